// bk_asm_apply.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// contig life cycle (contig.__init__ :417-426, buffer.add_contig :337-340); check_align's verdict (bk_decide, :459-503) and its application to the contig for ONE read (bk_retire: check_read :552-566, contig_overlap_read :506-528, read_overlap_contig :530-546).
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 8      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

// ---- contig life cycle ------------------------------------------------------------------------------
__device__ inline void bk_contig_new(int rank, int u, bool in_fifo)                  // contig.__init__ :417-426
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    bk_load_read(u);
    const int len = S->rlen, nreads = S->rn, indel = S->rindel;
    if (len > C_.MAXC) { bk_fail(BK_ST_CONTIG); return; }
    const int base = C_.MAXC - len;
    for (int t = BK_TID; t < len; t += BK_AT) L_CSEQ[base + t] = L_RSEQ[t];
    int32_t *io = bk_cnt_io(0) + base, *ot = bk_cnt_ot(0) + base;
    for (int t = BK_TID; t < len; t += BK_AT) { io[t] = indel ? nreads : 0; ot[t] = indel ? 0 : nreads; }      // :162-165
    BK_SYNC();
    if (BK_TID == 0) {
        S->cbase = base; S->clen = len; S->nbase = base; S->nlen = len; S->cbuf = 0;
        S->serial = ++S->serial_ctr; S->setup = 0; S->founder = u; S->founder_added = 0; S->in_fifo = in_fifo ? 1 : 0;
        S->nk = 0; S->nr = 0; S->nalt = 0; S->kscan = 0;
        C_.kstamp[3 * rank] = S->serial;                 // checked_kmers = [kmer_val]
        C_.ubuf[u] = S->serial;                          // buffer = set([read.id])
    }
    BK_SYNC();
    BK_ACC(9);
}
__device__ inline void bk_fifo_push(int rank, int u)                                 // buffer.add_contig :337-340 (thread 0)
{
    BkAsmShared *S = S_;
    if (C_.ufound[u] >= 0 || (C_.ufl[u] & BK_R_USED)) return;
    C_.pend[2 * S->ptail] = (uint32_t)rank; C_.pend[2 * S->ptail + 1] = (uint32_t)u;
    C_.ufound[u] = S->ptail; S->ptail++; C_.ufl[u] |= BK_R_USED;
}
__device__ inline void bk_add_used_mer(int rank)                                     // thread 0
{
    if (C_.kstate[rank] == BK_K_LIVE) { C_.kstate[rank] = BK_K_USED; C_.usedl[S_->nused++] = (uint32_t)rank; }
}

// check_align's verdict on one read from its two overlap DPs (sv_assembly.py:459-503); tie: the k-mer position rule decides
__device__ inline int bk_decide(const BkNwResult &v1, const BkNwResult &v2, int clen, int rl, int &ds, int &de, bool &tie)
{
    int dec = BK_DEC_NONE; ds = 0; de = 0; tie = false;
    const int minlen = min(clen, rl);
    const bool ok1 = (4 * v1.score >= minlen) && (200 * v1.score >= 179 * (clen - v1.j_start));
    const bool ok2 = (4 * v2.score >= minlen) && (200 * v2.score >= 179 * (rl - v2.j_start));
    if (!ok1 && !ok2) dec = BK_DEC_NONE;
    else if (v1.score == v2.score && v1.j_start == 0 && v1.i_start == 0 && clen == rl) dec = BK_DEC_SAME;
    else if (v1.score == v2.score) {
        if (clen < rl || v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; }
        else if (rl < clen || v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; }
        else tie = true;
    } else if (v1.score > v2.score) {
        if (v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; } else dec = BK_DEC_POST;
    } else {
        if (v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; } else dec = BK_DEC_PRE;
    }
    return dec;
}

// ---- check_align (sv_assembly.py:449-504) + check_read (:552-566) -------------------------------------
// Decision and state update for the read in look-ahead slot `sl`, whose two overlap DPs (v1 = nw(contig, read),
// v2 = nw(read, contig)) were computed against the CURRENT contig.  `rank` = the k-mer that recruited the read.
// Returns (uniform) whether the read matched.
__device__ inline bool bk_retire(int rank, int sl, bool grow)
{
    BkAsmShared *S = S_;
    const int k = C_.k;
    const int u = S->slot[sl].u, rl = S->slot[sl].rl, nreads = S->slot[sl].rn, indel = S->slot[sl].rindel;
    const uint8_t *rseq = L_RSEQ_S(sl);
    const int clen = S->clen;
    const uint8_t *cs = L_CSEQ + S->cbase;
    const int wv = BK_TID >> 6;
    const BkNwResult v1 = S->slot[sl].v1, v2 = S->slot[sl].v2;
    // thread 0 asks now for the per-read words its bookkeeping at the end needs: nobody else writes them meanwhile, and the
    // round trip hides behind the count updates instead of standing between two barriers
    uint32_t pre_fl = 0, pre_kc = 0; int pre_ureads = 0, pre_found = -1;
    if (BK_TID == 0) { pre_fl = C_.ufl[u]; pre_ureads = C_.ureads[u]; pre_kc = C_.kcnt[rank]; if (grow) pre_found = C_.ufound[u]; }
    int ds = 0, de = 0;                             // uniform: computed identically by every thread
    bool tie = false;
    bool synced = false;                            // (uniform) a barrier has been passed since this function was entered
    int dec = bk_decide(v1, v2, clen, rl, ds, de, tie);
    if (tie) {
        synced = true;
        // k-mer position tie-break: x.replace('-','') of the aligned strings are the plain slices
        if (wv == 0) {
            BkKey key; key.hi = C_.khi[rank]; key.lo = C_.klo[rank];
            int i11 = bk_find_kmer_wave(cs + v1.j_start, clen - v1.j_start, key, k);
            int i12 = bk_find_kmer_wave(rseq + v1.i_start, v1.i_end - v1.i_start, key, k);
            int i21 = bk_find_kmer_wave(rseq + v2.j_start, rl - v2.j_start, key, k);
            int i22 = bk_find_kmer_wave(cs + v2.i_start, v2.i_end - v2.i_start, key, k);
            int d = BK_DEC_NONE;
            if (i11 > -1 && i12 > -1) { if ((i21 == -1 && i22 == -1) || (abs(i21 - i22) > abs(i11 - i12))) d = BK_DEC_POST; }
            else if (i21 > -1 && i22 > -1) { if ((i11 == -1 && i12 == -1) || (abs(i21 - i22) < abs(i11 - i12))) d = BK_DEC_PRE; }
            if (BK_TID == 0) S->dec = d;
        }
        BK_SYNC();
        dec = S->dec;
        // contig_overlap_read / read_overlap_contig re-test the containment case (:508, :531)
        if (dec == BK_DEC_POST && v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; }
        if (dec == BK_DEC_PRE && v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; }
        BK_SYNC();
    }
    // ---- apply: one fused phase (disjoint index ranges, scalar state committed by thread 0, one barrier) -------
    const bool match = dec != BK_DEC_NONE;
    bool ext = false;                                   // contig was extended (POST / PRE): new k-mers in grow mode
    if (dec == BK_DEC_SUPER) {                                                    // aseq.set_superseq :232-236 (rare: separate phases)
        synced = true;
        bk_counts_superseq(rl, nreads, indel, ds, de);
        const int base = C_.MAXC - rl;
        for (int t = BK_TID; t < rl; t += BK_AT) L_CSEQ[base + t] = rseq[t];
        BK_SYNC();
        if (BK_TID == 0) { S->cbase = base; S->clen = rl; S->pc = S->slot[sl].pos; }
        BK_SYNC();
        if (grow) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);              // set_kmers(skmers) :479/:515
    } else if (dec != BK_DEC_NONE && dec != BK_DEC_SAME) {
        const int cbase = S->cbase, nbase = S->nbase, nlen = S->nlen;
        int32_t *io = bk_cnt_io(S->cbuf), *ot = bk_cnt_ot(S->cbuf);
        int32_t *cv = (indel ? io : ot) + nbase;
        int s0 = ds, s1 = de, pl = 0, at = 0; bool fail = false;
        if (dec == BK_DEC_POST) {                                                 // :520-527, add_postseq :243-250
            pl = max(rl - v1.i_end, 0); s0 = v1.j_start; s1 = clen; at = nbase + nlen;
            fail = cbase + clen + pl > 2 * C_.MAXC || clen + pl > C_.MAXC || at + pl > 2 * C_.MAXC;
            if (!fail) for (int t = BK_TID; t < pl; t += BK_AT) L_CSEQ[cbase + clen + t] = rseq[v1.i_end + t];
        } else if (dec == BK_DEC_PRE) {                                           // :538-545, add_preseq :255-262
            pl = v2.j_start; s0 = v2.i_start; s1 = v2.i_end; at = nbase - pl;
            fail = cbase - pl < 0 || clen + pl > C_.MAXC || at < 0;
            if (!fail) for (int t = BK_TID; t < pl; t += BK_AT) L_CSEQ[cbase - pl + t] = rseq[t];
        }
        if (fail) { bk_fail(BK_ST_CONTIG); synced = true; }
        else {
            for (int t = s0 + BK_TID; t < min(s1, nlen); t += BK_AT) cv[t] += nreads;          // set_counts :195-199 (old coordinates)
            for (int t = BK_TID; t < pl; t += BK_AT) { io[at + t] = indel ? nreads : 0; ot[at + t] = indel ? 0 : nreads; }   // extend_counts :201-221
            // Every wavefront has read the scalars above (cbase, nbase, clen, nlen) before thread 0 replaces them: without
            // this barrier a wavefront that is late into this function (starved by co-resident workgroups) reads the
            // NEW base and updates a range shifted by the prepended length.
            BK_SYNC();
            synced = true;
            if (BK_TID == 0 && dec != BK_DEC_SUB) {
                if (dec == BK_DEC_PRE) { S->cbase = cbase - pl; S->nbase = nbase - pl; S->pc += pl; }
                S->clen = clen + pl; S->nlen = nlen + pl;
            }
            ext = dec != BK_DEC_SUB;
        }
    }
    // check_read bookkeeping (:552-565).  It replaces words the CALLER's control flow reads right before this call, on every wavefront
    // for itself (S->last_dec in bk_retire_checked, the acceptance counters in bk_expect_reject: which retire path is taken, whether the
    // prediction held): a wavefront that is late into this call must have read them before thread 0 writes -- a read that changes
    // nothing (dec NONE / SAME) passes no barrier on its way here, and a late wavefront then took another path than the others,
    // with other barriers (found with a sleeping wavefront behind every barrier, -DBK_JITTER: wrong fixtures at any load; in the
    // field: faults and hangs once several noisy workgroups share a CU).
    if (!synced) BK_SYNC();
    if (BK_TID == 0) {
        S->last_dec = dec; S->hit = match ? 1 : 0;
        if (match) S->n_acc++; else S->n_rej++;
        if (S->n_acc + S->n_rej >= 64) { S->n_acc >>= 1; S->n_rej >>= 1; }
        C_.ubuf[u] = S->serial;                                                    // self.buffer.add(read.id)
        S->cells += 2ull * (unsigned long long)clen * (unsigned long long)rl; S->calls += 2;
        if (match) {
            C_.ufl[u] = (uint8_t)(pre_fl | BK_R_USED);
            if (pre_ureads != S->serial) { C_.ureads[u] = S->serial; C_.readl[S->nr++] = (uint32_t)u; }
            if (grow && pre_found >= 0 && BK_CHK((uint32_t)pre_found <= C_.U, 10, pre_found)) { C_.pend[2 * pre_found] = BK_EMPTY32; C_.ufound[u] = -1; }       // buff.remove_contig :638-639
        } else if (pre_kc > 2 && !(pre_fl & BK_R_USED)) {
            if (S->nalt < C_.MAXCAND) C_.altl[S->nalt++] = (uint32_t)u; else S->status = BK_ST_CAND;
        } else C_.ufl[u] = (uint8_t)(pre_fl | BK_R_DELETED);                       // rb.delete -> rb.clean :390
    }
    BK_SYNC();
    BK_ACC(3);
    if (ext && grow && !S->status) {
        const int k1 = k - 1;
        if (dec == BK_DEC_POST) { const int from = max(clen - k1, 0); bk_kmers_ordered(S->cbase + from, (clen - from) + (S->clen - clen), BK_ORD_FOR); }
        else bk_kmers_ordered(S->cbase, (S->clen - clen) + min(k1, clen), BK_ORD_REV);
    }
    return match;
}
