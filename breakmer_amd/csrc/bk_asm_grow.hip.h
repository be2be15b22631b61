// bk_asm_grow.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// check_alt_reads (:568-582), finalize (:584-599), contig.grow (:616-649), the record of a kept contig (init_assembly :53-59, set_kmer_locs :434-438) and setup_contigs (:11-26).
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 11      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

// ---- check_alt_reads (sv_assembly.py:568-582) + the adds of finalize (:590-592) -------------------------
BK_COLD void bk_check_alt_reads()
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const int k = C_.k;
    const int nalt = S->nalt;
    if (nalt == 0) return;
    BK_SYNC();
    if (BK_TID == 0) S->tmp2 = ++S->stamp_ctr;          // identifies mer_set of this call
    BK_SYNC();
    const int fin = S->tmp2;
    int *tmp = (int *)L_CAND;
    for (int a = 0; a < nalt; a++) {
        const int u = (int)C_.altl[a];
        bk_load_read(u);
        const int len = S->rlen, np = len - k;
        // x = get_read_kmers(read) - used_mers - mer_set   (set(self.kmers) holds tuples: removes nothing)
        BkKey best; best.hi = ~0ull; best.lo = ~0ull; int bestrk = -1; int anyx = 0;
        for (int x = BK_TID; x < np; x += BK_AT) {
            BkKey key; uint32_t st = 0; int rk = bk_bytes_kmer(L_RSEQ + x, k, key) ? bk_lookup_state(key, st) : -1;
            if (rk >= 0 && (st != BK_K_LIVE || C_.kstamp[3 * rk + 1] == fin)) rk = -1;
            tmp[x] = rk;
            if (rk >= 0) { anyx = 1; if (C_.kcnt[rk] > 1 && key_lt(key, best)) { best = key; bestrk = rk; } }   // sorted(x) (P2): smallest mer with count > 1
        }
        // reduce the minimum key over the block
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long oh = __shfl_xor(best.hi, o), ol = __shfl_xor(best.lo, o); int ork = __shfl_xor(bestrk, o);
            BkKey ob; ob.hi = oh; ob.lo = ol;
            if (ork >= 0 && (bestrk < 0 || key_lt(ob, best))) { best = ob; bestrk = ork; }
        }
        unsigned long long *red = (unsigned long long *)(tmp + 2 * C_.MAXCAND - 64);   // tail of the scratch: (hi, lo, rk) per wavefront (8 x 3 x 8 B <= 256 B)
        BK_SYNC();
        if ((BK_TID & 63) == 0) { int w = BK_TID >> 6; red[3 * w] = best.hi; red[3 * w + 1] = best.lo; red[3 * w + 2] = (unsigned long long)(long long)bestrk; }
        BK_SYNC();
        { best.hi = red[0]; best.lo = red[1]; bestrk = (int)(long long)red[2];
          for (int w = 1; w < BK_AT / 64; w++) { BkKey ob; ob.hi = red[3 * w]; ob.lo = red[3 * w + 1]; int ork = (int)(long long)red[3 * w + 2];
              if (ork >= 0 && (bestrk < 0 || key_lt(ob, best))) { best = ob; bestrk = ork; } } }
        BK_SYNC();
        (void)anyx;
        if (bestrk >= 0) {
            for (int x = BK_TID; x < np; x += BK_AT) if (tmp[x] >= 0) C_.kstamp[3 * tmp[x] + 1] = fin;       // mer_set = mer_set | x
            if (BK_TID == 0) bk_fifo_push(bestrk, u);
        }
        BK_SYNC();
    }
    BK_ACC(6);
}

// ---- finalize (sv_assembly.py:584-599) ----------------------------------------------------------------
__device__ inline void bk_finalize(bool setup)
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    if (setup) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);                 // set_kmers(akmers.smers_set)
    bk_check_alt_reads();
    BK_SYNC();
    if (BK_TID == 0) {
        if (!S->founder_added) {                                                  // batch_reads[0] = founder, aligned (:383)
            S->founder_added = 1; int u = S->founder;
            if (C_.ureads[u] != S->serial) { C_.ureads[u] = S->serial; C_.readl[S->nr++] = (uint32_t)u; }
        }
        S->nalt = 0;
    }
    BK_SYNC();
    BK_ACC(10);
}

// ---- contig.grow (sv_assembly.py:616-649) --------------------------------------------------------------
__device__ __forceinline__ void bk_grow()
{
    BK_ACC(S_->ctx); BK_CTX(15);
    BkAsmShared *S = S_;
    if (!S->setup) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);
    for (;;) {
        if (S->status) return;
        // refresh_kmers :601-602 -> snapshot list
        const int nk = S->nk, k0 = S->kscan;
        uint32_t T;
        // Every k-mer of a snapshot is in checked_kmers when its visits are over, and the list only grows at its end (a
        // replacement, set_kmers, starts it anew): the next snapshot can only hold what was appended since.  With sequencing
        // noise that is a handful of k-mers after every extension, a snapshot per round -- one wavefront takes them, the
        // candidate-less ones marked in the same pass (below), one barrier instead of seven.
        const bool small = nk - k0 <= 64;
        if (small) {
            if ((BK_TID >> 6) == 0) {
                const int idx = k0 + BK_TID;
                uint32_t en = idx < nk ? C_.klist[idx] : 0u;
                const bool unchecked = idx < nk && C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial;
                const unsigned long long bm = __ballot(unchecked);
                if (unchecked) {
                    const int rank = (int)(en & 0x3FFFFFFFu);
                    const uint32_t pb = C_.poff[rank], pe = C_.poff[rank + 1];
                    bool has = pe - pb > 16u;
                    for (uint32_t i = pb; !has && i < pe; i++) { const uint32_t u = C_.post[i] >> 10; has = !(C_.ufl[u] & BK_R_DELETED) && C_.ubuf[u] != S->serial; }
                    if (!has) { C_.kstamp[3 * rank] = S->serial; en |= 0x40000000u; }
                    C_.nklist[__popcll(bm & ((1ull << BK_TID) - 1ull))] = en;
                }
                if (BK_TID == 0) {
#ifdef BK_PHASE_STAMPS
                    S->acc[20] += 1; S->acc[22] += (unsigned long long)(nk - k0);
#endif
                    S->tmp0 = __popcll(bm); S->kscan = nk;
                    if (bm) {
                        if (!S->founder_added) { S->founder_added = 1; const int fu = S->founder; if (C_.ureads[fu] != S->serial) { C_.ureads[fu] = S->serial; C_.readl[S->nr++] = (uint32_t)fu; } }
                        if (S->plan_kind == 0) S->plan_ok = 0;
                    }
                }
            }
            BK_SYNC();
            T = (uint32_t)S->tmp0;
            BK_ACC(13);
            if (T == 0) break;
        } else {
        const int chunk = (nk + BK_AT - 1) / BK_AT, b = BK_TID * chunk, e = min(nk, b + chunk);
        uint32_t cnt = 0;
        if (chunk == 1) {                                   // the usual case: an entry per thread, looked at once
            uint32_t en = 0;
            if (b < e) { en = C_.klist[b]; cnt = C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial; }
            const uint32_t pre = bk_scan256(cnt, S->scan, &T);
            if (cnt) C_.nklist[pre] = en;
        } else {
        for (int t = b; t < e; t++) cnt += C_.kstamp[3 * (C_.klist[t] & 0x7FFFFFFFu)] != S->serial;
        uint32_t pre = bk_scan256(cnt, S->scan, &T);
        for (int t = b; t < e; t++) { uint32_t en = C_.klist[t]; if (C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial) C_.nklist[pre++] = en; }
        }
#ifdef BK_PHASE_STAMPS
        if (BK_TID == 0) { S->acc[21] += 1; S->acc[23] += (unsigned long long)nk; }
#endif
        if (BK_TID == 0) S->kscan = nk;
        BK_SYNC();
        BK_ACC(13);
        if (T == 0) break;
        // Visits without any candidate read.  Within one contig the candidate set of a k-mer only shrinks (reads get
        // buffered or deleted, never the reverse), so a k-mer whose short posting list holds no eligible read now has
        // none when its turn comes, and such a visit does nothing but mark the k-mer checked and, AT ITS TURN (the
        // used set is read by check_alt_reads of the visits before it), used (get_mer_reads :604-614 returns []).
        // With sequencing noise that is the majority of all visits: they are found here for the whole snapshot at once
        // (bit 30 of the entry), and the loop below retires whole runs of them with one wavefront.
        // The founder read joins the read list first, as the first finalize would do (:383).
        if (BK_TID == 0 && !S->founder_added) {
            S->founder_added = 1; const int fu = S->founder;
            if (C_.ureads[fu] != S->serial) { C_.ureads[fu] = S->serial; C_.readl[S->nr++] = (uint32_t)fu; }
        }
        for (uint32_t t = BK_TID; t < T; t += BK_AT) {
            const uint32_t en = C_.nklist[t]; int rank = (int)(en & 0x3FFFFFFFu);
            if (!BK_CHK((uint32_t)rank < C_.M, 7, ((unsigned long long)t << 32) | en)) rank = 0;
            const uint32_t pb = C_.poff[rank], pe = C_.poff[rank + 1];
            bool has = pe - pb > 16u;                                             // long lists take the ordinary visit
            for (uint32_t i = pb; !has && i < pe; i++) { const uint32_t u = C_.post[i] >> 10; has = !(C_.ufl[u] & BK_R_DELETED) && C_.ubuf[u] != S->serial; }
            if (!has) { C_.kstamp[3 * rank] = S->serial; C_.nklist[t] = en | 0x40000000u; }       // checked_kmers is only read by the next snapshot
        }
        BK_SYNC();
        if (BK_TID == 0 && S->plan_kind == 0) S->plan_ok = 0;                      // a plan of visits refers to one snapshot (and one contig)
        BK_SYNC();
        }
        uint32_t t = 0, en_next = C_.nklist[0];
        while (t < T) {
            if (S->status) return;
            const uint32_t en = en_next; int rank = (int)(en & 0x3FFFFFFFu); const bool rev = (en >> 31) != 0;
            if (!BK_CHK((uint32_t)rank < C_.M, 6, ((unsigned long long)t << 32) | en)) { S->status = S->status ? S->status : BK_ST_UNSPLIT; BK_SYNC(); return; }
            if (en & 0x40000000u) {
                // a run of candidate-less visits, in order: used_mers.add(mer) for each k-mer not yet in it
                if ((BK_TID >> 6) == 0) {
                    const uint32_t idx = t + BK_TID;
                    const uint32_t e2 = idx < T ? C_.nklist[idx] : 0u;
                    const unsigned long long bm = __ballot((e2 & 0x40000000u) != 0);
                    const int run = ~bm ? __ffsll((long long)~bm) - 1 : 64;                  // entries t .. t+run-1 are candidate-less
                    const bool in_run = BK_TID < run;
                    const int rk = (int)(e2 & 0x3FFFFFFFu);
                    bool ap = in_run && C_.kstate[in_run ? rk : 0] == BK_K_LIVE;
                    for (int j = 0; j < run; j++) { const int rj = __builtin_amdgcn_readlane(rk, j); if (j < BK_TID && rj == rk) ap = false; }   // listed twice: once
                    const unsigned long long am = __ballot(ap);
                    if (ap) { C_.kstate[rk] = BK_K_USED; C_.usedl[S->nused + __popcll(am & ((1ull << BK_TID) - 1ull))] = (uint32_t)rk; }
                    if (BK_TID == 0) { S->nused += __popcll(am); S->tmp1 = run; }
                }
                BK_SYNC();
                t += (uint32_t)S->tmp1;
                if (t < T) en_next = C_.nklist[t];
                BK_SYNC();
                continue;
            }
            const int vt = (int)t;
            t++;
            if (t < T) en_next = C_.nklist[t];                                     // fetched a whole visit ahead of its use
            bk_find_reads(rank, rev, true);                                    // get_mer_reads :604-614
            if (BK_TID == 0) bk_add_used_mer(rank);
            BK_SYNC();
            // position of this k-mer in the contig (prediction seed); -1 disables the look-ahead
            if ((BK_TID >> 6) == 0) {
                int pc = -1;
                // (a single candidate needs no prediction unless the round may go on into the following visits)
                if (!bk_expect_reject() && (S->ncand >= 2 || (S->ncand == 1 && S->la_pause == 0 && !(C_.flags & BK_F_NO_XVISIT)))) { BkKey key; key.hi = C_.khi[rank]; key.lo = C_.klo[rank]; pc = bk_find_kmer_wave(L_CSEQ + S->cbase, S->clen, key, C_.k); }
                if (BK_TID == 0) S->pc = pc;
            }
            BK_SYNC();
            BK_ACC(14);
            bk_run_candidates(rank, 0, S->ncand, true, vt, (int)T);
            bk_finalize(false);
            if (BK_TID == 0) C_.kstamp[3 * rank] = S->serial;                       // checked_kmers.append(mer): read by the next snapshot only
        }
        BK_SYNC();                                                                  // the stamps above are visible before the next snapshot reads them
    }
    BK_ACC(15); BK_CTX(0);
}

// ---- init_assembly keeps a contig iff support >= rc_thresh and len > read_len (sv_assembly.py:53-59);
//      set_kmer_locs (:434-438) and the record the host reads back ------------------------------------------
BK_COLD void bk_emit_contig()
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const int total = bk_total_reads();
    if (total < C_.rc_thresh || S->clen <= (int)C_.max_len) return;
    const int k = C_.k, len = S->clen, nlen = S->nlen, nk = S->nk, nr = S->nr;
    const uint32_t o_seq = (uint32_t)sizeof(BkContigRec), o_io = (uint32_t)bk_align_up(o_seq + len, 8), o_ot = o_io + 4u * nlen,
                   o_kl = o_ot + 4u * nlen, o_km = (uint32_t)bk_align_up(o_kl + 4u * len, 8), o_rd = o_km + 16u * nk, size = (uint32_t)bk_align_up(o_rd + 4u * nr, 8);
    BK_SYNC();
    if (BK_TID == 0) {
        uint64_t need = bk_align_up(size, 256);
        uint64_t off = atomicAdd(C_.out_top, (unsigned long long)need);
        // The result arena is full: the region goes on WITHOUT writing records (offset 0 = none; its status says so at once) -- the bump pointer
        // then ends at what the whole batch needs and the host grows the arena ONCE (round 6; until then a region stopped at its first
        // record that did not fit, the demand was unknown, and a noisy batch on a fresh handle grew 8 -> 33 -> 134 -> 536 MB in three re-runs)
        if (off + need > C_.out_cap) { off = 0; atomicCAS((int *)&C_.wk->status, BK_ST_OK, BK_ST_OUT); }
        S->scan[8] = (uint32_t)off; S->scan[9] = (uint32_t)(off >> 32);
    }
    BK_SYNC();
    const uint64_t off = ((uint64_t)S->scan[9] << 32) | S->scan[8];
    if (off == 0) return;                                 // offset 0 is reserved (out_top starts at 256)
    uint8_t *rec = C_.out + off;
    BkContigRec *h = (BkContigRec *)rec;
    char *oseq = (char *)(rec + o_seq); int32_t *oio = (int32_t *)(rec + o_io), *oot = (int32_t *)(rec + o_ot), *okl = (int32_t *)(rec + o_kl);
    uint64_t *okm = (uint64_t *)(rec + o_km); uint32_t *ord_ = (uint32_t *)(rec + o_rd);
    const uint8_t *cs = L_CSEQ + S->cbase;
    const int32_t *io = bk_cnt_io(S->cbuf) + S->nbase, *ot = bk_cnt_ot(S->cbuf) + S->nbase;
    for (int t = BK_TID; t < len; t += BK_AT) { oseq[t] = "ACGTN"[cs[t]]; okl[t] = 0; }
    for (int t = BK_TID; t < nlen; t += BK_AT) { oio[t] = io[t]; oot[t] = ot[t]; }
    for (int t = BK_TID; t < nk; t += BK_AT) { uint32_t rk = C_.klist[t] & 0x7FFFFFFFu; if (!BK_CHK(rk < C_.M, 4, ((unsigned long long)t << 32) | C_.klist[t])) rk = 0; okm[2 * t] = C_.klo[rk]; okm[2 * t + 1] = C_.khi[rk]; }
    for (int t = BK_TID; t < nr; t += BK_AT) ord_[t] = C_.urep[C_.readl[t]];
    // first occurrence of every sample k-mer in the contig (str.find over all len-k+1 positions)
    // (split regions: the stamps of other units' k-mers are left alone -- only the contig's own k-mers are read back below, and
    // those all belong to components this unit holds)
    for (int x = BK_TID; x + k <= len; x += BK_AT) { BkKey key; int rk = bk_bytes_kmer(cs + x, k, key) ? bk_lookup(key) : -1; if (rk >= 0 && (!C_.own || (C_.kroot[rk] != BK_EMPTY32 && bk_acc_mine(C_.kroot[rk])))) atomicMin(&C_.kstamp[3 * rk + 2], x); }
    BK_SYNC();
    for (int t = BK_TID; t < nk; t += BK_AT) {
        if (!BK_CHK((C_.klist[t] & 0x7FFFFFFFu) < C_.M, 5, C_.klist[t])) continue;
        int pos = C_.kstamp[3 * (C_.klist[t] & 0x7FFFFFFFu) + 2];
        if (pos == 0x7FFFFFFF) continue;                 // find() == -1: the python slice [-1:k-1] is empty for len >= k
        for (int q = pos; q < min(pos + k, len); q++) atomicAdd(&okl[q], 1);
    }
    BK_SYNC();
    for (int x = BK_TID; x + k <= len; x += BK_AT) { BkKey key; int rk = bk_bytes_kmer(cs + x, k, key) ? bk_lookup(key) : -1; if (rk >= 0 && (!C_.own || (C_.kroot[rk] != BK_EMPTY32 && bk_acc_mine(C_.kroot[rk])))) C_.kstamp[3 * rk + 2] = 0x7FFFFFFF; }
    if (BK_TID == 0) {
        h->root = S->ccomp; h->pass = C_.want;
        h->next = 0; h->hits_off = 0; h->seq_len = len; h->counts_len = nlen; h->n_kmers = nk; h->n_reads = nr; h->total_reads = total; h->n_hits = 0;
        h->o_seq = o_seq; h->o_io = o_io; h->o_ot = o_ot; h->o_klocs = o_kl; h->o_kmers = o_km; h->o_reads = o_rd; h->n_sec = 0; h->size = size;
        if (C_.split) {       // the units of a region emit side by side: (order key, record) pairs, ordered and linked by bk_link_kernel
            const uint32_t at = atomicAdd(&C_.wk->n_cidx, 1u);
            if (at < C_.wk->cidx_cap) { C_.cidx_key[at] = ((unsigned long long)(uint32_t)S->seed_rank << 20) | (unsigned long long)(uint32_t)min(S->emit_seq, 0xFFFFF); C_.cidx_key[C_.wk->cidx_cap + at] = off; }
            S->emit_seq++;
        } else {
            if (C_.wk->o_first_contig == 0) C_.wk->o_first_contig = off; else ((BkContigRec *)(C_.out + C_.wk->o_last_contig))->next = off;
            C_.wk->o_last_contig = off;
        }
        S->n_contigs++;
        const unsigned long long ci = atomicAdd(C_.n_clist, 1ull);              // work list of the realign stage (one workgroup per contig)
        if (ci < C_.clist_cap) C_.clist[ci] = off | ((unsigned long long)C_.region << 40);
    }
    BK_SYNC();
    BK_ACC(7);
}

// ---- setup_contigs (sv_assembly.py:11-26) -----------------------------------------------------------------
__device__ inline void bk_setup_contigs(int rank)
{
    BK_ACC(S_->ctx); BK_CTX(8);
    BkAsmShared *S = S_;
    bk_find_reads(rank, false, false);                                         // used_reads = set()
    if (BK_TID == 0) bk_add_used_mer(rank);
    BK_SYNC();
    const int n = S->ncand;
    if (n == 0 || S->status) return;
    // the candidate list must survive the check_read calls below: candu is not touched by them
    const int u0 = (int)(L_CANDU[0] & 0x3FFFFFu);
    const bool in_fifo = C_.ufound[u0] < 0 && !(C_.ufl[u0] & BK_R_USED);            // buff.add_contig :337-340
    BK_SYNC();
    bk_contig_new(rank, u0, in_fifo);
    if (BK_TID == 0 && in_fifo) C_.ufl[u0] |= BK_R_USED;
    BK_SYNC();
    if (BK_TID == 0) S->pc = (int)(L_CANDU[0] >> 22);      // the contig IS the first read: the k-mer sits where it sits in that read
    BK_SYNC();
    bk_run_candidates(rank, 1, n, false, -1, 0);
    bk_finalize(true);
    if (in_fifo) { bk_grow(); if (!S->status) bk_emit_contig(); }                // it is the FIFO head (:50-52)
    BK_ACC(8); BK_CTX(0);
}
