// bk_asm_round.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// the run retire (the slots of a round retired together), the plan of a round (bk_plan_round) and the candidate loop of setup_contigs / grow with its speculative look-ahead (bk_run_candidates).
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 10      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

// Run retire (BK_F_NO_RUN_RETIRE switches it off).  The slots of a round are retired together for as long as every one of
// them does what was predicted: the decisions of slots sl .. s1-1 are taken at once (a lane each, against the contig length
// each was aligned to), thread 0 walks the prediction chain over them -- a rejected, identical or contained read leaves the
// sequence alone, an extension (contig_overlap_read :520-527, read_overlap_contig :538-545) must be the predicted kind and
// length, which makes the bytes the predicted ones -- and the run is applied in ONE pass: the appended / prepended bases, the
// count entries of the new bases (assigned, extend_counts :201-221), then one summed count update per base over the ranges
// of all reads of the run in the coordinates each range was taken in (set_counts :195-199; old coordinates before a
// prepend), one lane per read for the per-read words, appends in slot order.  Left to the one-read path (bk_retire): a read
// that replaces the contig (superseq), a tie of the two scores (k-mer position rule), the first read of a seed planned
// ahead, an extension that fails a bound, and in grow mode an extension by two or more bases (it adds contig k-mers, in
// order, bk_kmers_ordered; an extension by one base adds none, Q1).  Four barriers per run instead of two to three per read.
// The number retired is left in S->tmp0 (uniform after the function's last barrier).
__device__ __forceinline__ void bk_retire_run(int sl, int s1, bool grow)
{
    BkAsmShared *S = S_;
    const int nbt = s1 - sl;
    const int cbase0 = S->cbase, clen0 = S->clen, nbase0 = S->nbase, nlen0 = S->nlen;      // read by every thread before thread 0 replaces them (two barriers on)
    // wavefront 0, a lane per slot: the decision, whether the slot does what it may do here, whether the contig it was aligned
    // against is what its predecessor leaves behind; the run = the slots before the first lane that says no.  The per-read
    // words the bookkeeping at the end needs are asked for now (nobody writes them in between).
    int my_u = 0, my_dec = BK_DEC_NONE, my_found = -1, my_ur = 0; uint32_t my_fl = 0, my_kc = 0; unsigned long long my_cells = 0;
    if ((BK_TID >> 6) == 0) {
        const int j = BK_TID;
        const bool have = j < nbt;
        bool ok = false; int dec = BK_DEC_NONE, pl = 0, a0 = 0, a1 = 0, cb_after = 0, cl_after = 0, pb = 0, plen = 0;
        if (have) {
            BkAsmShared::Slot &t = S->slot[sl + j];
            my_u = t.u; my_fl = C_.ufl[my_u]; my_ur = C_.ureads[my_u]; my_kc = C_.kcnt[t.rank]; if (grow) my_found = C_.ufound[my_u];
            pb = t.pb; plen = t.plen;
            int ds, de; bool tie;
            dec = bk_decide(t.v1, t.v2, plen, t.rl, ds, de, tie);
            if (dec == BK_DEC_POST) { pl = max(t.rl - t.v1.i_end, 0); ds = t.v1.j_start; de = plen; }
            else if (dec == BK_DEC_PRE) { pl = t.v2.j_start; ds = t.v2.i_start; de = t.v2.i_end; }
            const int nb_ = nbase0 - (cbase0 - pb), nl_ = nlen0 + (plen - clen0);          // the count vectors move with the contig
            ok = !t.first && !tie && dec != BK_DEC_SUPER && !(grow && pl >= 2);
            if (dec == BK_DEC_POST) ok = ok && !(pb + plen + pl > 2 * C_.MAXC || plen + pl > C_.MAXC || nb_ + nl_ + pl > 2 * C_.MAXC);
            else if (dec == BK_DEC_PRE) ok = ok && !(pb - pl < 0 || plen + pl > C_.MAXC || nb_ - pl < 0);
            if (dec != BK_DEC_NONE && dec != BK_DEC_SAME) { a0 = nb_ + ds; a1 = max(nb_ + min(de, nl_), a0); }      // absolute span of the count update
            cb_after = dec == BK_DEC_PRE ? pb - pl : pb; cl_after = (dec == BK_DEC_PRE || dec == BK_DEC_POST) ? plen + pl : plen;
            my_dec = dec; my_cells = 2ull * (unsigned long long)plen * (unsigned long long)t.rl;
            if (ok) { t.dec = dec; t.ds = a0; t.de = a1; t.hasn = pl; }      // hasn: only read when the slot is staged; from here on the length of the extension
        }
        // the contig this slot was aligned against = what the slot before it leaves (slot sl: the contig as it is)
        int pcb = __shfl_up(cb_after, 1), pcl = __shfl_up(cl_after, 1), pdec = __shfl_up(dec, 1);
        if (j == 0) { pcb = cbase0; pcl = clen0; pdec = S->last_dec; }
        if (have) {
            bool chain = pcb == pb && pcl == plen;
            if (sl + j > 0) {
                const int pk = S->slot[sl + j - 1].kind;
                chain = chain && ((pk == BK_PK_PRE && pdec == BK_DEC_PRE) || (pk == BK_PK_POST && pdec == BK_DEC_POST) ||
                                  (pk == BK_PK_SAME && (pdec == BK_DEC_NONE || pdec == BK_DEC_SAME || pdec == BK_DEC_SUB)));
            }
            ok = ok && chain;
        }
        const unsigned long long bad = ~__ballot(ok);
        const int m = bad ? __ffsll((long long)bad) - 1 : 64;              // lanes >= nbt say no: m <= nbt
        // span of all count updates of the run, geometry after its last slot
        int lo = (j < m && a1 > a0) ? a0 : 0x7FFFFFFF, hi = (j < m && a1 > a0) ? a1 : 0;
        for (int o = 1; o < 16; o <<= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }      // BK_SPEC <= 16 slots
        if (j == 0) { S->tmp0 = m; S->tmp1 = hi > 0 ? lo : 0; S->tmp2 = hi; }
        if (m > 0 && j == m - 1) { S->dstart = cb_after; S->dend = cl_after; }
    }
    BK_SYNC();
    const int m = S->tmp0;
    if (m == 0) return;
    const int lo = S->tmp1, hi = S->tmp2, cbase1 = S->dstart, clen1 = S->dend;
    int32_t *io = bk_cnt_io(S->cbuf), *ot = bk_cnt_ot(S->cbuf);
    // new bases and their count entries: a wavefront per slot (the bases of all slots but the round's last were pre-written by the staging; the same bytes)
    for (int j = BK_TID >> 6; j < m; j += BK_AT / 64) {
        const BkAsmShared::Slot &q = S->slot[sl + j];
        const int pl = q.hasn;
        if ((q.dec != BK_DEC_POST && q.dec != BK_DEC_PRE) || pl == 0) continue;
        const int nb_ = nbase0 - (cbase0 - q.pb), nl_ = nlen0 + (q.plen - clen0);
        const uint8_t *rs = L_RSEQ_S(sl + j);
        const int cat_io = q.rindel ? q.rn : 0, cat_ot = q.rindel ? 0 : q.rn;
        if (q.dec == BK_DEC_POST) {
            for (int t = BK_TID & 63; t < pl; t += 64) { L_CSEQ[q.pb + q.plen + t] = rs[q.rl - pl + t]; io[nb_ + nl_ + t] = cat_io; ot[nb_ + nl_ + t] = cat_ot; }
        } else {
            for (int t = BK_TID & 63; t < pl; t += 64) { L_CSEQ[q.pb - pl + t] = rs[t]; io[nb_ - pl + t] = cat_io; ot[nb_ - pl + t] = cat_ot; }
        }
    }
    BK_SYNC();
    for (int t = lo + BK_TID; t < hi; t += BK_AT) {
        int a = 0, b = 0;
        for (int j = 0; j < m; j++) {
            const BkAsmShared::Slot &q = S->slot[sl + j];
            if (t >= q.ds && t < q.de) { if (q.rindel) a += q.rn; else b += q.rn; }      // empty span for a read that changes no count
        }
        if (a) io[t] += a;
        if (b) ot[t] += b;
    }
    if ((BK_TID >> 6) == 0) {
        const int j = BK_TID;
        const bool have = j < m;
        const int u = my_u, dec = my_dec;
        const bool match = have && dec != BK_DEC_NONE;
        const bool to_list = match && my_ur != S->serial;
        const bool to_alt = have && !match && my_kc > 2 && !(my_fl & BK_R_USED);
        const unsigned long long lm = __ballot(to_list), am = __ballot(to_alt), mm = __ballot(match), below = (1ull << j) - 1ull;
        if (have) {
            C_.ubuf[u] = S->serial;
            if (match) {
                C_.ufl[u] = (uint8_t)(my_fl | BK_R_USED);
                if (to_list) { C_.ureads[u] = S->serial; C_.readl[S->nr + __popcll(lm & below)] = (uint32_t)u; }
                if (grow && my_found >= 0 && BK_CHK((uint32_t)my_found <= C_.U, 11, my_found)) { C_.pend[2 * my_found] = BK_EMPTY32; C_.ufound[u] = -1; }
            } else if (to_alt) {
                const int at = S->nalt + __popcll(am & below);
                if (at < C_.MAXCAND) C_.altl[at] = (uint32_t)u; else S->status = BK_ST_CAND;
            } else C_.ufl[u] = (uint8_t)(my_fl | BK_R_DELETED);
        }
        unsigned long long cells = have ? my_cells : 0ull;
        for (int o = 1; o < 16; o <<= 1) cells += __shfl_xor(cells, o);
        if (BK_TID == 0) {
            S->nr += __popcll(lm); S->nalt = min(S->nalt + __popcll(am), C_.MAXCAND);
            int na = S->n_acc, nr_ = S->n_rej;
            for (int i = 0; i < m; i++) { if ((mm >> i) & 1ull) na++; else nr_++; if (na + nr_ >= 64) { na >>= 1; nr_ >>= 1; } }
            S->n_acc = na; S->n_rej = nr_;
            S->cells += cells; S->calls += 2 * m;
            const bool lastm = ((mm >> (m - 1)) & 1ull) != 0;
            S->last_dec = S->slot[sl + m - 1].dec; S->hit = lastm ? 1 : 0;
            S->pc += cbase0 - cbase1;                                       // the recruiting k-mer moves with what was prepended
            S->nbase = nbase0 - (cbase0 - cbase1); S->nlen = nlen0 + (clen1 - clen0); S->cbase = cbase1; S->clen = clen1;
#ifdef BK_PHASE_STAMPS
            S->acc[17] += m;
#endif
        }
    }
    BK_SYNC();
}

// retire slot sl if the contig is what the slot was aligned against; uniform result: 1 retired, 0 prediction failed
__device__ __forceinline__ int bk_retire_checked(int sl, bool grow)
{
    BkAsmShared *S = S_;
    if (S->slot[sl].first) {                        // first read of a seed planned ahead: the contig is its founder, nothing came before
        if (S->cbase != S->slot[sl].pb || S->clen != S->slot[sl].plen) return 0;
    } else if (sl > 0) {
        const int pk = S->slot[sl - 1].kind, ld = S->last_dec;
        const bool kind_ok = (pk == BK_PK_PRE && ld == BK_DEC_PRE) || (pk == BK_PK_POST && ld == BK_DEC_POST) ||
                             (pk == BK_PK_SAME && (ld == BK_DEC_NONE || ld == BK_DEC_SAME || ld == BK_DEC_SUB));
        if (!kind_ok || S->cbase != S->slot[sl].pb || S->clen != S->slot[sl].plen) return 0;
    }
    (void)bk_retire(S->slot[sl].rank, sl, grow);    // ends with a barrier; the FIFO entry of a matched read is dropped in its bookkeeping
#ifdef BK_PHASE_STAMPS
    if (BK_TID == 0) S->acc[17] += 1;
#endif
    return 1;
}

// one step of the prediction chain on VALUES (bk_predict below is the same step on a slot in LDS): the read whose recruiting k-mer sits at pos of its rl
// bases is aligned against the contig [pb, pb + plen) in which that k-mer sits at ppc; kind / amt = what it is predicted to do to it
__device__ inline bool bk_predict_step(int pos, int rl, bool xrej, int maxc, int &pb, int &plen, int &ppc, int lo, int hi, int &kind, int &amt)
{
    amt = 0;
    if (xrej) { kind = BK_PK_SAME; return true; }
    const int left = pos - ppc, right = (rl - pos) - (plen - ppc);
    if (ppc < 0 || (left > 0 && right > 0)) { kind = BK_PK_STOP; return false; }
    if (left > 0) { kind = BK_PK_PRE; amt = left; if (pb - left < lo || plen + left > maxc) { kind = BK_PK_STOP; return false; } pb -= left; plen += left; ppc += left; }
    else if (right > 0) { kind = BK_PK_POST; amt = right; if (pb + plen + right > hi || plen + right > maxc) { kind = BK_PK_STOP; return false; } plen += right; }
    else kind = BK_PK_SAME;
    return true;
}

// the plan of one round (the FIRST WAVEFRONT): prediction chain over this visit's slots, then over the following visits' lists.  Out of
// line: its loops over slots and lists would otherwise sit in the register budget of the state machine's hot loop.
// Until round 6 thread 0 ran it alone, word by word in LDS: every slot it wrote made the compiler re-read the region's state (the writes may
// alias it), every list entry cost a dependent LDS round trip per slot already planned -- 5.2 us per round of the headline's 20 rounds per
// region, 11 % of the assembler (profiles/r06/headline_plan_phases.txt).  Now slot s of the round lives in LANE s (read, k-mer position,
// length), entry i of a look-ahead list in lane i: the chain itself stays serial, but on registers (v_readlane of the lane whose turn it
// is, the contig geometry in scalars), "is this read already in a slot" is one compare per slot for the whole list at once, and every
// lane writes the slot it ended up with.  The plan only decides what a round SPECULATES on; what it retires is checked against the contig
// as it is then (bk_retire_checked), so the results cannot depend on it.
#define BK_SWEEP_RETRY 8
BK_COLD void bk_plan_round(int q, int n, int nbmax, int cap, int vt, int la, int la_on)
{
    BkAsmShared *S = S_;
    if (BK_TID >= 64) return;
    const int lane = BK_TID;
#define BK_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
    const int cbase = BK_UNI(S->cbase), maxc = BK_UNI(C_.MAXC), flags = BK_UNI(C_.flags);
    const bool xrej = BK_UNI(bk_expect_reject()) != 0;
    int pb = cbase, plen = BK_UNI(S->clen), ppc = BK_UNI(S->pc), nb = 0, mx = 0;
    bool go = true;
    // does this round run the score sweep first (bk_nw.hip.h)?  Decided before the slots are planned: it sets how many slots a round over a
    // LONG contig may have (below)
    int dp_n = BK_UNI(S->dp_n), dp_redo = BK_UNI(S->dp_redo);
    if (dp_n >= 64) {
        if (lane == 0) { S->dp_tot += dp_n - (dp_n >> 1); S->dp_rtot += dp_redo - (dp_redo >> 1); S->dp_n = dp_n >> 1; S->dp_redo = dp_redo >> 1; }
        dp_n >>= 1; dp_redo >>= 1;
    }
    // Off is not for good (round 6): while the sweep is off nothing feeds its window, so until now one bad stretch -- the reads of an indel
    // against the first contigs of a region -- switched it off for the rest of the region (a noisy region with 27,000 reads swept 1,650 of
    // them and ran full DPs, four times the instructions per cell, for the rest: profiles/r06/noisy_sweep_retry.txt).  After BK_SWEEP_RETRY
    // rounds without it the window goes to the totals and starts afresh; a stretch that still needs the full DPs switches it off again
    // within five to seventeen reads.
    if (BK_SWEEP_OFF(dp_redo, dp_n)) {
        int off = BK_UNI(S->dp_off) + 1;
        if (off >= BK_SWEEP_RETRY) {
            if (lane == 0) { S->dp_tot += dp_n; S->dp_rtot += dp_redo; S->dp_n = 0; S->dp_redo = 0; }
            dp_n = 0; dp_redo = 0; off = 0;
        }
        if (lane == 0) S->dp_off = off;
    }
    const bool fast = !(flags & BK_F_NO_SCORE_SWEEP) && (!BK_SWEEP_OFF(dp_redo, dp_n) || (flags & BK_F_FORCE_REDO));      // (the diagnostic that flags every read keeps it on)
    // Slots of a round whose (predicted) contig is longer than the dual / pair kernels take (BK_NW_DUAL_COLS).  The full overlap DPs of
    // such a contig need TWO wavefronts per read (bk_nw_suffix + bk_nw_wave): half as many slots as wavefronts.  The score sweep
    // (bk_nw_score_long: column tiles, any length) needs ONE -- until round 5 the second wavefront of each slot idled through it, so a
    // round over a 650-column contig aligned 2 reads on the 4 wavefronts of the throughput build (configs[4]: 19,300 rounds of 2.5 reads
    // per region, the chain that bounds it; configs[3]'s translocation contigs of 1,650 columns alike).  Round 6: a wavefront per slot
    // while the sweep is on; the few reads it cannot settle are swept in full afterwards, BK_SPEC_WIDE at a time (bk_dp_redo).
    const int widecap = fast ? BK_WAVES : BK_SPEC_WIDE;
    // this visit's slots (staged by bk_run_candidates): lane s holds slot s
    int my_u = -1, my_pos = 0, my_rl = 0;
    if (lane < nbmax) { my_u = S->slot[lane].u; my_pos = S->slot[lane].pos; my_rl = S->slot[lane].rl; }
    int t_pb = 0, t_plen = 0, t_kind = 0, t_amt = 0;
    // both DPs of a slot run on one wavefront while the (predicted) contig fits its columns; else a wavefront (score sweep) or two per slot
    for (int sl = 0; sl < nbmax && go; sl++) {
        if (sl >= widecap && plen > BK_NW_DUAL_COLS) { go = false; break; }
        nb = sl + 1;
        const int pos = __builtin_amdgcn_readlane(my_pos, sl), rl = __builtin_amdgcn_readlane(my_rl, sl), pb0 = pb, plen0 = plen;
        int kind, amt;
        go = bk_predict_step(pos, rl, xrej, maxc, pb, plen, ppc, 0, 2 * maxc, kind, amt);
        if (lane == sl) { t_pb = pb0; t_plen = plen0; t_kind = kind; t_amt = amt; }
        mx = max(mx, plen0);
    }
    if (lane < nb) { BkAsmShared::Slot &t = S->slot[lane]; t.pb = t_pb; t.plen = t_plen; t.kind = t_kind; t.amt = t_amt; }
    const int ncur = nb, mxcur = mx;
    int upto = vt;
#ifdef BK_PHASE_STAMPS
    if (lane == 0) {
        if (la_on) S->acc[19] += 1ull;                                               // rounds with free slots
        if (la_on && !(la && go && q + ncur == n)) S->acc[19] += 1ull << 16;          // ... not looked ahead (paused / STOP / round does not finish the visit)
        else if (la_on && S->la_n[0] < 0) S->acc[19] += 1ull << 32;                  // ... next visit: none in the window or posting list too long
        else if (la_on && S->la_n[0] == 0) S->acc[19] += 1ull << 48;                 // ... next visit has no eligible read
    }
#endif
    if (la && vt >= 0 && go && q + ncur == n) {
        for (int w = 0; w < BK_AT / 64 && go; w++) {
            const int cn = BK_UNI(S->la_n[w]);
            if (cn < 0) break;
            // the visit's list once the reads planned before it are in the buffer: entry i in lane i
            const uint32_t e_cu = lane < cn ? BK_LA_CU(w)[lane] : 0u;
            const int e_u = (int)(e_cu & 0x3FFFFFu), e_pos = (int)(e_cu >> 22);
            bool inflight = false;
            for (int s2 = 0; s2 < nb; s2++) inflight = inflight || __builtin_amdgcn_readlane(my_u, s2) == e_u;
            const unsigned long long km = __ballot(lane < cn && !inflight);
            const int keep = __popcll(km);
            if (nb + keep > cap) break;                             // does not fit
            const int pc2 = BK_UNI(S->la_pc[w]);
            if (keep > 0 && pc2 < 0 && !xrej) break;
            ppc = pc2 + (cbase - pb);                       // the predicted contig starts cbase - pb bases before the current one
            const int nb0 = nb, mx0 = mx, lt = BK_UNI(S->la_t[w]), lrank = BK_UNI(S->la_rank[w]);
            int e_rl = 0, e_rn = 0, e_fl = 0;
            if (lane < cn) { e_rl = BK_LA_RL(w)[lane]; e_rn = BK_LA_RN(w)[lane]; e_fl = BK_LA_FL(w)[lane]; }
            int e_slot = -1;
            unsigned long long rest = km;
            while (rest && go) {
                const int i = (int)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                if (nb >= widecap && plen > BK_NW_DUAL_COLS) { go = false; break; }
                const int pos = __builtin_amdgcn_readlane(e_pos, i), rl = __builtin_amdgcn_readlane(e_rl, i), u = __builtin_amdgcn_readlane(e_u, i), pb0 = pb, plen0 = plen;
                int kind, amt;
                go = bk_predict_step(pos, rl, xrej, maxc, pb, plen, ppc, 0, 2 * maxc, kind, amt);
                if (lane == i) { e_slot = nb; t_pb = pb0; t_plen = plen0; t_kind = kind; t_amt = amt; }
                if (lane == nb) my_u = u;
                mx = max(mx, plen0);
                nb++;
            }
            if (nb - nb0 != keep) { nb = nb0; mx = mx0; break; }      // a visit is planned whole or not at all
            if (e_slot >= 0) {
                BkAsmShared::Slot &t = S->slot[e_slot];
                t.u = e_u; t.pos = e_pos; t.rl = e_rl; t.rn = e_rn; t.rindel = (e_fl & BK_R_INDEL) ? 1 : 0;
                t.hasn = (C_.n_nlist && (e_fl & BK_R_HASN)) ? 1 : 0;
                t.vt = lt; t.rank = lrank; t.fu = -1; t.first = 0;
                t.pb = t_pb; t.plen = t_plen; t.kind = t_kind; t.amt = t_amt;
            }
            upto = lt;
        }
    }
    int kind = 0;
    if (la && vt < 0 && q + ncur == n) {
        // the first round of the seeds that follow: a group per seed, aligned against its founder in a strip of its own
        // (lane 0, slot by slot in LDS as before: no "already in a slot" test here, and a seed's list is two or three reads)
        kind = 1;
        int nbc = nb, mxc = mx;
        if (lane == 0) {
            int g = 0;
            for (int w = 0; w < BK_AT / 64; w++) {
                const int cn = S->la_n[w];
                if (cn < 0) break;
                if (cn < 2) continue;                                   // a seed with its founder only has no DP
                if (nbc + cn - 1 > cap) break;
                const uint32_t fcu = BK_LA_CU(w)[0];
                const int base = BK_SEEDBUF(g) + C_.MAXR + 16, lo = BK_SEEDBUF(g), hi = BK_SEEDBUF(g) + 3 * (C_.MAXR + 16);
                int pb2 = base, plen2 = BK_LA_RL(w)[0], ppc2 = (int)(fcu >> 22);          // the contig IS the founder; the k-mer sits where it sits in that read
                const int nb0 = nbc, mx0 = mxc; bool whole = true;
                for (int i = 1; i < cn; i++) {
                    const uint32_t cu = BK_LA_CU(w)[i];
                    if (nbc >= widecap && plen2 > BK_NW_DUAL_COLS) { whole = false; break; }
                    BkAsmShared::Slot &t = S->slot[nbc];
                    const int fl = BK_LA_FL(w)[i];
                    t.u = (int)(cu & 0x3FFFFFu); t.pos = (int)(cu >> 22); t.rl = BK_LA_RL(w)[i]; t.rn = BK_LA_RN(w)[i]; t.rindel = (fl & BK_R_INDEL) ? 1 : 0;
                    t.hasn = (C_.n_nlist && (fl & BK_R_HASN)) ? 1 : 0;
                    t.vt = S->la_t[w]; t.rank = S->la_rank[w]; t.fu = (int)(fcu & 0x3FFFFFu); t.first = (i == 1) ? 1 : 0;
                    nbc++;
                    mxc = max(mxc, plen2);
                    if (!bk_predict(t, pb2, plen2, ppc2, lo, hi) && i + 1 < cn) { whole = false; break; }
                }
                if (!whole) { nbc = nb0; mxc = mx0; continue; }             // a seed is planned whole or not at all
                g++;
            }
        }
        nb = BK_UNI(nbc); mx = BK_UNI(mxc);
    }
    if (kind == 1 && mx > BK_NW_TILE_COLS) { nb = ncur; mx = mxcur; }      // a multi-tile DP would use the scratch the strips sit in
    // both DPs of a slot on one wavefront when more than BK_SPEC_WIDE reads are in the round; with fewer, the idle
    // wavefronts take the second DP (two 64-lane sweeps finish sooner than one half-wave pair)
    const bool dual = mx <= BK_NW_DUAL_COLS && !(flags & BK_F_NO_DUAL) && (nb > BK_SPEC_WIDE || (flags & BK_F_DUAL_ALWAYS));
    int nc = ncur;
    if (!dual && nb > widecap) { nc = min(nc, widecap); nb = nc; upto = vt; }      // a wavefront (score sweep) or two (full DPs) per slot: this visit's reads only
    if (lane == 0) {
        S->fast = fast ? 1 : 0; S->dual = dual ? 1 : 0;
        S->nb = nb; S->ncur = nc;
        S->plan_r = nc; S->plan_upto = upto; S->plan_ok = (nb > nc || upto > vt) ? 1 : 0; S->plan_kind = kind;
        int planned = S->la_planned + nb - nc, pause = S->la_pause;
        if (la_on && pause > 0) pause--;
        if (planned >= 64) {                        // one window: did the slots planned for later visits get used?
            if (4 * S->la_adopted < planned) { pause = S->la_backoff; S->la_backoff = min(2 * S->la_backoff, 4096); }
            else S->la_backoff = 32;
            planned = 0; S->la_adopted = 0;
        }
        S->la_planned = planned; S->la_pause = pause;
    }
#undef BK_UNI
}

// vt / T: index of this visit in the snapshot and the snapshot's length (grow); vt < 0: setup_contigs
__device__ __forceinline__ void bk_run_candidates(int rank, int first, int n, bool grow, int vt, int T)
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6;
    int q = first;
    // 0. a plan of an earlier round covers this visit: adopt its slots if it predicted exactly this candidate list
    //    (uniform: every thread evaluates the same LDS words)
    bool adopt = false;
    int r0 = 0;
    if (vt >= 0 && S->plan_ok && S->plan_kind == 0) {
        r0 = S->plan_r;
        adopt = vt <= S->plan_upto;
        int g = 0;
        if (adopt) { while (r0 + g < S->nb && S->slot[r0 + g].vt == vt) g++; adopt = g == n; }
        for (int i = 0; adopt && i < n; i++) { const uint32_t cu = L_CANDU[i]; adopt = S->slot[r0 + i].u == (int)(cu & 0x3FFFFFu) && S->slot[r0 + i].pos == (int)(cu >> 22); }
        if (!adopt || n == 0) {
            BK_SYNC();
            if (BK_TID == 0) { if (!adopt) S->plan_ok = 0; else if (r0 >= S->nb && vt >= S->plan_upto) S->plan_ok = 0; }
            BK_SYNC();
            adopt = false;
        }
    }
    if (vt == -1 && S->plan_ok && S->plan_kind == 1) {
        // a setup round planned the first round of this seed ahead: same founder, same candidates -> adopt
        const int nbp = S->nb;
        int r = S->plan_r;
        while (r < nbp && -2 - S->slot[r].vt < rank) r++;          // seeds that never came up (their k-mer was used up meanwhile)
        int g = 0; bool ad = false;
        if (r < nbp && -2 - S->slot[r].vt == rank) {
            while (r + g < nbp && S->slot[r + g].vt == S->slot[r].vt) g++;
            ad = n >= 2 && g == n - 1 && S->slot[r].first && S->slot[r].fu == (int)(L_CANDU[0] & 0x3FFFFFu);
            for (int i = 0; ad && i < g; i++) { const uint32_t cu = L_CANDU[1 + i]; ad = S->slot[r + i].u == (int)(cu & 0x3FFFFFu) && S->slot[r + i].pos == (int)(cu >> 22); }
        }
        BK_SYNC();
        if (BK_TID == 0) {
            if (ad) { const int delta = S->cbase - S->slot[r].pb; for (int i = 0; i < g; i++) S->slot[r + i].pb += delta; S->plan_r = r; }      // strip -> contig deque coordinates
            else { S->plan_r = r + g; if (S->plan_r >= nbp) S->plan_ok = 0; }
        }
        BK_SYNC();
        if (ad) { adopt = true; r0 = r; }
    }
    // one loop for both kinds of pass, so that the decision/apply step (bk_retire) is inlined once: a pass either retires
    // the adopted slots of this visit or plans, aligns and retires a fresh round
    while (adopt || q < n) {
        int s0 = r0, s1 = r0 + (n - q);             // the adopted slots of this visit / seed ...
        if (!adopt) {                               // ... or a fresh round:
        if (S->status) return;
        BK_ACC(S_->ctx);
        // slots of a round: one per wavefront while both DPs of a slot fit one wavefront (contig <= BK_NW_DUAL_COLS), else half
        // (a long contig: a slot per wavefront while the score sweep is on, bk_plan_round -- the same predicate on the same words, which
        //  nobody writes between the last barrier and the plan; the plan's own verdict is what the round runs with)
        const bool sweep_on = !(C_.flags & BK_F_NO_SCORE_SWEEP) && (!BK_SWEEP_OFF(S->dp_redo, S->dp_n) || (C_.flags & BK_F_FORCE_REDO));
        const int cap = (C_.flags & (BK_F_NO_DUAL | BK_F_SPEC4)) ? BK_SPEC_WIDE : S->clen > BK_NW_DUAL_COLS ? (sweep_on ? BK_WAVES : BK_SPEC_WIDE) : BK_SPEC;
        const int nbmax = min((C_.flags & (BK_F_NO_DUAL | BK_F_SPEC4)) ? BK_SPEC_WIDE : BK_SPEC, n - q);
        // look into the following visits when this one leaves slots free (needs the scratch for BK_AT/64 lists of 64 reads)
        // (split regions, bk_comp.hip.h: until round 4 the look-ahead was off inside them -- with it the assembler faulted about once
        //  in 25 runs of a 64-region noisy batch.  The causes were two missing barriers (bk_retire's bookkeeping, round 4; the
        //  S->foreign loop of bk_kmers_ordered, round 5), not the plans; BK_F_SPLIT_NO_LOOKAHEAD is the round-4 setting.)
        const bool la_on = n - q < cap && !(C_.flags & (vt >= 0 ? BK_F_NO_XVISIT : BK_F_NO_XSEED)) && 2 * C_.MAXCAND >= 4 * BK_AT && !(C_.split && (C_.flags & BK_F_SPLIT_NO_LOOKAHEAD));
        const bool la = la_on && S->la_pause == 0;
        // 1. stage the reads of this round (one lane per slot fetches the read's metadata), plan the predictions
        BK_SYNC();
        if (BK_TID < nbmax) {
            const uint32_t cu = L_CANDU[q + BK_TID]; const int u = (int)(cu & 0x3FFFFFu);
            const uint32_t ri = C_.urep[u];
            BkAsmShared::Slot &t = S->slot[BK_TID];
            t.u = u; t.pos = (int)(cu >> 22); t.rl = C_.rlen[ri]; t.rn = (int)C_.unr[u]; t.rindel = (C_.ufl[u] & BK_R_INDEL) ? 1 : 0;
            t.hasn = (C_.n_nlist && (C_.ufl[u] & BK_R_HASN)) ? 1 : 0;
            t.vt = vt; t.rank = rank; t.fu = -1; t.first = 0;
        }
        if (la) { if (vt >= 0) bk_lookahead_wave(wv, vt, T); else bk_seedahead_wave(wv, rank); }
        BK_SYNC();
        bk_plan_round(q, n, nbmax, cap, vt, la ? 1 : 0, la_on ? 1 : 0);
        // The reads of THIS visit's slots do not depend on the plan (staged above: slot s = candidate q + s): while the first wavefront plans, the
        // others unpack them -- their global round trips used to start after the plan's barrier (round 6).  What the plan adds (slots of later
        // visits / seeds, the bytes a slot is predicted to add to the contig) follows below as before.
        if (BK_WAVES > 1 && wv > 0) {
            const int ln = BK_TID & 63;
            for (int sl = wv - 1; sl < nbmax; sl += BK_WAVES - 1) {
                const int rl = S->slot[sl].rl; uint8_t *rs = L_RSEQ_S(sl);
                const uint32_t *w = C_.reads + (uint64_t)C_.urep[S->slot[sl].u] * C_.read_words;
                for (int t = ln; t < rl; t += 64) rs[t] = (uint8_t)seq_base(w, t);
            }
        }
        BK_SYNC();
        const int nb = S->nb, ncur = S->ncur;
        const int nstaged = BK_WAVES > 1 ? min(nbmax, ncur) : 0;      // slots [0, nstaged) are unpacked already (slots from ncur on hold other reads)
#ifdef BK_PHASE_STAMPS
        if (BK_TID == 0) { S->acc[16] += nb; S->acc[18] += 1; }
#endif
        {   // unpack the reads; pre-write the bytes slot sl is predicted to add.  One wavefront per slot: the global loads
            // of all slots are in flight together (slot after slot they were nb dependent round trips per round)
            const int ln = BK_TID & 63;
            for (int sl = BK_TID >> 6; sl < nb; sl += BK_WAVES) {
                const int rl = S->slot[sl].rl; uint8_t *rs = L_RSEQ_S(sl);
                const bool staged = sl < nstaged;
                if (staged) {                           // its bytes are in LDS (written before the barrier): the predicted contig bytes come from there
                    if (sl + 1 < nb) {
                        const int amt = S->slot[sl].amt, pb = S->slot[sl].pb, plen = S->slot[sl].plen;
                        if (S->slot[sl].kind == BK_PK_PRE) for (int t = ln; t < amt; t += 64) L_CSEQ[pb - amt + t] = rs[t];
                        else if (S->slot[sl].kind == BK_PK_POST) for (int t = ln; t < amt; t += 64) L_CSEQ[pb + plen + t] = rs[rl - amt + t];
                    }
                    continue;
                }
                const uint32_t *w = C_.reads + (uint64_t)C_.urep[S->slot[sl].u] * C_.read_words;
                for (int t = ln; t < rl; t += 64) rs[t] = (uint8_t)seq_base(w, t);
                if (S->slot[sl].first) {                 // founder of a seed planned ahead: the contig its reads are aligned against
                    const uint32_t fri = C_.urep[S->slot[sl].fu];
                    const uint32_t *fw = C_.reads + (uint64_t)fri * C_.read_words;
                    uint8_t *fs = L_CSEQ + S->slot[sl].pb; const int fl2 = S->slot[sl].plen;
                    for (int t = ln; t < fl2; t += 64) fs[t] = (uint8_t)seq_base(fw, t);
                    if (C_.n_nlist && (C_.ufl[S->slot[sl].fu] & BK_R_HASN)) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                        if (ln == 0) bk_patch_n(fri, fs, 0, fl2);
                    }
                }
                if (sl + 1 < nb) {
                    const int amt = S->slot[sl].amt, pb = S->slot[sl].pb, plen = S->slot[sl].plen;
                    if (S->slot[sl].kind == BK_PK_PRE) for (int t = ln; t < amt; t += 64) L_CSEQ[pb - amt + t] = (uint8_t)seq_base(w, t);
                    else if (S->slot[sl].kind == BK_PK_POST) for (int t = ln; t < amt; t += 64) L_CSEQ[pb + plen + t] = (uint8_t)seq_base(w, rl - amt + t);
                }
            }
        }
        BK_SYNC();
        if (C_.n_nlist) {                                 // reads with N calls (rare): code 4 over the unpacked bytes and over the predicted contig bytes
            bool any = false;
            for (int sl = 0; sl < nb; sl++) any = any || S->slot[sl].hasn;
            if (any) {
                if (BK_TID < nb && S->slot[BK_TID].hasn) {
                    const BkAsmShared::Slot &t = S->slot[BK_TID]; const uint32_t ri = C_.urep[t.u];
                    bk_patch_n(ri, L_RSEQ_S(BK_TID), 0, t.rl);
                    if (BK_TID + 1 < nb) {
                        if (t.kind == BK_PK_PRE) bk_patch_n(ri, L_CSEQ + t.pb - t.amt, 0, t.amt);
                        else if (t.kind == BK_PK_POST) bk_patch_n(ri, L_CSEQ + t.pb + t.plen, t.rl - t.amt, t.amt);
                    }
                }
                BK_SYNC();
            }
        }
        BK_ACC(1);
        // 2. the overlap DPs (:451-452) of every slot of this round
        bk_dp_round();
        BK_SYNC();
        if (!S->dual && S->fast) {                      // (uniform) a wavefront per slot ran the score sweep: did it leave a border cell open?
            int nflag = 0;
            for (int sl = 0; sl < nb; sl++) nflag += (S->slot[sl].v1.j_start == BK_NW_NEEDS_DP || S->slot[sl].v2.j_start == BK_NW_NEEDS_DP) ? 1 : 0;
            if (nflag) {
                BK_SYNC();                              // every wavefront has looked at the result words before they change
                if (BK_TID < nb) {
                    const int f = (S->slot[BK_TID].v1.j_start == BK_NW_NEEDS_DP || S->slot[BK_TID].v2.j_start == BK_NW_NEEDS_DP) ? 1 : 0;
                    S->slot[BK_TID].dec = f;
                    if (f) atomicAdd(&S->dp_redo, 1);
                }
                BK_SYNC();
                // the full overlap DPs of the flagged slots, two wavefronts each: BK_SPEC_WIDE slots per pass (nflag is uniform)
                for (int done = 0; done < nflag; done += BK_SPEC_WIDE) {
                    bk_dp_redo(done);
                    BK_SYNC();
                }
            }
        }
        BK_ACC(2);
        s0 = 0; s1 = ncur;
        }
        // 3. retire this visit's slots in order while the predictions hold; the slots of later visits wait for their turn
        int sl = s0;
        while (sl < s1) {
            if (S->status) return;
            // the reads that change nothing, together -- where there are runs of them: on clean data nearly every read
            // extends the contig by a base and the attempt only costs its two barriers (same-box A/B on the headline:
            // 1.82 ms with it always on, 1.70 ms without), so it waits until check_align has lately rejected most reads
            if (!(C_.flags & BK_F_NO_RUN_RETIRE) && (s1 - sl >= 2 || bk_expect_reject())) {
                bk_retire_run(sl, s1, grow);
                const int m = S->tmp0;
                sl += m; q += m;
                if (sl >= s1 || S->status) break;
            }
            if (!bk_retire_checked(sl, grow)) break;        // the read that extends / replaces the contig (or fails the prediction)
            sl++; q++;
        }
        if (adopt) {
            BK_SYNC();
            if (BK_TID == 0) {
                S->la_adopted += sl - s0;
                if (sl < s1) S->plan_ok = 0;
                else { S->plan_r = s1; if (S->plan_r >= S->nb && vt >= S->plan_upto) S->plan_ok = 0; }
            }
            BK_SYNC();
            adopt = false;
        } else if (sl < s1 && S->plan_ok) { BK_SYNC(); if (BK_TID == 0) S->plan_ok = 0; BK_SYNC(); }      // a prediction failed: what was planned behind it is void
    }
}
