// bk_asm_plan.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// the overlap DPs of one look-ahead round (dispatch to bk_nw.hip.h); the look-ahead lists of the following k-mer visits / the following seeds; one step of the prediction chain.
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 9      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

// ---- the read loop of setup_contigs (:16-23) / grow (:634-639) with speculative look-ahead --------------------
// The reference checks the candidate reads strictly one after the other: each accepted read changes the contig
// the next one is aligned to.  The alignment of read q+1 only depends on the contig SEQUENCE after read q, and
// that is predictable from where the recruiting k-mer sits in the read and in the contig (the read sticks out
// `pos - pc` bases to the left, or its tail beyond the contig end).  So each round aligns up to BK_SPEC reads
// at once -- slot s against the contig predicted after slots 0..s-1, two wavefronts per slot -- and then retires
// them in order, for as long as the contig really became what was predicted (same kind of change, same
// geometry => same bytes); the first misprediction discards the later slots, which are redone next round.
// Results are therefore bit-identical to the serial loop; only the DP latency chain gets shorter.
enum { BK_PK_SAME = 0, BK_PK_PRE = 1, BK_PK_POST = 2, BK_PK_STOP = 3 };
// The overlap DPs of one look-ahead round.  Everything it needs is in LDS (slots, contig deque, staged reads); it is kept
// OUT of line so that the dozens of DP variants it dispatches to (one function per column count) have ONE call site
// whose live state is nothing: inlined into the state machine they made the allocator spill around every variant.
__device__ __noinline__ void bk_dp_round()
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6, nb = S->nb;
    if (S->dual && nb > BK_WAVES) {                      // more reads than wavefronts (BK_PAIR builds): one score matrix per read, wavefront w takes slots 2w, 2w+1 (bk_nw_pair)
        const int a = 2 * wv, b = 2 * wv + 1;
        if (a < nb) {
            BkPairArgs A, B;
            A.contig = BK_O_CSEQ + S->slot[a].pb; A.clen = S->slot[a].plen; A.read = BK_O_RSEQ + a * (C_.MAXR + 16); A.n = S->slot[a].rl; A.res = (int)((uint8_t *)&S->slot[a].v1 - bk_lds);
            if (b < nb) { B.contig = BK_O_CSEQ + S->slot[b].pb; B.clen = S->slot[b].plen; B.read = BK_O_RSEQ + b * (C_.MAXR + 16); B.n = S->slot[b].rl; B.res = (int)((uint8_t *)&S->slot[b].v1 - bk_lds); }
            else { B.contig = 0; B.clen = 0; B.read = 0; B.n = 0; B.res = 0; }
            if (S->fast) {
                // the score sweep first: end cells and scores of both calls; the border cells follow without a traceback for overlaps
                // without a mismatch or an indel (bk_nw.hip.h).  A read it cannot settle is swept again in full -- with its partner
                const int nrd = b < nb ? 2 : 1;
                bk_nw_score_pair(A, B);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                bool redo = S->slot[a].v1.j_start == BK_NW_NEEDS_DP || S->slot[a].v2.j_start == BK_NW_NEEDS_DP;
                if (b < nb) redo = redo || S->slot[b].v1.j_start == BK_NW_NEEDS_DP || S->slot[b].v2.j_start == BK_NW_NEEDS_DP;
                redo = __builtin_amdgcn_readfirstlane((int)redo) != 0;
                if ((BK_TID & 63) == 0) { atomicAdd(&S->dp_n, nrd); if (redo) atomicAdd(&S->dp_redo, nrd); }
                if (redo) bk_nw_pair(A, B);
            } else bk_nw_pair(A, B);
        }
    } else if (S->dual) {                                // both DPs of slot wv on this wavefront
        if (wv < nb) {
            const int contig = BK_O_CSEQ + S->slot[wv].pb, clen = S->slot[wv].plen, rd = BK_O_RSEQ + wv * (C_.MAXR + 16), rl = S->slot[wv].rl, res = (int)((uint8_t *)&S->slot[wv].v1 - bk_lds);
            if (S->fast) {
                BkPairArgs A; A.contig = contig; A.clen = clen; A.read = rd; A.n = rl; A.res = res;
                bk_nw_score_one(A);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool redo = __builtin_amdgcn_readfirstlane((int)(S->slot[wv].v1.j_start == BK_NW_NEEDS_DP || S->slot[wv].v2.j_start == BK_NW_NEEDS_DP)) != 0;
                if ((BK_TID & 63) == 0) { atomicAdd(&S->dp_n, 1); if (redo) atomicAdd(&S->dp_redo, 1); }
                if (redo) bk_nw_dual(contig, clen, rd, rl, res);
            } else bk_nw_dual(contig, clen, rd, rl, res);
        }
    } else if (S->fast) {
        // A contig beyond the dual / pair kernels' columns, score sweep on: the sweep of the whole matrix (any contig length: registers up to
        // 640 columns, column tiles beyond) takes ONE wavefront, so every wavefront has a slot of its own (round 6; bk_plan_round: until
        // round 5 a slot had two wavefronts and the second one idled here).  What the sweep cannot settle is swept in full after the
        // round's barrier, two wavefronts per read (bk_dp_redo).
        // (a tile pipeline over two wavefronts of one slot -- alternate column tiles, the second ~128 steps behind on the edge column in
        //  LDS -- was built and measured in round 5: bit-exact, and no faster where long contigs occur: configs[4] 6,367 -> 6,394 ms per
        //  batch, configs[3] 945 -> 953; not kept: profiles/r05/score_sweep_ab.txt.  It shortened ONE read's sweep; this doubles the reads.)
        if (wv < nb) {
            bk_nw_score_long(BK_O_CSEQ + S->slot[wv].pb, S->slot[wv].plen, BK_O_RSEQ + wv * (C_.MAXR + 16), S->slot[wv].rl, (int)((uint8_t *)&S->slot[wv].v1 - bk_lds), L_BOUND_W(wv), (C_.flags & BK_F_FORCE_REDO) ? 2 : 0);
            if ((BK_TID & 63) == 0) atomicAdd(&S->dp_n, 1);
        }
    } else {                                             // two wavefronts per slot, both with the contig on the tile columns
        const int sl = wv >> 1;
        if (sl < nb) {
            const int cl = S->slot[sl].plen, rl = S->slot[sl].rl;
            {
                const uint8_t *cs = L_CSEQ + S->slot[sl].pb;
                // waves w and w+4 land on the same SIMD: give it one direct (heavier) and one transposed sweep
                if ((((wv & 1) ^ (wv >> 2)) & 1) == 0) { BkNwResult r = bk_nw_suffix(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v1 = r; }
                else { BkNwResult r = bk_nw_wave<true>(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v2 = r; }
            }
        }
    }
}
// the slots of a long-contig round whose score sweep left a border cell open: both overlap DPs in full, two wavefronts per slot -- wavefront
// pair p takes the (skip + p)-th flagged slot (the caller passes over the flagged slots BK_SPEC_WIDE at a time)
__device__ __noinline__ void bk_dp_redo(int skip)
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6;
    int sl = -1, want = skip + (wv >> 1);
    for (int s2 = 0; s2 < S->nb; s2++) if (S->slot[s2].dec && want-- == 0) { sl = s2; break; }      // (Slot::dec is free between the staging of a round and its retirement: here it says "sweep again")
    if (sl >= 0) {
        const uint8_t *cs = L_CSEQ + S->slot[sl].pb; const int cl = S->slot[sl].plen, rl = S->slot[sl].rl;
        if ((((wv & 1) ^ (wv >> 2)) & 1) == 0) { BkNwResult r = bk_nw_suffix(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v1 = r; }
        else { BkNwResult r = bk_nw_wave<true>(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v2 = r; }
    }
}
// The score sweep (bk_nw.hip.h: 3 instructions per cell) runs first while at most HALF of the reads of its window had to be swept again in full (12 per
// cell): a pair of reads is swept again when either is flagged, so at a read rate p the cost is 3 + 12 (1 - (1 - p)^2) against 12 -- it pays up
// to p = 0.5; a wide round (a wavefront per slot, then passes of BK_SPEC_WIDE flagged slots on two wavefronts each) breaks even there too.
// (a quarter until round 6, from the days of the 6-instruction cell.)  redo / n: S->dp_redo / S->dp_n, the window that is halved at 64.
#ifndef BK_SWEEP_OFF
#define BK_SWEEP_OFF(redo, n) (2 * (redo) > (n) + 8)
#endif
// noisy reads: check_align has lately rejected three reads out of four (prediction then is "nothing changes", bk_predict)
__device__ inline bool bk_expect_reject() { return S_->n_rej >= 24 && S_->n_rej >= 3 * S_->n_acc; }

// Look-ahead across the k-mer visits of grow.  A visit usually recruits only a handful of reads (clean data) or a single
// one (sequencing noise), far fewer than there are look-ahead slots.  The visits of a snapshot are known in advance
// (nklist), and so is the candidate list of a LATER visit: every read a visit looks at ends up in the contig's buffer
// (check_read :552 buffer.add, matched or not), so the list a later visit will see is its eligible reads now minus the
// reads of the visits in between -- which are exactly the slots planned before it.  A round therefore goes on planning
// into the following visits until the slots are full; their DPs run in the same round against the predicted contig and
// their results wait in the slots.  When such a visit comes up, its real candidate list (the ordinary find_reads) is
// compared with the planned one: equal -> the slots are retired in order under the usual prediction checks, no DP;
// anything else -> the plan is dropped and the visit runs as before.  State only ever changes in bk_retire / finalize,
// in the reference's order.
//
// One wavefront per following visit: its eligible reads (short posting lists only), ordered as find_reads orders them,
// with the per-read fields a slot needs, and the position of its k-mer in the current contig.
#define BK_LA_CH 16                                  // posting lists of up to 64 * BK_LA_CH entries are looked into
#define BK_LA_CU(w) ((uint32_t *)L_CAND + (w) * 64)
#define BK_LA_RL(w) ((int *)L_CAND + (BK_AT / 64 + (w)) * 64)
#define BK_LA_RN(w) ((int *)L_CAND + (2 * (BK_AT / 64) + (w)) * 64)
#define BK_LA_FL(w) ((int *)L_CAND + (3 * (BK_AT / 64) + (w)) * 64)
BK_COLD void bk_lookahead_wave(int w, int vt, int T)
{
    BkAsmShared *S = S_;
    const int lane = BK_TID & 63;
    const int idx = vt + 1 + lane;
    const uint32_t en = idx < T ? C_.nklist[idx] : 0x40000000u;
    unsigned long long m = __ballot(!(en & 0x40000000u));                      // visits that may have candidates, in order
    for (int i = 0; i < w; i++) m &= m - 1;
    int cnt = -1, tt = -1, rank2 = 0, pc2 = -1;
    if (m) {
        const int bit = __ffsll((long long)m) - 1;
        const uint32_t e2 = (uint32_t)__shfl((int)en, bit);
        tt = vt + 1 + bit; rank2 = (int)(e2 & 0x3FFFFFFFu);
        const bool rev = (e2 >> 31) != 0;
        const uint32_t b = C_.poff[rank2], e = C_.poff[rank2 + 1];
        if (e - b <= 64u * BK_LA_CH) {
            // eligible entries of the posting list (a k-mer of a deep region sits in a few hundred reads, nearly all of them
            // in the buffer already): BK_LA_CH entries per lane, loads issued together, compacted into the list
            const int np = (int)(e - b);
            uint32_t pe[BK_LA_CH]; uint32_t fl[BK_LA_CH]; int bs[BK_LA_CH];
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) pe[c] = lane + 64 * c < np ? C_.post[b + lane + 64 * c] : 0u;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) { const bool have = lane + 64 * c < np; fl[c] = have ? C_.ufl[pe[c] >> 10] : 0u; bs[c] = have ? C_.ubuf[pe[c] >> 10] : 0; }
            int nv = 0;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) {
                const bool okc = lane + 64 * c < np && !(fl[c] & BK_R_DELETED) && bs[c] != S->serial;
                const unsigned long long vm = __ballot(okc);
                const int at = nv + __popcll(vm & ((1ull << lane) - 1ull));
                if (okc && at < 64) { BK_LA_RL(w)[at] = (int)pe[c]; BK_LA_FL(w)[at] = (int)fl[c]; }      // staging: overwritten by the ordered list below
                nv += __popcll(vm);
            }
            if (nv <= 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool have = lane < nv;
                const uint32_t mypost = have ? (uint32_t)BK_LA_RL(w)[lane] : 0u; const uint32_t myfl = have ? (uint32_t)BK_LA_FL(w)[lane] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const uint32_t u = mypost >> 10; const int pos = (int)(mypost & 1023u);
                uint32_t len = 0; int rn = 0;
                if (have) { len = C_.ulen[u]; rn = (int)C_.unr[u]; }
                bool drop = false;                                                 // the k-mer twice in one read: first occurrence (re.search)
                for (int j = 0; j < nv; j++) {
                    const uint32_t oen = (uint32_t)__builtin_amdgcn_readlane((int)mypost, j);
                    drop = drop || ((oen >> 10) == u && (int)(oen & 1023u) < pos);
                }
                const bool valid = have && !drop;
                const unsigned long long pk = rev ? (unsigned long long)(0xFFFF - pos) : (unsigned long long)pos;
                const unsigned long long key = valid ? ((pk << 40) | ((0xFFFFull - len) << 24) | u) : ~0ull;
                int ord = 0;
                for (int j = 0; j < nv; j++) {
                    const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                    ord += kj < key;
                }
                if (valid) { BK_LA_CU(w)[ord] = u | ((uint32_t)pos << 22); BK_LA_RL(w)[ord] = (int)len; BK_LA_RN(w)[ord] = rn; BK_LA_FL(w)[ord] = (int)myfl; }
                cnt = __popcll(__ballot(valid));
                BkKey kk; kk.hi = C_.khi[rank2]; kk.lo = C_.klo[rank2];
                if (!bk_expect_reject()) pc2 = bk_find_kmer_wave(L_CSEQ + S->cbase, S->clen, kk, C_.k);      // the geometry is not used while rejections are expected
            }
        }
    }
    if (lane == 0) { S->la_n[w] = cnt; S->la_t[w] = tt; S->la_rank[w] = rank2; S->la_pc[w] = pc2; }
}

// The same for the SEEDS that follow (setup_contigs :11-26).  With sequencing noise most DP rounds are the first round of a
// seed: an error k-mer shared by two or three reads, one of them the founder.  The seeds to come are the next live k-mers in
// (count, mer) order, their candidate lists are find_reads results without a buffer filter (only deleted reads are left out).
// Wavefront w takes the (w+1)-th live k-mer after `rank`: list in find_reads order (entry 0 = founder) with the slot fields.
BK_COLD void bk_seedahead_wave(int w, int rank)
{
    BkAsmShared *S = S_;
    const int lane = BK_TID & 63;
    // (the seeds that follow in THIS unit's list: S->head is where the running seed sits in it)
    const int li = S->head + 1 + lane;
    const int r = li < C_.n_my ? bk_seed_at(li) : -1;
    const bool live = r >= 0 && C_.kstate[r] == BK_K_LIVE && C_.kcnt[r] >= 2;
    unsigned long long m = __ballot(live);
    for (int i = 0; i < w; i++) m &= m - 1;
    int cnt = -1, rank2 = 0;
    (void)rank;
    if (m) {
        rank2 = __shfl(r, __ffsll((long long)m) - 1);
        const uint32_t b = C_.poff[rank2], e = C_.poff[rank2 + 1];
        if (e - b <= 64u * BK_LA_CH) {
            const int np = (int)(e - b);
            uint32_t pe[BK_LA_CH]; uint32_t fl[BK_LA_CH];
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) pe[c] = lane + 64 * c < np ? C_.post[b + lane + 64 * c] : 0u;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) fl[c] = lane + 64 * c < np ? C_.ufl[pe[c] >> 10] : 0u;
            int nv = 0;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) {
                const bool okc = lane + 64 * c < np && !(fl[c] & BK_R_DELETED);             // used_reads = set(): no buffer filter
                const unsigned long long vm = __ballot(okc);
                const int at = nv + __popcll(vm & ((1ull << lane) - 1ull));
                if (okc && at < 64) { BK_LA_RL(w)[at] = (int)pe[c]; BK_LA_FL(w)[at] = (int)fl[c]; }
                nv += __popcll(vm);
            }
            if (nv <= 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool have = lane < nv;
                const uint32_t mypost = have ? (uint32_t)BK_LA_RL(w)[lane] : 0u; const uint32_t myfl = have ? (uint32_t)BK_LA_FL(w)[lane] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const uint32_t u = mypost >> 10; const int pos = (int)(mypost & 1023u);
                uint32_t len = 0; int rn = 0;
                if (have) { len = C_.ulen[u]; rn = (int)C_.unr[u]; }
                bool drop = false;
                for (int j = 0; j < nv; j++) {
                    const uint32_t oen = (uint32_t)__builtin_amdgcn_readlane((int)mypost, j);
                    drop = drop || ((oen >> 10) == u && (int)(oen & 1023u) < pos);
                }
                const bool valid = have && !drop;
                const unsigned long long key = valid ? (((unsigned long long)pos << 40) | ((0xFFFFull - len) << 24) | u) : ~0ull;
                int ord = 0;
                for (int j = 0; j < nv; j++) {
                    const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                    ord += kj < key;
                }
                if (valid) { BK_LA_CU(w)[ord] = u | ((uint32_t)pos << 22); BK_LA_RL(w)[ord] = (int)len; BK_LA_RN(w)[ord] = rn; BK_LA_FL(w)[ord] = (int)myfl; }
                cnt = __popcll(__ballot(valid));
            }
        }
    }
    if (lane == 0) { S->la_n[w] = cnt; S->la_t[w] = -2 - rank2; S->la_rank[w] = rank2; S->la_pc[w] = 0; }
}

// one step of the prediction chain (thread 0): slot t is aligned against the contig [pb, pb+plen) in which its k-mer sits
// at ppc; what the read is predicted to do to it.  Returns false when nothing can be predicted past this slot.
__device__ inline bool bk_predict(BkAsmShared::Slot &t, int &pb, int &plen, int &ppc, int lo, int hi)
{
    const int pos = t.pos, rl = t.rl;
    t.pb = pb; t.plen = plen;
    // noisy reads: where check_align has lately rejected three reads out of four, the best guess for the next one is that
    // it is rejected too and the contig stays as it is (a rejection after a predicted extension would void the later slots)
    if (bk_expect_reject()) { t.kind = BK_PK_SAME; t.amt = 0; return true; }
    const int left = pos - ppc, right = (rl - pos) - (plen - ppc);
    if (ppc < 0 || (left > 0 && right > 0)) { t.kind = BK_PK_STOP; t.amt = 0; return false; }
    if (left > 0) { t.kind = BK_PK_PRE; t.amt = left; if (pb - left < lo || plen + left > C_.MAXC) { t.kind = BK_PK_STOP; return false; } pb -= left; plen += left; ppc += left; }
    else if (right > 0) { t.kind = BK_PK_POST; t.amt = right; if (pb + plen + right > hi || plen + right > C_.MAXC) { t.kind = BK_PK_STOP; return false; } plen += right; }
    else { t.kind = BK_PK_SAME; t.amt = 0; }
    return true;
}
