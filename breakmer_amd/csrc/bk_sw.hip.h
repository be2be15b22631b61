// bk_sw.hip.h -- stage BK_STAGE_REALIGN: contig -> reference-window realignment (replaces the external
// BLAT call, sv_processor.py:823-851; BLAT parity is UNPINNED, the contract is oracle/bk_oracle.h R2).
//
// Iterated gap-free local Smith-Waterman (maximal scoring segment, match +1 / mismatch -2) of the
// still-unaligned query intervals against every window on both strands.  With no gap transitions
// every cell depends only on its diagonal predecessor, so the diagonals of the DP matrix are
// independent: a thread walks a whole diagonal (H = max(0, H + s)) and keeps its best
// (score, query end, run); one 64-bit max-reduction applies the contract's tie-break.
// Exact pruning: the best segment of a diagonal cannot score more than the diagonal has matching
// positions, and that count U costs ~0.5 op/base on the 2-bit packed sequences (xor + popcount).
// Per staged target chunk: (1) U of every diagonal, (2) the diagonals with the block-wide largest
// U are walked and raise the lower bound L (an achieved score), (3) every diagonal with U >= L is
// walked.  A diagonal with U < L cannot hold the maximum, ties (U == L) are walked, so the result
// equals the brute-force scan.  Query interval and target are staged in LDS as packed words; the target stays staged
// across passes and contigs while it is the same chunk.
// One workgroup per CONTIG: persistent workgroups pull entries of the contig list the assembler appended to
// (contigs are independent; a noisy region has thousands of them), query intervals of a contig are processed in sequence.
// Secondary alignments (contract step 5: BLAT prints every alignment >= -minScore and the caller counts them per query base):
// after the iteration one more sweep over ALL diagonals of the WHOLE query reports every positive excursion whose peak is
// >= min_score and that is not the same alignment as a step-1 hit (same target, strand, diagonal, overlapping query bases).
// A diagonal is only walked when a word-granular upper bound of H (16 bases per step: matches by popcount) can reach
// min_score somewhere on it -- on unrelated sequence it never does, so the sweep costs ~1 op per base of the matrix.
// Chaining of collinear hits into PSL records is host code (bk_api.hip), restated in the oracle.
#pragma once
#include "bk_common.h"
#undef BK_SRC_ID
#define BK_SRC_ID 6      // barrier sites of this file (bk_common.h: BK_SYNC)

#define BK_ST_TMAX 512                 // largest workgroup (the LONG tier and batches of few contigs); the SHORT tier runs smaller ones
#define BK_ST_T ((int)blockDim.x)
#define BK_SW_MIN_SEG 20
#define BK_SW_FLAGS 1024

// Step-1 hits tile the query; a hit holds at least min_score bases (it scores >= min_score, +1 per match) and is only looked for
// in intervals of >= BK_SW_MIN_SEG bases, so a contig of at most max_contig bases has at most max_contig / min(BK_SW_MIN_SEG,
// min_score) of them: the hit list and the interval stack are sized for that in the dynamic LDS block and cannot overflow
// (should they ever, the region fails with BK_ST_HITS: never a silent drop).  Secondary alignments have no such bound (a microsatellite in
// the window: hundreds per contig): BK_SEC_LDS of them are collected in LDS; when there are more the sweep is repeated
// writing straight into the result arena (their number is known by then).  Neither is a cap.
#define BK_SEC_LDS 256
__host__ __device__ inline int bk_sw_max_hits(int max_contig, int min_score) { const int ms = min_score < 1 ? 1 : min_score; return max_contig / (ms < BK_SW_MIN_SEG ? ms : BK_SW_MIN_SEG) + 2; }
struct BkSwShared {
    int nseg;
    int nhits;
    int nsec;                                    // secondary alignments (step 5) collected in the LDS list `sec` (dynamic block), in no particular order
    int nflag;                                   // diagonals of the whole query on which H can reach min_score (found by the first pass): flag_off / flag_ts (dynamic block): (offset, target << 1 | strand)
    unsigned long long red[BK_ST_TMAX / 64]; int red_run[BK_ST_TMAX / 64];
    unsigned long long best_key; int best_run;
    unsigned long long cells;
    int L, umax;                 // lower bound on the pass maximum (an achieved score); block maximum of U
    int staged_ti, staged_t0, staged_t1, staged_region;   // which target bases the packed staging buffer currently holds
    unsigned long long rec_off;
    int status;
    unsigned long long qidx; int skip;
};

// Two tiers (round 4).  SHORT: contigs of up to `contig_cap` bases on small workgroups with an LDS block sized for THAT (about
// 12 KB instead of 40: a dozen contigs per CU in flight, and the block fits next to the assembler's workgroups of the other
// batches in flight); a longer contig is appended to `sw_long` and taken by the LONG tier, a second launch of 512-thread
// workgroups with the block sized for max_contig (idle when that list is empty, the rule).  Both run the same code.
struct BkSwTier { uint32_t tw_cap; int contig_cap, sec_lds, n_flags, mode; };      // mode 0: the contig list, longer ones deferred; 1: the deferred ones
struct BkSwLayout { uint32_t qf, qpk, tp, tnb, hits, seg, sec, foff, fts, total; int qpw, max_hits; };
__host__ __device__ inline BkSwLayout bk_sw_layout(int contig_cap, uint32_t tw_cap, int sec_lds, int n_flags, int min_score)
{
    BkSwLayout L;
    L.qpw = contig_cap / 16 + 2; L.max_hits = bk_sw_max_hits(contig_cap, min_score);
    uint32_t o = (uint32_t)((sizeof(BkSwShared) + 15) / 16) * 16;
    L.qf = o; o += 2 * (uint32_t)contig_cap; o = (o + 15) / 16 * 16;            // contig, forward and reverse complement (codes)
    L.qpk = o; o += 4 * 4 * (uint32_t)L.qpw;                                   // packed query interval + N masks, both strands
    L.tp = o; o += 4 * (tw_cap / 16 + 8);                                      // staged target chunk, packed words
    L.tnb = o; o += 4 * (tw_cap / 16 + 8);                                     // its N mask
    L.hits = o; o += (uint32_t)L.max_hits * (uint32_t)sizeof(BkHit);           // step-1 hits
    L.seg = o; o += 2 * (2 * (uint32_t)L.max_hits + 4) * 4;                    // stack of query intervals still to be aligned
    o = (o + 15) / 16 * 16;
    L.sec = o; o += (uint32_t)sec_lds * (uint32_t)sizeof(BkHit);               // secondary alignments collected in the LDS
    L.foff = o; o += 4 * (uint32_t)n_flags; L.fts = o; o += 4 * (uint32_t)n_flags;
    L.total = (o + 15) / 16 * 16;
    return L;
}

__device__ inline unsigned long long bk_sw_key(int score, int tidx, int strand, int a, long long b)
{   // larger key wins: score desc, target index asc, '+' first, smallest query end, smallest target end
    // [score:15 | 15-tidx:4 | 1-strand:1 | 0x7FFF-a:15 | 0x1FFFFFFF-b:29]   (contigs < 32,768 bases, windows < 512 Mb)
    return ((unsigned long long)score << 49) | ((unsigned long long)(15 - tidx) << 45) | ((unsigned long long)(1 - strand) << 44) |
           ((unsigned long long)(0x7FFF - a) << 29) | (unsigned long long)(0x1FFFFFFFll - b);
}

// N mask of the staged target under the 16 bases that start at word i0, shift sh (as the packed words themselves are fetched); bit
// 2*(15-t) set where base t is an N.  tn == nullptr: the window has no N (the rule).
__device__ inline uint32_t bk_sw_tn(const uint32_t *tn, int i0, int tpn, int sh)
{
    if (!tn) return 0u;
    const uint32_t m0 = (unsigned)i0 < (unsigned)tpn ? tn[i0] : 0u, m1 = (unsigned)(i0 + 1) < (unsigned)tpn ? tn[i0 + 1] : 0u;
    return sh ? (m0 << sh) | (m1 >> (32 - sh)) : m0;
}
// matching positions on diagonal `off` (target b = query a + off) of a packed query interval (n bases, qp) against the
// staged packed target words tp[0..tpn) = target words tpw0.. (16 bases per word, MSB first)
// qn: N mask of the packed query (bit 2*(15-t) set where base t of the word is an N: an N matches nothing here)
__device__ inline int bk_sw_diag_count_fwd(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off);
__device__ inline int bk_sw_diag_matches(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off)
{
    return bk_sw_diag_count_fwd(qp, qn, tp, tn, tpw0, tpn, n, m, off);
}
// walk one diagonal on the packed words: best positive run (strict '>': smallest query end among equal scores).  Inside
// a run of matches the score rises strictly, so only the end of each run can become the new best: the loop advances
// from mismatch to mismatch (count-leading-zeros on the mismatch mask) instead of base by base.
__device__ inline void bk_sw_walk(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off, int &bh, int &ba, int &br)
{
    const int a0 = off < 0 ? -off : 0, a1 = min(n, m - off);
    int h = 0, run = 0;
    bh = 0; ba = 0; br = 0;
    if (a1 <= a0) return;
    for (int wq = a0 >> 4; wq <= (a1 - 1) >> 4; wq++) {
        const int aw = wq << 4, pb = aw + off;
        const int i0 = (pb >> 4) - tpw0, sh = 2 * (pb & 15);
        const uint32_t w0 = (unsigned)i0 < (unsigned)tpn ? tp[i0] : 0u, w1 = (unsigned)(i0 + 1) < (unsigned)tpn ? tp[i0 + 1] : 0u;
        const uint32_t tb = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        const uint32_t x = (qp[wq] ^ tb) | bk_sw_tn(tn, i0, tpn, sh);        // an N of the window matches nothing
        const int lo = max(a0 - aw, 0), hi = min(a1 - aw, 16);
        uint32_t stop = (x | (x >> 1) | qn[wq] | (hi < 16 ? 0xFFFFFFFFu >> (2 * hi) : 0u)) & 0x55555555u;   // mismatches (an N is one) and everything from `hi` on
        int pos = lo;
        while (pos < hi) {
            const uint32_t rest = stop & (0xFFFFFFFFu >> (2 * pos));
            const int ts = rest ? (__clz((int)rest) - 1) >> 1 : 16;           // first non-matching position >= pos
            const int r = ts - pos;
            if (r > 0) { h += r; run += r; if (h > bh) { bh = h; ba = aw + ts; br = run; } }
            if (ts >= hi) break;
            h -= 2; run++;
            if (h <= 0) { h = 0; run = 0; }
            pos = ts + 1;
        }
    }
}

// Can H = max(0, H + s) reach `thresh` anywhere on diagonal `off`?  Word-granular upper bound: inside a word H rises by at
// most its matches; after the word it is at most max(H + matches - 2 mismatches, matches) (the second term: a fresh
// start inside the word).  false => no excursion of the diagonal peaks at >= thresh (exact walks are only needed for `true`).
__device__ inline bool bk_sw_diag_maybe(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off, int thresh)
{
    const int a0 = off < 0 ? -off : 0, a1 = min(n, m - off);
    if (a1 - a0 < thresh) return false;
    int hub = 0;
    for (int wq = a0 >> 4; wq <= (a1 - 1) >> 4; wq++) {
        const int aw = wq << 4, pb = aw + off;
        const int i0 = (pb >> 4) - tpw0, sh = 2 * (pb & 15);
        const uint32_t w0 = (unsigned)i0 < (unsigned)tpn ? tp[i0] : 0u, w1 = (unsigned)(i0 + 1) < (unsigned)tpn ? tp[i0 + 1] : 0u;
        const uint32_t tb = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        const uint32_t x = (qp[wq] ^ tb) | bk_sw_tn(tn, i0, tpn, sh);        // an N of the window matches nothing
        const uint32_t eq = ~(x | (x >> 1) | qn[wq]) & 0x55555555u;
        const int lo = max(a0 - aw, 0), hi = min(a1 - aw, 16);
        uint32_t vm = 0xFFFFFFFFu >> (2 * lo);
        if (hi < 16) vm &= ~(0xFFFFFFFFu >> (2 * hi));
        const int mm = __popc(eq & vm);
        if (hub + mm >= thresh) return true;
        hub = max(hub + mm - 2 * (hi - lo - mm), mm);
    }
    return false;
}
// match count of a diagonal (as bk_sw_diag_matches) and -- SCAN -- in the same pass over its words, the word-granular upper bound of
// bk_sw_diag_maybe: `maybe` = H can reach thresh somewhere on it.  Branch-free (the loads of the next words are not held up).
// The sweeps of the realigner spend their time here (a contig of Q bases against a window of W: 2 (Q + W) diagonals of Q / 16 words
// per pass), so the words BETWEEN a diagonal's first and last one take a short path: all 16 bases lie inside the diagonal, both
// target words are staged (the staged range covers every diagonal of the chunk), the shift is the same for every word of the
// diagonal ((16 wq + off) & 15 = off & 15), and the second target word of one step is the first of the next.
template <bool SCAN>
__device__ inline int bk_sw_diag_count(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off, int thresh, bool &maybe)
{
    const int a0 = off < 0 ? -off : 0, a1 = min(n, m - off);
    maybe = false;
    if (a1 <= a0) return 0;
    int u = 0, hub = 0, top = 0;
    auto edge = [&](int wq) {
        const int aw = wq << 4, pb = aw + off;
        const int i0 = (pb >> 4) - tpw0, sh = 2 * (pb & 15);
        const uint32_t w0 = (unsigned)i0 < (unsigned)tpn ? tp[i0] : 0u, w1 = (unsigned)(i0 + 1) < (unsigned)tpn ? tp[i0 + 1] : 0u;
        const uint32_t tb = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        const uint32_t x = (qp[wq] ^ tb) | bk_sw_tn(tn, i0, tpn, sh);        // an N of the window matches nothing
        const uint32_t eq = ~(x | (x >> 1) | qn[wq]) & 0x55555555u;
        const int lo = max(a0 - aw, 0), hi = min(a1 - aw, 16);
        uint32_t vm = 0xFFFFFFFFu >> (2 * lo);
        if (hi < 16) vm &= ~(0xFFFFFFFFu >> (2 * hi));
        const int mm = __popc(eq & vm);
        u += mm;
        if (SCAN) { top = max(top, hub + mm); hub = max(hub + 3 * mm - 2 * (hi - lo), mm); }
    };
    const int wlo = a0 >> 4, whi = (a1 - 1) >> 4;
    if (tn) { for (int wq = wlo; wq <= whi; wq++) edge(wq); }               // a window that holds an N (rare): every word the general way
    else {
        edge(wlo);
        if (whi - wlo >= 2) {
            const int pb = ((wlo + 1) << 4) + off, sh = 2 * (pb & 15);
            int i0 = (pb >> 4) - tpw0;
            uint32_t w0 = tp[i0];
            for (int wq = wlo + 1; wq < whi; wq++, i0++) {
                const uint32_t w1 = tp[i0 + 1];                              // (sh == 0 at the end of the staged range: one of the buffer's 8 slack words, not used)
                const uint32_t tb = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
                const uint32_t x = qp[wq] ^ tb;
                const int mm = __popc(~(x | (x >> 1) | qn[wq]) & 0x55555555u);
                u += mm;
                if (SCAN) { top = max(top, hub + mm); hub = max(hub + 3 * mm - 32, mm); }
                w0 = w1;
            }
        }
        if (whi > wlo) edge(whi);
    }
    maybe = top >= thresh;
    return u;
}
__device__ inline int bk_sw_diag_count_fwd(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off)
{
    bool mb; return bk_sw_diag_count<false>(qp, qn, tp, tn, tpw0, tpn, n, m, off, 0, mb);
}
__device__ inline int bk_sw_diag_scan(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off, int thresh, bool &maybe)
{
    return bk_sw_diag_count<true>(qp, qn, tp, tn, tpw0, tpn, n, m, off, thresh, maybe);
}
// walk one diagonal over the whole query and report every positive excursion (reset to reset / end of the diagonal) whose
// peak is >= thresh: emit(peak, query end of the first position of the peak, length of the segment from the excursion's start)
template <class F>
__device__ inline void bk_sw_walk_all(const uint32_t *qp, const uint32_t *qn, const uint32_t *tp, const uint32_t *tn, int tpw0, int tpn, int n, int m, int off, int thresh, F emit)
{
    const int a0 = off < 0 ? -off : 0, a1 = min(n, m - off);
    int h = 0, run = 0, bh = 0, ba = 0, br = 0;
    if (a1 <= a0) return;
    for (int wq = a0 >> 4; wq <= (a1 - 1) >> 4; wq++) {
        const int aw = wq << 4, pb = aw + off;
        const int i0 = (pb >> 4) - tpw0, sh = 2 * (pb & 15);
        const uint32_t w0 = (unsigned)i0 < (unsigned)tpn ? tp[i0] : 0u, w1 = (unsigned)(i0 + 1) < (unsigned)tpn ? tp[i0 + 1] : 0u;
        const uint32_t tb = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        const uint32_t x = (qp[wq] ^ tb) | bk_sw_tn(tn, i0, tpn, sh);        // an N of the window matches nothing
        const int lo = max(a0 - aw, 0), hi = min(a1 - aw, 16);
        uint32_t stop = (x | (x >> 1) | qn[wq] | (hi < 16 ? 0xFFFFFFFFu >> (2 * hi) : 0u)) & 0x55555555u;
        int pos = lo;
        while (pos < hi) {
            const uint32_t rest = stop & (0xFFFFFFFFu >> (2 * pos));
            const int ts = rest ? (__clz((int)rest) - 1) >> 1 : 16;
            const int r = ts - pos;
            if (r > 0) { h += r; run += r; if (h > bh) { bh = h; ba = aw + ts; br = run; } }
            if (ts >= hi) break;
            h -= 2; run++;
            if (h <= 0) { if (bh >= thresh) emit(bh, ba, br); h = 0; run = 0; bh = 0; }
            pos = ts + 1;
        }
    }
    if (bh >= thresh) emit(bh, ba, br);
}
// packed copies (and N masks) of query interval [qs, qe) of the contig, both strands
__device__ inline void bk_sw_pack_query(const uint8_t *qf, const uint8_t *qr, int Q, int qs, int qe, uint32_t *qpk, uint32_t *qnm, int qpw, int tid)
{
    const int n = qe - qs;
    for (int w = tid; w < 2 * ((n + 15) / 16); w += BK_ST_T) {
        const int st = w >= (n + 15) / 16, wi = st ? w - (n + 15) / 16 : w;
        const uint8_t *q = st ? qr + (Q - qe) : qf + qs;
        uint32_t x = 0, nm = 0;
        for (int t = 0; t < 16; t++) { const int a = wi * 16 + t; const uint32_t c = a < n ? (uint32_t)q[a] : 0u; x = (x << 2) | (c & 3u); nm = (nm << 2) | (c >> 2); }
        qpk[st * qpw + wi] = x; qnm[st * qpw + wi] = nm;
    }
}

extern "C" __global__ void __launch_bounds__(BK_ST_TMAX) bk_sw_kernel(BkParams p, BkSwTier T)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sl[];
    const int tid = threadIdx.x;
    const uint32_t tw_cap = T.tw_cap;
    const BkSwLayout LY = bk_sw_layout(T.contig_cap, T.tw_cap, T.sec_lds, T.n_flags, p.sw_min_score);
    BkSwShared *S = (BkSwShared *)sl;
    uint8_t *qf = sl + LY.qf;                                          // contig forward (codes)
    uint8_t *qr = qf + T.contig_cap;                                   // contig reverse complement
    uint32_t *qpk = (uint32_t *)(sl + LY.qpk);                         // packed query interval, both strands
    const int qpw = LY.qpw;
    uint32_t *qnm = qpk + 2 * qpw;                                     // N masks of the packed query interval, both strands
    uint32_t *tp = (uint32_t *)(sl + LY.tp);                           // staged target chunk, packed words (kept across passes and contigs)
    uint32_t *tnb = (uint32_t *)(sl + LY.tnb);                         // its N mask, filled (and used) only for windows that hold an N
    const int max_hits = LY.max_hits;
    BkHit *hits = (BkHit *)(sl + LY.hits);                             // step-1 hits of the contig (cannot overflow: see BkSwShared)
    int *seg = (int *)(sl + LY.seg);                                   // stack of query intervals still to be aligned: 2 * (2 * max_hits + 4) ints
    BkHit *sec = (BkHit *)(sl + LY.sec);
    int *flag_off = (int *)(sl + LY.foff), *flag_ts = (int *)(sl + LY.fts);
    if (tid == 0) { S->staged_ti = -1; S->staged_t0 = 0; S->staged_t1 = 0; S->staged_region = -1; }
    const unsigned long long n_list = T.mode == 0 ? min(*p.n_clist, (unsigned long long)p.clist_cap) : min(*p.n_sw_long, (unsigned long long)p.clist_cap);
    const unsigned long long *list = T.mode == 0 ? p.clist : p.sw_long;
    for (;;) {
        BK_SYNC();
        if (tid == 0) {
            const unsigned long long qi = atomicAdd(T.mode == 0 ? p.sw_head : p.sw_long_head, 1ull);
            S->qidx = qi; S->cells = 0; S->status = 0; S->skip = 0;
            if (qi < n_list) {                                          // one reader of the (concurrently written) region status
                const unsigned long long e0 = list[qi];
                const int rr = (int)(e0 >> 40);
                const int rst = p.work[rr].status;
                S->skip = rst != BK_ST_OK && rst != BK_ST_REDO;          // (a split region that awaits a repair pass: most of its contigs stay, and this launch is the one that sees them)
                if (!S->skip && T.mode == 0 && ((const BkContigRec *)(p.out + (e0 & ((1ull << 40) - 1ull))))->seq_len > T.contig_cap) {
                    const unsigned long long li = atomicAdd(p.n_sw_long, 1ull);        // too long for this tier's LDS block: the LONG tier takes it
                    if (li < p.clist_cap) p.sw_long[li] = e0;
                    S->skip = 1;
                }
                if (!S->skip && p.work[rr].split) {                       // a contig of a component that ran again in a later pass (bk_comp.hip.h) is dead: not in the region's list, nobody reads its hits
                    const BkContigRec *c0 = (const BkContigRec *)(p.out + (e0 & ((1ull << 40) - 1ull)));
                    if (c0->root != BK_EMPTY32) {
                        const uint32_t *rroot = (const uint32_t *)(p.arena + p.work[rr].o_rroot), *cinfo = (const uint32_t *)(p.arena + p.work[rr].o_cinfo);
                        if ((cinfo[rroot[c0->root]] & 0xFFFFu) != c0->pass) S->skip = 1;
                    }
                }
                if (!S->skip && S->staged_region != rr) { S->staged_ti = -1; S->staged_region = rr; }
            }
        }
        BK_SYNC();
        if (S->qidx >= n_list) break;
        if (S->skip) continue;
        const unsigned long long ent = list[S->qidx];
        const int r = (int)(ent >> 40);
        const unsigned long long roff = ent & ((1ull << 40) - 1ull);
        BkRegionWork *wk = &p.work[r];
        const BkRegionDesc d = p.desc[r];
        BkContigRec *rec = (BkContigRec *)(p.out + roff);
        const int Q = rec->seq_len;
        const char *seq = (const char *)(p.out + roff + rec->o_seq);
        for (int i = tid; i < Q; i += BK_ST_T) { char ch = seq[i]; uint8_t c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : BK_CODE_N; qf[i] = c; qr[Q - 1 - i] = c == BK_CODE_N ? (uint8_t)BK_CODE_N : (uint8_t)(3 - c); }
        if (tid == 0) { S->nseg = 1; seg[0] = 0; seg[1] = Q; S->nhits = 0; S->nsec = 0; S->nflag = 0; }
        BK_SYNC();
        while (S->nseg > 0 && S->status == 0) {
            const int qs = seg[2 * (S->nseg - 1)], qe = seg[2 * (S->nseg - 1) + 1], n = qe - qs;
            BK_SYNC();
            if (tid == 0) { S->nseg--; S->best_key = 0; S->best_run = 0; S->L = 0; }
            BK_SYNC();
            if (n < BK_SW_MIN_SEG) continue;
            const bool first = qs == 0 && qe == Q && S->nhits == 0;              // the first pass: the whole query
            // packed copies of the query interval (both strands) for the match counts
            bk_sw_pack_query(qf, qr, Q, qs, qe, qpk, qnm, qpw, tid);
            unsigned long long bkey = 0; int brun = 0;
            for (int ti = 0; ti <= (int)d.n_partners; ti++) {
                const uint32_t *gw; int m;
                const uint32_t *wn; int nn;                                  // N positions of this window (none, as a rule)
                if (ti == 0) { gw = p.windows + d.win_word_off; m = (int)d.win_len; wn = p.wnlist + d.win_n_off; nn = (int)d.n_win_n; }
                else { const BkPartnerDesc pd = p.partners[d.part_desc_off + ti - 1]; gw = p.windows + pd.word_off; m = (int)pd.len; wn = p.wnlist + pd.n_off; nn = (int)pd.n_n; }
                const uint32_t *tnp = nn ? tnb : nullptr;
                // diagonals off = b - a in [-(n-1), m-1] are independent: they are processed in chunks whose target
                // bases [o0, o0 + CH + n - 1) fit the staging buffer (one chunk when the window is short)
                const int CH = (int)tw_cap - n, mw = (m + 15) / 16;
                for (int o0 = -(n - 1); o0 < m; o0 += CH) {
                    const int o1 = min(o0 + CH, m), t0 = max(o0, 0), t1 = min(m, o1 - 1 + n);
                    const int tpw0 = t0 >> 4, tpn = ((t1 + 15) >> 4) - tpw0;
                    BK_SYNC();
                    const bool staged = S->staged_ti == ti && S->staged_t0 == t0 && S->staged_t1 == t1;   // e.g. one short window: staged once per region
                    BK_SYNC();
                    if (!staged) for (int i = tid; i < tpn; i += BK_ST_T) { tp[i] = tpw0 + i < mw ? gw[tpw0 + i] : 0u; if (nn) tnb[i] = 0u; }
                    if (tid == 0) { S->umax = 0; S->staged_ti = ti; S->staged_t0 = t0; S->staged_t1 = t1; }
                    BK_SYNC();
                    if (!staged && nn) {
                        for (int e = tid; e < nn; e += BK_ST_T) { const int pn = (int)wn[e], wi = (pn >> 4) - tpw0; if ((unsigned)wi < (unsigned)tpn) atomicOr(&tnb[wi], 1u << (2 * (15 - (pn & 15)))); }
                        BK_SYNC();
                    }
                    const int nd = o1 - o0;
                    // (1) match counts; this thread's two best diagonals and the largest count among its others.  The first pass
                    // (the whole query) also notes the diagonals on which H can reach min_score at all: step 5 walks only those.
                    int myu = -1, myD = 0, myu2 = -1, myD2 = 0, myu3 = -1;
                    for (int D = tid; D < 2 * nd; D += BK_ST_T) {
                        const int st = D >= nd, off = o0 + (st ? D - nd : D);
                        int u;
                        if (first) {
                            bool maybe; u = bk_sw_diag_scan(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off, p.sw_min_score, maybe);
                            if (maybe) { const int fi = atomicAdd(&S->nflag, 1); if (fi < T.n_flags) { flag_off[fi] = off; flag_ts[fi] = (ti << 1) | st; } }
                        } else u = bk_sw_diag_matches(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off);
                        if (u > myu) { myu3 = myu2; myu2 = myu; myD2 = myD; myu = u; myD = D; }
                        else if (u > myu2) { myu3 = myu2; myu2 = u; myD2 = D; }
                        else if (u > myu3) myu3 = u;
                    }
                    if (myu > 0) atomicMax(&S->umax, myu);
                    BK_SYNC();
                    // (2) the diagonals with the largest count raise the lower bound
                    int walked = -1;
                    if (myu == S->umax && myu >= S->L && myu > 0) {
                        const int st = myD >= nd, off = o0 + (st ? myD - nd : myD);
                        int bh, ba, br; bk_sw_walk(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off, bh, ba, br);
                        if (bh > 0) { const unsigned long long key = bk_sw_key(bh, ti, st, ba, (long long)ba + off); if (key > bkey) { bkey = key; brun = br; } atomicMax(&S->L, bh); }
                        walked = myD;
                    }
                    BK_SYNC();
                    // (3) every diagonal that can still hold the maximum (count >= L; ties are walked): this thread's best one if (2)
                    // did not take it, its second best, and -- only if even the largest of its other counts reaches L -- a
                    // rescan of all its diagonals (rare: the counts of unrelated diagonals are far below an achieved score)
                    auto walk_one = [&](int D) {
                        const int st = D >= nd, off = o0 + (st ? D - nd : D);
                        int bh, ba, br; bk_sw_walk(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off, bh, ba, br);
                        if (bh > 0) { const unsigned long long key = bk_sw_key(bh, ti, st, ba, (long long)ba + off); if (key > bkey) { bkey = key; brun = br; } if (bh > *(volatile int *)&S->L) atomicMax(&S->L, bh); }
                    };
                    if (walked < 0 && myu > 0 && myu >= *(volatile int *)&S->L) walk_one(myD);
                    if (myu2 > 0 && myu2 >= *(volatile int *)&S->L) walk_one(myD2);
                    if (myu3 > 0 && myu3 >= *(volatile int *)&S->L) {
                        for (int D = tid; D < 2 * nd; D += BK_ST_T) {
                            if (D == myD || D == myD2) continue;
                            const int st = D >= nd, off = o0 + (st ? D - nd : D);
                            const int u = bk_sw_diag_matches(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off);
                            if (u < *(volatile int *)&S->L || u == 0) continue;
                            walk_one(D);
                        }
                    }
                }
                if (tid == 0) S->cells += 2ull * (unsigned long long)n * (unsigned long long)m;
                BK_SYNC();
            }
            // block reduction of the best key (unique: (target, strand, a, b) identify a cell)
            for (int o = 32; o > 0; o >>= 1) { unsigned long long ok = __shfl_xor(bkey, o); int orun = __shfl_xor(brun, o); if (ok > bkey) { bkey = ok; brun = orun; } }
            if ((tid & 63) == 0) { S->red[tid >> 6] = bkey; S->red_run[tid >> 6] = brun; }
            BK_SYNC();
            if (tid == 0) {
                for (int w = 0; w < (BK_ST_T + 63) / 64; w++) if (S->red[w] > S->best_key) { S->best_key = S->red[w]; S->best_run = S->red_run[w]; }
                const unsigned long long key = S->best_key; const int score = (int)(key >> 49);
                if (score >= p.sw_min_score) {
                    const int tidx = 15 - (int)((key >> 45) & 15), st = 1 - (int)((key >> 44) & 1), a1 = 0x7FFF - (int)((key >> 29) & 0x7FFF), b1 = (int)(0x1FFFFFFFll - (long long)(key & 0x1FFFFFFFull)), run = S->best_run;
                    const int offq = st ? Q - qe : qs;
                    BkHit hgt; hgt.qs = offq + a1 - run; hgt.qe = offq + a1; hgt.ts = b1 - run; hgt.te = b1; hgt.strand = st; hgt.tidx = tidx; hgt.score = score;
                    const int fs = st ? Q - hgt.qe : hgt.qs, fe = st ? Q - hgt.qs : hgt.qe;
                    hgt.fq = fs;
                    if (S->nhits < max_hits) {                                                // always (the hits tile the query)
                        hits[S->nhits++] = hgt;
                        seg[2 * S->nseg] = fe; seg[2 * S->nseg + 1] = qe; S->nseg++;            // right remainder (after the left one)
                        seg[2 * S->nseg] = qs; seg[2 * S->nseg + 1] = fs; S->nseg++;
                    } else S->status = BK_ST_HITS;                                            // (unreachable by the sizing above; loud if it ever is)
                }
            }
            BK_SYNC();
        }
        // ---- step 5: secondary alignments (a contig without a step-1 hit has none: the first pass saw every diagonal) ----
        BK_SYNC();
        if (S->nhits > 0 && S->status == 0) {
            const int n = Q, nh1 = S->nhits;
            bk_sw_pack_query(qf, qr, Q, 0, Q, qpk, qnm, qpw, tid);
            // dst / cap: where the segments go (the LDS list first; the result arena when that list was too short)
            auto sweep = [&](BkHit *dst, int cap, bool count_cells) {
            for (int ti = 0; ti <= (int)d.n_partners; ti++) {
                const uint32_t *gw; int m;
                const uint32_t *wn; int nn;                                  // N positions of this window (none, as a rule)
                if (ti == 0) { gw = p.windows + d.win_word_off; m = (int)d.win_len; wn = p.wnlist + d.win_n_off; nn = (int)d.n_win_n; }
                else { const BkPartnerDesc pd = p.partners[d.part_desc_off + ti - 1]; gw = p.windows + pd.word_off; m = (int)pd.len; wn = p.wnlist + pd.n_off; nn = (int)pd.n_n; }
                const uint32_t *tnp = nn ? tnb : nullptr;
                const int CH = (int)tw_cap - n, mw = (m + 15) / 16;
                for (int o0 = -(n - 1); o0 < m; o0 += CH) {
                    const int o1 = min(o0 + CH, m), t0 = max(o0, 0), t1 = min(m, o1 - 1 + n);
                    const int tpw0 = t0 >> 4, tpn = ((t1 + 15) >> 4) - tpw0;
                    BK_SYNC();
                    const bool staged = S->staged_ti == ti && S->staged_t0 == t0 && S->staged_t1 == t1;
                    BK_SYNC();
                    if (!staged) for (int i = tid; i < tpn; i += BK_ST_T) { tp[i] = tpw0 + i < mw ? gw[tpw0 + i] : 0u; if (nn) tnb[i] = 0u; }
                    if (tid == 0) { S->staged_ti = ti; S->staged_t0 = t0; S->staged_t1 = t1; }
                    BK_SYNC();
                    if (!staged && nn) {
                        for (int e = tid; e < nn; e += BK_ST_T) { const int pn = (int)wn[e], wi = (pn >> 4) - tpw0; if ((unsigned)wi < (unsigned)tpn) atomicOr(&tnb[wi], 1u << (2 * (15 - (pn & 15)))); }
                        BK_SYNC();
                    }
                    const int nd = o1 - o0;
                    const int nfl = S->nflag, nwork = nfl <= T.n_flags ? nfl : 2 * nd;      // the diagonals the first pass flagged; all of them if that list overflowed
                    for (int W = tid; W < nwork; W += BK_ST_T) {
                        int st, off;
                        if (nfl <= T.n_flags) {
                            const int ts = flag_ts[W]; off = flag_off[W]; st = ts & 1;
                            if ((ts >> 1) != ti || off < o0 || off >= o1) continue;
                        } else {
                            st = W >= nd; off = o0 + (st ? W - nd : W);
                            if (!bk_sw_diag_maybe(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off, p.sw_min_score)) continue;
                        }
                        bk_sw_walk_all(qpk + st * qpw, qnm + st * qpw, tp, tnp, tpw0, tpn, n, m, off, p.sw_min_score, [&](int sc, int aend, int run) {
                            const int sqs = aend - run, sqe = aend;                      // strand coordinates (the whole query: no interval offset)
                            for (int x = 0; x < nh1; x++) {
                                const BkHit &an = hits[x];
                                if (an.tidx == ti && an.strand == st && an.ts - an.qs == off && an.qs < sqe && sqs < an.qe) return;   // the same alignment
                            }
                            const int idx = atomicAdd(&S->nsec, 1);
                            if (idx < cap) { BkHit e; e.qs = sqs; e.qe = sqe; e.ts = sqs + off; e.te = sqe + off; e.strand = st; e.tidx = ti; e.score = sc; e.fq = st ? Q - sqe : sqs; dst[idx] = e; }
                        });
                    }
                }
                if (tid == 0 && count_cells) S->cells += 2ull * (unsigned long long)n * (unsigned long long)m;
                BK_SYNC();
            }
            };
            sweep(sec, T.sec_lds, true);
            BK_SYNC();
            const int total = S->nsec;                                       // every thread reads it before it is reset
            BK_SYNC();
            if (total > T.sec_lds) {
                // more than the LDS list holds: room for all of them (behind the step-1 hits) in the result arena, same sweep again
                if (tid == 0) {
                    const uint64_t need = bk_align_up((uint64_t)(nh1 + total) * sizeof(BkHit), 256);
                    const uint64_t off = atomicAdd(p.out_top, (unsigned long long)need);
                    S->rec_off = off + need > p.out_cap ? 0ull : off;
                    if (!S->rec_off) S->status = BK_ST_OUT;
                    S->nsec = 0;
                }
                BK_SYNC();
                if (S->rec_off) sweep((BkHit *)(p.out + S->rec_off) + nh1, total, false);
                BK_SYNC();
                if (tid == 0) S->nsec = S->rec_off ? -total : 0;             // negative: already in the arena
            }
        }
        // write the raw hits next to the contig record
        BK_SYNC();
        if (tid == 0) {
            const int nh = S->nhits; int ns = S->nsec;
            rec->n_hits = nh; rec->n_sec = (uint32_t)(ns < 0 ? -ns : ns); rec->hits_off = 0;
            if (S->status == BK_ST_OUT) { rec->n_hits = 0; rec->n_sec = 0; }
            else if (ns < 0) {                                                  // the secondary alignments are in the arena already: the step-1 hits go in front of them
                BkHit *o = (BkHit *)(p.out + S->rec_off); for (int i = 0; i < nh; i++) o[i] = hits[i]; rec->hits_off = S->rec_off;
            } else if (nh > 0) {
                uint64_t need = bk_align_up((uint64_t)(nh + ns) * sizeof(BkHit), 256);
                uint64_t off = atomicAdd(p.out_top, (unsigned long long)need);
                if (off + need > p.out_cap) { S->status = BK_ST_OUT; rec->n_hits = 0; rec->n_sec = 0; }
                else { BkHit *o = (BkHit *)(p.out + off); for (int i = 0; i < nh; i++) o[i] = hits[i]; for (int i = 0; i < ns; i++) o[nh + i] = sec[i]; rec->hits_off = off; }
            }
            atomicAdd((unsigned long long *)&wk->sw_cells, S->cells);
            if (S->status) wk->status = S->status;
        }
    }
}
