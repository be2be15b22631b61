// bk_sw.hip.h -- stage BK_STAGE_REALIGN: contig -> reference-window realignment (replaces the external
// BLAT call, sv_processor.py:823-851; BLAT parity is UNPINNED, the contract is oracle/bk_oracle.h R2).
//
// Iterated gap-free local Smith-Waterman (maximal scoring segment, match +1 / mismatch -2) of the
// still-unaligned query intervals against every window on both strands.  With no gap transitions
// every cell depends only on its diagonal predecessor, so the diagonals of the DP matrix are
// independent: each of the 256 threads of the workgroup walks whole diagonals
// (H = max(0, H + s)), keeping the best (score, query end, target end) under the contract's
// tie-break, followed by one 64-bit max-reduction.  Sequences are staged in LDS as bytes.
// One workgroup per region; contigs and query intervals are processed in sequence.
// Chaining of collinear hits into PSL records is host code (bk_api.hip), restated in the oracle.
#pragma once
#include "bk_common.h"

#define BK_ST_T 512
#define BK_SW_MIN_SEG 20

struct BkSwShared {
    int nseg; int seg[2 * (2 * BK_MAX_HITS + 4)];
    int nhits; BkHit hits[BK_MAX_HITS];
    unsigned long long red[BK_ST_T / 64]; int red_run[BK_ST_T / 64];
    unsigned long long best_key; int best_run;
    unsigned long long cells;
    unsigned long long rec_off;
    int status;
};

__device__ inline unsigned long long bk_sw_key(int score, int tidx, int strand, int a, long long b)
{   // larger key wins: score desc, target index asc, '+' first, smallest query end, smallest target end
    // [score:13 | 15-tidx:4 | 1-strand:1 | 0x1FFF-a:13 | 0x1FFFFFFFF-b:33]
    return ((unsigned long long)score << 51) | ((unsigned long long)(15 - tidx) << 47) | ((unsigned long long)(1 - strand) << 46) |
           ((unsigned long long)(0x1FFF - a) << 33) | (unsigned long long)(0x1FFFFFFFFll - b);
}

extern "C" __global__ void __launch_bounds__(BK_ST_T) bk_sw_kernel(BkParams p, uint32_t tw_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sl[];
    const int r = blockIdx.x, tid = threadIdx.x;
    BkRegionWork *wk = &p.work[r];
    if (wk->status != BK_ST_OK) return;
    const BkRegionDesc d = p.desc[r];
    BkSwShared *S = (BkSwShared *)sl;
    uint8_t *qf = sl + ((sizeof(BkSwShared) + 15) / 16) * 16;          // contig forward (codes)
    uint8_t *qr = qf + p.max_contig;                                   // contig reverse complement
    uint8_t *tw = qr + p.max_contig;                                   // current target window
    if (tid == 0) { S->cells = 0; S->status = 0; S->rec_off = wk->o_first_contig; }
    __syncthreads();
    while (S->rec_off != 0) {
        const unsigned long long roff = S->rec_off;
        BkContigRec *rec = (BkContigRec *)(p.out + roff);
        const int Q = rec->seq_len;
        const char *seq = (const char *)(p.out + roff + rec->o_seq);
        for (int i = tid; i < Q; i += BK_ST_T) { char ch = seq[i]; uint8_t c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; qf[i] = c; qr[Q - 1 - i] = (uint8_t)(3 - c); }
        if (tid == 0) { S->nseg = 1; S->seg[0] = 0; S->seg[1] = Q; S->nhits = 0; }
        __syncthreads();
        while (S->nseg > 0 && S->nhits < BK_MAX_HITS) {
            const int qs = S->seg[2 * (S->nseg - 1)], qe = S->seg[2 * (S->nseg - 1) + 1], n = qe - qs;
            __syncthreads();
            if (tid == 0) { S->nseg--; S->best_key = 0; S->best_run = 0; }
            __syncthreads();
            if (n < BK_SW_MIN_SEG) continue;
            unsigned long long bkey = 0; int brun = 0;
            for (int ti = 0; ti <= (int)d.n_partners; ti++) {
                const uint32_t *gw; int m;
                if (ti == 0) { gw = p.windows + d.win_word_off; m = (int)d.win_len; }
                else { const BkPartnerDesc pd = p.partners[d.part_desc_off + ti - 1]; gw = p.windows + pd.word_off; m = (int)pd.len; }
                // diagonals off = b - a in [-(n-1), m-1] are independent: they are processed in chunks whose target
                // bases [o0, o0 + CH + n - 1) fit the staging buffer (one chunk when the window is short)
                const int CH = (int)tw_cap - n;
                for (int o0 = -(n - 1); o0 < m; o0 += CH) {
                    const int o1 = min(o0 + CH, m), t0 = max(o0, 0), t1 = min(m, o1 - 1 + n);
                    __syncthreads();
                    for (int i = t0 + tid; i < t1; i += BK_ST_T) tw[i - t0] = (uint8_t)seq_base(gw, i);
                    __syncthreads();
                    const int nd = o1 - o0;
                    for (int D = tid; D < 2 * nd; D += BK_ST_T) {
                        const int st = D >= nd, off = o0 + (st ? D - nd : D);
                        const uint8_t *q = st ? qr + (Q - qe) : qf + qs;
                        int a = off < 0 ? -off : 0, b = a + off;            // first cell of the diagonal (0-based)
                        int h = 0, run = 0;
                        for (; a < n && b < m; a++, b++) {
                            h += (q[a] == tw[b - t0]) ? 1 : -2; run++;
                            if (h <= 0) { h = 0; run = 0; }
                            else { unsigned long long key = bk_sw_key(h, ti, st, a + 1, b + 1); if (key > bkey) { bkey = key; brun = run; } }
                        }
                    }
                }
                if (tid == 0) S->cells += 2ull * (unsigned long long)n * (unsigned long long)m;
                __syncthreads();
            }
            // block reduction of the best key (unique: (target, strand, a, b) identify a cell)
            for (int o = 32; o > 0; o >>= 1) { unsigned long long ok = __shfl_xor(bkey, o); int orun = __shfl_xor(brun, o); if (ok > bkey) { bkey = ok; brun = orun; } }
            if ((tid & 63) == 0) { S->red[tid >> 6] = bkey; S->red_run[tid >> 6] = brun; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 0; w < BK_ST_T / 64; w++) if (S->red[w] > S->best_key) { S->best_key = S->red[w]; S->best_run = S->red_run[w]; }
                const unsigned long long key = S->best_key; const int score = (int)(key >> 51);
                if (score >= p.sw_min_score) {
                    const int tidx = 15 - (int)((key >> 47) & 15), st = 1 - (int)((key >> 46) & 1), a1 = 0x1FFF - (int)((key >> 33) & 0x1FFF), b1 = (int)(0x1FFFFFFFFll - (long long)(key & 0x1FFFFFFFFull)), run = S->best_run;
                    const int offq = st ? Q - qe : qs;
                    BkHit hgt; hgt.qs = offq + a1 - run; hgt.qe = offq + a1; hgt.ts = b1 - run; hgt.te = b1; hgt.strand = st; hgt.tidx = tidx; hgt.score = score;
                    const int fs = st ? Q - hgt.qe : hgt.qs, fe = st ? Q - hgt.qs : hgt.qe;
                    hgt.fq = fs;
                    S->hits[S->nhits++] = hgt;
                    S->seg[2 * S->nseg] = fe; S->seg[2 * S->nseg + 1] = qe; S->nseg++;      // right remainder (after the left one)
                    S->seg[2 * S->nseg] = qs; S->seg[2 * S->nseg + 1] = fs; S->nseg++;
                }
            }
            __syncthreads();
        }
        // write the raw hits next to the contig record
        __syncthreads();
        if (tid == 0) {
            const int nh = S->nhits;
            rec->n_hits = nh; rec->hits_off = 0;
            if (nh > 0) {
                uint64_t need = bk_align_up((uint64_t)nh * sizeof(BkHit), 256);
                uint64_t off = atomicAdd(p.out_top, (unsigned long long)need);
                if (off + need > p.out_cap) { S->status = BK_ST_OUT; rec->n_hits = 0; }
                else { BkHit *o = (BkHit *)(p.out + off); for (int i = 0; i < nh; i++) o[i] = S->hits[i]; rec->hits_off = off; }
            }
            S->rec_off = rec->next;
        }
        __syncthreads();
    }
    if (tid == 0) { wk->sw_cells = S->cells; if (S->status) wk->status = S->status; }
}
