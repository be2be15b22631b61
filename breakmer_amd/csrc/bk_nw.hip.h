// bk_nw.hip.h -- olc.nw (olc.py:40-107) on one 64-lane wavefront, no pointer matrix.
//
// The reference fills an (n+1)x(m+1) score matrix and a pointer matrix, picks the end cell in the
// last column (largest row among equal maxima, olc.py:79-83) and walks the pointers back until a
// border is reached (olc.py:90-105).  Its callers only consume (j_start, i_end, i_start, score)
// -- the aligned strings stripped of '-' are exactly seq1[j_start:m] and seq2[i_start:i_end] -- so
// instead of storing pointers each cell carries the border cell its traceback would end in:
//
//     cell word = [ score : 14 signed | priority : 2 | origin : 16 ]
//     origin    = j            for the top border    (0, j)
//               = 0x8000 | i   for the left border   (i, 0)
//
// One signed max3 over (diag + s, left + gap, up + gap) picks the predecessor with the reference's
// tie-break (diagonal 3 > S[i][j-1] 2 > S[i-1][j] 1, olc.py:69-74) because the 2-bit priority sits
// directly below the score; the origin rides along in the low bits and never influences the
// comparison (priorities are distinct).
//
// Mapping: lane l owns C consecutive columns (C = ceil(cols/64) <= 8, registers); the wave sweeps
// the rows as a skewed pipeline (lane l works on row t-l at step t) and hands the right edge of
// its block to lane l+1 with one DPP wave_shr:1 per step.  Wider matrices are processed in column
// tiles of 64*8 with the tile edge column staged in LDS.  Integer VALU + DPP only (no MFMA: a
// max-plus recurrence is not a dense contraction).
#pragma once
#include "bk_common.h"

#define BK_NW_PRIO_MASK 0x00030000
#define BK_NW_MATCH ((1 << 18) + (2 << 16))
#define BK_NW_MISM ((int)(((unsigned)-2) << 18) + (2 << 16))
#define BK_NW_CU ((int)(((unsigned)-2) << 18) + (1 << 16))     // S[i][j-1] + gap, pointer 2
#define BK_NW_CL ((int)(((unsigned)-2) << 18))                 // S[i-1][j] + gap, pointer 1
#define BK_NW_TILE_C 8
#define BK_NW_TILE_COLS (64 * BK_NW_TILE_C)

struct BkNwResult { int j_start, i_end, i_start, score; };

__device__ inline int bk_dpp_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // lane l <- lane l-1

// One column tile.  cols/rows: LDS byte arrays of base codes (0..3).  Columns j0+1 .. j0+mt.
// bound_in[i]  (i=1..n): word of cell (i, j0) from the previous tile (ignored when j0 == 0)
// bound_out[i] (i=1..n): word of cell (i, j0+mt), written when !last
// best_*: running end-cell selection on the last tile.
// Kept out of line: one function per C keeps the register budget of every caller at the C=8 size
// (inlining all eight variants into one kernel made the allocator spill: 232+ VGPRs).
template <int C>
__device__ __noinline__ int2 bk_nw_tile(const uint8_t *cols, const uint8_t *rows, int n, int j0, int mt,
                                        const int *bound_in, int *bound_out, bool last, int best_word, int best_i)
{
    const int lane = threadIdx.x & 63;
    const int lm = (mt - 1) / C, xm = (mt - 1) % C;
    int H[C]; int cb[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        int jj = lane * C + x;                          // 0-based column inside the tile
        H[x] = j0 + jj + 1;                             // row 0: score 0, origin = top border (0, j)
        cb[x] = jj < mt ? (int)cols[j0 + jj] : 4;       // 4 never matches
    }
    int dprev = j0 + lane * C;                          // cell (0, j0 + l*C): score 0, origin (0, j)
    int out_prev = 0;
    const int steps = n + lm;                           // lanes 0..lm
    for (int t = 0; t < steps; t++) {
        const int recv = bk_dpp_shr1(out_prev);
        const int i = t - lane + 1;                     // row handled by this lane at this step
        const bool active = (lane <= lm) && (i >= 1) && (i <= n);
        if (active) {
            int left_in;
            if (lane == 0) left_in = (j0 == 0) ? (0x8000 | i) : bound_in[i];
            else left_in = recv;
            const int rb = (int)rows[i - 1];
            int diag = dprev, u_in = left_in;
#pragma unroll
            for (int x = 0; x < C; x++) {
                int cd = diag + (cb[x] == rb ? BK_NW_MATCH : BK_NW_MISM);
                int cu = u_in + BK_NW_CU;
                int cl = H[x] + BK_NW_CL;
                int nv = max(max(cd, cu), cl) & ~BK_NW_PRIO_MASK;
                diag = H[x]; H[x] = nv; u_in = nv;
            }
            dprev = left_in;
            out_prev = H[C - 1];
            if (lane == lm) {
                int v = H[0];
#pragma unroll
                for (int x = 1; x < C; x++) if (x == xm) v = H[x];
                if (last) { if ((v >> 18) >= (best_word >> 18)) { best_word = v; best_i = i; } }   // olc.py:81 '>=': last row wins
                else bound_out[i] = v;
            }
        }
    }
    // hand the end-cell state to every lane
    if (last) { best_word = __shfl(best_word, lm); best_i = __shfl(best_i, lm); }
    return make_int2(best_word, best_i);
}

template <int C>
__device__ inline int2 bk_nw_tile_call(int c, const uint8_t *cols, const uint8_t *rows, int n, int j0, int mt,
                                       const int *bi, int *bo, bool last, int bw, int bidx)
{
    if (c == C) return bk_nw_tile<C>(cols, rows, n, j0, mt, bi, bo, last, bw, bidx);
    if constexpr (C < BK_NW_TILE_C) return bk_nw_tile_call<C + 1>(c, cols, rows, n, j0, mt, bi, bo, last, bw, bidx);
    return make_int2(bw, bidx);
}

// nw(seq1 = cols (m), seq2 = rows (n)) executed by the calling wave (all 64 lanes must call).
// bound: LDS scratch of 2*(n+1) ints, needed only when m > BK_NW_TILE_COLS.
__device__ inline BkNwResult bk_nw_wave(const uint8_t *cols, int m, const uint8_t *rows, int n, int *bound)
{
    int best_word = 0, best_i = 0;                      // row 0 of the last column: score 0 (olc.py:79-83)
    int *bi = bound, *bo = bound ? bound + (n + 1) : nullptr;
    for (int j0 = 0; j0 < m; j0 += BK_NW_TILE_COLS) {
        const int mt = min(m - j0, BK_NW_TILE_COLS);
        const bool last = j0 + mt >= m;
        const int c = (mt + 63) / 64;
        const int2 b = bk_nw_tile_call<1>(c, cols, rows, n, j0, mt, bi, bo, last, best_word, best_i);
        best_word = b.x; best_i = b.y;
        int *tswap = bi; bi = bo; bo = tswap;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    BkNwResult r;
    if (best_i == 0) {                                  // Q5: one forced traceback step from (0, m): pointer[0][m] = 2
        r.j_start = m - 1; r.i_end = 0; r.i_start = 0; r.score = 0;
    } else {
        const int org = best_word & 0xFFFF;
        r.score = best_word >> 18; r.i_end = best_i;
        if (org & 0x8000) { r.i_start = org & 0x7FFF; r.j_start = 0; } else { r.j_start = org; r.i_start = 0; }
    }
    return r;
}
