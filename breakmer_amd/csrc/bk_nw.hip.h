// bk_nw.hip.h -- olc.nw (olc.py:40-107) on one 64-lane wavefront, no pointer matrix.
//
// The reference fills an (n+1)x(m+1) score matrix and a pointer matrix, picks the end cell in the
// last column (largest row among equal maxima, olc.py:79-83) and walks the pointers back until a
// border is reached (olc.py:90-105).  Its callers only consume (j_start, i_end, i_start, score)
// -- the aligned strings stripped of '-' are exactly seq1[j_start:m] and seq2[i_start:i_end] -- so
// instead of storing pointers each cell carries the border cell its traceback would end in:
//
//     cell word = [ score : 14 signed | priority : 2 | origin : 16 ]
//     origin    = j            for the border row    (0, j)   of the reference matrix
//               = 0x8000 | i   for the border column (i, 0)
//
// One signed max3 over (diag + s, S[i][j-1] + gap, S[i-1][j] + gap) picks the predecessor with the
// reference's tie-break (diagonal 3 > S[i][j-1] 2 > S[i-1][j] 1, olc.py:69-74) because the 2-bit
// priority sits directly below the score; the origin rides along in the low bits and never
// influences the comparison (priorities are distinct).
//
// Mapping: lane l owns C consecutive "tile columns" (C = ceil(cols/64) <= 8, registers); the wave
// sweeps the "tile rows" as a skewed pipeline (lane l works on row t-l at step t) and hands the right
// edge of its block to lane l+1 with one DPP wave_shr:1 per step; the row symbol travels the same
// way (lane 0 picks it with a v_readlane from a register block, so there is no LDS access in the
// loop).  Both DPs of check_align put the CONTIG on the tile columns and the read on the tile rows:
//   TR = false : nw(contig, read) -- tile columns = reference columns (seq1);
//   TR = true  : nw(read, contig) -- the reference matrix transposed (tile columns = reference rows,
//                seq2): the two gap priorities swap places, the "last column" becomes the last tile
//                row, read out of the registers after the sweep.
// Wider contigs are processed in column tiles of 64*8 with the tile edge column staged in LDS.
// Integer VALU + DPP only (no MFMA: a max-plus recurrence is not a dense contraction).
#pragma once
#include "bk_common.h"
#undef BK_SRC_ID
#define BK_SRC_ID 3      // barrier sites of this file (bk_common.h: BK_SYNC)

#define BK_NW_PRIO_MASK 0x00030000
#define BK_NW_MATCH ((1 << 18) + (2 << 16))
#define BK_NW_MISM ((int)(((unsigned)-2) << 18) + (2 << 16))
#define BK_NW_G2 ((int)(((unsigned)-2) << 18) + (1 << 16))     // S[i][j-1] + gap, pointer 2
#define BK_NW_G1 ((int)(((unsigned)-2) << 18))                 // S[i-1][j] + gap, pointer 1
#define BK_NW_TILE_C 8
#ifndef BK_NW_STILE_C
#define BK_NW_STILE_C 13            // columns per lane of a column tile of the SCORE sweep (bk_nw_score_tile_c: 832 columns per tile).  A step costs ~24 + 3 C
                                    // instructions and a tile n + 63 steps whatever its width: wide tiles pay the step's overhead and the skew's fill / drain
                                    // fewer times (a 1,650-column translocation contig: 2 tiles of 13 instead of 4 of 8 columns per lane, 27 k instead of 41 k
                                    // instructions per read; round 6)
#endif
// "closed" left border: a score no path can recover from (14-bit signed score field; rows <= 1024 cost at most -2048 more)
#define BK_NW_CLOSED ((int)(((unsigned)-4096) << 18))
#define LEFTB_OF(TR) ((TR) ? 0 : 0x8000)
#define BK_NW_TILE_COLS (64 * BK_NW_TILE_C)

struct BkNwResult { int j_start, i_end, i_start, score; };

__device__ inline int bk_dpp_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }   // lane l <- lane l-1 (lane 0 keeps its own value)

// One column tile.  cols/rows: LDS byte arrays of base codes (0..3).  Tile columns j0+1 .. j0+mt, tile rows 1..n.
// bound_in[i]  (i=1..n): word of tile cell (i, j0) from the previous tile (ignored when j0 == 0)
// bound_out[i] (i=1..n): word of tile cell (i, j0+mt), written when !last
// best (x = word, y = index): running end-cell selection.
// Kept out of line: one function per (C, TR, XM) keeps the register budget of every caller at the C=8 size
// (inlining all variants into one kernel made the allocator spill: 232+ VGPRs).  XM = register that holds the
// last tile column in lane lm; a template parameter because a run-time select costs C-1 instructions per step
// (direct sweep only; the transposed sweep reads the end cells after the loop and uses XM = 0).
// Single-tile sweep (contig <= 512 columns: every sweep of the benchmark workload).  Same recurrence and the same
// results as bk_nw_tile below, with everything a one-tile matrix does not need removed from the loop: no tile-edge
// column, no loads/stores, no uniform branches; the neighbour's
// edge cell is read straight out of its H register, the row index is a running counter, the end-cell selection of the
// direct sweep is branch-free.  ~45 instructions per step at C = 4 (general variant: ~75).
template <int C, bool TR, int XM>
__device__ __noinline__ int2 bk_nw_tile_st(const uint8_t *cols, const uint8_t *rows, int n_, int mt_, int lb_)
{
    const int n = __builtin_amdgcn_readfirstlane(n_), mt = __builtin_amdgcn_readfirstlane(mt_);
    const int lbase = (LEFTB_OF(TR) + 1) + __builtin_amdgcn_readfirstlane(lb_);      // left border word of row 1 (lb: 0, or BK_NW_CLOSED)
    const int lane = threadIdx.x & 63;
    const int lm = (mt - 1) / C;
    constexpr int GH = TR ? BK_NW_G1 : BK_NW_G2, GV = TR ? BK_NW_G2 : BK_NW_G1;
    constexpr int TOPB = TR ? 0x8000 : 0;
    int H[C]; int cb[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = lane * C + x;
        H[x] = TOPB | (jj + 1);
        cb[x] = jj < mt ? (int)cols[jj] : 8;                // 8 never matches (codes: 0..3, N = 4: N matches N, olc.py:32-38)
    }
    int dprev = lane ? (TOPB | (lane * C)) : 0;
    int rb = 0;
    int im1 = -lane;                                    // tile row handled at this step, minus 1
    int best_word = 0, best_im1 = -1;                   // border cell of the last reference column: score 0 (olc.py:79-83)
    const bool inlanes = lane <= lm;
    const int steps = n + lm;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const int rblk = (t0 + lane < n) ? (int)rows[t0 + lane] : 0;       // row symbols of this block of 64 steps
        const int te = min(64, steps - t0);
        for (int tl = 0; tl < te; tl++) {
            const int recv = __builtin_amdgcn_mov_dpp(H[C - 1], 0x138, 0xf, 0xf, true);       // lane l <- lane l-1: its edge cell of the same tile row
            rb = __builtin_amdgcn_update_dpp(rb, rb, 0x138, 0xf, 0xf, false);
            { const int rb0 = __builtin_amdgcn_readlane(rblk, tl); if (lane == 0) rb = rb0; }       // rows[t] enters at lane 0
            if (inlanes && (unsigned)im1 < (unsigned)n) {
                const int left_in = lane == 0 ? lbase + im1 : recv;
                int cd[C], cv[C];
#pragma unroll
                for (int x = 0; x < C; x++) {
                    cd[x] = (x ? H[x - 1] : dprev) + (cb[x] == rb ? BK_NW_MATCH : BK_NW_MISM);
                    cv[x] = H[x] + GV;
                }
                int u_in = left_in;
#pragma unroll
                for (int x = 0; x < C; x++) {
                    const int nv = max(max(cd[x], u_in + GH), cv[x]) & ~BK_NW_PRIO_MASK;
                    H[x] = nv; u_in = nv;
                }
                dprev = left_in;
                if constexpr (!TR) {                    // every lane tracks its own column XM; only lane lm's is read (olc.py:81 '>=': last row wins)
                    const int v = H[XM];
                    const bool take = v >= (best_word & ~0x3FFFF);
                    best_word = take ? v : best_word; best_im1 = take ? im1 : best_im1;
                }
            }
            im1++;
        }
    }
    int best_i = best_im1 + 1;
    if (!TR) { best_word = __shfl(best_word, lm); best_i = __shfl(best_i, lm); }
    else {
        best_word = 0; best_i = 0;
#pragma unroll
        for (int x = 0; x < C; x++) {
            const int jj = lane * C + x;
            if (jj < mt && lane <= lm && (H[x] >> 18) >= (best_word >> 18)) { best_word = H[x]; best_i = jj + 1; }
        }
        for (int o = 1; o < 64; o <<= 1) {
            const int ow = __shfl_xor(best_word, o), oi = __shfl_xor(best_i, o);
            const int sc = best_word >> 18, os = ow >> 18;
            if (os > sc || (os == sc && oi > best_i)) { best_word = ow; best_i = oi; }
        }
    }
    return make_int2(best_word, best_i);
}
template <int C, bool TR, int XM>
__device__ inline int2 bk_nw_st_xm(int xm, const uint8_t *cols, const uint8_t *rows, int n, int mt, int lb)
{
    if (TR || xm == XM) return bk_nw_tile_st<C, TR, TR ? 0 : XM>(cols, rows, n, mt, lb);
    if constexpr (!TR && XM + 1 < C) return bk_nw_st_xm<C, TR, XM + 1>(xm, cols, rows, n, mt, lb);
    return make_int2(0, 0);
}
template <int C, bool TR>
__device__ inline int2 bk_nw_st_call(int c, const uint8_t *cols, const uint8_t *rows, int n, int mt, int lb)
{
    if (c == C) return bk_nw_st_xm<C, TR, 0>((mt - 1) % C, cols, rows, n, mt, lb);
    if constexpr (C < BK_NW_TILE_C) return bk_nw_st_call<C + 1, TR>(c, cols, rows, n, mt, lb);
    return make_int2(0, 0);
}

// ST = the whole matrix is one tile (contig <= 512: every sweep of the benchmark workload): no tile-edge column, no
// loads/stores and no uniform branches in the loop (the general variant is ~75 instructions per step at C = 4, this one ~50).
template <int C, bool TR, int XM, bool ST>
__device__ __noinline__ int2 bk_nw_tile(const uint8_t *cols, const uint8_t *rows, int n_, int j0_, int mt_,
                                        const int *bound_in, int *bound_out, bool last_, int best_word, int best_i, int lb_)
{
    const int lb = __builtin_amdgcn_readfirstlane(lb_);
    // the arguments of an out-of-line function arrive in VGPRs: tell the compiler they are wave-uniform so that
    // the loop control and the lane predicates stay on the scalar unit
    const int n = __builtin_amdgcn_readfirstlane(n_), j0 = ST ? 0 : __builtin_amdgcn_readfirstlane(j0_), mt = __builtin_amdgcn_readfirstlane(mt_);
    const bool last = ST ? true : __builtin_amdgcn_readfirstlane((int)last_) != 0;
    const int lane = threadIdx.x & 63;
    const int lm = (mt - 1) / C;
    // which reference neighbour is "horizontal" (previous tile column) / "vertical" (previous tile row)
    constexpr int GH = TR ? BK_NW_G1 : BK_NW_G2, GV = TR ? BK_NW_G2 : BK_NW_G1;
    constexpr int TOPB = TR ? 0x8000 : 0, LEFTB = TR ? 0 : 0x8000;   // origin tag of the tile's top / left border
    int H[C]; int cb[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = lane * C + x;                    // 0-based tile column
        H[x] = TOPB | (j0 + jj + 1);                    // tile row 0: score 0, origin = that border cell
        cb[x] = jj < mt ? (int)cols[j0 + jj] : 8;       // 8 never matches (N = 4 matches N)
    }
    int dprev = (j0 + lane * C) ? (TOPB | (j0 + lane * C)) : 0;      // tile cell (0, j0 + l*C); the corner (0,0) is origin 0
    int out_prev = 0, rb_prev = 0;
    int rblk = (lane < n) ? (int)rows[lane] : 0;        // row symbols t = 0..63
    const int steps = n + lm;                           // lanes 0..lm
    for (int t = 0; t < steps; t++) {
        const int tl = t & 63;
        if (tl == 0 && t) rblk = (t + lane < n) ? (int)rows[t + lane] : 0;
        const int rb0 = __builtin_amdgcn_readlane(rblk, tl);       // rows[t]: the symbol of tile row t+1, enters at lane 0
        const int recv = bk_dpp_shr1(out_prev);
        int rb = bk_dpp_shr1(rb_prev);
        const int i = t - lane + 1;                     // tile row handled by this lane at this step
        if (lane == 0) rb = rb0;
        rb_prev = rb;
        const bool active = (lane <= lm) && ((unsigned)(i - 1) < (unsigned)n);
        if (active) {
            int left_in;
            if constexpr (ST) left_in = lane == 0 ? (LEFTB | i) : recv;
            else {
                if (lane == 0) left_in = (j0 == 0) ? (LEFTB | i) + lb : bound_in[i];
                else left_in = recv;
            }
            // candidates that only need the previous tile row first (off the dependency chain), then the
            // serial chain along the row writes H[x] in place (no register shuffling at the loop end)
            int cd[C], cv[C];
#pragma unroll
            for (int x = 0; x < C; x++) {
                cd[x] = (x ? H[x - 1] : dprev) + (cb[x] == rb ? BK_NW_MATCH : BK_NW_MISM);
                cv[x] = H[x] + GV;
            }
            int u_in = left_in;
#pragma unroll
            for (int x = 0; x < C; x++) {
                const int nv = max(max(cd[x], u_in + GH), cv[x]) & ~BK_NW_PRIO_MASK;
                H[x] = nv; u_in = nv;
            }
            dprev = left_in;
            out_prev = H[C - 1];
            if constexpr (ST) {
                if constexpr (!TR) {                    // last tile column of this tile row, branch-free (olc.py:81 '>=': last row wins)
                    const int v = H[XM];
                    const bool take = (lane == lm) && ((v >> 18) >= (best_word >> 18));
                    best_word = take ? v : best_word; best_i = take ? i : best_i;
                }
            } else if (lane == lm && (!TR || !last)) {  // last tile column of this tile row
                int v = H[XM];
                if constexpr (TR) {                     // transposed sweep, not the last tile (contig > 512): rare, run-time select
                    const int xm = (mt - 1) % C;
#pragma unroll
                    for (int x = 1; x < C; x++) if (x == xm) v = H[x];
                }
                if (!TR && last) { if ((v >> 18) >= (best_word >> 18)) { best_word = v; best_i = i; } }   // olc.py:81 '>=': last row wins
                else bound_out[i] = v;
            }
        }
    }
    if (!TR) {
        if (last) { best_word = __shfl(best_word, lm); best_i = __shfl(best_i, lm); }
    } else {
        // reference last column = last tile row: scan tile columns in ascending order, '>=' keeps the largest index
#pragma unroll
        for (int x = 0; x < C; x++) {
            const int jj = lane * C + x;
            if (jj < mt && lane <= lm && (H[x] >> 18) >= (best_word >> 18)) { best_word = H[x]; best_i = j0 + jj + 1; }
        }
        // wave reduction: max score, then largest index (indices are unique, lanes own ascending ranges)
        for (int o = 1; o < 64; o <<= 1) {
            const int ow = __shfl_xor(best_word, o), oi = __shfl_xor(best_i, o);
            const int s = best_word >> 18, os = ow >> 18;
            if (os > s || (os == s && oi > best_i)) { best_word = ow; best_i = oi; }
        }
    }
    return make_int2(best_word, best_i);
}

template <int C, bool TR, int XM, bool ST>
__device__ inline int2 bk_nw_tile_xm(int xm, const uint8_t *cols, const uint8_t *rows, int n, int j0, int mt,
                                     const int *bi, int *bo, bool last, int bw, int bidx, int lb)
{
    if (TR || xm == XM) return bk_nw_tile<C, TR, TR ? 0 : XM, ST>(cols, rows, n, j0, mt, bi, bo, last, bw, bidx, lb);
    if constexpr (!TR && XM + 1 < C) return bk_nw_tile_xm<C, TR, XM + 1, ST>(xm, cols, rows, n, j0, mt, bi, bo, last, bw, bidx, lb);
    return make_int2(bw, bidx);
}
template <int C, bool TR, bool ST>
__device__ inline int2 bk_nw_tile_call(int c, const uint8_t *cols, const uint8_t *rows, int n, int j0, int mt,
                                       const int *bi, int *bo, bool last, int bw, int bidx, int lb)
{
    if (c == C) return bk_nw_tile_xm<C, TR, 0, ST>((mt - 1) % C, cols, rows, n, j0, mt, bi, bo, last, bw, bidx, lb);
    if constexpr (C < BK_NW_TILE_C) return bk_nw_tile_call<C + 1, TR, ST>(c, cols, rows, n, j0, mt, bi, bo, last, bw, bidx, lb);
    return make_int2(bw, bidx);
}

// Executed by the calling wave (all 64 lanes).  `tcols` (length m) goes on the tile columns, `trows` (length n) on
// the tile rows.  TR = false computes olc.nw(seq1 = tcols, seq2 = trows); TR = true computes
// olc.nw(seq1 = trows, seq2 = tcols).  bound: LDS scratch of 2*(n+1) ints, needed only when m > BK_NW_TILE_COLS.
// lb = BK_NW_CLOSED closes the border column of the reference matrix (direct sweep only): bk_nw_suffix below.
template <bool TR>
__device__ inline BkNwResult bk_nw_wave(const uint8_t *tcols, int m, const uint8_t *trows, int n, int *bound, int lb = 0)
{
    int best_word = 0, best_i = 0;                      // border cell of the last reference column: score 0 (olc.py:79-83)
    if (m <= BK_NW_TILE_COLS) {
        const int2 b = bk_nw_st_call<1, TR>((m + 63) / 64, tcols, trows, n, m, lb);
        best_word = b.x; best_i = b.y;
    } else {
        int *bi = bound, *bo = bound ? bound + (n + 1) : nullptr;
        for (int j0 = 0; j0 < m; j0 += BK_NW_TILE_COLS) {
            const int mt = min(m - j0, BK_NW_TILE_COLS);
            const bool last = j0 + mt >= m;
            const int c = (mt + 63) / 64;
            const int2 b = bk_nw_tile_call<1, TR, false>(c, tcols, trows, n, j0, mt, bi, bo, last, best_word, best_i, lb);
            best_word = b.x; best_i = b.y;
            int *tswap = bi; bi = bo; bo = tswap;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int m_ref = TR ? n : m;                       // len(seq1) of the reference call
    BkNwResult r;
    if (best_i == 0) {                                  // Q5: one forced traceback step from (0, m): pointer[0][m] = 2
        r.j_start = m_ref - 1; r.i_end = 0; r.i_start = 0; r.score = 0;
    } else {
        const int org = best_word & 0xFFFF;
        r.score = best_word >> 18; r.i_end = best_i;
        if (org & 0x8000) { r.i_start = org & 0x7FFF; r.j_start = 0; } else { r.j_start = org; r.i_start = 0; }
    }
    return r;
}

// olc.nw(seq1 = contig, seq2 = read) needs only the last floor(1.5 n) + 2 columns of a long contig.  Proof: the end cell
// lies in the last column and scores >= 0 (the border cell (0, m) does).  A path that spans L columns over at most n rows
// has a <= n diagonal steps and >= L - a column gaps, so it scores <= a - 2 (L - a) <= 3 n - 2 L, which is negative for
// L > 1.5 n: the traceback path of the end cell -- a maximum-score path to it -- starts on the top border at a column
// >= m - 1.5 n.  Every cell ON that path has the same value in the matrix restricted to the last K columns with the
// cut column closed (no path may start there), all other cells can only score lower there, so the end-cell rule
// (largest row among equal maxima) and every pointer decision along the path (priority among equal candidates) come out
// the same.  Columns before m - K are never computed.
__device__ inline BkNwResult bk_nw_suffix(const uint8_t *contig, int m, const uint8_t *read, int n, int *bound)
{
    const int K = n + (n >> 1) + 2;
    if (m <= K) return bk_nw_wave<false>(contig, m, read, n, bound, 0);
    const int off = m - K;
    BkNwResult r = bk_nw_wave<false>(contig + off, K, read, n, bound, BK_NW_CLOSED);
    r.j_start += off;                                   // top-border origins (and the forced step of Q5) are column numbers
    return r;
}

// ---- both overlap DPs of check_align (sv_assembly.py:451-452) on ONE wavefront ----------------------------------------
// Lanes 0..31 run the direct sweep of nw(contig[off1:], read) (closed cut column when off1 > 0, see bk_nw_suffix), lanes
// 32..63 the transposed sweep of nw(read, contig): the same row symbols (the read) enter both halves at the same step, so
// the skew is 31 lanes instead of 63 (fill/drain 17 % of the steps instead of 28 %) and a candidate read needs one
// wavefront instead of two.  Each lane owns C = ceil(contig / 32) <= BK_NW_DUAL_C columns.  The columns are RIGHT-aligned
// in the lanes of their half: the last column is always register C-1 of the last lane (no run-time register select for
// the end-cell rule of the direct sweep); the `pad` unused registers at the left of lane 0 hold a symbol that matches
// nothing and take a horizontal gap constant of 0, so each row's border word simply travels through them.
#define BK_NW_DUAL_C 10
#define BK_NW_DUAL_COLS (32 * BK_NW_DUAL_C)
extern __shared__ __attribute__((aligned(16))) uint8_t bk_dyn_lds[];      // the dynamic LDS block of the running kernel
template <int C>
__device__ __noinline__ void bk_nw_dual_c(int contig_off, int clen_, int off1_, int rows_off, int n_, int res_off /* 8 ints in LDS: v1, v2 */)
{
    // sequences and results are addressed as offsets into the dynamic LDS block: plain LDS instructions (a generic pointer
    // makes every access a FLAT one, whose completion order against s_barrier proved fragile for the result stores)
    const uint8_t *contig = bk_dyn_lds + contig_off, *rows = bk_dyn_lds + rows_off;
    int *res = (int *)(bk_dyn_lds + res_off);
    const int clen = __builtin_amdgcn_readfirstlane(clen_), off1 = __builtin_amdgcn_readfirstlane(off1_), n = __builtin_amdgcn_readfirstlane(n_);
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const bool tr = lane >= 32;                                        // half 1: transposed sweep (v2)
    const int mt = tr ? clen : clen - off1;                            // columns of this half
    const uint8_t *cols = tr ? contig : contig + off1;
    const int nl = (mt + C - 1) / C, lm = nl - 1, pad = nl * C - mt;
    const int lm_max = max((clen - off1 + C - 1) / C, (clen + C - 1) / C) - 1;      // wave-uniform
    const int GH = tr ? BK_NW_G1 : BK_NW_G2, GV = tr ? BK_NW_G2 : BK_NW_G1;
    const int TOPB = tr ? 0x8000 : 0, LEFTB = tr ? 0 : 0x8000;
    const int lbase = (LEFTB + 1) + ((!tr && off1 > 0) ? BK_NW_CLOSED : 0);
    int H[C], cb[C], gh[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = hl * C + x - pad;                               // 0-based column, < 0: padding
        const bool real = jj >= 0 && jj < mt && hl <= lm;
        H[x] = real ? (TOPB | (jj + 1)) : 0;                           // row 0: the border cell (0, jj+1); padding holds the corner
        cb[x] = real ? (int)cols[jj] : 8;
        gh[x] = jj >= 0 ? GH : 0;
    }
    int dprev = (hl * C - pad) > 0 ? (TOPB | (hl * C - pad)) : 0;
    int rb = 0, im1 = -hl;
    int best_word = 0, best_im1 = -1;
    const bool inl = hl <= lm;
    const int steps = n + lm_max;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const int rblk = (t0 + lane < n) ? (int)rows[t0 + lane] : 0;
        const int te = min(64, steps - t0);
        for (int tl = 0; tl < te; tl++) {
            const int recv = __builtin_amdgcn_mov_dpp(H[C - 1], 0x138, 0xf, 0xf, true);
            rb = __builtin_amdgcn_update_dpp(rb, rb, 0x138, 0xf, 0xf, false);
            { const int rb0 = __builtin_amdgcn_readlane(rblk, tl); if (hl == 0) rb = rb0; }
            if (inl && (unsigned)im1 < (unsigned)n) {
                const int left_in = hl == 0 ? lbase + im1 : recv;
                int cd[C], cv[C];
#pragma unroll
                for (int x = 0; x < C; x++) {
                    cd[x] = (x ? H[x - 1] : dprev) + (cb[x] == rb ? BK_NW_MATCH : BK_NW_MISM);
                    cv[x] = H[x] + GV;
                }
                int u_in = left_in;
#pragma unroll
                for (int x = 0; x < C; x++) {
                    const int nv = max(max(cd[x], u_in + gh[x]), cv[x]) & ~BK_NW_PRIO_MASK;
                    H[x] = nv; u_in = nv;
                }
                dprev = left_in;
                const int v = H[C - 1];                                // direct half, lane lm: the last column (olc.py:81 '>=': last row wins)
                const bool take = v >= (best_word & ~0x3FFFF);
                best_word = take ? v : best_word; best_im1 = take ? im1 : best_im1;
            }
            im1++;
        }
    }
    // direct half: lane lm of half 0
    const int lm1 = (clen - off1 + C - 1) / C - 1;
    const int w1 = __shfl(best_word, lm1), i1 = __shfl(best_im1, lm1) + 1;
    // transposed half: the last tile row, scanned in ascending column order ('>=' keeps the largest index), then reduced
    int w2 = (int)0x80000000, i2 = 0;
    if (tr) {
        w2 = 0; i2 = 0;
#pragma unroll
        for (int x = 0; x < C; x++) {
            const int jj = hl * C + x - pad;
            if (jj >= 0 && jj < mt && hl <= lm && (H[x] >> 18) >= (w2 >> 18)) { w2 = H[x]; i2 = jj + 1; }
        }
    }
    for (int o = 1; o < 32; o <<= 1) {
        const int ow = __shfl_xor(w2, o), oi = __shfl_xor(i2, o);
        const int sc = w2 >> 18, os = ow >> 18;
        if (os > sc || (os == sc && oi > i2)) { w2 = ow; i2 = oi; }
    }
    w2 = __shfl(w2, 32); i2 = __shfl(i2, 32);
    if (lane == 0) {
        // v1 = nw(contig, read): m_ref = clen; v2 = nw(read, contig): m_ref = n
        if (i1 == 0) { res[0] = clen - 1; res[1] = 0; res[2] = 0; res[3] = 0; }
        else { const int org = w1 & 0xFFFF; res[3] = w1 >> 18; res[1] = i1; if (org & 0x8000) { res[2] = org & 0x7FFF; res[0] = 0; } else { res[0] = org + off1; res[2] = 0; } }
        if (i2 == 0) { res[4] = n - 1; res[5] = 0; res[6] = 0; res[7] = 0; }
        else { const int org = w2 & 0xFFFF; res[7] = w2 >> 18; res[5] = i2; if (org & 0x8000) { res[6] = org & 0x7FFF; res[4] = 0; } else { res[4] = org; res[6] = 0; } }
    }
}
// ---- ONE score matrix per read, both tie-break orders; TWO reads per wavefront ------------------------------------------
// nw(read, contig) is nw(contig, read) transposed: the two reference matrices hold the SAME scores, only the pointers differ
// where candidates tie (olc.py:69-74 in each call's own orientation: with the contig on the columns, nw(contig, read) prefers
// diagonal > S[i][j-1] > S[i-1][j], nw(read, contig) prefers diagonal > S[i-1][j] > S[i][j-1]).  So a cell is computed once: the
// word [score | priority | origin] follows the first order as in every sweep of this file, and a second register per cell
// carries the border cell the traceback of the OTHER order would reach -- of the candidates that reach the cell's score:
// the diagonal one, else the vertical one, else the horizontal one (two compares against the score, two selects).  End
// cells: nw(contig, read) the last column (tracked per step), nw(read, contig) the last row (read out of the registers).
// 12 instructions per cell for both DPs instead of 2 x 7, no column of the contig is swept twice, and with half a wavefront
// per read (32 lanes, right-aligned columns as in bk_nw_dual_c) a wavefront aligns TWO reads of a round.  For the assembler's
// throughput mode, where the SIMDs are bound by instruction issue; a round alone finishes sooner with bk_nw_dual_c (one read
// per wavefront).  par_off: 10 ints in LDS, per half {contig offset, contig length, read offset, read length, result offset}
// (offsets into the dynamic LDS block; read length 0 = this half is idle); results: v1 then v2 as (j_start, i_end, i_start, score).
struct BkPairArgs { int contig, clen, read, n, res; };      // one half: offsets into the dynamic LDS block; n = 0: the half is idle
template <int C>
__device__ __noinline__ void bk_nw_pair_c(BkPairArgs A_, BkPairArgs B_)
{
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    // the arguments of an out-of-line function arrive in vector registers: make them wave-uniform again, then every lane takes its half's
    const int a0 = __builtin_amdgcn_readfirstlane(A_.contig), a1 = __builtin_amdgcn_readfirstlane(A_.clen), a2 = __builtin_amdgcn_readfirstlane(A_.read), a3 = __builtin_amdgcn_readfirstlane(A_.n), a4 = __builtin_amdgcn_readfirstlane(A_.res);
    const int b0_ = __builtin_amdgcn_readfirstlane(B_.contig), b1 = __builtin_amdgcn_readfirstlane(B_.clen), b2 = __builtin_amdgcn_readfirstlane(B_.read), b3 = __builtin_amdgcn_readfirstlane(B_.n), b4 = __builtin_amdgcn_readfirstlane(B_.res);
    const uint8_t *cols = bk_dyn_lds + (half ? b0_ : a0), *rows = bk_dyn_lds + (half ? b2 : a2);
    const int mt = half ? b1 : a1, n = half ? b3 : a3;
    int *res = (int *)(bk_dyn_lds + (half ? b4 : a4));
    const int nmax = max(a3, b3);
    const int nl = (mt + C - 1) / C, lm = nl - 1, pad = nl * C - mt;
    const int lm_max = max((a1 + C - 1) / C, (b1 + C - 1) / C) - 1;
    int H[C], O2[C], cb[C], gh[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = hl * C + x - pad;                                            // 0-based column, < 0: padding
        const bool real = jj >= 0 && jj < mt && hl <= lm;
        H[x] = real ? (jj + 1) : 0;                                                 // row 0: score 0, origin (first order) = column j; padding: the corner
        O2[x] = real ? (0x8000 | (jj + 1)) : 0;                                     // the same border cell in the second order's encoding
        cb[x] = real ? (int)cols[jj] : 8;
        gh[x] = jj >= 0 ? BK_NW_G2 : 0;
    }
    const int c0 = hl * C - pad;
    int dprev1 = c0 > 0 ? c0 : 0, dprev2 = c0 > 0 ? (0x8000 | c0) : 0;
    int rb = 0, im1 = -hl;
    int best_word = 0, best_im1 = -1;
    const bool inl = hl <= lm && n > 0;
    const int steps = nmax + lm_max;
    for (int t0 = 0; t0 < steps; t0 += 32) {
        const int rblk = (t0 + hl < n) ? (int)rows[t0 + hl] : 0;                     // lanes 0..31: symbols of read A for these 32 steps, lanes 32..63: of read B
        const int te = min(32, steps - t0);
        for (int tl = 0; tl < te; tl++) {
            const int recv1 = __builtin_amdgcn_mov_dpp(H[C - 1], 0x138, 0xf, 0xf, true);
            const int recv2 = __builtin_amdgcn_mov_dpp(O2[C - 1], 0x138, 0xf, 0xf, true);
            rb = __builtin_amdgcn_update_dpp(rb, rb, 0x138, 0xf, 0xf, false);
            { const int ra = __builtin_amdgcn_readlane(rblk, tl), rbb = __builtin_amdgcn_readlane(rblk, 32 + tl); if (hl == 0) rb = half ? rbb : ra; }
            if (inl && (unsigned)im1 < (unsigned)n) {
                const int left1 = hl == 0 ? (0x8000 + 1) + im1 : recv1;              // border column (i, 0): score 0, origin 0x8000 | i
                const int left2 = hl == 0 ? im1 + 1 : recv2;                        // ... = i in the second order's encoding
                int u_in = left1, o_in = left2, hprev = dprev1, oprev = dprev2;        // this row's left neighbour; the previous row's (x-1) cell
#pragma unroll
                for (int x = 0; x < C; x++) {
                    const int hold = H[x], oold = O2[x];
                    const int cd = hprev + (cb[x] == rb ? BK_NW_MATCH : BK_NW_MISM);
                    const int cu = hold + BK_NW_G1;
                    const int nv = max(max(cd, u_in + gh[x]), cu) & ~BK_NW_PRIO_MASK;
                    const int nvs = nv & ~0x3FFFF;                                  // the cell's score; a candidate word >= it reaches that score
                    const int t = cu >= nvs ? oold : o_in;                          // second order: vertical before horizontal ...
                    const int o2 = cd >= nvs ? oprev : t;                           // ... and the diagonal before both
                    H[x] = nv; O2[x] = o2; u_in = nv; o_in = o2; hprev = hold; oprev = oold;
                }
                dprev1 = left1; dprev2 = left2;
                const int v = H[C - 1];                                            // lane lm: the last column (olc.py:81 '>=': last row wins)
                const bool take = v >= (best_word & ~0x3FFFF);
                best_word = take ? v : best_word; best_im1 = take ? im1 : best_im1;
            }
            im1++;
        }
    }
    // nw(contig, read): lane lm of the half
    const int w1 = __shfl(best_word, (half << 5) + lm), i1 = __shfl(best_im1, (half << 5) + lm) + 1;
    // nw(read, contig): the last row, columns ascending ('>=' keeps the largest index), reduced over the half
    int w2 = 0, i2 = 0, o2s = 0;
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = hl * C + x - pad;
        if (jj >= 0 && jj < mt && inl && (H[x] >> 18) >= (w2 >> 18)) { w2 = H[x]; i2 = jj + 1; o2s = O2[x]; }
    }
    for (int o = 1; o < 32; o <<= 1) {
        const int ow = __shfl_xor(w2, o), oi = __shfl_xor(i2, o), oo = __shfl_xor(o2s, o);
        const int sc = w2 >> 18, os = ow >> 18;
        if (os > sc || (os == sc && oi > i2)) { w2 = ow; i2 = oi; o2s = oo; }
    }
    if (hl == 0 && n > 0) {
        if (i1 == 0) { res[0] = mt - 1; res[1] = 0; res[2] = 0; res[3] = 0; }
        else { const int org = w1 & 0xFFFF; res[3] = w1 >> 18; res[1] = i1; if (org & 0x8000) { res[2] = org & 0x7FFF; res[0] = 0; } else { res[0] = org; res[2] = 0; } }
        if (i2 == 0) { res[4] = n - 1; res[5] = 0; res[6] = 0; res[7] = 0; }
        else { const int org = o2s & 0xFFFF; res[7] = w2 >> 18; res[5] = i2; if (org & 0x8000) { res[6] = org & 0x7FFF; res[4] = 0; } else { res[4] = org; res[6] = 0; } }
    }
}
template <int C>
__device__ inline void bk_nw_pair_call(int c, const BkPairArgs &A, const BkPairArgs &B)
{
    if (c <= C) { bk_nw_pair_c<C>(A, B); return; }
    if constexpr (C < BK_NW_DUAL_C) bk_nw_pair_call<C + 1>(c, A, B);
}
// both contigs <= BK_NW_DUAL_COLS; registers per lane = ceil(longer contig / 32)
__device__ inline void bk_nw_pair(const BkPairArgs &A, const BkPairArgs &B) { bk_nw_pair_call<3>((max(A.clen, B.n ? B.clen : 0) + 31) / 32, A, B); }

// ---- round 5: the SCORE sweep -- one plain score matrix per read, no pointers, no origins ---------------------------------------
// What check_align consumes of an overlap DP is (j_start, i_end, i_start, score): the END CELL of the traceback (largest row among the
// maxima of the last column, olc.py:79-83) and the border cell the traceback reaches.  The end cells of BOTH calls -- nw(contig, read):
// last column; nw(read, contig): last row of the same matrix, largest column among the maxima -- need the scores only: 3 instructions
// per cell (bit-field extract, add3, max3: see bk_nw_score_c; 6 in the sweep's first version) instead of the 12 of bk_nw_pair_c.
// The border cell then follows WITHOUT a traceback in the case clean data consists of:
//   Let the end cell be (i, m) with score s, d = min(i, m) the most diagonal steps any path into it can have.  A path with a matches,
//   x mismatches and g gaps scores a - 2x - 2g <= d - 2g.  s == d  =>  a = d, x = g = 0: the ONLY path that reaches the score is the
//   pure diagonal of d matches from the border -- and along it the diagonal candidate wins every cell strictly (a gap candidate would
//   need a neighbour scoring 2 more than the cell, which exceeds that neighbour's own bound min(i, j)), so the reference's pointer
//   walk follows it whatever the tie-break order: origin = (0, m - i) if i <= m, else (i - m, 0).
// The same holds with substitutions on that diagonal: if the pure diagonal from the border to the end cell, with its x mismatches, scores
// d - 3x == s, it is an optimal path, so every prefix of it is optimal for the cell it ends in (a better prefix would make a better
// path), i.e. at every cell on it the diagonal candidate reaches the cell's score -- and the diagonal has the highest priority in both
// tie-break orders (olc.py:69-74): the pointer walk follows it to the border.  x is counted on the sequences (d comparisons, the
// lanes of the read in parallel); this settles the overlaps of reads with sequencing errors, whose alignments have no gaps.
// Otherwise the origin is only needed when the score can pass check_align's first test (4 * score >= min(len(contig), len(read)),
// sv_assembly.py:459-461): else the decision never looks at it (bk_decide: ok_k false; a call whose partner wins is only read for
// its score).  What is left -- an overlap with a mismatch or an indel in it -- is FLAGGED (j_start = -1) and the caller runs the full
// sweep (bk_nw_pair_c / bk_nw_dual_c) for that read: bit-identical by construction, since every value reported here IS the
// reference's.  LPR = lanes per read: 64 (one read per wavefront, the latency build) or 32 (two reads per wavefront).
#define BK_NW_NEEDS_DP (-1)
// mismatches on the d cells of the diagonal that ends in matrix cell (row ie, column je): read symbol rows[ie - 1 - t] against contig
// symbol cols[je - 1 - t]; executed by the LPR lanes that hold the read (hl = lane within them), result in all of them
template <int LPR>
__device__ inline int bk_nw_diag_mismatches(const uint8_t *cols, const uint8_t *rows, int ie, int je, int d, int hl)
{
    int c = 0;
    for (int t = hl; t < d; t += LPR) c += rows[ie - 1 - t] != cols[je - 1 - t] ? 1 : 0;
    for (int o = 1; o < LPR; o <<= 1) c += __shfl_xor(c, o);
    return c;
}
__device__ inline void bk_nw_score_results(int *res, int mt, int n, int s1, int i1, int s2, int i2, int x1, int x2, int force);
template <int C, int LPR>
__device__ __noinline__ void bk_nw_score_c(BkPairArgs A_, BkPairArgs B_, int force_)
{
    const int lane = threadIdx.x & 63, hl = lane & (LPR - 1), half = LPR == 32 ? lane >> 5 : 0;
    const int a0 = __builtin_amdgcn_readfirstlane(A_.contig), a1 = __builtin_amdgcn_readfirstlane(A_.clen), a2 = __builtin_amdgcn_readfirstlane(A_.read), a3 = __builtin_amdgcn_readfirstlane(A_.n), a4 = __builtin_amdgcn_readfirstlane(A_.res);
    const int b0_ = __builtin_amdgcn_readfirstlane(B_.contig), b1 = __builtin_amdgcn_readfirstlane(B_.clen), b2 = __builtin_amdgcn_readfirstlane(B_.read), b3 = __builtin_amdgcn_readfirstlane(B_.n), b4 = __builtin_amdgcn_readfirstlane(B_.res);
    const int force = __builtin_amdgcn_readfirstlane(force_);          // (tests: every origin is wanted, whatever the score)
    const uint8_t *cols = bk_dyn_lds + (half ? b0_ : a0), *rows = bk_dyn_lds + (half ? b2 : a2);
    const int mt = half ? b1 : a1, n = half ? b3 : a3;
    int *res = (int *)(bk_dyn_lds + (half ? b4 : a4));
    const int nmax = LPR == 32 ? max(a3, b3) : a3;
    const int nl = (mt + C - 1) / C, lm = nl - 1, pad = nl * C - mt;
    const int lm_max = (LPR == 32 ? max((a1 + C - 1) / C, (b1 + C - 1) / C) : (a1 + C - 1) / C) - 1;
    // The cells hold V(i, j) = S(i, j) + 2 (i + j) instead of the score S: a gap move then adds NOTHING (S - 2 one cell further) and
    // the diagonal adds 4 + (1 | -2) = 2 + 3 [match].  All three candidates of a cell carry the same offset, so the maximum picks the
    // same one: S = V - 2 (i + j) exactly, taken where a score is read (the last column's running maximum, the last row at the end).
    // The row's symbol travels as the word 3 << 4 * code, a column keeps 4 * code: one bit-field extract yields 3 [match] or 0, and a
    // cell is v_bfe_u32, v_add3_u32, v_max3_i32 -- 3 instructions where the first version of the sweep had 6, and no condition code.
    // (a mismatch on the diagonal adds -2 + 4 = 2.  The padding stands for the border column 0, V(i, 0) = 2 i: there a "diagonal" step of + 2 IS
    //  the step down the border -- 2 (i - 1) + 2 = 2 i, what the left neighbour hands in anyway --, so the 2 is a literal of every column and the
    //  per-column register of rounds 5 is gone)
    int H[C], cb[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = hl * C + x - pad;                                            // 0-based column, < 0: padding (right-aligned columns: the last one is register C-1 of lane lm)
        const bool real = jj >= 0 && jj < mt && hl <= lm;
        H[x] = jj >= 0 ? 2 * (jj + 1) : 0;                                          // row 0: score 0 in matrix column jj + 1; the padding stands for column 0
        cb[x] = real ? 4 * (int)cols[jj] : 28;                                      // the column's symbol as a nibble position (28: no symbol sits there, matches nothing)
    }
    int dprev = 2 * max(hl * C - pad, 0), rb = 0, im1 = -hl;                        // V(0, column left of this lane's first)
    // The end cell of the last column (olc.py:79-83: the LARGEST row among the maxima) as ONE word per lane: key = score << 11 | row, so that a plain
    // signed maximum keeps the later row among equal scores (rows <= 1,024; a negative score makes a negative key, which never beats the border
    // cell (0, m): score 0, row 0 = key 0).  The last column's score is V - 2 (row + m): key = (V << 11) + koff with koff = row - (2 (row + m) << 11),
    // which moves by 1 - 4096 per step -- shift-add, max and the counter: 3 instructions per step where the compare-and-two-selects form of
    // round 5 took 5 (round 6).
    int koff = (1 - hl) - ((2 * (1 - hl + mt)) << 11);
    int bkey = 0;
    const bool inl = hl <= lm && n > 0;
    const int steps = nmax + lm_max;
    for (int t0 = 0; t0 < steps; t0 += LPR) {
        // The symbols of the next LPR rows, one per lane, already as the word the cells test (3 << 4 code).  They rotate TOWARDS the read's first
        // lane by one lane per step (DPP wave_shl:1), so that lane always holds the word of the row that enters the pipeline now; the row
        // words themselves move down the lanes by one (wave_shr:1) and the first lane takes the new one: three instructions per step -- two
        // DPP moves and one select on a loop-invariant mask -- for one read or two (round 5: v_readlane, shift, move and select per read, nine
        // for the pair).  A block ends after LPR steps, before what enters the rotation from the neighbouring half could reach a first lane.
        int rrot = 3 << (4 * ((t0 + hl < n) ? (int)rows[t0 + hl] : 0));
        const int te = __builtin_amdgcn_readfirstlane(min(LPR, steps - t0));          // (uniform: the step counter stays in a scalar register)
        for (int tl = 0; tl < te; tl++) {
            const int recv = __builtin_amdgcn_mov_dpp(H[C - 1], 0x138, 0xf, 0xf, true);
            rb = __builtin_amdgcn_update_dpp(rb, rb, 0x138, 0xf, 0xf, false);
            rb = hl == 0 ? rrot : rb;
            rrot = __builtin_amdgcn_update_dpp(rrot, rrot, 0x130, 0xf, 0xf, false);
            if (inl && (unsigned)im1 < (unsigned)n) {
                const int left = hl == 0 ? 2 * im1 + 2 : recv;                      // border column (i, 0): score 0, V = 2 i
                // the diagonal candidates of the whole row first (from the previous row's values), then the chain of max3 IN PLACE: left
                // to itself the compiler interleaves them and pays a register copy per column to put the new values back where the
                // loop carries them
                int dg[C];
                dg[0] = dprev + (int)__builtin_amdgcn_ubfe((unsigned)rb, (unsigned)cb[0], 4u) + 2;
#pragma unroll
                for (int x = 1; x < C; x++) dg[x] = H[x - 1] + (int)__builtin_amdgcn_ubfe((unsigned)rb, (unsigned)cb[x], 4u) + 2;
                __builtin_amdgcn_sched_barrier(0);
                int u_in = left;
#pragma unroll
                for (int x = 0; x < C; x++) { H[x] = max(max(dg[x], u_in), H[x]); u_in = H[x]; }
                dprev = left;
                bkey = max(bkey, (H[C - 1] << 11) + koff);                          // lane lm: the last column's (score, row); '>=' of olc.py:81 = the later row wins
            }
            im1++; koff -= 4095;
        }
    }
    const int bk1 = __shfl(bkey, (half * LPR) + lm);
    const int s1 = bk1 >> 11, i1 = bk1 & 2047;
    // nw(read, contig): the last row, columns ascending ('>=' keeps the largest index), reduced over the read's lanes
    int s2 = 0, i2 = 0;
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = hl * C + x - pad, sc = H[x] - 2 * (n + jj + 1);             // the score of (n, jj + 1)
        if (jj >= 0 && jj < mt && inl && sc >= s2) { s2 = sc; i2 = jj + 1; }
    }
    for (int o = 1; o < LPR; o <<= 1) {
        const int os = __shfl_xor(s2, o), oi = __shfl_xor(i2, o);
        if (os > s2 || (os == s2 && oi > i2)) { s2 = os; i2 = oi; }
    }
    // the end cells' diagonals (uniform per read: s, i are): v1 ends in (row i1, column mt), v2 in (row n, column i2)
    int x1 = 0, x2 = 0;
    if (n > 0 && i1 > 0 && s1 != min(i1, mt)) x1 = bk_nw_diag_mismatches<LPR>(cols, rows, i1, mt, min(i1, mt), hl);
    if (n > 0 && i2 > 0 && s2 != min(i2, n)) x2 = bk_nw_diag_mismatches<LPR>(cols, rows, n, i2, min(i2, n), hl);
    if (hl == 0 && n > 0) bk_nw_score_results(res, mt, n, s1, i1, s2, i2, x1, x2, force);
}
template <int C, int LPR>
__device__ inline void bk_nw_score_call(int c, const BkPairArgs &A, const BkPairArgs &B, int force)
{
    if (c <= C) { bk_nw_score_c<C, LPR>(A, B, force); return; }
    if constexpr (C < BK_NW_DUAL_C) bk_nw_score_call<C + 1, LPR>(c, A, B, force);
}
// two reads per wavefront (both contigs <= BK_NW_DUAL_COLS) / one read per wavefront (contig <= BK_NW_DUAL_COLS: C <= 5)
__device__ inline void bk_nw_score_pair(const BkPairArgs &A, const BkPairArgs &B, int force = 0) { bk_nw_score_call<3, 32>((max(A.clen, B.n ? B.clen : 0) + 31) / 32, A, B, force); }
__device__ inline void bk_nw_score_one(const BkPairArgs &A, int force = 0) { BkPairArgs B; B.contig = 0; B.clen = 0; B.read = 0; B.n = 0; B.res = 0; bk_nw_score_call<2, 64>((A.clen + 63) / 64, A, B, force); }

// ---- the score sweep for contigs of any length: column tiles of 64 x C columns, one wavefront, the tile's edge column staged in LDS --
// Same recurrence, end cells and rules as bk_nw_score_c; the columns of a tile are right-aligned in the lanes (the last column is
// register C-1 of the last lane: a full tile has no padding, the last tile's padding passes the edge column through).  bound_in /
// bound_out: n ints each in LDS (the scores of the column left of the tile / of its last column); the edge values and the row
// symbols of 64 steps are fetched as a block, so the loop itself has no LDS read.  s2 / i2 (the last row: largest column among the
// maxima) are carried from tile to tile in ascending column order; best / best_im1 (the last column) come from the last tile.
struct BkScoreCarry { int s2, i2, best, best_im1; };
template <int C>
__device__ __noinline__ BkScoreCarry bk_nw_score_tile_c(const uint8_t *cols, const uint8_t *rows, int n_, int j0_, int mt_, const int *bound_in, int *bound_out, int last_, int s2_, int i2_)
{
    const int n = __builtin_amdgcn_readfirstlane(n_), j0 = __builtin_amdgcn_readfirstlane(j0_), mt = __builtin_amdgcn_readfirstlane(mt_);
    const bool last = __builtin_amdgcn_readfirstlane(last_) != 0;
    const int lane = threadIdx.x & 63;
    const int nl = (mt + C - 1) / C, lm = nl - 1, pad = nl * C - mt;
    // (cells hold V = S + 2 (i + column within the tile), as in bk_nw_score_c: 4 instructions per cell; the edge columns in LDS hold scores)
    // (no per-column mismatch register either, as in bk_nw_score_c: padding only ever occurs in the FIRST tile (j0 = 0: bk_nw_score_long makes every
    //  later tile a whole number of columns per lane), where it stands for the zero border and the literal 2 is exact; next to an edge column
    //  with scores of its own a + 2 inside the padding could exceed the edge's value one row down)
    int H[C], cb[C];
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = lane * C + x - pad;
        const bool real = jj >= 0 && jj < mt && lane <= lm;
        H[x] = jj >= 0 ? 2 * (jj + 1) : 0;
        cb[x] = real ? 4 * (int)cols[j0 + jj] : 28;
    }
    int dprev = 2 * max(lane * C - pad, 0), rb = 0, im1 = -lane;
    int voff = 2 * (1 - lane + mt);
    int best = 0, best_im1 = -1;
    const bool inl = lane <= lm;
    const int steps = n + lm;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        int rrot = 3 << (4 * ((t0 + lane < n) ? (int)rows[t0 + lane] : 0));          // the rows' symbol words rotate towards lane 0 (bk_nw_score_c)
        const int bblk = (j0 > 0 && t0 + lane < n) ? bound_in[t0 + lane] : 0;      // S[i][j0] for the rows lane 0 handles in this block (the border column: 0)
        const int te = min(64, steps - t0);
        for (int tl = 0; tl < te; tl++) {
            const int recv = __builtin_amdgcn_mov_dpp(H[C - 1], 0x138, 0xf, 0xf, true);
            rb = __builtin_amdgcn_update_dpp(rb, rb, 0x138, 0xf, 0xf, false);
            const int bi = __builtin_amdgcn_readlane(bblk, tl);
            rb = lane == 0 ? rrot : rb;
            rrot = __builtin_amdgcn_update_dpp(rrot, rrot, 0x130, 0xf, 0xf, false);
            if (inl && (unsigned)im1 < (unsigned)n) {
                const int left = lane == 0 ? bi + 2 * im1 + 2 : recv;               // lane 0 handles row t0 + tl at this step: its edge value is entry tl of the block
                // the diagonal candidates of the whole row first (from the previous row's values), then the chain of max3 IN PLACE: left
                // to itself the compiler interleaves them and pays a register copy per column to put the new values back where the
                // loop carries them
                int dg[C];
                dg[0] = dprev + (int)__builtin_amdgcn_ubfe((unsigned)rb, (unsigned)cb[0], 4u) + 2;
#pragma unroll
                for (int x = 1; x < C; x++) dg[x] = H[x - 1] + (int)__builtin_amdgcn_ubfe((unsigned)rb, (unsigned)cb[x], 4u) + 2;
                __builtin_amdgcn_sched_barrier(0);
                int u_in = left;
#pragma unroll
                for (int x = 0; x < C; x++) { H[x] = max(max(dg[x], u_in), H[x]); u_in = H[x]; }
                dprev = left;
                const int v = H[C - 1] - voff;
                if (last) { const bool take = v >= best; best = take ? v : best; best_im1 = take ? im1 : best_im1; }
                else if (lane == lm) bound_out[im1] = v;
            }
            im1++; voff += 2;
        }
    }
    int s2 = s2_, i2 = i2_;                                                         // carried: compared with '>=' in ascending column order
    int mx = -0x40000000, mi = 0;
#pragma unroll
    for (int x = 0; x < C; x++) {
        const int jj = lane * C + x - pad, sc = H[x] - 2 * (n + jj + 1);
        if (jj >= 0 && jj < mt && inl && sc >= mx) { mx = sc; mi = j0 + jj + 1; }
    }
    for (int o = 1; o < 64; o <<= 1) {
        const int os = __shfl_xor(mx, o), oi = __shfl_xor(mi, o);
        if (os > mx || (os == mx && oi > mi)) { mx = os; mi = oi; }
    }
    if (mx >= s2) { s2 = mx; i2 = mi; }
    BkScoreCarry c; c.s2 = s2; c.i2 = i2; c.best = __shfl(best, lm); c.best_im1 = __shfl(best_im1, lm);
    return c;
}
template <int C>
__device__ inline BkScoreCarry bk_nw_score_tile_call(int c, const uint8_t *cols, const uint8_t *rows, int n, int j0, int mt, const int *bi, int *bo, int last, int s2, int i2)
{
    if (c <= C) return bk_nw_score_tile_c<C>(cols, rows, n, j0, mt, bi, bo, last, s2, i2);
    if constexpr (C < BK_NW_STILE_C) return bk_nw_score_tile_call<C + 1>(c, cols, rows, n, j0, mt, bi, bo, last, s2, i2);
    BkScoreCarry z; z.s2 = s2; z.i2 = i2; z.best = 0; z.best_im1 = -1; return z;
}
// the epilogue of bk_nw_score_c as a function of the end cells (lane 0 writes the 8 result ints)
__device__ inline void bk_nw_score_results(int *res, int mt, int n, int s1, int i1, int s2, int i2, int x1, int x2, int force)
{
    // x1 / x2: mismatches on the pure diagonal from the border to the end cell of v1 / v2 (bk_nw_diag_mismatches; 0 where it was not counted)
    // v1 = nw(contig, read): m_ref = mt, end cell (i1, mt); v2 = nw(read, contig): m_ref = n, end cell (contig position i2, read column n)
    const int minlen = min(mt, n);
    int j1 = 0, r1 = 0, j2 = 0, r2 = 0; bool k1 = false, k2 = false;             // border cells (j_start, i_start); known without a traceback?
    // Which border cells does check_align look at (bk_decide, sv_assembly.py:459-503)?  ok_k = 4 s_k >= minlen and 200 s_k >= 179 *
    // overlap_k (needs j_start_k); both false: no match, nothing else is read.  Else the call with the larger score decides, alone;
    // equal scores: both are read.  So the sweep first establishes what it can of ok_1 / ok_2 (below), then: a match is certain -> the
    // winner's border cell; no match is certain -> none; else the border cells of the calls whose ok is still open.
    if (i1 == 0) { j1 = mt - 1; k1 = true; }                                     // Q5: one forced traceback step from (0, m), score 0
    else if (s1 == min(i1, mt) - 3 * x1) { j1 = i1 <= mt ? mt - i1 : 0; r1 = i1 <= mt ? 0 : i1 - mt; k1 = true; }
    if (i2 == 0) { j2 = n - 1; k2 = true; }
    else if (s2 == min(i2, n) - 3 * x2) { j2 = i2 <= n ? n - i2 : 0; r2 = i2 <= n ? 0 : i2 - n; k2 = true; }
    // ok_k as far as it is known here: 1 true, 0 false, -1 open.  False without the border cell where the score cannot pass the ratio
    // test whatever the path: a path into end cell (i, .) that starts on the top border consumes i rows = a diagonal steps + gr row gaps
    // and scores s <= a - 2 gr = i - 3 gr, so gr <= (i - s) / 3 and it spans >= a = i - gr >= (2 i + s) / 3 columns (its overlap); one
    // that starts on the left border spans all m_ref columns.  200 s < 179 * overlap for both => ok_k is false (the reads of a noisy
    // region that check_align rejects: ~10 % mismatches in an overlap score 0.7 per base against the 0.895 it asks for).
    const bool p1 = i1 > 0 && 4 * s1 >= minlen, p2 = i2 > 0 && 4 * s2 >= minlen;
    int o1 = !p1 ? 0 : k1 ? (200 * s1 >= 179 * (mt - j1) ? 1 : 0) : ((600 * s1 < 179 * (2 * i1 + s1) && 200 * s1 < 179 * mt) ? 0 : -1);
    int o2 = !p2 ? 0 : k2 ? (200 * s2 >= 179 * (n - j2) ? 1 : 0) : ((600 * s2 < 179 * (2 * i2 + s2) && 200 * s2 < 179 * n) ? 0 : -1);
    bool n1 = false, n2 = false;
    if (force) { n1 = true; n2 = true; }
    else if (o1 == 1 || o2 == 1) { n1 = s1 >= s2; n2 = s2 >= s1; }                // a match: the call with the larger score decides (both at equal scores)
    else if (o1 == 0 && o2 == 0) { }                                             // no match: nothing else is read
    else { n1 = o1 < 0; n2 = o2 < 0; }                                           // open: the border cell of the call(s) whose ok is not known
    res[0] = (n1 && !k1) ? BK_NW_NEEDS_DP : j1; res[1] = i1; res[2] = r1; res[3] = i1 == 0 ? 0 : s1;
    res[4] = (n2 && !k2) ? BK_NW_NEEDS_DP : j2; res[5] = i2; res[6] = r2; res[7] = i2 == 0 ? 0 : s2;
    if (force == 2) { res[0] = BK_NW_NEEDS_DP; res[4] = BK_NW_NEEDS_DP; }        // diagnostic (BK_F_FORCE_REDO): the caller sweeps every read again in full
}
// One read against a contig of any length, executed by the calling wavefront: up to 640 columns in registers (bk_nw_score_c), beyond
// that in column tiles of up to 64 x BK_NW_STILE_C columns.  contig / read / res: offsets into the dynamic LDS block; bound: 2 * n ints of
// LDS scratch (used from the second tile on).
__device__ inline void bk_nw_score_long(int contig, int clen, int read, int n, int res, int *bound, int force = 0)
{
    if (clen <= 64 * BK_NW_DUAL_C) { BkPairArgs A; A.contig = contig; A.clen = clen; A.read = read; A.n = n; A.res = res; bk_nw_score_one(A, force); return; }
    const uint8_t *cols = bk_dyn_lds + contig, *rows = bk_dyn_lds + read;
    int *bi = bound, *bo = bound + n;
    BkScoreCarry c; c.s2 = 0; c.i2 = 0; c.best = 0; c.best_im1 = -1;
    // balanced tiles: as few as 64 x BK_NW_STILE_C columns allow; every tile but the FIRST is 64 lanes x a whole number of columns (no
    // padding), the first one takes what is left (its padding stands for the zero border: bk_nw_score_tile_c)
    const int nt_ = (clen + 64 * BK_NW_STILE_C - 1) / (64 * BK_NW_STILE_C), tw = 64 * (((clen + nt_ - 1) / nt_ + 63) / 64), w0 = clen - (nt_ - 1) * tw;
    for (int j0 = 0; j0 < clen; j0 += (j0 == 0 ? w0 : tw)) {
        const int mt = j0 == 0 ? w0 : tw;
        const bool last = j0 + mt >= clen;
        c = bk_nw_score_tile_call<1>((mt + 63) / 64, cols, rows, n, j0, mt, bi, bo, last ? 1 : 0, c.s2, c.i2);
        int *t = bi; bi = bo; bo = t;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const int s1 = c.best, i1 = c.best_im1 + 1, s2 = c.s2, i2 = c.i2, hl = threadIdx.x & 63;
    int x1 = 0, x2 = 0;
    if (i1 > 0 && s1 != min(i1, clen)) x1 = bk_nw_diag_mismatches<64>(cols, rows, i1, clen, min(i1, clen), hl);
    if (i2 > 0 && s2 != min(i2, n)) x2 = bk_nw_diag_mismatches<64>(cols, rows, n, i2, min(i2, n), hl);
    if (hl == 0) bk_nw_score_results((int *)(bk_dyn_lds + res), clen, n, s1, i1, s2, i2, x1, x2, force);
}

// Tried in round 3 and not kept, second attempt (profiles/r03/pair_two_word_ab.txt): bk_nw_pair_c with the second order carried as a
// second WORD [score | priority | origin] through its own v_max3 (12 instructions per cell, no compares against the score, no
// register copies, no hazard nops) instead of the origin register with its two compares and selects (14 + nops).  Bit-exact,
// 23 % faster while the SIMDs have wavefronts to spare (54.7 vs 70.9 us per round), but 10 % SLOWER where this kernel is used,
// at saturation (4.36 vs 4.85 TCUPS): 11 % fewer VALU instructions, each 24 % more expensive (SQ_INSTS_VALU, SQ_WAVE_CYCLES) --
// a second v_max3 per cell costs more than the three compare/select/copy instructions it replaces; neither literals moved to
// registers nor forming the next column's diagonal candidate early (which removes the copies of the kept version) changed that.
// Tried in round 3 and not kept (tools/dp_bench_dual.py history, profiles/r03/valu_rate.txt): the same sweep with two rows per
// iteration in a column-shifted frame (stored word = H + 2 j, so the left neighbour's word IS the horizontal candidate and the
// chain along a row is max3 -> v_and_or), laid out cell by cell so that the two rows' chains interleave.  Bit-exact, but no
// faster (0.9-1.0x): the loop is bound by instruction ISSUE, not by the dependency chain -- one wavefront alone issues a VALU
// instruction every ~5 cycles whatever the dependencies, a saturated SIMD issues a 2-operand instruction every ~2.4 cycles
// and a 3-operand one (v_max3_i32, v_and_or_b32) every ~4.2 -- and trading the add + and for one v_and_or saves nothing.
template <int C>
__device__ inline void bk_nw_dual_call(int c, int contig, int clen, int off1, int rows, int n, int res)
{
    if (c <= C) { bk_nw_dual_c<C>(contig, clen, off1, rows, n, res); return; }
    if constexpr (C < BK_NW_DUAL_C) bk_nw_dual_call<C + 1>(c, contig, clen, off1, rows, n, res);
}
// contig <= BK_NW_DUAL_COLS.  All three are byte offsets into the dynamic LDS block; res: 8 ints written by lane 0: v1 then
// v2 as (j_start, i_end, i_start, score)
__device__ inline void bk_nw_dual(int contig, int clen, int read, int n, int res)
{
    const int K = n + (n >> 1) + 2, off1 = clen > K ? clen - K : 0;
    bk_nw_dual_call<3>((clen + 31) / 32, contig, clen, off1, read, n, res);
}
