"""The cell encoding of the score sweep (csrc/bk_nw.hip.h: bk_nw_score_c / bk_nw_score_tile_c) restated lane by lane in Python and held
against the plain olc.nw recurrence (olc.py:62-74: diagonal +1 / -2, gaps -2, free leading gaps): cells hold V = S + 2 (i + j), a gap
move adds nothing, the diagonal adds 2 + bfe(3 << 4 code_row, 4 code_col, 4); columns right-aligned in the lanes with the padding
standing for column 0; lanes skewed by one step; long contigs in column tiles whose edge columns are handed over as SCORES.  What the
kernel reports from it -- (s1, i1): the largest row among the maxima of the last column; (s2, i2): the largest column among the maxima
of the last row -- must be those of the plain matrix.  (The kernels themselves are held against the reference's known answers and
the C oracle on the GPU: tests/test_hip_gpu.py, bk_nw_batch modes 9-20.)"""
import random


def plain_end_cells(cols, rows):
    m, n = len(cols), len(rows)
    S = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            S[i][j] = max(S[i - 1][j - 1] + (1 if cols[j - 1] == rows[i - 1] else -2), S[i][j - 1] - 2, S[i - 1][j] - 2)
    s1, i1 = 0, 0
    for i in range(0, n + 1):
        if S[i][m] >= s1:
            s1, i1 = S[i][m], i
    s2, i2 = 0, 0
    for j in range(1, m + 1):
        if S[n][j] >= s2:
            s2, i2 = S[n][j], j
    return (s1, i1, s2, i2), [S[i][m] for i in range(1, n + 1)]


def bfe(word, off, width):
    return (word >> off) & ((1 << width) - 1)


def sweep_tile(cols, rows, C, lanes, j0, bound_in, last, carry):
    """one tile of mt = len(cols) columns (matrix columns j0 + 1 .. j0 + mt) as the wavefront executes it; bound_in[i - 1] = S[i][j0]
    (None: the border column); returns (bound_out or None, (best, best_im1) of the last column if last, (s2, i2) carried on)"""
    mt, n = len(cols), len(rows)
    nl = (mt + C - 1) // C
    lm, pad = nl - 1, nl * C - mt
    assert nl <= lanes
    H = [[0] * C for _ in range(lanes)]
    cb = [[28] * C for _ in range(lanes)]
    ms = [[0] * C for _ in range(lanes)]
    for l in range(lanes):
        for x in range(C):
            jj = l * C + x - pad
            real = 0 <= jj < mt and l <= lm
            H[l][x] = 2 * (jj + 1) if jj >= 0 else 0
            cb[l][x] = 4 * cols[jj] if real else 28
            ms[l][x] = 2 if jj >= 0 else 0
    dprev = [2 * max(l * C - pad, 0) for l in range(lanes)]
    rb = [0] * lanes
    im1 = [-l for l in range(lanes)]
    voff = [2 * (1 - l + mt) for l in range(lanes)]
    best, best_im1 = [0] * lanes, [-1] * lanes
    bound_out = [None] * n
    for t in range(n + lm):
        recv = [0] + [H[l - 1][C - 1] for l in range(1, lanes)]            # DPP wave_shr:1 of the last column register (before this step's update)
        rb = [rb[0]] + rb[:-1]
        if t < n:
            rb[0] = 3 << (4 * rows[t])
        newH = [row[:] for row in H]
        for l in range(lanes):
            if l <= lm and 0 <= im1[l] < n:
                edge = (bound_in[im1[l]] if bound_in is not None else 0) + 2 * im1[l] + 2
                left = edge if l == 0 else recv[l]
                dg = [dprev[l] + bfe(rb[l], cb[l][0], 4) + ms[l][0]] + [H[l][x - 1] + bfe(rb[l], cb[l][x], 4) + ms[l][x] for x in range(1, C)]
                u = left
                for x in range(C):
                    newH[l][x] = max(dg[x], u, H[l][x])
                    u = newH[l][x]
                dprev[l] = left
                v = newH[l][C - 1] - voff[l]
                if last:
                    if v >= best[l]:
                        best[l], best_im1[l] = v, im1[l]
                elif l == lm:
                    bound_out[im1[l]] = v
            im1[l] += 1
            voff[l] += 2
        H = newH
    s2, i2 = carry
    mx, mi = -(1 << 30), 0
    for l in range(lm + 1):
        for x in range(C):
            jj = l * C + x - pad
            sc = H[l][x] - 2 * (n + jj + 1)
            if 0 <= jj < mt and (sc > mx or (sc == mx and j0 + jj + 1 > mi)):
                mx, mi = sc, j0 + jj + 1
    if mx >= s2:
        s2, i2 = mx, mi
    return bound_out, (best[lm], best_im1[lm] + 1), (s2, i2)


def test_sweep_encoding_equals_the_plain_recurrence():
    rnd = random.Random(5)
    for case in range(160):
        n = rnd.randint(1, 40)
        m = rnd.randint(1, 70)
        cols = [rnd.choice([0, 1, 2, 3, 3, 2, 4]) for _ in range(m)]
        rows = cols[rnd.randint(0, m - 1):][:n] if rnd.random() < 0.5 else []
        rows = (rows + [rnd.choice([0, 1, 2, 3, 4]) for _ in range(n)])[:n]
        if rnd.random() < 0.5:                                     # a few substitutions / an indel in an overlap
            for _ in range(rnd.randint(0, 3)):
                rows[rnd.randrange(n)] = rnd.choice([0, 1, 2, 3])
            if n > 4 and rnd.random() < 0.4:
                del rows[rnd.randrange(n)]
                rows.append(rnd.choice([0, 1, 2, 3]))
        want, _lastcol = plain_end_cells(cols, rows)
        # one tile holding every column (bk_nw_score_c: the register version), at several columns per lane
        for C in (1, 2, 3, 5, 8):
            lanes = (m + C - 1) // C
            _bo, (s1, i1), (s2, i2) = sweep_tile(cols, rows, C, lanes, 0, None, True, (0, 0))
            assert (s1, i1, s2, i2) == want, (case, C, cols, rows)
        # column tiles (bk_nw_score_long): edge columns handed over as scores, the last row's maximum carried on
        for tile, C in ((16, 4), (9, 3), (23, 5)):
            bound, carry, j0 = None, (0, 0), 0
            while j0 < m:
                mt = min(tile, m - j0)
                last = j0 + mt == m
                bound, end1, carry = sweep_tile(cols[j0:j0 + mt], rows, C, (tile + C - 1) // C, j0, bound, last, carry)
                j0 += mt
            assert (end1[0], end1[1], carry[0], carry[1]) == want, (case, tile, C, cols, rows)
