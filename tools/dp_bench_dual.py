"""Throughput of the dual overlap-DP kernel (both DPs of check_align on one wavefront, bk_nw_batch mode 3) on independent
problems, from one wavefront per CU to eight per SIMD: what the sweep itself sustains when nothing else is in its way (the
practical ceiling the assembler's DP rounds are measured against; profiles/r03/dp_bench_dual.txt)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb
rnd = random.Random(1)
eng = hb.Engine(kmer_size=31)
base = "".join(rnd.choice("ACGT") for _ in range(600))
for (m, n) in ((298, 150), (224, 150), (160, 150), (298, 151)):
    a, b = base[:m], base[m - 100:m - 100 + n]
    for nb in (256, 512, 2048, 8192):
        pairs = [(a, b)] * nb
        reps = 146
        out, ms = eng.nw_batch(pairs, reps=reps, transposed=3)
        out2, ms2 = eng.nw_batch(pairs, reps=reps, transposed=5)          # bk_nw_pair: every wavefront aligns TWO reads (both DPs of each)
        cells = 2 * nb * reps * m * n
        print("cols %d rows %d wavefronts %5d (%.1f per SIMD): dual %.3f ms, %.1f us per read, %.0f GCUPS | pair %.3f ms, %.1f us per TWO reads, %.0f GCUPS (x%.2f)" % (
            m, n, nb, nb / 1024.0, ms, ms * 1e3 / reps, cells / ms / 1e6, ms2, ms2 * 1e3 / reps, 2 * cells / ms2 / 1e6, 2 * ms / ms2))
