"""A/B of two builds of the pair overlap-DP kernel (bk_nw_batch mode 5) at SIMD saturation; BK_LIB = the library to load."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from breakmer_amd import hip_backend as hb
if os.environ.get("BK_LIB"):
    hb.load_library(os.environ["BK_LIB"])
rnd = random.Random(1)
eng = hb.Engine(kmer_size=31)
base = "".join(rnd.choice("ACGT") for _ in range(600))
for (m, n) in ((298, 150), (270, 150), (224, 150)):
    a, b = base[:m], base[m - 100:m - 100 + n]
    for nb in (512, 4096, 8192):
        reps = 146
        out2, ms2 = eng.nw_batch([(a, b)] * nb, reps=reps, transposed=5)
        cells = 2 * nb * reps * m * n
        print("%s cols %d rows %d wavefronts %5d: %.3f ms, %.0f GCUPS" % (os.path.basename(hb.LIB_PATH), m, n, nb, ms2, 2 * cells / ms2 / 1e6))
