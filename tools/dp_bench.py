import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb
rnd = random.Random(1)
eng = hb.Engine(kmer_size=31)
base = "".join(rnd.choice("ACGT") for _ in range(600))
for (m, n) in ((298, 150), (150, 298), (224, 150), (150, 224)):
    a, b = base[:m], base[m - 100:m - 100 + n]
    for nb in (256, 2048, 8192):
        pairs = [(a, b)] * nb
        reps = 146
        out, ms = eng.nw_batch(pairs, reps=reps, transposed=(m < n))
        cells = nb * reps * m * n
        print("cols %d rows %d blocks %5d reps %d: %.3f ms  -> %.1f us/DP/wave, %.1f GCUPS" % (m, n, nb, reps, ms, ms * 1e3 / reps, cells / ms / 1e6))
