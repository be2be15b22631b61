"""Stress probe: bk_nw_dual (modes 3/4 of bk_nw_batch) against the single-DP sweeps on many concurrent wavefronts."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakmer_amd import hip_backend as hb
rnd = random.Random(5)
eng = hb.Engine(kmer_size=31)
pairs = []
base = "".join(rnd.choice("ACGT") for _ in range(4000))
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20000):
    m, n = rnd.randint(100, 320), rnd.randint(80, 160)
    s = rnd.randint(0, 3000)
    a = base[s:s + m]
    o = rnd.randint(-100, 100)
    b = base[max(0, s + m - n // 2 + o):][:n]
    pairs.append((a, b))
for rep in range(3):
    d0, _ = eng.nw_batch(pairs)
    d3, _ = eng.nw_batch(pairs, transposed=3)
    r0, _ = eng.nw_batch([(b, a) for a, b in pairs])
    d4, _ = eng.nw_batch(pairs, transposed=4)
    bad3 = np.nonzero((d0 != d3).any(axis=1))[0]
    bad4 = np.nonzero((r0 != d4).any(axis=1))[0]
    print("rep", rep, "v1 mismatches", len(bad3), bad3[:5], "v2 mismatches", len(bad4), bad4[:5], flush=True)
    for i in list(bad3[:3]):
        print("  v1", i, len(pairs[i][0]), len(pairs[i][1]), d0[i], d3[i])
    for i in list(bad4[:3]):
        print("  v2", i, len(pairs[i][0]), len(pairs[i][1]), r0[i], d4[i])
