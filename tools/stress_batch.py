"""Stress probe (not a test): one large mixed batch run repeatedly; every region's contigs must be identical across
runs and equal to the oracle.  python tools/stress_batch.py [flags] [n_regions] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant
_variant.use()
from breakmer_amd import hip_backend as hb, synth
from oracle import bk_oracle as bo
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
regions = [synth.make_region(5000 + i, sv_type=synth.SV_TYPES[i % 5], depth=(200 if i % 97 == 0 else 24), W=700, L=100,
                             noise=(0.01 if i % 50 == 7 else 0.0)) for i in range(n)]
eng = hb.Engine(kmer_size=25, flags=flags)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regions])
ref = None
bad = set()
for rep in range(reps):
    eng.run(7)
    cur = [[(c["seq"], tuple(c["others"]), tuple(c["reads"])) for c in eng.contigs(i)] for i in range(n)]
    if ref is None:
        ref = cur
    else:
        d = [i for i in range(n) if cur[i] != ref[i]]
        for i in d[:4]:
            for (sa, oa, ra), (sb, ob, rb_) in zip(cur[i], ref[i]):
                dif = [t for t in range(min(len(oa), len(ob))) if oa[t] != ob[t]]
                print("   region", i, "seq equal", sa == sb, len(sa), len(sb), "others differ at", (dif[0], dif[-1], len(dif)) if dif else None,
                      "delta", sorted({oa[t] - ob[t] for t in dif}), "reads only in cur", sorted(set(ra) - set(rb_)), "only in ref", sorted(set(rb_) - set(ra)), flush=True)
        print("flags", flags, "rep", rep, "regions differing from rep 0:", d[:20], len(d), flush=True)
        bad.update(d)
for i in sorted(bad)[:6]:
    r = regions[i]
    want, _ = bo.assemble_region(r.read_strs(), [r.window_str], 25, 2)
    w = [(c["seq"], tuple(c["others"]), tuple(c["reads"])) for c in want]
    print("region", i, r.sv_type, "depth", r.reads.shape[0], "rep0 == oracle:", ref[i] == w, "n contigs oracle/rep0", len(w), len(ref[i]), [len(x[0]) for x in w], [len(x[0]) for x in ref[i]])
print("done flags", flags, "bad", len(bad))
