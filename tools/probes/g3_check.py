"""diagnostic: the reference-generated assembly fixtures (tests/golden/assembly.json) through the library named by BK_LIB (default: the product),
both workgroup sizes; prints how many cases come out as the fixtures say (no pytest, no torch: a few seconds)"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tools"))
import _variant
_variant.use()                       # BK_VARIANT / BK_LIB: the build of the library to check (default: the product)
from breakmer_amd import hip_backend as hb, synth
d = json.load(open(os.path.join(root, "tests", "golden", "assembly.json")))
by = {}
for c in d["cases"]:
    by.setdefault((c["k"], c["rc_thresh"]), []).append(c)
bad = tot = 0
for (k, rc), cases in by.items():
    regions = [synth.make_region(**c["gen"]) for c in cases]
    for wg in (512, 256):
        eng = hb.Engine(kmer_size=k, rc_thresh=rc, wg_threads=wg)
        eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only) for r in regions])
        eng.run(3)
        for i, c in enumerate(cases):
            got = [{kk: v for kk, v in x.items() if kk not in ("total_reads", "n_hits")} for x in eng.contigs(i)]
            tot += 1
            if got != c["contigs"]:
                bad += 1
        eng.close()
print("G3 cases x workgroup sizes: %d, wrong: %d" % (tot, bad), flush=True)
