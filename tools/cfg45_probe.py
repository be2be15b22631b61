"""Timing probe for BASELINE configs[3]/[4]-shaped regions (not a test): python tools/cfg45_probe.py cfg4|cfg5 depth nregions"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
cfg, depth, nreg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if cfg == "cfg5":
    regs = [synth.make_region(700 + i, sv_type="del", depth=depth, W=3000, L=250, noise=0.05) for i in range(nreg)]; k = 41
else:
    regs = [synth.make_region(800 + i, sv_type=["del", "ins", "inv", "dup", "trl"][i % 5], depth=depth, W=3000, L=150) for i in range(nreg)]; k = 31
eng = hb.Engine(kmer_size=k)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regs])
eng.run(7)
t = time.time(); eng.run(7); dt = time.time() - t
print(cfg, "depth", depth, "regions", nreg, "reads/region", regs[0].reads.shape[0], "wall s %.3f" % dt, "contigs", [len(eng.contigs(i)) for i in range(min(nreg, 10))],
      "kernel ms", [round(eng.kernel_ms(j), 2) for j in (1, 2, 3)], "nw calls", eng.stat(1), flush=True)
