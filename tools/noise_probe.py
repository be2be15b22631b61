"""Timing probe (not a test): configs[1] regions with sequencing noise: python tools/noise_probe.py noise nregions"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
noise, nreg = float(sys.argv[1]), int(sys.argv[2])
regs = [synth.make_region(900 + i, sv_type="del", depth=500, W=3000, L=150, noise=noise) for i in range(nreg)]
eng = hb.Engine(kmer_size=31)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regs])
eng.run(7)
t = time.time(); eng.run(7); dt = time.time() - t
print("noise", noise, "regions", nreg, "wall s %.3f" % dt, "M", len(eng.kmers(0)[0]), "contigs", [len(eng.contigs(i)) for i in range(min(nreg, 8))],
      "kernel ms", [round(eng.kernel_ms(j), 2) for j in (1, 2, 3)], "nw calls/region", eng.stat(1) // nreg, flush=True)
