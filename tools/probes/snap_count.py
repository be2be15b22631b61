import sys, os
sys.path.insert(0, "/root/repo")
from breakmer_amd import hip_backend as hb, synth
hb.LIB_PATH = "tools/probes/libbk_stamps_probe"
for (noise, depth, L, k) in ((0.05, 2000, 250, 41), (0.005, 500, 150, 31)):
    r = synth.make_region(40000, sv_type="del", depth=depth, W=3000, L=L, noise=noise)
    eng = hb.Engine(kmer_size=k)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens)])
    eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
    print(noise, "asm ms", eng.kernel_ms(2), "small snapshots", eng.stat(116), "entries", eng.stat(118), "| large", eng.stat(117), "entries", eng.stat(119), "| snapshot us", eng.stat(113) / 100.0, "grow(own) us", eng.stat(115) / 100.0)
