import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
hb.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libbk_stamps_probe")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 200
regions = [synth.make_region(700, sv_type="del", depth=depth, W=3000, L=250, noise=0.05)]
eng = hb.Engine(kmer_size=41)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
print("asm kernel ms", eng.kernel_ms(2), "nw calls", eng.stat(1), "cells", eng.stat(0), "contigs", len(eng.contigs(0)))
print("rounds %d planned slots %d retired %d" % (eng.stat(106), eng.stat(104), eng.stat(105)))
acc = [eng.stat(100 + i) / 100.0 for i in range(4)]
print("asm region 0: outside check_read %.1f us | load_read %.1f | DP %.1f | decide+apply+bookkeeping %.1f" % tuple(acc))
