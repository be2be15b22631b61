"""Per-phase time shares of the assembler kernel on the headline batch (256 regions of configs[1]); needs the diagnostic build
the build `python breakmer_amd/build.py stamps`.  BK_WG=256|512 selects the workgroup size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
from breakmer_amd import build as _bk_build; hb.load_library(_bk_build.lib_path("stamps"))      # the diagnostic build with phase stamps (python breakmer_amd/build.py stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(n)]           # bench.py's default batch
eng = hb.Engine(kmer_size=31, wg_threads=int(os.environ.get("BK_WG", "0")))
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
print("asm kernel ms", eng.kernel_ms(2), "nw calls", eng.stat(1), "cells", eng.stat(0))
names = ["misc", "load_read", "DP", "decide+apply", "find_reads", "kmers_ordered", "check_alt", "emit", "setup_contigs(own)", "contig_new", "finalize(own)",
         "head scan", "remove_kmers", "grow snapshot", "grow pre-cand (used_mer, find_bytes)", "grow(own)"]
acc = [eng.stat(100 + i) / 100.0 / n for i in range(20)]
tot = sum(acc[:16])
for nm, v in zip(names, acc):
    print("%-38s %10.1f us/region  %5.1f %%" % (nm, v, 100 * v / tot))
print("per region: total %.1f us, rounds %.1f slots %.1f retired %.1f" % (tot, acc[18] * 100, acc[16] * 100, acc[17] * 100))
