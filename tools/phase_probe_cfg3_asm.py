"""Per-phase times of the assembler kernel on ONE configs[3] region (BK_CFG3_ID, default 3 = a translocation: 20,000 x 150 bp, partner window); stamps build."""

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
from breakmer_amd import build as _bk_build; hb.load_library(_bk_build.lib_path("stamps"))      # the diagnostic build with phase stamps (python breakmer_amd/build.py stamps)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 250
k = 31
import bench; regions = [bench.cfg3_region(synth, int(os.environ.get("BK_CFG3_ID", "3")))]
eng = hb.Engine(kmer_size=k, wg_threads=int(os.environ.get("BK_WG", "0")), flags=int(os.environ.get("BK_FLAGS", "0")))
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p_[4] for p_ in r.partners]) for r in regions])
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
print("kmer kernel ms", eng.kernel_ms(1), "asm kernel ms", eng.kernel_ms(2), "nw calls", eng.stat(1), "cells", eng.stat(0), "contigs", eng.contig_count(0),
      "U", eng.stat(4), "M", eng.stat(5), "T", eng.stat(7))
names = ["misc", "load_read", "DP", "decide+apply", "find_reads", "kmers_ordered", "check_alt", "emit", "setup_contigs(own)", "contig_new", "finalize(own)",
         "head scan", "remove_kmers", "grow snapshot", "grow pre-cand (used_mer, find_bytes)", "grow(own)"]
acc = [eng.stat(100 + i) / 100.0 for i in range(20)]
tot = sum(acc[:16])
for n, v in zip(names, acc):
    print("%-38s %12.1f us  %5.1f %%" % (n, v, 100 * v / tot))
print("rounds %d slots %d retired %d" % (acc[18] * 100, acc[16] * 100, acc[17] * 100))
d = eng.stat(119)
print("rounds with free slots %d; of these: not looked ahead %d, next visit unusable %d, next visit empty %d" % (d & 0xFFFF, (d >> 16) & 0xFFFF, (d >> 32) & 0xFFFF, (d >> 48) & 0xFFFF))
print("score sweeps (reads) %d, of these swept again in full %d" % (eng.stat(30), eng.stat(31)))
