"""Per-phase times of the k-mer kernel on ONE configs[4] region (24,000 x 250 bp, k = 41, 5 % noise); needs the diagnostic
build `python breakmer_amd/build.py stamps`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from breakmer_amd import hip_backend as hb, synth
from breakmer_amd import build as _bk_build; hb.load_library(_bk_build.lib_path("stamps"))      # the diagnostic build with phase stamps (python breakmer_amd/build.py stamps)
regions = [bench.cfg4_region(synth, 0)]
eng = hb.Engine(kmer_size=41)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
for it in range(2):
    eng.run(hb.BK_STAGE_KMER)
st = [eng.stat(100 + i) for i in range(12)]
mers, counts, U = eng.kmers(0)
print("kmer kernel ms", eng.kernel_ms(1), "U", U, "M", len(mers), "nslow", st[10])
seq = [(1, "P1 group"), (2, "P2 compact"), (0, "P0 ref table"), (3, "P3a classify"), (8, "P3a slow count"), (9, "sum"), (4, "P3b alloc+record"), (11, "P3b insert+sc"), (5, "P4 sort"), (6, "P5 postings"), (7, None)]
for (a, name), (b, _) in zip(seq[:-1], seq[1:]):
    print("%-18s %8.1f us" % (name, (st[b] - st[a]) / 100.0))
