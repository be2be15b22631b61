"""Per-phase times of the k-mer kernel on N regions of the configs[1] shape with sequencing noise (default 1 region at 0.5 %);
needs the diagnostic build `python breakmer_amd/build.py stamps`.  With N > 1 the stamps are summed over the regions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
from breakmer_amd import build as _bk_build; hb.load_library(_bk_build.lib_path("stamps"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
regions = [synth.make_region(50000 + i, depth=500, L=150, sv_type="del", noise=noise) for i in range(n)]
eng = hb.Engine(kmer_size=31)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
for it in range(2):
    eng.run(hb.BK_STAGE_KMER)
st = [eng.stat(100 + i) for i in range(12)]
mers, counts, U = eng.kmers(0)
print("regions", n, "noise", noise, "kmer kernel ms", eng.kernel_ms(1), "U", U, "M", len(mers), "M2", sum(1 for c in counts if c >= 2), "T", eng.stat(7) // n, "tcap", eng.stat(8) // n, "nslow (sum)", st[10])
seq = [(1, "P1 group"), (2, "P2 compact"), (0, "P0 ref table"), (3, "P3a classify"), (8, "P3a slow count"), (9, "sum"), (4, "P3b alloc+record"), (11, "P3b insert+sc"), (5, "P4 sort"), (6, "P5 postings"), (7, None)]
for (a, name), (b, _) in zip(seq[:-1], seq[1:]):
    print("%-18s %8.1f us (mean per region)" % (name, (st[b] - st[a]) / 100.0 / n))
