"""only the k-mer stage (read grouping + sample k-mer selection: bk_kmer_kernel) of the headline batch, a few launches -- the program
behind the rocprofv3 --pmc passes that attribute the kernel's waiting (tools/profile_round.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(n)]
eng = hb.Engine(kmer_size=31, wg_threads=int(os.environ.get("BK_WG", "0")))
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
for _ in range(reps):
    eng.run(hb.BK_STAGE_KMER)
print("k-mer kernel ms", eng.kernel_ms(1), "U", eng.stat(4), "M", eng.stat(5))
