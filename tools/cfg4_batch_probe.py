"""configs[4] (24,000 x 250 bp, k = 41, 5 % noise): regions/s of the whole GPU path against the batch size and the assembler's
workgroup size (one region's chain takes ~3 s whatever runs beside it, so the batch size IS the throughput until the arena --
~350 MB per region -- fills the HBM).  python tools/cfg4_batch_probe.py <distinct> <batch> [wg]   (batch = copies of the distinct regions)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from breakmer_amd import hip_backend as hb, synth
nd, nb = int(sys.argv[1]), int(sys.argv[2])
wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = time.time()
regs = [bench.cfg4_region(synth, i) for i in range(nd)]
print("generated %d regions in %.1f s" % (nd, time.time() - t), flush=True)
regs = [regs[i % nd] for i in range(nb)]
eng = hb.Engine(kmer_size=41, wg_threads=wg)
t = time.time()
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regs])
print("submit %.1f s" % (time.time() - t), flush=True)
eng.run(7)
for it in range(2):
    t = time.time(); eng.run(7); dt = time.time() - t
    print("batch %d wg %d: wall %.3f s = %.1f regions/s; kernel ms kmer %.0f asm %.0f sw %.0f; contigs %d failed %d" % (
        nb, wg, dt, nb / dt, eng.kernel_ms(1), eng.kernel_ms(2), eng.kernel_ms(3), eng.stat(6), eng.stat(22)), flush=True)
