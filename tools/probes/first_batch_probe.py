"""What the FIRST batch on a fresh handle costs (the arenas start sized for clean reads and learn their size by overflowing: the batch is run
again after every growth step): wall time of the first and of the second bk_run + bk_sync of the same noisy batch, several fresh handles in a row.
python tools/probes/first_batch_probe.py <regions> <noise> [flags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _variant
_variant.use()
from breakmer_amd import hip_backend as hb, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
    flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    regions = [synth.make_region(50000 + i, depth=500, L=150, sv_type="del", noise=noise) for i in range(n)]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
    for trial in range(3):
        eng = hb.Engine(kmer_size=31, flags=flags)
        eng.submit(ins)
        t0 = time.perf_counter(); eng.run(hb.BK_STAGE_ALL); t1 = time.perf_counter(); eng.run(hb.BK_STAGE_ALL); t2 = time.perf_counter()
        print("fresh handle %d: %d regions at %g, flags %d: first run %.1f ms, second run %.1f ms (ratio %.1f), contigs %d, failed %d" % (trial, n, noise, flags, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t1 - t0) / (t2 - t1), eng.stat(6), eng.stat(22)), flush=True)
        eng.close()

if __name__ == "__main__":
    main()
