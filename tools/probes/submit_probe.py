"""bk_submit_regions of the headline batch on a warm handle: wall time, the library's own packing / copy split (stats 20, 21)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from breakmer_amd import hip_backend as hb, synth
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(256)]
ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
eng = hb.Engine(kmer_size=31)
for it in range(5):
    t = time.perf_counter(); eng.submit(ins); dt = time.perf_counter() - t
    print("submit %d: %.2f ms wall; packing %.2f ms, copies + waits %.2f ms" % (it, dt * 1e3, eng.stat(20) / 1e3, eng.stat(21) / 1e3), flush=True)
    eng.run(7)
