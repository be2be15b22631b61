import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from breakmer_amd import hip_backend as hb, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from breakmer_amd.sv_processor import params as bk_params
n = 256
regions = [synth.make_region(i) for i in range(n)]
opts = dict(bk_params.DEFAULTS); opts["var_filter"] = ["indel", "rearrangement", "trl"]
eng = hb.Engine(kmer_size=31)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
eng.set_call_context(bench.call_context_text(regions, opts))
acc = [0.0] * 5
for it in range(12):
    t0 = time.perf_counter(); eng.run(hb.BK_STAGE_ALL, sync=False); t1 = time.perf_counter()
    eng.sync(); t2 = time.perf_counter()
    eng.L.bk_call(eng.h); t3 = time.perf_counter()
    rows = eng.call(); t4 = time.perf_counter()
    blob = np.frombuffer("\n".join("%d\t%s" % (r, "\t".join(x)) for r in sorted(rows) for x in rows[r]).encode(), dtype=np.uint8); t5 = time.perf_counter()
    if it >= 2:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
print("launch %.3f  sync(wait+D2H work) %.3f  bk_call(fetch+chain+call) %.3f  call again+parse %.3f  blob %.3f ms" % tuple(a / 10 * 1e3 for a in acc))
print("gpu kernels", eng.kernel_ms(0))
