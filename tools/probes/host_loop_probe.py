"""diagnostic (GPU box): where the host thread of bench.py's timed loop spends a step -- waiting for the GPU (fetch) or working
(launching, call tail) -- and whether the call tail on the library's thread (bk_call_async) changes the rate.
   python3 tools/probes/host_loop_probe.py [steps] [inflight]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from breakmer_amd import hip_backend as hb, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
infl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(256)]
ctx = bench.call_context_text(regions, bench.default_opts())
ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
engs = []
for _ in range(infl):
    e = hb.Engine(kmer_size=31, rc_thresh=2, wg_threads=256)
    e.submit(ins); e.set_call_context(ctx); e.run(hb.BK_STAGE_ALL); e.call_blob()
    engs.append(e)

def loop(k, mode):
    t = {"fetch": 0.0, "run": 0.0, "tail": 0.0}
    for e in engs:
        e.run(hb.BK_STAGE_ALL, sync=False)
        if mode == "async":
            e._pending = False
    t0 = time.perf_counter()
    n = 0
    for s in range(k):
        e = engs[s % infl]
        a = time.perf_counter()
        if mode == "async" and e._pending:
            raw = e.call_blob()                       # the tail of the step before, made on the library's thread meanwhile
            n += raw.count(b"\n")
        b = time.perf_counter()
        e.fetch()
        c = time.perf_counter()
        e.run(hb.BK_STAGE_ALL, sync=False)
        d = time.perf_counter()
        if mode == "async":
            e.call_async(); e._pending = True
        else:
            raw = e.call_blob(); n += raw.count(b"\n")
        f = time.perf_counter()
        t["fetch"] += c - b; t["run"] += d - c; t["tail"] += (b - a) + (f - d)
    for e in engs:
        e.sync()
        if mode == "async" and e._pending:
            n += e.call_blob().count(b"\n")
    dt = time.perf_counter() - t0
    print("%-5s %d steps x 256 regions, %d in flight: %.0f regions/s, %.3f ms/step; host per step: fetch (wait + copy) %.3f, launch %.3f, call tail %.3f ms; calls %d"
          % (mode, k, infl, 256 * k / dt, dt / k * 1e3, t["fetch"] / k * 1e3, t["run"] / k * 1e3, t["tail"] / k * 1e3, n), flush=True)

for rep in range(2):
    loop(steps, "sync")
    loop(steps, "async")
