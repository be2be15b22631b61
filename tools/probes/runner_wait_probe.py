"""Where the driver's main thread spends a steady-state run (32 batches of 256 headline regions): wall time inside each
library call (submit / run / sync / call / set_call_context) and in the driver's own per-batch functions."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("BK_PROBE_TORCH"): import torch
import bench
from breakmer_amd import hip_backend as hb, synth, sv_processor as sp
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally: acc[cls.__name__ + "." + name] += time.perf_counter() - t0; cnt[cls.__name__ + "." + name] += 1
    setattr(cls, name, g)
for n in ("submit", "run", "sync", "call", "set_call_context", "fetch"):
    if hasattr(hb.Engine, n): wrap(hb.Engine, n)
for n in ("_submit_batch", "_launch_batch", "_finish_batch", "create_targets"):
    wrap(sp.runner, n)
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(512)]
out = bench.time_runner(synth, regions, 31, cycles=16)
print({k: out[k] for k in ("value", "seconds", "batches")})
for k in sorted(acc, key=lambda x: -acc[x]): print("%-28s %8.1f ms total  %4d calls  %7.3f ms/call" % (k, acc[k] * 1e3, cnt[k], acc[k] * 1e3 / cnt[k]))
