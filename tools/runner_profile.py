"""cProfile of breakmer_amd.sv_processor.runner.run() on n synthetic regions of the headline shape (GPU box):
   python3 tools/runner_profile.py [n_regions] [repeats]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from breakmer_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cyc = int(sys.argv[3]) if len(sys.argv) > 3 else 1
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(n)]
for i in range(rep):
    print("run", i, bench.time_runner(synth, regions, 31, cyc))
pr = cProfile.Profile()
pr.enable()
out = bench.time_runner(synth, regions, 31, cyc)
pr.disable()
print("profiled", out)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
