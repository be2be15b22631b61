"""cProfile of runner.run() over noisy targets with a deep launch queue (run_depth, throughput_mode): where the driver's time goes
   python3 tools/probes/runner_noisy_profile.py [n_regions] [cycles] [run_depth]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from breakmer_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cyc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    regions = bench.make_regions_parallel("noisy", n)
    extra = {"run_depth": depth, "throughput_mode": "1"}
    for i in range(2):
        out = bench.time_runner(synth, regions, 31, cyc, extra=extra)
        print("run", i, {k: out[k] for k in ("value", "seconds", "batches")}, flush=True)
    pr = cProfile.Profile()
    pr.enable()
    out = bench.time_runner(synth, regions, 31, cyc, extra=extra)
    pr.disable()
    print("profiled", {k: out[k] for k in ("value", "seconds", "batches")})
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(s.getvalue())


if __name__ == "__main__":
    main()
