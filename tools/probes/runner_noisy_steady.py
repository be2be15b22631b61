"""runner.run() over noisy targets with a deep launch queue: regions/s against the length of the run (first-batch effects against steady state)
   python3 tools/probes/runner_noisy_steady.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from breakmer_amd import synth  # noqa: E402


def main():
    regions = bench.make_regions_parallel("noisy", 256)
    for depth, cyc in ((8, 16), (8, 64), (8, 128), (4, 64), (12, 64)):
        extra = {"run_depth": depth, "throughput_mode": "1"}
        out = bench.time_runner(synth, regions, 31, cyc, extra=extra)
        print("run_depth", depth, "batches", out["batches"], {k: out[k] for k in ("value", "seconds")}, flush=True)


if __name__ == "__main__":
    main()
