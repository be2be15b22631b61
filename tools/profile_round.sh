#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun, from the repo root):
#   bash tools/profile_round.sh r01
# Three separate passes (kernel stats, FETCH_SIZE, WRITE_SIZE): PMC collection is never combined with other tracing.
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
mkdir -p "$out"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --cpu-sample 0 > "$out/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --inflight 1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --inflight 1 > "$out/pmc_write.log" 2>&1
python3 tools/summarize_profile.py "$out" "$tag"
