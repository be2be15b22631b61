#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun, from the repo root):
#   bash tools/profile_round.sh r04
# The PMC passes run one launch at a time (--inflight 1: counter collection serialises kernels anyway); the timed loop then
# launches the 256-thread assembler (bk_asm_kernel_w4, the kernel of the default bench line), the one-step-at-a-time pass the
# 512-thread one, so every pass has per-launch figures of both.
# Separate passes (kernel stats; FETCH_SIZE; WRITE_SIZE; SQ instruction counters): PMC collection is never combined with
# other tracing, and the profiled program is python3 itself (no launcher in between).
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
mkdir -p "$out"
B="--cpu-sample 0 --other-configs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $B > "$out/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 > "$out/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 > "$out/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$out/pmc_sq3" -- python3 bench.py --steps 12 --warmup 6 $B > "$out/pmc_sq3.log" 2>&1
python3 tools/summarize_profile.py "$out" "$tag"           # writes profiles/traffic.json + profiles/valu.json from THIS tree's counters ...
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"      # ... which the default bench line of the same session then quotes
python3 tools/summarize_profile.py "$out" "$tag" > /dev/null
# the summaries are written into profiles/ of THIS copy of the repo; only gpurun_out/ travels back from a gpurun box, so a copy goes there
mkdir -p "$out/summaries" && cp -r profiles/$tag "$out/summaries/" 2>/dev/null; cp profiles/traffic.json profiles/valu.json profiles/kernel_ms.json "$out/summaries/" 2>/dev/null
# raw per-dispatch counter CSVs are tens of MB: keep the small ones only
find "$out" -name "*.csv" -size +4M -delete 2>/dev/null; find "$out" -name "*.db" -delete 2>/dev/null

