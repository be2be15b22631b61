#!/bin/bash
# round 6, step 7: the sweep's lane-to-lane traffic as two DPP selects on VCC, the row guard on the key counter (41 -> 38 instructions per step): parity, then headline A/B
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step7; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g1_nw or nw_random or g3_assembly or lookahead or both_workgroup or config3 or config4 or long_reads or overflow_a_cap or arena or noisy_regions_and_both or redo_passes or batch_vs_oracle or config2_properties or more_regions" > $O/pytest_dp.log 2>&1
echo "pytest rc $?" >> $O/pytest_dp.log
for rep in 1 2 3; do
  for v in "" olddp; do
    lib=breakmer_amd/libbreakmer_hip${v:+_$v}.so
    timeout 200 python bench.py --lib $lib --cpu-sample 0 --other-configs 0 --steps 60 --warmup 6 > $O/bench_${v:-newdp}_$rep.json 2> /dev/null
  done
done
tail -n 3 $O/pytest_dp.log
python3 - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r06_step7/bench_*.json")):
    try:
        d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
        print(fn.split("/")[-1], d["value"], d["value_100_steps"]["value"], d["kernels_ms"], d["kernels_ms_inflight"])
    except Exception as e:
        print(fn, "ERR", e)
PY
