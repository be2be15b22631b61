#!/bin/bash
# round 6, step 8: the plan of a round on the first wavefront (slot s in lane s, list entry i in lane i) instead of thread 0: parity, phases, headline A/B
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step8; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g1_nw or nw_random or g3_assembly or lookahead or both_workgroup or config3 or config4 or long_reads or overflow_a_cap or arena or noisy_regions_and_both or redo_passes or batch_vs_oracle or config2_properties or more_regions" > $O/pytest_dp.log 2>&1
echo "pytest rc $?" >> $O/pytest_dp.log
BK_WG=256 timeout 200 python3 tools/phase_probe_headline.py > $O/asm_phases_wg256.txt 2>&1
for rep in 1 2 3; do
  for v in "" oldplan; do
    lib=breakmer_amd/libbreakmer_hip${v:+_$v}.so
    timeout 200 python bench.py --lib $lib --cpu-sample 0 --other-configs 0 --steps 60 --warmup 6 > $O/bench_${v:-newplan}_$rep.json 2> /dev/null
  done
done
tail -n 3 $O/pytest_dp.log; cat $O/asm_phases_wg256.txt
python3 - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r06_step8/bench_*.json")):
    try:
        d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
        print(fn.split("/")[-1], d["value"], d["value_100_steps"]["value"], d["kernels_ms"], d["kernels_ms_inflight"])
    except Exception as e:
        print(fn, "ERR", e)
PY
