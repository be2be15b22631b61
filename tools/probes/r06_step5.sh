#!/bin/bash
# round 6, step 5: headline A/B of the score-sweep tile width (register pressure of the call graph), and the forced-redo check
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step5; mkdir -p $O; rm -f $O/*
timeout 400 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 300 --timeout-method thread -k "redo_passes or g1_nw or nw_random" > $O/pytest_redo.log 2>&1
echo "pytest rc $?" >> $O/pytest_redo.log
for rep in 1 2; do
  for v in "" stile8 stile10; do
    lib=breakmer_amd/libbreakmer_hip${v:+_$v}.so
    timeout 200 python bench.py --lib $lib --cpu-sample 0 --other-configs 0 --steps 60 --warmup 6 > $O/bench_${v:-stile13}_$rep.json 2> /dev/null
  done
done
tail -n 3 $O/pytest_redo.log
python3 - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r06_step5/bench_*.json")):
    try:
        d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
        print(fn.split("/")[-1], d["value"], d["value_100_steps"]["value"], d["kernels_ms"], d["kernels_ms_inflight"])
    except Exception as e:
        print(fn, "ERR", e)
PY
