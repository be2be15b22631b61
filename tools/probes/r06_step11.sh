#!/bin/bash
# round 6, step 11: this visit's reads are unpacked while the first wavefront plans: parity, headline phases and A/B (oldplan = the tree before), side configurations
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step11; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g1_nw or nw_random or g3_assembly or lookahead or both_workgroup or config3 or config4 or long_reads or overflow_a_cap or arena or noisy_regions_and_both or redo_passes or batch_vs_oracle or config2_properties or more_regions or split" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
BK_WG=256 timeout 200 python3 tools/phase_probe_headline.py > $O/asm_phases_wg256.txt 2>&1
for rep in 1 2; do
  for v in "" oldplan; do
    lib=breakmer_amd/libbreakmer_hip${v:+_$v}.so
    timeout 200 python bench.py --lib $lib --cpu-sample 0 --other-configs 0 --steps 100 --warmup 6 > $O/bench_${v:-new}_$rep.json 2> /dev/null
  done
done
timeout 900 python bench.py --side-configs-only 1 --cpu-sample 0 > $O/side.json 2> $O/side.err
tail -n 3 $O/pytest.log; cat $O/asm_phases_wg256.txt
python3 - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r06_step11/bench_*.json")):
    try:
        d = json.loads([l for l in open(fn) if l.startswith("{")][-1])
        print(fn.split("/")[-1], d["value"], d["kernels_ms"], d["kernels_ms_inflight"])
    except Exception as e:
        print(fn, "ERR", e)
d = json.loads([l for l in open("gpurun_out/r06_step11/side.json") if l.startswith("{")][-1])
for k, v in d.items():
    print(k, (v.get("value"), v.get("ms_per_batch"), (v.get("in_flight") or {}).get("value"), ((v.get("runner_end_to_end") or {}).get("steady_state") or {}).get("value")) if isinstance(v, dict) else v)
PY
