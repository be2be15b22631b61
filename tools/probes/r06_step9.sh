#!/bin/bash
# round 6, step 9: the score sweep comes back on after BK_SWEEP_RETRY rounds without it: parity, the two noisy probe regions, side configurations
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step9; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_hip_gpu.py -x -q -m gpu -p timeout --timeout 400 --timeout-method thread -k "g1_nw or nw_random or g3_assembly or lookahead or both_workgroup or config3 or config4 or long_reads or overflow_a_cap or arena or noisy_regions_and_both or redo_passes or batch_vs_oracle or config2_properties or more_regions or split" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
BK_WG=512 BK_FLAGS=128 BK_RID=50215 timeout 200 python3 tools/phase_probe_noise.py 0.005 > $O/asm_region_50215_one_unit.txt 2>&1
BK_WG=512 BK_FLAGS=128 timeout 200 python3 tools/phase_probe_noise.py 0.005 > $O/asm_noise_one_unit.txt 2>&1
timeout 900 python bench.py --side-configs-only 1 --cpu-sample 0 > $O/side.json 2> $O/side.err
tail -n 3 $O/pytest.log; cat $O/asm_region_50215_one_unit.txt $O/asm_noise_one_unit.txt
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_step9/side.json") if l.startswith("{")][-1])
for k, v in d.items():
    print(k, (v.get("value"), v.get("ms_per_batch"), (v.get("in_flight") or {}).get("value"), ((v.get("runner_end_to_end") or {}).get("steady_state") or {}).get("value")) if isinstance(v, dict) else v)
PY
