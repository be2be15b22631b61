#!/bin/bash
# round 6, step 10: sweep threshold a half instead of a quarter, retry after 8 rounds: the noisy probe regions and the side configurations
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_step10; mkdir -p $O; rm -f $O/*
BK_WG=512 BK_FLAGS=128 BK_RID=50215 timeout 200 python3 tools/phase_probe_noise.py 0.005 > $O/asm_region_50215_one_unit.txt 2>&1
timeout 300 python3 tools/phase_probe_cfg4_asm.py > $O/asm_cfg4_region.txt 2>&1
timeout 900 python bench.py --side-configs-only 1 --cpu-sample 0 > $O/side.json 2> $O/side.err
tail -n 4 $O/asm_region_50215_one_unit.txt; head -1 $O/asm_region_50215_one_unit.txt; tail -n 4 $O/asm_cfg4_region.txt; head -1 $O/asm_cfg4_region.txt
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_step10/side.json") if l.startswith("{")][-1])
for k, v in d.items():
    print(k, (v.get("value"), v.get("ms_per_batch"), (v.get("in_flight") or {}).get("value"), ((v.get("runner_end_to_end") or {}).get("steady_state") or {}).get("value")) if isinstance(v, dict) else v)
PY
