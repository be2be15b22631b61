#!/bin/bash
# "expect a rejection" predictor on noisy reads: noise probes, other configs, against the build before it
for nz in 0.005 0.02 0.05; do echo NOISE $nz; BK_WG=512 timeout 300 python3 tools/phase_probe_noise.py $nz 2>&1 | grep -E "asm kernel|DP |rounds"; done
B="--cpu-sample 0 --steps 40"
for lib in "" "--lib oldlibs/lib_before_thr.so"; do
python3 bench.py $B $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or 'new', d['value'], d['one_step_at_a_time']['kernels_ms']['bk_asm_kernel']);
for k,v in d['other_configs'].items(): print('  ', k, v['value'], v['kernels_ms'])" "$lib"
done
timeout 900 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "noisy or config4_config5 or lookahead or batch_vs_oracle" 2>&1 | tail -2
