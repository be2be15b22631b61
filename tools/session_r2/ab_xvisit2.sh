#!/bin/bash
# other configs + noise with the cross-visit look-ahead against the build before it
for nz in 0.005 0.05; do BK_WG=512 timeout 300 python3 tools/phase_probe_noise.py $nz 2>&1 | grep -E "asm kernel|rounds"; done
for lib in "" "--lib oldlibs/lib_before_xvisit.so"; do
python3 bench.py --cpu-sample 0 --steps 40 $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or 'new', d['value'], d['one_step_at_a_time']['kernels_ms']['bk_asm_kernel']);
for k,v in d['other_configs'].items(): print('  ', k, v['value'], v['kernels_ms'])" "$lib"
done
