#!/bin/bash
# bench A/B of the in-tree library against another build ($1), on one box: headline, one-step kernels, other configs
B="--cpu-sample 0 --steps 60"
for rep in 1 2; do for lib in "" "--lib $1"; do
python3 bench.py $B $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or 'in-tree', d['value'], d['one_step_at_a_time']['kernels_ms'], [(k, v['value'], v['kernels_ms']['asm']) for k, v in d['other_configs'].items()])" "$lib"
done; done
