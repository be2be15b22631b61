#!/bin/bash
B="--cpu-sample 0 --steps 20 --cfg4-regions 8"
for rep in 1 2 3; do
for lib in "" "--lib $1"; do
python3 bench.py $B $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['other_configs']['configs[3]']; print(sys.argv[1] or 'new', 'cfg3', v['value'], v['kernels_ms'])" "$lib"
done; done
