#!/bin/bash
# throughput of the headline bench against the assembler's workgroups per CU (sized through the candidate-list cap -> LDS)
for mc in 2048 1024 512 3072 6000; do
  python3 bench.py --cpu-sample 0 --other-configs 0 --max-candidates $mc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['asm_workgroups_per_cu'])" $mc
done
