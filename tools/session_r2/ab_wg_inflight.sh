#!/bin/bash
# headline throughput against the workgroup size of the batches in flight and the number of handles (one box)
for wg in 256 512; do for inf in 4 6 8; do
python3 bench.py --cpu-sample 0 --other-configs 0 --wg $wg --inflight $inf 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wg', sys.argv[1], 'inflight', sys.argv[2], d['value'], d['ms_per_step'])" $wg $inf
done; done
