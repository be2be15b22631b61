#!/bin/bash
# diagnostic (GPU box): the working caps of the assembler (LDS per workgroup) against the headline with six batches in flight
out=gpurun_out/r4caps; mkdir -p $out; : > $out/ab.txt
for rep in 1 2; do
for v in "0 0" "1024 0" "1024 2048" "512 1024"; do
  set -- $v
  python3 bench.py --cpu-sample 0 --other-configs 0 --max-candidates $1 --max-contig $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('max_cand $1 max_contig $2:', d['value'], 'regions/s', d['ms_per_step'], 'ms/step', json.dumps(d.get('kernels_ms','')), 'failed', d.get('failed_regions'))" >> $out/ab.txt
done
done
cat $out/ab.txt
