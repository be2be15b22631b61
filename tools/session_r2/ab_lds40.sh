#!/bin/bash
# assembler LDS back to 40 KB at 256 threads (four workgroups per CU again): identity, parity, A/B
for f in 0; do timeout 900 python3 tools/stress_batch.py $f 1500 3 2>&1 | tail -1; done
bad=0; for i in $(seq 1 8); do r=$(timeout 300 python3 tools/session_r2/dbg_xvisit.py 2>&1 | grep -c -E "fault|EXC|MISMATCH|bad [1-9]"); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "G3 fixtures x8: runs with problems $bad"
timeout 1500 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "g3 or batch_vs_oracle or both_workgroup or edge or noisy or long_reads or more_regions or reads_with_n or lookahead or large_windows or config4_config5" 2>&1 | tail -1
timeout 900 python3 tools/fuzz_parity.py 96 93 2>&1 | tail -1
BK_FUZZ_WG=256 timeout 900 python3 tools/fuzz_parity.py 64 94 2>&1 | tail -1
python3 bench.py --cpu-sample 0 --other-configs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs per cu', d['config']['asm_workgroups_per_cu'])"
bash tools/session_r2/ab_lib.sh $1
