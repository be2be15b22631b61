#!/bin/bash
# bk_retire: per-read words fetched at the start (thread 0), FIFO removal folded into the bookkeeping: identity, parity, A/B
for f in 0 24; do timeout 900 python3 tools/stress_batch.py $f 1500 4 2>&1 | tail -1; done
bad=0; for i in $(seq 1 8); do r=$(timeout 300 python3 tools/session_r2/dbg_xvisit.py 2>&1 | grep -c -E "fault|EXC|MISMATCH|bad [1-9]"); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "G3 fixtures x8: runs with problems $bad"
timeout 1500 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "g3 or batch_vs_oracle or both_workgroup or edge or noisy or more_regions or reads_with_n or lookahead or runner or fetch" 2>&1 | tail -1
timeout 900 python3 tools/fuzz_parity.py 96 91 2>&1 | tail -1
bash tools/session_r2/ab_lib.sh $1
