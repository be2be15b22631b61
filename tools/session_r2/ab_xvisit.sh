#!/bin/bash
# cross-visit look-ahead: stability of the G3 fixtures over repeated runs, parity subset, phase probes, bench A/B against
# the previous build (oldlibs/lib_before_xvisit.so)
out=gpurun_out/ab_xvisit; mkdir -p $out
bad=0; for i in $(seq 1 12); do r=$(timeout 300 python3 tools/session_r2/dbg_xvisit.py 2>&1 | grep -c -E "fault|EXC|MISMATCH|bad [1-9]"); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "G3 fixtures x12 (flags 8 and 0, both workgroup sizes): runs with problems $bad"
timeout 1500 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "g3 or batch_vs_oracle or both_workgroup or edge or noisy or long_reads or more_regions or reads_with_n or config4_config5" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for wg in 512 256; do echo HEADLINE WG $wg; BK_WG=$wg python3 tools/phase_probe_headline.py 256 2>&1 | grep -E "asm kernel|DP |decide|load_read|per region"; done
for nz in 0.005 0.05; do echo NOISE $nz; BK_WG=512 timeout 300 python3 tools/phase_probe_noise.py $nz 2>&1 | grep -E "asm kernel|DP |rounds"; done
B="--cpu-sample 0 --other-configs 0"
for rep in 1 2; do
  python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['one_step_at_a_time']['kernels_ms'])"
  python3 bench.py $B --lib oldlibs/lib_before_xvisit.so 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['value'], d['one_step_at_a_time']['kernels_ms'])"
  python3 bench.py $B --flags 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new, flag 8 (no cross-visit)', d['value'], d['one_step_at_a_time']['kernels_ms'])"
done
