#!/bin/bash
# look-ahead into the next seeds: stability, parity subset, noise probes, bench A/B against the build before it
out=gpurun_out/ab_xseed; mkdir -p $out
bad=0; for i in $(seq 1 10); do r=$(timeout 300 python3 tools/session_r2/dbg_xvisit.py 2>&1 | grep -c -E "fault|EXC|MISMATCH|bad [1-9]"); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "G3 fixtures x10: runs with problems $bad"
timeout 1500 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "g3 or batch_vs_oracle or both_workgroup or edge or noisy or long_reads or more_regions or reads_with_n or config4_config5 or lookahead" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for nz in 0.005 0.05; do for f in 0 16; do echo NOISE $nz flags $f; BK_FLAGS=$f BK_WG=512 timeout 300 python3 tools/phase_probe_noise.py $nz 2>&1 | grep -E "asm kernel|DP |rounds"; done; done
B="--cpu-sample 0 --steps 40"
for lib in "" "--lib oldlibs/lib_before_xseed.so"; do
python3 bench.py $B $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or 'new', d['value'], d['one_step_at_a_time']['kernels_ms']['bk_asm_kernel']);
for k,v in d['other_configs'].items(): print('  ', k, v['value'], v['kernels_ms'])" "$lib"
done
