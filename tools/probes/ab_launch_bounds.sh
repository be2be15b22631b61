#!/bin/bash
# diagnostic (GPU box): the assembler kernels built for 3 workgroups per CU (__launch_bounds__(T, 3): 168-VGPR budget, no spills)
# against the product build (4 per CU, 128 VGPRs, a few spilled), same box, alternating runs of the default bench line.
set -u
out=gpurun_out/r4lb; mkdir -p $out
alt=$out/libbreakmer_hip_minb3.so
( cd breakmer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -DBK_ASM_MINB=3 -o ../../$alt bk_api.hip ) > $out/build.log 2>&1
: > $out/ab.txt
for rep in 1 2 3; do
  for lib in product minb3; do
    if [ $lib = product ]; then L=""; else L="--lib $alt"; fi
    python3 bench.py --cpu-sample 0 --other-configs 0 $L 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'regions/s', d['ms_per_step'], 'ms/step', json.dumps(d.get('kernels_ms','')))" >> $out/ab.txt
  done
done
cat $out/ab.txt
