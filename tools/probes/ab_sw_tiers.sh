# diagnostic: the realigner's SHORT-tier workgroup size (BK_SW_T; 1 = one 512-thread tier with the full LDS block), same box.
# BK_SW_T is read by the DIAGNOSTIC build of the library only (python breakmer_amd/build.py diag): the product reads no environment.
python3 breakmer_amd/build.py diag > /dev/null
mkdir -p gpurun_out/r4p
: > gpurun_out/r4p/ab.log
for t in ${SW_TS:-1 128 256 384 512}; do
  echo "== BK_SW_T=$t" >> gpurun_out/r4p/ab.log
  for rep in 1 2; do
  BK_SW_T=$t timeout 300 python bench.py --cpu-sample 0 --other-configs 0 --lib breakmer_amd/libbreakmer_hip_diag.so 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], json.dumps(d.get('kernels_ms', d.get('stage_ms',''))))" >> gpurun_out/r4p/ab.log 2>&1
  done
  BK_VARIANT=diag BK_SW_T=$t timeout 200 python tools/probes/split_probe.py full 64 0.005 2>&1 | grep "split   wg256" >> gpurun_out/r4p/ab.log
done
cat gpurun_out/r4p/ab.log
