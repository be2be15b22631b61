set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p4
free -g > gpurun_out/r2p4/mem.log 2>&1; nproc >> gpurun_out/r2p4/mem.log
for a in "cfg4 1000 2048" "cfg5 2000 16"; do
  timeout 900 python3 tools/cfg45_probe.py $a >> gpurun_out/r2p4/cfg45.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p4/cfg45.log
done
