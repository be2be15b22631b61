set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p1
for a in "cfg4 1000 8" "cfg4 1000 64" "cfg5 200 2" "cfg5 2000 1" "cfg5 2000 2"; do
  timeout 600 python3 tools/cfg45_probe.py $a >> gpurun_out/r2p1/cfg45.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p1/cfg45.log
done
for a in "0.005 64" "0.02 16"; do
  timeout 300 python3 tools/noise_probe.py $a >> gpurun_out/r2p1/noise.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p1/noise.log
done
timeout 600 python3 bench.py > gpurun_out/r2p1/bench.json 2> gpurun_out/r2p1/bench.err
