set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p2
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "not config4_regions" > gpurun_out/r2p2/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r2p2/pytest.log
for a in "cfg4 1000 64" "cfg4 1000 256" "cfg4 1000 1024"; do
  timeout 600 python3 tools/cfg45_probe.py $a >> gpurun_out/r2p2/cfg45.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p2/cfg45.log
done
for a in "0.005 64" "0.005 256"; do
  timeout 300 python3 tools/noise_probe.py $a >> gpurun_out/r2p2/noise.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p2/noise.log
done
timeout 900 python3 bench.py --other-configs 0 > gpurun_out/r2p2/bench.json 2> gpurun_out/r2p2/bench.err
