set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p3
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2p3/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r2p3/pytest.log
for a in "cfg4 1000 256" "cfg4 1000 1024"; do
  timeout 600 python3 tools/cfg45_probe.py $a >> gpurun_out/r2p3/cfg45.log 2>&1
  echo "rc=$? $a" >> gpurun_out/r2p3/cfg45.log
done
timeout 900 python3 bench.py --cfg3-regions 1024 > gpurun_out/r2p3/bench.json 2> gpurun_out/r2p3/bench.err
