set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p6
timeout 2400 python3 -m pytest tests -m gpu -x -q -k "not config4_regions" > gpurun_out/r2p6/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r2p6/pytest.log
timeout 900 python3 bench.py --other-configs 0 --cpu-sample 0 > gpurun_out/r2p6/bench.json 2> gpurun_out/r2p6/bench.err
timeout 300 python3 tools/noise_probe.py 0.005 64 > gpurun_out/r2p6/noise.log 2>&1
timeout 300 python3 tools/dp_bench.py > gpurun_out/r2p6/dp.log 2>&1
