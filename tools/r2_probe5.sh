set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p5
timeout 2400 python3 -m pytest tests -m gpu -x -q -k "not config4_regions and not config3_regions and not more_regions" > gpurun_out/r2p5/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r2p5/pytest.log
timeout 900 python3 bench.py --other-configs 0 --cpu-sample 0 > gpurun_out/r2p5/bench.json 2> gpurun_out/r2p5/bench.err
