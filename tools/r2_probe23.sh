export TMPDIR=/tmp
mkdir -p gpurun_out/r2p23
timeout 2700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2p23/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2p23/pytest.log
