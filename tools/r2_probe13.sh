export TMPDIR=/tmp
mkdir -p gpurun_out/r2p13
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2p13/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2p13/pytest.log
bash tools/profile_round.sh r02 > gpurun_out/r2p13/profile.log 2>&1
