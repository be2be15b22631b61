export TMPDIR=/tmp
out=gpurun_out/r2p35; mkdir -p $out
timeout 2700 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
BK_FUZZ_WG=256 timeout 1500 python3 tools/fuzz_parity.py 320 11 > $out/fuzz256.log 2>&1
