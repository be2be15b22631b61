export TMPDIR=/tmp
out=gpurun_out/r2p11; mkdir -p $out
L=tools/probes/libbk_at256_probe
B="--cpu-sample 0 --other-configs 0 --lib $L --steps 3 --warmup 1"
timeout 300 python3 bench.py $B --regions 4 --inflight 1 > $out/a.json 2> $out/a.err; echo "a rc=$?" >> $out/log
timeout 300 python3 bench.py $B --regions 256 --inflight 1 > $out/b.json 2> $out/b.err; echo "b rc=$?" >> $out/log
timeout 300 python3 bench.py $B --regions 4 --inflight 3 > $out/c.json 2> $out/c.err; echo "c rc=$?" >> $out/log
BREAKMER_HIP_LIB=$PWD/$L timeout 900 python3 -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "full_size_config2 or fetch_then or realign_vs or runner_end" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/log
