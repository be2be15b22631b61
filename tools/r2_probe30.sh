export TMPDIR=/tmp
out=gpurun_out/r2p30; mkdir -p $out
timeout 2700 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
B="--cpu-sample 0 --other-configs 0"
for rep in 1 2; do
python3 bench.py $B > $out/bench_default_$rep.json 2>/dev/null
python3 bench.py $B --wg 512 --inflight 4 > $out/bench_wg512_$rep.json 2>/dev/null
done
timeout 300 python3 tools/stress_batch.py 0 1500 3 > $out/stress.log 2>&1
