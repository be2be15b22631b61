export TMPDIR=/tmp
out=gpurun_out/r2p33; mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
for rep in 1 2; do
python3 bench.py $B > $out/new_$rep.json 2>/dev/null
python3 bench.py $B --lib tools/probes/libbk_prev_probe > $out/prev_$rep.json 2>/dev/null
done
timeout 300 python3 tools/stress_batch.py 0 1500 3 > $out/stress.log 2>&1
timeout 900 python3 -m pytest tests -m gpu -x -q -k "g3 or batch_vs or both_workgroup or more_regions or reads_with_n or noisy" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
