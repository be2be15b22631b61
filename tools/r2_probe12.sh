export TMPDIR=/tmp
out=gpurun_out/r2p12; mkdir -p $out
L=tools/probes/libbk_at256_probe
B="--cpu-sample 0 --other-configs 0"
timeout 300 python3 bench.py $B > $out/r512_i3.json 2> $out/r512_i3.err; echo "512 i3 rc=$?" >> $out/log
for i in 2 3 3 4; do
timeout 300 python3 bench.py $B --lib $L --inflight $i > $out/r256_i$i.json 2> $out/r256_i$i.err; echo "256 i$i rc=$?" >> $out/log
done
timeout 300 python3 bench.py $B --lib $L --inflight 3 --flags 1 > $out/r256_i3_f1.json 2> $out/r256_i3_f1.err; echo "256 i3 f1 rc=$?" >> $out/log
