export TMPDIR=/tmp
out=gpurun_out/r2p28; mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
L=tools/probes/libbk_at256_probe
python3 bench.py $B > $out/n512_i4.json 2>/dev/null; echo "512 rc=$?" >> $out/log
for i in 2 4 6 8; do
timeout 300 python3 bench.py $B --lib $L --inflight $i > $out/n256_i$i.json 2> $out/n256_i$i.err; echo "256 i$i rc=$?" >> $out/log
done
timeout 300 python3 bench.py $B --lib $L --inflight 4 --regions 512 > $out/n256_i4_r512.json 2> /dev/null; echo "256 i4 r512 rc=$?" >> $out/log
timeout 300 python3 bench.py $B --inflight 4 --regions 512 > $out/n512_i4_r512.json 2> /dev/null; echo "512 i4 r512 rc=$?" >> $out/log
BREAKMER_HIP_LIB=$PWD/$L timeout 300 python3 tools/stress_batch.py 0 600 3 > $out/stress256.log 2>&1
