export TMPDIR=/tmp
out=gpurun_out/r2p29; mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
L=tools/probes/libbk_at256_probe
for rep in 1 2; do
for i in 4 5 6 7; do
timeout 300 python3 bench.py $B --lib $L --inflight $i > $out/n256_i${i}_$rep.json 2>/dev/null
timeout 300 python3 bench.py $B --inflight $i > $out/n512_i${i}_$rep.json 2>/dev/null
done
done
