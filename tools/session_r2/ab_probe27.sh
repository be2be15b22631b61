export TMPDIR=/tmp
out=gpurun_out/r2p27; mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
for i in 1 2; do
python3 bench.py $B > $out/new_$i.json 2>/dev/null
python3 bench.py $B --lib tools/probes/libbk_inlinecold_probe > $out/old_$i.json 2>/dev/null
done
python3 bench.py $B --flags 4 > $out/new_f4.json 2>/dev/null
python3 bench.py $B --inflight 4 > $out/new_i4.json 2>/dev/null
python3 bench.py $B --inflight 5 > $out/new_i5.json 2>/dev/null
python3 bench.py $B --inflight 6 > $out/new_i6.json 2>/dev/null
timeout 300 python3 tools/noise_probe.py 0.005 64 > $out/noise_new.log 2>&1
BREAKMER_HIP_LIB=$PWD/tools/probes/libbk_inlinecold_probe timeout 300 python3 tools/noise_probe.py 0.005 64 > $out/noise_old.log 2>&1
timeout 300 python3 tools/cfg45_probe.py cfg4 1000 256 > $out/cfg3_new.log 2>&1
BREAKMER_HIP_LIB=$PWD/tools/probes/libbk_inlinecold_probe timeout 300 python3 tools/cfg45_probe.py cfg4 1000 256 > $out/cfg3_old.log 2>&1
