export TMPDIR=/tmp
out=gpurun_out/r2p34; mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
for rep in 1 2; do
python3 bench.py $B > $out/kt512_$rep.json 2>/dev/null
python3 bench.py $B --lib tools/probes/libbk_kt1024_probe > $out/kt1024_$rep.json 2>/dev/null
done
BREAKMER_HIP_LIB=$PWD/tools/probes/libbk_kt1024_probe timeout 600 python3 -m pytest tests -m gpu -x -q -k "g3 or g4 or edge or reads_with_n or large_windows or long_reads" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
