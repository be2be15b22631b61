set -u
export TMPDIR=/tmp
out=gpurun_out/r2p9
mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
python3 bench.py $B > $out/bench_512.json 2> $out/bench_512.err
python3 bench.py $B --lib tools/probes/libbk_at256_probe > $out/bench_256.json 2> $out/bench_256.err
python3 bench.py $B --lib tools/probes/libbk_at256_probe --inflight 4 > $out/bench_256_i4.json 2> $out/bench_256_i4.err
python3 bench.py $B --lib tools/probes/libbk_at256_probe --inflight 6 > $out/bench_256_i6.json 2> $out/bench_256_i6.err
python3 bench.py $B --inflight 4 > $out/bench_512_i4.json 2> $out/bench_512_i4.err
