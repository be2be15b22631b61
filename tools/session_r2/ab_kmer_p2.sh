#!/bin/bash
# A/B on one box: k-mer kernel with the per-unique-read arrays written by index (new) vs from the per-thread chunks (old lib
# given as $1): bench value, exclusive k-mer kernel time, and WRITE_SIZE / FETCH_SIZE of the k-mer kernel.
set -u
old=$1
out=gpurun_out/ab_kmer_p2
export TMPDIR=/tmp
mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
timeout 600 python3 -m pytest tests/test_hip_gpu.py -x -q -m gpu -k "kmer or config0 or config1 or whole_gene or reads_with_n or edge_cases or more_regions" > $out/pytest.log 2>&1
for rep in 1 2; do
  python3 bench.py $B > $out/new_$rep.json 2> $out/new_$rep.err
  python3 bench.py $B --lib $old > $out/old_$rep.json 2> $out/old_$rep.err
done
for which in new old; do
  L=""; [ $which = old ] && L="--lib $old"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w_$which -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 --wg 512 $L > $out/w_$which.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f_$which -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 --wg 512 $L > $out/f_$which.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in ("w_new","w_old","f_new","f_old"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/ab_kmer_p2/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, "n", len(v), "avg KB", sum(v)/len(v))
PY
