set -u
export TMPDIR=/tmp
out=gpurun_out/r2p8
mkdir -p $out
B="--cpu-sample 0 --other-configs 0"
for fl in 0 1; do
  python3 bench.py $B --flags $fl > $out/bench_f$fl.json 2> $out/bench_f$fl.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/sq_f$fl -- python3 bench.py --steps 6 --warmup 2 $B --inflight 1 --flags $fl > $out/sq_f$fl.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/sq3_f$fl -- python3 bench.py --steps 9 --warmup 3 $B --inflight 3 --flags $fl > $out/sq3_f$fl.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r2p8"
for sub in ("sq_f0", "sq_f1", "sq3_f0", "sq3_f1"):
    fs = glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True)
    if not fs: print(sub, "no file"); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        a = acc.setdefault(r["Kernel_Name"], {})
        c = a.setdefault(r["Counter_Name"], [0.0, set()]); c[0] += float(r["Counter_Value"]); c[1].add(r["Dispatch_Id"])
    with open(os.path.join(out, sub + "_summary.txt"), "w") as f:
        for k, a in sorted(acc.items()):
            if k.startswith("bk_"):
                f.write(k + " " + " ".join("%s=%.3g" % (n, v[0] / max(1, len(v[1]))) for n, v in sorted(a.items())) + "\n")
PY
