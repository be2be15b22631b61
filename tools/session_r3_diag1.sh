#!/bin/bash
# round-3 diagnostics batch 1 (one GPU box): counters that exist, per-phase times (headline at both workgroup sizes, one noisy
# region, one configs[4] region), A/B of the default library
export TMPDIR=/tmp
o=gpurun_out/r3d; mkdir -p $o
rocprofv3-avail list > $o/avail.txt 2>&1 || rocprofv3 --list-avail > $o/avail.txt 2>&1
BK_WG=512 python3 tools/phase_probe_headline.py > $o/phase_headline_512.txt 2>&1
BK_WG=256 python3 tools/phase_probe_headline.py > $o/phase_headline_256.txt 2>&1
python3 tools/phase_probe_noise.py 0.005 > $o/phase_noise.txt 2>&1
timeout 600 python3 tools/phase_probe_cfg4_asm.py > $o/phase_cfg4.txt 2>&1
timeout 600 python3 tools/phase_probe_k_cfg4.py > $o/phase_k_cfg4.txt 2>&1
python3 bench.py --other-configs 0 --cpu-sample 0 > $o/bench_a.json 2> $o/bench_a.err
python3 bench.py --other-configs 0 --cpu-sample 0 > $o/bench_b.json 2> $o/bench_b.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $o/pmc_a -- python3 bench.py --steps 12 --warmup 6 --cpu-sample 0 --other-configs 0 > $o/pmc_a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES --kernel-trace --output-format csv -d $o/pmc_b -- python3 bench.py --steps 12 --warmup 6 --cpu-sample 0 --other-configs 0 > $o/pmc_b.log 2>&1
python3 - <<'PY'
import csv, glob, os
for sub in ("pmc_a", "pmc_b"):
    fs = glob.glob("gpurun_out/r3d/%s/**/*_counter_collection.csv" % sub, recursive=True)
    if not fs: print(sub, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        a = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0.0, set()])
        a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    with open("gpurun_out/r3d/%s_summary.txt" % sub, "w") as f:
        for k, d in sorted(acc.items()):
            f.write(k + ": " + ", ".join("%s=%.0f" % (n, v[0] / max(1, len(v[1]))) for n, v in sorted(d.items())) + "\n")
    os.system("rm -rf gpurun_out/r3d/%s" % sub)
PY
tail -3 $o/*_summary.txt
