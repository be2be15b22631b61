#!/bin/bash
# rocprofv3 counter passes on the k-mer stage ALONE (tools/kmer_only.py: bk_kmer_kernel of the 256-region headline batch, one launch
# at a time): what its waves wait for.  Separate passes (SQ has 8 slots, TCC 4), the profiled program is python3 itself.
#   bash tools/kmer_pmc.sh r04      (on the GPU box, from the repo root; writes gpurun_out/kpmc_<tag>/summary.txt)
set -u
tag=${1:-r04}
O=gpurun_out/kpmc_$tag
export TMPDIR=/tmp
mkdir -p "$O"
P="python3 tools/kmer_only.py"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$O/p1" -- $P > "$O/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d "$O/p2" -- $P > "$O/p2.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --kernel-trace --output-format csv -d "$O/p3" -- $P > "$O/p3.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d "$O/p4" -- $P > "$O/p4.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum --kernel-trace --output-format csv -d "$O/p4b" -- $P > "$O/p4b.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/p5" -- $P > "$O/p5.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/p6" -- $P > "$O/p6.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $P > "$O/stats.log" 2>&1
python3 - "$O" > "$O/summary.txt" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
print("bk_kmer_kernel, 256 regions x 10,000 x 150 bp reads, one launch at a time; counters per launch (sum over XCDs / SEs)")
for p in sorted(glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"] == "bk_kmer_kernel":
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for k, (v, ids) in sorted(acc.items()):
        print("%-4s %-36s %16.0f  (%d launches)" % (p.split("/")[2], k, v / max(1, len(ids)), len(ids)))
for p in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Name"].startswith("bk_"):
            print("stats", r["Name"], "calls", r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
cat "$O/summary.txt"
tail -n 1 "$O"/p1.log
find "$O" -name "*.csv" -size +2M -delete 2>/dev/null; find "$O" -name "*.db" -delete 2>/dev/null
