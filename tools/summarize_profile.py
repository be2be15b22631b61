"""Reduce the rocprofv3 output of tools/profile_round.sh to the small files committed under profiles/<tag>/:
kernel_stats.csv (per-kernel averages of the default bench command), pmc_traffic_summary.csv and ../traffic.json
(HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are in KB and gfx950 reports
half of the fetched bytes, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
KERNELS = ("bk_kmer_kernel", "bk_kmer_kernel_g", "bk_asm_kernel", "bk_sw_kernel")


def one(pattern):
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None        # gpurun merges every call's files: take the latest


st = one("stats/**/*_kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        if r["Name"].startswith("bk_"):
            print("stats", r["Name"], "calls", r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
tot = {}
for cname, pat in (("FETCH_SIZE", "pmc_fetch/**/*_counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*_counter_collection.csv")):
    fn = one(pat)
    if not fn:
        continue
    acc = {}
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == cname and r["Kernel_Name"] in KERNELS:
            a = acc.setdefault(r["Kernel_Name"], [0.0, set()])
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
    for k, (v, ids) in acc.items():
        tot.setdefault(k, {})[cname] = v / max(1, len(ids))
if tot:
    with open(os.path.join(dst, "pmc_traffic_summary.csv"), "w") as f:
        f.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch\n")
        tj = {}
        for k, d in sorted(tot.items()):
            fe, wr = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
            b = int((2 * fe + wr) * 1024)
            f.write("%s,%.1f,%.1f,%d\n" % (k, fe, wr, b))
            tj[k + "_bytes_per_launch"] = b
            print("pmc", k, "fetch KB %.0f write KB %.0f -> %.1f MB/launch" % (fe, wr, b / 1e6))
    tj["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 6 --warmup 2 --inflight 1, 256 regions); units KB; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of the fetched bytes; calibrated there for wide coalesced "
                  "streams only, our loads are 4 B/lane: upper bound)")
    json.dump(tj, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
bj = os.path.join(src, "bench_default.json")
if os.path.isfile(bj) and os.path.getsize(bj):
    open(os.path.join(dst, "bench_default.json"), "w").write(open(bj).read())
