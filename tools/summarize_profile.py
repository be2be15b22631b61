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
KERNELS = ("bk_kmer_kernel", "bk_kmer_kernel_g", "bk_asm_kernel", "bk_asm_kernel_w4", "bk_sw_kernel", "bk_sched_kernel")


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
    # average launch duration of every kernel of the library in THIS profile run: quoted by the bench line (roofline.kernel_ms_profiled), so that
    # its roofline fraction can be recomputed from tracked files alone
    kj = {r["Name"] + "_avg_ms": round(float(r["AverageNs"]) / 1e6, 4) for r in rows if r["Name"].startswith("bk_")}
    kj.update({r["Name"] + "_calls": int(r["Calls"]) for r in rows if r["Name"].startswith("bk_")})
    kj["kernel_ms_source"] = "profiles/%s/kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python3 bench.py --cpu-sample 0 --other-configs 0`, tools/profile_round.sh %s)" % (tag, tag)
    json.dump(kj, open(os.path.join(root, "profiles", "kernel_ms.json"), "w"), indent=1)
# ---- overlap of the batches in flight, from the kernel trace of the same run: per window of 100 assembler launches of the timed loop -- the span
#      (first start .. last end), the summed durations per kernel, the time during which at least one kernel of the library ran (union-busy) and how
#      many launches were resident on average.  This is what reconciles "2.2 ms per assembler launch" with "0.8 ms per step": launches overlap.
tr = one("stats/**/*_kernel_trace.csv")
if tr:
    ev = []
    for r in csv.DictReader(open(tr)):
        if r["Kernel_Name"] in KERNELS:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    main = "bk_asm_kernel_w4" if sum(1 for e in ev if e[2] == "bk_asm_kernel_w4") >= sum(1 for e in ev if e[2] == "bk_asm_kernel") else "bk_asm_kernel"
    idx = [i for i, e in enumerate(ev) if e[2] == main]
    with open(os.path.join(dst, "overlap_summary.csv"), "w") as f:
        f.write("window,first_launch,launches_of_%s,span_ms,span_ms_per_launch,union_busy_ms,busy_frac,sum_ms_%s,resident_launches_avg\n" % (main, ",sum_ms_".join(KERNELS)))
        W = 100
        for w0 in range(0, max(0, len(idx) - W + 1), W):
            a, b = idx[w0], idx[w0 + W - 1]
            t0, t1 = ev[a][0], max(e[1] for e in ev[a:b + 1])
            win = [e for e in ev if e[0] >= t0 and e[0] <= ev[b][0]]
            sums = {k: 0 for k in KERNELS}
            for s_, e_, k in win:
                sums[k] += e_ - s_
            busy, cur_s, cur_e = 0, None, None
            for s_, e_, _k in win:
                if cur_e is None or s_ > cur_e:
                    busy += (cur_e - cur_s) if cur_e is not None else 0
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            busy += (cur_e - cur_s) if cur_e is not None else 0
            span = t1 - t0
            f.write("%d,%d,%d,%.3f,%.4f,%.3f,%.4f,%s,%.2f\n" % (w0 // W, w0, W, span / 1e6, span / 1e6 / W, busy / 1e6, busy / max(1, span), ",".join("%.3f" % (sums[k] / 1e6) for k in KERNELS), sum(sums.values()) / max(1, span)))
            print("overlap window %d: span %.2f ms (%.3f per launch), union-busy %.2f ms, %.2f launches resident on average" % (w0 // W, span / 1e6, span / 1e6 / W, busy / 1e6, sum(sums.values()) / max(1, span)))
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
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of the fetched bytes); the factor was re-measured for this "
                  "repo's 4 B/lane and row-strided loads with tools/pmc_calibrate.hip (profiles/r02/pmc_calibration.txt: 0.500-0.502); "
                  "WRITE_SIZE is exact for coalesced stores/atomics and counts a whole 64-B line per isolated 4-byte store")
    tj["traffic_source"] = "profiles/%s/pmc_traffic_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh %s)" % (tag, tag)
    json.dump(tj, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
# ---- SQ instruction counters (one pass, one handle in flight): VALU issue roofline of the assembler
cells = None
for cand in ("bench_default.json", "pmc_sq.log", "stats.log"):      # the bench line of any of the runs: the algorithmic cell count is a property of the workload
    bj0 = os.path.join(src, cand)
    if cells is None and os.path.isfile(bj0) and os.path.getsize(bj0):
        try:
            line = [ln for ln in open(bj0, errors="replace").read().splitlines() if ln.startswith("{") and "dp_cells_per_step" in ln][-1]
            cells = json.loads(line).get("dp_cells_per_step")
        except Exception:
            cells = None
for sub, label in (("pmc_sq", "one launch at a time: the 256-thread build (timed loop) and the 512-thread build (one-step-at-a-time pass)"), ("pmc_sq3", "default: 6 handles in flight, 256-thread workgroups")):
    fn = one(sub + "/**/*_counter_collection.csv")
    if not fn:
        continue
    acc = {}
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"] in KERNELS:
            a = acc.setdefault(r["Kernel_Name"], {})
            c = a.setdefault(r["Counter_Name"], [0.0, set()])
            c[0] += float(r["Counter_Value"])
            c[1].add(r["Dispatch_Id"])
    names = sorted({c for a in acc.values() for c in a})
    with open(os.path.join(dst, sub + "_summary.csv"), "w") as f:
        f.write("kernel," + ",".join(n + "_per_launch" for n in names) + "\n")
        for k, a in sorted(acc.items()):
            f.write(k + "," + ",".join("%.0f" % (a[n][0] / max(1, len(a[n][1])) if n in a else 0.0) for n in names) + "\n")
    if sub == "pmc_sq" and cells:
        vj = {"dp_cells_per_launch": cells,
              "valu_source": "profiles/%s/pmc_sq_summary.csv (rocprofv3 --pmc SQ_INSTS_VALU ... pass of tools/profile_round.sh %s, one launch at a time)" % (tag, tag),
              "valu_note": "wave-level VALU instructions of one launch x 64 lanes / algorithmic DP cells of the launch (sum len(seq1)*len(seq2) over the reference's nw calls)"}
        for kn in ("bk_asm_kernel", "bk_asm_kernel_w4"):
            if kn in acc and "SQ_INSTS_VALU" in acc[kn]:
                a = acc[kn]
                valu = a["SQ_INSTS_VALU"][0] / max(1, len(a["SQ_INSTS_VALU"][1]))
                vj[kn + "_valu_insts_per_launch"] = valu
                vj[kn + "_valu_laneops_per_cell"] = round(valu * 64.0 / cells, 3)
                for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                    if n in a:
                        vj[kn + "_" + n + "_per_launch"] = a[n][0] / max(1, len(a[n][1]))
                print("valu", kn, vj[kn + "_valu_laneops_per_cell"], "lane-ops per algorithmic cell")
        json.dump(vj, open(os.path.join(root, "profiles", "valu.json"), "w"), indent=1)
bj = os.path.join(src, "bench_default.json")
if os.path.isfile(bj) and os.path.getsize(bj):
    open(os.path.join(dst, "bench_default.json"), "w").write(open(bj).read())
