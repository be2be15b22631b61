"""Time the REAL reference's hot path (imported from /root/reference through oracle/ref_loader.py: lib2to3 translation, CPython 3)
on BASELINE configs[0] -- one region, 10,000 x 150 bp reads, planted 200 bp deletion, k = 31 -- and on a 4-region subset of
configs[1]: T1 grouping + k-mer selection (the reference's set algebra on a Jellyfish stand-in) + sv_assembly.init_assembly
(olc.nw inside).  Build container only (the reference cannot travel); the result is committed as
profiles/<round>/reference_python_timing.json and quoted by bench.py's cpu_baseline block.
    PYTHONHASHSEED=0 python tools/time_reference.py r03"""
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from breakmer_amd import synth  # noqa: E402
from oracle import bk_oracle as bo  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
for i in range(n):
    r = synth.make_region(i)
    reads = r.read_strs()
    t0 = time.perf_counter()
    mers = rh.ref_kmer_select(reads, [r.window_str], 31)
    t1 = time.perf_counter()
    cdicts, _ = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
    t2 = time.perf_counter()
    want, _ = bo.assemble_region(reads, [r.window_str], 31, 2)
    assert cdicts == want, "the C oracle and the reference disagree on region %d" % i
    rows.append({"region": i, "kmer_select_s": round(t1 - t0, 3), "init_assembly_s": round(t2 - t1, 3), "contigs": len(cdicts)})
    print(rows[-1], flush=True)
tot = sum(x["kmer_select_s"] + x["init_assembly_s"] for x in rows)
out = {"what": "the reference's own Python hot path (lib2to3 translation of /root/reference, CPython %s), one core" % platform.python_version(),
       "regions": n, "seconds": round(tot, 2), "regions_per_s": round(n / tot, 4), "per_region": rows, "cores": 1,
       "where": "build container (%s)" % (platform.processor() or platform.machine()),
       "workload": "configs[0] / configs[1] regions: 10,000 x 150 bp reads (500x), planted 200 bp deletion, k=31; realign and call tail not included (BLAT absent)",
       "parity": "contigs equal to oracle/bk_oracle.c on every timed region"}
d = os.path.join(ROOT, "profiles", tag)
os.makedirs(d, exist_ok=True)
json.dump(out, open(os.path.join(d, "reference_python_timing.json"), "w"), indent=1)
print(json.dumps(out)[:300])
