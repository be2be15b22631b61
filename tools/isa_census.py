"""Spill / barrier / memory instruction census of the device code of the PRODUCT build, per function (runs anywhere: hipcc
cross-compiles gfx950 without a GPU).  python tools/isa_census.py [extra -D flags ...] > profiles/<round>/asm_isa_spill_census.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "breakmer_amd", "csrc", "bk_api.hip")
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "bk.s")
    log = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:] + [src, "-o", asm],
                         capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
    lines = open(asm).read().split("\n")
print("kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage):")
cur = None
for ln in log.splitlines():
    m = re.search(r"remark: (Function Name: \S+|\s+(?:TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): \d+)", ln)
    if m:
        t = m.group(1).strip()
        if t.startswith("Function Name"):
            cur = t.split(": ")[1]
            print("  " + cur + ":", end="")
        elif cur and cur.startswith("bk_"):
            print("  " + t, end="")
        if t.startswith("VGPRs Spill"):
            print()
starts = [(i, re.search(r"\.type\s+(\S+),@function", l).group(1)) for i, l in enumerate(lines) if "@function" in l and ".type" in l] + [(len(lines), "END")]
print("\nper function: instructions (lines), scratch_load, scratch_store, v_writelane, v_readlane, s_barrier, global_load, calls (s_swappc)")
rows = []
for (a, name), (b, _x) in zip(starts, starts[1:]):
    t = "\n".join(l for l in lines[a:b] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"))
    rows.append((t.count("\n") + 1, name, t.count("scratch_load"), t.count("scratch_store"), t.count("v_writelane"), t.count("v_readlane"), t.count("s_barrier"), t.count("global_load"), t.count("s_swappc")))
for r in sorted(rows, reverse=True)[:40]:
    print("  %6d  %-72s %4d %4d %5d %5d %4d %4d %4d" % (r[0], r[1][:72], *r[2:]))
