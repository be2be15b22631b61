"""The HOST half of libbreakmer_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer (CPU box
only: GPU-side sanitizers are not available on this pool).  bk_api.hip is compiled `hipcc --cuda-host-only -fsanitize=...` and
linked with a stand-in for the HIP runtime (tests/sanitize/hip_stub.cpp: device memory = zeroed host memory, launches do
nothing) and a driver (tests/sanitize/host_driver.cpp) that goes through the C-ABI: 2-bit packing on both paths, synchronous
and asynchronous submits on reused staging (also at the size where a submit goes to the device in chunks while its helper threads
fill the rest, with ASCII reads and 2-bit packed rows, and the fault each bad window / partner window / N list is reported with),
two handles on two threads, getters on empty results, error paths, bk_trim, and
the native call tail on the G5 / G8m fixture texts -- whose rows must still equal the reference's.  Two synchronisation bugs of
round 2 sat in code paths like these; this is the guard the host side did not have."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _tuplify(mask):
    if mask is None:
        return None
    if isinstance(mask, dict):
        return {k: [tuple(x) for x in v] for k, v in mask.items()}
    return [tuple(x) for x in mask]


def _write_inputs(d, golden_dir):
    from breakmer_amd import call_context as cc, synth
    from breakmer_amd.sv_processor import params as bk_params
    opts = dict(bk_params.DEFAULTS)
    opts["var_filter"] = ["indel", "rearrangement", "trl"]
    for n in (8, 5):
        regions = [synth.make_region(60 + i, depth=10, W=700, sv_type=("trl" if i % 3 == 0 else "del")) for i in range(n)]
        genes = {}
        for r in regions:
            genes[r.name] = ["chr" + r.chrom, r.start, r.end]
        lines = [cc.opts_line(opts)] + cc.tables_lines(genes, None)
        for i, r in enumerate(regions):
            qr = (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, "exon")])
            lines += cc.region_lines(i, qr, None, r.disc_reads, [("9", 1000)] if i % 3 == 0 else [], r.read_ids)
        open(os.path.join(d, "ctx_%d.txt" % n), "w").write("\n".join(lines) + "\n")
    want = []
    cases = json.load(open(os.path.join(golden_dir, "caller.json")))["cases"]
    for c in cases:
        qr = c["query_region"]
        query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
        cd = c["contig"]
        tags = set(i.split("/")[1] for i in c["read_ids"])
        lines = [cc.opts_line(c["opts"])] + cc.tables_lines(c["genes"], _tuplify(c["all_repeat_mask"]))
        lines += cc.region_lines(0, query_region, _tuplify(c["target_repeat_mask"]), c["disc_reads"])
        lines += cc.contig_lines(c["contig_id"], cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], len(cd["kmers"]), len(tags) == 1,
                                 c["psl_rows"], c["offset"], c["tname"])
        open(os.path.join(d, "case_%d.txt" % len(want)), "w").write("\n".join(lines) + "\n")
        want.append(c["expected"])
    mh = json.load(open(os.path.join(golden_dir, "realign_multihit.json")))
    for c in mh["cases"]:
        qr = c["query_region"]
        query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
        cd = c["contig"]
        e = c["contract"]
        lines = [cc.opts_line(mh["opts"])] + cc.tables_lines(c["genes"], None) + cc.region_lines(0, query_region, None, c["disc_reads"])
        lines += cc.contig_lines("contig1", cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], len(cd["kmers"]),
                                 len(set(i.split("/")[1] for i in c["read_ids"])) == 1, e["psl_rows"], e["offset"], e["tname"])
        open(os.path.join(d, "case_%d.txt" % len(want)), "w").write("\n".join(lines) + "\n")
        want.append(e["expected"])
    return want


def _build(d, tag, san):
    obj = os.path.join(d, "bk_api_%s.o" % tag)
    subprocess.check_call([HIPCC, "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-Wno-unused-result",
                           "-c", os.path.join(ROOT, "breakmer_amd", "csrc", "bk_api.hip"), "-o", obj], stderr=subprocess.DEVNULL)
    sym = [ln.split()[-1] for ln in subprocess.check_output(["nm", "-u", obj], text=True).splitlines() if "__hip_fatbin_" in ln]
    fat = os.path.join(d, "fatbin_%s.cpp" % tag)
    open(fat, "w").write("".join('extern "C" { extern const char %s[64]; const char %s[64] = {0}; }\n' % (s, s) for s in sym))
    exe = os.path.join(d, "host_driver_" + tag)
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp"), obj,
                           os.path.join(ROOT, "tests", "sanitize", "hip_stub.cpp"), fat, "-lpthread", "-o", exe], stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.skipif(not (os.path.isfile(HIPCC) and os.path.isfile(CLANG)), reason="needs the ROCm clang for the host-only sanitizer build")
@pytest.mark.parametrize("tag,san", [("asan_ubsan", "address,undefined"), ("tsan", "thread")])
def test_host_half_under_sanitizers(tmp_path, golden_dir, tag, san):
    d = str(tmp_path)
    want = _write_inputs(d, golden_dir)
    exe = _build(d, tag, san)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    p = subprocess.run([exe, d], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-4000:])
    for marker in ("AddressSanitizer", "runtime error:", "ThreadSanitizer", "LeakSanitizer"):
        assert marker not in p.stderr, p.stderr[-4000:]
    assert p.stdout.strip().endswith("DONE")
    rows = {}
    for ln in p.stdout.splitlines():
        if ln.startswith("ROW\t"):
            f = ln.split("\t")
            rows[int(f[1])] = f[3:] if len(f) > 3 and f[3] != "" else None
    assert len(rows) == len(want) >= 40
    for i, w in enumerate(want):
        assert rows[i] == w, i
    shutil.rmtree(d, ignore_errors=True)
