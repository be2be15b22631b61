"""CPU-only checks of the drop-in boundary: the built library loads, exports every symbol that
include/breakmer_hip.h declares, and refuses to run without a gfx950 GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from breakmer_amd import build, hip_backend
    build.build_hip()                       # hipcc cross-compiles for gfx950 without a GPU
    return hip_backend.load_library()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "breakmer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 15 and "bk_run" in names and "bk_get_hits" in names
    for n in names:
        assert getattr(lib, n) is not None, n


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from breakmer_amd import hip_backend as hb
    with pytest.raises(hb.BreakmerHipError) as e:
        hb.Engine(kmer_size=31)
    assert "no HIP device" in str(e.value) or "gfx950" in str(e.value)


def test_product_never_imports_the_oracle():
    """breakmer_amd/ must not reference oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "breakmer_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
                assert "bk_oracle" not in txt.replace("oracle/bk_oracle", ""), fn


def test_product_library_reads_no_environment(lib):
    """A drop-in's behaviour is a function of its arguments: the product build of the library imports no getenv at all (every
    diagnostic switch sits behind -DBK_DIAG: breakmer_amd/build.py VARIANTS), every workgroup barrier of the device code is
    BK_SYNC() (bk_common.h: the barrier-check and jitter builds then cover every kernel), and the Python layer names a diagnostic
    build explicitly (load_library(path)) instead of reading a variable."""
    import subprocess
    from breakmer_amd import build
    syms = subprocess.run(["nm", "-D", "--undefined-only", build.HIP_LIB], capture_output=True, text=True).stdout
    assert "getenv" not in syms and "hipLaunchKernel" in syms
    csrc = os.path.join(ROOT, "breakmer_amd", "csrc")
    for fn in os.listdir(csrc):
        txt = open(os.path.join(csrc, fn)).read()
        if fn != "bk_common.h":
            assert "__syncthreads" not in txt, fn
        if fn == "bk_api.hip":
            assert txt.count("getenv(") == 1 and "static inline const char *bk_diag_env(const char *name) { return getenv(name); }" in txt
        else:
            assert "getenv(" not in txt, fn
    for fn in os.listdir(os.path.join(ROOT, "breakmer_amd")):
        if fn.endswith(".py"):
            txt = open(os.path.join(ROOT, "breakmer_amd", fn)).read()
            for m in re.finditer(r"environ(?:\.get|\.setdefault|\[)\(?\s*[\"']([A-Z_0-9]+)", txt):
                assert m.group(1) in ("GPU_MAX_HW_QUEUES", "HIPCC", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"), (fn, m.group(1))


def test_pack_sequence_simd_equals_table_path():
    """bk_pack_sequence (host code of the library, no GPU): the SSSE3 path of the 2-bit packing -- whole 16-base blocks of
    plain A/C/G/T -- and the table path -- blocks with an N, tails -- against a Python restatement; other characters fail."""
    import random
    import numpy as np
    from breakmer_amd import hip_backend as hb
    rng = random.Random(5)

    def ref(seq):
        words = [0] * ((len(seq) + 15) // 16 + 1)
        npos = []
        for i, ch in enumerate(seq):
            c = "ACGTN".index(ch)
            if c == 4:
                npos.append(i)
                c = 0
            words[i // 16] |= c << (30 - 2 * (i % 16))
        return words, npos

    for it in range(400):
        n = rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 100, 150, 151, 250, 1000, 3000]) if it < 60 else rng.randrange(0, 400)
        seq = "".join(rng.choice("ACGT") if rng.random() > 0.01 else "N" for _ in range(n))
        want_w, want_n = ref(seq)
        w, npos = hb.pack_sequence(seq)
        assert w.tolist() == want_w and npos == want_n, (it, n)
        codes = np.array(["ACGTN".index(ch) for ch in seq], dtype=np.uint8)
        w2, npos2 = hb.pack_sequence(codes.tobytes(), codes=True)
        assert w2.tolist() == want_w and npos2 == want_n, (it, n)
    for bad in ("ACGTx", "acgt", "ACGTACGTACGTACGTACGR", "ACG-"):
        with pytest.raises(hb.BreakmerHipError):
            hb.pack_sequence(bad)
    with pytest.raises(hb.BreakmerHipError):
        hb.pack_sequence(bytes([0, 1, 2, 3, 7]), codes=True)


def test_pack_reads_matches_the_library_packing():
    """hip_backend.pack_reads (numpy; what BK_SUBMIT_PACKED takes) == bk_pack_sequence (the library's own packing) word for word, N
    positions included, for ragged reads with N calls."""
    import numpy as np
    from breakmer_amd import hip_backend as hb, synth
    r = synth.make_region(5, depth=30, W=900, n_frac=0.2, var_len=0.5)
    w, l, nl = hb.pack_reads(r.reads, r.read_lens)
    assert w.dtype == np.uint32 and l.tolist() == r.read_lens.tolist() and len(nl) > 10 and (np.diff(nl.astype(np.int64)) > 0).all()
    for i in range(r.reads.shape[0]):
        ww, npos = hb.pack_sequence(bytes(r.reads[i, :r.read_lens[i]]), codes=True)
        nw = (int(r.read_lens[i]) + 15) // 16
        assert (ww[:nw] == w[i, :nw]).all() and (w[i, nw:] == 0).all(), i
        assert [int(x & 1023) for x in nl if (x >> 10) == i] == npos, i


def test_struct_packed_region_array_equals_the_ctypes_one():
    """hip_backend.region_array_packed (one struct.pack per target) describes a batch exactly as RegionInput.fill does: same
    bk_region bytes apart from the window pointer, which points to the same characters"""
    import ctypes as C
    import numpy as np
    from breakmer_amd import hip_backend as hb, synth
    regs = [synth.make_region(60 + i, depth=20, W=700 + 50 * i, n_frac=(0.1 if i % 2 else 0.0), var_len=0.3) for i in range(5)]
    packed = [hb.pack_reads(r.reads, r.read_lens) for r in regs]
    assert all(type(p_) is hb.PackedReads and p_.maxlen == int(r.read_lens.max()) for p_, r in zip(packed, regs))
    io = [None, np.zeros(regs[1].reads.shape[0], dtype=np.uint8), None, np.ones(regs[3].reads.shape[0], dtype=np.uint8), None]
    arr, wbuf = hb.region_array_packed([(p_, r.window_str.encode(), i_) for p_, r, i_ in zip(packed, regs, io)])
    assert C.sizeof(arr) == len(regs) * C.sizeof(hb.BkRegion)
    ref = (hb.BkRegion * len(regs))()
    views = [hb.RegionInput(r.reads, r.window_str, read_lens=r.read_lens, indel_only=i_, packed=p_) for p_, r, i_ in zip(packed, regs, io)]
    for g, v in zip(ref, views):
        v.fill(g)
    for a, b, r in zip(arr, ref, regs):
        for name, _t in hb.BkRegion._fields_:
            if name == "window":
                assert C.string_at(C.cast(a.window, C.c_void_p).value, a.window_len) == r.window_str.encode() == C.string_at(C.cast(b.window, C.c_void_p).value, b.window_len)
            elif name == "partners":
                assert not a.partners and not b.partners
            else:
                assert getattr(a, name) == getattr(b, name), name


def test_struct_layouts_of_the_header_equal_the_ctypes_mirrors(tmp_path):
    """ABI 5: every structure of include/breakmer_hip.h as a C compiler lays it out (gcc on the header itself: sizeof and the offset of
    every member) against the ctypes mirrors a binding uses (breakmer_amd/hip_backend.py; INTEGRATION.md shows the same declarations)."""
    import subprocess
    from breakmer_amd import hip_backend as hb
    structs = {"bk_config": hb.BkConfig, "bk_region": hb.BkRegion, "bk_contig_info": hb.BkContigInfo, "bk_psl": hb.BkPsl}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "breakmer_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (name, name))
        for f, _t in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append('printf("abi %d known %d diag %d\\n", BK_ABI_VERSION, BK_CFG_KNOWN_MASK, BK_CFG_DIAG_MASK); return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = {}
    for ln in out:
        w = ln.split()
        if len(w) == 3 and w[0] in structs:
            got[(w[0], w[1])] = int(w[2])
    for name, cls in structs.items():
        assert got[(name, "sizeof")] == C.sizeof(cls), name
        for f, _t in cls._fields_:
            assert got[(name, f)] == getattr(cls, f).offset, (name, f)
    assert ("abi %d " % hb.BK_ABI_VERSION) in out[-2]
    # the names the Python layer uses for the flag bits are the header's
    hdr = open(os.path.join(ROOT, "include", "breakmer_hip.h")).read()
    for nm in ("BK_CFG_NO_SPLIT", "BK_CFG_TEST_SPLIT_ALWAYS", "BK_CFG_TEST_FULL_CALLER", "BK_CFG_TEST_HOST_REPAIR", "BK_CFG_TEST_PREQUEUE_UNITS"):
        m = re.search(r"\b%s\s*=\s*(\d+)" % nm, hdr)
        assert m and int(m.group(1)) == getattr(hb, nm), nm
    # ... and the device code's BK_F_* values are the same bits
    com = open(os.path.join(ROOT, "breakmer_amd", "csrc", "bk_common.h")).read()
    cfg_bits = {int(v) for v in re.findall(r"\bBK_CFG_(?:DIAG|TEST)?_?[A-Z0-9_]+\s*=\s*(\d+)\s*,", hdr)}
    f_bits = {int(v) for v in re.findall(r"\bBK_F_[A-Z0-9_]+\s*=\s*(\d+)", com)}
    assert f_bits == cfg_bits and len(f_bits) == 16, (sorted(f_bits), sorted(cfg_bits))


def test_bk_create_checks_its_configuration_before_it_looks_for_a_device(lib):
    """ABI 5: a non-zero `reserved` word, an unknown flag bit, a diagnostic-only bit (product build), a workgroup size other than 0 / 256 /
    512 are BK_E_ARG on any machine; a valid configuration then fails with BK_E_NOGPU here (no CPU fallback) or succeeds on a GPU box."""
    from breakmer_amd import hip_backend as hb

    def create(**kw):
        cfg = hb.BkConfig()
        cfg.abi_version, cfg.kmer_size, cfg.rc_thresh = hb.BK_ABI_VERSION, 31, 2
        for k, v in kw.items():
            if k == "reserved":
                cfg.reserved[v[0]] = v[1]
            else:
                setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.bk_create(0, C.byref(cfg), C.byref(h))
        text = lib.bk_last_error(None).decode() if rc else ""
        if rc == 0:
            lib.bk_destroy(h)
        return rc, text

    for bad, word in ((dict(reserved=(0, 1)), "reserved"), (dict(reserved=(1, 7)), "reserved"), (dict(flags=1 << 20), "unknown bit"), (dict(flags=512), "diagnostic-only"),
                      (dict(flags=1), "diagnostic-only"), (dict(asm_wg_threads=128), "asm_wg_threads"), (dict(no_escalation=2), "no_escalation"), (dict(submit_threads=65), "submit_threads"),
                      (dict(abi_version=4), "ABI version")):
        rc, text = create(**bad)
        assert rc == hb.BK_E_ARG and word in text, (bad, rc, text)
    rc, text = create(flags=hb.BK_CFG_NO_SPLIT | hb.BK_CFG_TEST_SPLIT_ALWAYS, asm_wg_threads=256, no_escalation=1, submit_threads=4)
    assert rc in (0, -3), (rc, text)
    assert lib.bk_last_error_region(None) == -1
