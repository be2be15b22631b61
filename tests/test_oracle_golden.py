"""The CPU oracle (oracle/bk_oracle.c) against golden vectors produced by the REAL reference
(tools/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from breakmer_amd import synth
from oracle import bk_oracle as bo


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_g1_nw_kats(golden_dir):
    d = _load(golden_dir, "nw_kats.json")
    assert len(d["cases"]) > 150
    for c in d["cases"]:
        assert list(bo.nw(c["seq1"], c["seq2"])) == c["out"], c["tag"]


def test_g2_check_align_branches(golden_dir):
    d = _load(golden_dir, "check_align.json")
    for c in d["cases"]:
        got = bo.check_align_case(c["contig"], c["read"], c["mer"], c["k"], c["mode"], c["nreads"], c["indel_only"],
                                  c["founder_nreads"], c["pre"])
        assert got["match"] == c["match"], (c["tag"], c["mode"])
        assert got["seq"] == c["seq"], (c["tag"], c["mode"])
        assert got["io"] == c["io"] and got["ot"] == c["ot"], (c["tag"], c["mode"])
        assert got["kmers"] == c["kmers"], (c["tag"], c["mode"])


def test_g4_kmer_select(golden_dir):
    d = _load(golden_dir, "kmer_select.json")
    for c in d["cases"]:
        mers, counts = bo.kmer_select(c["reads"], [c["ref"]], c["k"], c["sc"])
        assert dict(zip(mers, counts.tolist())) == c["mers"]


def _case_inputs(c):
    r = synth.make_region(**c["gen"])
    reads = r.read_strs()
    assert hashlib.sha256(("\n".join(reads)).encode()).hexdigest() == c["reads_sha256"], "generator drifted: " + c["tag"]
    return r, reads


def test_g3_assembly(golden_dir):
    d = _load(golden_dir, "assembly.json")
    assert len(d["cases"]) >= 14
    for c in d["cases"]:
        r, reads = _case_inputs(c)
        contigs, info = bo.assemble_region(reads, [r.window_str], c["k"], c["rc_thresh"], indel_only=r.indel_only)
        got = dict(zip(info["mers"], info["counts"].tolist()))
        assert len(got) == c["n_mers"], c["tag"]
        msum = hashlib.sha256(("\n".join("%s %d" % (m, got[m]) for m in sorted(got))).encode()).hexdigest()
        assert msum == c["mers_sha256"], c["tag"]
        assert contigs == c["contigs"], c["tag"]


def test_find_index_equals_scan(golden_dir):
    """The oracle's optional k-mer -> reads index behind find_reads (needed for full-size noisy regions) gives what the
    reference's scan over all reads gives: on every G3 fixture (outputs of the real reference) and on noisy regions with
    hundreds of contigs, short and ragged reads."""
    d = _load(golden_dir, "assembly.json")
    for c in d["cases"]:
        r, reads = _case_inputs(c)
        contigs, info = bo.assemble_region(reads, [r.window_str], c["k"], c["rc_thresh"], indel_only=r.indel_only, find_index=True)
        assert contigs == c["contigs"], c["tag"]
    for rid, kw, k in ((71, dict(depth=300, W=900, noise=0.01), 31), (72, dict(depth=80, W=700, L=250, noise=0.05), 41),
                       (73, dict(depth=150, W=1200, noise=0.02, var_len=0.4, sv_type="inv"), 21)):
        r = synth.make_region(rid, **kw)
        a, ia = bo.assemble_region(r.read_strs(), [r.window_str], k, 2)
        b, ib = bo.assemble_region(r.read_strs(), [r.window_str], k, 2, find_index=True)
        assert a == b and len(a) >= 2, rid
        assert (ia["flags"] == ib["flags"]).all()
