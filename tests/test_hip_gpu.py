"""GPU parity tests: the HIP path, called through the C-ABI, against the oracle and the golden
fixtures generated from the real reference.  Run with -m gpu on an MI355X."""
import hashlib
import json
import os
import random

import pytest

from breakmer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hb():
    from breakmer_amd import hip_backend
    # BK_TEST_VARIANT (test infrastructure, e.g. "checkjit"): the whole suite through a diagnostic build of the library
    # (breakmer_amd/build.py VARIANTS: barrier-site check, sleeping wavefronts); unset = the product build
    v = os.environ.get("BK_TEST_VARIANT", "")
    if v:
        from breakmer_amd import build
        hip_backend.load_library(build.lib_path(v))
    hip_backend.load_library()          # fails loudly if the extension is missing
    return hip_backend


def _race_check(args, timeout=1500):
    """tools/race_check.py in a child process (a device fault or a hang is then a red test, not a dead suite)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "race_check.py")] + args, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        pytest.fail("race_check %s hung (killed after %d s): %s" % (args, timeout, (e.stdout or b"")[-1500:]))
    assert r.returncode == 0 and "RACE CHECK RESULT: ok" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-1500:])
    return r.stdout


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_g1_nw_kats_gpu(hb, golden_dir):
    """olc.nw on the wavefront DP == the reference's 7-tuple (fields 2..6; 0,1 are the slices)."""
    d = _load(golden_dir, "nw_kats.json")
    eng = hb.Engine(kmer_size=31)
    pairs = [(c["seq1"], c["seq2"]) for c in d["cases"]]
    out, _ = eng.nw_batch(pairs)
    out_t, _ = eng.nw_batch(pairs, transposed=True)          # the transposed sweep used for nw(read, contig)
    assert out.tolist() == out_t.tolist()
    assert eng.nw_batch(pairs, transposed=2)[0].tolist() == out.tolist()      # suffix-restricted sweep used for nw(contig, read)
    small = [p for p in pairs if len(p[0]) <= 320]                            # both DPs on one wavefront (contig <= 320 columns)
    o3, o4 = eng.nw_batch(small, transposed=3)[0].tolist(), eng.nw_batch(small, transposed=4)[0].tolist()
    assert o3 == eng.nw_batch(small)[0].tolist() and o4 == eng.nw_batch([(b, a) for a, b in small])[0].tolist()
    # one score matrix with both tie-break orders, two pairs per wavefront (bk_nw_pair): pair i in half A, pair i+1 in half B
    rot = lambda x: x[1:] + x[:1]
    assert eng.nw_batch(small, transposed=5)[0].tolist() == o3 and eng.nw_batch(small, transposed=6)[0].tolist() == o4
    assert eng.nw_batch(small, transposed=7)[0].tolist() == rot(o3) and eng.nw_batch(small, transposed=8)[0].tolist() == rot(o4)
    # the score sweep (+ the full sweep for what it flags): two reads per wavefront, one read per wavefront
    assert eng.nw_batch(small, transposed=9)[0].tolist() == o3 and eng.nw_batch(small, transposed=10)[0].tolist() == o4
    assert eng.nw_batch(small, transposed=11)[0].tolist() == rot(o3) and eng.nw_batch(small, transposed=12)[0].tolist() == rot(o4)
    assert eng.nw_batch(small, transposed=13)[0].tolist() == o3 and eng.nw_batch(small, transposed=14)[0].tolist() == o4
    for c, o in zip(d["cases"], out.tolist()):
        exp = c["out"]
        assert o == [exp[3], exp[4], exp[5], exp[6]], c["tag"]
        # the aligned strings stripped of '-' are the plain slices (what check_align consumes)
        assert exp[0].replace("-", "") == c["seq1"][exp[3]:exp[2]]
        assert exp[1].replace("-", "") == c["seq2"][exp[5]:exp[4]]


def test_nw_random_vs_oracle_gpu(hb):
    from oracle import bk_oracle as bo
    rnd = random.Random(11)
    pairs = []
    for t in range(400):
        m, n = rnd.randint(1, 700), rnd.randint(1, 300)
        a = "".join(rnd.choice("ACGT") for _ in range(m))
        if rnd.random() < 0.7:
            ov = rnd.randint(1, min(m, n))
            b = a[m - ov:] + "".join(rnd.choice("ACGT") for _ in range(n - ov))
            b = "".join((ch if rnd.random() > 0.03 else rnd.choice("ACGT")) for ch in b)
        else:
            b = "".join(rnd.choice("ACGT") for _ in range(n))
        pairs.append((a, b))
        pairs.append((b, a))
    # column tiles: more than 512 columns, and a 4000-column case
    big = "".join(rnd.choice("ACGT") for _ in range(4000))
    pairs.append((big, big[3800:] + "ACGTACGT" * 10))
    pairs.append((big[:1500], big[1300:1500] + "TTTT" * 20))
    # long seq1 against short seq2 (the suffix-restricted sweep cuts columns there): planted overlaps at the end, in the
    # middle (must lose against the end), periodic and low-complexity sequences (ties everywhere), gaps near the cut
    for t in range(300):
        m, n = rnd.randint(40, 2500), rnd.randint(1, 160)
        alpha = "ACGT" if t % 3 else "AC"
        a = "".join(rnd.choice(alpha) for _ in range(m))
        if t % 5 == 0:
            a = ("ACGTTGCA" * 400)[:m]
        ov = rnd.randint(1, min(m, n))
        src = a[m - ov:] if t % 4 else a[max(0, m - ov - rnd.randint(0, 300)):][:ov]
        b = src + "".join(rnd.choice(alpha) for _ in range(n - len(src)))
        if t % 2:
            b = "".join((ch if rnd.random() > 0.05 else rnd.choice("ACGT")) for ch in b)
        if t % 7 == 0 and len(b) > 12:
            b = b[:5] + b[7:]                                # a gap
        pairs.append((a, b[:n] if b[:n] else "A"))
    eng = hb.Engine(kmer_size=31)
    out, _ = eng.nw_batch(pairs)
    out_t, _ = eng.nw_batch(pairs, transposed=True)
    out_s, _ = eng.nw_batch(pairs, transposed=2)
    # 18 / 19: the score sweep for contigs of any length (column tiles beyond 640 columns) + the full sweep for what it flags
    out_l, _ = eng.nw_batch(pairs, transposed=18)
    out_l2, _ = eng.nw_batch(pairs, transposed=19)
    for (a, b), o, ot, os_, ol, ol2 in zip(pairs, out.tolist(), out_t.tolist(), out_s.tolist(), out_l.tolist(), out_l2.tolist()):
        e = bo.nw(a, b)
        assert o == [e[3], e[4], e[5], e[6]], (len(a), len(b))
        assert ot == o, (len(a), len(b), "transposed")
        assert os_ == o, (len(a), len(b), "suffix")
        assert ol == o, (len(a), len(b), "score sweep, any length")
        if len(b) <= 1024:
            e2 = bo.nw(b, a)
            assert ol2 == [e2[3], e2[4], e2[5], e2[6]], (len(a), len(b), "score sweep, any length, nw(seq2, seq1)")
    # both overlap DPs of check_align on one wavefront: (contig, read) -> nw(contig, read) and nw(read, contig)
    rnd2 = random.Random(12)
    dual = [(a, b) for a, b in pairs if len(a) <= 320 and len(b) <= 400]
    for t in range(300):
        m, n = rnd2.randint(1, 320), rnd2.randint(1, 260)
        a = "".join(rnd2.choice("ACGT" if t % 3 else "AC") for _ in range(m))
        kind = t % 4
        ov = rnd2.randint(1, min(m, n))
        if kind == 0:
            b = a[m - ov:] + "".join(rnd2.choice("ACGT") for _ in range(n - ov))          # read hangs over the contig end
        elif kind == 1:
            b = "".join(rnd2.choice("ACGT") for _ in range(n - ov)) + a[:ov]              # ... over its start
        elif kind == 2:
            s0 = rnd2.randint(0, m - ov); b = a[s0:s0 + ov]                               # contained
        else:
            b = "".join(rnd2.choice("ACGT") for _ in range(n))
        if t % 2:
            b = "".join((ch if rnd2.random() > 0.04 else rnd2.choice("ACGTN")) for ch in b)
        dual.append((a, b if b else "A"))
    # odd and even read lengths, contigs around every register count (32 * C columns), the shapes of the headline
    for m in list(range(60, 321, 13)) + [96, 97, 128, 129, 160, 161, 288, 289, 319, 320]:
        for n in (149, 150, 151, 2, 1):
            a = "".join(rnd2.choice("ACGT") for _ in range(m))
            ov = min(m, n, rnd2.randint(1, 150))
            dual.append((a, (a[:ov][::-1][::-1] if (m + n) % 3 == 0 else a[m - ov:]) + "".join(rnd2.choice("ACGT") for _ in range(n - ov))))
    # overlaps with an indel in them (the score sweep cannot settle those: it must flag them) and with substitutions only (it can)
    for t in range(40):
        m = rnd2.randint(160, 320)
        a = "".join(rnd2.choice("ACGT") for _ in range(m))
        ov = rnd2.randint(80, 149)
        src = a[m - ov:]
        cut = rnd2.randint(20, ov - 20)
        src = (src[:cut] + src[cut + rnd2.randint(1, 3):]) if t % 2 else (src[:cut] + "ACG"[:rnd2.randint(1, 3)] + src[cut:])
        dual.append((a, (src + "".join(rnd2.choice("ACGT") for _ in range(150)))[:150]))
    want = [(bo.nw(a, b), bo.nw(b, a)) for a, b in dual]
    # 9..14: the score sweep of round 5 (one plain score matrix per read; the border cell without a traceback where the end
    # cell's score equals its diagonal; everything else swept again in full): two reads per wavefront (9..12), one (13, 14)
    for mode1, mode2, shift, tag in ((3, 4, 0, "dual"), (5, 6, 0, "pair, half A"), (7, 8, 1, "pair, half B"),
                                     (9, 10, 0, "score sweep, two reads per wavefront, half A"), (11, 12, 1, "score sweep, half B"), (13, 14, 0, "score sweep, one read per wavefront")):
        d1, _ = eng.nw_batch(dual, transposed=mode1)
        d2, _ = eng.nw_batch(dual, transposed=mode2)
        for q, (x1, x2) in enumerate(zip(d1.tolist(), d2.tolist())):
            (a, b), (e1, e2) = dual[(q + shift) % len(dual)], want[(q + shift) % len(dual)]
            assert x1 == [e1[3], e1[4], e1[5], e1[6]], (len(a), len(b), "v1", tag)
            assert x2 == [e2[3], e2[4], e2[5], e2[6]], (len(a), len(b), "v2", tag)
    # 15: the score sweep alone, as the assembler calls it: scores and end cells always the reference's; a border cell that
    # check_align looks at (bk_decide: none when both ok tests fail; else the winner's, both at equal scores) is either the
    # reference's or flagged -1 (the full sweep follows); and the sweep settles the exact overlaps by itself
    raw, _ = eng.nw_batch(dual, transposed=15)
    settled = flagged = 0
    for (a, b), (e1, e2), x in zip(dual, want, raw.tolist()):
        j1, s1, j2, s2 = x
        assert s1 == e1[6] and s2 == e2[6], (len(a), len(b))
        minlen = min(len(a), len(b))
        p1, p2 = e1[4] > 0 and 4 * s1 >= minlen, e2[4] > 0 and 4 * s2 >= minlen
        ok1 = p1 and 200 * s1 >= 179 * (len(a) - e1[3])
        ok2 = p2 and 200 * s2 >= 179 * (len(b) - e2[3])
        if not (ok1 or ok2):
            need = (False, False)                            # no match: check_align reads nothing else
        else:
            need = (s1 >= s2, s2 >= s1)                      # the call with the larger score decides (both at equal scores)
        for j, e, nd in ((j1, e1, need[0]), (j2, e2, need[1])):
            if j == -1:
                flagged += 1
            elif nd:
                settled += 1
                assert j == e[3], (len(a), len(b), j, e, need)
    assert settled > 300 and flagged >= 20, (settled, flagged)
    # substitutions only: settled by the sweep itself (the diagonal of the end cell scores d - 3 x = the end cell's score)
    subs = []
    for t in range(80):
        m = rnd2.randint(150, 320)
        a = "".join(rnd2.choice("ACGT") for _ in range(m))
        ov = rnd2.randint(60, 149)
        b = list(a[m - ov:] + "".join(rnd2.choice("ACGT") for _ in range(150 - ov)))
        for _ in range(rnd2.randint(1, 4)):
            q = rnd2.randint(5, ov - 5)
            b[q] = "ACGT"[("ACGT".index(b[q]) + 1) % 4]
        subs.append((a, "".join(b)))
    rs, _ = eng.nw_batch(subs, transposed=15)
    ws = [(bo.nw(a, b), bo.nw(b, a)) for a, b in subs]
    nflag = sum(1 for x in rs.tolist() if -1 in (x[0], x[2]))
    assert nflag <= 8, nflag                                 # (a substitution next to the overlap's start can make a gapped path co-optimal: those are flagged)
    for (a, b), (e1, e2), x in zip(subs, ws, rs.tolist()):
        assert x[1] == e1[6] and x[3] == e2[6]
        if x[0] != -1 and e1[6] > e2[6]:
            assert x[0] == e1[3], (x, e1)
    exact = [(a, b) for (a, b), (e1, e2) in zip(dual, want) if e1[6] == min(e1[4], len(a)) and e1[4] > 0]
    r2, _ = eng.nw_batch(exact, transposed=15)
    assert len(exact) > 100 and all(x[0] != -1 for x in r2.tolist())
    # the headline's reads: the read hangs over the contig end by one base (or starts one base before it): nw(read, contig) then
    # ends in a gap and loses -- nothing is flagged
    rnd3 = random.Random(13)
    ext = []
    for t in range(60):
        m = rnd3.randint(150, 320)
        a = "".join(rnd3.choice("ACGT") for _ in range(m))
        k_ = rnd3.randint(1, 3)
        ext.append((a, a[m - (150 - k_):] + "".join(rnd3.choice("ACGT") for _ in range(k_))))
        ext.append((a, "".join(rnd3.choice("ACGT") for _ in range(k_)) + a[:150 - k_]))
    r3, _ = eng.nw_batch(ext, transposed=15)
    assert all(x[0] != -1 and x[2] != -1 for x in r3.tolist()), [x for x in r3.tolist() if -1 in (x[0], x[2])][:3]


def _run_regions(hb, regions, k, rc_thresh=2, stages=3, **limits):
    eng = hb.Engine(kmer_size=k, rc_thresh=rc_thresh, **limits)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only,
                               partners=[p[4] for p in r.partners]) for r in regions])
    eng.run(stages)
    return eng


def _strip(contigs):
    return [{k: v for k, v in c.items() if k not in ("total_reads", "n_hits")} for c in contigs]


def test_g3_assembly_golden_gpu(hb, golden_dir):
    """k-mer selection + assembly on the GPU == contigs the REAL reference produced (fixtures)."""
    d = _load(golden_dir, "assembly.json")
    by_cfg = {}
    for c in d["cases"]:
        by_cfg.setdefault((c["k"], c["rc_thresh"]), []).append(c)
    for (k, rc), cases in by_cfg.items():
        regions = [synth.make_region(**c["gen"]) for c in cases]
        for c, r in zip(cases, regions):
            assert hashlib.sha256(("\n".join(r.read_strs())).encode()).hexdigest() == c["reads_sha256"]
        eng = _run_regions(hb, regions, k, rc)
        for i, c in enumerate(cases):
            mers, counts, U = eng.kmers(i)
            assert len(mers) == c["n_mers"], c["tag"]
            got = dict(zip(mers, counts.tolist()))
            msum = hashlib.sha256(("\n".join("%s %d" % (m, got[m]) for m in sorted(got))).encode()).hexdigest()
            assert msum == c["mers_sha256"], c["tag"]
            # visit order of init_assembly: (count, mer) descending
            assert mers == [m for m, _ in sorted(got.items(), key=lambda x: (x[1], x[0]), reverse=True)], c["tag"]
            assert _strip(eng.contigs(i)) == c["contigs"], c["tag"]


def test_both_workgroup_sizes_gpu(hb, golden_dir):
    """The assembler exists in two workgroup sizes (512 threads / 8 look-ahead slots, 256 threads / 4 slots; bk_config
    asm_wg_threads): the reference fixtures, a noisy region and long contigs give the same records through both."""
    from oracle import bk_oracle as bo
    d = _load(golden_dir, "assembly.json")
    cases = [c for c in d["cases"] if c["k"] == 31 and c["rc_thresh"] == 2]
    regions = [synth.make_region(**c["gen"]) for c in cases]
    extra = [synth.make_region(970, sv_type="trl", depth=150, W=2400), synth.make_region(971, sv_type="del", depth=300, W=1000, noise=0.01)]
    for wg in (256, 512):
        eng = _run_regions(hb, regions + extra, 31, stages=7, wg_threads=wg)
        assert eng.stat(25) == wg
        for i, c in enumerate(cases):
            assert _strip(eng.contigs(i)) == c["contigs"], (wg, c["tag"])
        for j, r in enumerate(extra):
            want, _ = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
            assert _strip(eng.contigs(len(cases) + j)) == want and len(want) >= 1, (wg, j)


def test_split_regions_are_bit_identical_gpu(hb, golden_dir):
    """The shipped path of noisy regions (default since round 5; flag 128 = one unit per region, as everything ran until round 4).
    Noisy regions are split over up to 16 assembler workgroups (bk_comp.hip.h): unit 0 runs the high-count seeds (the SV's own
    k-mers) alone and in order, then the components of what is left of the read / k-mer graph are dealt to the units;
    components that meet across units are merged and run again.  The result must be the serial one:
    (a) every reference fixture (G3) with the split FORCED on its small graph (flag 256), both workgroup sizes;
    (b) a mixed batch with mid-size noisy regions against the oracle, split forced;
    (c) full-size regions at 0.2 % / 0.5 % / 1 % substitutions, split (the default) against one unit (flag 128): the same
        contigs in the same order, the same realign records -- and the split really happened, with repair passes: inside the
        assembler kernel (the last unit of a region merges, re-deals and re-queues: the default) and driven by the host (flag 4096:
        the fallback when the unit queue has no room)."""
    from oracle import bk_oracle as bo
    d = _load(golden_dir, "assembly.json")
    by_cfg = {}
    for c in d["cases"]:
        by_cfg.setdefault((c["k"], c["rc_thresh"]), []).append(c)
    nsplit = 0
    for (k, rc), cases in by_cfg.items():
        regions = [synth.make_region(**c["gen"]) for c in cases]
        for wg in (256, 512):
            eng = _run_regions(hb, regions, k, rc, flags=256, wg_threads=wg)
            assert eng.sync() == 0
            for i, c in enumerate(cases):
                assert _strip(eng.contigs(i)) == c["contigs"], (c["tag"], wg)
            nsplit += eng.stat(28)
            eng.close()
    assert nsplit > 0
    regions = [synth.make_region(600 + i, sv_type=synth.SV_TYPES[i % 5], depth=(200, 300)[i % 2], W=1200, noise=(0.004, 0.008, 0.015)[i % 3]) for i in range(9)]
    eng = _run_regions(hb, regions, 31, stages=7, flags=256)
    assert eng.sync() == 0 and eng.stat(28) >= 6
    for i, r in enumerate(regions):
        want, _ = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        got = eng.contigs(i)
        assert _strip(got) == want, (i, len(got), len(want))
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        for ci in range(0, len(got), 7):
            assert eng.hits(i, ci) == bo.realign(got[ci]["seq"], targets), (i, ci)
    eng.close()
    full = [synth.make_region(50000 + i, depth=500, L=150, sv_type=("del", "ins", "inv")[i % 3], noise=nz) for i, nz in enumerate((0.002, 0.005, 0.005, 0.005, 0.01, 0.0))]
    one = _run_regions(hb, full, 31, stages=7, flags=128)
    many = _run_regions(hb, full, 31, stages=7)
    assert one.sync() == 0 and many.sync() == 0
    assert one.stat(28) == 0 and many.stat(28) >= 4, many.stat(28)          # the clean region and the percolated one (1 %) stay one unit
    assert many.stat(27) >= 1 and many.stat(29) == 0                         # components met across units: at least one repair pass, all of them inside the assembler kernel
    assert many.stat(1) >= one.stat(1)                                       # (the DPs of components that ran again are counted too)
    host = _run_regions(hb, full, 31, stages=7, flags=4096)                  # the fallback: repair passes driven by the host (no room in the unit queue)
    assert host.sync() == 0 and host.stat(28) >= 4 and host.stat(29) >= 1
    preq = _run_regions(hb, full, 31, stages=7, flags=hb.BK_CFG_TEST_PREQUEUE_UNITS)      # the round-5 queue: the units of a region are entries of the launch and wait for unit 0
    assert preq.sync() == 0 and preq.stat(28) >= 4
    for i in range(len(full)):
        a, b, c, e = one.contigs(i), many.contigs(i), host.contigs(i), preq.contigs(i)
        assert len(a) == len(b) and a == b and c == a and e == a, (i, len(a), len(b), len(c), len(e))
        for ci in range(0, len(a), 50):
            assert one.hits(i, ci) == many.hits(i, ci) == host.hits(i, ci) == preq.hits(i, ci), (i, ci)
    # (d) ... and DIRECTLY against the oracle at full size (round 6: until then full-size noisy regions were only held split-vs-one-unit
    #     and run-vs-run; the oracle comparison of noisy regions stopped at depth 300): the 0.2 % region and the three 0.5 % regions,
    #     10,000 reads each, ~1,100-1,200 contigs per region, every field of every contig; realign records of every 40th contig
    for i in (0, 1, 2, 3):
        r = full[i]
        want, _ = bo.assemble_region(synth.BASES[r.reads], [r.window_str], 31, 2, find_index=True)
        got = many.contigs(i)
        assert len(got) == len(want) and len(want) > 300, (i, len(got), len(want))
        assert _strip(got) == want, i
        for ci in range(0, len(got), 40):
            assert many.hits(i, ci) == bo.realign(got[ci]["seq"], [r.window_str]), (i, ci)


def test_split_batches_on_concurrent_handles_gpu(hb):
    """with the unit queue of round 5 a workgroup that found the queue empty waited for *pending while the first
    entries were tied to workgroups by block index -- two assembler kernels of two handles, each only partly resident, could wait for
    each other for ever.  Since round 6 every entry of a batch with split regions is handed out through the queue head (bk_sched.hip.h).
    Four handles, each with a batch whose units (64 regions x 16) exceed what the chip holds resident, split forced, launched back to
    back and waited for together, three times over; a fifth handle runs the same regions with the round-5 queue order
    (BK_CFG_TEST_PREQUEUE_UNITS) in the same crowd.  Every handle must come back -- the test runs under the suite's timeout -- with the
    records of a handle that ran alone, and those are the oracle's."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(8800 + i, sv_type=synth.SV_TYPES[i % 5], depth=(120, 200)[i % 2], W=1200, noise=(0.004, 0.008, 0.006)[i % 3]) for i in range(64)]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regions]
    alone = hb.Engine(kmer_size=31, flags=hb.BK_CFG_TEST_SPLIT_ALWAYS)
    alone.submit(ins); alone.run(7)
    assert alone.stat(22) == 0 and alone.stat(28) >= 48
    ref = [alone.contigs(i) for i in range(len(regions))]
    for i in range(0, len(regions), 9):
        want, _ = bo.assemble_region(regions[i].read_strs(), [regions[i].window_str], 31, 2)
        assert _strip(ref[i]) == want, i
    engs = [hb.Engine(kmer_size=31, flags=hb.BK_CFG_TEST_SPLIT_ALWAYS | (hb.BK_CFG_TEST_PREQUEUE_UNITS if j == 4 else 0), wg_threads=(512, 256)[j % 2]) for j in range(5)]
    for e in engs:
        e.submit(ins)
    for rep in range(3):
        for e in engs:
            e.run(7, sync=False)
        for j, e in enumerate(engs):
            assert e.sync() == 0, (rep, j)
            assert e.stat(28) >= 48
        for j, e in enumerate(engs):
            for i in range(rep, len(regions), 5):
                assert e.contigs(i) == ref[i], (rep, j, i)
    for e in engs + [alone]:
        e.close()


def test_barrier_discipline_of_every_kernel_gpu():
    """Every workgroup barrier of the library is BK_SYNC() (bk_common.h).  The `checkjit` build verifies at EVERY barrier that all
    wavefronts of the workgroup stand at the same barrier site (a divergence is reported by bk_sync with both sites) while a
    pseudo-random subset of the wavefronts sleeps behind each barrier -- legal at any time, so whatever goes wrong with it is a
    race.  Three seeds (which wavefront sleeps where): the reference fixtures on both workgroup sizes, as shipped and with the
    component split forced; a mixed batch of small regions against the oracle incl. the realign records; regions that overflow a
    working cap (the give-up paths).  This is the class of defect found by accident in rounds 2, 3 and 4 (bk_retire twice,
    bk_find_reads) and twice more in round 5: the S->foreign loop of bk_kmers_ordered (found by reading FOR the rule) and bk_fail's
    bare write of S->status on the contig-overflow path of bk_retire (found by this build: BK_TEST_VARIANT=checkjit runs the WHOLE
    suite through it)."""
    from breakmer_amd import build
    # (built by __graft_entry__.build() and shipped with the tree: a missing library is a failure, never a compile on the GPU box)
    assert os.path.isfile(build.lib_path("checkjit")), "libbreakmer_hip_checkjit.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` before the GPU suite"
    out = _race_check(["--variant", "checkjit", "--seeds", "1,2,3", "g3", "mixed", "caps"])
    assert out.count("ok    ") >= 3 * (2 + 4 + 4)


def test_redo_passes_of_long_contig_rounds_gpu():
    """Round 6: a look-ahead round over a contig beyond the dual / pair kernels' 320 columns has a slot per wavefront while the score sweep is on;
    the reads the sweep cannot settle are swept again in full afterwards, two wavefronts per read, BK_SPEC_WIDE slots per pass (bk_dp_redo).  On
    real data that is ~2 % of such reads -- hardly ever more than one pass.  With BK_CFG_DIAG_FORCE_REDO (accepted by diagnostic builds only: this
    runs through the barrier-check + jitter build, in a child process) EVERY read of those rounds is flagged: all passes run, and the records must
    still be the oracle's, on both workgroup sizes; the counter of reads swept again must show it."""
    from breakmer_amd import build
    assert os.path.isfile(build.lib_path("checkjit")), "libbreakmer_hip_checkjit.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` before the GPU suite"
    out = _race_check(["--variant", "checkjit", "--seeds", "1", "redo"])
    assert out.count("ok    ") == 4


def test_score_sweep_comes_back_on_gpu(hb):
    """Round 6: the score sweep switches itself off while more than half of the reads of its window have to be swept again in full -- and, until
    this round, stayed off for the rest of the region, because nothing fed the window any more.  Region 50215 (0.5 % noise; the one whose serial
    prefix bounds the 256-region noisy batch) hit such a stretch among its first reads and then ran 26,000 of its 27,400 slots through the full
    DPs.  Now the window starts afresh after BK_SWEEP_RETRY rounds: nearly every read goes through the sweep, and the contigs are the oracle's."""
    from oracle import bk_oracle as bo
    r = synth.make_region(50215, sv_type="del", depth=500, W=3000, L=150, noise=0.005)
    for wg in (512, 256):
        eng = _run_regions(hb, [r], 31, stages=3, flags=128, wg_threads=wg)
        assert eng.sync() == 0
        reads_aligned, swept, again = eng.stat(1) // 2, eng.stat(30), eng.stat(31)
        assert swept >= 0.9 * reads_aligned and again <= 0.1 * swept, (wg, reads_aligned, swept, again)
        if wg == 512:
            want, _ = bo.assemble_region(synth.BASES[r.reads], [r.window_str], 31, 2, find_index=True)
            assert _strip(eng.contigs(0)) == want and len(want) > 1000
        eng.close()


def test_batches_that_faulted_in_round_4_gpu():
    """The shape that faulted or hung until the barrier fixes (profiles/r04/split_fault/README.md; no test had it): several workgroups
    of small NOISY regions per CU.  720 and 1,024 regions at 1 % noise, four runs each on both workgroup sizes: every run gives the
    records of the first, eight sampled regions equal the oracle; and full-size noisy regions split over 16 workgroups each
    against the one-unit run.  Product build, in a child process."""
    out = _race_check(["shape:720:4", "shape:1024:4", "noisy:6"])
    assert out.count("ok    ") == 6


def test_batch_vs_oracle_gpu(hb):
    """A mixed batch (config-2-like regions at reduced depth) against the C oracle."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(100 + i, sv_type=synth.SV_TYPES[i % 4], depth=100, W=3000) for i in range(16)]
    eng = _run_regions(hb, regions, 31)
    for i, r in enumerate(regions):
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        assert _strip(eng.contigs(i)) == want, i
    assert eng.stat(0) > 0 and eng.stat(1) > 0


def test_realign_vs_oracle_gpu(hb):
    """R2: the realign kernel + host chaining == the oracle's contract (BLAT parity itself is unpinned)."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(200 + i, sv_type=synth.SV_TYPES[i % 5], depth=60, W=1500, noise=(0.02 if i >= 10 else 0.0)) for i in range(14)]
    eng = _run_regions(hb, regions, 31, stages=7)
    nrec = 0
    for i, r in enumerate(regions):
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        for ci, c in enumerate(eng.contigs(i)):
            want = bo.realign(c["seq"], targets)
            got = eng.hits(i, ci)
            assert got == want, (i, ci, r.sv_type)
            nrec += len(got)
    assert nrec > 14 and eng.stat(2) > 0


def _multihit_regions():
    mk = lambda i, **kw: synth.make_region(i, depth=60, W=1500, **kw)
    return [mk(3, sv_type="del", flank_dups=1), mk(3, sv_type="del", flank_dups=2), mk(3, sv_type="del", flank_dups=3), mk(3, sv_type="del", flank_dups=6),
            mk(3, sv_type="del", flank_dups=7), mk(3, sv_type="trl", trl_repeat_copies=5), mk(5, sv_type="trl", trl_repeat_copies=3),
            synth.make_region(3, depth=60, W=3000, microsat=100), synth.make_region(4, depth=60, W=3000, microsat=250),
            mk(7, sv_type="del", flank_dups=3, noise=0.01), mk(8, sv_type="del", flank_dups=5, noise=0.02), mk(3, sv_type="del")]


def test_realign_secondary_alignments_vs_oracle_gpu(hb):
    """R2 steps 5-6: secondary alignments (every other gap-free segment >= min_score: duplicated flanks on either strand, a
    repeated partner half, a microsatellite next to the junction with ~100-200 of them) and the placement of ambiguous hits --
    the kernel's sweep with its word-granular pruning + the host's chaining == the oracle's brute force, record for record."""
    from oracle import bk_oracle as bo
    regions = _multihit_regions()
    eng = _run_regions(hb, regions, 31, stages=7)
    assert eng.sync() == 0
    nrec = 0
    for i, r in enumerate(regions):
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        cs = eng.contigs(i)
        assert cs, i
        for ci, c in enumerate(cs):
            want = bo.realign(c["seq"], targets)
            got = eng.hits(i, ci)
            assert got == want, (i, ci)
            nrec += len(got)
        if i < 9:
            assert max(len(eng.hits(i, ci)) for ci in range(len(cs))) >= 2, i       # these really have secondary alignments
    assert nrec > 300


def test_g8m_multi_mapping_rows_gpu(hb, golden_dir):
    """The rows the REAL reference's caller made of multi-mapping contigs (tests/golden/realign_multihit.json) from the GPU
    path end to end: assembly + realign with secondary alignments on the device, native call tail."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from breakmer_amd.sv_processor import params as bk_params
    d = _load(golden_dir, "realign_multihit.json")
    tags = {c["tag"]: c for c in d["cases"]}
    mk = lambda **kw: synth.make_region(3, depth=60, W=1500, **kw)
    cases = [("del_unique", mk(sv_type="del")), ("del_left_flank_dup", mk(sv_type="del", flank_dups=1)), ("del_both_flanks_dup", mk(sv_type="del", flank_dups=3)),
             ("del_right_flank_dup_rc", mk(sv_type="del", flank_dups=6)), ("del_right_flank_dup", mk(sv_type="del", flank_dups=2)),
             ("trl_partner_repeat_disc", mk(sv_type="trl", trl_repeat_copies=5)),
             # contract step 8 (BLAT's -minIdentity default 90): the ~80 % copy of the flank is no record, the ~93 % copy is
             ("del_left_flank_diverged_copy_80pct", mk(sv_type="del", flank_dups=1, flank_div=4)), ("del_left_flank_diverged_copy_93pct", mk(sv_type="del", flank_dups=1, flank_div=12))]
    regions = [r for _t, r in cases]
    opts = dict(bk_params.DEFAULTS); opts["var_filter"] = ["indel", "rearrangement", "trl"]
    eng = hb.Engine(kmer_size=31)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regions])
    eng.set_call_context(bench.call_context_text(regions, opts))
    eng.run(hb.BK_STAGE_ALL)
    rows = eng.call()
    for i, (tag, r) in enumerate(cases):
        c = tags[tag]
        cs = eng.contigs(i)
        assert len(cs) == 1 and cs[0]["seq"] == c["contig"]["seq"], tag
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        assert targets == c["targets"], tag
        assert eng.hits(i, 0) == c["contract"]["records"], tag
        want = c["contract"]["expected"]
        got = rows.get(i, [])
        assert got == ([want] if want is not None else []), tag


def test_realign_has_no_hit_caps_gpu(hb):
    """The reference's BLAT prints as many alignments as there are.  (a) A 600-base microsatellite: ~350 secondary alignments of one
    contig (more than the 256 the kernel collects in LDS: they spill to the result arena) -- every record equals the oracle's,
    the neighbours in the batch are untouched, no region fails.  (b) A contig with more step-1 hits than the 32 the library used
    to keep: a window made of the contig's 24-base pieces in shuffled order with spacers; (c) a chained record with more than the
    32 blocks bk_psl holds: the same pieces in order, one base deleted between neighbours -- bk_get_hits_flat returns it whole,
    bk_get_hits says BK_E_LIMIT for that contig only."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(3, depth=60, W=1500), synth.make_region(3, depth=60, W=3000, microsat=600), synth.make_region(4, depth=60, W=1500, sv_type="ins")]
    want1 = bo.realign(bo.assemble_region(regions[1].read_strs(), [regions[1].window_str], 31, 2)[0][0]["seq"], [regions[1].window_str])
    assert len(want1) > 257
    eng = hb.Engine(kmer_size=31)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    eng.run(hb.BK_STAGE_ALL, sync=False)
    assert eng.sync() == 0 and eng.stat(22) == 0
    for i, r in enumerate(regions):
        assert eng.region_status(i) == (0, "ok")
        cs = eng.contigs(i)
        assert cs
        for ci, c in enumerate(cs):
            assert eng.hits(i, ci) == bo.realign(c["seq"], [r.window_str]), (i, ci)
    assert eng.hits(1, 0) == want1
    # (b), (c): a 1,040-base sequence of 40 random 26-mers; the reads are its own 150-mers, the window is built from its pieces
    rnd = random.Random(5)
    sp = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    q = "".join(sp(26) for _ in range(40))
    lp = [q[i:i + 26] for i in range(0, len(q), 26)]
    order = list(range(len(lp))); rnd.shuffle(order)
    win_b = sp(40) + "".join(lp[j] + sp(30) for j in order)       # shuffled, 30 random bases between them: every piece is a step-1 hit of its own
    win_c = sp(50) + "A".join(lp) + sp(50)                        # in order, one extra window base between neighbours: ONE chained record, a block per piece
    reads = [q[i:i + 150] for i in range(0, len(q) - 149, 3)] * 2
    for name, win in (("many hits", win_b), ("many blocks", win_c)):
        e2 = hb.Engine(kmer_size=31)
        e2.submit([hb.RegionInput(reads, win)])
        e2.run(hb.BK_STAGE_ALL, sync=False)
        assert e2.sync() == 0, name
        cs = e2.contigs(0)
        assert cs and max(len(c["seq"]) for c in cs) > 1000, name
        seen = 0
        for ci, c in enumerate(cs):
            w2 = bo.realign(c["seq"], [win])
            got = e2.hits(0, ci)
            assert got == w2, (name, ci)
            if name == "many hits":
                seen += sum(len(x["block_sizes"]) for x in w2)
            else:
                nb = max([len(x["block_sizes"]) for x in w2] + [0])
                seen = max(seen, nb)
                if nb > 32:
                    with pytest.raises(hb.BreakmerHipError):
                        e2.hits_fixed(0, ci)
                else:
                    assert e2.hits_fixed(0, ci) == w2
        assert seen > (20 if name == "many hits" else 32), (name, seen)
        e2.close()


def test_soft_masked_window_reports_rep_matches_gpu(hb):
    """BLAT is run with -repeats=lower (sv_processor.py:843): matches on lower-case target bases are PSL column 3 (repMatches),
    which the caller reads (sv_caller.py:913, 975-986).  A window whose left flank is soft-masked: same contigs and hits as the
    upper-case window, matches split into matches + rep_matches exactly as the oracle's contract does; the native call tail
    and the Python tail agree on the rows."""
    from oracle import bk_oracle as bo
    r = synth.make_region(3, depth=60, W=1500)
    w = r.window_str
    c = len(w) // 2
    soft = w[:c - 160] + w[c - 160:c - 100].lower() + w[c - 100:]
    e_up, e_lo = hb.Engine(kmer_size=31), hb.Engine(kmer_size=31)
    for e, win in ((e_up, w), (e_lo, soft)):
        e.submit([hb.RegionInput(r.reads, win, read_lens=r.read_lens)])
        e.run(hb.BK_STAGE_ALL)
    assert e_up.contigs(0) == e_lo.contigs(0) and len(e_lo.contigs(0)) == 1
    up, lo = e_up.hits(0, 0), e_lo.hits(0, 0)
    assert lo == bo.realign(e_lo.contigs(0)[0]["seq"], [soft])
    assert up == bo.realign(e_up.contigs(0)[0]["seq"], [w])
    assert sum(x["rep_matches"] for x in up) == 0 and lo[0]["rep_matches"] == 60
    for a, b in zip(up, lo):
        assert a["matches"] == b["matches"] + b["rep_matches"]
        assert {k: v for k, v in a.items() if k not in ("matches", "rep_matches")} == {k: v for k, v in b.items() if k not in ("matches", "rep_matches")}


def test_regions_that_overflow_a_cap_are_rerun_with_larger_caps_gpu(hb):
    """The reference has no caps (find_reads sv_assembly.py:111-122, contig growth :506-546).  (a) 6,000x of ragged reads: more
    than 3,000 unique reads hold the seed k-mer (default cap 2,048 candidates per visit); (b) a 5,000-base insertion: a contig of
    more than 4,096 bases.  Both overflow the default LDS-sized caps, are run again by the library under the larger ones and
    come out bit-exact against the oracle; the clean neighbour is untouched; with the re-run switched off they fail loudly."""
    from oracle import bk_oracle as bo
    deep = synth.make_region(21, depth=6000, W=800, var_len=1.0)
    longc = synth.make_region(22, sv_type="ins", sv_size=5000, W=1200, n_reads=1600)
    plain = synth.make_region(3, depth=60, W=1500)
    regions = [deep, plain, longc]
    wants = [bo.assemble_region(r.read_strs(), [r.window_str], 31, 2) for r in regions]
    # the seed k-mer of (a): how many unique reads hold it
    seed = wants[0][1]["mers"][0]
    assert len({s for s in deep.read_strs() if seed in s}) >= 3000
    assert max(len(c["seq"]) for c in wants[2][0]) > 4096
    eng = hb.Engine(kmer_size=31)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    eng.run(hb.BK_STAGE_ALL, sync=False)
    assert eng.sync() == 0 and eng.stat(22) == 0 and eng.stat(26) == 2
    for i, r in enumerate(regions):
        assert eng.region_status(i) == (0, "ok")
        got = eng.contigs(i)
        assert _strip(got) == wants[i][0], i
        for ci, c in enumerate(got):
            assert eng.hits(i, ci) == bo.realign(c["seq"], [r.window_str]), (i, ci)
    off = hb.Engine(kmer_size=31, no_escalation=1)
    off.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    off.run(hb.BK_STAGE_ALL, sync=False)
    assert off.sync() == 2
    assert off.region_status(0)[0] == 4 and off.region_status(2)[0] == 3 and off.region_status(1) == (0, "ok")
    assert off.contigs(0) == [] and _strip(off.contigs(1)) == wants[1][0]


def test_runner_end_to_end_gpu(hb, golden_dir, tmp_path):
    """breakmer.py-level run on the GPU: config file + BED + annotation -> runner.run() -> the same
    13-field rows the REAL reference's caller produced for these contigs (tests/golden/caller.json),
    and the per-target / run-level output files."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_pipeline import make_inputs
    from breakmer_amd import sv_processor as sp
    gold = {c["tag"]: c["expected"] for c in _load(golden_dir, "caller.json")["cases"]}
    cfg, data = make_inputs(tmp_path, [(3, "del")])
    rows = sp.runner(cfg, region_data=data).run()
    assert rows == [gold["del_indelmode_c0"]]
    (tmp_path / "py").mkdir()
    cfg2, data2 = make_inputs(tmp_path / "py", [(3, "del")])
    assert sp.runner(cfg2, region_data=data2, native_calls=False).run() == rows          # Python tail == native tail
    for sv, tag in (("ins", "ins"), ("inv", "inv_disc"), ("dup", "dup"), ("trl", "trl")):
        d = tmp_path / sv
        d.mkdir()
        cfg, data = make_inputs(d, [(3, sv)])
        rows = sp.runner(cfg, region_data=data).run()
        assert rows == [gold[k] for k in sorted(gold) if k.startswith(tag + "_c") and gold[k] is not None], sv
        (d / "py").mkdir()
        cfg2, data2 = make_inputs(d / "py", [(3, sv)])
        assert sp.runner(cfg2, region_data=data2, native_calls=False).run() == rows, sv
    out = tmp_path / "analysis" / "output"
    assert (out / "synth_indel_svs.out").is_file() and (out / "synth_summary.out").is_file()
    # N1: per-target and per-contig files of the reference (sv_processor.py:668-683, 747-799), native and Python tail alike
    for base in (tmp_path, tmp_path / "py"):
        tdir = base / "analysis" / "targets" / "GENE00003"
        assert (tdir / "contigs" / "contig1" / "contig1.fa").read_text().startswith(">contig1\n")
        assert (tdir / "contigs" / "contig1" / "contig1.fq").read_text().count("\n+\n") >= 2
        assert (tdir / "contigs" / "contig1" / "contig1_svs.out").read_text().split("\t")[6] == "indel"
        assert (tdir / "kmers" / "GENE00003_sample_kmers_merged.out").read_text().startswith("contig1 ")
        o = base / "analysis" / "output" / "GENE00003"
        assert (o / "GENE00003_indel_svs.out").read_text().splitlines()[1].split("\t")[1].endswith("(D200)")
        assert (o / "contig1_svs.out").is_file()


def test_runner_from_alignment_file_gpu(hb, tmp_path):
    """N2 -> hot path -> call on the GPU: reads selected from a SAM file (read_extraction.py, pinned by G6), both k=15
    (the reference's default config) and both call tails."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_pipeline import check_sam_run, make_sam_inputs
    from breakmer_amd import sv_processor as sp
    cfg, r = make_sam_inputs(tmp_path)
    rows = sp.runner(cfg).run()
    check_sam_run(rows, r, tmp_path)
    (tmp_path / "py").mkdir()
    cfg2, r2 = make_sam_inputs(tmp_path / "py")
    assert sp.runner(cfg2, native_calls=False).run() == rows


def test_cli_end_to_end_gpu(hb, tmp_path):
    """`python -m breakmer_amd.breakmer -a <config>` (breakmer.py:50-96): key=value config file on disk, reads from
    the alignment file, the reference's output files written."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_pipeline import make_sam_inputs
    from breakmer_amd import breakmer
    cfg, r = make_sam_inputs(tmp_path)
    cfg.pop("keep_repeat_regions")
    (tmp_path / "run.cfg").write_text("".join("%s=%s\n" % kv for kv in cfg.items()))
    breakmer.main(["-a", "-l", "ERROR", str(tmp_path / "run.cfg")])
    out = tmp_path / "analysis" / "output"
    lines = (out / "fromsam_indel_svs.out").read_text().splitlines()
    assert lines[0].split("\t")[0] == "genes" and len(lines) == 2
    f = lines[1].split("\t")
    assert f[0] == r.name and f[1].endswith("(D120)") and f[6] == "indel" and max(int(x) for x in f[10].split(",")) > 0
    summ = (out / "fromsam_summary.out").read_text().splitlines()
    assert len(summ) == 2 and summ[1].split("\t")[0] == r.name
    assert (out / r.name / (r.name + "_indel_svs.out")).is_file()


def test_g4_kmer_select_gpu(hb, golden_dir):
    """K1/K2 incl. a separate soft-clip set (case_sc) on the GPU == the reference's set algebra (G4)."""
    d = _load(golden_dir, "kmer_select.json")
    by_k = {}
    for c in d["cases"]:
        by_k.setdefault(c["k"], []).append(c)
    for k, cases in by_k.items():
        eng = hb.Engine(kmer_size=k)
        eng.submit([hb.RegionInput(c["reads"], c["ref"], sc_seqs=c["sc"]) for c in cases])
        eng.run(hb.BK_STAGE_KMER)
        for i, c in enumerate(cases):
            mers, counts, U = eng.kmers(i)
            assert dict(zip(mers, counts.tolist())) == c["mers"], (k, i)
            assert U == len(set(c["reads"]))


def test_edge_cases_gpu(hb):
    """Empty region, region without SV (no sample k-mers), ragged reads shorter than k, one-read region,
    all in one batch next to a normal region; and loud failures for the documented limits."""
    from oracle import bk_oracle as bo
    normal = synth.make_region(3, sv_type="del", depth=60, W=1500)
    nosv = synth.make_region(9, sv_type="del", sv_size=0, depth=40, W=900)
    w = normal.window_str
    ins = [hb.RegionInput(normal.reads, normal.window, read_lens=normal.read_lens),
           hb.RegionInput([], w),
           hb.RegionInput(nosv.reads, nosv.window, read_lens=nosv.read_lens),
           hb.RegionInput([w[100:120], w[300:450], "ACGT" * 10 + w[500:600], w[300:450]], w),
           hb.RegionInput([w[10:160]], w)]
    eng = hb.Engine(kmer_size=31)
    eng.submit(ins)
    eng.run(hb.BK_STAGE_ALL)
    want, _ = bo.assemble_region(normal.read_strs(), [w], 31, 2)
    assert _strip(eng.contigs(0)) == want and len(want) == 1
    for r in (1, 2, 4):
        assert eng.contigs(r) == []
    assert eng.kmers(1)[0] == [] and eng.kmers(2)[0] == []
    reads3 = [w[100:120], w[300:450], "ACGT" * 10 + w[500:600], w[300:450]]
    want3, info3 = bo.assemble_region(reads3, [w], 31, 2)
    mers3, counts3, U3 = eng.kmers(3)
    assert dict(zip(mers3, counts3.tolist())) == dict(zip(info3["mers"], info3["counts"].tolist())) and U3 == 3
    assert _strip(eng.contigs(3)) == want3
    # limits fail loudly
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=31).submit([hb.RegionInput(["ACGTR" * 10], w)])           # a character other than A/C/G/T/N
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=31).submit([hb.RegionInput([w[:100]], w[:500] + "Y" + w[501:])])   # ... in a window too (an N is fine there: test_windows_with_n_gpu)
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=31).submit([hb.RegionInput(["A" * 2000], w)])             # read longer than max_read_len
    e5 = hb.Engine(kmer_size=31)                                                      # 20 kb window of a 4-mer repeat: global-memory k-mer set
    e5.submit([hb.RegionInput([w[:150]], "ACGT" * 5000)])
    e5.run(hb.BK_STAGE_ALL)
    want5, info5 = bo.assemble_region([w[:150]], ["ACGT" * 5000], 31, 2)
    assert sorted(e5.kmers(0)[0]) == sorted(info5["mers"]) and _strip(e5.contigs(0)) == want5
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=99)


def test_config4_config5_shapes_gpu(hb):
    """configs[3]/[4] of BASELINE.json at reduced size: mixed SV set at 1,000x (multi-contig regions, long trl
    contigs) and 250 bp reads / k=41 / 5 % noise (two-word k-mers, thousands of sample k-mers) vs the oracle."""
    from oracle import bk_oracle as bo
    regs = [synth.make_region(300 + i, sv_type=synth.SV_TYPES[i % 5], depth=1000, W=600, L=150) for i in range(5)]
    eng = _run_regions(hb, regs, 31, stages=7)
    for i, r in enumerate(regs):
        want, _ = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        assert _strip(eng.contigs(i)) == want, ("cfg4", i)
    # one region at the FULL configs[3] depth (20,000 reads: beyond the in-LDS read-grouping table -> global-memory table)
    big = synth.make_region(310, sv_type="del", depth=1000, W=3000, L=150)
    assert big.reads.shape[0] == 20000
    eng = _run_regions(hb, [big], 31, stages=7)
    want, info = bo.assemble_region(big.read_strs(), [big.window_str], 31, 2)
    assert eng.kmers(0)[0] == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)]
    assert _strip(eng.contigs(0)) == want and eng.hits(0, 0) == bo.realign(want[0]["seq"], [big.window_str])
    regs = [synth.make_region(400 + i, sv_type="del", depth=100, W=700, L=250, noise=0.05) for i in range(3)]
    eng = _run_regions(hb, regs, 41, stages=7)
    for i, r in enumerate(regs):
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 41, 2)
        assert eng.kmers(i)[0] == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)]
        assert _strip(eng.contigs(i)) == want, ("cfg5", i)


def test_large_windows_gpu(hb, oracle_bg):
    """Whole-gene windows: beyond the LDS k-mer set (40 kb, 303 kb -> bk_kmer_kernel_g) and beyond the realigner's
    staging buffer (chunked diagonals), mixed in one batch with an ordinary region and a 120 kb partner window.  (The oracle's side
    -- the realign of every contig against 300 kb targets on one core -- comes from the session's background pool.)"""
    from oracle_worker import large_window_regions
    regions = large_window_regions()
    eng = _run_regions(hb, regions, 31, stages=7)
    nrec = 0
    for i, r in enumerate(regions):
        ora = oracle_bg.get(("largewin", i))
        assert eng.kmers(i)[0] == ora["mers"], i
        assert _strip(eng.contigs(i)) == ora["contigs"], i
        for ci, c in enumerate(eng.contigs(i)):
            got = eng.hits(i, ci)
            assert got == ora["hits"][ci], (i, ci)
            nrec += len(got)
    assert nrec >= 5


def test_full_size_config2_properties_gpu(hb):
    """BASELINE.json configs[1] at full size (256 regions x 10,000 x 150 bp, k=31) through size-independent
    properties: (1) every region yields the planted call -- one contig spanning the junction, chained into one PSL
    record with a single 200 bp target gap at the planted position; (2) a second run on the same handle is identical;
    (3) results do not depend on the position of a region in the batch (reversed submission order); (4) ALL 256 regions
    equal the oracle."""
    from oracle import bk_oracle as bo
    n = 256
    regions = [synth.make_region(i) for i in range(n)]
    eng = _run_regions(hb, regions, 31, stages=7)
    first = []
    for i, r in enumerate(regions):
        cs = eng.contigs(i)
        assert len(cs) == 1, i
        hs = eng.hits(i, 0)
        assert len(hs) == 1, i
        h = hs[0]
        c = len(r.window) // 2
        assert h["t_num_insert"] == 1 and h["t_base_insert"] == 200 and h["q_num_insert"] == 0 and h["mismatches"] <= 4, (i, h)
        assert h["matches"] + h["mismatches"] == len(cs[0]["seq"]) and len(h["block_sizes"]) == 2, (i, h)
        # the gap may slide over the micro-homology (or a chance near-match) at the junction, never further
        end0 = h["t_starts"][0] + h["block_sizes"][0]
        assert abs(end0 - (c - 100)) <= 12 and h["t_starts"][1] - end0 == 200, (i, h)
        first.append((cs, hs, eng.kmers(i)[0]))
    eng.run(7)
    for i in (0, 17, 255):
        assert (eng.contigs(i), eng.hits(i, 0), eng.kmers(i)[0]) == first[i], i
    rev = _run_regions(hb, regions[::-1], 31, stages=7)
    for i in range(n):
        assert (rev.contigs(n - 1 - i), rev.hits(n - 1 - i, 0), rev.kmers(n - 1 - i)[0]) == first[i], i
    # (4) EVERY region against the oracle (contigs with both count vectors, k-mer lists, read sets; realign records): ~20 s
    # of oracle time, spread over the host cores
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_worker import oracle_regions
    ora = oracle_regions("cfg1", range(n))
    for i in range(n):
        want, hits = ora[i]
        assert _strip(first[i][0]) == want, i
        assert [first[i][1]] == hits, i


def test_noisy_regions_and_both_arenas_grow_gpu(hb):
    """Sequencing noise (1 %, hundreds of recurrent-error contigs per region) against the oracle, started with
    deliberately tiny scratch AND result arenas: both must grow (several times) and the rerun must be exact."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(901 + i, sv_type="del", depth=300, W=1000, L=150, noise=0.01) for i in range(3)]
    eng = hb.Engine(kmer_size=31, arena_bytes=1 << 16, out_kbytes=64)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    eng.run(hb.BK_STAGE_ALL)
    for i in (0, 2):
        want, info = bo.assemble_region(regions[i].read_strs(), [regions[i].window_str], 31, 2)
        assert len(want) > 100
        assert _strip(eng.contigs(i)) == want, i
        for ci in (0, len(want) // 2, len(want) - 1):
            assert eng.hits(i, ci) == bo.realign(want[ci]["seq"], [regions[i].window_str]), (i, ci)


def test_long_reads_gpu(hb):
    """Reads beyond 256 bp (the register fast paths of the k-mer kernel end there) up to ~900 bp, incl. noise and
    trimmed ends: grouping, k-mer selection, assembly (multi-tile DPs) and realignment against the oracle."""
    from oracle import bk_oracle as bo
    regs = [synth.make_region(950, sv_type="del", depth=40, W=2400, L=400),
            synth.make_region(951, sv_type="ins", depth=40, W=2400, L=400, noise=0.01, var_len=0.3),
            synth.make_region(952, sv_type="del", depth=30, W=4000, L=900, sv_size=300)]
    eng = _run_regions(hb, regs, 31, stages=7)
    for i, r in enumerate(regs):
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        assert len(want) >= 1, i
        assert eng.kmers(i)[0] == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)], i
        assert _strip(eng.contigs(i)) == want, i
        for ci, c in enumerate(want[:4]):
            assert eng.hits(i, ci) == bo.realign(c["seq"], [r.window_str]), (i, ci)


def test_fetch_then_rerun_then_call_gpu(hb):
    """bk_fetch: the records of a run are on the host, the handle runs again at once, bk_call still reports the run that
    was fetched (the bench overlaps the call tail of batch i with the kernels of batch i+1 this way)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from breakmer_amd.sv_processor import params as bk_params
    regions = [synth.make_region(40 + i, depth=80, W=1500, sv_type=synth.SV_TYPES[i % 3]) for i in range(6)]
    opts = dict(bk_params.DEFAULTS); opts["var_filter"] = ["indel", "rearrangement", "trl"]
    eng = hb.Engine(kmer_size=31)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    eng.set_call_context(bench.call_context_text(regions, opts))
    eng.run(hb.BK_STAGE_ALL)
    want = eng.call()
    assert sum(len(v) for v in want.values()) >= 4
    eng.run(hb.BK_STAGE_ALL, sync=False)
    eng.fetch()
    eng.run(hb.BK_STAGE_ALL, sync=False)               # in flight while the tail of the fetched run is computed
    assert eng.call() == want
    assert eng.call() == want                          # no snapshot held any more: refers to (and waits for) the newest run
    # the getters refer to the newest run again, which equals a fresh handle's run of the same batch; every call row
    # belongs to a contig of its region
    fresh = _run_regions(hb, regions, 31, stages=7)
    for i in range(6):
        assert eng.contigs(i) == fresh.contigs(i), i
        assert len(want.get(i, [])) <= len(eng.contigs(i)), i
        ids = {"%s_contig%d" % (regions[i].name, c + 1) for c in range(len(eng.contigs(i)))}
        assert all(row[11] in ids for row in want.get(i, [])), i


def test_native_tail_equals_python_tail_on_noisy_regions_gpu(hb, tmp_path):
    """Noisy regions: hundreds of contigs made of a few reads that share a sequencing error, each with one gap-free alignment over
    its whole length.  The native tail decides those from the raw hit (no record is built); the rows must still be the ones the
    Python tail (sv_caller.py, every contig through align_manager) makes."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from breakmer_amd import sv_processor as sp

    def inputs(d):
        bed, genes, data = [], ["header"], {}
        for i, (sv, noise) in enumerate([("del", 0.01), ("ins", 0.006), ("inv", 0.01), ("dup", 0.004), ("del", 0.0)]):
            r = synth.make_region(900 + i, sv_type=sv, depth=150, W=1500, noise=noise)
            bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
            genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
            data[r.name.upper()] = sp.RegionData(r.read_ids, r.read_strs(), r.indel_only.tolist(), None, r.window_str, [], r.disc_reads)
        d.mkdir()
        (d / "t.bed").write_text("\n".join(bed) + "\n"); (d / "g.txt").write_text("\n".join(genes) + "\n")
        return {"analysis_name": "noisy", "targets_bed_file": str(d / "t.bed"), "gene_annotation_file": str(d / "g.txt"), "kmer_size": "31", "keep_repeat_regions": True}, data
    cfg, data = inputs(tmp_path / "native")
    r1 = sp.runner(cfg, region_data=data)
    rows = r1.run()
    cfg2, data2 = inputs(tmp_path / "python")
    r2 = sp.runner(cfg2, region_data=data2, native_calls=False)
    assert r2.run() == rows and len(rows) >= 3
    assert r1.summary == r2.summary and sum(int(v.split("\t")[1]) for v in r1.summary.values()) > 300       # N_contigs: the noise made hundreds


def test_call_shortcut_equals_full_caller_gpu(hb, golden_dir):
    """bk_call decides a contig whose only alignment is ONE gap-free hit over its whole length on the target window from the raw
    hit (no records, no target_hit / get_result: bk_api.hip call_impl).  With the shortcut switched off (bk_config.flags bit
    2048) every contig goes through the full caller: the call records must be byte-identical -- on noisy regions (hundreds of
    such contigs), on clean ones of every SV type, and on the regions of the reference-made surface fixtures."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    regions = [synth.make_region(900 + i, sv_type=sv, depth=150, W=1500, noise=noise)
               for i, (sv, noise) in enumerate([("del", 0.01), ("ins", 0.006), ("inv", 0.01), ("dup", 0.004), ("del", 0.0), ("trl", 0.0), ("trl", 0.008), ("ins", 0.0)])]
    regions += [synth.make_region(**c["gen"]) for c in _load(golden_dir, "assembly.json")["cases"] if c["k"] == 31 and c["rc_thresh"] == 2]
    opts = bench.default_opts()
    blobs = []
    for flags in (0, 2048):
        eng = hb.Engine(kmer_size=31, rc_thresh=2, flags=flags)
        eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only, partners=[p[4] for p in r.partners]) for r in regions])
        eng.set_call_context(bench.call_context_text(regions, opts))
        eng.run(hb.BK_STAGE_ALL)
        blobs.append(eng.call_blob())
        ncontigs = eng.stat(6)
        eng.close()
    assert blobs[0] == blobs[1] and blobs[0].count(b"\n") >= 6 and ncontigs > 300


def test_call_async_makes_the_same_calls_gpu(hb):
    """bk_call_async: the wait for the run, the copy back and the call tail on the handle's thread; bk_call afterwards finds the
    calls made (same text as the synchronous tail); a new run invalidates them; an error of the tail is reported by the next call."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from breakmer_amd.sv_processor import params as bk_params
    regions = [synth.make_region(140 + i, depth=80, W=1500, sv_type=synth.SV_TYPES[i % 3]) for i in range(9)]
    opts = dict(bk_params.DEFAULTS); opts["var_filter"] = ["indel", "rearrangement", "trl"]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
    eng = hb.Engine(kmer_size=31)
    eng.submit(ins)
    eng.set_call_context(bench.call_context_text(regions, opts))
    eng.run(hb.BK_STAGE_ALL)
    want = eng.call_blob()
    assert want.count(b"\n") >= 6
    for _ in range(3):
        eng.run(hb.BK_STAGE_ALL, sync=False)
        eng.call_async()                                   # returns at once
        assert eng.sync() == 0                             # joins the thread
        assert eng.call_blob() == want
    # a batch of other regions on the same handle: nothing of the earlier calls is left
    eng.submit(ins[:4])
    eng.set_call_context(bench.call_context_text(regions[:4], opts))
    eng.run(hb.BK_STAGE_ALL, sync=False)
    eng.call_async()
    got = eng.call()
    assert sorted(got) == sorted(k for k in range(4) if k in got) and all(k < 4 for k in got)
    fresh = hb.Engine(kmer_size=31)
    fresh.submit(ins[:4]); fresh.set_call_context(bench.call_context_text(regions[:4], opts)); fresh.run(hb.BK_STAGE_ALL)
    assert fresh.call() == got
    # without a context the call is refused where it is made
    other = hb.Engine(kmer_size=31)
    other.submit(ins[:2]); other.run(hb.BK_STAGE_ALL)
    with pytest.raises(hb.BreakmerHipError):
        other.call_async()


def test_arena_growth_and_rerun_gpu(hb):
    """A deliberately tiny scratch arena: the library must notice the overflow, grow the arena and rerun --
    results identical to a run with the default arena; repeated bk_run on one handle is idempotent."""
    regions = [synth.make_region(500 + i, sv_type=synth.SV_TYPES[i % 4], depth=80, W=1500, noise=(0.02 if i % 2 else 0.0)) for i in range(6)]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
    big = hb.Engine(kmer_size=31)
    big.submit(ins)
    big.run(hb.BK_STAGE_ALL)
    small = hb.Engine(kmer_size=31, arena_bytes=1 << 16)
    small.submit(ins)
    small.run(hb.BK_STAGE_ALL)
    small.run(hb.BK_STAGE_ALL)
    for i in range(len(regions)):
        assert small.contigs(i) == big.contigs(i), i
        assert [small.hits(i, c) for c in range(len(small.contigs(i)))] == [big.hits(i, c) for c in range(len(big.contigs(i)))]


def test_bench_dist_path_gathers_call_records_gpu(hb, tmp_path):
    """bench.py's multi-rank code path (RCCL all_gather_into_tensor of [length | records] buffers, double buffered)
    on ONE GPU: the collated bytes of the last step equal the rank's own bk_call records, and the JSON line reports
    them; `--gpus` must equal the real world size."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "collated.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1", "--regions", "12", "--depth", "80",
           "--cpu-sample", "0", "--other-configs", "0", "--inflight", "2", "--dump-collated", dump]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    got, own = open(dump, "rb").read(), open(dump + ".rank0", "rb").read()
    assert got == own and len(got) > 0
    assert line["config"]["collated_bytes_per_step"] == len(got) and line["n_gpus"] == 1
    assert line["config"]["sv_calls_per_step"] == got.count(b"\n") == 12
    # a launcher-less `--gpus 2` on a one-GPU box must fail (no silent single-GPU run with n_gpus: 1)
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--cpu-sample", "0"], env=env2,
                        capture_output=True, text=True, timeout=300)
    assert p2.returncode != 0 and "--gpus 2" in (p2.stderr + p2.stdout)


def test_config2_per_gpu_share_through_dist_path_gpu(hb, tmp_path, oracle_bg):
    """BASELINE configs[2] is 4,096 regions over 8 GPUs: 512 full-size regions per GPU per step.  That share, through
    bench.py's multi-rank code path (--force-dist: RCCL all-gather of the step's records) on this one GPU: the collated bytes
    hold, for every one of the 512 regions, exactly the row that the CPU oracle's contigs and realign records give through
    the Python call logic (breakmer_amd.sv_caller, pinned by G5/G7/G8) -- the runner over an oracle-backed engine.  N > 1
    itself stays unmeasured until a multi-GPU node exists."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "collated.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1", "--regions", "512",
           "--cpu-sample", "0", "--other-configs", "0", "--dump-collated", dump]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    got = open(dump, "rb").read().decode()
    assert line["config"]["regions_total_per_step"] == 512 and line["config"]["sv_calls_per_step"] == 512
    rows = {}
    for ln in got.split("\n"):
        if ln:
            f = ln.split("\t")
            rows.setdefault(int(f[0]), []).append(f[2:])
    # the same 512 regions through the driver surface with the oracle as the engine and the Python call logic: eight chunks of 64 on the
    # session's background pool (tests/oracle_worker.py "c2rows")
    n = 512
    by_name = {}
    for b0 in range(0, n, 64):
        by_name.update(oracle_bg.get(("c2rows", tuple(range(b0, b0 + 64)))))
    assert len(by_name) == n and all(len(v) == 1 for v in by_name.values())
    for i in range(n):
        assert rows.get(i) == by_name["GENE%05d" % i], i


def _bench_mod():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench


def test_full_size_config3_regions_gpu(hb):
    """BASELINE.json configs[3] at FULL size (20,000 x 150 bp reads = 1,000x, W = 3,000): the mixed SV set bench.py
    times -- deletion, insertion, inversion, tandem duplication, translocation with its partner window -- k-mers,
    contigs and realign records bit-exact against the C oracle."""
    from oracle import bk_oracle as bo
    bench = _bench_mod()
    regions = [bench.cfg3_region(synth, i) for i in range(8)]
    assert [r.sv_type for r in regions] == ["del", "inv", "dup", "trl", "ins", "inv", "dup", "trl"]
    assert all(r.reads.shape == (20000, 150) for r in regions) and len(regions[3].partners) == 1
    eng = _run_regions(hb, regions, 31, stages=7)
    nrec = 0
    for i, r in enumerate(regions):
        assert eng.region_status(i) == (0, "ok")
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2, find_index=True)
        assert eng.kmers(i)[0] == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)], i
        got = eng.contigs(i)
        assert _strip(got) == want and len(want) >= 1, (i, r.sv_type)
        for ci, c in enumerate(want):
            assert eng.hits(i, ci) == bo.realign(c["seq"], targets), (i, ci, r.sv_type)
            nrec += 1
    assert nrec >= 10
    assert max(len(c["seq"]) for c in eng.contigs(3)) > 500          # the translocation contig runs far into the partner half


def test_full_size_config3_64_regions_gpu(hb):
    """64 full-size configs[3] regions (16 of each SV class, 16 translocations with their partner windows; 20,000 reads each)
    in one batch on the persistent queue: contigs and realign records of every region bit-exact against the oracle (the
    translocations cost the oracle ~10 s each: the host cores share them)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_worker import oracle_regions
    bench = _bench_mod()
    ids = list(range(8, 72))
    regions = [bench.cfg3_region(synth, i) for i in ids]
    eng = _run_regions(hb, regions, 31, stages=7)
    assert eng.sync() == 0
    ora = oracle_regions("cfg3", ids)
    nrec = 0
    for j, i in enumerate(ids):
        want, hits = ora[i]
        assert _strip(eng.contigs(j)) == want and len(want) >= 1, (i, regions[j].sv_type)
        assert [eng.hits(j, ci) for ci in range(len(want))] == hits, (i, regions[j].sv_type)
        nrec += sum(len(h) for h in hits)
    assert nrec >= 100


def test_full_size_config4_regions_gpu(hb, oracle_bg):
    """BASELINE.json configs[4] at FULL size: 24,000 x 250 bp reads (2,000x), k = 41 (two-word keys), 5 % substitution
    noise -> millions of sample k-mers and hundreds of recurrent-error contigs per region.  Two regions, everything
    bit-exact against the C oracle (whose find_reads is answered from its index: the plain scan needs hours here; a minute per region
    on one core: the session's background pool has been at it since the session started)."""
    bench = _bench_mod()
    regions = [bench.cfg4_region(synth, i) for i in range(2)]
    assert all(r.reads.shape == (24000, 250) for r in regions)
    eng = _run_regions(hb, regions, 41, stages=7)
    for i, r in enumerate(regions):
        assert eng.region_status(i) == (0, "ok")
        ora = oracle_bg.get(("cfg4", i))
        want = ora["contigs"]
        mers, counts, U = eng.kmers(i)
        assert U == ora["U"] and len(mers) == ora["M"] > 1000000
        h = hashlib.sha256()
        for m, c in sorted(zip(mers, counts.tolist()), key=lambda x: (x[1], x[0]), reverse=True):
            h.update(("%s %d\n" % (m, c)).encode())
        assert h.hexdigest() == ora["mers_sha256"], i
        got = eng.contigs(i)
        assert len(got) == len(want) > 300, (i, len(got), len(want))
        assert _strip(got) == want, i
        for ci, rec in ora["hits"].items():
            assert eng.hits(i, ci) == rec, (i, ci)


def test_more_regions_than_workgroups_gpu(hb):
    """A batch far larger than the number of resident assembler workgroups (persistent workgroups pulling regions from
    the cost-ordered queue): every region's result equals what a small batch gives, light and heavy regions mixed."""
    from oracle import bk_oracle as bo
    n = 1500
    regions = [synth.make_region(5000 + i, sv_type=synth.SV_TYPES[i % 5], depth=(200 if i % 97 == 0 else 24), W=700, L=100,
                                 noise=(0.01 if i % 50 == 7 else 0.0)) for i in range(n)]
    big = _run_regions(hb, regions, 25, stages=7)
    assert big.stat(22) == 0
    # repeated runs of the same batch are identical, region by region (region -> workgroup assignment and the relative
    # timing of the wavefronts differ from run to run: this is the test that exposed a missing barrier in bk_retire)
    snap = [big.contigs(i) for i in range(n)]
    for _rep in range(3):
        big.run(7)
        assert [i for i in range(n) if big.contigs(i) != snap[i]] == []
    picks = list(range(0, n, 97)) + [7, 57, 1, 2, 3, 4, n - 1]
    small = _run_regions(hb, [regions[i] for i in picks], 25, stages=7)
    for j, i in enumerate(picks):
        assert big.contigs(i) == small.contigs(j), i
        assert [big.hits(i, c) for c in range(len(big.contigs(i)))] == [small.hits(j, c) for c in range(len(small.contigs(j)))], i
    for i in (0, 7, 97, 1499):
        r = regions[i]
        want, _ = bo.assemble_region(r.read_strs(), [r.window_str], 25, 2)
        assert _strip(big.contigs(i)) == want, i
    assert sum(len(big.contigs(i)) for i in range(n)) >= n // 2


def test_reads_with_n_gpu(hb):
    """Reads with N calls: grouping (the N is part of the string), k-mer counting (no k-mer spans an N), the overlap DP
    (N matches N only), contigs that contain N, realignment of such contigs -- against the oracle; the fixtures from the
    real reference with N reads are part of test_g3_assembly_golden_gpu."""
    from oracle import bk_oracle as bo
    regions = [synth.make_region(960, sv_type="del", depth=80, W=1500, n_frac=0.2),
               synth.make_region(961, sv_type="ins", depth=80, W=1500, n_frac=0.6, noise=0.01, var_len=0.3),
               synth.make_region(962, sv_type="trl", depth=60, W=1200, n_frac=0.4),
               synth.make_region(963, sv_type="del", depth=40, W=2400, L=400, n_frac=0.5),      # reads beyond the register fast paths
               synth.make_region(964, sv_type="inv", depth=300, W=900, n_frac=1.0, noise=0.01)]
    # duplicates that differ only in an N, and identical reads with the same N: grouped exactly like the strings
    r0 = regions[0]
    r0.reads[5] = r0.reads[4]; r0.reads[5, 70] = 4
    r0.reads[6] = r0.reads[5]
    eng = _run_regions(hb, regions, 31, stages=7)
    ncontig_n = 0
    for i, r in enumerate(regions):
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        mers, counts, U = eng.kmers(i)
        assert U == len(info["rep"]) == len(set(r.read_strs())), i
        assert dict(zip(mers, counts.tolist())) == dict(zip(info["mers"], info["counts"].tolist())), i
        assert _strip(eng.contigs(i)) == want and len(want) >= 1, i
        for ci, c in enumerate(want[:6]):
            assert eng.hits(i, ci) == bo.realign(c["seq"], targets), (i, ci)
            ncontig_n += "N" in c["seq"]
    assert ncontig_n >= 1


def test_windows_with_n_gpu(hb):
    """Reference windows (and a partner window) with N calls -- an assembly gap within reach of the target: no window k-mer spans an
    N (Jellyfish skips them, utils.py:151-178), so the reads over the gap are sample-only there and get assembled; in the
    realignment an N matches nothing.  K-mer sets, contigs and realign records against the oracle, for LDS-sized windows, a
    whole-gene window (global-memory k-mer set, chunked realign), N next to the SV junction, runs of N, N at the window ends."""
    import numpy as np
    from oracle import bk_oracle as bo
    regions = []
    spots = [[700], [700, 701, 702, 703], [0, 1499], [640, 660], [300, 900, 1200], [745], [100 + 7 * j for j in range(40)]]
    for i, sp_ in enumerate(spots):
        r = synth.make_region(8100 + i, sv_type=synth.SV_TYPES[i % 5], depth=60, W=1500, noise=(0.01 if i == 4 else 0.0), n_frac=(0.2 if i == 3 else 0.0))
        r.window = r.window.copy()
        r.window[np.array(sp_)] = 4
        if r.partners:
            pw = r.partners[0][4].copy(); pw[[760, 761, 1100]] = 4
            r.partners[0] = r.partners[0][:4] + (pw,)
        regions.append(r)
    r = synth.make_region(8120, sv_type="del", depth=60, W=1500)             # whole-gene sized window with N
    fl = synth.rand_bases(synth.stream_key(5, 8120, 9), 2 * 15000)
    r.window = np.concatenate([fl[:15000], r.window, fl[15000:]]).astype(np.uint8)
    r.window[[15700, 15701, 200, 29000]] = 4
    regions.append(r)
    eng = _run_regions(hb, regions, 31, stages=7)
    assert eng.sync() == 0
    ncontig = 0
    for i, r in enumerate(regions):
        assert "N" in r.window_str
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        assert eng.kmers(i)[0] == [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)], i
        got = eng.contigs(i)
        assert _strip(got) == want, i
        for ci, c in enumerate(want):
            assert eng.hits(i, ci) == bo.realign(c["seq"], targets), (i, ci)
        ncontig += len(want)
    assert ncontig >= len(regions) + 3                        # the gaps produce contigs of their own


def test_translocation_partner_discovery_gpu(hb, tmp_path):
    """N4 on the GPU: reads from a SAM file, target window cut from the genome FASTA, partner window discovered from the
    discordant pairs, both call tails."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_pipeline import check_trl_run, make_trl_inputs
    from breakmer_amd import sv_processor as sp
    cfg, r = make_trl_inputs(tmp_path)
    run = sp.runner(cfg)
    rows = run.run()
    check_trl_run(run, rows, r, tmp_path)
    (tmp_path / "py").mkdir()
    cfg2, r2 = make_trl_inputs(tmp_path / "py")
    assert sp.runner(cfg2, native_calls=False).run() == rows


def test_batch_lane_of_the_driver_gpu(hb, tmp_path):
    """runner.run over 160 targets with packed reads (every SV type, a third with noise, one without reads; three translocation
    targets with a partner window, which are not plain: their batch goes per target) in batches of 48 on several handles: the batch
    lane (no target objects, the batch as a table) gives the rows, summary lines and target objects of the per-target way
    (batch_lane=False), and the rows of the per-target way over the C oracle with the Python call tail."""
    import numpy as np
    from breakmer_amd import sv_processor as sp
    from fake_engine import FakeEngine
    n = 160
    regions = [synth.make_region(900 + i, depth=(60, 100)[i % 2], W=1500, sv_type="trl" if i in (100, 110, 120) else synth.SV_TYPES[i % 4], noise=(0.0, 0.0, 0.004)[i % 3]) for i in range(n)]
    bed, genes = [], ["header"]
    for r in regions:
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
        genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
        for p_ in r.partners:
            genes.append("\t".join(["0", p_[3], "chr" + p_[0], "+", str(p_[1]), str(p_[2])] + ["x"] * 6 + [p_[3]]))
    (tmp_path / "t.bed").write_text("\n".join(bed) + "\n")
    (tmp_path / "g.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "lane", "targets_bed_file": str(tmp_path / "t.bed"), "gene_annotation_file": str(tmp_path / "g.txt"), "kmer_size": "31",
           "keep_repeat_regions": True, "batch_regions": 48}

    def inputs(packed):
        data = {}
        for i, r in enumerate(regions):
            reads, lens, ids = (r.reads, r.read_lens, r.read_ids) if i != 17 else (r.reads[:0], r.read_lens[:0], r.read_ids[:0])
            pk = None if not packed else hb.pack_reads(reads, lens) if len(lens) else hb.PackedReads(np.zeros((0, 10), np.uint32), lens, None)
            data[r.name.upper()] = sp.RegionData(ids, None, None, None, r.window_str, [(p_[0], p_[1], p_[2], p_[3], synth.codes_to_str(p_[4])) for p_ in r.partners],
                                                 r.disc_reads, read_codes=reads, read_lens=lens, read_packed=pk)
        return data
    out = {}
    for way in (True, False):
        run = sp.runner(cfg, region_data=inputs(True), batch_lane=way)
        rows = run.run()
        assert set(run.targets._made) == {r.name.upper() for r in (regions[96:144] if way else regions)}      # objects only where a batch went per target
        objs = {k: (t.name, t.chrom, t.start, t.end, t.results, len(t.kmers.get('clusters', [])), t.svs, t.failed) for k, t in run.targets.items()}
        out[way] = (rows, run.summary, run.summary_header, run.failed_targets, objs)
    assert out[True] == out[False]
    rows = out[True][0]
    assert len(rows) >= n - 10 and not out[True][3]
    want = sp.runner(cfg, region_data=inputs(False), engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')), native_calls=False).run()
    assert [[str(x) for x in w] for w in want] == rows


def test_translocation_without_discordant_pairs_genome_search_gpu(hb, tmp_path, monkeypatch):
    """N4, the genome-wide second pass as a GPU path: a translocation that only split reads speak of (every pair with its ends on two
    chromosomes removed from the alignment file) -- the first pass leaves the partner half of the contig unaligned, the driver looks
    that segment up in the genome index ON THE DEVICE (bk_index_find: ranges, hits sorted by (sequence, diagonal, position), loci with
    >= 2 hits in a diagonal band), runs the target again with the window found, and the contig is explained.  The same run with the
    look-ups on the host (numpy) must find the same window and give the same rows.  (A stand-in for the reference's whole-genome
    gfServer, sv_processor.py:829-831: unpinned)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_pipeline import make_trl_inputs
    from breakmer_amd import refseq, sv_processor as sp

    def run_once(d, on_device):
        d.mkdir()
        cfg, r = make_trl_inputs(d)
        sam = (d / "sample.sam").read_text().splitlines()
        kept = [ln for ln in sam if ln.startswith("@") or ln.split("\t")[6] == "="]
        assert 0 < len(kept) < len(sam)
        (d / "sample.sam").write_text("\n".join(kept) + "\n")
        made = []
        orig = refseq.GenomeIndex.__init__

        def init(self, fasta, *a, **kw):
            if not on_device:
                kw["device"] = None
            kw["cache"] = False
            orig(self, fasta, *a, **kw)
            made.append(self)
        monkeypatch.setattr(refseq.GenomeIndex, "__init__", init)
        run = sp.runner(cfg)
        rows = run.run()
        monkeypatch.setattr(refseq.GenomeIndex, "__init__", orig)
        t = run.targets[r.name.upper()]
        assert t.genome_searched and len(made) == 1 and (made[0]._dev_no is not None) == on_device
        assert made[0]._dev_loci == on_device and (made[0].probe_ms > 0) == on_device          # the loci came from bk_index_find / from numpy
        return rows, [(w[0], w[1], w[2], w[3]) for w in t.partner_windows], t
    rows_d, wins_d, t = run_once(tmp_path / "dev", True)
    rows_h, wins_h, _t = run_once(tmp_path / "host", False)
    assert wins_d == wins_h and len(wins_d) == 1 and rows_d == rows_h
    pc, ps, pe, pn = wins_d[0]
    assert pc == "2" and pn == "PARTNERX" and ps < 5000 + 600 < pe


@pytest.mark.gpu
def test_async_submit_gpu(hb):
    """BK_SUBMIT_ASYNC (the driver's overlap of packing + H2D with the previous batch): same results as the blocking submit;
    an input error of the asynchronous submit is raised by the next call on the handle, which then takes a new batch."""
    regions = [synth.make_region(7100 + i, sv_type=synth.SV_TYPES[i % 5], depth=40, W=700, L=100) for i in range(24)]
    ref = _run_regions(hb, regions, 25, stages=7)

    def mk(rs):
        return [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only, partners=[p[4] for p in r.partners]) for r in rs]
    eng = hb.Engine(kmer_size=25)
    for _rep in range(3):                                       # resubmission on the same handle while nothing else is pending
        eng.submit(mk(regions), wait=False)
        eng.run(7)
        for i in range(len(regions)):
            assert eng.contigs(i) == ref.contigs(i), i
            assert [eng.hits(i, c) for c in range(len(eng.contigs(i)))] == [ref.hits(i, c) for c in range(len(ref.contigs(i)))], i
    bad = regions[3].read_strs()
    bad[5] = bad[5][:10] + "x" + bad[5][11:]
    ins = mk(regions[:3]) + [hb.RegionInput(bad, regions[3].window_str)]
    eng.submit(ins, wait=False)
    with pytest.raises(hb.BreakmerHipError, match="region 3 read 5"):
        eng.run(7)
    with pytest.raises(hb.BreakmerHipError):                    # no batch on the handle after a failed submit
        eng.run(7)
    eng.submit(mk(regions), wait=False)
    eng.submit(mk(regions[:5]), wait=False)     # supersedes the unfinished one
    eng.run(7)
    assert eng.n_regions == 5 and eng.contigs(4) == ref.contigs(4)
    eng.close()


@pytest.mark.gpu
def test_lookahead_across_visits_changes_nothing_gpu(hb):
    """The look-ahead across k-mer visits and into the next seeds only moves DPs to an earlier round: contigs, realign
    records and the DP work counted (the reference's nw calls and cells) are the same with either or both switched off
    (bk_config.flags = 8 / 16 / 24), for both workgroup sizes, on clean, noisy and N-carrying regions.  The same for
    the run retire (64 = off: every read retired on its own): reads that leave the contig sequence as it is -- rejected,
    identical, contained -- are retired together; on the noisy regions whole runs are rejections (the shape that made the
    first, unguarded version of it crawl: a run without a contained read has an empty count range)."""
    regions = [synth.make_region(7300 + i, sv_type=synth.SV_TYPES[i % 5], depth=(300 if i % 3 == 0 else 60), W=1200, L=100,
                                 noise=(0.0, 0.006, 0.004, 0.03)[i % 4], n_frac=(0.2 if i % 7 == 0 else 0.0)) for i in range(40)]
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, indel_only=r.indel_only, partners=[p[4] for p in r.partners]) for r in regions]
    ref = None
    for wg in (512, 256):
        for flags in (24 | 64, 24, 8, 16, 64, 0):
            eng = hb.Engine(kmer_size=31, flags=flags, wg_threads=wg)
            eng.submit(ins)
            eng.run(7)
            assert eng.stat(22) == 0
            got = ([eng.contigs(i) for i in range(len(regions))], [[eng.hits(i, c) for c in range(eng.contig_count(i))] for i in range(len(regions))],
                   eng.stat(0), eng.stat(1))
            if ref is None:
                ref = got
            assert got[0] == ref[0], (wg, flags)
            assert got[1] == ref[1], (wg, flags)
            assert got[2:] == ref[2:], (wg, flags)
            eng.close()
    assert sum(len(c) for c in ref[0]) >= 20


@pytest.mark.gpu
def test_bucket_sort_of_seed_kmers_gpu(hb, golden_dir):
    """Regions with millions of seed-capable k-mers order them with a bucket sort (count classes, leading mer bits, one
    wavefront per bucket) instead of the in-LDS bitonic sort.  bk_config.flags = 32 forces that path on the small
    G3 fixtures: same k-mer order ((count, mer) descending) and same contigs as the reference produced; on a noisy
    region with ~10^5 seed k-mers the order is checked directly."""
    d = _load(golden_dir, "assembly.json")
    by_cfg = {}
    for c in d["cases"]:
        by_cfg.setdefault((c["k"], c["rc_thresh"]), []).append(c)
    for (k, rc), cases in by_cfg.items():
        regions = [synth.make_region(**c["gen"]) for c in cases]
        eng = _run_regions(hb, regions, k, rc, flags=32)
        for i, c in enumerate(cases):
            mers, counts, U = eng.kmers(i)
            got = dict(zip(mers, counts.tolist()))
            assert len(mers) == c["n_mers"], c["tag"]
            seeds = [(m, n) for m, n in zip(mers, counts.tolist()) if n >= 2]
            assert [m for m, _ in seeds] == [m for m, _ in sorted(seeds, key=lambda x: (x[1], x[0]), reverse=True)], c["tag"]
            assert hashlib.sha256(("\n".join("%s %d" % (m, got[m]) for m in sorted(got))).encode()).hexdigest() == c["mers_sha256"], c["tag"]
            assert _strip(eng.contigs(i)) == c["contigs"], c["tag"]
        eng.close()
    noisy = [synth.make_region(7700 + i, sv_type="del", depth=300, W=2000, L=150, noise=0.02) for i in range(2)]
    plain = _run_regions(hb, noisy, 31, stages=3)
    forced = _run_regions(hb, noisy, 31, stages=3, flags=32)
    for i in range(2):
        mers, counts, _u = forced.kmers(i)
        seeds = [(m, n) for m, n in zip(mers, counts.tolist()) if n >= 2]
        assert len(seeds) > 20000
        assert [m for m, _ in seeds] == [m for m, _ in sorted(seeds, key=lambda x: (x[1], x[0]), reverse=True)]
        assert forced.contigs(i) == plain.contigs(i)


@pytest.mark.gpu
def test_assembler_occupancy_of_the_headline_shape_gpu(hb):
    """150 bp reads with the default limits: the 256-thread assembler build must fit four workgroups per CU (40 KB of LDS
    each; look-ahead buffers once pushed it to 44 KB = three, 4 % fewer regions/s), the 512-thread build two."""
    regions = [synth.make_region(7900 + i, sv_type="del", depth=60, W=1500, L=150) for i in range(8)]
    for wg, want in ((256, 4), (512, 2)):
        eng = _run_regions(hb, regions, 31, stages=7, wg_threads=wg)
        if "check" in os.environ.get("BK_TEST_VARIANT", "") and wg == 256:
            want -= 1                                        # (the barrier-check build keeps its site table in static LDS: 3 instead of 4 per CU)
        assert eng.stat(25) == wg and eng.stat(23) == want, (wg, eng.stat(23))
        eng.close()


def test_packed_submit_gpu(hb):
    """BK_SUBMIT_PACKED: reads handed over 2 bit/base (hip_backend.pack_reads) -- ragged lengths, reads with N calls, duplicates --
    give the same k-mers, contigs and realign records as the same reads handed over as base codes or as strings; a batch is all
    packed or not at all; a broken N list is refused."""
    import numpy as np
    regions = [synth.make_region(3, depth=60, W=1500), synth.make_region(31, depth=80, W=1200, var_len=0.6, noise=0.004),
               synth.make_region(32, depth=80, W=1200, sv_type="ins", n_frac=0.15), synth.make_region(33, depth=60, W=1500, sv_type="inv", var_len=0.3, n_frac=0.05)]
    a = hb.Engine(kmer_size=31)
    a.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    a.run(hb.BK_STAGE_ALL)
    b = hb.Engine(kmer_size=31)
    b.submit([hb.RegionInput(None, r.window, packed=hb.pack_reads(r.reads, r.read_lens)) for r in regions])
    b.run(hb.BK_STAGE_ALL)
    c = hb.Engine(kmer_size=31)
    c.submit([hb.RegionInput(None, r.window, packed=hb.pack_reads(r.reads, r.read_lens)) for r in regions], wait=False)      # through the library's submit thread
    c.run(hb.BK_STAGE_ALL)
    for i in range(len(regions)):
        ka, kb = a.kmers(i), b.kmers(i)
        assert ka[0] == kb[0] and ka[1].tolist() == kb[1].tolist() and ka[2] == kb[2], i
        ca = a.contigs(i)
        assert ca and ca == b.contigs(i) == c.contigs(i), i
        for ci in range(len(ca)):
            assert a.hits(i, ci) == b.hits(i, ci), (i, ci)
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=31).submit([hb.RegionInput(regions[0].reads, regions[0].window), hb.RegionInput(None, regions[1].window, packed=hb.pack_reads(regions[1].reads, regions[1].read_lens))])
    w, l, nl = hb.pack_reads(regions[2].reads, regions[2].read_lens)
    assert len(nl) > 2
    with pytest.raises(hb.BreakmerHipError):
        hb.Engine(kmer_size=31).submit([hb.RegionInput(None, regions[2].window, packed=(w, l, nl[::-1].copy()))])        # not ascending


def test_genome_index_device_probe_gpu(hb, tmp_path):
    """N4: the genome-wide seed look-up on the device (bk_index_probe_kernel: sorted sampled 16-mers in HBM, binary search per query
    k-mer) returns the ranges numpy.searchsorted returns on the host, and refseq.GenomeIndex.find gives the same loci through
    either -- on a 4 Mb genome with an assembly gap and a repeated segment, for planted segments of both strands, absent k-mers
    and k-mers beyond the occurrence cap."""
    import numpy as np
    from breakmer_amd import refseq
    rng = np.random.default_rng(3)
    chroms = []
    for c in range(4):
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1000000, dtype=np.uint8)].copy()
        s[300000:300400] = ord("N")
        s[500000:500300] = s[100000:100300]                       # a duplicated segment: two loci
        chroms.append(s.tobytes().decode())
    fn = tmp_path / "g.fa"
    fn.write_text("".join(">chr%d\n%s\n" % (i + 1, "\n".join(s[j:j + 80] for j in range(0, len(s), 80))) for i, s in enumerate(chroms)))
    fa = refseq.FastaIndex(str(fn))
    host = refseq.GenomeIndex(fa, cache=False)
    dev = refseq.GenomeIndex(fa, cache=True, device=0)
    assert (tmp_path / "g.fa.bkidx.k16s8.npz").is_file()
    again = refseq.GenomeIndex(fa, cache=True, device=0)          # from the cache file
    assert (again.code == host.code).all() and (again.pos == host.pos).all() and (again.seqno == host.seqno).all()
    q = np.concatenate([host.code[::1000], np.array([0, 0xFFFFFFFF, 12345], dtype=np.uint32), host.code[:5] + np.uint32(1)])
    di = hb.DeviceIndex(host.code)
    lo, hi = di.probe(q)
    assert (lo == np.searchsorted(host.code, q, side="left")).all() and (hi == np.searchsorted(host.code, q, side="right")).all()
    nloc = 0
    for i in range(60):
        c = int(rng.integers(0, 4)); p = int(rng.integers(1000, 990000)) if i % 5 else 100000 + 20 * i
        seg = chroms[c][p:p + 90]
        if "N" in seg:
            continue
        seg = seg if i % 2 == 0 else refseq.revcomp(seg)
        a, b = host.find(seg), dev.find(seg)
        assert a == b, (i, a[:2], b[:2])
        nloc += len(a)
        assert a and any(x[1] == "chr%d" % (c + 1) and x[3] >= p and x[4] <= p + 90 for x in a), (i, a[:3])
    assert nloc > 60 and dev.probe_ms > 0
