"""Host logic (runner/target/contig mirror, config parsing, output files, multi-rank collation) on
CPU with the oracle-backed FakeEngine.  The expected rows are pinned independently: they must equal
what the REAL reference's caller produced for the same contigs in tests/golden/caller.json."""
import json
import os
import sys

import pytest

from breakmer_amd import sv_processor as sp, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_engine import FakeEngine  # noqa: E402


def make_inputs(tmp_path, ids_types, write_header=True):
    bed, genes, data = [], ["header"], {}
    for rid, sv in ids_types:
        r = synth.make_region(rid, sv_type=sv, depth=60, W=1500)
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
        genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
        for p in r.partners:
            genes.append("\t".join(["0", p[3], "chr" + p[0], "+", str(p[1]), str(p[2])] + ["x"] * 6 + [p[3]]))
        data[r.name.upper()] = sp.RegionData(r.read_ids, r.read_strs(), r.indel_only.tolist(), None, r.window_str,
                                             [(p[0], p[1], p[2], p[3], synth.codes_to_str(p[4])) for p in r.partners], r.disc_reads)
    (tmp_path / "targets.bed").write_text("\n".join(bed) + "\n")
    (tmp_path / "genes.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "synth", "targets_bed_file": str(tmp_path / "targets.bed"), "analysis_dir": str(tmp_path / "analysis"),
           "reference_data_dir": str(tmp_path / "ref"), "gene_annotation_file": str(tmp_path / "genes.txt"), "kmer_size": "31",
           "keep_repeat_regions": True}
    return cfg, data


CASES = [(3, "del"), (3, "ins"), (3, "inv"), (3, "dup"), (3, "trl")]


def test_runner_matches_reference_rows(tmp_path, golden_dir):
    gold = {c["tag"]: c["expected"] for c in json.load(open(os.path.join(golden_dir, "caller.json")))["cases"]}
    for rid, sv in CASES:
        d = tmp_path / sv
        d.mkdir()
        cfg, data = make_inputs(d, [(rid, sv)])
        r = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
        rows = r.run()
        tag = {"del": "del_indelmode", "ins": "ins", "inv": "inv_disc", "dup": "dup", "trl": "trl"}[sv]
        want = [gold[k] for k in sorted(gold) if k.startswith(tag + "_c") and gold[k] is not None]
        assert rows == want, sv
        out = d / "analysis" / "output"
        assert (out / "synth_summary.out").is_file()
        if rows:
            kind = rows[0][6]
            lines = (out / ("synth_%s_svs.out" % kind)).read_text().splitlines()
            assert lines[0].split("\t") == sp.HEADER_FIELDS and lines[1].split("\t") == rows[0]


def test_runner_leaves_the_garbage_collector_as_it_found_it(tmp_path):
    """runner.run switches the cyclic collector off while it runs (its passes over a large heap were half of the driver's time
    in a process with torch imported) and restores the state it found -- on success and on failure"""
    import gc
    cfg, data = make_inputs(tmp_path, [(3, "del")])
    seen = []

    class Watching(FakeEngine):
        def submit(self, *a, **k):
            seen.append(gc.isenabled())
            return FakeEngine.submit(self, *a, **k)

    for was in (True, False):
        (gc.enable if was else gc.disable)()
        try:
            sp.runner(cfg, region_data=data, engine_factory=lambda prm: Watching(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
            assert gc.isenabled() == was
            bad = sp.runner(cfg, region_data=data, engine_factory=lambda prm: (_ for _ in ()).throw(RuntimeError("no device")))
            with pytest.raises(RuntimeError):
                bad.run()
            assert gc.isenabled() == was
        finally:
            gc.enable()
    assert seen and not any(seen)


def test_config_and_cli_parsing(tmp_path):
    from breakmer_amd import breakmer
    (tmp_path / "c.cfg").write_text("analysis_name=x\nkmer_size=31\n")
    args = breakmer.build_parser().parse_args(["-s", "7", "-k", str(tmp_path / "c.cfg")])
    cfgfn = args.config
    del args.config
    d = breakmer.parse_config_f(cfgfn, args)
    assert d["kmer_size"] == "31" and d["indel_size"] == 7 and d["keep_intron_vars"] is True and d["trl_sr_thresh"] == 2
    (tmp_path / "bad.cfg").write_text("no_equals_sign\n")
    with pytest.raises(SystemExit):
        breakmer.parse_config_f(str(tmp_path / "bad.cfg"), args)


SKEW = [(3, "del"), (5, "ins"), (7, "inv"), (9, "del"), (11, "del"), (13, "ins")]


def _skew_inputs(d):
    """six targets, the first with ten times the reads of the others (depth 600 vs 60)"""
    cfg, data = make_inputs(d, SKEW)
    r = synth.make_region(3, sv_type="del", depth=600, W=1500)
    data[r.name.upper()] = sp.RegionData(r.read_ids, r.read_strs(), r.indel_only.tolist(), None, r.window_str, [], r.disc_reads)
    return cfg, data


def _rank_main(rank, world, port, base, q, mode):
    import torch.distributed as td
    from breakmer_amd.collate import collate_results, exchange_status
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    import pathlib
    d = pathlib.Path(base) / ("rank%d" % rank)
    d.mkdir()
    if mode in ("w8", "w8fail"):                          # 4,096 plain targets of mixed cost on eight ranks (configs[2]: 4,096 regions sharded region-per-GPU over 8)
        cfg, data, script = _many_inputs(d)

        class FailingScripted(ScriptedEngine):
            def run(self, stages, sync=True):
                raise RuntimeError("injected device failure")
        cls = FailingScripted if (mode == "w8fail" and rank == 5) else ScriptedEngine
        r = sp.runner(cfg, region_data=data, engine_factory=lambda prm: cls(script, []), rank=rank, world=world, collate=collate_results, status_exchange=exchange_status)
        try:
            rows = r.run()
            import hashlib
            q.put((rank, "ok", hashlib.sha256(repr(rows).encode()).hexdigest(), len(rows), (r.assigned_cost, hashlib.sha256("\n".join(r.assigned_targets).encode()).hexdigest(), len(r.assigned_targets), tuple(r.rank_loads))))
        except Exception as ex:
            q.put((rank, "raised", "%s: %s" % (type(ex).__name__, ex), None, getattr(r, "assigned_cost", None)))
        td.destroy_process_group()
        return
    if mode == "lane":                                   # plain targets with packed reads through a scripted engine: the batch lane on every rank
        cfg, data, script = _lane_inputs(d)
        r = sp.runner(cfg, region_data=data, engine_factory=lambda prm: ScriptedEngine(script, []), rank=rank, world=world, collate=collate_results, status_exchange=exchange_status)
        rows = r.run()
        q.put((rank, "ok" if not r.targets._made else "objects were made", rows, sorted(r.summary.items()), r.assigned_cost))
        td.destroy_process_group()
        return
    cfg, data = _skew_inputs(d) if mode != "plain" else make_inputs(d, [(3, "del"), (5, "ins"), (7, "inv"), (9, "del")])

    class Failing(FakeEngine):
        def run(self, stages=7, sync=True):
            raise RuntimeError("injected device failure")
    def factory(prm):
        cls = Failing if (mode == "fail" and rank == 1) else FakeEngine
        return cls(prm.get_kmer_size(), prm.get_sr_thresh('min'))
    r = sp.runner(cfg, region_data=data, engine_factory=factory, rank=rank, world=world, collate=collate_results, status_exchange=exchange_status)
    try:
        rows = r.run()
        q.put((rank, "ok", rows, sorted(r.summary), r.assigned_cost))
    except Exception as ex:
        q.put((rank, "raised", "%s: %s" % (type(ex).__name__, ex), None, r.assigned_cost))
    td.destroy_process_group()


def _lane_inputs(d):
    """seven plain targets of different depth (packed reads, no files) and the script of a ScriptedEngine for them"""
    import numpy as np
    from breakmer_amd import hip_backend as hb
    regs = [synth.make_region(20 + i, sv_type=("del", "ins", "inv")[i % 3], depth=10 + 7 * i, W=600) for i in range(7)]
    bed, genes, data, script = [], ["header"], {}, {}
    for n, r in enumerate(regs):
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
        genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
        data[r.name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=hb.pack_reads(r.reads, r.read_lens))
        rows = [[r.name, "%s:%d" % (r.chrom, r.start + 100 * c), "D10", "0", "+", "0", "indel", "7", "30", "0", "40", "%s_contig%d" % (r.name, c), "ACGT"] for c in range(1, 1 + n % 3)]
        script[r.window_str.encode()] = (rows, len(rows) + 1, None)
    (d / "t.bed").write_text("\n".join(bed) + "\n")
    (d / "g.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "lane", "targets_bed_file": str(d / "t.bed"), "gene_annotation_file": str(d / "g.txt"), "kmer_size": "31", "keep_repeat_regions": True, "batch_regions": 2}
    return cfg, data, script


def _many_inputs(d, n=4096):
    """4,096 plain targets (packed reads, no files) of a configs[3]-like cost mix -- sixteen base regions with 30 to 480 reads, cycled under
    4,096 target names in a scrambled order -- and the script of a ScriptedEngine for them (rows per window)"""
    from breakmer_amd import hip_backend as hb
    regs = [synth.make_region(200 + i, sv_type=("del", "ins", "inv", "trl")[i % 4], depth=(3, 6, 12, 24, 48)[i % 5], W=500, L=50) for i in range(16)]
    packs = [hb.pack_reads(r.reads, r.read_lens) for r in regs]
    bed, genes, data, script = [], ["header"], {}, {}
    for j, r in enumerate(regs):
        script[r.window_str.encode()] = ([["T", "%s:%d" % (r.chrom, r.start + 7 * j), "D%d" % (10 + j), "0", "+", "0", "indel", "7", "30", "0", "40", "c%d" % j, "ACGT"]] * (j % 3), j % 3 + 1, None)
    for t in range(n):
        j = (t * 7 + t // 16) % 16
        r, name = regs[j], "T%04d" % t
        bed.append("\t".join([r.chrom, str(r.start + t), str(r.end + t), name, "exon"]))
        genes.append("\t".join(["0", name, "chr" + r.chrom, "+", str(r.start + t), str(r.end + t)] + ["x"] * 6 + [name]))
        data[name] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=packs[j])
    (d / "t.bed").write_text("\n".join(bed) + "\n")
    (d / "g.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "many", "targets_bed_file": str(d / "t.bed"), "gene_annotation_file": str(d / "g.txt"), "kmer_size": "31", "keep_repeat_regions": True, "batch_regions": 256}
    return cfg, data, script


def _run_ranks(tmp_path, mode, tag, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + hash(tag)) % 2000
    base = tmp_path / tag
    base.mkdir()
    procs = [ctx.Process(target=_rank_main, args=(rk, world, port, str(base), q, mode)) for rk in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert not p.is_alive()
    return sorted(got)


def test_two_rank_collation_gloo(tmp_path):
    """world_size 2 on CPU (gloo): regions dealt to the ranks, rows all-gathered == single-process run."""
    single = tmp_path / "single"
    single.mkdir()
    cfg, data = make_inputs(single, [(3, "del"), (5, "ins"), (7, "inv"), (9, "del")])
    want = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
    for rank, state, rows, names, _cost in _run_ranks(tmp_path, "plain", "plain"):
        assert state == "ok" and rows == want, rank
        assert len(names) == 4
    assert len(want) >= 3


def test_two_rank_cost_skew_and_failed_rank_gloo(tmp_path):
    """(a) A 10:1 cost skew: the targets are dealt by estimated cost (number of reads), heaviest first to the least loaded
    rank -- the heavy target sits alone on one rank, the five light ones on the other (name stripes would have put it with
    two of them) -- and the collated rows equal the single-process run.  (b) A rank whose engine fails: EVERY rank raises
    (naming the failed rank) instead of the healthy one waiting in the collation until the launcher kills it."""
    single = tmp_path / "single"
    single.mkdir()
    cfg, data = _skew_inputs(single)
    want = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
    got = _run_ranks(tmp_path, "skew", "skew")
    costs = sorted(g[4] for g in got)
    light = 5 * ((60 * 1500) // 150)
    assert costs == [light, (600 * 1500) // 150], costs
    for rank, state, rows, names, _cost in got:
        assert state == "ok" and rows == want, rank
        assert len(names) == len(SKEW)
    got = _run_ranks(tmp_path, "fail", "fail")
    assert [g[1] for g in got] == ["raised", "raised"], got
    assert "injected device failure" in got[1][2]                     # the failing rank re-raises its own exception
    assert "rank 1" in got[0][2] and "injected device failure" in got[0][2]      # the healthy rank names it


def test_two_rank_batch_lane_gloo(tmp_path):
    """the batch lane under two ranks (gloo): targets dealt by cost without a target object being made, rows and summary lines
    collated == the single-process run"""
    single = tmp_path / "single"
    single.mkdir()
    cfg, data, script = _lane_inputs(single)
    one = sp.runner(cfg, region_data=data, engine_factory=lambda prm: ScriptedEngine(script, []))
    want = one.run()
    assert len(want) == 6 and not one.targets._made
    got = _run_ranks(tmp_path, "lane", "lane")
    for rank, state, rows, summary, _cost in got:
        assert state == "ok" and rows == want and summary == sorted(one.summary.items()), rank
    assert sorted(g[4] for g in got) != [0, 0] and sum(g[4] for g in got) == sum(len(d_.read_ids) for d_ in data.values())


def test_run_depth_keeps_rows_and_order(tmp_path):
    """`run_depth` (launched batches in flight before the oldest is picked up: what a sample of noisy reads wants) changes when a
    batch is picked up, not what comes out: rows, summary and the per-target files' order are those of the default depth -- through the batch
    lane (scripted engine, batches of 2) and through the per-target way (oracle-backed engine, batches of 1 and 2)."""
    for depth in (1, 2, 4):
        d = tmp_path / ("lane%d" % depth)
        d.mkdir()
        cfg, data, script = _lane_inputs(d)
        cfg["run_depth"] = depth
        engines = []

        def factory(prm, script=script, engines=engines):
            e = ScriptedEngine(script, [])
            engines.append(e)
            return e
        r = sp.runner(cfg, region_data=data, engine_factory=factory)
        rows = r.run()
        if depth == 1:
            want, want_summary = rows, sorted(r.summary.items())
        assert rows == want and sorted(r.summary.items()) == want_summary, depth
        assert len(engines) <= depth + 3, (depth, len(engines))          # a handle per batch in flight (launched + submitted), reused afterwards
    ref = None
    for depth, bsz in ((1, 1), (3, 1), (2, 2)):
        d = tmp_path / ("obj%d_%d" % (depth, bsz))
        d.mkdir()
        cfg, data = make_inputs(d, [(3, "del"), (5, "ins"), (7, "inv"), (9, "del"), (11, "del")])
        cfg["run_depth"], cfg["batch_regions"] = depth, bsz
        rows = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
        ref = rows if ref is None else ref
        assert rows == ref and len(rows) >= 4, (depth, bsz)


def test_eight_ranks_4096_targets_gloo(tmp_path):
    """configs[2]'s shape on CPU: 4,096 targets of mixed cost dealt over EIGHT ranks (gloo), batches of 256 through the batch lane.
    Every rank computes the same assignment without communication (each reports the loads of all eight: identical tuples; the eight
    target lists are disjoint and cover the 4,096); the load imbalance by cost is below 5 %; every rank ends with the rows of the
    single-process run; and with one failed rank (rank 5's engine raises) ALL EIGHT raise, the healthy ones naming it, instead of
    waiting in the collation.  No 8-GPU node was available to any round: on hardware this path is unmeasured."""
    import hashlib
    single = tmp_path / "single"
    single.mkdir()
    cfg, data, script = _many_inputs(single)
    one = sp.runner(cfg, region_data=data, engine_factory=lambda prm: ScriptedEngine(script, []))
    want = one.run()
    assert len(want) == sum(len(script[d_.window.encode() if isinstance(d_.window, str) else d_.window][0]) for d_ in data.values()) and len(want) > 3000
    got = _run_ranks(tmp_path, "w8", "w8", world=8)
    assert [g[1] for g in got] == ["ok"] * 8, got
    for rank, _state, digest, nrows, _info in got:
        assert nrows == len(want) and digest == hashlib.sha256(repr(want).encode()).hexdigest(), rank
    loads = {g[4][3] for g in got}
    assert len(loads) == 1                                              # the same assignment on every rank
    loads = list(loads)[0]
    total = sum(len(d_.read_ids) for d_ in data.values())
    assert sum(loads) == total and [g[4][0] for g in got] == list(loads)
    assert max(loads) <= 1.05 * (total / 8.0), loads
    assert sum(g[4][2] for g in got) == 4096 and len({g[4][1] for g in got}) == 8      # eight different lists that add up to all targets
    got = _run_ranks(tmp_path, "w8fail", "w8fail", world=8)
    assert [g[1] for g in got] == ["raised"] * 8, got
    for rank, _state, text, _n, _c in got:
        assert "injected device failure" in text and (rank == 5 or "rank 5" in text), (rank, text)


def make_sam_inputs(tmp_path, rid=3, sv="del", size=120, n_pairs=400):
    """Config whose reads come from an alignment file (N2): SAM text + the reference-window FASTA."""
    r = synth.make_region(rid, W=1200, L=100, depth=5, sv_type=sv, sv_size=size)
    (tmp_path / "targets.bed").write_text("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]) + "\n")
    (tmp_path / "genes.txt").write_text("header\n" + "\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]) + "\n")
    (tmp_path / "sample.sam").write_text(synth.make_sam(r, n_pairs))
    ref = tmp_path / "ref" / r.name
    ref.mkdir(parents=True)
    (ref / (r.name + "_forward_refseq.fa")).write_text(">w\n" + r.window_str + "\n")
    cfg = {"analysis_name": "fromsam", "targets_bed_file": str(tmp_path / "targets.bed"), "analysis_dir": str(tmp_path / "analysis"),
           "reference_data_dir": str(tmp_path / "ref"), "gene_annotation_file": str(tmp_path / "genes.txt"), "kmer_size": "15",
           "keep_repeat_regions": True, "sample_bam_file": str(tmp_path / "sample.sam")}
    return cfg, r


def check_sam_run(rows, r, tmp_path):
    assert len(rows) >= 1
    c = len(r.window) // 2
    gpos = r.start - 200
    bps = rows[0][1]
    assert rows[0][6] == "indel" and rows[0][0] == r.name
    lo = gpos + c - r.sv_size // 2
    assert bps == "chr%s:%d-%d (D%d)" % (r.chrom, lo + 1, lo + 1 + r.sv_size, r.sv_size), bps
    cov = [int(x) for x in rows[0][10].split(",")]
    assert len(cov) == 2 and max(cov) > 0, rows[0][10]                # breakpoint coverages from the alignment file
    d = tmp_path / "analysis" / "targets" / r.name / "data"
    assert (d / (r.name + "_sv_reads.fastq")).stat().st_size > 0 and (d / (r.name + "_sv_sc_seqs.fa")).stat().st_size > 0


def test_runner_from_alignment_file(tmp_path):
    cfg, r = make_sam_inputs(tmp_path)
    run = sp.runner(cfg, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows = run.run()
    check_sam_run(rows, r, tmp_path)


def test_reads_with_n_are_kept(tmp_path):
    """Real alignment files contain N calls; the reference keeps such reads (utils.py:203-246) and so does this path:
    every N read reaches the engine, the run finds the planted call; only characters that are not A/C/G/T/N are dropped."""
    cfg, r = make_sam_inputs(tmp_path)
    sam = (tmp_path / "sample.sam").read_text().splitlines()
    out, n = [], 0
    for ln in sam:
        f = ln.split("\t")
        if not ln.startswith("@") and n < 12 and "S" in f[5]:
            f[9] = f[9][:40] + ("N" if n < 10 else "R") + f[9][41:]
            n += 1
        out.append("\t".join(f))
    (tmp_path / "sample.sam").write_text("\n".join(out) + "\n")
    seen = {}

    class Spy(FakeEngine):
        def submit(self, ins):
            seen["n"] = sum(1 for g in ins for i in range(g.reads.shape[0]) if b"N" in bytes(g.reads[i, :g.lens[i]]))
            seen["other"] = sum(1 for g in ins for i in range(g.reads.shape[0]) if b"R" in bytes(g.reads[i, :g.lens[i]]))
            FakeEngine.submit(self, ins)
    rows = sp.runner(cfg, engine_factory=lambda prm: Spy(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
    assert n == 12 and len(rows) >= 1 and rows[0][1].endswith("(D120)")
    assert seen["n"] >= 8 and seen["other"] == 0


def test_g9_annotation_and_repeat_mask_loaders(tmp_path, golden_dir):
    """N3: gene table / extra regions / repeat-mask loaders == what the REAL reference's loaders produced from the same
    files (tests/golden/loaders.json), incl. the widest-record rule for repeated gene ids, set_gene, the per-target mask
    with its chromosome-name quirk (a target given as 'chr1' matches no repeat), the bed file it writes and its read-back."""
    g = json.load(open(os.path.join(golden_dir, "loaders.json")))
    (tmp_path / "genes.txt").write_text(g["gene_table"])
    (tmp_path / "other.bed").write_text(g["regions_bed"])
    (tmp_path / "rmask.bed").write_text(g["repeat_mask_bed"])
    an = sp.anno()
    an.add_genes(str(tmp_path / "genes.txt"))
    assert {k: list(v) for k, v in an.genes.items()} == g["genes_after_add_genes"]
    an.add_regions(str(tmp_path / "other.bed"))
    assert {k: list(v) for k, v in an.genes.items()} == g["genes_after_add_regions"] and list(an.genes) == g["gene_order"]
    for chrom, pos, want in g["set_gene"]:
        assert an.set_gene(chrom, pos) == want, (chrom, pos)
    allm = sp.setup_rmask_all(str(tmp_path / "rmask.bed"))
    assert {k: [list(x) for x in v] for k, v in allm.items()} == g["rmask_all"]
    for t in g["rmask_targets"]:
        chrom, s, e, name = t["coords"]
        ref = tmp_path / ("ref_" + name)
        first = sp.setup_rmask((chrom, s, e, name, []), str(ref), allm)
        assert [list(x) for x in first] == t["first"], name
        assert (ref / (name + "_rep_mask.bed")).read_text() == t["bed"], name
        assert [list(x) for x in sp.setup_rmask((chrom, s, e, name, []), str(ref), allm)] == t["second"], name


def test_runner_batches_and_code_matrix_inputs(tmp_path):
    """runner.run in bounded batches on alternating handles (batch_regions) and with code-matrix inputs (no per-read host
    work, read objects made on demand) gives the rows and files of the one-batch string path."""
    ids = [(3, "del"), (5, "ins"), (7, "inv"), (9, "del"), (11, "dup"), (13, "del"), (15, "ins")]
    one = tmp_path / "one"
    one.mkdir()
    cfg, data = make_inputs(one, ids)
    fac = lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min'))
    want = sp.runner(cfg, region_data=data, engine_factory=fac).run()
    assert len(want) >= 4
    for tag, bsz, codes in (("b2", 2, False), ("b3c", 3, True)):
        d = tmp_path / tag
        d.mkdir()
        cfg2, data2 = make_inputs(d, ids)
        cfg2["batch_regions"] = bsz
        if codes:
            for (rid, sv), key in zip(ids, list(data2)):
                pass
            for rid, sv in ids:
                r = synth.make_region(rid, sv_type=sv, depth=60, W=1500)
                data2[r.name.upper()] = sp.RegionData(r.read_ids, None, r.indel_only.tolist(), None, r.window_str,
                                                      [(p[0], p[1], p[2], p[3], synth.codes_to_str(p[4])) for p in r.partners], r.disc_reads,
                                                      read_codes=r.reads, read_lens=r.read_lens)
        run = sp.runner(cfg2, region_data=data2, engine_factory=fac)
        assert run.run() == want, tag
        a = (one / "analysis" / "output" / "synth_indel_svs.out").read_text()
        assert (d / "analysis" / "output" / "synth_indel_svs.out").read_text() == a, tag
        fq1 = sorted((one / "analysis" / "targets" / "GENE00003" / "contigs" / "contig1" / "contig1.fq").read_text().split("\n"))
        assert sorted((d / "analysis" / "targets" / "GENE00003" / "contigs" / "contig1" / "contig1.fq").read_text().split("\n")) == fq1, tag


def make_trl_inputs(tmp_path):
    """Translocation from files only: alignment file (SAM), genome FASTA with the target and the partner chromosome, BED,
    gene table -- no reference-window file and no partner window are handed over (N4)."""
    import numpy as np
    r = synth.make_region(21, sv_type="trl", W=1200, L=100, depth=5)
    r.chrom, r.start = "1", 2200
    r.end = r.start + (len(r.window) - 400)
    pw = r.partners[0][4]
    r.partners[0] = ("2", 5000, 5000 + len(pw), "PARTNERX", pw)
    fl = synth.rand_bases(synth.stream_key(3, 21, 9), 12000)
    chr1 = np.concatenate([fl[:2000], r.window, fl[2000:4000]])
    chr2 = np.concatenate([fl[4000:9000], pw, fl[9000:12000]])
    with open(tmp_path / "genome.fa", "w") as f:
        for name, seq in (("chr1", chr1), ("chr2", chr2)):
            s = synth.codes_to_str(seq)
            f.write(">" + name + " test\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    (tmp_path / "targets.bed").write_text("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]) + "\n")
    genes = ["header", "\t".join(["0", r.name, "chr1", "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]),
             "\t".join(["0", "PARTNERX", "chr2", "+", "4000", "7500"] + ["x"] * 6 + ["PARTNERX"])]
    (tmp_path / "genes.txt").write_text("\n".join(genes) + "\n")
    (tmp_path / "sample.sam").write_text(synth.make_sam_trl(r, 400))
    cfg = {"analysis_name": "trlrun", "targets_bed_file": str(tmp_path / "targets.bed"), "analysis_dir": str(tmp_path / "analysis"),
           "reference_data_dir": str(tmp_path / "ref"), "gene_annotation_file": str(tmp_path / "genes.txt"), "kmer_size": "15",
           "keep_repeat_regions": True, "sample_bam_file": str(tmp_path / "sample.sam"), "reference_fasta": str(tmp_path / "genome.fa")}
    return cfg, r


def check_trl_run(run, rows, r, tmp_path):
    from breakmer_amd import refseq
    t = run.targets[r.name.upper()]
    ref = tmp_path / "ref" / r.name
    fwd = (ref / (r.name + "_forward_refseq.fa")).read_text()
    assert fwd == ">" + r.name + "\n" + r.window_str + "\n"                        # utils.extract_refseq_fa: [start-200, end+200)
    assert (ref / (r.name + "_reverse_refseq.fa")).read_text() == ">" + r.name + "\n" + refseq.revcomp(r.window_str) + "\n"
    assert len(t.partner_windows) == 1
    pc, ps, pe, pn, pseq = t.partner_windows[0]
    assert pc == "2" and pn == "PARTNERX" and ps < 5000 + 600 < pe and pseq == refseq.FastaIndex(str(tmp_path / "genome.fa")).fetch("2", ps, pe)
    assert len(rows) >= 1
    row = rows[0]
    assert row[6] == "rearrangement" and set(row[0].split(",")) == {r.name, "PARTNERX"}, row[:7]
    bps = row[1].split(",")
    assert any(b.startswith("chr1:") and abs(int(b.split(":")[1]) - (r.start - 200 + 600)) <= 12 for b in bps), row[1]
    assert any(b.startswith("chr2:") and abs(int(b.split(":")[1]) - (5000 + 600)) <= 12 for b in bps), row[1]
    assert int(row[9]) >= 2                                                         # discordant pairs counted


def test_translocation_partner_discovery_from_files(tmp_path):
    """N4: target window extracted from the genome FASTA (utils.extract_refseq_fa), partner window discovered from the
    discordant pairs of the alignment file, contig realigned against both, the reference's caller reports the translocation."""
    cfg, r = make_trl_inputs(tmp_path)
    run = sp.runner(cfg, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows = run.run()
    check_trl_run(run, rows, r, tmp_path)


def test_translocation_without_discordant_pairs_genome_search(tmp_path):
    """N4, the genome-wide part: the same translocation, but every pair of the alignment file that has its ends on two
    chromosomes is removed -- only the split (soft-clipped) reads speak of it, so no partner window comes from discordant
    pairs and the first pass leaves the partner half of the contig unaligned.  The driver then looks that segment up in a
    sampled k-mer index of the whole genome FASTA (refseq.GenomeIndex; the reference asks a whole-genome gfServer,
    sv_processor.py:829-831), finds the locus on chr2, runs the target again with that window, and the contig is explained:
    one record on the target, one on chr2 at the planted breakpoint.  (Whether the reference's filter_trl then REPORTS a
    translocation with zero discordant pairs is its own matter, sv_caller.py:383-422; `genome_search = False` switches the
    second pass off.)"""
    cfg, r = make_trl_inputs(tmp_path)
    sam = (tmp_path / "sample.sam").read_text().splitlines()
    kept = [ln for ln in sam if ln.startswith("@") or ln.split("\t")[6] == "="]
    assert 0 < len(kept) < len(sam)
    (tmp_path / "sample.sam").write_text("\n".join(kept) + "\n")
    seen = []

    class Spy(FakeEngine):
        def run(self, stages=7, sync=True):
            FakeEngine.run(self, stages, sync)
            seen.append([(len(reg[4]), [[(h["t_index"], h["q_start"], h["q_end"], h["t_start"], h["t_end"]) for h in hs] for hs in out[4]])
                         for reg, out in zip(self.regions, self.out)])
    run = sp.runner(cfg, engine_factory=lambda prm: Spy(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    run.run()
    t = run.targets[r.name.upper()]
    assert len(seen) == 2                                             # first pass without, second pass with the partner window
    assert seen[0][0][0] == 0 and seen[1][0][0] == 1
    assert len(t.partner_windows) == 1
    pc, ps, pe, pn, pseq = t.partner_windows[0]
    assert pc == "2" and pn == "PARTNERX" and ps < 5000 + 600 < pe
    first = [h for hs in seen[0][0][1] for h in hs]
    second = [h for hs in seen[1][0][1] for h in hs]
    assert all(h[0] == 0 for h in first)
    on_partner = [h for h in second if h[0] == 1]
    assert on_partner and any(abs((ps + h[3]) - (5000 + 600)) <= 12 for h in on_partner), second      # the partner half starts at the planted breakpoint
    # switched off: one pass, no partner window
    d2 = tmp_path / "off"
    d2.mkdir()
    cfg2, r2 = make_trl_inputs(d2)
    (d2 / "sample.sam").write_text("\n".join(kept) + "\n")
    cfg2["genome_search"] = False
    seen.clear()
    run2 = sp.runner(cfg2, engine_factory=lambda prm: Spy(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    run2.run()
    assert len(seen) == 1 and not run2.targets[r2.name.upper()].partner_windows


def test_lazy_views_of_engine_records():
    """hip_backend.KmerStrings / sv_assembly._KmerTuples / LazyContigs behave like the lists they stand for (the driver
    builds per-target objects only when something looks at them)."""
    from breakmer_amd import hip_backend as hb, sv_assembly as sa
    ks = hb.KmerStrings(b"ACGTTGCAAAAA", 4, 3)
    assert len(ks) == 3 and list(ks) == ["ACGT", "TGCA", "AAAA"] and ks[1] == "TGCA" and ks[-1] == "AAAA" and ks[0:2] == ["ACGT", "TGCA"]
    assert ks == ["ACGT", "TGCA", "AAAA"]
    with pytest.raises(IndexError):
        ks[3]
    kt = sa._KmerTuples(ks)
    assert len(kt) == 3 and kt[0] == ("ACGT",) and [x[0] for x in kt] == list(ks) and kt[1:] == [("TGCA",), ("AAAA",)]

    class Eng(object):
        batch_serial = 4
        calls = 0

        def contig_count(self, region):
            return 2

        def contigs(self, region, lazy_kmers=False):
            Eng.calls += 1
            return [{"seq": "ACGTACGT", "indel_only": [0] * 8, "others": [2] * 8, "kmer_locs": [0] * 8, "kmers": hb.KmerStrings(b"ACGTCGTA", 4, 2), "reads": [0, 1]}] * 2

    e = Eng()
    reads = [sa.fq_read("@a/1_0", "ACGT", "IIII", False), sa.fq_read("@b/1_0", "ACGT", "IIII", False)]
    lz = sa.LazyContigs(e, 0, reads, 4)
    assert len(lz) == 2 and Eng.calls == 0                     # the count alone touches no record
    assert lz[0].get_contig_seq() == "ACGTACGT" and Eng.calls == 1 and len(lz[0].kmers) == 2 and {r.id for r in lz[1].reads} == {"@a/1_0", "@b/1_0"}
    lz2 = sa.LazyContigs(e, 0, reads, 4)
    e.batch_serial = 5                                         # the engine took another batch before anything was read
    with pytest.raises(RuntimeError):
        lz2[0]
    lz2.detach()
    assert len(lz2) == 2


def test_window_with_n_is_processed_and_other_characters_skip_that_target_only(tmp_path):
    """A reference window with an N (an assembly gap within 200 bp of the target) goes through like any other (no window k-mer
    spans the N, so the reads over it are sample-only there: the reference behaves the same, Jellyfish skips such k-mers).  A
    window with any OTHER character cannot be packed: that target is skipped ALONE -- error in the log, listed in
    runner.failed_targets -- and the other targets of the run come out as without it."""
    cfg, data = make_inputs(tmp_path, [(3, "del"), (4, "del"), (5, "ins")])
    names = sorted(data)
    bad = data[names[1]]
    bad.window = bad.window[:100] + "R" + bad.window[101:]
    withn = data[names[0]]
    withn.window = withn.window[:60] + "N" + withn.window[61:]
    r = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows = r.run()
    assert [t.upper() for t in r.failed_targets] == [names[1]]
    assert "A/C/G/T/N" in list(r.failed_targets.values())[0]
    d2 = tmp_path / "clean"
    d2.mkdir()
    cfg2, data2 = make_inputs(d2, [(3, "del"), (5, "ins")])
    r2 = sp.runner(cfg2, region_data=data2, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows2 = r2.run()
    assert len(rows) >= 2 and not r2.failed_targets
    assert [x for x in rows if x[0] == rows2[-1][0]] == [rows2[-1]]                  # the untouched target: the same row


def test_packed_targets_are_checked_at_submit_and_give_the_same_rows(tmp_path):
    """Targets that come with packed reads (hip_backend.pack_reads) have their windows checked once per batch at submit; a window
    with a foreign character still skips that target ALONE, and the rows of the others equal those of the string inputs."""
    from breakmer_amd import hip_backend as hb
    spec = [(3, "del"), (4, "del"), (5, "ins"), (6, "del")]
    cfg, data = make_inputs(tmp_path, spec)
    r0 = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows0 = r0.run()
    d2 = tmp_path / "packed"
    d2.mkdir()
    cfg2, _ = make_inputs(d2, spec)
    data2 = {}
    for rid, sv in spec:
        r = synth.make_region(rid, sv_type=sv, depth=60, W=1500)
        pk = hb.pack_reads(r.reads, r.read_lens)
        assert type(pk) is hb.PackedReads
        data2[r.name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=pk)
        assert data2[r.name.upper()].checked_at_submit() and data2[r.name.upper()].max_read_len() == int(r.read_lens.max())
    names = sorted(data2)
    r1 = sp.runner(cfg2, region_data=data2, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows1 = r1.run()
    assert rows1 == rows0 and not r1.failed_targets and len(rows1) >= 3
    bad = data2[names[2]]
    bad.window = bad.window[:300] + "Y" + bad.window[301:]
    r2 = sp.runner(cfg2, region_data=data2, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')))
    rows2 = r2.run()
    assert [t.upper() for t in r2.failed_targets] == [names[2]] and "A/C/G/T/N" in list(r2.failed_targets.values())[0]
    assert rows2 == [x for x in rows0 if not x[11].upper().startswith(names[2])] and len(rows2) == len(rows0) - 1


def test_soft_masked_refseq_file_keeps_its_case(tmp_path):
    """A <name>_forward_refseq.fa written by the reference tool from a soft-masked genome keeps lower case
    (utils.py:366-371: str(seq)); refseq.extract_refseq_fa writes the bytes of the genome file as the reference does, and the
    window goes to the library WITH its case (BLAT -repeats=lower reports matches on lower-case bases as repMatches)."""
    from breakmer_amd import refseq
    (tmp_path / "g.fa").write_text(">chr1\nACGTacgtnnACGT\nggccAATT\n")
    fa = refseq.FastaIndex(str(tmp_path / "g.fa"))
    assert fa.fetch("1", 2, 18) == "GTACGTNNACGTGGCC" and fa.fetch("1", 2, 18, upper=False) == "GTacgtnnACGTggcc"
    fn = refseq.extract_refseq_fa(("1", 204, 212, "T1", []), str(tmp_path / "ref"), fa, "forward")
    assert open(fn).read() == ">T1\nacgtnnACGTggccAATT\n"
    assert sp.read_fasta_first(fn) == "acgtnnACGTggccAATT"
    fn = refseq.extract_refseq_fa(("1", 204, 212, "T1", []), str(tmp_path / "ref"), fa, "reverse")
    assert open(fn).read() == ">T1\nAATTggccACGTnnacgt\n"


def _frame_rank(rank, world, port, q):
    """one rank of the bench.py collation: frame this rank's steps, all_gather_into_tensor, de-frame every rank's buffer"""
    import numpy as np
    import torch
    import torch.distributed as td
    from breakmer_amd import collate
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cap = 4096
        steps = [("rank%d step%d " % (rank, s)).encode() * (s + 1 + rank) for s in range(3 + rank)]      # ragged: counts and lengths differ per rank
        steps.insert(1, b"")                                                                              # a step without a call
        hv, used = collate.frame_steps([np.frombuffer(b, dtype=np.uint8) for b in steps], cap)
        dev = torch.from_numpy(hv.copy())
        out = torch.zeros(world * cap, dtype=torch.uint8)
        td.all_gather_into_tensor(out, dev)
        got = collate.deframe_gathered(out.numpy(), world, cap)
        q.put((rank, used, got, steps))
    finally:
        td.destroy_process_group()


def test_framed_step_buffers_of_two_ranks_deframe_in_rank_order_gloo():
    """bench.py's multi-rank collation ([n_steps | (length, records)*] per rank, one all_gather_into_tensor of fixed-capacity
    buffers): two gloo ranks with different step counts and record lengths; every rank recovers every rank's steps, in rank
    order and step order, byte for byte."""
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert not p.is_alive()
    want = [res[0][3], res[1][3]]
    for rank, used, got, _mine in res:
        assert got == want, rank
        assert used == 8 + sum(8 + len(b) for b in want[rank])
    from breakmer_amd import collate
    with pytest.raises(RuntimeError):
        collate.frame_steps([b"x" * 100], 64)


class ScriptedEngine(object):
    """an engine that answers from a script {window bytes: (rows, number of contigs, why the region failed or None)} through both
    ways runner.run hands batches over (objects: submit; the batch lane: submit_packed): what the two ways make of the SAME answers
    must be the same"""
    batch_serial = 0

    def __init__(self, script, log):
        self.script, self.log, self.wins = script, log, []

    def submit(self, ins, wait=True):
        self.wins = [bytes(g.window) for g in ins]
        self.batch_serial += 1
        self.log.append(("submit", len(ins)))

    def submit_packed(self, items, wait=False):
        self.wins = [w for _p, w, _io in items]
        self.batch_serial += 1
        self.log.append(("submit", len(items)))

    def run(self, stages, sync=True):
        from breakmer_amd import hip_backend as hb
        for i, w in enumerate(self.wins):                  # as the library: the (asynchronous) submit fails, the next call on the handle says so
            if w.translate(None, b"ACGTNacgtn"):
                ex = hb.BreakmerHipError("bk_run failed (-1): the submit was refused for region %d" % i)      # (the driver branches on the CODE, not on this text)
                ex.code, ex.region = hb.BK_E_ARG, i
                raise ex
    def sync(self): pass
    def close(self): pass
    def contigs(self, r): return []
    def hits(self, r, c): return []

    def set_call_context(self, text):
        self.log.append(("ctx", [ln for ln in text.split("\n") if ln.split(" ")[0] in ("region", "iv", "inv", "td", "other", "disc", "partner", "rtags")]))

    def call(self):
        return {i: [list(r) for r in self.script[w][0]] for i, w in enumerate(self.wins) if self.script[w][0]}

    def contig_counts(self): return [self.script[w][1] for w in self.wins]
    def contig_count(self, r): return self.script[self.wins[r]][1]
    def stat(self, i): return sum(1 for w in self.wins if self.script[w][2]) if i == 22 else 0

    def region_status(self, r):
        why = self.script[self.wins[r]][2]
        return (0, "ok") if why is None else (3, why)


def test_batch_lane_equals_the_per_target_way(tmp_path):
    """runner.run takes batches of plain targets (packed reads, text window, nothing else) through as rows of a table without
    making a target object for each (the batch lane); rows, summary lines, skipped and failed targets, the call context handed to
    the library and the target objects looked at AFTERWARDS are those of the per-target way (batch_lane=False).  Covered: targets
    of several intervals, a lower-case BED name, a target without reads, without rows, with two rows, a region that failed on the
    device, a window with a foreign character (the library refuses the batch, which then goes the per-target way and skips it alone), a target with soft-clip
    sequences (not plain: its batch goes the per-target way), batches of 3."""
    import numpy as np
    from breakmer_amd import hip_backend as hb
    ids = [(3, "del"), (4, "ins"), (5, "del"), (6, "inv"), (7, "del"), (8, "dup"), (9, "del"), (10, "del"), (11, "ins"), (12, "del"), (13, "del")]
    regs = [synth.make_region(rid, sv_type=sv, depth=20, W=600) for rid, sv in ids]
    bed, genes = [], ["header"]
    for n, r in enumerate(regs):
        name = r.name.lower() if n == 2 else r.name
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), name, "exon"]))
        if n == 1:                                                       # a second interval of the same target
            bed.append("\t".join([r.chrom, str(r.end + 500), str(r.end + 900), name, "intron"]))
        genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
    (tmp_path / "t.bed").write_text("\n".join(bed) + "\n")
    (tmp_path / "g.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "lane", "targets_bed_file": str(tmp_path / "t.bed"), "gene_annotation_file": str(tmp_path / "g.txt"), "kmer_size": "31",
           "keep_repeat_regions": True, "batch_regions": 3}

    def inputs():
        data = {}
        for n, r in enumerate(regs):
            reads, lens, ids_ = (r.reads, r.read_lens, r.read_ids) if n != 4 else (r.reads[:0], r.read_lens[:0], r.read_ids[:0])      # target 4: no reads
            w = r.window_str if n != 7 else r.window_str[:50] + "R" + r.window_str[51:]                                                # target 7: a foreign character
            data[r.name.upper()] = sp.RegionData(ids_, None, None, ["ACGT" * 12] if n == 9 else None, w, [], r.disc_reads, read_codes=reads, read_lens=lens,
                                                 read_packed=hb.pack_reads(reads, lens) if len(lens) else hb.PackedReads(np.zeros((0, 10), np.uint32), lens, None))
        return data
    row = lambda r, c, tag: [r.name, "%s:%d" % (r.chrom, r.start + 100 * c), "D10", "0", "+", "0", tag, "7", "30", "0", "40", "%s_contig%d" % (r.name, c), "ACGT"]
    script = {}
    for n, r in enumerate(regs):
        rows = [] if n == 3 else [row(r, 1, "indel"), row(r, 2, "rearrangement_inversion")] if n == 5 else [row(r, 1, "trl" if n == 6 else "indel")]
        w = inputs()[r.name.upper()].window.encode()
        script[w] = ([], 0, "contig longer than max_contig_len") if n == 8 else (rows, len(rows) + (n % 2), None)
    out = {}
    for way in (False, True):
        log = []
        run = sp.runner(cfg, region_data=inputs(), engine_factory=lambda prm: ScriptedEngine(script, log), batch_lane=way)
        rows = run.run()
        # batches 1, 2 and 4 of the lane run are plain (no object made); 3 holds the bad window, 4 the soft-clip target: per target
        assert set(run.targets._made) == ({r.name.upper() for r in regs[6:]} if way else {r.name.upper() for r in regs})
        assert [n for k, n in log if k != "ctx"] == ([3, 2, 3, 2, 2] if way else [3, 2, 2, 2])
        objs = {}
        for k in run.targets:
            t = run.targets[k]
            objs[k] = (t.name, t.chrom, t.start, t.end, t.results, t.has_results(), len(t.kmers.get('clusters', [])), t.svs, t.failed, t.data is None or len(t.data.read_ids) == 0)
        out[way] = (rows, run.summary, run.summary_header, run.failed_targets, objs, [x for x in log if x[0] == "ctx"])
    assert out[True] == out[False]
    rows, summary, header, failed, objs, _ctx = out[True]
    assert len(rows) == 8 and [r[11] for r in rows] == sorted(r[11] for r in rows)
    assert set(failed) == {regs[7].name, regs[8].name} and "A/C/G/T/N" in failed[regs[7].name] and "max_contig_len" in failed[regs[8].name]
    assert regs[2].name.lower() in summary and regs[4].name not in summary and regs[8].name not in summary
    assert summary[regs[5].name].split("\t")[2:6] == ["2", "1", "1", "0"] and objs[regs[1].name][3] == regs[1].end + 900
