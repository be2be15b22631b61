"""TEST INFRASTRUCTURE: the CPU oracle over many regions on the host cores (a spawned pool: the workers never touch the GPU).
Regions are described by the arguments of their generator, so nothing big crosses the pipe on the way in."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _one(spec):
    kind, rid, k, with_hits = spec
    from breakmer_amd import synth
    from oracle import bk_oracle as bo
    if kind == "cfg1":
        r = synth.make_region(rid)
    elif kind == "cfg3":
        import bench
        r = bench.cfg3_region(synth, rid)
    else:
        raise ValueError(kind)
    targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
    want, _ = bo.assemble_region(synth.BASES[r.reads], [r.window_str], k, 2, find_index=(kind != "cfg1"))
    hits = [bo.realign(c["seq"], targets) for c in want] if with_hits else None
    return rid, want, hits


def oracle_regions(kind, ids, k=31, with_hits=True, procs=None):
    """{region id: (contigs, realign records per contig)} from the oracle, regions in parallel"""
    import multiprocessing as mp
    procs = procs or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    specs = [(kind, i, k, with_hits) for i in ids]
    if procs == 1:
        res = [_one(s) for s in specs]
    else:
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(_one, specs, chunksize=1)
    return {rid: (want, hits) for rid, want, hits in res}


# ---- one oracle pass per session, in the background (round 6): the full-size GPU tests ask for results that a spawned pool has been
#      computing since the session started, instead of each test waiting for the CPU oracle with the GPU idle --------------------------------

def large_window_regions():
    """the regions of test_large_windows_gpu (whole-gene windows, a 120 kb partner window): built the same way by the test and by the worker"""
    import numpy as np
    from breakmer_amd import synth

    def widen(r, flank, salt):
        fl = synth.rand_bases(synth.stream_key(7, r.region_id, salt), 2 * flank)
        r.window = np.concatenate([fl[:flank], r.window, fl[flank:]]).astype(np.uint8)
        return r
    regions = [synth.make_region(600, sv_type="del", depth=60, W=3000),
               widen(synth.make_region(601, sv_type="ins", depth=60, W=3000), 18500, 0),
               widen(synth.make_region(602, sv_type="del", depth=60, W=3000, noise=0.01), 150000, 0),
               widen(synth.make_region(603, sv_type="inv", depth=60, W=3000), 40000, 0)]
    t = synth.make_region(604, sv_type="trl", depth=60, W=3000)
    pc, ps, pe, pn, pw = t.partners[0]
    fl = synth.rand_bases(synth.stream_key(7, 604, 1), 117000)
    t.partners[0] = (pc, ps, pe, pn, np.concatenate([fl[:60000], pw, fl[60000:]]).astype(np.uint8))
    regions.append(t)
    return regions


def _task(spec):
    """worker of OracleBackground: spec = (kind, argument) -> whatever the test of that kind compares with"""
    import hashlib
    kind, arg = spec
    from breakmer_amd import synth
    from oracle import bk_oracle as bo
    if kind == "cfg4":                                        # one full-size configs[4] region: contigs, k-mer digest, sampled realign records
        import bench
        r = bench.cfg4_region(synth, arg)
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 41, 2, find_index=True)
        h = hashlib.sha256()
        for m, c in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True):
            h.update(("%s %d\n" % (m, c)).encode())
        sample = sorted({0, 1, len(want) // 2, len(want) - 1}) if want else []
        return {"contigs": want, "U": len(info["rep"]), "M": len(info["mers"]), "mers_sha256": h.hexdigest(), "hits": {ci: bo.realign(want[ci]["seq"], [r.window_str]) for ci in sample}}
    if kind == "largewin":                                    # region `arg` of large_window_regions(): ordered k-mers, contigs, every realign record
        r = large_window_regions()[arg]
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        want, info = bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        mers = [m for m, _ in sorted(zip(info["mers"], info["counts"].tolist()), key=lambda x: (x[1], x[0]), reverse=True)]
        return {"contigs": want, "mers": mers, "hits": [bo.realign(c["seq"], targets) for c in want]}
    if kind == "c2rows":                                      # configs[1] regions `arg` through the driver surface with the oracle as the engine: rows by target name
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fake_engine import FakeEngine
        from breakmer_amd import sv_processor as sp
        d = tempfile.mkdtemp()
        regions = [synth.make_region(i) for i in arg]
        bed, genes, data = [], ["header"], {}
        for r in regions:
            bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
            genes.append("\t".join(["0", r.name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [r.name]))
            data[r.name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens)
        open(os.path.join(d, "t.bed"), "w").write("\n".join(bed) + "\n")
        open(os.path.join(d, "g.txt"), "w").write("\n".join(genes) + "\n")
        cfg = {"analysis_name": "c2", "targets_bed_file": os.path.join(d, "t.bed"), "gene_annotation_file": os.path.join(d, "g.txt"), "kmer_size": "31",
               "keep_repeat_regions": True, "batch_regions": 64}
        want = sp.runner(cfg, region_data=data, engine_factory=lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min'))).run()
        by_name = {}
        for w in want:
            by_name.setdefault(w[11].rsplit("_", 1)[0], []).append([str(x) for x in w])
        return {r.name: by_name.get(r.name, []) for r in regions}
    raise ValueError(kind)


class OracleBackground(object):
    """A spawned pool (the workers never touch the GPU) that starts on the oracle work of the full-size GPU tests when the session starts;
    get(spec) waits for one result.  Longest tasks first."""
    SPECS = [("cfg4", 0), ("cfg4", 1)] + [("largewin", i) for i in (2, 4, 3, 1, 0)] + [("c2rows", tuple(range(b, b + 64))) for b in range(0, 512, 64)]

    def __init__(self, procs=None):
        import multiprocessing as mp
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        self.pool = mp.get_context("spawn").Pool(procs or max(1, min(10, n - 2)))
        self.res = {spec: self.pool.apply_async(_task, (spec,)) for spec in self.SPECS}

    def get(self, spec, timeout=1200):
        if spec not in self.res:                               # (not one of the planned ones: computed now)
            self.res[spec] = self.pool.apply_async(_task, (spec,))
        return self.res[spec].get(timeout)

    def close(self):
        self.pool.terminate()
        self.pool.join()
