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
