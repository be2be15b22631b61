#!/usr/bin/env python3
"""bench.py -- BreaKmer hot path on MI355X: target regions/s at 500x 150 bp reads, 31-mers.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  `--gpus N` with N > 1 and no launcher in the environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (before anything
touches the GPU) and exits with its code; under a launcher (RANK set) the process is one rank.
A "step" is one pass of the hot path (read grouping -> k-mer selection -> assembly -> realign on the GPU,
SV-call tail in host C++) over one batch of synthetic regions, inputs already packed and resident in HBM:
  N = 1 : BASELINE.json configs[1], 256 regions x 10,000 x 150 bp reads (500x), planted 200 bp deletion, k = 31;
  N > 1 : the SAME 256 regions per GPU per step (N = 8: configs[2]'s 4,096 regions sharded region-per-GPU = two steps of 8 x 256).
Regions are independent, so ranks get disjoint region ids (weak scaling) and the only exchange is the
all-gather of the per-region result records at the end of each step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The steps in flight live on separate HIP streams.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); once torch and RCCL have created their own streams the library's 6 streams share queues and the batches
# serialise.  Measured on one GPU through the multi-rank code path: 4 queues 121 k regions/s, 8 queues 158 k, 12 or more
# 202 k (= the single-process figure).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# integer VALU issue peak: 256 CU x 4 SIMD x 32 lanes/clk (a 64-wide wavefront instruction issues over 2 clocks) x 2.4 GHz
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--regions", type=int, default=0, help="regions per GPU per step (0: 256 = configs[1] on every GPU, whatever N: the scaling curve compares equal per-GPU batches)")
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=31)
    ap.add_argument("--cpu-sample", type=int, default=24, help="regions timed through the CPU oracle on ONE core (0 = skip the CPU baseline)")
    ap.add_argument("--inflight", type=int, default=0, help="steps in flight (independent batches on separate HIP streams); 0 = 12 for runs of 60 steps or more, 8 for shorter ones (a 20-step window cannot fill and drain a 12-deep pipeline: profiles/r06/inflight_depth.txt; 6 until round 6)")
    ap.add_argument("--max-candidates", type=int, default=0, help="diagnostic: device cap on one k-mer's candidate reads (0 = library default 2048); sizes the assembler's LDS and so its workgroups per CU")
    ap.add_argument("--max-contig", type=int, default=0, help="diagnostic: device cap on the contig length (0 = library default 4096); sizes the assembler's and the realigner's LDS")
    ap.add_argument("--wg", type=int, default=256, help="assembler workgroup size of the batches in flight (256: 4 per CU; 512: 2 per CU); the one-step-at-a-time pass always uses 512")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (testing)")
    ap.add_argument("--other-configs", type=int, default=1, help="also time configs[3]/[4] of BASELINE.json (one GPU only; 0 = skip)")
    ap.add_argument("--cfg3-regions", type=int, default=4096, help="batch size of the configs[3] side measurement (heavy regions: the chip fills at a few thousand)")
    ap.add_argument("--cfg4-regions", type=int, default=768, help="batch size of the configs[4] side measurement: a region's chain of ~19,000 dependent rounds takes ~3 s whatever runs beside it, so regions in flight ARE the throughput (256: 69 regions/s, 512: 101, 768: 121) until the scratch arena (~270 MB per region) fills the HBM (896 no longer fit)")
    ap.add_argument("--noisy-inflight", type=int, default=8, help="batches in flight of the noisy 256-region side measurement (other_configs.noise_0.5pct_256_regions.in_flight)")
    ap.add_argument("--side-configs-only", type=int, default=0, help="internal: print the side measurements (configs[3], configs[4], noisy batch) as one JSON object and exit")
    ap.add_argument("--split-experimental", type=int, default=0, help="(ignored: the component split of noisy regions is the default since round 5)")
    ap.add_argument("--flags", type=int, default=0, help="library flags (bk_config.flags: BK_CFG_*); 0 in every reported number")
    ap.add_argument("--lib", default=None, help="diagnostic: path of an alternative build of the library (A/B runs on one box)")
    ap.add_argument("--dump-collated", default=None, help="write the bytes collated in the last step to this file (testing)")
    return ap.parse_args(argv)


def self_launch(a):
    """`bench.py --gpus N` without a launcher: run the ranks as a child process group (never exec: nothing of this
    process has touched the GPU yet, and it never will)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def call_context_text(regions, opts, keep_tables=False):
    """query_region / annotation / discordant-pair context of the batch for the native call tail.  keep_tables: the handle keeps the
    gene table of its previous context (the same for every batch of a run: sv_processor.runner sends it once per handle too)"""
    from breakmer_amd import call_context as cc
    genes = {}
    for r in regions:
        genes[r.name] = ["chr" + r.chrom, r.start, r.end]
        for p in r.partners:
            genes[p[3]] = ["chr" + p[0], p[1], p[2]]
    lines = [cc.opts_line(opts)] + (["keep_tables"] if keep_tables else cc.tables_lines(genes, None))
    for i, r in enumerate(regions):
        qr = (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, "exon")])
        lines += cc.region_lines(i, qr, None, r.disc_reads, [(p[0], p[1]) for p in r.partners], r.read_ids)
    return "\n".join(lines) + "\n"


def default_opts():
    from breakmer_amd.sv_processor import params as bk_params
    opts = dict(bk_params.DEFAULTS)
    opts["var_filter"] = ["indel", "rearrangement", "trl"]
    return opts


def cfg3_region(synth, i, depth=1000):
    """BASELINE.json configs[3] (SURVEY 8d "Config 4"): SV type = region_id % 4 in {indel (del 200 / ins 60 alternating),
    inversion 200, tandem-dup 200, translocation with an explicit 3,000 bp partner window}, 1,000x (20,000 reads)."""
    kind = i % 4
    sv = ("del" if (i // 4) % 2 == 0 else "ins") if kind == 0 else ("inv", "dup", "trl")[kind - 1]
    return synth.make_region(30000 + i, sv_type=sv, depth=depth, W=3000, L=150)


def cfg4_region(synth, i, depth=2000):
    """BASELINE.json configs[4] (SURVEY 8d "Config 5"): 250 bp reads at 2,000x (24,000 reads), k = 41, 5 % substitutions."""
    return synth.make_region(40000 + i, sv_type="del", depth=depth, W=3000, L=250, noise=0.05)


def noisy_region(synth, i, depth=500, L=150, noise=0.005):
    """a configs[1]-shaped region (10,000 x 150 bp, planted 200 bp deletion) with substitution noise per base: what real reads look like"""
    return synth.make_region(50000 + i, depth=depth, L=L, sv_type="del", noise=noise)


SIDE_K = {"cfg3": 31, "cfg4": 41, "noisy": 31}


def _gen_region(spec):
    kind, i = spec
    from breakmer_amd import synth
    return cfg4_region(synth, i) if kind == "cfg4" else noisy_region(synth, i) if kind == "noisy" else cfg3_region(synth, i)


def _cpu_side_region(spec):
    """one region of a side configuration through the C oracle (worker of cpu_side_baselines): contigs, seconds of THIS region"""
    kind, i = spec
    from breakmer_amd import synth
    from oracle import bk_oracle as bo
    if i < 0:                                               # start-up (imports, library load) outside the clock
        r = synth.make_region(3, sv_type="del", depth=40, W=1200)
        bo.assemble_region(r.read_strs(), [r.window_str], 31, 2)
        return 0, 0.0
    r = _gen_region((kind, i))
    targets = [r.window_str] + [synth.codes_to_str(p_[4]) for p_ in r.partners]
    t0 = time.perf_counter()
    want, _ = bo.assemble_region(synth.BASES[r.reads], [r.window_str], SIDE_K[kind], 2, find_index=True)
    for c in want:
        bo.realign(c["seq"], targets)
    return len(want), time.perf_counter() - t0


def cpu_side_baselines(samples):
    """CPU figure beside every side configuration (a GPU figure with nothing beside it cannot be read): the C oracle (oracle/bk_oracle.c,
    kind "port") over a bounded sample of the SAME regions on all host cores -- one spawned pool, a region per task; `value` = regions /
    wall time of the pool (cores stated), `value_1core` = 1 / mean seconds of one region inside its worker.  Runs BEFORE this process
    touches the GPU.  samples: {key: (kind, number of regions)}."""
    import multiprocessing as mp
    cores = usable_cores()
    out = {}
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_cpu_side_region, [("noisy", -1)] * cores, chunksize=1)
        for key, (kind, n) in samples.items():
            t0 = time.perf_counter()
            res = pool.map(_cpu_side_region, [(kind, i) for i in range(n)], chunksize=1)
            wall = time.perf_counter() - t0
            per = [x[1] for x in res]
            out[key] = {"value": round(n / wall, 4), "unit": "regions/s", "cores": min(cores, n), "kind": "port", "value_1core": round(len(per) / sum(per), 4),
                        "sample": "oracle/bk_oracle.c (T1+K1/K2+init_assembly with its k-mer -> reads index for find_reads+realign) on regions 0..%d of this configuration, one region per task over %d worker processes (%.1f s wall); value_1core from the seconds each region took inside its worker" % (n - 1, min(cores, n), wall),
                        "contigs_in_sample": int(sum(x[0] for x in res))}
    return out


def make_regions_parallel(kind, n):
    """synthetic regions generated on the host cores (a configs[4] region is 6 M random draws: ~0.1-0.3 s in numpy); the workers
    are SPAWNED and only run numpy -- nothing of this process's GPU state is inherited"""
    import multiprocessing as mp
    cores = usable_cores()
    if cores <= 1 or n < 16:
        return [_gen_region((kind, i)) for i in range(n)]
    with mp.get_context("spawn").Pool(min(cores, 16)) as pool:
        return pool.map(_gen_region, [(kind, i) for i in range(n)], chunksize=4)


def time_inflight(hb, regions, k, opts, device, handles, steps, flags, wg):
    """sustained rate of one configuration with `handles` batches in flight (one handle = one HIP stream each, the way the headline is
    timed): a single noisy batch is bound by the serial chain of its slowest region (round 6: one region of 256 takes 224 of the
    batch's 247 ms), batches in flight fill the chip meanwhile -- and then one workgroup per region (BK_CFG_NO_SPLIT) on 256-thread
    workgroups is the cheaper way through the same work than 16 units per region"""
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regions]
    ctx = call_context_text(regions, opts)
    engs, ref = [], None
    for _ in range(handles):
        e = hb.Engine(kmer_size=k, rc_thresh=2, device=device, flags=flags, wg_threads=wg)
        e.submit(ins)
        e.set_call_context(ctx)
        e.run(hb.BK_STAGE_ALL)
        raw = e.call_blob()
        ref = raw if ref is None else ref
        engs.append(e)

    def run_steps(kk):
        launched, same = 0, True
        for j in range(min(handles, kk)):
            engs[j].run(hb.BK_STAGE_ALL, sync=False)
            launched += 1
        for s_ in range(kk):
            e = engs[s_ % handles]
            e.fetch()
            if launched < kk:
                e.run(hb.BK_STAGE_ALL, sync=False)
                launched += 1
            same = (e.call_blob() == ref) and same
        return same
    run_steps(handles)
    t0 = time.perf_counter()
    same = run_steps(steps)
    dt = time.perf_counter() - t0
    out = {"value": round(len(regions) * steps / dt, 1), "unit": "regions/s", "ms_per_batch": round(dt / steps * 1e3, 2), "batches_in_flight": handles, "steps": steps,
           "flags": flags, "asm_workgroup_threads": int(engs[0].stat(25)), "failed_regions": int(sum(e.stat(22) for e in engs)), "sv_calls": int(ref.count(b"\n")),
           "rows_identical_across_batches": bool(same)}
    for e in engs:
        e.close()
    return out


def side_configs(a, hb, synth, opts, local):
    """BASELINE configs[3] / configs[4] and the noisy batches on one GPU (whole path incl. call tail, inputs resident)"""
    oc = {}
    try:
        regs3 = [cfg3_region(synth, i) for i in range(a.cfg3_regions)]
        oc["configs[3]"] = time_other_config(hb, regs3, 31, opts, 2, local, flags=a.flags)
        oc["configs[3]"]["workload"] = "mixed SV set (indel/inv/dup/trl + partner window), 1,000x 150 bp, k=31; value = ONE batch of %d regions at a time; in_flight = two batches of %d on two handles (the k-mer stage, the copy back and the call tail of one overlap the assembler of the other)" % (len(regs3), len(regs3) // 2)
        if len(regs3) >= 512:
            oc["configs[3]"]["in_flight"] = time_inflight(hb, regs3[:len(regs3) // 2], 31, opts, local, handles=2, steps=6, flags=a.flags, wg=0)
        del regs3
        regs4 = make_regions_parallel("cfg4", a.cfg4_regions)
        oc["configs[4]"] = time_other_config(hb, regs4, 41, opts, 1, local, flags=a.flags)
        oc["configs[4]"]["workload"] = "250 bp reads at 2,000x, k=41, 5 % substitution noise"
        del regs4
        # the standing round-1 bar: one launch of 64 configs[1]-shaped regions at 0.5 % substitution noise (< 0.1 s asked for)
        regsn = [synth.make_region(50000 + i, depth=a.depth, L=a.read_len, sv_type="del", noise=0.005) for i in range(64)]
        oc["noise_0.5pct_64_regions"] = time_other_config(hb, regsn, a.kmer, opts, 2, local, flags=a.flags)
        oc["noise_0.5pct_64_regions"]["workload"] = "64 regions of the configs[1] shape with 0.5 % substitution noise per base, one launch (seconds per launch = ms_per_batch / 1000)"
        del regsn
        # realistic reads at the size of the headline batch: 256 regions at 0.5 %.  One batch alone (as shipped: noisy regions split into
        # units) is the LATENCY of a batch; `in_flight` is the sustained rate with batches on several handles, as the headline is timed
        regs256 = make_regions_parallel("noisy", 256)
        oc["noise_0.5pct_256_regions"] = time_other_config(hb, regs256, a.kmer, opts, 2, local, flags=a.flags)
        oc["noise_0.5pct_256_regions"]["workload"] = "256 regions of the configs[1] shape with 0.5 % substitution noise per base; value = ONE batch at a time (a batch's latency: bound by the serial chain of its slowest region); in_flight = sustained rate with batches in flight"
        oc["noise_0.5pct_256_regions"]["in_flight"] = time_inflight(hb, regs256, a.kmer, opts, local, handles=a.noisy_inflight, steps=3 * a.noisy_inflight, flags=128 | a.flags, wg=256)
        try:                                                  # the same through the retained driver surface: runner.run() with its launch queue eight deep
            r_short = time_runner(synth, regs256, a.kmer, cycles=24, extra={"run_depth": a.noisy_inflight, "throughput_mode": "1"})
            oc["noise_0.5pct_256_regions"]["runner_end_to_end"] = time_runner(synth, regs256, a.kmer, cycles=96, extra={"run_depth": a.noisy_inflight, "throughput_mode": "1"},
                note="runner.run() wall time over 96 x 256 noisy targets (submit of 2-bit packed reads + GPU stages + native call tail + per-target Python), run_depth = %d launched batches in flight, throughput_mode (BK_CFG_NO_SPLIT, 256-thread workgroups), after one untimed warm-up run" % a.noisy_inflight)
            r_long = oc["noise_0.5pct_256_regions"]["runner_end_to_end"]
            # a run starts with ~10 fresh handles (device and pinned allocations, the first batch of each sizes its arenas): a fixed few seconds;
            # the rate a long sample sees is the marginal one, from the difference of the two run lengths
            r_long["short_run"] = {"value": r_short["value"], "regions": r_short["regions"], "seconds": r_short["seconds"]}
            if r_long["seconds"] > r_short["seconds"]:
                r_long["steady_state"] = {"value": round((r_long["regions"] - r_short["regions"]) / (r_long["seconds"] - r_short["seconds"]), 1), "unit": "regions/s",
                                          "note": "(regions of the long run - regions of the short run) / (their wall times' difference): what every further batch of a sample costs; the fixed start-up is %.1f s" % max(0.0, r_short["seconds"] - r_short["regions"] * (r_long["seconds"] - r_short["seconds"]) / (r_long["regions"] - r_short["regions"]))}
        except Exception as ex:
            oc["noise_0.5pct_256_regions"]["runner_end_to_end"] = {"error": repr(ex)}
        oc["noise_0.5pct_256_regions"]["in_flight"]["note"] = "bk_config.flags = BK_CFG_NO_SPLIT, asm_wg_threads = 256: with the chip full of other batches one workgroup per region costs less than 16 units per region (same rows: rows_identical_across_batches compares every batch with a batch that ran alone)"
        del regs256
    except Exception as ex:                      # never lose what was measured to a later side measurement
        oc["error"] = repr(ex)
    return oc


def time_other_config(hb, regions, k, opts, reps, device, flags=0):
    """whole path (GPU stages + native call tail) over one batch, inputs resident; returns regions/s and details"""
    eng = hb.Engine(kmer_size=k, rc_thresh=2, device=device, flags=flags)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens, partners=[p[4] for p in r.partners]) for r in regions])
    eng.set_call_context(call_context_text(regions, opts))
    eng.run(hb.BK_STAGE_ALL)
    eng.call_blob()
    t0 = time.perf_counter()
    ncalls = 0
    for _ in range(reps):
        eng.run(hb.BK_STAGE_ALL, sync=False)
        eng.fetch()
        raw = eng.call_blob()
        ncalls = raw.count(b"\n")
    dt = (time.perf_counter() - t0) / reps
    out = {"regions": len(regions), "reads_per_region": int(regions[0].reads.shape[0]), "k": k, "value": round(len(regions) / dt, 1), "unit": "regions/s",
           "ms_per_batch": round(dt * 1e3, 2), "kernels_ms": {"kmer": round(eng.kernel_ms(1), 2), "asm": round(eng.kernel_ms(2), 2), "sw": round(eng.kernel_ms(3), 2)},
           "contigs": int(eng.stat(6)), "sv_calls": int(ncalls), "nw_cells": int(eng.stat(0)), "failed_regions": int(eng.stat(22)),
           "dp_tcups": round(eng.stat(0) / dt / 1e12, 3),
           "dp_reads_through_score_sweep": int(eng.stat(30)), "dp_reads_swept_again_in_full": int(eng.stat(31))}
    eng.close()
    return out


def reference_python_timing():
    """the reference's OWN Python hot path timed on this workload -- in the build container, where the reference is (it cannot
    travel to the GPU box in any form): the committed result of tools/time_reference.py, quoted with its source"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "reference_python_timing.json")))
    if not fs:
        return None
    try:
        d = json.load(open(fs[-1]))
        return {"value": d["regions_per_s"], "unit": "regions/s", "cores": 1, "regions": d["regions"], "seconds": d["seconds"], "where": d["where"],
                "what": d["what"], "source": os.path.relpath(fs[-1], ROOT)}
    except Exception:
        return None


def usable_cores():
    """cores this process may really use: the affinity mask, capped by the container's CPU quota (cgroup cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for fn in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(fn).read().split()
            if fn.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def time_runner(synth, regions, kmer, cycles=1, extra=None, note=None):
    """the retained driver surface end to end (breakmer_amd.sv_processor.runner.run: per-target objects, batches on two handles,
    submit = host 2-bit packing + H2D included, native call tail, rows in target order), code-matrix inputs, no output files.
    cycles > 1: the same regions again under further target names (more batches per run without more host memory)"""
    import tempfile
    from breakmer_amd import sv_processor as sp
    d = tempfile.mkdtemp()
    bed, genes, data = [], ["header"], {}
    from breakmer_amd import hip_backend as hb_
    packed = {id(r): hb_.pack_reads(r.reads, r.read_lens) for r in regions}      # what a read extraction that packs as it goes hands over (BK_SUBMIT_PACKED); outside the clock like the extraction itself
    for c in range(cycles):
        for r in regions:
            name = r.name + ("C%d" % c if c else "")
            bed.append("\t".join([r.chrom, str(r.start), str(r.end), name, "exon"]))
            genes.append("\t".join(["0", name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [name]))
            data[name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=packed[id(r)])
    def config(tag, nb):
        open(os.path.join(d, tag + ".bed"), "w").write("\n".join(bed[:nb]) + "\n")
        open(os.path.join(d, tag + ".txt"), "w").write("\n".join(genes[:nb + 1]) + "\n")
        c = {"analysis_name": tag, "targets_bed_file": os.path.join(d, tag + ".bed"), "gene_annotation_file": os.path.join(d, tag + ".txt"),
             "kmer_size": str(kmer), "keep_repeat_regions": True, "batch_regions": 256}
        c.update(extra or {})
        return c
    if cycles > 1:                                      # untimed: creates the handles the process keeps between runs
        sp.runner(config("warmup", min(1536, len(bed))), region_data=data).run()
    cfg = config("bench", len(bed))
    t0 = time.perf_counter()
    rows = sp.runner(cfg, region_data=data).run()
    dt = time.perf_counter() - t0
    return {"value": round(len(data) / dt, 1), "unit": "regions/s", "regions": len(data), "rows": len(rows), "seconds": round(dt, 3),
            "batches": (len(data) + 255) // 256,
            "note": note or "runner.run() wall time: submit of 2-bit packed reads (BK_SUBMIT_PACKED: row copies + H2D) + GPU stages + native call tail + per-target Python objects (no output files), "
                    "after one untimed warm-up run of 6 batches (the process keeps its handles between runs); 2 x 256 distinct regions cycled under 16 sets of target names; "
                    "the timed `value` above excludes submit (inputs resident, SURVEY 8d)"}


def _cpu_region(args):
    """one region through the C oracle (worker of the all-core CPU baseline); returns the number of contigs"""
    rid, depth, read_len, kmer = args
    from breakmer_amd import synth
    from oracle import bk_oracle as bo
    r = synth.make_region(rid, depth=depth, L=read_len, sv_type="del")
    t0 = time.perf_counter()
    want, _ = bo.assemble_region(synth.BASES[r.reads], [r.window_str], kmer, 2)
    for c in want:
        bo.realign(c["seq"], [r.window_str])
    return len(want), time.perf_counter() - t0


def main():
    a = parse()
    if a.side_configs_only:                                  # child of the default run: the side measurements in a process of their own
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        from breakmer_amd import hip_backend as hb, synth
        if a.side_configs_only == 2:                         # the same noisy batch with one unit per region (how every region ran until round 4): the comparison figure
            regsn = [synth.make_region(50000 + i, depth=a.depth, L=a.read_len, sv_type="del", noise=0.005) for i in range(64)]
            oc = {"noise_0.5pct_64_regions_one_unit": time_other_config(hb, regsn, a.kmer, default_opts(), 2, int(os.environ.get("LOCAL_RANK", "0")), flags=128)}
            oc["noise_0.5pct_64_regions_one_unit"]["workload"] = "the same 64 noisy regions with bk_config.flags = BK_CFG_NO_SPLIT (128): no component split, one assembler workgroup per region (the default until round 4)"
            print(json.dumps(oc), flush=True)
            return
        cpu = {}
        if a.cpu_sample > 0:                                     # the CPU figures first: nothing of this process has touched the GPU yet
            try:
                nc = usable_cores()
                cpu = cpu_side_baselines({"configs[3]": ("cfg3", max(16, nc)), "configs[4]": ("cfg4", min(nc, 16)), "noise": ("noisy", max(16, nc))})
            except Exception as ex:
                cpu = {"error": repr(ex)}
        oc = side_configs(a, hb, synth, default_opts(), int(os.environ.get("LOCAL_RANK", "0")))      # (--flags applies: diagnostic A/B runs)
        for key, ent in oc.items():
            if isinstance(ent, dict) and "value" in ent:
                src = cpu.get("noise" if key.startswith("noise") else key) if "error" not in cpu else {"error": cpu["error"]}
                if src:
                    ent["cpu_baseline"] = src
        print(json.dumps(oc), flush=True)
        return
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a))
    # ---- CPU baseline on ALL host cores (process pool over regions through the C oracle).  Runs first, before anything
    #      initialises the GPU in this process: the workers are spawned (fork + exec) from a GPU-free parent.
    cpu_all, cpu_cores = None, 1
    if a.gpus == 1 and not a.force_dist and a.cpu_sample > 0:
        cpu_cores = usable_cores()
        if cpu_cores > 1:
            import multiprocessing as mp
            nall = cpu_cores * 32                                # ~13 regions/s/core: a few seconds of wall time per core
            with mp.get_context("spawn").Pool(cpu_cores) as pool:
                pool.map(_cpu_region, [(i, 60, a.read_len, a.kmer) for i in range(cpu_cores)])      # start-up (imports, library build/load) outside the clock
                t2 = time.perf_counter()
                res = pool.map(_cpu_region, [(i % 256, a.depth, a.read_len, a.kmer) for i in range(nall)], chunksize=1)
                all_dt = time.perf_counter() - t2
            cpu_all = {"value": round(nall / all_dt, 3), "regions": nall, "contigs": int(sum(x[0] for x in res))}
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    dist = world > 1 or a.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    if dist:
        import torch.distributed as td
        if "RANK" not in os.environ:                       # --force-dist on one GPU without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1])); s.close()
            td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        else:
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
    from breakmer_amd import hip_backend as hb, synth, collate
    if a.lib:
        hb.load_library(os.path.abspath(a.lib))
    # The SAME batch per GPU per step at every N (weak scaling: the 1 -> 8 GPU curve compares like with like; a larger batch
    # raises per-GPU throughput by itself on this latency-bound path).  256 = configs[1] on each GPU; at N = 8 configs[2]'s
    # 4,096 regions are two consecutive steps of 8 x 256 (with 6 steps in flight 1,536 regions per GPU are resident anyway).
    n_regions = a.regions if a.regions > 0 else 256

    # ---- inputs: disjoint region ids per rank (weak scaling), packed + resident before timing -------
    ids = range(rank * n_regions, (rank + 1) * n_regions)
    regions = [synth.make_region(i, depth=a.depth, L=a.read_len, sv_type="del") for i in ids]
    stages = hb.BK_STAGE_ALL
    opts = default_opts()
    ctx_text = call_context_text(regions, opts)
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
    engs = []
    submit_ms = []
    if a.inflight <= 0:
        a.inflight = 12 if a.steps >= 60 else 8
    for _ in range(max(1, a.inflight)):                # one handle (= one HIP stream + its own buffers) per step in flight
        e = hb.Engine(kmer_size=a.kmer, rc_thresh=2, device=local, flags=a.flags, wg_threads=a.wg, max_candidates=a.max_candidates, max_contig_len=a.max_contig)
        t0 = time.perf_counter()
        e.submit(ins)
        submit_ms.append((time.perf_counter() - t0) * 1e3)
        e.set_call_context(ctx_text)
        engs.append(e)
    eng = engs[0]
    pack_ms, h2d_ms = eng.stat(20) / 1e3, eng.stat(21) / 1e3
    last_rows = {}

    def take_rows(raw):
        last_rows["n"] = raw.count(b"\n") + (1 if raw and not raw.endswith(b"\n") else 0)
        last_rows["raw"] = raw
        if dist:                                       # collate the variable-length records of all ranks (RCCL all-gather)
            gather(np.frombuffer(raw, dtype=np.uint8))

    def finish_step(e, relaunch):
        """consume one step of handle e: wait + copy its records to the host, start the handle's next step at once (its
        device buffers are free again), and have the SV-call tail made on the host copy while those kernels execute -- on the
        library's own thread (bk_call_async), so this thread goes on to the next handle; the calls of a step are picked up when
        its handle comes round again (or by collect_tails() at the end of the run: every step's tail is inside the timed region)"""
        if tails.get(id(e)):
            take_rows(e.call_blob())                   # the tail of this handle's previous step: made meanwhile
        e.fetch()
        ms = [e.kernel_ms(j + 1) for j in range(3)]
        if relaunch:
            e.run(stages, sync=False)                  # group + k-mer select + assemble + realign on the GPU (async)
        e.call_async()                                 # SV-call tail (host C++) over the fetched copy: one tab-separated record per call, region order
        tails[id(e)] = True
        return ms

    tails = {}

    def collect_tails(k):
        for j in range(len(engs)):                     # in step order: the oldest outstanding step is on handle k % inflight
            e = engs[(k + j) % len(engs)]
            if tails.get(id(e)):
                take_rows(e.call_blob())
                tails[id(e)] = False

    # Collation: the records of GATHER_EVERY steps go out in one all-gather, framed per step ([n_steps | (length, records)*]
    # per rank, fixed capacity), two slots so that a gather overlaps the kernels of the following steps (the records are only
    # consumed after the run).  The driver surface collates once per run (breakmer_amd/collate.py: sizes, then padded
    # payload); one RCCL launch per step cost the host loop 0.25 ms of 1.3 ms on one GPU.
    GATHER_EVERY = 8
    CAP = GATHER_EVERY * (1 << 18) + 64
    slots = []
    if dist:
        for _ in range(2):
            slots.append({"host": torch.zeros(CAP, dtype=torch.uint8).pin_memory(), "dev": torch.zeros(CAP, dtype=torch.uint8, device="cuda"),
                          "out": torch.zeros(world * CAP, dtype=torch.uint8, device="cuda"), "work": None})
    gstate = {"n": 0, "collated": 0, "last": b"", "pending": []}

    def flush_gather():
        blobs = gstate["pending"]
        if not blobs:
            return
        gstate["pending"] = []
        sl = slots[gstate["n"] % 2]
        gstate["n"] += 1
        if sl["work"] is not None:
            sl["work"].wait()
        _hv, o = collate.frame_steps(blobs, CAP, out=sl["host"].numpy())
        sl["dev"][:o].copy_(sl["host"][:o], non_blocking=True)
        sl["work"] = td.all_gather_into_tensor(sl["out"], sl["dev"], async_op=True)

    def gather(blob):
        gstate["pending"].append(blob)
        if len(gstate["pending"]) >= GATHER_EVERY:
            flush_gather()

    def drain():
        """send what is pending, wait for the outstanding all-gathers; returns the bytes of the LAST step collated over
        all ranks (rank order)"""
        tot = 0
        if dist:
            flush_gather()
        for sl in slots:
            if sl["work"] is not None:
                sl["work"].wait()
                sl["work"] = None
        if slots and gstate["n"]:
            sl = slots[(gstate["n"] - 1) % 2]
            parts = [steps[-1] if steps else b"" for steps in collate.deframe_gathered(sl["out"].cpu().numpy(), world, CAP)]
            tot = sum(len(x) for x in parts)
            gstate["last"] = b"".join(parts)
        gstate["collated"] = tot
        return tot

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        """K steps, step s on handle s % inflight; every handle always has its next step queued before the host turns
        to the call tail of the step it just collected"""
        acc = [0.0, 0.0, 0.0]
        launched = 0
        for j in range(min(len(engs), k)):
            engs[j].run(stages, sync=False)
            launched += 1
        for s in range(k):
            relaunch = launched < k
            ms = finish_step(engs[s % len(engs)], relaunch)
            launched += 1 if relaunch else 0
            for j in range(3):
                acc[j] += ms[j]
        collect_tails(k)
        drain()
        return acc

    def allmax(x):
        if not dist:
            return x
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    run_steps(a.warmup)
    barrier()
    t0 = time.perf_counter()
    kmer_ms, asm_ms, sw_ms = run_steps(a.steps)
    barrier()
    dt = allmax(time.perf_counter() - t0)
    total_regions = n_regions * world * a.steps
    value = total_regions / dt
    collated = gstate["collated"]
    # The same loop over >= 100 steps: with K = 20 the timed region is 22 ms and holds the fill and drain of the 6-deep pipeline
    # (about a step's worth each); `value` keeps the contract (EXACTLY K steps), this leg says what the loop sustains.
    long_run = None
    if a.steps < 100:
        barrier()
        tl = time.perf_counter()
        run_steps(100)
        barrier()
        ldt = allmax(time.perf_counter() - tl)
        long_run = {"value": round(n_regions * world * 100 / ldt, 1), "unit": "regions/s", "steps": 100, "ms_per_step": round(ldt / 100 * 1e3, 3)}
    if a.dump_collated and rank == 0:
        with open(a.dump_collated, "wb") as f:
            f.write(gstate["last"])
        with open(a.dump_collated + ".rank0", "wb") as f:
            f.write(last_rows.get("raw", b""))
    # ---- the same loop with every batch SUBMITTED again (the drop-in's sustained rate: PCIe-inclusive).  The reads go over 2-bit
    #      packed (BK_SUBMIT_PACKED: 96 MB per 256-region batch; one byte per base would be 384 MB), the submit of step s + inflight
    #      runs on the library's thread (BK_SUBMIT_ASYNC) while the host collects the steps before it.
    with_submit = None
    if not dist:
        pins = [hb.RegionInput(None, r.window, packed=hb.pack_reads(r.reads, r.read_lens)) for r in regions]
        ctx_keep = call_context_text(regions, opts, keep_tables=True)      # (every handle got the tables with ctx_text above)
        def run_steps_submit(k):
            """step s: pick up the calls of engine s % n and hand it its next batch (asynchronous submit on the library's thread);
            the engine half a turn ahead, whose submit was started n/2 steps ago, gets its context and is launched, and the
            library's thread of that handle waits for its kernels, takes the records and makes the calls (bk_call_async) -- the way
            sv_processor.runner drives the library.  So n/2 submits and n/2 runs are in flight at any time."""
            n = len(engs); half = max(1, n // 2)
            state = ["idle"] * n                            # idle -> submitted -> running
            raw = b""
            done = started = 0
            t = 0
            while done < k:
                i = t % n; e = engs[i]
                if state[i] == "running":
                    raw = e.call_blob(); done += 1; state[i] = "idle"
                if state[i] == "idle" and started < k:
                    e.submit(pins, wait=False); state[i] = "submitted"; started += 1
                j = (t + half) % n; f = engs[j]
                if state[j] == "submitted":
                    f.set_call_context(ctx_keep); f.run(stages, sync=False); f.call_async(); state[j] = "running"
                t += 1
            return raw
        run_steps_submit(len(engs))
        barrier()
        tws = time.perf_counter()
        ksub = 48 if a.steps >= 8 else max(len(engs), a.steps)      # its own number of steps (a side figure): 20 steps of 3 ms are a 60 ms window
        raw_ws = run_steps_submit(ksub)
        barrier()
        wdt = time.perf_counter() - tws
        with_submit = {"value": round(n_regions * ksub / wdt, 1), "unit": "regions/s", "steps": ksub, "ms_per_step": round(wdt / ksub * 1e3, 3),
                       "bytes_per_step": int(sum(p_.reads.nbytes for p_ in pins)), "same_rows_as_resident": raw_ws == last_rows.get("raw"),
                       "note": "every step submits its batch again (BK_SUBMIT_PACKED | BK_SUBMIT_ASYNC: 2-bit packed rows copied + H2D on the library's thread), context set again, then the same stages + call tail (bk_call_async, as sv_processor.runner drives it)"}
        for e in engs:                                      # back to the resident inputs for what follows
            e.submit(ins); e.set_call_context(ctx_text); e.run(stages)
    # ---- the same steps strictly one after the other (one handle, nothing in flight): the kernel durations of THIS pass
    #      are exclusive (no co-running batches stretch them) and are what the roofline figures use
    ks = max(2, min(a.steps, 8))
    lat = hb.Engine(kmer_size=a.kmer, rc_thresh=2, device=local, flags=a.flags, wg_threads=512)       # the latency-tuned workgroup size
    lat.submit(ins)
    lat.set_call_context(ctx_text)
    saved, engs[:] = list(engs), [lat]
    run_steps(2)
    barrier()
    ts = time.perf_counter()
    s_k, s_a, s_w = run_steps(ks)
    barrier()
    sdt = allmax(time.perf_counter() - ts)
    engs[:] = saved
    serial = {"value": round(n_regions * world * ks / sdt, 1), "ms_per_step": round(sdt / ks * 1e3, 3), "steps": ks,
              "kernels_ms": {"bk_kmer_kernel": round(s_k / ks, 3), "bk_asm_kernel": round(s_a / ks, 3), "bk_sw_kernel": round(s_w / ks, 3)}}

    if rank == 0:
        alg_bytes = eng.stat(3)                         # 2-bit reads + 4 B/read + window fwd+rc + ~2 KB out, summed over regions
        cells, calls = eng.stat(0), eng.stat(1)
        asm_excl_s = s_a / ks / 1e3                     # exclusive duration of the dominant kernel (one handle)
        step_s = dt / a.steps
        achieved = alg_bytes / asm_excl_s / 1e9
        prof = {}
        for name in ("traffic.json", "valu.json", "kernel_ms.json"):
            fn = os.path.join(ROOT, "profiles", name)
            if os.path.isfile(fn):
                try:
                    prof.update(json.load(open(fn)))
                except Exception:
                    pass
        # The dominant kernel OF THE TIMED REGION: the 256-thread build when batches are in flight (default), the 512-thread build
        # otherwise.  Its average launch duration comes from the HIP events the library records on the handle's stream around
        # every launch of the timed steps (bk_last_kernel_ms); the counters (HBM traffic, VALU instructions) are per-launch
        # figures of THAT kernel from the committed rocprofv3 passes of the round (`source`), collected with one launch at a
        # time because PMC collection serialises kernels (tools/profile_round.sh regenerates them with the bench line).
        kname = "bk_asm_kernel_w4" if int(eng.stat(25)) == 256 else "bk_asm_kernel"
        k_ms = asm_ms / a.steps
        achieved = alg_bytes / (k_ms / 1e3) / 1e9
        traffic = prof.get(kname + "_bytes_per_launch")
        # the same kernel's average launch duration in the committed rocprofv3 --kernel-trace --stats run of this command (profiles/kernel_ms.json,
        # written by tools/summarize_profile.py from the round's profile run): the live figure and this one differ by what the batches in flight
        # do to each other in a given run and by what the HIP events bracket (they include the scheduling kernel and the gaps between launches)
        k_ms_prof = prof.get(kname + "_avg_ms")
        kmer_excl_s = s_k / ks / 1e3
        kmer_traffic = prof.get("bk_kmer_kernel_bytes_per_launch")
        # integer-VALU roofline of the assembler.  Algorithmic lane-ops per DP cell of olc.nw (olc.py:62-74): three candidate
        # sums, the match/mismatch compare + select, one three-way max = 6; `peak` = the chip's VALU issue rate / 6.  What the
        # kernel really issues per algorithmic cell (SQ_INSTS_VALU x 64 / cells: the 7-op cell of this encoding, pipeline
        # fill/drain, planning, retire) gives `utilisation` = issued lane-ops / issue peak.
        ALG_OPS = 6.0
        lpc = prof.get(kname + "_valu_laneops_per_cell")
        valu = {"bound": "valu", "unit": "TCUPS", "kernel": kname, "peak_laneops_per_s": VALU_PEAK_LANEOPS, "algorithmic_laneops_per_cell": ALG_OPS,
                "peak": round(VALU_PEAK_LANEOPS / ALG_OPS / 1e12, 3),
                "achieved": round(cells / step_s / 1e12, 4), "achieved_kernel_exclusive": round(cells / asm_excl_s / 1e12, 4),
                "measured_laneops_per_cell": lpc, "source": prof.get("valu_source")}
        valu["frac"] = round(valu["achieved"] / valu["peak"], 4)
        if lpc:
            valu["utilisation"] = round(lpc * cells / step_s / VALU_PEAK_LANEOPS, 4)
        out = {
            "metric": "target regions/sec at 500x 150bp, 31-mers; achieved HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "regions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d regions/GPU x %d reads x %d bp (%dx), planted 200 bp deletion, k=%d"
                                   % ("configs[1]" if world == 1 else ("configs[2] (4,096 regions = %d steps)" % (4096 // (world * n_regions)) if 4096 % (world * n_regions) == 0 else "configs[2]-shaped"),
                                      n_regions, regions[0].reads.shape[0], a.read_len, a.depth, a.kmer),
                       "stages": "group reads + k-mer select + assemble (olc.nw) + realign on the GPU, SV-call tail in host C++, rows collated",
                       "regions_per_gpu_per_step": n_regions, "regions_total_per_step": n_regions * world,
                       "sv_calls_per_step": last_rows.get("n", 0),
                       "sv_calls_parity": "k-mer selection, assembly and the caller are pinned by vectors generated from the reference itself; the realign records the calls are made from follow the contract of oracle/bk_oracle.c (BLAT is absent: that stage is parity-unpinned, DESIGN 4.3)",
                       "steps_in_flight": len(engs),
                       "asm_workgroup_threads": int(eng.stat(25)), "asm_workgroups_per_cu": int(eng.stat(23)),
                       "one_step_at_a_time_workgroup_threads": 512,
                       "collated_bytes_per_step": collated if dist else None,
                       "parallelism": "regions sharded per GPU, all-gather of result records"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": prof.get("traffic_source"),
                         "kernel": kname, "kernel_ms": round(k_ms, 3),
                         "kernel_ms_profiled": k_ms_prof, "frac_profiled": (round(alg_bytes / (k_ms_prof / 1e3) / 1e9 / HBM_PEAK_GBS, 6) if k_ms_prof else None),
                         "kernel_ms_profiled_source": prof.get("kernel_ms_source"),
                         "chip_ms_per_step": round(step_s * 1e3, 3),
                         "chip_ms_per_step_note": "wall time of the timed region / steps = the time the chip spends per step with %d batches in flight (profiles/<round>/overlap_summary.csv: span of a 100-step window of the kernel trace / 100, the device never idle inside it); algorithmic bytes / chip_ms_per_step is `hbm_path`" % len(engs),
                         "kernel_ms_note": "average duration of ONE launch of this kernel over the timed steps (HIP events on its stream), not chip time per step: %d batches are in flight and their launches overlap (the durations of the three kernels sum to more than ms_per_step x steps in flight)" % len(engs),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "exclusive": {"kernel": "bk_asm_kernel", "kernel_ms": round(s_a / ks, 3), "achieved": round(alg_bytes / asm_excl_s / 1e9, 3),
                                       "traffic": prof.get("bk_asm_kernel_bytes_per_launch"),
                                       "note": "the one-step-at-a-time pass below (one handle, 512-thread build): nothing co-runs"},
                         "note": "the path is integer-VALU/latency bound, not HBM bound (SURVEY 8d): see roofline_valu"},
            "roofline_valu": valu,
            # the k-mer kernel is the one stage SURVEY 8d calls HBM-bound (one streaming pass over the packed reads): its own line
            "roofline_kmer": {"bound": "hbm", "kernel": "bk_kmer_kernel", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                              "kernel_ms": round(s_k / ks, 4), "kernel_ms_note": "exclusive: the one-step-at-a-time pass (one handle, 1,024-thread workgroups, nothing co-runs)",
                              "kernel_ms_inflight": round(kmer_ms / a.steps, 4),
                              "algorithmic_bytes_per_launch": alg_bytes, "achieved": round(alg_bytes / kmer_excl_s / 1e9, 2), "frac": round(alg_bytes / kmer_excl_s / 1e9 / HBM_PEAK_GBS, 5),
                              "traffic": kmer_traffic, "traffic_over_algorithmic": (round(kmer_traffic / alg_bytes, 3) if kmer_traffic else None),
                              "traffic_rate": (round(kmer_traffic / kmer_excl_s / 1e9, 1) if kmer_traffic else None), "traffic_source": prof.get("traffic_source")},
            "hbm_path": {"achieved": round(alg_bytes / step_s / 1e9, 3), "unit": "GB/s", "frac": round(alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, 6),
                         "note": "algorithmic bytes of one step / ms_per_step (all kernels, batches in flight)"},
            "kernels_ms": serial["kernels_ms"],
            "kernels_ms_inflight": {"bk_kmer_kernel": round(kmer_ms / a.steps, 3), "bk_asm_kernel": round(asm_ms / a.steps, 3), "bk_sw_kernel": round(sw_ms / a.steps, 3)},
            "dp_gcups": round(cells / step_s / 1e9, 1), "dp_cells_per_step": cells, "nw_calls_per_step": calls,
            "submit_ms": round(sum(submit_ms) / len(submit_ms), 2), "submit_pack_ms": round(pack_ms, 2), "h2d_ms": round(h2d_ms, 2),
            "submit_note": "bk_submit_regions of one %d-region batch (host 2-bit packing + H2D), outside the timed region" % n_regions,
            "one_step_at_a_time": serial,
            "value_100_steps": long_run if long_run else {"value": round(value, 1), "unit": "regions/s", "steps": a.steps, "ms_per_step": round(step_s * 1e3, 3)},
            "value_with_submit": with_submit,
        }
        if world == 1 and a.other_configs:
            try:
                out["runner_end_to_end"] = time_runner(synth, regions + [synth.make_region(n_regions + i, depth=a.depth, L=a.read_len, sv_type="del") for i in range(n_regions)], a.kmer, cycles=16)
            except Exception as ex:
                out["runner_end_to_end"] = {"error": repr(ex)}
        # ---- other BASELINE configs on one GPU (not the headline; whole path incl. call tail, inputs resident) -----
        if world == 1 and a.other_configs:
            # In a process of their own: a device fault in one of these side measurements ("known defect": noisy regions under
            # load) must not take the headline line with it.  The child prints one JSON object; anything else becomes an "error" entry.
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--side-configs-only", "1", "--kmer", str(a.kmer), "--depth", str(a.depth), "--read-len", str(a.read_len),
                   "--cfg3-regions", str(a.cfg3_regions), "--cfg4-regions", str(a.cfg4_regions), "--noisy-inflight", str(a.noisy_inflight), "--cpu-sample", str(a.cpu_sample)]
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=2400)
                lines = [ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")]
                oc = json.loads(lines[-1]) if (pr.returncode == 0 and lines) else {"error": "side measurements ended with code %d" % pr.returncode, "stderr_tail": pr.stderr[-400:]}
            except Exception as ex:
                oc = {"error": repr(ex)}
            try:                                             # ... and the one-unit comparison run of the noisy batch
                cmd2 = [sys.executable, os.path.abspath(__file__), "--side-configs-only", "2", "--kmer", str(a.kmer), "--depth", str(a.depth), "--read-len", str(a.read_len), "--cpu-sample", "0"]
                pr2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=300)
                lines2 = [ln for ln in pr2.stdout.strip().splitlines() if ln.startswith("{")]
                oc.update(json.loads(lines2[-1]) if (pr2.returncode == 0 and lines2) else {"noise_0.5pct_64_regions_one_unit": {"error": "ended with code %d" % pr2.returncode}})
            except Exception as ex:
                oc["noise_0.5pct_64_regions_one_unit"] = {"error": repr(ex)}
            if isinstance(oc.get("noise_0.5pct_64_regions"), dict) and "cpu_baseline" in oc["noise_0.5pct_64_regions"] and isinstance(oc.get("noise_0.5pct_64_regions_one_unit"), dict):
                oc["noise_0.5pct_64_regions_one_unit"].setdefault("cpu_baseline", oc["noise_0.5pct_64_regions"]["cpu_baseline"])      # (the same regions)
            out["other_configs"] = oc
        # ---- CPU baseline: the oracle (C port of the reference algorithm) on the host cores, bounded sample -------
        if world == 1 and a.cpu_sample > 0:
            from oracle import bk_oracle as bo
            bo.lib()
            ns = min(a.cpu_sample, n_regions)
            asc = [synth.BASES[regions[i].reads] for i in range(ns)]
            wins = [regions[i].window_str for i in range(ns)]
            t1 = time.perf_counter()
            wants = [bo.assemble_region(asc[i], [wins[i]], a.kmer, 2)[0] for i in range(ns)]
            for i in range(ns):
                for c in wants[i]:
                    bo.realign(c["seq"], [wins[i]])
            cpu_dt = time.perf_counter() - t1
            ok = all([{k: v for k, v in c.items() if k not in ("total_reads", "n_hits")} for c in eng.contigs(i)] == wants[i] for i in range(ns))
            one = ns / cpu_dt
            allc, cores = cpu_all, cpu_cores
            out["cpu_baseline"] = {"value": allc["value"] if allc else round(one, 3), "unit": "regions/s", "cores": cores if allc else 1, "kind": "port",
                                   "value_1core": round(one, 3),
                                   "sample": "oracle/bk_oracle.c (T1+K1/K2+init_assembly+realign): %d regions of the same batch on 1 core; %s"
                                             % (ns, ("%d regions over %d worker processes (all host cores)" % (allc["regions"], cores)) if allc else "single-core host"),
                                   "parity_on_sample": bool(ok),
                                   "reference_python": reference_python_timing()}
    else:
        out = None
    if dist:
        td.destroy_process_group()
    if out is not None:
        # the ONE JSON line is the last thing on stdout: RCCL writes its version banner through C stdio, which a pipe
        # buffers until exit -- push it out first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
