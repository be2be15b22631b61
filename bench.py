#!/usr/bin/env python3
"""bench.py -- BreaKmer hot path on MI355X: target regions/s at 500x 150 bp reads, 31-mers.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run).  A "step" is one pass of
the hot path (read grouping -> k-mer selection -> assembly [-> realign -> call]) over one batch of
synthetic regions (BASELINE.json configs[1]: 256 regions per GPU, 10,000 x 150 bp reads each, planted
200 bp deletion, k = 31), inputs already packed and resident in HBM.  Regions are independent, so
ranks get disjoint region ids (weak scaling) and the only exchange is the all-gather of the per-region
result records at the end of each step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The steps in flight live on separate HIP streams.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4); once RCCL has created its own streams the default leaves the library's streams sharing queues and the
# batches serialise (measured: 86 k -> 61 k regions/s).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=256, help="regions per GPU per step (configs[1])")
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=31)
    ap.add_argument("--cpu-sample", type=int, default=24, help="regions timed through the CPU oracle (0 = skip)")
    ap.add_argument("--inflight", type=int, default=3, help="steps in flight (independent batches on separate HIP streams)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (testing)")
    return ap.parse_args()


def call_context_text(regions, opts):
    """query_region / annotation / discordant-pair context of the batch for the native call tail."""
    from breakmer_amd import call_context as cc
    genes = {}
    for r in regions:
        genes[r.name] = ["chr" + r.chrom, r.start, r.end]
        for p in r.partners:
            genes[p[3]] = ["chr" + p[0], p[1], p[2]]
    lines = [cc.opts_line(opts)] + cc.tables_lines(genes, None)
    for i, r in enumerate(regions):
        qr = (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, "exon")])
        lines += cc.region_lines(i, qr, None, r.disc_reads, [(p[0], p[1]) for p in r.partners], r.read_ids)
    return "\n".join(lines) + "\n"


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1 or a.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    if dist:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
    from breakmer_amd import hip_backend as hb, synth

    # ---- inputs: disjoint region ids per rank (weak scaling), packed + resident before timing -------
    ids = range(rank * a.regions, (rank + 1) * a.regions)
    regions = [synth.make_region(i, depth=a.depth, L=a.read_len, sv_type="del") for i in ids]
    stages = hb.BK_STAGE_ALL
    from breakmer_amd.sv_processor import params as bk_params
    opts = dict(bk_params.DEFAULTS)
    opts["var_filter"] = ["indel", "rearrangement", "trl"]
    ctx_text = call_context_text(regions, opts)
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]
    engs = []
    for _ in range(max(1, a.inflight)):                # one handle (= one HIP stream + its own buffers) per step in flight
        e = hb.Engine(kmer_size=a.kmer, rc_thresh=2, device=local)
        e.submit(ins)
        e.set_call_context(ctx_text)
        engs.append(e)
    eng = engs[0]
    last_rows = {}

    def finish_step(e, relaunch):
        """consume one step of handle e: wait + copy its records to the host, start the handle's next step at once (its
        device buffers are free again), then run the SV-call tail on the host copy while those kernels execute"""
        e.fetch()
        ms = [e.kernel_ms(j + 1) for j in range(3)]
        if relaunch:
            e.run(stages, sync=False)                  # group + k-mer select + assemble + realign on the GPU (async)
        raw = e.call_blob()                            # SV-call tail (host C++): one tab-separated record per call, region order
        last_rows["n"] = raw.count(b"\n") + (1 if raw and not raw.endswith(b"\n") else 0)
        blob = np.frombuffer(raw, dtype=np.uint8)
        if dist:                                       # collate the variable-length records of all ranks (RCCL all-gather)
            gather(blob)
        return ms

    # Collation buffers: [8-byte length | records] per rank, fixed capacity, two slots so that the all-gather of step s
    # overlaps the kernels of step s+1 (the records are only consumed after the run).  The general two-phase
    # (sizes, then padded payload) exchange of the product path is breakmer_amd/collate.py.
    CAP = 1 << 20
    slots = []
    if dist:
        for _ in range(2):
            slots.append({"host": torch.zeros(CAP, dtype=torch.uint8).pin_memory(), "dev": torch.zeros(CAP, dtype=torch.uint8, device="cuda"),
                          "out": torch.zeros(world * CAP, dtype=torch.uint8, device="cuda"), "work": None})
    gstate = {"n": 0, "collated": 0}

    def gather(blob):
        sl = slots[gstate["n"] % 2]
        gstate["n"] += 1
        if sl["work"] is not None:
            sl["work"].wait()
        if blob.size + 8 > CAP:
            raise RuntimeError("collation record larger than %d bytes" % CAP)
        hv = sl["host"].numpy()
        hv[:8] = np.frombuffer(np.int64(blob.size).tobytes(), dtype=np.uint8)
        hv[8:8 + blob.size] = blob
        sl["dev"][:8 + blob.size].copy_(sl["host"][:8 + blob.size], non_blocking=True)
        sl["work"] = td.all_gather_into_tensor(sl["out"], sl["dev"], async_op=True)

    def drain():
        """wait for the outstanding all-gathers; returns the bytes collated by the last one (all ranks)"""
        tot = 0
        for sl in slots:
            if sl["work"] is not None:
                sl["work"].wait()
                sl["work"] = None
        if slots and gstate["n"]:
            sl = slots[(gstate["n"] - 1) % 2]
            heads = sl["out"].view(world, CAP)[:, :8].contiguous().cpu().numpy()
            tot = int(sum(int(np.frombuffer(heads[r].tobytes(), dtype=np.int64)[0]) for r in range(world)))
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
        drain()
        return acc

    run_steps(a.warmup)
    barrier()
    t0 = time.perf_counter()
    kmer_ms, asm_ms, sw_ms = run_steps(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    total_regions = a.regions * world * a.steps
    value = total_regions / dt
    # the same steps strictly one after the other (one handle, nothing in flight) for reference
    serial = None
    if len(engs) > 1:
        ks = max(2, min(a.steps, 8))
        only = engs[:1]
        saved, engs[:] = list(engs), only
        barrier()
        ts = time.perf_counter()
        s_k, s_a, s_w = run_steps(ks)
        barrier()
        sdt = time.perf_counter() - ts
        engs[:] = saved
        if dist:
            t = torch.tensor([sdt], device="cuda", dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            sdt = float(t.item())
        serial = {"value": round(a.regions * world * ks / sdt, 1), "ms_per_step": round(sdt / ks * 1e3, 3), "steps": ks,
                  "kernels_ms": {"bk_kmer_kernel": round(s_k / ks, 3), "bk_asm_kernel": round(s_a / ks, 3), "bk_sw_kernel": round(s_w / ks, 3)}}

    if rank == 0:
        # ---- roofline of the dominant kernel (assembler): algorithmic HBM bytes per launch / kernel time -----
        alg_bytes = eng.stat(3)                         # 2-bit reads + 4 B/read + window fwd+rc + ~2 KB out, summed over regions
        asm_s = asm_ms / a.steps / 1e3
        achieved = alg_bytes / asm_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tf):
            try:
                traffic = json.load(open(tf)).get("bk_asm_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        cells, calls = eng.stat(0), eng.stat(1)
        out = {
            "metric": "target regions/sec at 500x 150bp, 31-mers; achieved HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "regions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d regions/GPU x %d reads x %d bp (%dx), planted 200 bp deletion, k=%d"
                                   % (a.regions, regions[0].reads.shape[0], a.read_len, a.depth, a.kmer),
                       "stages": "group reads + k-mer select + assemble (olc.nw) + realign on the GPU, SV-call tail in host C++, rows collated",
                       "sv_calls_per_step": last_rows.get("n", 0),
                       "steps_in_flight": len(engs),
                       "collated_bytes_per_step": gstate["collated"] if dist else None,
                       "parallelism": "regions sharded per GPU, all-gather of result records"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "bk_asm_kernel", "kernel_ms": round(asm_ms / a.steps, 3),
                         "note": "path is integer-DP/latency bound, not HBM bound (SURVEY 8d); see dp_gcups"},
            "kernels_ms": {"bk_kmer_kernel": round(kmer_ms / a.steps, 3), "bk_asm_kernel": round(asm_ms / a.steps, 3),
                           "bk_sw_kernel": round(sw_ms / a.steps, 3)},
            "dp_gcups": round(cells / asm_s / 1e9, 1), "dp_cells_per_step": cells, "nw_calls_per_step": calls,
            "one_step_at_a_time": serial,
        }
        # ---- CPU baseline: the oracle (C port of the reference algorithm), 1 core, bounded sample -------
        if world == 1 and a.cpu_sample > 0:
            from oracle import bk_oracle as bo
            bo.lib()
            ns = min(a.cpu_sample, a.regions)
            asc = [synth.BASES[regions[i].reads] for i in range(ns)]
            wins = [regions[i].window_str for i in range(ns)]
            t1 = time.perf_counter()
            wants = [bo.assemble_region(asc[i], [wins[i]], a.kmer, 2)[0] for i in range(ns)]
            for i in range(ns):
                for c in wants[i]:
                    bo.realign(c["seq"], [wins[i]])
            cpu_dt = time.perf_counter() - t1
            ok = all([{k: v for k, v in c.items() if k not in ("total_reads", "n_hits")} for c in eng.contigs(i)] == wants[i] for i in range(ns))
            out["cpu_baseline"] = {"value": round(ns / cpu_dt, 3), "unit": "regions/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the %d regions of the same batch through oracle/bk_oracle.c (T1+K1/K2+init_assembly+realign), 1 thread" % (ns, a.regions),
                                   "parity_on_sample": bool(ok)}
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
