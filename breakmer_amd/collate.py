"""Collation of per-region SV rows across ranks: regions are independent, so the only
exchange of the whole path is one all-gather of variable-length records at the end --
sizes first, then a padded all-gather (RCCL has no gatherv).  Works on any torch.distributed
backend: `nccl` (= RCCL over xGMI) on the GPU box, `gloo` in the CPU tests."""
from __future__ import annotations

import json


def all_gather_bytes(payload: bytes, device=None):
    """-> list of bytes objects, one per rank (rank order)."""
    import torch
    import torch.distributed as td
    world = td.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    td.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    td.all_gather(out, buf)
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(out, sizes)]


def collate_results(results, summary, device=None):
    """runner hook: merge (results, summary) of all ranks (concatenated in rank order; the runner tags every
    row with its target's position and restores the reference's output order, sv_processor.py:175-176, 212-224)."""
    parts = all_gather_bytes(json.dumps({"r": results, "s": summary}).encode(), device)
    all_results, all_summary = [], {}
    for p in parts:
        d = json.loads(p.decode())
        all_results.extend(d["r"])
        all_summary.update(d["s"])
    return all_results, all_summary


def exchange_status(err, device=None):
    """runner hook, called by every rank before the collation: `err` = None (this rank is through its targets) or the text
    of what stopped it; -> the list of all ranks' values (rank order).  A rank that failed still takes part, so nobody waits
    in a collective for a rank that is gone."""
    parts = all_gather_bytes(json.dumps(err).encode(), device)
    return [json.loads(p.decode()) for p in parts]


# ---- framed step buffers (bench.py's multi-rank path): the records of several steps travel in ONE fixed-capacity buffer per
#      rank, [n_steps:int64 | (length:int64, records)*], gathered with all_gather_into_tensor (equal sizes: no size exchange)
def frame_steps(blobs, cap, out=None):
    """-> (uint8 numpy array of `cap` bytes, bytes used).  `blobs`: one bytes-like / uint8 array per step, step order;
    `out`: write into this uint8 array of at least `cap` bytes (a pinned staging buffer) instead of a fresh one."""
    import numpy as np
    need = 8 + sum(8 + len(b) for b in blobs)
    if need > cap:
        raise RuntimeError("collation records larger than %d bytes" % cap)
    hv = np.zeros(cap, dtype=np.uint8) if out is None else out
    hv[:8] = np.frombuffer(np.int64(len(blobs)).tobytes(), dtype=np.uint8)
    o = 8
    for b in blobs:
        b = np.frombuffer(bytes(b), dtype=np.uint8) if not hasattr(b, "dtype") else b
        hv[o:o + 8] = np.frombuffer(np.int64(b.size).tobytes(), dtype=np.uint8)
        hv[o + 8:o + 8 + b.size] = b
        o += 8 + b.size
    return hv, o


def deframe_steps(row):
    """one rank's framed buffer -> list of bytes, one per step (step order)"""
    import numpy as np
    row = np.asarray(row, dtype=np.uint8)
    ns = int(np.frombuffer(row[:8].tobytes(), dtype=np.int64)[0])
    o, out = 8, []
    for _ in range(ns):
        n = int(np.frombuffer(row[o:o + 8].tobytes(), dtype=np.int64)[0])
        out.append(row[o + 8:o + 8 + n].tobytes())
        o += 8 + n
    return out


def deframe_gathered(allbuf, world, cap):
    """the output of all_gather_into_tensor over the ranks' framed buffers -> per rank (rank order) the list of its steps"""
    import numpy as np
    a = np.asarray(allbuf, dtype=np.uint8).reshape(world, cap)
    return [deframe_steps(a[r]) for r in range(world)]
