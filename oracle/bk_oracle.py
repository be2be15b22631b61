"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the C oracle (oracle/bk_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libbk_oracle.so")
    src = os.path.join(_HERE, "bk_oracle.c")
    hdr = os.path.join(_HERE, "bk_oracle.h")
    if force or not os.path.isfile(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"], cwd=_HERE)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.bko_nw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_char_p]
        L.bko_nw.restype = None
        L.bko_cells.argtypes = [C.c_int]
        L.bko_cells.restype = C.c_uint64
        L.bko_nw_calls.argtypes = [C.c_int]
        L.bko_nw_calls.restype = C.c_uint64
        L.bko_group_reads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.bko_group_reads.restype = C.c_int
        L.bko_kmer_select.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_char_p), C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.bko_kmer_select.restype = C.c_int
        L.bko_init_assembly.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.bko_init_assembly.restype = C.c_void_p
        for f in ("ncontigs",):
            getattr(L, "bko_asm_" + f).argtypes = [C.c_void_p]
            getattr(L, "bko_asm_" + f).restype = C.c_int
        for f in ("contig_len", "contig_clen", "contig_nkmers", "contig_nreads"):
            getattr(L, "bko_asm_" + f).argtypes = [C.c_void_p, C.c_int]
            getattr(L, "bko_asm_" + f).restype = C.c_int
        L.bko_asm_contig_get.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.bko_asm_contig_get.restype = None
        L.bko_asm_read_flags.argtypes = [C.c_void_p, C.c_void_p]
        L.bko_asm_read_flags.restype = None
        L.bko_asm_free.argtypes = [C.c_void_p]
        L.bko_asm_free.restype = None
        _LIB = L
    return _LIB


def nw(seq1, seq2):
    """olc.nw -> the reference's 7-tuple (olc.py:107)."""
    L = lib()
    out = (C.c_int * 7)()
    a1 = C.create_string_buffer(len(seq1) + len(seq2) + 2)
    a2 = C.create_string_buffer(len(seq1) + len(seq2) + 2)
    L.bko_nw(seq1.encode(), len(seq1), seq2.encode(), len(seq2), out, a1, a2)
    return (a1.value.decode(), a2.value.decode(), out[2], out[3], out[4], out[5], out[6])


def _rows(seqs):
    """list[str] or uint8 ASCII matrix -> (contiguous bytes matrix, stride, lens)."""
    if isinstance(seqs, np.ndarray):
        m = np.ascontiguousarray(seqs, dtype=np.uint8)
        return m, m.shape[1], np.full(m.shape[0], m.shape[1], dtype=np.int32)
    n = len(seqs)
    stride = max([len(s) for s in seqs] + [1])
    m = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for i, s in enumerate(seqs):
        b = np.frombuffer(s.encode(), dtype=np.uint8)
        m[i, :len(b)] = b
        lens[i] = len(b)
    return m, stride, lens


def group_reads(seqs):
    """T1 (utils.py:239-244) -> (rep index per unique, nreads per unique)."""
    m, stride, lens = _rows(seqs)
    n = m.shape[0]
    rep = np.zeros(max(n, 1), dtype=np.int32)
    cnt = np.zeros(max(n, 1), dtype=np.int32)
    U = lib().bko_group_reads(m.ctypes.data, stride, lens.ctypes.data, n, rep.ctypes.data, cnt.ctypes.data)
    return rep[:U].copy(), cnt[:U].copy()


def kmer_select(seqs, refs, k, sc_seqs=None):
    """K1+K2 -> (list of mers sorted ascending, counts)."""
    m, stride, lens = _rows(seqs)
    n = m.shape[0]
    if sc_seqs is None:
        scm, scs, scl, nsc = None, 0, None, -1
    else:
        scm, scs, scl = _rows(sc_seqs)
        nsc = scm.shape[0]
    refs_b = [r.encode() for r in refs]
    arr = (C.c_char_p * len(refs_b))(*refs_b)
    rl = np.array([len(r) for r in refs_b], dtype=np.int32)
    cap = 1 << 16
    while True:
        om = np.zeros(cap * k, dtype=np.uint8)
        oc = np.zeros(cap, dtype=np.int32)
        got = lib().bko_kmer_select(m.ctypes.data, stride, lens.ctypes.data, n,
                                    scm.ctypes.data if scm is not None else None, scs,
                                    scl.ctypes.data if scl is not None else None, nsc,
                                    arr, rl.ctypes.data, len(refs_b), k, om.ctypes.data, oc.ctypes.data, cap)
        if got <= cap:
            break
        cap = got
    mers = [om[i * k:(i + 1) * k].tobytes().decode() for i in range(got)]
    return mers, oc[:got].copy()


def init_assembly(useqs, unreads, uindel, mers, counts, k, rc_thresh, read_len):
    """sv_assembly.init_assembly on grouped reads -> list of contig dicts + read flags."""
    L = lib()
    m, stride, lens = _rows(useqs)
    U = m.shape[0]
    unreads = np.ascontiguousarray(unreads, dtype=np.int32)
    uindel = np.ascontiguousarray(uindel, dtype=np.uint8)
    M = len(mers)
    mm = np.frombuffer("".join(mers).encode(), dtype=np.uint8).copy() if M else np.zeros(1, dtype=np.uint8)
    cc = np.ascontiguousarray(counts, dtype=np.int32) if M else np.zeros(1, dtype=np.int32)
    h = L.bko_init_assembly(m.ctypes.data, stride, lens.ctypes.data, unreads.ctypes.data, uindel.ctypes.data, U,
                            mm.ctypes.data, cc.ctypes.data, M, k, rc_thresh, read_len)
    out = []
    try:
        for c in range(L.bko_asm_ncontigs(h)):
            ln, cl = L.bko_asm_contig_len(h, c), L.bko_asm_contig_clen(h, c)
            nk, nr = L.bko_asm_contig_nkmers(h, c), L.bko_asm_contig_nreads(h, c)
            seq = np.zeros(ln, dtype=np.uint8)
            io = np.zeros(max(cl, 1), dtype=np.int32)
            ot = np.zeros(max(cl, 1), dtype=np.int32)
            kl = np.zeros(max(ln, 1), dtype=np.int32)
            ki = np.zeros(max(nk, 1), dtype=np.int32)
            ri = np.zeros(max(nr, 1), dtype=np.int32)
            L.bko_asm_contig_get(h, c, seq.ctypes.data, io.ctypes.data, ot.ctypes.data, kl.ctypes.data, ki.ctypes.data, ri.ctypes.data)
            out.append({"seq": seq.tobytes().decode(), "indel_only": io[:cl].tolist(), "others": ot[:cl].tolist(),
                        "kmer_locs": kl[:ln].tolist(), "kmers": [mers[i] for i in ki[:nk]], "reads": ri[:nr].tolist()})
        flags = np.zeros(max(U, 1), dtype=np.uint8)
        L.bko_asm_read_flags(h, flags.ctypes.data)
    finally:
        L.bko_asm_free(h)
    return out, flags[:U]


def assemble_region(read_seqs, refs, k, rc_thresh=2, indel_only=None, sc_seqs=None, find_index=False):
    """T1 -> K1/K2 -> init_assembly for one region; returns (contigs, info).  find_index: answer find_reads from a
    k-mer -> reads index instead of the reference's scan over all reads (same result; for full-size noisy regions)."""
    lib().bko_set_find_index(1 if find_index else 0)
    rep, cnt = group_reads(read_seqs)
    if isinstance(read_seqs, np.ndarray):
        useqs = read_seqs[rep]
        read_len = read_seqs.shape[1]
    else:
        useqs = [read_seqs[i] for i in rep]
        read_len = max(len(s) for s in read_seqs) if len(read_seqs) else 0
    uind = np.zeros(len(rep), dtype=np.uint8) if indel_only is None else np.asarray(indel_only, dtype=np.uint8)[rep]
    mers, counts = kmer_select(read_seqs, [r.upper() for r in refs], k, sc_seqs)      # Jellyfish counts soft-masked (lower-case) bases like any others
    contigs, flags = init_assembly(useqs, cnt, uind, mers, counts, k, rc_thresh, read_len)
    for c in contigs:
        c["reads"] = [int(rep[u]) for u in c["reads"]]          # representative read index (FASTQ order)
    return contigs, {"rep": rep, "nreads": cnt, "mers": mers, "counts": counts, "flags": flags}


def check_align_case(contig, read, mer, k, mode, nreads, indel_only, founder_nreads, pre=None):
    """G2 hook: state of a contig after one check_align (sv_assembly.py:449-504)."""
    L = lib()
    L.bko_check_align_case.restype = C.c_int
    cap = len(contig) + len(read) + (len(pre) if pre else 0) + 8
    seq = C.create_string_buffer(cap)
    ln, cl, nk = C.c_int(), C.c_int(), C.c_int()
    io = (C.c_int * cap)()
    ot = (C.c_int * cap)()
    km = C.create_string_buffer(cap * 4 * k)
    meta = (C.c_int * (cap * 16))()
    pb = pre.encode() if pre else None
    m = L.bko_check_align_case(contig.encode(), len(contig), read.encode(), len(read), pb, len(pre) if pre else 0,
                               mer.encode(), k, 1 if mode == "grow" else 0, nreads, 1 if indel_only else 0, founder_nreads,
                               seq, C.byref(ln), io, ot, C.byref(cl), km, meta, C.byref(nk))
    order = {0: "for", 1: "rev", 2: "mid"}
    kmers = [[km.raw[t * k:(t + 1) * k].decode(), meta[4 * t], meta[4 * t + 1], meta[4 * t + 2], order[meta[4 * t + 3]]] for t in range(nk.value)]
    return {"match": bool(m), "seq": seq.raw[:ln.value].decode(), "io": list(io[:cl.value]), "ot": list(ot[:cl.value]), "kmers": kmers}


class OPsl(C.Structure):
    _fields_ = [("matches", C.c_int32), ("mismatches", C.c_int32), ("rep_matches", C.c_int32), ("n_count", C.c_int32),
                ("q_num_insert", C.c_int32), ("q_base_insert", C.c_int32), ("t_num_insert", C.c_int32), ("t_base_insert", C.c_int32),
                ("strand", C.c_int32), ("q_size", C.c_int32), ("q_start", C.c_int32), ("q_end", C.c_int32),
                ("t_index", C.c_int32), ("t_size", C.c_int32), ("t_start", C.c_int32), ("t_end", C.c_int32),
                ("block_count", C.c_int32), ("block_sizes", C.c_int32 * 512), ("q_starts", C.c_int32 * 512),
                ("t_starts", C.c_int32 * 512), ("score", C.c_int32)]


def psl_to_dict(r):
    n = r.block_count
    return {"matches": r.matches, "mismatches": r.mismatches, "rep_matches": r.rep_matches, "n_count": r.n_count,
            "q_num_insert": r.q_num_insert, "q_base_insert": r.q_base_insert, "t_num_insert": r.t_num_insert,
            "t_base_insert": r.t_base_insert, "strand": chr(r.strand), "q_size": r.q_size, "q_start": r.q_start,
            "q_end": r.q_end, "t_index": r.t_index, "t_size": r.t_size, "t_start": r.t_start, "t_end": r.t_end,
            "block_sizes": list(r.block_sizes[:n]), "q_starts": list(r.q_starts[:n]), "t_starts": list(r.t_starts[:n]),
            "score": r.score}


def realign(contig, targets, min_score=20, min_seg=20):
    """R2 contract (bk_oracle.h): contig vs [target window, partner windows...] -> PSL-equivalent dicts."""
    L = lib()
    L.bko_realign.restype = C.c_int
    tb = [t.encode() for t in targets]
    arr = (C.c_char_p * len(tb))(*tb)
    tl = (C.c_int * len(tb))(*[len(t) for t in tb])
    cap = 64
    while True:
        out = (OPsl * cap)()
        n = L.bko_realign(contig.encode(), len(contig), arr, tl, len(tb), min_score, min_seg, out, cap)
        if n < 0:
            raise RuntimeError("bko_realign: a chained record needs more than 512 blocks")
        if n <= cap:
            return [psl_to_dict(out[i]) for i in range(n)]
        cap = n
