"""Thin ctypes binding of libbreakmer_hip.so (include/breakmer_hip.h).

There is no CPU fallback: importing is harmless, but creating an engine without the built
library or without a gfx950 GPU raises.  The library is built in-tree by
`__graft_entry__.build()` / `breakmer_amd.build.build_hip()`.
"""
from __future__ import annotations

import ctypes as C
import struct
import os

import numpy as np

# Several handles (HIP streams) per process next to torch/RCCL: with the runtime's default of 4 hardware queues their
# batches serialise (bench.py measured 121 k / 158 k / 202 k regions/s at 4 / 8 / 12+ queues).  Only effective if the HIP
# runtime has not initialised yet; a value from the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbreakmer_hip.so")      # the product build; diagnostic builds (build.VARIANTS) are named explicitly: load_library(path)

BK_STAGE_KMER, BK_STAGE_ASSEMBLE, BK_STAGE_REALIGN, BK_STAGE_ALL = 1, 2, 4, 7
BK_MAX_BLOCKS = 32
BK_ABI_VERSION = 5
BK_E_ARG = -1
# bk_config.flags (include/breakmer_hip.h: BK_CFG_*); production leaves 0
BK_CFG_NO_SPLIT, BK_CFG_TEST_SPLIT_ALWAYS, BK_CFG_TEST_FULL_CALLER, BK_CFG_TEST_HOST_REPAIR, BK_CFG_TEST_PREQUEUE_UNITS = 128, 256, 2048, 4096, 16384
BK_CFG_DIAG_FORCE_REDO = 32768      # diagnostic builds only
BK_PSL_FLAT_HEAD = 18
BK_W_REGIONS_FAILED = 1


class BkConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("kmer_size", C.c_int32), ("rc_thresh", C.c_int32),
                ("max_contig_len", C.c_int32), ("max_read_len", C.c_int32), ("max_candidates", C.c_int32),
                ("arena_bytes", C.c_int64), ("sw_min_score", C.c_int32), ("out_kbytes", C.c_int32),
                ("flags", C.c_uint32), ("asm_wg_threads", C.c_int32), ("no_escalation", C.c_int32), ("submit_threads", C.c_int32), ("reserved", C.c_int32 * 2)]


class BkRegion(C.Structure):
    _fields_ = [("reads", C.c_void_p), ("read_lens", C.c_void_p), ("indel_only", C.c_void_p),
                ("n_reads", C.c_int32), ("read_stride", C.c_int32),
                ("sc_seqs", C.c_void_p), ("sc_lens", C.c_void_p), ("n_sc", C.c_int32), ("sc_stride", C.c_int32),
                ("window", C.c_char_p), ("window_len", C.c_int32), ("n_partners", C.c_int32),
                ("partners", C.POINTER(C.c_char_p)), ("partner_lens", C.c_void_p), ("read_n", C.c_void_p), ("n_read_n", C.c_int32)]


class BkContigInfo(C.Structure):
    _fields_ = [("seq_len", C.c_int32), ("counts_len", C.c_int32), ("n_kmers", C.c_int32), ("n_reads", C.c_int32),
                ("total_reads", C.c_int32), ("n_hits", C.c_int32)]


class BkPsl(C.Structure):
    _fields_ = [("matches", C.c_int32), ("mismatches", C.c_int32), ("rep_matches", C.c_int32), ("n_count", C.c_int32),
                ("q_num_insert", C.c_int32), ("q_base_insert", C.c_int32), ("t_num_insert", C.c_int32), ("t_base_insert", C.c_int32),
                ("strand", C.c_int32), ("q_size", C.c_int32), ("q_start", C.c_int32), ("q_end", C.c_int32),
                ("t_index", C.c_int32), ("t_size", C.c_int32), ("t_start", C.c_int32), ("t_end", C.c_int32),
                ("block_count", C.c_int32), ("block_sizes", C.c_int32 * BK_MAX_BLOCKS), ("q_starts", C.c_int32 * BK_MAX_BLOCKS),
                ("t_starts", C.c_int32 * BK_MAX_BLOCKS), ("score", C.c_int32)]


def call_text(text):
    """Native call tail on ONE fully described contig (CPU only; used by the G5 parity test)."""
    L = load_library()
    out = C.create_string_buffer(1 << 16)
    hit = C.c_int(-1)
    rc = L.bk_call_text(text.encode(), out, len(out), C.byref(hit))
    if rc != 0:
        raise BreakmerHipError("bk_call_text failed (%d): %s" % (rc, L.bk_last_error(None).decode()))
    row = out.value.decode()
    return (row.split("\t") if row else None), (None if hit.value < 0 else bool(hit.value))


EXPORTS = ["bk_create", "bk_destroy", "bk_last_error", "bk_abi_version", "bk_submit_regions", "bk_submit_regions_ex", "bk_run", "bk_sync", "bk_fetch",
           "bk_last_kernel_ms", "bk_get_region_status", "bk_get_kmer_count", "bk_get_kmers", "bk_get_contig_count", "bk_get_contig_info",
           "bk_get_contig", "bk_get_hits", "bk_get_stat", "bk_call_text", "bk_set_call_context", "bk_call", "bk_call_async", "bk_get_calls", "bk_get_contig_counts",
           "bk_nw_batch", "bk_pack_sequence", "bk_trim", "bk_get_hits_flat", "bk_index_create", "bk_index_probe", "bk_index_destroy", "bk_index_set_loci", "bk_index_find", "bk_last_error_region"]

_lib = None


class BreakmerHipError(RuntimeError):
    """code: the BK_E_* value the library returned; region: the region a failed submit names (bk_last_error_region), else -1"""
    code = 0
    region = -1


def load_library(path=None):
    """dlopen the in-tree library; raises if it has not been built (no fallback).  `path`: another build of the library
    (a diagnostic variant of build.VARIANTS) -- named by tests / tools BEFORE the first engine is made; the product never passes it
    and reads no environment variable to choose one."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != _lib._name:
            raise BreakmerHipError("load_library(%s): %s is loaded already" % (path, _lib._name))
        return _lib
    lp = os.path.abspath(path) if path else LIB_PATH
    if not os.path.isfile(lp):
        raise BreakmerHipError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % lp)
    L = C.CDLL(lp)
    L.bk_last_error.restype = C.c_char_p
    L.bk_last_error.argtypes = [C.c_void_p]
    L.bk_last_error_region.restype = C.c_int32
    L.bk_last_error_region.argtypes = [C.c_void_p]
    L.bk_create.argtypes = [C.c_int, C.POINTER(BkConfig), C.POINTER(C.c_void_p)]
    L.bk_destroy.argtypes = [C.c_void_p]
    L.bk_submit_regions.argtypes = [C.c_void_p, C.POINTER(BkRegion), C.c_int32]
    L.bk_submit_regions_ex.argtypes = [C.c_void_p, C.POINTER(BkRegion), C.c_int32, C.c_uint32]
    L.bk_run.argtypes = [C.c_void_p, C.c_uint32]
    L.bk_sync.argtypes = [C.c_void_p]
    L.bk_fetch.argtypes = [C.c_void_p]
    L.bk_last_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.bk_get_region_status.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]
    L.bk_get_kmer_count.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.bk_get_kmers.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    L.bk_get_contig_count.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.bk_get_contig_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.bk_get_contig_info.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(BkContigInfo)]
    L.bk_get_contig.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6
    L.bk_get_hits.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(BkPsl), C.c_int32]
    L.bk_get_stat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.bk_get_hits_flat.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bk_index_create.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.bk_index_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.bk_index_destroy.argtypes = [C.c_void_p]
    L.bk_index_set_loci.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.bk_index_find.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.bk_call_text.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.bk_pack_sequence.argtypes = [C.c_char_p, C.c_int32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.bk_trim.argtypes = [C.c_void_p, C.c_uint64]
    L.bk_set_call_context.argtypes = [C.c_void_p, C.c_char_p]
    L.bk_call.argtypes = [C.c_void_p]
    L.bk_call_async.argtypes = [C.c_void_p]
    L.bk_get_calls.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bk_nw_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_float)]
    _lib = L
    return L


def _ascii_matrix(seqs):
    """list[str] | uint8 code matrix (+lens) -> (ASCII uint8 matrix, lens uint16)."""
    n = len(seqs)
    stride = max([len(s) for s in seqs] + [1])
    m = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, s in enumerate(seqs):
        b = np.frombuffer(s.encode(), dtype=np.uint8)
        m[i, :len(b)] = b
        lens[i] = len(b)
    return m, lens


_ACGT = np.frombuffer(b"ACGTN", dtype=np.uint8)       # code 4 = N
_CODE2ASCII = bytes(b"ACGTN"[i] if i < 5 else ord("?") for i in range(256))


def _as_c(a, dtype):
    """`a` as a C-contiguous array of `dtype`; the array itself when it already is one (the per-target fast path)"""
    if type(a) is np.ndarray and a.dtype == dtype and a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a, dtype=dtype)



class RegionInput(object):
    """Host-side view of one target region, kept alive until submit returns."""

    def __init__(self, reads, window, *, read_lens=None, indel_only=None, sc_seqs=None, partners=(), packed=None):
        """packed: (words uint32 [N, W], read_lens, n_list uint32) as pack_reads() makes them -- handed over as they are
        (BK_SUBMIT_PACKED): a quarter of the bytes of a code matrix and no packing on the submit path"""
        self.packed = packed is not None
        self.read_n = None
        self.codes = (not self.packed) and isinstance(reads, np.ndarray)          # uint8 codes 0..3 (4 = N), [N, L]: handed over as they are (bk_submit_regions_ex)
        if self.packed:
            words, plens, nl = packed
            self.reads = _as_c(words, np.uint32)
            self.lens = _as_c(plens, np.uint16)
            self.read_n = _as_c(nl, np.uint32) if nl is not None and len(nl) else None
        elif self.codes:
            self.reads = _as_c(reads, np.uint8) if reads.size else np.zeros((0, 1), dtype=np.uint8)
            self.lens = (_as_c(read_lens, np.uint16) if read_lens is not None
                         else np.full(reads.shape[0], reads.shape[1], dtype=np.uint16))
        else:
            self.reads, self.lens = _ascii_matrix(list(reads))
            self.lens = _as_c(self.lens, np.uint16)
        self.indel_only = None if indel_only is None else _as_c(indel_only, np.uint8)
        self.window = window.encode() if isinstance(window, str) else bytes(_ACGT[np.asarray(window)])
        if sc_seqs is None:
            self.sc, self.sc_lens = None, None
        else:
            self.sc, self.sc_lens = _ascii_matrix(list(sc_seqs))
        self.partners = [pw.encode() if isinstance(pw, str) else bytes(_ACGT[np.asarray(pw)]) for pw in partners]
        self._filled = None                                 # the bk_region of this object, made once (a driver that keeps the object submits it again)

    def fill(self, g: BkRegion):
        """g: an element of a BkRegion array.  The struct is made once and copied: this object holds every buffer it points to."""
        f = self._filled
        if f is None:
            f = BkRegion()
            self._fill(f)
            self._filled = f
        C.memmove(C.addressof(g), C.addressof(f), C.sizeof(BkRegion))

    def _fill(self, g: BkRegion):
        reads = self.reads
        g.reads = reads.ctypes.data
        g.read_lens = self.lens.ctypes.data
        g.indel_only = self.indel_only.ctypes.data if self.indel_only is not None else None
        g.n_reads, g.read_stride = reads.shape
        if self.packed:
            g.read_stride = reads.shape[1] * 4                  # bytes between rows
            g.read_n = self.read_n.ctypes.data if self.read_n is not None else None
            g.n_read_n = 0 if self.read_n is None else len(self.read_n)
        if self.sc is None:
            g.sc_seqs, g.sc_lens, g.n_sc, g.sc_stride = None, None, -1, 0
        else:
            g.sc_seqs, g.sc_lens, g.n_sc, g.sc_stride = self.sc.ctypes.data, self.sc_lens.ctypes.data, self.sc.shape[0], self.sc.shape[1]
        g.window = self.window
        g.window_len = len(self.window)
        g.n_partners = len(self.partners)
        if self.partners:
            self._parr = (C.c_char_p * len(self.partners))(*self.partners)
            self._plens = np.array([len(x) for x in self.partners], dtype=np.int32)
            g.partners = self._parr
            g.partner_lens = self._plens.ctypes.data
        else:                                               # the common case: nothing to build per target (never dereferenced)
            g.partners, g.partner_lens = None, None


class KmerStrings(object):
    """The k-mers of one contig as a read-only sequence of str over the bytes the library returned; a string is only made
    when an element is asked for (the driver's per-target objects rarely look at them)."""
    __slots__ = ("_b", "_k", "_n")

    def __init__(self, raw, k, n):
        self._b, self._k, self._n = raw, k, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._b[i * self._k:(i + 1) * self._k].decode()

    def __iter__(self):
        b, k = self._b, self._k
        return (b[i * k:(i + 1) * k].decode() for i in range(self._n))

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))


class Engine(object):
    """One handle = one GPU + one stream (bk_create ... bk_destroy)."""

    def __init__(self, kmer_size, rc_thresh=2, device=0, **limits):
        self.L = load_library()
        cfg = BkConfig()
        cfg.abi_version = BK_ABI_VERSION
        cfg.kmer_size = int(kmer_size)
        cfg.rc_thresh = int(rc_thresh)
        cfg.max_contig_len = int(limits.get("max_contig_len", 0))
        cfg.max_read_len = int(limits.get("max_read_len", 0))
        cfg.max_candidates = int(limits.get("max_candidates", 0))
        cfg.arena_bytes = int(limits.get("arena_bytes", 0))
        cfg.out_kbytes = int(limits.get("out_kbytes", 0))
        cfg.sw_min_score = int(limits.get("sw_min_score", 0))
        cfg.flags = int(limits.get("flags", 0))                     # BK_CFG_* (0 in production)
        cfg.asm_wg_threads = int(limits.get("wg_threads", 0))       # assembler workgroup size: 0 = library default, 256 (throughput) or 512 (latency)
        cfg.no_escalation = int(limits.get("no_escalation", 0))     # 1: regions that overflow an assembler cap fail at once instead of being re-run under larger caps
        cfg.submit_threads = int(limits.get("submit_threads", 0))   # host threads filling the staging buffer of a submit (0: library default)
        self.k = int(kmer_size)
        self.rc_thresh, self.device = int(rc_thresh), int(device)
        self._inputs = None
        self.h = C.c_void_p()
        self.batch_serial = 0
        rc = self.L.bk_create(int(device), C.byref(cfg), C.byref(self.h))
        if rc != 0:
            ex = BreakmerHipError("bk_create failed (%d): %s" % (rc, self.L.bk_last_error(None).decode()))
            ex.code = rc
            raise ex
        self.n_regions = 0
        self.n_failed = 0

    def _chk(self, rc, what):
        if rc != 0:
            ex = BreakmerHipError("%s failed (%d): %s" % (what, rc, self.L.bk_last_error(self.h).decode()))
            ex.code, ex.region = rc, int(self.L.bk_last_error_region(self.h))
            raise ex

    def close(self):
        if self.h:
            self.L.bk_destroy(self.h)
            self.h = C.c_void_p()
            self._inputs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, regions, wait=True):
        """wait=False: the library packs and copies on its own thread (BK_SUBMIT_ASYNC); the inputs are kept alive here and
        any error of the submit is raised by the next call on this engine."""
        arr = (BkRegion * len(regions))()
        for g, r in zip(arr, regions):
            r.fill(g)
        pk = {bool(r.packed) for r in regions if r.reads.shape[0]}
        if pk == {True}:
            self._chk(self.L.bk_submit_regions_ex(self.h, arr, len(regions), 4 | (0 if wait else 2)), "bk_submit_regions")
            self._inputs = None if wait else (arr, regions)
            self.n_regions = len(regions)
            self.batch_serial += 1
            return
        if True in pk:
            raise BreakmerHipError("a batch is either all packed reads (RegionInput(packed=...)) or none")
        kinds = {bool(r.codes) for r in regions if r.reads.shape[0]}
        if len(kinds) > 1:                                  # mixed batch: the code matrices become ASCII (one C-speed byte translation each)
            for g, r in zip(arr, regions):
                if r.codes:
                    r.reads = np.frombuffer(r.reads.tobytes().translate(_CODE2ASCII), dtype=np.uint8).reshape(r.reads.shape)
                    r.codes = False
                    r._filled = None
                    r.fill(g)
            kinds = {False}
        flags = (1 if kinds == {True} else 0) | (0 if wait else 2)
        self._chk(self.L.bk_submit_regions_ex(self.h, arr, len(regions), flags), "bk_submit_regions")
        # the sequences must outlive the library's packing thread (the previous batch's were still referenced during the call
        # above, which waits for an unfinished earlier submit)
        self._inputs = None if wait else (arr, regions)
        self.n_regions = len(regions)
        self.batch_serial += 1                               # lazily read results (sv_assembly.LazyContigs) belong to one batch

    def submit_packed(self, items, wait=False):
        """a batch of (PackedReads, window bytes, indel_only uint8 array or None) -- see region_array_packed; otherwise like
        submit(..., wait) of RegionInput(packed=...) objects"""
        arr, wbuf = region_array_packed(items)
        self._chk(self.L.bk_submit_regions_ex(self.h, arr, len(items), 4 | (0 if wait else 2)), "bk_submit_regions")
        self._inputs = None if wait else (arr, wbuf, items)
        self.n_regions = len(items)
        self.batch_serial += 1

    def contig_counts(self):
        """the number of contigs of every region of the batch (one library call)"""
        n = np.zeros(max(self.n_regions, 1), dtype=np.int32)
        self._chk(self.L.bk_get_contig_counts(self.h, n.ctypes.data, len(n)), "bk_get_contig_counts")
        return n[:self.n_regions].tolist()

    def run(self, stages=BK_STAGE_KMER | BK_STAGE_ASSEMBLE, sync=True):
        self._chk(self.L.bk_run(self.h, stages), "bk_run")
        if sync:
            self.sync()

    def sync(self):
        """waits for the run; returns the number of regions that hit a device cap (they report no contigs; region_status()
        names the cap) -- 0 in the normal case"""
        rc = self.L.bk_sync(self.h)
        if rc == BK_W_REGIONS_FAILED:
            self.n_failed = int(self.stat(22))
            return self.n_failed
        self._chk(rc, "bk_sync")
        self.n_failed = 0
        return 0

    def trim(self, keep_bytes=1 << 30):
        """free the device / pinned buffers larger than keep_bytes (they are sized anew by the next submit)"""
        self._chk(self.L.bk_trim(self.h, int(keep_bytes)), "bk_trim")
        self._inputs = None

    def fetch(self):
        """Wait for the last run and copy its records to the host; the handle may be run again before call()."""
        self._chk(self.L.bk_fetch(self.h), "bk_fetch")

    def kernel_ms(self, which=0):
        ms = C.c_float()
        self._chk(self.L.bk_last_kernel_ms(self.h, which, C.byref(ms)), "bk_last_kernel_ms")
        return ms.value

    def stat(self, which):
        v = C.c_uint64()
        self._chk(self.L.bk_get_stat(self.h, which, C.byref(v)), "bk_get_stat")
        return v.value

    def region_status(self, region):
        """(status, text) of one region of the last run: 0/'ok', or the device cap it hit (it then has no contigs)."""
        st, tx = C.c_int32(), C.c_char_p()
        self._chk(self.L.bk_get_region_status(self.h, region, C.byref(st), C.byref(tx)), "bk_get_region_status")
        return st.value, (tx.value or b"").decode()

    def kmers(self, region):
        n, u = C.c_int32(), C.c_int32()
        self._chk(self.L.bk_get_kmer_count(self.h, region, C.byref(n), C.byref(u)), "bk_get_kmer_count")
        mers = np.zeros(max(n.value * self.k, 1), dtype=np.uint8)
        cnt = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self.L.bk_get_kmers(self.h, region, mers.ctypes.data, cnt.ctypes.data, n.value), "bk_get_kmers")
        return [mers[i * self.k:(i + 1) * self.k].tobytes().decode() for i in range(n.value)], cnt[:n.value].copy(), u.value

    def contig_count(self, region):
        n = C.c_int32()
        self._chk(self.L.bk_get_contig_count(self.h, region, C.byref(n)), "bk_get_contig_count")
        return n.value

    def contigs(self, region, lazy_kmers=False):
        """records of the contigs of one region; lazy_kmers: the 'kmers' entry is a KmerStrings view instead of a list"""
        n = C.c_int32()
        self._chk(self.L.bk_get_contig_count(self.h, region, C.byref(n)), "bk_get_contig_count")
        out = []
        for ci in range(n.value):
            info = BkContigInfo()
            self._chk(self.L.bk_get_contig_info(self.h, region, ci, C.byref(info)), "bk_get_contig_info")
            seq = np.zeros(max(info.seq_len, 1), dtype=np.uint8)
            io = np.zeros(max(info.counts_len, 1), dtype=np.int32)
            ot = np.zeros(max(info.counts_len, 1), dtype=np.int32)
            kl = np.zeros(max(info.seq_len, 1), dtype=np.int32)
            km = np.zeros(max(info.n_kmers * self.k, 1), dtype=np.uint8)
            rd = np.zeros(max(info.n_reads, 1), dtype=np.int32)
            self._chk(self.L.bk_get_contig(self.h, region, ci, seq.ctypes.data, io.ctypes.data, ot.ctypes.data, kl.ctypes.data,
                                           km.ctypes.data, rd.ctypes.data), "bk_get_contig")
            out.append({"seq": seq[:info.seq_len].tobytes().decode(), "indel_only": io[:info.counts_len].tolist(),
                        "others": ot[:info.counts_len].tolist(), "kmer_locs": kl[:info.seq_len].tolist(),
                        "kmers": (KmerStrings(km[:info.n_kmers * self.k].tobytes(), self.k, info.n_kmers) if lazy_kmers
                                  else [km[i * self.k:(i + 1) * self.k].tobytes().decode() for i in range(info.n_kmers)]),
                        "reads": rd[:info.n_reads].tolist(), "total_reads": info.total_reads, "n_hits": info.n_hits})
        return out

    def set_call_context(self, text):
        self._chk(self.L.bk_set_call_context(self.h, text.encode()), "bk_set_call_context")

    def call_async(self):
        """start the wait for the run + the native SV-call tail on the library's thread; call() / call_blob() pick the result up"""
        self._chk(self.L.bk_call_async(self.h), "bk_call_async")

    def call_blob(self):
        """Native SV-call tail over every contig of the batch -> the serialised records (bytes; one line per call:
        region, contig number, then the 13 result fields, tab separated) as they are collated across ranks."""
        self._chk(self.L.bk_call(self.h), "bk_call")
        need = C.c_size_t()
        self.L.bk_get_calls(self.h, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        self._chk(self.L.bk_get_calls(self.h, buf, need.value, C.byref(need)), "bk_get_calls")
        return buf.value

    def call(self):
        """Native SV-call tail over every contig of the batch -> {region: [13-field rows]} (contig order)."""
        out = {}
        for ln in self.call_blob().decode().split("\n"):
            if ln:
                f = ln.split("\t")
                out.setdefault(int(f[0]), []).append(f[2:])
        return out

    def hits(self, region, contig):
        """PSL-equivalent records of one contig (realign stage) as dicts (any number of blocks: bk_get_hits_flat)."""
        need = C.c_size_t()
        n = self.L.bk_get_hits_flat(self.h, region, contig, None, 0, C.byref(need))
        if n < 0:
            self._chk(n, "bk_get_hits_flat")
        buf = np.zeros(max(need.value, 1), dtype=np.int32)
        if n > 0:
            n = self.L.bk_get_hits_flat(self.h, region, contig, buf.ctypes.data, buf.size, C.byref(need))
            if n < 0:
                self._chk(n, "bk_get_hits_flat")
        out, o = [], 0
        v = buf.tolist()
        for _ in range(n):
            hd = v[o:o + BK_PSL_FLAT_HEAD]
            nb = hd[17]
            o += BK_PSL_FLAT_HEAD
            out.append({"matches": hd[0], "mismatches": hd[1], "rep_matches": hd[2], "n_count": hd[3],
                        "q_num_insert": hd[4], "q_base_insert": hd[5], "t_num_insert": hd[6], "t_base_insert": hd[7],
                        "strand": chr(hd[8]), "q_size": hd[9], "q_start": hd[10], "q_end": hd[11], "t_index": hd[12], "t_size": hd[13],
                        "t_start": hd[14], "t_end": hd[15], "block_sizes": v[o:o + nb], "q_starts": v[o + nb:o + 2 * nb],
                        "t_starts": v[o + 2 * nb:o + 3 * nb], "score": hd[16]})
            o += 3 * nb
        return out

    def hits_fixed(self, region, contig):
        """the same through bk_get_hits (bk_psl: at most BK_MAX_BLOCKS blocks per record; raises BK_E_LIMIT beyond)"""
        cap = 16
        while True:
            arr = (BkPsl * cap)()
            n = self.L.bk_get_hits(self.h, region, contig, arr, cap)
            if n < 0:
                self._chk(n, "bk_get_hits")
            if n <= cap:
                break
            cap = n
        out = []
        for r in arr[:n]:
            nb = r.block_count
            out.append({"matches": r.matches, "mismatches": r.mismatches, "rep_matches": r.rep_matches, "n_count": r.n_count,
                        "q_num_insert": r.q_num_insert, "q_base_insert": r.q_base_insert, "t_num_insert": r.t_num_insert,
                        "t_base_insert": r.t_base_insert, "strand": chr(r.strand), "q_size": r.q_size, "q_start": r.q_start,
                        "q_end": r.q_end, "t_index": r.t_index, "t_size": r.t_size, "t_start": r.t_start, "t_end": r.t_end,
                        "block_sizes": list(r.block_sizes[:nb]), "q_starts": list(r.q_starts[:nb]), "t_starts": list(r.t_starts[:nb]),
                        "score": r.score})
        return out

    def nw_batch(self, pairs, reps=1, transposed=False):
        """olc.nw on (seq1, seq2) pairs -> int32 [n,4] (j_start, i_end, i_start, score), kernel ms.
        transposed: False = direct sweep, True = the transposed sweep the assembler uses for nw(read, contig),
        2 = the direct sweep restricted to the last 1.5 len(seq2) + 2 columns of seq1 (what the assembler runs for nw(contig, read))."""
        blob = bytearray()
        o1, l1, o2, l2 = [], [], [], []
        for a, b in pairs:
            o1.append(len(blob)); l1.append(len(a)); blob += a.encode()
            o2.append(len(blob)); l2.append(len(b)); blob += b.encode()
        o1, l1, o2, l2 = (np.array(x, dtype=np.uint32) for x in (o1, l1, o2, l2))
        out = np.zeros((len(pairs), 4), dtype=np.int32)
        ms = C.c_float()
        self._chk(self.L.bk_nw_batch(self.h, bytes(blob), len(blob), o1.ctypes.data, l1.ctypes.data, o2.ctypes.data, l2.ctypes.data,
                                     len(pairs), reps, int(transposed), out.ctypes.data, C.byref(ms)), "bk_nw_batch")
        return out, ms.value


# ---- handles kept between driver runs of one process ---------------------------------------------------------------------
# Creating and destroying a handle costs ~35 ms (stream, pinned staging, device buffers sized by the first batch); a process
# that runs the driver repeatedly (one sample after the other) keeps up to four per (device, k, rc_thresh) -- of ONE such
# key at a time (a run with another k closes the others), and no pooled handle keeps a buffer above _POOL_KEEP_BYTES (the
# scratch arena of a heavy batch can reach tens of GB: it is given back and re-sized by the next batch).
_POOL = {}
_POOL_MAX = 4                                   # the driver has up to four batches on their way (sv_processor.runner: two submitted, one launched, one being read)
_POOL_KEEP_BYTES = 2 << 30


def acquire_engine(kmer_size, rc_thresh=2, device=0, flags=0, wg_threads=0):
    key = (device, int(kmer_size), int(rc_thresh), int(flags), int(wg_threads))
    for other in [k_ for k_ in _POOL if k_ != key]:
        for e in _POOL.pop(other):
            e.close()
    lst = _POOL.get(key)
    if lst:
        return lst.pop()
    eng = Engine(kmer_size=kmer_size, rc_thresh=rc_thresh, device=device, flags=flags, wg_threads=wg_threads)
    eng._pool_key = key
    return eng


def release_engine(eng):
    key = getattr(eng, "_pool_key", (eng.device, eng.k, eng.rc_thresh, 0, 0))
    lst = _POOL.setdefault(key, [])
    if eng.h and len(lst) < _POOL_MAX:
        eng._inputs = None                      # nothing of the last batch is pending on a handle the driver gives back
        try:
            eng.trim(_POOL_KEEP_BYTES)
        except BreakmerHipError:
            eng.close()
            return
        lst.append(eng)
    else:
        eng.close()


def close_pool():
    for lst in _POOL.values():
        for e in lst:
            e.close()
    _POOL.clear()


import atexit  # noqa: E402
atexit.register(close_pool)


class DeviceIndex(object):
    """sorted sampled k-mer codes of a genome resident in HBM; probe(queries) -> (lo, hi) index ranges by binary search on the
    device (bk_index_*; the host's numpy.searchsorted over the same array is what it is pinned against)"""

    def __init__(self, sorted_codes, device=0):
        self.L = load_library()
        codes = _as_c(sorted_codes, np.uint32)
        self.h = C.c_void_p()
        self.n = len(codes)
        rc = self.L.bk_index_create(int(device), codes.ctypes.data, len(codes), C.byref(self.h))
        if rc != 0:
            raise BreakmerHipError("bk_index_create failed (%d): %s" % (rc, self.L.bk_last_error(None).decode()))
        self.kernel_ms = 0.0

    def probe(self, queries):
        q = _as_c(queries, np.uint32)
        lo = np.zeros(len(q), dtype=np.uint32)
        hi = np.zeros(len(q), dtype=np.uint32)
        ms = C.c_float()
        rc = self.L.bk_index_probe(self.h, q.ctypes.data, len(q), lo.ctypes.data, hi.ctypes.data, C.byref(ms))
        if rc != 0:
            raise BreakmerHipError("bk_index_probe failed (%d)" % rc)
        self.kernel_ms = ms.value
        return lo, hi

    def set_loci(self, seqno, pos):
        """sequence number (uint16) and position (uint32) of every index entry, parallel to the sorted codes: what find() needs"""
        sq, ps = _as_c(seqno, np.uint16), _as_c(pos, np.uint32)
        if len(sq) != self.n or len(ps) != self.n:
            raise BreakmerHipError("DeviceIndex.set_loci: arrays must be parallel to the codes")
        rc = self.L.bk_index_set_loci(self.h, sq.ctypes.data, ps.ctypes.data)
        if rc != 0:
            raise BreakmerHipError("bk_index_set_loci failed (%d)" % rc)

    def find(self, codes, ok, max_occ=64, band=32, min_hits=2):
        """loci of one strand of a query on the device: [(hits, sequence number, first index position, last index position)] in
        (sequence, diagonal, position) order (bk_index_find)"""
        q = _as_c(codes, np.uint32)
        okb = _as_c(np.asarray(ok).astype(np.uint8), np.uint8)
        cap = 256
        while True:
            out = np.zeros((cap, 4), dtype=np.uint32)
            n, ms = C.c_uint32(), C.c_float()
            rc = self.L.bk_index_find(self.h, q.ctypes.data, okb.ctypes.data, len(q), int(max_occ), int(band), int(min_hits), out.ctypes.data, cap, C.byref(n), C.byref(ms))
            if rc != 0:
                raise BreakmerHipError("bk_index_find failed (%d)" % rc)
            self.kernel_ms = ms.value
            if n.value <= cap:
                return [tuple(int(v) for v in row) for row in out[:n.value]]
            cap = int(n.value)

    def close(self):
        if self.h:
            self.L.bk_index_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackedReads(tuple):
    """(words, lens, N list) as pack_reads makes them, with what a submit needs of them looked up ONCE (buffer addresses, sizes, the
    longest read): a driver hands thousands of these over per second, and asking numpy for an address costs more than the struct"""

    def __new__(cls, words, lens, nl):
        words, lens = _as_c(words, np.uint32), _as_c(lens, np.uint16)
        nl = _as_c(nl, np.uint32) if nl is not None and len(nl) else None
        self = tuple.__new__(cls, (words, lens, nl))
        self.n_reads, self.stride = int(words.shape[0]), int(words.shape[1]) * 4
        self.p_words, self.p_lens = words.ctypes.data, lens.ctypes.data
        self.p_n, self.n_n = (nl.ctypes.data, len(nl)) if nl is not None else (0, 0)
        self.maxlen = int(lens.max()) if len(lens) else 0
        return self


# bk_region as bytes (the ctypes structure above, field for field; C.sizeof(BkRegion) with its tail padding)
_REGION_PACK = struct.Struct("@PPPiiPPiiPiiPPPi%dx" % (C.sizeof(BkRegion) - struct.calcsize("@PPPiiPPiiPiiPPPi"))).pack


def region_array_packed(items):
    """BkRegion array + what it points to, for a batch of (PackedReads, window bytes, indel_only uint8 array or None): one
    struct.pack per target instead of a ctypes structure filled field by field (no soft-clip sequences, no partner windows:
    targets that have those go through RegionInput)"""
    wbuf = b"".join(w for _p, w, _io in items)
    base = C.cast(C.c_char_p(wbuf), C.c_void_p).value or 0
    parts, off, pack = [], 0, _REGION_PACK
    for pr, w, io in items:
        parts.append(pack(pr.p_words, pr.p_lens, io.ctypes.data if io is not None else 0, pr.n_reads, pr.stride, 0, 0, -1, 0, base + off, len(w), 0, 0, 0, pr.p_n, pr.n_n))
        off += len(w)
    return (BkRegion * len(items)).from_buffer_copy(b"".join(parts)), wbuf


def pack_reads(codes, lens=None):
    """uint8 code matrix [N, L] (0..3, 4 = N) -> (uint32 words [N, ceil(L/16)], uint16 lens, uint32 N list) in the layout
    bk_submit_regions_ex(BK_SUBMIT_PACKED) takes: 16 bases per word, the first in the most significant bits, an N packed as A,
    bases beyond a read's length zero.  Vectorised numpy (what a read extraction that packs as it goes would hand over)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, L = codes.shape
    lens = np.full(n, L, dtype=np.uint16) if lens is None else np.ascontiguousarray(lens, dtype=np.uint16)
    W = (L + 15) // 16
    pad = np.zeros((n, W * 16), dtype=np.uint8)
    pad[:, :L] = codes
    pad[np.arange(W * 16)[None, :] >= lens[:, None]] = 0
    isn = pad == 4
    rr, pp = np.nonzero(isn)
    nl = ((rr.astype(np.uint32) << np.uint32(10)) | pp.astype(np.uint32)).astype(np.uint32)      # row-major: ascending
    pad[isn] = 0
    v = pad.reshape(n, W, 16).astype(np.uint32)
    sh = (30 - 2 * np.arange(16, dtype=np.uint32))[None, None, :]
    words = np.bitwise_or.reduce(v << sh, axis=2).astype(np.uint32)
    return PackedReads(words, lens, nl)


def pack_sequence(seq, codes=False):
    """(words, N positions) of one sequence as bk_submit_regions packs it (host code of the library, no GPU); raises on any
    character other than A/C/G/T/N."""
    L = load_library()
    raw = seq if isinstance(seq, (bytes, bytearray)) else (bytes(bytearray(seq)) if codes else seq.encode())
    n = len(raw)
    words = np.zeros((n + 15) // 16 + 1, dtype=np.uint32)
    npos = np.zeros(max(n, 1), dtype=np.uint32)
    nn = C.c_int32()
    rc = L.bk_pack_sequence(raw, n, 1 if codes else 0, words.ctypes.data, len(words), npos.ctypes.data, len(npos), C.byref(nn))
    if rc != 0:
        raise BreakmerHipError("bk_pack_sequence failed (%d)" % rc)
    return words, npos[:nn.value].tolist()
