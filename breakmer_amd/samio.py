"""Minimal SAM/BAM reader with the attribute names the reference uses from pysam's AlignedRead / Samfile
(sv_processor.py:12-93, 422-540): no pysam in this image.  BAM is streamed through gzip (BGZF is a multi-member gzip
stream) record by record; no index is used.  With `regions` (the [start-200, end+200) windows of the run's targets) only the
records that can matter are kept: those overlapping a window and, found in a second streaming pass, the mates of kept
records that lie elsewhere (bam.mate() for the discordant-pair evidence) -- a multi-GB panel BAM is then neither held in
memory nor turned into one Python object per record."""
from __future__ import annotations

import copy
import gzip
import struct

CIGAR_OPS = "MIDNSHP=X"
_SEQ16 = "=ACMGRSVTWYHKDBN"


class AlignedRead(object):
    def __init__(self, qname, flag, tid, pos, mapq, cigar, rnext, pnext, tlen, seq, qual):
        self.qname, self.flag, self.tid, self.pos, self.mapq = qname, int(flag), tid, int(pos), int(mapq)
        self.cigar = cigar                      # list of (op code, length) or None
        self.rnext, self.pnext, self.tlen = rnext, int(pnext), int(tlen)
        self.seq, self.qual = seq, qual         # qual: phred+33 string
        self.mate_is_unmapped = bool(self.flag & 0x8)

    # flag views
    @property
    def is_duplicate(self): return bool(self.flag & 0x400)
    @property
    def is_qcfail(self): return bool(self.flag & 0x200)
    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def is_reverse(self): return bool(self.flag & 0x10)
    @property
    def mate_is_reverse(self): return bool(self.flag & 0x20)
    @property
    def is_read1(self): return bool(self.flag & 0x40)
    @property
    def is_read2(self): return bool(self.flag & 0x80)
    # pysam aliases
    @property
    def mrnm(self): return self.rnext
    @property
    def mpos(self): return self.pnext
    @property
    def isize(self): return self.tlen

    def ref_end(self):
        n = 0
        for op, ln in (self.cigar or []):
            if op in (0, 2, 3, 7, 8):
                n += ln
        return self.pos + max(n, 1)


def parse_cigar(text):
    if text == "*":
        return None
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((CIGAR_OPS.index(ch), int(num)))
            num = ""
    return out


_NIB = [_SEQ16[b >> 4] + _SEQ16[b & 15] for b in range(256)]           # two bases per packed byte
_QUAL = bytes((33 + (x if x != 255 else 0)) & 255 for x in range(256))


class Samfile(object):
    def __init__(self, fn, mode="r", regions=None, **_kw):
        """regions: optional [(chrom, start, end), ...] (0-based half-open, chrom with or without 'chr'): keep only the
        records overlapping one of them, plus the mates of those records wherever they lie."""
        self.references, self._tid, self.reads = [], {}, []
        self._by_tid = self._mates = None
        self._regions = regions
        self._fn, self._is_bam = fn, False
        if fn is None:
            return
        with open(fn, "rb") as f:
            magic = f.read(2)
        if magic == b"\x1f\x8b":
            self._is_bam = True
            self._read_bam(fn)
        else:
            self._read_sam(fn)

    # ---- region filter ------------------------------------------------------------------------------------------
    def _windows(self):
        """tid -> sorted [(start, end)] of the requested regions (None: keep everything)"""
        if self._regions is None:
            return None
        w = {}
        for chrom, s, e in self._regions:
            c = str(chrom)
            tid = self._tid.get(c, self._tid.get("chr" + c, self._tid.get(c.replace("chr", ""), -2)))
            w.setdefault(tid, []).append((int(s), int(e)))
        for v in w.values():
            v.sort()
        return w

    @staticmethod
    def _hits(wins, tid, pos, end):
        for s, e in (wins.get(tid) or ()):
            if pos < e and end > s:
                return True
        return False

    @classmethod
    def from_records(cls, references, reads):
        s = cls(None)
        s.references = list(references)
        s._tid = {n: i for i, n in enumerate(s.references)}
        s.reads = list(reads)
        return s

    def _read_sam(self, fn):
        with open(fn) as f:
            for ln in f:
                if ln.startswith("@"):
                    if ln.startswith("@SQ"):
                        name = [x[3:] for x in ln.strip().split("\t") if x.startswith("SN:")][0]
                        self._tid[name] = len(self.references)
                        self.references.append(name)
                    continue
                p = ln.rstrip("\n").split("\t")
                if len(p) < 11:
                    continue
                tid = self._tid.get(p[2], -1)
                rnext = tid if p[6] == "=" else self._tid.get(p[6], -1)
                self.reads.append(AlignedRead(p[0], p[1], tid, int(p[3]) - 1, p[4], parse_cigar(p[5]), rnext, int(p[7]) - 1, p[8],
                                              p[9], p[10]))
        wins = self._windows()
        if wins is not None:                             # SAM text is small: filter after the parse, same rule as the BAM stream
            keep = [r for r in self.reads if self._hits(wins, r.tid, r.pos, r.pos + 1 if r.is_unmapped else r.ref_end())]
            names = {r.qname for r in keep}
            kept = set(map(id, keep))
            self.reads = [r for r in self.reads if id(r) in kept or r.qname in names]

    def _read_bam(self, fn):
        wins = None
        for want_names in (None, "mates"):                # pass 1: records in the windows; pass 2 (only with a filter): their mates elsewhere
            if want_names == "mates":
                if wins is None:
                    break
                names = {r.qname for r in self.reads}
                have = {(r.qname, r.flag & 0xC0, r.tid, r.pos) for r in self.reads}
                first_pass = self.reads
                self.reads = []
            with gzip.open(fn, "rb") as f:
                assert f.read(4) == b"BAM\x01", "not a BAM file"
                l_text, = struct.unpack("<i", f.read(4)); f.read(l_text)
                n_ref, = struct.unpack("<i", f.read(4))
                refs, tids = [], {}
                for _ in range(n_ref):
                    l_name, = struct.unpack("<i", f.read(4))
                    name = f.read(l_name)[:-1].decode(); f.read(4)
                    tids[name] = len(refs)
                    refs.append(name)
                if want_names is None:
                    self.references, self._tid = refs, tids
                    wins = self._windows()
                while True:
                    head = f.read(4)
                    if len(head) < 4:
                        break
                    bs, = struct.unpack("<i", head)
                    rec = f.read(bs)
                    tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, rnext, pnext, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
                    q = 32
                    if wins is not None:
                        if want_names is None:
                            end = pos + 1
                            if not (flag & 0x4) and n_cig:
                                n = 0
                                for v in struct.unpack_from("<%dI" % n_cig, rec, q + l_rn):
                                    if (v & 0xF) in (0, 2, 3, 7, 8):
                                        n += v >> 4
                                end = pos + max(n, 1)
                            if not self._hits(wins, tid, pos, end):
                                continue
                        else:
                            qn = rec[q:q + l_rn - 1].decode()
                            if qn not in names or (qn, flag & 0xC0, tid, pos) in have:
                                continue
                    qname = rec[q:q + l_rn - 1].decode(); q += l_rn
                    cig = [(v & 0xF, v >> 4) for v in struct.unpack_from("<%dI" % n_cig, rec, q)] if n_cig else []
                    q += 4 * n_cig
                    nb = (l_seq + 1) // 2
                    seq = "".join([_NIB[b_] for b_ in rec[q:q + nb]])[:l_seq]; q += nb
                    qual = rec[q:q + l_seq].translate(_QUAL).decode("latin-1")
                    self.reads.append(AlignedRead(qname, flag, tid, pos, mapq, cig or None, rnext, pnext, tlen, seq, qual))
            if want_names == "mates":
                self.reads = first_pass + self.reads       # file order within each pass; fetch() orders by position per chromosome

    # pysam.Samfile surface used by the reference
    def getrname(self, tid):
        return self.references[tid]

    def _index(self):
        if self._by_tid is None:
            import numpy as np
            by = {}
            for n, r in enumerate(self.reads):
                by.setdefault(r.tid, []).append(n)
            self._by_tid = {}
            for tid, ids in by.items():
                ids = np.asarray(ids, dtype=np.int64)
                pos = np.asarray([self.reads[i].pos for i in ids], dtype=np.int64)
                end = np.asarray([self.reads[i].pos + 1 if self.reads[i].is_unmapped else self.reads[i].ref_end() for i in ids], dtype=np.int64)
                self._by_tid[tid] = (ids, pos, end)
            self._mates = {}
            for r in self.reads:
                self._mates.setdefault(r.qname, []).append(r)

    def fetch(self, chrom=None, start=None, end=None):
        """Records of `chrom` overlapping [start, end) in file order (unplaced-but-positioned reads count as 1 bp)."""
        if chrom is None:
            return [copy.copy(r) for r in self.reads]
        self._index()
        tid = self._tid.get(str(chrom), self._tid.get("chr" + str(chrom), -2))
        if tid not in self._by_tid:
            return []
        ids, pos, rend = self._by_tid[tid]
        # Copies, not the cached records: the reference re-opens the alignment file per target (sv_processor.py:426), so
        # the in-place quality trimming of fq_line/trim_qual (utils.py:414-443) and the mate_is_unmapped normalisation of
        # process_reads (:16-17) never leak from one target into the next one whose window overlaps it.
        return [copy.copy(self.reads[i]) for i in ids[(pos < end) & (rend > start)]]

    def covered(self, chrom, start, end):
        """True when [start, end) of `chrom` lies inside a window whose records were all kept (always without a filter)"""
        if self._regions is None:
            return True
        wins = self._windows()
        c = str(chrom)
        tid = self._tid.get(c, self._tid.get("chr" + c, self._tid.get(c.replace("chr", ""), -2)))
        return any(s <= start and end <= e for s, e in (wins.get(tid) or ()))

    def count_region(self, chrom, start, end, pred):
        """number of records of the FILE overlapping [start, end) that satisfy pred(flag, mapq) -- one streaming pass, for
        queries outside the kept windows (breakpoint coverage at a translocation partner, sv_caller.py:118-133)"""
        c = str(chrom)
        tid = self._tid.get(c, self._tid.get("chr" + c, self._tid.get(c.replace("chr", ""), -2)))
        n = 0
        if self._fn is None or not self._is_bam:
            return sum(1 for r in self.fetch(chrom, start, end) if pred(r.flag, r.mapq))
        with gzip.open(self._fn, "rb") as f:
            f.read(4)
            l_text, = struct.unpack("<i", f.read(4)); f.read(l_text)
            n_ref, = struct.unpack("<i", f.read(4))
            for _ in range(n_ref):
                l_name, = struct.unpack("<i", f.read(4)); f.read(l_name + 4)
            while True:
                head = f.read(4)
                if len(head) < 4:
                    break
                bs, = struct.unpack("<i", head)
                rec = f.read(bs)
                rtid, pos, l_rn, mapq, _bin, n_cig, flag = struct.unpack_from("<iiBBHHH", rec, 0)
                if rtid != tid or pos >= end:
                    continue
                rend = pos + 1
                if not (flag & 0x4) and n_cig:
                    m = 0
                    for v in struct.unpack_from("<%dI" % n_cig, rec, 32 + l_rn):
                        if (v & 0xF) in (0, 2, 3, 7, 8):
                            m += v >> 4
                    rend = pos + max(m, 1)
                if rend > start and pred(flag, mapq):
                    n += 1
        return n

    def mate(self, read):
        self._index()
        cands = [r for r in self._mates.get(read.qname, ()) if r.is_read1 != read.is_read1]
        for r in cands:                                    # pysam returns the primary record of the mate
            if not (r.flag & 0x900):
                return r
        if cands:
            return cands[0]
        raise ValueError("mate not found for " + read.qname)

    def write(self, _read): pass
    def close(self): pass
