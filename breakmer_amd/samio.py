"""Minimal SAM/BAM reader with the attribute names the reference uses from pysam's AlignedRead / Samfile
(sv_processor.py:12-93, 422-540): no pysam in this image, and only whole-file scans are needed here.
BAM is read through gzip (BGZF is a multi-member gzip stream); no index is used."""
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


class Samfile(object):
    def __init__(self, fn, mode="r", **_kw):
        self.references, self._tid, self.reads = [], {}, []
        self._by_tid = self._mates = None
        if fn is None:
            return
        with open(fn, "rb") as f:
            magic = f.read(2)
        if magic == b"\x1f\x8b":
            self._read_bam(fn)
        else:
            self._read_sam(fn)

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

    def _read_bam(self, fn):
        with gzip.open(fn, "rb") as f:
            data = f.read()
        assert data[:4] == b"BAM\x01", "not a BAM file"
        o = 4
        l_text, = struct.unpack_from("<i", data, o); o += 4 + l_text
        n_ref, = struct.unpack_from("<i", data, o); o += 4
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", data, o); o += 4
            name = data[o:o + l_name - 1].decode(); o += l_name + 4
            self._tid[name] = len(self.references)
            self.references.append(name)
        while o + 4 <= len(data):
            bs, = struct.unpack_from("<i", data, o); o += 4
            tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, rnext, pnext, tlen = struct.unpack_from("<iiBBHHHiiii", data, o)
            q = o + 32
            qname = data[q:q + l_rn - 1].decode(); q += l_rn
            cig = []
            for _i in range(n_cig):
                v, = struct.unpack_from("<I", data, q); q += 4
                cig.append((v & 0xF, v >> 4))
            raw = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
            seq = "".join(_SEQ16[(raw[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
            qual = "".join(chr(33 + (x if x != 255 else 0)) for x in data[q:q + l_seq])
            self.reads.append(AlignedRead(qname, flag, tid, pos, mapq, cig or None, rnext, pnext, tlen, seq, qual))
            o += bs

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
