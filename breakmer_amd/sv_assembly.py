"""Host-side objects of the assembly seam (S2 of SURVEY.md 8b): what `init_assembly` returns to its
caller in the reference (sv_assembly.py:30-63, consumers sv_processor.py:737-741, 760-761, 864 and
sv_caller.py:159, 242, 271-277, 634), rebuilt from the records the HIP assembler writes.
The algorithm itself lives in csrc/bk_asm.hip.h; nothing here computes an assembly."""
from __future__ import annotations


class assembly_counts(object):
    """Per-base supporting-read counts of a contig (sv_assembly.py:160-221), read-only view."""

    def __init__(self, indel_only, others):
        self.indel_only = indel_only if isinstance(indel_only, list) else list(indel_only)     # read-only view: no copy of a list
        self.others = others if isinstance(others, list) else list(others)

    def get_counts(self, p1, p2, sv_type):                          # sv_assembly.py:167-176
        if sv_type in ('indel', 'rearr'):
            if p1 == p2:
                return self.indel_only[p1] + self.others[p1]
            return [a + b for a, b in zip(self.indel_only[p1:p2], self.others[p1:p2])]
        if p1 == p2:
            return self.others[p1]
        return self.others[p1:p2]

    def get_total_reads(self):                                      # :178-179
        return max(self.indel_only) + max(self.others)


class fq_read(object):
    """utils.py:681-688 -- the fields downstream code reads."""

    def __init__(self, header, seq, qual, indel_only):
        self.id = header
        self.seq = str(seq)
        self.qual = str(qual)
        self.used = False
        self.dup = False
        self.indel_only = indel_only


class _Aseq(object):
    def __init__(self, seq, counts):
        self.seq = seq
        self.counts = counts


class _KmerTuples(object):
    """contig.kmers of the reference is a list of (kmer, ...) tuples of which only x[0] and len() are read downstream
    (sv_processor.py:760-761, 864); this is that view over a sequence of k-mer strings, tuples made when asked for."""
    __slots__ = ("_m",)

    def __init__(self, mers):
        self._m = mers

    def __len__(self):
        return len(self._m)

    def __getitem__(self, i):
        return [(m,) for m in self._m[i]] if isinstance(i, slice) else (self._m[i],)

    def __iter__(self):
        return ((m,) for m in self._m)

    def __eq__(self, other):
        return list(self) == list(other)


class contig(object):
    """What sv_processor.contig copies from an assembled contig (sv_processor.py:737-741)."""

    def __init__(self, seq, indel_only, others, kmer_locs, kmers, reads, kmer_len):
        self.aseq = _Aseq(seq, assembly_counts(indel_only, others))
        self.kmer_locs = kmer_locs if isinstance(kmer_locs, list) else list(kmer_locs)
        self.kmers = _KmerTuples(kmers)
        self._reads = None
        self._read_src = reads                      # (source sequence, indices) or an iterable of fq_read: the set is built when asked for
        self.kmer_len = kmer_len

    @property
    def reads(self):                                # set of fq_read (sv_assembly.py:423); the writers and the Python call tail read it
        if self._reads is None:
            src = self._read_src
            self._reads = set(src[0][i] for i in src[1]) if isinstance(src, tuple) else set(src)
        return self._reads

    def get_total_read_support(self): return self.aseq.counts.get_total_reads()
    def get_contig_len(self): return len(self.aseq.seq)
    def get_kmer_locs(self): return self.kmer_locs
    def get_contig_seq(self): return self.aseq.seq
    def get_contig_counts(self): return self.aseq.counts


def contigs_from_engine(engine, region, reads, kmer_len):
    """Rebuild contig objects of one region from the HIP engine's records.
    `reads`: list of fq_read in FASTQ order (the representative of each supporting sequence is
    indexed by its position)."""
    out = []
    try:
        recs = engine.contigs(region, lazy_kmers=True)
    except TypeError:                               # an engine without the lazy view (tests/fake_engine.py)
        recs = engine.contigs(region)
    for c in recs:
        out.append(contig(c["seq"], c["indel_only"], c["others"], c["kmer_locs"], c["kmers"], (reads, c["reads"]), kmer_len))
    return out


class LazyContigs(object):
    """The contigs of one region as a read-only sequence whose length is known at once and whose objects are only built
    when one is looked at -- the driver with the native call tail and no per-contig files never does.  The records live in
    the engine's host copy of the batch, so they must be read before the engine takes its next batch."""

    def __init__(self, engine, region, reads, kmer_len, n=None):
        self._eng, self._region, self._reads, self._k = engine, region, reads, kmer_len
        self._serial = engine.batch_serial
        self._n = engine.contig_count(region) if n is None else n
        self._items = None

    @classmethod
    def counted(cls, n):
        """the contigs of a target whose engine has moved on: their number, nothing to look at any more"""
        self = cls.__new__(cls)
        self._eng = self._reads = self._items = None
        self._region = self._k = self._serial = None
        self._n = n
        return self

    def _get(self):
        if self._items is None:
            if self._eng is None or self._eng.batch_serial != self._serial:
                raise RuntimeError("the contigs of this target were not read before its engine took the next batch")
            self._items = contigs_from_engine(self._eng, self._region, self._reads, self._k)
            self._eng = self._reads = None
        return self._items

    def detach(self):
        """the engine moves on: keep what was built, forget the rest"""
        self._eng = self._reads = None

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, i):
        return self._get()[i]
