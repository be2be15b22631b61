"""Reference-genome access for the steps around the hot path (host-side Python, nothing here runs on the GPU):

  * FastaIndex        -- random access to a (multi-)FASTA through a .fai-style index (built by one scan, or read from
                         <fasta>.fai when present); stands in for Biopython's SeqIO.to_dict (utils.py:363).
  * extract_refseq_fa -- utils.extract_refseq_fa (utils.py:355-381): the target window [start-200, end+200) of the target's
                         chromosome as <name>_forward_refseq.fa / <name>_reverse_refseq.fa (+ marker files).
  * discover_partners -- where the reference hands a contig that its target window does not explain to a whole-genome
                         gfServer (sv_processor.py:829-831, utils.py:620-657), this path realigns against explicit partner
                         windows.  They are found here from the evidence the reference itself collects while extracting reads
                         (read_d['disc'], sv_processor.py:58-66: pairs whose mate maps to another chromosome or > 1 kb away):
                         the mate positions are clustered and each cluster with enough pairs becomes a window of the genome.
                         A heuristic stand-in for the genome-wide search, NOT pinned against gfServer (absent binary, no
                         genome): DESIGN.md section 8.
  * GenomeIndex       -- the genome-wide part of that search for contigs no discordant pair explains (a translocation
                         supported by split reads only): a sampled k-mer index of the whole FASTA (every `step`-th 16-mer,
                         sorted; the analogue of gfServer's -stepSize tiles, utils.py:620-657), queried with every k-mer of
                         the contig segment the target window leaves unaligned; loci with >= 2 hits on one diagonal band (BLAT's
                         -minMatch=2) become partner windows, and the target is run again with them.
"""
from __future__ import annotations

import os

_COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


class FastaIndex(object):
    def __init__(self, path):
        self.path = path
        self.index = {}                                  # name -> (length, offset, line_bases, line_bytes)
        fai = path + ".fai"
        if os.path.isfile(fai) and os.path.getmtime(fai) >= os.path.getmtime(path):
            with open(fai) as f:
                for ln in f:
                    p = ln.rstrip("\n").split("\t")
                    if len(p) >= 5:
                        self.index[p[0]] = (int(p[1]), int(p[2]), int(p[3]), int(p[4]))
        else:
            self._scan()
        self._f = open(path, "rb")

    def _scan(self):
        name, length, offset, lb, lby = None, 0, 0, 0, 0
        pos = 0
        with open(self.path, "rb") as f:
            for raw in f:
                if raw.startswith(b">"):
                    if name is not None:
                        self.index[name] = (length, offset, lb, lby)
                    name = raw[1:].split()[0].decode()
                    length, offset, lb, lby = 0, pos + len(raw), 0, 0
                else:
                    n = len(raw.rstrip(b"\r\n"))
                    if lb == 0 and n:
                        lb, lby = n, len(raw)
                    length += n
                pos += len(raw)
        if name is not None:
            self.index[name] = (length, offset, lb, lby)

    def _key(self, chrom):
        c = str(chrom)
        for k in (c, "chr" + c, c.replace("chr", "")):
            if k in self.index:
                return k
        raise KeyError("chromosome %s not in %s" % (chrom, self.path))

    def length(self, chrom):
        return self.index[self._key(chrom)][0]

    def fetch(self, chrom, start, end, upper=True):
        """bases [start, end) (0-based, clipped to the sequence); upper case unless upper=False (the bytes of the file:
        soft-masked genomes keep their lower case, as Biopython's str(seq) does in utils.extract_refseq_fa)"""
        length, offset, lb, lby = self.index[self._key(chrom)]
        start, end = max(0, int(start)), min(length, int(end))
        if end <= start or lb == 0:
            return ""
        b0 = offset + (start // lb) * lby + start % lb
        b1 = offset + ((end - 1) // lb) * lby + (end - 1) % lb + 1
        self._f.seek(b0)
        seq = self._f.read(b1 - b0).replace(b"\n", b"").replace(b"\r", b"").decode()
        return seq.upper() if upper else seq

    def close(self):
        self._f.close()


class GenomeIndex(object):
    """Sampled k-mer index of a FastaIndex: the k-mers (k <= 16, 2 bit/base in a uint32) starting at every `step`-th base of
    every sequence, N-free, sorted by code with (sequence number, position).  Built once per run, on first use (a 3 Gb genome
    at step 8: ~375 M entries, ~3.4 GB, about a minute of numpy); tests use kilobase genomes."""

    def __init__(self, fasta, k=16, step=8, max_occ=64, cache=True, device=None):
        """cache: keep the sorted index next to the FASTA (<fasta>.bkidx.k<k>s<step>.npz, valid while size and mtime of the FASTA
        stay) -- a 3 Gb genome takes minutes to index and every rank of a run needs the same one.  device: GPU number for the
        look-ups (hip_backend.DeviceIndex: the sorted codes in HBM, binary search per query k-mer on the device); None = host
        numpy.searchsorted (the same ranges: tests/test_hip_gpu.py pins one against the other)."""
        import numpy as np
        self.k, self.step, self.max_occ, self.fasta = int(k), int(step), int(max_occ), fasta
        self.names = list(fasta.index.keys())
        self._dev, self._dev_no, self._dev_loci = None, device, False
        self.probe_ms = 0.0
        cfn = None
        if cache and getattr(fasta, "path", None):
            st = os.stat(fasta.path)
            cfn = "%s.bkidx.k%ds%d.npz" % (fasta.path, self.k, self.step)
            if os.path.isfile(cfn):
                try:
                    z = np.load(cfn)
                    if int(z["size"]) == st.st_size and int(z["mtime_ns"]) == st.st_mtime_ns and list(z["names"]) == self.names:
                        self.code, self.seqno, self.pos = z["code"], z["seqno"], z["pos"]
                        return
                except Exception:
                    pass
        codes, seqno, pos = [], [], []
        lut = np.full(256, 4, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = i
            lut[ch + 32] = i                              # lower case (soft-masked genomes)
        for si, name in enumerate(self.names):
            length = fasta.index[name][0]
            CH = 1 << 24                                  # chunks of 16 Mb (+ k - 1 bases of overlap)
            for c0 in range(0, length, CH):
                s_ = fasta.fetch(name, c0, min(length, c0 + CH + self.k - 1), upper=False)
                b = lut[np.frombuffer(s_.encode(), dtype=np.uint8)]
                n = len(b) - self.k + 1
                if n <= 0:
                    continue
                first = (-c0) % self.step                 # positions that are multiples of `step` in sequence coordinates
                starts = np.arange(first, n, self.step, dtype=np.int64)
                if not len(starts):
                    continue
                code = np.zeros(len(starts), dtype=np.uint32)
                bad = np.zeros(len(starts), dtype=bool)
                for t in range(self.k):
                    col = b[starts + t]
                    bad |= col > 3
                    code = (code << np.uint32(2)) | (col & 3).astype(np.uint32)
                keep = ~bad
                codes.append(code[keep]); seqno.append(np.full(int(keep.sum()), si, dtype=np.uint16)); pos.append((starts[keep] + c0).astype(np.uint32))
        if codes:
            code = np.concatenate(codes); order = np.argsort(code, kind="stable")
            self.code, self.seqno, self.pos = code[order], np.concatenate(seqno)[order], np.concatenate(pos)[order]
        else:
            self.code = np.zeros(0, dtype=np.uint32); self.seqno = np.zeros(0, dtype=np.uint16); self.pos = np.zeros(0, dtype=np.uint32)
        if cfn:
            try:
                tmp = cfn + ".%d.tmp.npz" % os.getpid()
                np.savez(tmp, code=self.code, seqno=self.seqno, pos=self.pos, size=st.st_size, mtime_ns=st.st_mtime_ns, names=np.array(self.names))
                os.replace(tmp, cfn)                      # several ranks may build at once: whoever finishes last wins, every file is complete
            except OSError:
                pass

    def _ranges(self, code):
        """index ranges [lo, hi) of the entries equal to each query code"""
        import numpy as np
        if self._dev_no is not None:
            self._device()
            lo, hi = self._dev.probe(code)
            self.probe_ms += self._dev.kernel_ms
            return lo.astype(np.int64), hi.astype(np.int64)
        return np.searchsorted(self.code, code, side="left"), np.searchsorted(self.code, code, side="right")

    def _device(self):
        if self._dev is None:
            from . import hip_backend
            self._dev = hip_backend.DeviceIndex(self.code, self._dev_no)
        return self._dev

    def _codes(self, seq):
        import numpy as np
        lut = np.full(256, 4, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = i
        b = lut[np.frombuffer(seq.encode(), dtype=np.uint8)]
        n = len(b) - self.k + 1
        if n <= 0:
            return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=bool)
        code = np.zeros(n, dtype=np.uint32); bad = np.zeros(n, dtype=bool)
        for t in range(self.k):
            col = b[t:t + n]
            bad |= col > 3
            code = (code << np.uint32(2)) | (col & 3).astype(np.uint32)
        return code, ~bad

    def find(self, seq, min_hits=2, band=32):
        """loci of `seq` in the genome: [(hits, sequence name, strand, start, end)], best first; a locus = index hits of one
        strand on diagonals within `band` of each other, `min_hits` or more (k-mers that occur more than max_occ times in
        the index are repeats and do not count)"""
        import numpy as np
        out = []
        for strand, q in (("+", seq), ("-", revcomp(seq))):
            code, ok = self._codes(q)
            if not len(code):
                continue
            if self._dev_no is not None and len(code) <= 32768:
                # the whole look-up on the device (bk_index_find: ranges, hits sorted by (sequence, diagonal, position), loci cut at
                # sequence changes / diagonal jumps, >= min_hits): the same loci in the same order as the numpy path below
                self._device()
                if not self._dev_loci:
                    self._dev.set_loci(self.seqno, self.pos)
                    self._dev_loci = True
                for nh, sq_, p0, p1 in self._dev.find(code, ok, self.max_occ, band, min_hits):
                    out.append((nh, self.names[sq_], strand, p0, p1 + self.k))
                self.probe_ms += self._dev.kernel_ms
                continue
            lo, hi = self._ranges(code)
            use = np.nonzero(ok & (hi > lo) & (hi - lo <= self.max_occ))[0]
            if not len(use):
                continue
            cnt = (hi[use] - lo[use]).astype(np.int64)
            qp = np.repeat(use, cnt)                                                  # query position of every hit
            e = np.repeat(lo[use], cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))      # its index entry
            sq = self.seqno[e].astype(np.int64); ps = self.pos[e].astype(np.int64); dg = ps - qp
            order = np.lexsort((ps, dg, sq))                                          # by (sequence, diagonal, position): as sorted tuples
            sq, dg, ps = sq[order], dg[order], ps[order]
            brk = np.nonzero((sq[1:] != sq[:-1]) | (dg[1:] - dg[:-1] > band))[0] + 1  # a locus ends where the sequence changes or the diagonal jumps
            starts = np.concatenate(([0], brk)); ends = np.concatenate((brk, [len(sq)]))
            for a, b in zip(starts.tolist(), ends.tolist()):
                if b - a >= min_hits:
                    out.append((b - a, self.names[int(sq[a])], strand, int(ps[a:b].min()), int(ps[a:b].max()) + self.k))
        out.sort(key=lambda x: (-x[0], x[1], x[3]))
        return out


def revcomp(seq):
    return seq.encode().translate(_COMP)[::-1].decode()


def extract_refseq_fa(gene_coords, ref_path, fasta, direction):      # utils.py:355-381
    chrom, s, e, name = gene_coords[:4]
    fa_fn = os.path.join(ref_path, name + '_' + direction + '_refseq.fa')
    marker = os.path.join(ref_path, "." + name + '_' + direction + '_refseq.fa')
    if not os.path.isfile(marker):
        # the bytes the reference writes: str(seq) keeps the case of a soft-masked genome (utils.py:366-371); readers of the
        # file upper-case (sv_processor.read_fasta_first), as Jellyfish and BLAT ignore case.  Only difference: a window that
        # starts before base 0 is clipped here, where Python's negative slice index would wrap around in the reference.
        seq = fasta.fetch(chrom, int(s) - 200, int(e) + 200, upper=False)
        if direction == "reverse":
            seq = revcomp(seq)
        os.makedirs(ref_path, exist_ok=True)
        with open(fa_fn, 'w') as f:
            f.write(">" + name + "\n" + seq + "\n")
        open(marker, 'w').close()
    return fa_fn


def discover_partners(disc, fasta, annotations, target_chrom, target_start, target_end, min_pairs=2, join=1000, flank=1500, max_windows=8, skipped=None):
    """disc: {mate chromosome: [(read position, mate position), ...]} of one target (sv_processor.py:60-66).
    -> [(chrom, start, end, name, sequence)] in genome coordinates, most supported first: one window per cluster of mate
    positions (neighbours <= `join` apart) with >= `min_pairs` pairs, `flank` bases around it, clipped to the chromosome;
    clusters inside the target's own window [start-200, end+200) are the target itself and are skipped."""
    out = []
    tc = str(target_chrom).replace("chr", "")
    for chrom, pairs in disc.items():
        pos = sorted(int(p[1]) for p in pairs)
        i = 0
        while i < len(pos):
            j = i
            while j + 1 < len(pos) and pos[j + 1] - pos[j] <= join:
                j += 1
            n = j - i + 1
            lo, hi = pos[i], pos[j]
            i = j + 1
            if n < min_pairs:
                continue
            c = str(chrom).replace("chr", "")
            if c == tc and hi >= target_start - 200 and lo <= target_end + 200:
                continue
            try:
                clen = fasta.length(c)
            except KeyError:
                continue
            s, e = max(0, lo - flank), min(clen, hi + flank)
            seq = fasta.fetch(c, s, e)
            if len(seq) < 64 or seq.strip("ACGTNacgtn"):
                if skipped is not None and len(seq) >= 64:
                    skipped.append((c, s, e))              # IUPAC codes other than N cannot be packed: the caller logs these
                continue
            name = annotations.set_gene(c, [(lo + hi) // 2]) if annotations is not None else "intergenic"
            out.append((n, (c, s, e, name, seq)))
    out.sort(key=lambda x: (-x[0], x[1][0], x[1][1]))
    return [w for _n, w in out[:max_windows]]
