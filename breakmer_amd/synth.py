"""Synthetic misaligned-read regions -- build-owned generator.

Counter-based RNG (splitmix64 keyed by (global_seed, region_id, stream)) so that
this container and the GPU box generate byte-identical inputs without Python's
`random`.  A region mimics what the reference hands to its hot path for one
target: a reference window (target interval + 200 bp flanks, utils.py:367 /
sv_processor.py:431), the cleaned reads (utils.py:203-246), read ids in the
`@<qname>/<1|2>_<0|1>` convention (utils.py:436-443), discordant-pair evidence
(sv_processor.py:376-408) and a one-gene-per-window annotation (utils.py:727-773).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)      # code 4 = N (no-call)
SV_TYPES = ("del", "ins", "inv", "dup", "trl")


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def stream_key(global_seed: int, region_id: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        k = _mix(np.uint64(global_seed) * _GOLD + np.uint64(1))
        k = _mix(k ^ (np.uint64(region_id) * _GOLD + np.uint64(2)))
        k = _mix(k ^ (np.uint64(stream) * _GOLD + np.uint64(3)))
    return np.uint64(k)


def rand_u64(key: np.uint64, n: int, offset: int = 0) -> np.ndarray:
    """n counter-based 64-bit values: splitmix64(key + (offset+i)*GOLD)."""
    with np.errstate(over="ignore"):
        ctr = (np.arange(offset, offset + n, dtype=np.uint64) + np.uint64(1)) * _GOLD + np.uint64(key)
    return _mix(ctr)


def rand_bases(key, n):
    return (rand_u64(key, n) >> np.uint64(62)).astype(np.uint8)          # codes 0..3


def rand_below(key, n, bound):
    # 53-bit multiply-shift; bias is irrelevant here, determinism is what matters
    return ((rand_u64(key, n) >> np.uint64(11)).astype(np.float64) * (float(bound) / 9007199254740992.0)).astype(np.int64)


def revcomp_codes(c):
    return (3 - c[::-1]).astype(np.uint8)


def codes_to_str(c) -> str:
    return BASES[np.asarray(c, dtype=np.uint8)].tobytes().decode()


def str_to_codes(s: str) -> np.ndarray:
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    out = lut[np.frombuffer(s.encode(), dtype=np.uint8)]
    if (out == 255).any():
        raise ValueError("non-ACGT base in sequence (unsupported on this path)")
    return out


class SynthReadIds(object):
    """read ids `@S:1:1:<region>:<i>/1_0` (utils.py:436-443 convention), materialised on demand: a region has 10-24
    thousand of them and the batched paths only need to know that they all carry the same `/1_0` tag"""
    uniform_tag = "1_0"

    def __init__(self, region_id, n):
        self.region_id, self.n = region_id, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return "@S:1:1:%d:%d/1_0" % (self.region_id, i)

    def __iter__(self):
        return ("@S:1:1:%d:%d/1_0" % (self.region_id, i) for i in range(self.n))


class Region(object):
    """One synthetic target region.

    Attributes
    ----------
    name, chrom, start, end : target interval (BED-like); window = [start-200, end+200)
    window    : uint8 codes, reference window (forward)
    partners  : list of (chrom, start, end, name, codes) extra windows in genome coords
    donor     : uint8 codes, sample haplotype the reads are drawn from
    reads     : uint8 [N, L] codes (reference orientation, FASTQ order)
    read_ids  : list[str]
    sv_type   : one of SV_TYPES
    disc_reads: dict as built by sv_processor.py:476-531
    """

    def __init__(self):
        self.partners = []

    def read_strs(self):
        return [codes_to_str(r[:n]) for r, n in zip(self.reads, self.read_lens)]

    @property
    def window_str(self):
        return codes_to_str(self.window)


def make_region(region_id: int, *, global_seed: int = 1, W: int = 3000, L: int = 150, depth: int = 500,
                sv_type: str = "del", sv_size: int | None = None, noise: float = 0.0,
                n_reads: int | None = None, var_len: float = 0.0, indel_only_frac: float = 0.0, n_frac: float = 0.0,
                flank_dups: int = 0, trl_repeat_copies: int = 0, microsat: int = 0, flank_div: int = 0) -> Region:
    """Generate region `region_id` (config 1-3 defaults).

    Multi-mapping variants (realign contract step 5; off by default, the default regions are unchanged):
      flank_dups (del only): bit 0 = the L bases left of the deletion also sit at window[20:20+L], bit 1 = the L bases right
        of it also at window[W-20-L:W-20], bit 2 = that right copy is reverse-complemented;
      flank_div (with flank_dups bit 0): the left copy is DIVERGED -- its last 30 bases (next to where the junction would be) are
        exact (two of BLAT's index tiles wherever the grid falls), of the bases before them every flank_div-th is substituted:
        4 -> an ~80 % copy (a +1/-2 segment still runs through it; BLAT's -minIdentity=90 drops it), 12 -> a ~94 % copy;
      trl_repeat_copies (trl only): a second partner window holding that many copies of the partner half of the donor;
      microsat (del only): the `microsat` bases left of the deletion are a (CA)n repeat -- a contig across the junction then
        aligns on every second diagonal of the repeat (tens to hundreds of secondary alignments)."""
    assert sv_type in SV_TYPES
    r = Region()
    r.region_id = region_id
    r.sv_type = sv_type
    r.name = "GENE%05d" % region_id
    r.chrom = "%d" % (1 + region_id % 22)            # BED chrom without "chr" (sv_caller.py:917,1124)
    flank = 200
    r.start = 100000 + 20000 * (region_id // 22) + flank
    r.end = r.start + (W - 2 * flank)
    win = rand_bases(stream_key(global_seed, region_id, 0), W)
    c = W // 2
    if sv_size is None:
        sv_size = 60 if sv_type == "ins" else 200
    h = sv_size // 2
    r.sv_size = sv_size
    if flank_dups:
        assert sv_type == "del" and W >= 2 * (h + 2 * L + 40)
        win = win.copy()
        if flank_dups & 1:
            cp = win[c - h - L:c - h].copy()
            if flank_div:
                at = np.arange(L - 31, -1, -flank_div)
                cp[at] = (cp[at] + 1) & 3
            win[20:20 + L] = cp
        if flank_dups & 2:
            cp = win[c + h:c + h + L]
            win[W - 20 - L:W - 20] = revcomp_codes(cp) if flank_dups & 4 else cp
    if microsat:
        assert sv_type == "del" and microsat <= c - h
        win = win.copy()
        win[c - h - microsat:c - h] = np.tile(np.array([1, 0], dtype=np.uint8), (microsat + 1) // 2)[:microsat]
    r.window = win
    if sv_type == "del":
        donor = np.concatenate([win[:c - h], win[c + h:]])
    elif sv_type == "ins":
        ins = rand_bases(stream_key(global_seed, region_id, 1), sv_size)
        donor = np.concatenate([win[:c], ins, win[c:]])
    elif sv_type == "inv":
        donor = np.concatenate([win[:c - h], revcomp_codes(win[c - h:c + h]), win[c + h:]])
    elif sv_type == "dup":
        donor = np.concatenate([win[:c + h], win[c - h:c + h], win[c + h:]])
    else:  # trl: left half of target window joined to right half of a partner window
        pw = rand_bases(stream_key(global_seed, region_id, 1), W)
        pchrom = "%d" % (1 + (region_id + 7) % 22)
        pstart = 50000000 + 20000 * region_id
        r.partners.append((pchrom, pstart, pstart + W, "PARTNER%05d" % region_id, pw))
        donor = np.concatenate([win[:c], pw[c:]])
        if trl_repeat_copies:
            sp = rand_bases(stream_key(global_seed, region_id, 7), 40 * (trl_repeat_copies + 1))
            parts = [sp[:40]]
            for i in range(trl_repeat_copies):
                parts += [pw[c:], sp[40 * (i + 1):40 * (i + 2)]]
            rw = np.concatenate(parts)
            rchrom, rstart = "%d" % (1 + (region_id + 13) % 22), 90000000 + 20000 * region_id
            r.partners.append((rchrom, rstart, rstart + len(rw), "REPEAT%05d" % region_id, rw))

    r.donor = donor
    N = n_reads if n_reads is not None else (depth * W) // L
    starts = rand_below(stream_key(global_seed, region_id, 2), N, len(donor) - L + 1)
    reads = np.lib.stride_tricks.sliding_window_view(donor, L)[starts]       # rows donor[s:s+L] (a row gather of a strided view)
    if noise > 0.0:
        u = rand_u64(stream_key(global_seed, region_id, 3), N * L).reshape(N, L)
        # (u >> 11) * 2**-53 < noise, evaluated on the integers (x * 2**-53 is exact; for an integer x, x < y <=> x < ceil(y))
        import math
        flip = (u >> np.uint64(11)) < np.uint64(math.ceil(noise * 9007199254740992.0))
        delta = ((u & np.uint64(0x3FF)) % np.uint64(3)).astype(np.uint8) + np.uint8(1)  # 1..3 -> always a different base
        reads = np.where(flip, (reads + delta) & 3, reads).astype(np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    if n_frac > 0.0:
        # no-calls as real alignment files have them: a fraction of the reads carries one N (code 4), a few of them a second one
        u = rand_u64(stream_key(global_seed, region_id, 6), N)
        sel = np.nonzero((u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) < n_frac)[0]
        p1 = ((u[sel] & np.uint64(0xFFFF)) % np.uint64(L)).astype(np.int64)
        reads[sel, p1] = 4
        two = ((u[sel] >> np.uint64(16)) & np.uint64(7)) == np.uint64(0)
        p2 = (((u[sel] >> np.uint64(20)) & np.uint64(0xFFFF)) % np.uint64(L)).astype(np.int64)
        reads[sel[two], p2[two]] = 4
    lens = np.full(N, L, dtype=np.int32)
    if var_len > 0.0:
        # quality/adapter-trimmed reads (utils.py:385-443, cutadapt): a fraction loses up to 40 bases
        # from its 3' or 5' end; rows stay left-aligned, the tail is padding (code 0, never read)
        u = rand_u64(stream_key(global_seed, region_id, 4), N)
        sel = (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) < var_len
        cut = ((u & np.uint64(0xFFFF)) % np.uint64(41)).astype(np.int32)
        five = ((u >> np.uint64(16)) & np.uint64(1)).astype(bool)
        for i in np.nonzero(sel & (cut > 0))[0]:
            c_ = int(cut[i])
            if five[i]:
                reads[i, :L - c_] = reads[i, c_:].copy()
            reads[i, L - c_:] = 0
            lens[i] = L - c_
    r.reads = reads
    r.read_lens = lens.astype(np.uint16)            # the dtype the library takes: no conversion per submit
    r.indel_only = np.zeros(N, dtype=np.uint8)
    if indel_only_frac > 0.0:
        u = rand_u64(stream_key(global_seed, region_id, 5), N)
        r.indel_only = ((u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) < indel_only_frac).astype(np.uint8)
    r.read_starts = starts
    r.read_ids = SynthReadIds(region_id, N)
    # discordant-pair evidence normally derived from the BAM (sv_processor.py:376-408)
    npairs = -(-depth // 50)
    gpos = r.start - flank  # genome coordinate of window[0]
    disc = {"disc": {}, "inv": [], "td": [], "other": []}
    if sv_type == "inv":
        b0, b1 = gpos + c - h, gpos + c + h
        for i in range(npairs):
            disc["inv"].append((b0 - 50 - i, b0 + 20 + i, 1, 1, "P%d" % i))
    elif sv_type == "trl":
        pchrom, pstart = r.partners[0][0], r.partners[0][1]
        disc["disc"][pchrom] = [(gpos + c - 100 - i, pstart + c + 100 + i) for i in range(npairs)]
    r.disc_reads = disc
    return r


def make_sam(r: Region, n_pairs: int = 200, L: int = 100, global_seed: int = 1, frag: int = 260) -> str:
    """SAM text of a toy aligner's view of read pairs drawn from `r.donor` (input of the read-extraction step,
    sv_processor.py:422-540): reads crossing the junction are soft-clipped on their shorter side, a few pairs
    overlap, have an unmapped mate, are duplicates / QC failures, carry low-quality tails, or are discordant.
    Only del / ins regions (collinear donor) are mapped; coordinates are 1-based in the text."""
    assert r.sv_type in ("del", "ins")
    W = len(r.window)
    c, h = W // 2, r.sv_size // 2
    gpos = r.start - 200
    chrom = r.chrom                                  # BED / BAM names without "chr" (sv_caller.py:917,1124)
    other = "%d" % (1 + (int(r.chrom) + 4) % 22)
    key = stream_key(global_seed, r.region_id, 11)
    u = rand_u64(key, n_pairs * 4).reshape(n_pairs, 4)
    D = len(r.donor)

    def to_ref(d):                                   # donor offset -> window offset of the base to its right
        if r.sv_type == "del":
            return d if d < c - h else d + 2 * h
        return d if d < c else (c if d < c + r.sv_size else d - r.sv_size)

    junctions = [c - h] if r.sv_type == "del" else [c, c + r.sv_size]

    def place(d0):
        """(pos0, cigar text) of donor[d0:d0+L]"""
        for jn in junctions:
            b = jn - d0
            if 0 < b < L:
                if r.sv_type == "del":
                    return (gpos + to_ref(d0), "%dM%dS" % (b, L - b)) if b >= L - b else (gpos + to_ref(jn), "%dS%dM" % (b, L - b))
                if jn == c:                          # read runs from reference into the inserted bases
                    return gpos + to_ref(d0), "%dM%dS" % (b, L - b)
                return gpos + c, "%dS%dM" % (b, L - b)
        if r.sv_type == "ins" and c <= d0 and d0 + L <= c + r.sv_size:
            return None, "*"
        return gpos + to_ref(d0), "%dM" % L

    out = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:%s\tLN:250000000" % chrom, "@SQ\tSN:%s\tLN:250000000" % other]
    for i in range(n_pairs):
        kind = int(u[i, 1] % np.uint64(16))
        fl = frag if kind != 1 else L + 30 + int(u[i, 2] % np.uint64(40))        # kind 1: overlapping pair
        if kind == 7:
            fl = L - 10                                                         # read-through (adapter) pair
        span = max(fl, L)
        # two thirds of the pairs sit near a junction
        if int(u[i, 0] % np.uint64(3)) < 2:
            d1 = junctions[0] - L + 5 + int((u[i, 0] >> np.uint64(8)) % np.uint64(L + span - 10)) - (span - L)
            d1 = min(max(d1, 0), D - span)
        else:
            d1 = int((u[i, 0] >> np.uint64(8)) % np.uint64(D - span + 1))
        d2 = d1 + span - L
        s1, s2 = codes_to_str(r.donor[d1:d1 + L]), codes_to_str(r.donor[d2:d2 + L])
        q1 = q2 = "I" * L
        if kind == 2:
            q1 = "I" * (L - 12) + "#" * 12
        if kind == 3:
            q2 = "#" * 9 + "I" * (L - 9)
        if kind == 8:
            q1 = "#" * L
        (p1, c1), (p2, c2) = place(d1), place(d2)
        name = "S:1:1:%d:%d" % (r.region_id, i)
        f1, f2 = 99, 147
        if int(u[i, 3] & np.uint64(1)):              # read 1 on the reverse strand: it is the rightmost read
            f1, f2 = 83, 163
            (p1, c1, s1, q1), (p2, c2, s2, q2) = (p2, c2, s2, q2), (p1, c1, s1, q1)
        if p1 is None or p2 is None:
            continue
        t = (max(p1, p2) + L) - min(p1, p2)
        t1 = t if p1 <= p2 and f1 == 99 else -t
        if kind == 4:
            f1 |= 0x400; f2 |= 0x400
        if kind == 5:
            f2 |= 0x200
        mq1 = mq2 = 60
        ch1 = ch2 = chrom
        if kind == 6:                                # mate unmapped: read 2 is stored at read 1's position
            f1, f2 = 73, 133
            out.append("\t".join([name, str(f1), chrom, str(p1 + 1), "60", c1, "=", str(p1 + 1), "0", s1, q1]))
            out.append("\t".join([name, str(f2), chrom, str(p1 + 1), "0", "*", "=", str(p1 + 1), "0", s2, q2]))
            continue
        if kind == 9:                                # mate on another chromosome
            f1 &= ~2; f2 &= ~2
            out.append("\t".join([name, str(f1), chrom, str(p1 + 1), "60", c1, other, str(5000 + i), "0", s1, q1]))
            out.append("\t".join([name, str(f2), other, str(5000 + i), "37", "%dM" % L, chrom, str(p1 + 1), "0", s2, q2]))
            continue
        if kind == 10:                               # same-strand pair (inversion evidence)
            f1, f2 = 65 | 0x0, 129
        if kind == 11:                               # everted pair (tandem-duplication evidence)
            f1, f2 = (81, 161) if p1 < p2 else (97, 145)
        if kind == 12:
            mq1 = 0
        out.append("\t".join([name, str(f1), ch1, str(p1 + 1), str(mq1), c1, "=", str(p2 + 1), str(t1), s1, q1]))
        out.append("\t".join([name, str(f2), ch2, str(p2 + 1), str(mq2), c2, "=", str(p1 + 1), str(-t1), s2, q2]))
    return "\n".join(out) + "\n"


def make_sam_trl(r: Region, n_pairs: int = 300, L: int = 100, global_seed: int = 1, frag: int = 260) -> str:
    """SAM text of a toy aligner's view of read pairs drawn from the donor of a TRANSLOCATION region (left half of the
    target window joined to the right half of the partner window): reads left of the junction map to the target chromosome,
    reads right of it to the partner chromosome, reads across it are soft-clipped on their shorter side; a pair with one end
    on each chromosome is the discordant evidence the reference collects (sv_processor.py:58-66).  Coordinates are 1-based."""
    assert r.sv_type == "trl" and len(r.partners) == 1
    W = len(r.window)
    c = W // 2
    gpos = r.start - 200
    pchrom, pstart = r.partners[0][0], r.partners[0][1]
    u = rand_u64(stream_key(global_seed, r.region_id, 12), n_pairs * 2).reshape(n_pairs, 2)
    D = len(r.donor)

    def place(d0):
        b = c - d0                                  # bases of the read left of the junction
        if b >= L:
            return r.chrom, gpos + d0, "%dM" % L
        if b <= 0:
            return pchrom, pstart + d0, "%dM" % L
        if b >= L - b:
            return r.chrom, gpos + d0, "%dM%dS" % (b, L - b)
        return pchrom, pstart + c, "%dS%dM" % (b, L - b)

    out = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:%s\tLN:250000000" % r.chrom, "@SQ\tSN:%s\tLN:250000000" % pchrom]
    for i in range(n_pairs):
        span = frag
        if int(u[i, 0] % np.uint64(4)) < 3:          # three quarters of the pairs sit around the junction
            d1 = c - span + 10 + int((u[i, 0] >> np.uint64(8)) % np.uint64(span + L - 20)) - L // 2
        else:
            d1 = int((u[i, 0] >> np.uint64(8)) % np.uint64(D - span + 1))
        d1 = min(max(d1, 0), D - span)
        d2 = d1 + span - L
        s1, s2 = codes_to_str(r.donor[d1:d1 + L]), codes_to_str(r.donor[d2:d2 + L])
        (c1, p1, g1), (c2, p2, g2) = place(d1), place(d2)
        name = "T:1:1:%d:%d" % (r.region_id, i)
        q = "I" * L
        if c1 == c2:
            t = (p2 + L) - p1
            out.append("\t".join([name, "99", c1, str(p1 + 1), "60", g1, "=", str(p2 + 1), str(t), s1, q]))
            out.append("\t".join([name, "147", c2, str(p2 + 1), "60", g2, "=", str(p1 + 1), str(-t), s2, q]))
        else:
            out.append("\t".join([name, "97", c1, str(p1 + 1), "60", g1, c2, str(p2 + 1), "0", s1, q]))
            out.append("\t".join([name, "145", c2, str(p2 + 1), "60", g2, c1, str(p1 + 1), "0", s2, q]))
    return "\n".join(out) + "\n"
