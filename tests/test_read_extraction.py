"""N2: read extraction + get_fastq_reads restatement against fixtures generated from the real
reference (tools/make_golden.py g6), plus the SAM/BAM reader itself.  CPU only."""
import gzip
import hashlib
import json
import os
import struct

import pytest

from breakmer_amd import read_extraction as rx
from breakmer_amd import samio, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "read_extraction.json")
CASES = json.load(open(GOLD))["cases"]


def _sam_for(case, tmp_path):
    r = synth.make_region(**{k: v for k, v in case["region"].items()})
    sam = synth.make_sam(r, case["n_pairs"])
    assert hashlib.sha1(sam.encode()).hexdigest() == case["sam_sha1"], "generator drifted from the fixture"
    fn = tmp_path / (case["tag"] + ".sam")
    fn.write_text(sam)
    return r, str(fn)


def _tuples(x):
    return json.loads(json.dumps(x))


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_extraction_matches_reference(case, tmp_path):
    r, fn = _sam_for(case, tmp_path)
    exp = case["expected"]
    sv, fq, fa, disc = rx.extract_reads(samio.Samfile(fn), r.chrom, r.start, r.end, case["kmer"])
    assert list(sv) == exp["sv_order"]
    assert fq == exp["fastq"]
    assert fa == exp["sc_fasta"]
    assert _tuples(disc) == exp["disc_reads"]
    for q, (rd, sc, cc, io) in sv.items():
        e = exp["sv_reads"][q]
        assert hashlib.sha1(rd.seq.encode()).hexdigest()[:10] == e[0] and sc == e[1] and cc == e[2] and io == e[3], q
    from breakmer_amd import sv_caller
    cf = sv_caller.bam_coverage_fn(samio.Samfile(fn))
    for tbp, want in exp["brkpt_coverages"].items():
        assert sv_caller.brkpt_coverages(tbp, cf) == want, tbp
    for variant in ("as_extracted", "trimmed"):
        e = exp[variant]
        recs, read_len = rx.get_fastq_reads(e["cleaned"] if e["cleaned"] is not None else fq, sv)
        assert [x[0] for x in recs] == e["kept"], variant
        assert read_len == e["read_len"]
        assert [bool(x[3]) for x in recs] == e["indel_only"]


def _bam_bytes(refs, reads):
    out = [b"BAM\x01", struct.pack("<i", 0), struct.pack("<i", len(refs))]
    for n in refs:
        out += [struct.pack("<i", len(n) + 1), n.encode() + b"\0", struct.pack("<i", 250000000)]
    for r in reads:
        cig = r.cigar or []
        seq = r.seq
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= "=ACMGRSVTWYHKDBN".index(ch) << (4 if i % 2 == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", r.tid, r.pos, len(r.qname) + 1, r.mapq, 0, len(cig), r.flag, len(seq), r.rnext, r.pnext, r.tlen)
        body += r.qname.encode() + b"\0" + b"".join(struct.pack("<I", ln << 4 | op) for op, ln in cig) + bytes(packed)
        body += bytes(ord(c) - 33 for c in r.qual)
        out += [struct.pack("<i", len(body)), body]
    return b"".join(out)


def test_bam_reader_equals_sam_reader(tmp_path):
    case = CASES[0]
    r, fn = _sam_for(case, tmp_path)
    s = samio.Samfile(fn)
    raw = _bam_bytes(s.references, s.reads)
    bfn = tmp_path / "x.bam"
    half = len(raw) // 2                              # two gzip members, like BGZF blocks
    with open(bfn, "wb") as f:
        f.write(gzip.compress(raw[:half]) + gzip.compress(raw[half:]))
    b = samio.Samfile(str(bfn))
    assert b.references == s.references and len(b.reads) == len(s.reads)
    for x, y in zip(b.reads, s.reads):
        assert (x.qname, x.flag, x.tid, x.pos, x.mapq, x.cigar, x.rnext, x.pnext, x.tlen, x.seq, x.qual) == \
               (y.qname, y.flag, y.tid, y.pos, y.mapq, y.cigar, y.rnext, y.pnext, y.tlen, y.seq, y.qual)
    a1 = rx.extract_reads(b, r.chrom, r.start, r.end, case["kmer"])
    assert a1[1] == case["expected"]["fastq"]


def test_trim_helpers():
    assert rx.trim_coords("###III##", 3) == (3, 6, 3)
    assert rx.trim_coords("####", 3) == (0, 0, 0)
    assert samio.parse_cigar("5S90M2D5M") == [(4, 5), (0, 90), (2, 2), (0, 5)]
    assert samio.parse_cigar("*") is None


def test_overlapping_targets_see_pristine_reads(tmp_path):
    """Two targets whose [start-200, end+200) windows overlap, served by ONE cached Samfile (params.open_bam): the
    second extraction must equal an extraction from a freshly parsed file, as in the reference, which re-opens the
    alignment file per target (sv_processor.py:426) -- trim_qual / process_reads mutate the records they are given."""
    case = CASES[0]
    r, fn = _sam_for(case, tmp_path)
    k = case["kmer"]
    shared = samio.Samfile(fn)
    first = rx.extract_reads(shared, r.chrom, r.start, r.end, k)
    assert "#" in "".join(x.qual for x in samio.Samfile(fn).reads), "fixture has no low-quality ends"
    assert any(len(rd.seq) < len(o.seq) for (rd, _s, _c, _i) in first[0].values()
               for o in shared.reads if o.qname == rd.qname and o.is_read1 == rd.is_read1), "nothing was trimmed"
    for shift in (0, 150):                           # the same window again, and one that overlaps it
        got = rx.extract_reads(shared, r.chrom, r.start + shift, r.end + shift, k)
        want = rx.extract_reads(samio.Samfile(fn), r.chrom, r.start + shift, r.end + shift, k)
        assert got[1] == want[1] and got[2] == want[2] and _tuples(got[3]) == _tuples(want[3]), shift
        assert list(got[0]) == list(want[0])
    assert rx.extract_reads(shared, r.chrom, r.start, r.end, k)[1] == case["expected"]["fastq"]


def test_region_filtered_reader_keeps_what_extraction_needs(tmp_path):
    """Samfile(regions=...) (what params.open_bam passes: the targets' windows) streams the BAM and keeps only the records
    in the windows plus their mates elsewhere: extraction results equal those of the unfiltered reader, although most of
    the file (reads far away, on other chromosomes) is never turned into objects."""
    case = CASES[0]
    r, fn = _sam_for(case, tmp_path)
    full = samio.Samfile(fn)
    far = []
    for i in range(300):                                  # bulk that no target needs: far away on the same chromosome, and on another one
        far.append(samio.AlignedRead("far%d" % i, 99 if i % 2 else 147, full._tid[r.chrom] if i % 3 else 1, 5000000 + 37 * i, 60, [(0, 100)],
                                     full._tid[r.chrom] if i % 3 else 1, 5000200 + 37 * i, 300, "ACGT" * 25, "I" * 100))
    raw = _bam_bytes(full.references, full.reads + far)
    bfn = tmp_path / "big.bam"
    third = len(raw) // 3
    with open(bfn, "wb") as f:
        f.write(gzip.compress(raw[:third]) + gzip.compress(raw[third:2 * third]) + gzip.compress(raw[2 * third:]))
    everything = samio.Samfile(str(bfn))
    assert len(everything.reads) == len(full.reads) + 300
    filt = samio.Samfile(str(bfn), regions=[(r.chrom, r.start - 200, r.end + 200)])
    assert len(filt.reads) < len(everything.reads) - 250 and not any(x.qname.startswith("far") for x in filt.reads)
    a = rx.extract_reads(everything, r.chrom, r.start, r.end, case["kmer"])
    b = rx.extract_reads(filt, r.chrom, r.start, r.end, case["kmer"])
    assert b[1] == a[1] == case["expected"]["fastq"] and b[2] == a[2] and _tuples(b[3]) == _tuples(a[3]) == case["expected"]["disc_reads"]
    sfilt = samio.Samfile(fn, regions=[(r.chrom, r.start - 200, r.end + 200)])                  # the SAM-text path applies the same rule
    assert rx.extract_reads(sfilt, r.chrom, r.start, r.end, case["kmer"])[1] == case["expected"]["fastq"]


def test_coverage_outside_kept_windows_counts_in_the_file(tmp_path):
    """Breakpoint coverage (sv_caller.py:118-133) at a position outside the windows a region-filtered reader kept (the partner
    side of a translocation) is counted in the file, not in the kept subset."""
    from breakmer_amd import sv_caller
    case = CASES[0]
    r, fn = _sam_for(case, tmp_path)
    full = samio.Samfile(fn)
    far = [samio.AlignedRead("far%d" % i, 99, full._tid[r.chrom], 7000000 + 10 * i, 60 if i % 4 else 5, [(0, 100)], full._tid[r.chrom], 7000300, 300,
                             "ACGT" * 25, "I" * 100) for i in range(40)]
    bfn = tmp_path / "c.bam"
    with open(bfn, "wb") as f:
        f.write(gzip.compress(_bam_bytes(full.references, full.reads + far)))
    filt = samio.Samfile(str(bfn), regions=[(r.chrom, r.start - 200, r.end + 200)])
    everything = samio.Samfile(str(bfn))
    for tbp in ("chr%s:%d" % (r.chrom, 7000150), "chr%s:%d-%d" % (r.chrom, r.start + 300, 7000200)):
        want = sv_caller.brkpt_coverages(tbp, sv_caller.bam_coverage_fn(everything))
        assert sv_caller.brkpt_coverages(tbp, sv_caller.bam_coverage_fn(filt)) == want and any(int(x) > 0 for x in want.split(",")), tbp
