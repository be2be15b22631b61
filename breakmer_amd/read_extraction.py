"""Selection of the reads that carry structural-variant evidence (the step in front of the hot path:
SURVEY.md 8f N2).  Restates, on `samio.AlignedRead` records, what the reference does with pysam:
  process_reads / pe_meta / add_discordant_pe      sv_processor.py:12-93
  target.extract_bam_reads, check_pair_overlap     sv_processor.py:422-583
  trim_coords / trim_qual / fq_line                utils.py:385-443
  get_fastq_reads                                  utils.py:203-246
Host-side Python; nothing here runs on the GPU.  Dict iteration is insertion ordered (P4 of SURVEY 8c)."""
from __future__ import annotations


def seq_trim(qual, min_qual):                                       # utils.py:385-391
    n = 0
    while ord(qual[n]) - 33 < min_qual:
        n += 1
        if n == len(qual):
            break
    return n


def get_seq_readname(read):                                         # utils.py:394-398
    return read.qname + "/" + ('2' if read.is_read2 else '1')


def trim_coords(qual, min_qual):                                    # utils.py:401-411
    start = seq_trim(qual, min_qual)
    if start == len(qual):
        return (0, 0, 0)
    end = len(qual) - seq_trim(qual[::-1], min_qual)
    return (start, end, end - start)


def trim_qual(read, min_qual, min_len):                             # utils.py:414-433 (mutates the read)
    q = read.qual
    start = seq_trim(q, min_qual)
    if start == len(q):
        return None
    end = len(q) - seq_trim(q[::-1], min_qual)
    if end - start < min_len:
        return None
    read.seq, read.qual = read.seq[start:end], q[start:end]
    return read


def fq_line(read, indel_only, min_len, trim=True):                  # utils.py:436-443
    if trim:
        read = trim_qual(read, 5, min_len)
    if not read:
        return None
    return "@" + get_seq_readname(read) + "_" + ('1' if indel_only else '0') + "\n" + read.seq + "\n+\n" + read.qual + "\n"


def pe_meta(r):                                                     # sv_processor.py:43-51
    proper = overlap = False
    if (r.flag in (83, 147) and r.tlen < 0) or (r.flag in (99, 163) and r.tlen > 0):
        proper = True
        overlap = abs(r.tlen) < 2 * len(r.seq)
    return proper, overlap


def add_discordant_pe(r, read_d, bam):                              # sv_processor.py:58-93
    if r.mapq > 0 and ((r.rnext != -1 and r.tid != r.rnext) or abs(r.tlen) > 1000) and not r.mate_is_unmapped:
        mate_ref = bam.getrname(r.rnext)
        if bam.mate(r).mapq > 0:
            read_d['disc'].setdefault(mate_ref, []).append((r.pos, r.pnext))
    if r.mapq > 0 and not r.mate_is_unmapped and r.tid == r.mrnm and r.is_read1:
        pos = None
        if r.is_reverse and r.mate_is_reverse:
            pos = (r.mpos, r.pos, 0, 0, r.qname) if r.mpos < r.pos else (r.pos, r.mpos, 0, 0, r.qname)
            read_d['inv_reads'].append(pos)
        elif not r.is_reverse and not r.mate_is_reverse:
            pos = (r.mpos, r.pos, 1, 1, r.qname) if r.mpos < r.pos else (r.pos, r.mpos, 1, 1, r.qname)
            read_d['inv_reads'].append(pos)
        elif r.is_reverse and not r.mate_is_reverse and r.pos < r.mpos:
            pos = (r.pos, r.mpos, 0, 1, r.qname)
            read_d['td_reads'].append(pos)
        elif not r.is_reverse and r.mate_is_reverse and r.mpos < r.pos:
            pos = (r.mpos, r.pos, 1, 0, r.qname)
            read_d['td_reads'].append(pos)
        if pos:
            read_d['other'].append(pos)


def process_reads(areads, read_d, bam):                             # sv_processor.py:12-40
    pair_indices, valid = {}, []
    for r in areads:
        if r.mate_is_unmapped or r.rnext == -1:
            r.mate_is_unmapped = True
        skip = r.is_duplicate or r.is_qcfail
        if r.is_unmapped:
            read_d['unmapped'][r.qname] = r
            skip = True
        if skip:
            continue
        proper = overlap = False
        fresh = r.qname not in pair_indices and not r.mate_is_unmapped
        if fresh:
            add_discordant_pe(r, read_d, bam)
            proper, overlap = pe_meta(r)
        valid.append((r, proper, overlap))
        if fresh:
            pair_indices[r.qname] = {}
        if r.qname in pair_indices:
            pair_indices[r.qname][int(r.is_read1)] = len(valid) - 1
    return pair_indices, valid


def check_pair_overlap(mate_seq, read, coords, trim_dir):           # sv_processor.py:548-583
    sc_seq = read.seq[coords[0]:coords[1]]
    sc_len = coords[1] - coords[0]
    if abs(read.isize) < len(read.seq):
        return not (abs(len(read.seq) - (abs(read.isize) + 1)) >= sc_len)        # adapter read-through
    misses = 0

    def off(s):                                                     # check_overlap :542-545
        return mate_seq.find(s) != (len(mate_seq) - len(s)) if trim_dir == 'back' else mate_seq.find(s) != 0
    while off(sc_seq) and misses < 5 and len(sc_seq) > 0:
        sc_seq = sc_seq[:-1] if trim_dir == 'back' else sc_seq[1:]
        misses += 1
    return len(sc_seq) == 0 or misses == 5


def extract_reads(bam, chrom, start, end, kmer_size):
    """target.extract_bam_reads (sv_processor.py:422-540) -> (sv_reads, fastq_text, sc_fasta_text, disc_reads)."""
    read_d = {'unmapped': {}, 'disc': {}, 'sv': {}, 'unmapped_keep': [], 'inv_reads': [], 'td_reads': [], 'other': []}
    areads = bam.fetch(chrom, start - 200, end + 200)
    pair_indices, valid = process_reads(areads, read_d, bam)
    for r, proper, overlap in valid:
        if r.cigar and len(r.cigar) > 1:
            tc = trim_coords(r.qual, 3)
            coords = [0, 0]
            for i, (code, clen) in enumerate(r.cigar):
                if code != 2 and code != 4:
                    coords[1] += clen
                if code == 4 and i == 0:
                    coords[0] = clen
                    coords[1] += clen
            s, e = coords
            if s > tc[0] or e < tc[1]:
                sc_seq = {'clipped': [], 'buffered': []}
                clip_coords = [0, 0]
                add = [False, False]
                indel_only = False
                start_sc, end_sc = s > 0, e < len(r.qual)
                seq = r.seq
                ll = len(seq)
                if start_sc and end_sc:
                    add = [True, True]
                elif start_sc:
                    add[0] = True
                    clip_coords = [0, s]
                    if overlap and r.is_reverse:
                        mate_seq = valid[pair_indices[r.qname][int(r.is_read1)]][0].seq
                        add[0] = check_pair_overlap(mate_seq, r, [0, s], 'back')
                    if proper:
                        indel_only = bool(r.is_reverse)
                elif end_sc:
                    clip_coords = [e, ll]
                    add[1] = True
                    if overlap and not r.is_reverse:
                        mate_seq = valid[pair_indices[r.qname][int(r.is_read1)]][0].seq
                        add[1] = check_pair_overlap(mate_seq, r, [e, ll], 'front')
                    if proper:
                        indel_only = False                           # `indel_only and ...` with indel_only False (:490-491)
                if add[0]:
                    sc_seq['buffered'].append(seq[0:(s + kmer_size)])
                    sc_seq['clipped'].append(seq[0:s])
                if add[1]:
                    sc_seq['buffered'].append(seq[(e - kmer_size):ll])
                    sc_seq['clipped'].append(seq[e:ll])
                if add[0] or add[1]:
                    read_d['sv'][get_seq_readname(r)] = (r, sc_seq, clip_coords, indel_only)
        if start <= r.pos <= end and r.mapq > 0 and r.mate_is_unmapped:
            read_d['unmapped_keep'].append(r.qname)
    fq, fa = [], []
    for qname in read_d['unmapped_keep']:
        if qname in read_d['unmapped']:
            rd = read_d['unmapped'][qname]
            read_d['sv'][get_seq_readname(rd)] = (rd, None, None, False)
            fa.append(">" + rd.qname + "\n" + str(rd.seq) + "\n")
    sv_reads = {}
    for qname, (r, sc_seq, _cc, indel_only) in read_d['sv'].items():
        sv_reads[qname] = read_d['sv'][qname]
        line = fq_line(r, indel_only, kmer_size, True)
        if line:
            fq.append(line)
        if sc_seq:
            for sc in sc_seq['buffered']:
                fa.append(">" + qname + "\n" + sc + "\n")
    disc = {'disc': read_d['disc'], 'inv': read_d['inv_reads'], 'td': read_d['td_reads'], 'other': read_d['other']}
    return sv_reads, "".join(fq), "".join(fa), disc


def get_fastq_reads(fastq_text, sv_reads):
    """utils.get_fastq_reads (:203-246) on the (cleaned) FASTQ text -> (kept records, read_len).
    A record is (header, seq, qual, indel_only)."""
    lines = fastq_text.split("\n")
    out, read_len = [], 0
    add = True
    for i in range(0, len(lines) - 3, 4):
        header, seq, qual = lines[i].strip(), lines[i + 1].strip(), lines[i + 3].strip()
        parts = header.lstrip("@").split("_")
        qname = "_".join(parts[:-1])
        indel_meta = False
        if qname in sv_reads:
            oseq, sc_seqs, _clip, indel_meta = sv_reads[qname]
            add = True
            old = oseq.seq
            if str(seq) != str(old) and sc_seqs:
                clips = sc_seqs['clipped']
                idx = old.find(seq)
                trimmed = old[len(seq):] if idx == 0 else old[0:idx]
                sc_lens = 0
                for sc in clips:
                    sc_lens += len(sc)
                    if trimmed.find(sc) > -1:
                        add = False
                if len(seq) == len(old) - sc_lens:
                    for sc in clips:
                        if seq.find(sc) == -1:
                            add = False
        if add:
            out.append((header, seq, qual, indel_meta))
            read_len = max(read_len, len(seq))
    return out, read_len
