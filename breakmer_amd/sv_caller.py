"""SV calling from PSL-equivalent alignment records -- host-side restatement (Python 3) of the
reference's call logic, behind the reference's own names (`align_manager(meta_dict).get_result()`
-> the 13-field result row).

Reference semantics followed (file:line under the reference tree):
  blat_res            sv_caller.py:881-1142   one PSL record, CIGAR/breakpoints/identity
  blat_repeat_manager sv_caller.py:838-875    overlap with the repeat mask
  blat_manager        sv_caller.py:523-779    hit ranking, indel test, gap-filling merge of clipped hits
  align_manager       sv_caller.py:785-833    entry point
  sv_event            sv_caller.py:13-517     13-field row, rearrangement typing and filters
  filter_by_feature / check_intervals / calc_contig_complexity   utils.py:20-39, 58-94
Quirks kept on purpose: Q10 (Python-2 round), Q11, Q13 (tandem-dup support is always 0), Q14.
Records come either from a PSL text file (`query_res_fn`, 21 tab-separated columns, no header) or
directly as lists of the same 21 fields (`psl_records`) as produced by `records_from_hits`.
Breakpoint coverages come from `meta_dict['coverage_fn']` (sv_processor supplies one over the alignment file when
`sample_bam_file` is set; 0 otherwise).
"""
from __future__ import annotations

import logging
import math
from decimal import ROUND_HALF_UP, Decimal

RESULT_FIELDS = ['anno_genes', 'target_breakpoints', 'align_cigar', 'mismatches', 'strands', 'repeat_matching', 'sv_type',
                 'split_read_count', 'nkmers', 'disc_read_count', 'breakpoint_coverages', 'contig_id', 'contig_seq']


def round2(x, n=0):
    """Python 2.7 round(): exact halves away from zero, float result."""
    return float(Decimal(float(x)).quantize(Decimal(1).scaleb(-int(n)), rounding=ROUND_HALF_UP))


def _is_digit(ch):
    return ch in "0123456789"


# --------------------------------------------------------------------------- utils.py helpers
def count_nmers(seq, n):                                            # utils.py:32-39
    seen = {}
    for i in range(len(seq) - (n - 1)):
        m = str(seq[i:i + n]).upper()
        seen[m] = seen.get(m, 0) + 1
    return seen


def calc_contig_complexity(seq, N=3, w=6):                          # utils.py:20-30
    vec = []
    L = len(seq)
    for i in range(L):
        s, e = max(0, i - w), min(L, i + w)
        vec.append(round2(float(len(count_nmers(seq[s:e], N))) / float(e - s), 2))
    return sum(vec) / len(vec), vec


def check_intervals(breakpts, query_region):                        # utils.py:77-94
    inside = [False, [], []]
    spanning = [False, [], []]
    for bp in breakpts:
        for iv in query_region[4]:
            if iv[1] - 20 <= int(bp) <= iv[2] + 20:
                inside[0] = True
                inside[1].append(iv[4])
                inside[2].append(iv)
            if iv[2] <= max(breakpts) and iv[1] >= min(breakpts):
                spanning[0] = True
                spanning[1].append(iv[4])
                spanning[2].append(iv)
    return inside, spanning


def filter_by_feature(brkpts, query_region, keep_intron_vars):      # utils.py:58-74
    in_filter = span_filter = False
    if not keep_intron_vars:
        inside, spanning = check_intervals(brkpts, query_region)
        in_filter = not (inside[0] and 'exon' in inside[1])
        span_filter = not (spanning[0] and 'exon' in spanning[1])
    return in_filter, span_filter


# --------------------------------------------------------------------------- repeat mask
class blat_repeat_manager(object):                                  # sv_caller.py:838-875
    def __init__(self):
        self.breakpoint_in_rep = [False, False]
        self.total_rep_overlap = 0.0
        self.simple_rep_overlap = 0.0
        self.other_values = [False, 0.0, [], [False, False]]

    def setup(self, coords, repeat_locs):
        start, end = coords
        seg_len = float(end - start)
        hit = False
        total = simple = 0.0
        rep_coords = []
        edges = [False, False]
        for _chr, b1, b2, name in repeat_locs:
            if (start <= b1 <= end) or (start <= b2 <= end) or (b1 <= start and b2 >= end):
                hit = True
                ov = float(min(b2, end) - max(b1, start))
                total += ov
                rep_coords.append((b1, b2))
                if ")n" in name or "_rich" in name:                 # simple / low-complexity repeat
                    simple += ov
                    if b1 <= start <= b2:
                        edges[0] = True
                    elif b1 <= end <= b2:
                        edges[1] = True
        pct = round2(float(min(total, seg_len)) / float(seg_len) * 100, 2)
        self.total_rep_overlap = pct
        self.simple_rep_overlap = round2(float(min(simple, seg_len)) / float(seg_len) * 100, 2)
        self.breakpoint_in_rep = edges
        self.other_values = [hit, pct, rep_coords, edges]


# --------------------------------------------------------------------------- one PSL record
class blat_res(object):                                             # sv_caller.py:881-1142
    def __init__(self, res_d):
        v = list(res_d['blat_values'])
        self.blat_values = v
        self.matches = {'match': int(v[0]), 'mis': int(v[1]), 'rep': int(v[2])}
        self.gaps = {'hit': [int(v[6]), int(v[7])], 'query': [int(v[4]), int(v[5])]}
        tname = str(v[13]).replace('chr', '')
        if 'tname' in res_d:
            tname = res_d['tname']
            v[13] = tname
        off = res_d.get('offset', 0)
        tcoords = [off + int(v[15]), off + int(v[16])]
        v[15], v[16] = tcoords
        self.vals = {'hit': {'name': tname, 'size': int(v[14]), 'coords': tcoords},
                     'query': {'name': v[9], 'size': int(v[10]), 'coords': [int(v[11]), int(v[12])]}}
        self.strand = v[8]
        self.query_blocksizes = [int(x) for x in str(v[18]).rstrip(",").split(",")]
        tstarts = [off + int(x) for x in str(v[20]).rstrip(",").split(",")]
        v[20] = ",".join(str(x) for x in tstarts) + ","
        qstarts = [int(x) for x in str(v[19]).rstrip(",").split(",")]
        self.fragments = {'hit': [], 'query': []}
        for qs, ts, bs in zip(qstarts, tstarts, self.query_blocksizes):
            self.fragments['hit'].append((ts, ts + bs))
            self.fragments['query'].append((qs, qs + bs))
        self.fragments['count'] = len(self.query_blocksizes)
        self.genes = ''
        self.in_target = False
        self.valid = True
        self.rep_man = None
        self.in_repeat = False
        self.repeat_overlap = 0.0
        self.repeat_coords = None
        self.filter_reps_edges = [False, False]
        self.breakpts = []
        self.query_brkpts = []
        self.indel_sizes = []
        self.mean_cov = 0.0
        self.seg_overlap = [0, 0]
        self.cigar = ''
        self.indel_flank_match = [0, 0]
        if 'params' in res_d:
            self.set_gene_anno(res_d['params'].gene_annotations, res_d['query_region'])
            if 'repeat_mask' in res_d:
                self.set_repeat(res_d['repeat_mask'], res_d['params'].repeat_mask)
        self.set_indel_locs()
        self.perc_ident = 100.0 - self.calcMilliBad()

    # ---- plain accessors
    def get_coords(self, which): return self.vals[which]['coords']
    def qstart(self): return self.vals['query']['coords'][0]
    def qend(self): return self.vals['query']['coords'][1]
    def tstart(self): return self.vals['hit']['coords'][0]
    def tend(self): return self.vals['hit']['coords'][1]
    def get_name(self, which): return self.vals[which]['name']
    def get_size(self, which): return self.vals[which]['size']
    def get_query_span(self): return self.qend() - self.qstart()
    def get_query_coverage(self): return round2(float(self.get_query_span()) / float(self.get_size('query')) * 100, 2)
    def spans_query(self): return self.get_size('query') == self.qend() - self.qstart()
    def get_ngap_total(self): return self.gaps['hit'][1] + self.gaps['query'][1]
    def get_num_gaps(self): return self.gaps['hit'][0] + self.gaps['query'][0]
    def get_nmatch_total(self): return self.matches['match'] + self.matches['rep']
    def get_nmatches(self, which): return self.matches[which]
    def get_gene_anno(self): return self.genes
    def get_blat_output(self): return "\t".join(str(x) for x in self.blat_values)
    def set_segment_overlap(self, right, left): self.seg_overlap = [left, right]          # argument order as in the reference (:970)

    def calcMilliBad(self):                                          # :954-968 (BLAT's identity formula)
        q_ali = self.qend() - self.qstart()
        t_ali = self.tend() - self.tstart()
        if min(q_ali, t_ali) <= 0:
            return 0.0
        size_dif = max(q_ali - t_ali, 0)
        total = self.matches['match'] + self.matches['rep'] + self.matches['mis']
        bad = 0.0
        if total != 0:
            bad = (1000 * (self.matches['mis'] + self.gaps['query'][0] + round2(3 * math.log(1 + size_dif)))) / total
        return bad * 0.1

    def set_gene_anno(self, annotations, query_region):              # :1119-1142
        s, e = self.get_coords('hit')
        lo, hi = query_region[1] - 200, query_region[2] + 200
        if query_region[0] == self.get_name('hit') and (lo <= s <= hi or lo <= e <= hi):
            self.in_target = True
            self.genes = query_region[3]
            return
        chrom = self.get_name('hit')
        if 'chr' not in chrom:
            chrom = 'chr' + str(chrom)
        found = []
        for g, (gchrom, gs, ge) in annotations.genes.items():       # first match in table order
            if chrom == gchrom and gs <= int(s) <= ge:
                found.append(g)
                break
        if not found:
            found = ['intergenic']
            self.valid = False
        self.genes = ",".join(found)

    def set_repeat(self, target_rep_mask, all_rep_mask):             # :973-986
        self.rep_man = blat_repeat_manager()
        if self.matches['rep'] > 0:
            self.in_repeat = True
        if not self.in_repeat and target_rep_mask and all_rep_mask:
            rmask = target_rep_mask
            if not self.in_target:
                rmask = all_rep_mask.get(self.get_name('hit'))
            if rmask:
                self.rep_man.setup(self.get_coords('hit'), rmask)
                self.in_repeat, self.repeat_overlap, self.repeat_coords, self.filter_reps_edges = self.rep_man.other_values

    def _flank_m_sum(self, piece):                                   # :1022-1034
        total = 0
        for i, ch in enumerate(piece):
            if ch != "M":
                continue
            j, digits = i - 1, ''
            while j > -1 and _is_digit(piece[j]):
                digits = piece[j] + digits
                j -= 1
            total += int(digits)
        return total

    def add_query_brkpt(self, b):
        if b not in self.query_brkpts:
            self.query_brkpts.append(b)

    def set_indel_locs(self):                                        # :1044-1085
        q, t = self.fragments['query'], self.fragments['hit']
        sizes = []
        biggest = [0, '']
        for i in range(self.fragments['count'] - 1):
            if i == 0 and q[0][0] > 0:
                self.cigar = str(q[0][0]) + "S"
            ins_bp = q[i + 1][0] - q[i][1]
            del_bp = t[i + 1][0] - t[i][1]
            self.cigar += str(self.query_blocksizes[i]) + "M"
            if ins_bp > 0:
                self.breakpts.append([t[i][1]])
                sizes.append("I" + str(ins_bp))
                self.add_query_brkpt(q[i][1])
                self.add_query_brkpt(q[i + 1][0])
                self.cigar += str(ins_bp) + "I"
                if ins_bp > biggest[0]:
                    biggest = [ins_bp, "I"]
            if del_bp > 0:
                self.breakpts.append([t[i][1], t[i + 1][0]])
                sizes.append("D" + str(del_bp))
                self.add_query_brkpt(q[i][1])
                self.cigar += str(del_bp) + "D"
                if del_bp > biggest[0]:
                    biggest = [del_bp, "D"]
        self.cigar += str(self.query_blocksizes[-1]) + "M"
        tail = self.get_size('query') - self.qend()
        if tail > 0:
            self.cigar += str(tail) + "S"
        if biggest[0] > 0:                                           # flank matches around the largest event (:1036-1042)
            parts = self.cigar.split(str(biggest[0]) + biggest[1])
            self.indel_flank_match[0] += self._flank_m_sum(parts[0])
            self.indel_flank_match[1] += self._flank_m_sum(parts[-1])
        if sizes:
            self.indel_sizes.append(",".join(sizes))
        if self.strand == "-":
            self.query_brkpts = [self.get_size('query') - b for b in self.query_brkpts]

    def get_brkpt_str(self, with_sizes=False):                       # :1091-1103 (only the first breakpoint pairs with the size string)
        out = []
        if self.breakpts:
            text = ''
            for b, s in zip(self.breakpts, self.indel_sizes):
                text = 'chr' + str(self.get_name('hit')) + ":" + ("-".join(str(x) for x in b) if len(b) > 1 else str(b[0]))
                if with_sizes:
                    text += " (" + s + ")"
            out.append(text)
        return ",".join(out)

    def get_brkpt_locs(self):
        return [x for b in self.breakpts for x in b]


# --------------------------------------------------------------------------- event = one or more records
def brkpt_coverages(tbp, coverage_fn):
    """sv_event.get_brkpt_coverages (sv_caller.py:99-133): one count per breakpoint position of the
    `target_breakpoints` field; coverage_fn(chrom_without_chr, start, end) -> reads (None: 0)."""
    if "(" in tbp:
        tbp = tbp.split()[0]
    pts = []
    for bp in tbp.split(','):
        chrom, locs = bp.split(':')
        chrom = chrom.replace('chr', '')
        ll = locs.split('-')
        pts.append((chrom, int(ll[0]), int(ll[0]) + 1))
        if len(ll) > 1:
            pts.append((chrom, int(ll[1]), int(ll[1]) + 1))
    return ",".join(str(int(coverage_fn(c, s, e)) if coverage_fn else 0) for c, s, e in pts)


def bam_coverage_fn(bam):
    """The read filter of sv_caller.py:121-126 over `samio.Samfile.fetch`."""
    def cov(chrom, start, end):
        if hasattr(bam, "covered") and not bam.covered(chrom, start, end):     # outside the kept windows of a region-filtered reader: count in the file
            return bam.count_region(chrom, start, end, lambda flag, mapq: not (flag & 0x604) and mapq >= 10)
        return sum(1 for r in bam.fetch(str(chrom), start, end)
                   if not (r.is_duplicate or r.is_qcfail or r.is_unmapped or r.mapq < 10))
    return cov


class sv_event(object):                                             # sv_caller.py:13-517
    def __init__(self, br, query_region, contig_vals, sample_bam, coverage_fn=None):
        self.blat_res = []
        self.br_sorted = []
        self.sample_bam = sample_bam
        self.coverage_fn = coverage_fn
        self.qlen = 0
        self.nmatch = 0
        self.in_target = False
        self.query_region = query_region
        (self.contig_seq, self.contig_rcounts, self.contig_id, self.contig_reads, self.nkmers, self.contig_kmer_locs) = contig_vals[:6]
        self.logger = logging.getLogger('root')
        self.valid = True
        self.in_rep = False
        self.query_size = None
        self.query_cov = [0] * len(self.contig_seq)
        self.result_values = {'anno_genes': None, 'target_breakpoints': None, 'align_cigar': '', 'mismatches': 0, 'strands': None,
                              'repeat_matching': None, 'sv_type': None, 'split_read_count': None, 'nkmers': self.nkmers,
                              'disc_read_count': 0, 'contig_id': query_region[3] + "_" + self.contig_id, 'contig_seq': self.contig_seq,
                              'sv_subtype': None, 'breakpoint_coverages': 0}
        self.add(br)

    def add(self, br):                                               # :40-52
        qs, qe = br.get_coords('query')
        self.blat_res.append((qs, br))
        for i in range(qs, qe):
            self.query_cov[i] += 1
        if not self.query_size:
            self.query_size = br.get_size('query')
        self.qlen += br.get_query_span()
        self.nmatch += br.get_nmatch_total()
        self.in_target = self.in_target or br.in_target
        self.in_rep = self.in_rep and (br.repeat_overlap > 75.0)
        self.valid = self.valid and br.valid
        self.br_sorted.append((br, br.get_nmatch_total()))

    def check_previous_add(self, br):                                # :55-72
        nc = br.get_coords('query')
        prev, prev_n = self.br_sorted[-1]
        pc = prev.get_coords('query')
        if nc[0] == pc[0] and nc[1] == pc[1]:
            n = br.get_nmatch_total()
            if abs(prev_n - n) < 10 and not prev.in_target and br.in_target:
                self.br_sorted[-1] = (br, n)
                self.blat_res[-1] = (nc[0], br)
                self.in_target = True

    def format_result(self, values):                                 # :88-95
        if values:
            for key, val in values.items():
                if not isinstance(val, list):
                    val = [val]
                self.result_values[key] = ",".join(str(x) for x in val)
        if self.result_values['sv_subtype']:
            self.result_values['sv_type'] += '_' + self.result_values['sv_subtype']
        return self.get_values()

    def get_brkpt_coverages(self):                                   # :99-133
        return brkpt_coverages(self.result_values['target_breakpoints'], self.coverage_fn)

    def get_values(self):                                            # :137-146
        row = []
        for key in RESULT_FIELDS:
            if self.result_values[key] == 'trl':
                self.result_values[key] = 'rearrangement'
            row.append(str(self.result_values[key]))
            if key == 'target_breakpoints':
                self.result_values['breakpoint_coverages'] = self.get_brkpt_coverages()
        return row

    def get_indel_result(self):                                      # :150-160
        br = self.blat_res[0][1]
        rv = self.result_values
        rv['anno_genes'] = br.get_gene_anno()
        rv['repeat_matching'] = '0.0:' + str(br.get_nmatch_total())
        rv['mismatches'] = br.get_nmatches('mis')
        rv['strands'] = br.strand
        rv['target_breakpoints'] = br.get_brkpt_str(True)
        rv['align_cigar'] = br.cigar
        rv['sv_type'] = 'indel'
        rv['split_read_count'] = ",".join(str(self.contig_rcounts.get_counts(x, x, 'indel')) for x in br.query_brkpts)
        return self.format_result(None)

    def _brkpt_info(self, br, d, i, last):                           # :164-203
        ts, te = br.get_coords('hit')
        qs, qe = br.get_coords('query')
        d['chrs'].append(br.get_name('hit'))
        d['tcoords'].append((ts, te))
        rep_start = None
        if i == 0:
            d['q'][0] = [max(0, qs - 1), qe]
            d['q'][1].append([qe, qe - d['q'][0][0], None])
            pts = [ts] if br.strand == '-' else [te]
            rep_start = br.filter_reps_edges[0]
        elif last:
            d['q'][1][-1][2] = qe - d['q'][1][-1][0]
            d['q'][1].append([qs, qs - d['q'][0][0], qe - qs])
            pts = [te] if br.strand == '-' else [ts]
            rep_start = br.filter_reps_edges[1] if br.strand == '-' else br.filter_reps_edges[0]
        else:
            d['q'][1][-1][2] = qe - d['q'][1][-1][1]
            d['q'][1].append([qs, qs - d['q'][0][0], qe - qs])
            d['q'][1].append([qe, qe - qs, None])
            d['q'][0] = [qs, qe]
            pts = [ts, te]
            if br.strand == "-":
                rep_start = br.filter_reps_edges[1]
                pts = [te, ts]
        text = 'chr' + str(br.get_name('hit')) + ":" + "-".join(str(x) for x in pts)
        d['brkpt_str'].append(text)
        d['r'].extend(pts)
        d['f'].append(rep_start)
        d['t']['in_target' if br.in_target else 'other'] = (br.get_name('hit'), pts[0])
        d['formatted'].append(text)
        return d

    def _brkpt_counts(self, brkpts, sv_type):                        # :256-284
        avg, vec = calc_contig_complexity(self.contig_seq)
        rep_filt = False
        counts = {'n': [], 'd': [], 'b': []}
        kmers = []
        rc = self.contig_rcounts
        for qb in brkpts['q'][1]:
            left, right = qb[0] - min(qb[1], 5), qb[0] + min(qb[2], 5)
            counts['n'].append(min(rc.get_counts(left, right, sv_type)))
            counts['d'].append(min(rc.get_counts(qb[0] - 1, qb[0] + 1, sv_type)))
            counts['b'].append(rc.get_counts(qb[0], qb[0], sv_type))
            kmers.append(self.contig_kmer_locs[qb[0]])
            rep_filt = rep_filt or (vec[qb[0]] < (avg / 2))
        rep_filt = rep_filt or any(brkpts['f'])
        return counts, kmers, rep_filt

    def get_svs_result(self, query_region, params, disc_reads):       # :207-252
        ordered = sorted(self.blat_res, key=lambda x: x[0])
        brkpts = {'t': {'in_target': None, 'other': None}, 'formatted': [], 'r': [], 'q': [[0, 0], []], 'chrs': [], 'brkpt_str': [],
                  'tcoords': [], 'f': []}
        res = {'target_breakpoints': [], 'align_cigar': [], 'sv_type': '', 'strands': [], 'mismatches': [], 'repeat_matching': [],
               'anno_genes': [], 'disc_read_count': 0}
        all_valid, all_simple = True, True
        max_repeat = 0.0
        for i, (_qs, br) in enumerate(ordered):
            all_valid = all_valid and br.valid
            all_simple = all_simple and (br.rep_man.simple_rep_overlap > 75.0)
            max_repeat = max(max_repeat, br.repeat_overlap)
            res['repeat_matching'].append(":".join([str(br.repeat_overlap), str(br.get_nmatch_total()), str(round2(br.mean_cov, 3))]))
            res['anno_genes'].append(br.get_gene_anno())
            res['align_cigar'].append(br.cigar)
            res['strands'].append(br.strand)
            res['mismatches'].append(br.get_nmatches('mis'))
            brkpts = self._brkpt_info(br, brkpts, i, i == len(ordered) - 1)
        result = None
        self.br_sorted = sorted(self.br_sorted, key=lambda x: x[1])
        if not self.multiple_genes(brkpts['chrs'], brkpts['r'], res['anno_genes']):
            counts, kmers, _rep = self._brkpt_counts(brkpts, 'rearr')
            rtype, support = self.define_rearr(brkpts['r'], res['strands'], brkpts['tcoords'], disc_reads)
            if not self.filter_rearr(query_region, params, brkpts['r'], counts, kmers, rtype, support):
                res['sv_type'] = 'rearrangement'
                if rtype != 'rearrangement':
                    res['sv_subtype'] = rtype
                res['disc_read_count'] = support
                res['anno_genes'] = _unique_keep_set_order(res['anno_genes'])
                res['target_breakpoints'] = brkpts['brkpt_str']
                res['split_read_count'] = counts['b']
                if 'rearrangement' in params.opts['var_filter']:
                    result = self.format_result(res)
        elif max(self.contig_rcounts.others) >= params.get_sr_thresh('trl'):
            counts, kmers, rep_filt = self._brkpt_counts(brkpts, 'trl')
            disc = self.check_disc_reads(brkpts['t'], query_region, disc_reads['disc'])
            if not self.filter_trl([all_valid, all_simple], query_region, params, counts, kmers, disc, res['anno_genes'], max_repeat, rep_filt):
                res['disc_read_count'] = disc
                res['sv_type'] = ['trl']
                res['target_breakpoints'] = brkpts['brkpt_str']
                res['split_read_count'] = counts['b']
                if 'trl' in params.opts['var_filter']:
                    result = self.format_result(res)
        return result

    @staticmethod
    def _contained(a, b):                                            # check_overlap :317-323
        return (a[0] >= b[0] and a[1] <= b[1]) or (b[0] >= a[0] and b[1] <= a[1])

    def define_rearr(self, brkpts, strands, tcoords, disc_reads):     # :327-367
        kind, support, typed = 'rearrangement', 0, False
        if len(strands) < 3 and not self._contained(tcoords[0], tcoords[1]):
            if strands[0] != strands[1] and brkpts[0] < brkpts[1]:
                typed, kind = True, 'inversion'
                for p1, p2, s1, s2, _q in disc_reads['inv']:
                    if s1 == 1 and s2 == 1:
                        if p1 <= brkpts[0] and brkpts[0] <= p2 <= brkpts[1]:
                            support += 1
                    elif brkpts[0] <= p1 <= brkpts[1] and p2 >= brkpts[1]:
                        support += 1
            elif strands[0] == "+" and strands[1] == "+" and brkpts[0] > brkpts[1]:
                typed, kind = True, 'tandem_dup'                      # support test ends in `and ()`: never counts (Q13)
        if not typed:
            per = [0] * len(brkpts)
            for i, b in enumerate(brkpts):
                for p1, p2, _s1, _s2, _q in disc_reads['other']:
                    if abs(p1 - b) <= 300 or abs(p2 - b) <= 300:
                        per[i] += 1
            support = max(per)
        return kind, support

    def filter_rearr(self, query_region, params, brkpts, counts, kmers, rtype, disc):   # :371-379
        in_ff, span_ff = filter_by_feature(brkpts, query_region, params.opts['keep_intron_vars'])
        return (min(counts['n']) < params.get_sr_thresh('rearrangement')) or self.br_sorted[0][1] < params.get_min_segment_length('rearr') \
            or (in_ff and span_ff) or (disc < 1) or (rtype == 'rearrangement') or (min(kmers) == 0)

    def filter_trl(self, br_valid, query_region, params, counts, kmers, disc, anno_genes, max_repeat, rep_filt):   # :383-422
        drop = br_valid[1] or (max(counts['d']) < params.get_sr_thresh('trl'))
        if not drop and disc < 2:
            shortest, short_n = self.br_sorted[0]
            if short_n < params.get_min_segment_length('trl') or min(counts['n']) < params.get_sr_thresh('trl') or min(kmers) == 0 or rep_filt:
                drop = True
            elif disc == 0:
                checks = [self.minseq_complexity(self.contig_seq[shortest.qstart():shortest.qend()], 3) < 25.0,
                          self.missing_query_coverage() > 5.0,
                          short_n <= round2(float(len(self.contig_seq)) / 4.0),
                          max(shortest.seg_overlap) > 5,
                          max(shortest.gaps['query'][0], shortest.gaps['hit'][0]) > 0,
                          self.check_uniqueness(),
                          self.check_read_strands(),
                          'intergenic' in anno_genes]
                if sum(1 for c in checks if c) > 1:
                    drop = True
        return drop

    def check_uniqueness(self):                                      # :426-433
        return any((br.mean_cov > 10) if br.in_target else (br.mean_cov > 4) for br, _n in self.br_sorted)

    def check_read_strands(self):                                    # :437-447
        return len(set(r.id.split("/")[1] for r in self.contig_reads)) == 1

    def minseq_complexity(self, seq, N):                             # :451-459
        kinds = set(str(seq[i:i + N]).upper() for i in range(len(seq) - (N - 1)))
        return round2(float(len(kinds)) / float(len(seq) - 2) * 100, 4)

    def missing_query_coverage(self):                                # :463-479
        miss = 0
        for c in self.query_cov:
            if c:
                break
            miss += 1
        for c in reversed(self.query_cov):
            if c:
                break
            miss += 1
        return round2(float(miss) / float(len(self.contig_seq)) * 100, 4)

    def multiple_genes(self, chrs, brkpts, anno_genes):               # :483-492
        if len(set(anno_genes)) == 1:
            return False
        if self.dup_gene_names(anno_genes) and len(set(chrs)) == 1 and (max(brkpts) - min(brkpts)) < 10000:
            return False
        return True

    @staticmethod
    def dup_gene_names(genes):                                       # :496-503
        return any((g1.find(g2) > -1) or (g2.find(g1) > -1) for i, g1 in enumerate(genes[:-1]) for g2 in genes[i + 1:])

    def check_disc_reads(self, brkpts, query_region, disc_reads):     # :507-514
        n = 0
        other = brkpts['other']
        if other[0] in disc_reads:
            for p1, p2 in disc_reads[other[0]]:
                if abs(p1 - brkpts['in_target'][1]) <= 1000 and abs(p2 - other[1]) <= 1000:
                    n += 1
        return n


def _unique_keep_set_order(items):
    """list(set(x)) in the reference (sv_caller.py:237): only reached when the set has one element or
    similar names; a deterministic order (sorted) is used for the multi-name case."""
    u = set(items)
    return list(u) if len(u) == 1 else sorted(u)


# --------------------------------------------------------------------------- all records of one contig
class blat_manager(object):                                         # sv_caller.py:523-779
    def __init__(self, meta_dict):
        self.meta_dict = meta_dict
        self.fn = meta_dict.get('query_res_fn')
        self.qsize = 0
        self.hit_freq = []
        self.nmismatches = 0
        self.ngaps = 0
        self.has_blat_results = True
        self.blat_results = []
        self.clipped_qs = []
        self.se = None
        self.logger = logging.getLogger('root')
        self.set_values()

    def _lines(self):
        if self.meta_dict.get('psl_records') is not None:
            return [list(r) for r in self.meta_dict['psl_records']]
        if not self.fn:
            return []
        with open(self.fn) as f:
            return [ln.strip().split("\t") for ln in f.readlines()]

    def set_values(self):                                            # :569-596
        rows = self._lines()
        if not rows:
            self.has_blat_results = False
        for fields in rows:
            self.meta_dict['blat_values'] = fields
            br = blat_res(self.meta_dict)
            raw = br.get_nmatch_total()
            self.blat_results.append((raw + float(raw) / float(br.get_size('query')), br.get_ngap_total(), 1 if br.in_target else 0, br, br.perc_ident))
            self.nmismatches += br.get_nmatches('mis')
            self.ngaps += br.get_num_gaps()
            if not self.qsize:
                self.qsize = br.get_size('query')
                self.hit_freq = [0] * self.qsize
            for i in range(br.qstart(), br.qend()):
                self.hit_freq[i] += 1
        self.blat_results.sort(key=lambda x: (-x[0], -x[4], x[1]))

    def target_hit(self):                                            # :539-565
        return self.blat_results[0][3].spans_query() or (len(self.blat_results) == 1 and self.get_query_coverage() >= 90.0)

    def write_mod_result_file(self, fn):
        with open(fn, 'w') as f:
            for r in self.blat_results:
                f.write(r[3].get_blat_output() + "\n")

    def get_query_coverage(self):
        return round2(float(sum(1 for x in self.hit_freq if x > 0)) / float(self.qsize) * 100, 2)

    def get_mean_cov(self, s, e):
        return float(sum(self.hit_freq[s:e])) / float(len(self.hit_freq[s:e]))

    def _new_event(self, br):
        md = self.meta_dict
        return sv_event(br, md['query_region'], md['contig_vals'], md.get('sbam'), md.get('coverage_fn'))

    def check_blat_indel(self, br):                                  # :621-651
        md = self.meta_dict
        params = md['params']
        if not (br.spans_query() or (len(self.blat_results) == 1 and br.in_target)):
            return False
        keep = br.valid and br.mean_cov < 2 and br.in_target and br.get_ngap_total() >= int(params.opts['indel_size']) \
            and not br.rep_man.breakpoint_in_rep[0] and not br.rep_man.breakpoint_in_rep[1]
        if keep:
            cov = [md['contig_vals'][1].get_counts(x, x, 'indel') for x in br.query_brkpts]
            low_cov = min(cov) < params.get_sr_thresh('indel')
            flanks_ok = all(round2(float(fm) / float(br.get_size('query')) * 100, 2) >= 10.0 for fm in br.indel_flank_match)
            in_ff, _span = filter_by_feature(br.get_brkpt_locs(), md['query_region'], params.opts['keep_intron_vars'])
            if not in_ff and not low_cov and flanks_ok:
                self.se = self._new_event(br)
        return True

    def get_indel_result(self):
        return self.se.get_indel_result() if self.se else None

    def get_svs_result(self):
        md = self.meta_dict
        return self.se.get_svs_result(md['query_region'], md['params'], md['disc_reads']) if self.se else None

    def check_indels(self):                                          # :671-688
        for i, (_sc, _ng, _it, br, _pi) in enumerate(self.blat_results):
            br.mean_cov = self.get_mean_cov(br.qstart(), br.qend())
            if i == 0 and self.check_blat_indel(br):
                return True
            self.clipped_qs.append((br.qstart(), br.qend(), br, i))
        return False

    def check_svs(self):                                             # :692-711
        gaps = [(0, self.qsize)]
        if len(self.clipped_qs) > 1:
            best = [0, None]
            for i, cq in enumerate(self.clipped_qs):
                gaps = self.iter_gaps(gaps, cq, i)
                if self.se.qlen > best[0]:
                    best = [self.se.qlen, self.se]
            self.se = best[1]
        return self.se_valid()

    def se_valid(self):                                              # :715-721
        if self.se and len(self.se.blat_res) > 1 and self.se.in_target:
            return sum(1 for x in self.se.query_cov if x == 0) < self.meta_dict['params'].get_min_segment_length('trl')
        return False

    def check_add_br(self, qs, qe, gs, ge, br):                      # :725-741
        over = round2(float(min(qe, ge) - max(qs, gs)) / float(qe - qs) * 100)
        ov_right = abs(qe - ge) if qe > ge else 0
        ov_left = abs(qs - gs) if qs < gs else 0
        br.set_segment_overlap(ov_left, ov_right)
        return over >= 50 and (max(ov_right, ov_left) < 15 or (br.in_target and self.se.in_target))

    def iter_gaps(self, gaps, cq, it):                               # :745-777
        qs, qe, br, _idx = cq
        out, hit = [], False
        for gs, ge in gaps:
            if (gs <= qs <= ge) or (gs <= qe <= ge):
                rest = []
                if qs > gs and (qs - 1 - gs) > 10:
                    rest.append((gs, qs - 1))
                if qe < ge and (ge - qe + 1) > 10:
                    rest.append((qe + 1, ge))
                if it == 0:
                    self.se = self._new_event(br)
                    out.extend(rest)
                    hit = True
                elif self.check_add_br(qs, qe, gs, ge, br):
                    out.extend(rest)
                    self.se.add(br)
                    hit = True
                else:
                    out.append((gs, ge))
            else:
                out.append((gs, ge))
        if not hit:
            self.se.check_previous_add(br)
        return out


class align_manager(object):                                        # sv_caller.py:785-833
    def __init__(self, meta_dict):
        self.meta_dict = meta_dict
        self.query_res_fn = meta_dict.get('query_res_fn')
        self.logger = logging.getLogger('root')
        self.result = None
        self.bm = blat_manager(meta_dict)

    def check_target_results(self):                                  # :794-816
        if not self.bm.has_blat_results:
            self.query_res_fn = None
            return True, self.query_res_fn
        hit = False
        if self.query_res_fn:
            self.bm.write_mod_result_file(self.query_res_fn + '.mod')
        if self.bm.target_hit():
            hit = True
            if self.query_res_fn:
                self.query_res_fn += '.mod'
        return hit, self.query_res_fn

    def get_result(self):                                            # :820-832
        if self.bm.has_blat_results:
            if self.bm.check_indels():
                self.result = self.bm.get_indel_result()
            elif self.bm.check_svs():
                self.result = self.bm.get_svs_result()
        return self.result


# --------------------------------------------------------------------------- realign records -> PSL columns
def psl_fields(rec, qname, tname, t_offset=0, repeats_lower=True):
    """One realign record (dict from hip_backend.Engine.hits / oracle realign) -> the 21 PSL columns
    BLAT writes (sv_processor.py:843), target coordinates shifted by t_offset.  repeats_lower=False: the genome-wide
    gfClient call (sv_processor.py:840) has no -repeats=lower, so matches on soft-masked bases are plain matches there."""
    bs = ",".join(str(x) for x in rec["block_sizes"]) + ","
    qs = ",".join(str(x) for x in rec["q_starts"]) + ","
    ts = ",".join(str(x + t_offset) for x in rec["t_starts"]) + ","
    m, rp = (rec["matches"], rec["rep_matches"]) if repeats_lower else (rec["matches"] + rec["rep_matches"], 0)
    return [str(m), str(rec["mismatches"]), str(rp), str(rec["n_count"]),
            str(rec["q_num_insert"]), str(rec["q_base_insert"]), str(rec["t_num_insert"]), str(rec["t_base_insert"]),
            rec["strand"], qname, str(rec["q_size"]), str(rec["q_start"]), str(rec["q_end"]),
            tname, str(rec["t_size"]), str(rec["t_start"] + t_offset), str(rec["t_end"] + t_offset),
            str(len(rec["block_sizes"])), bs, qs, ts]
