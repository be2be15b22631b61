"""TEST INFRASTRUCTURE -- drive the *real* reference (via oracle/ref_loader.py) on synthetic
regions.  Runs only where /root/reference exists (this container).  Used by
tools/make_golden.py to produce tests/golden/*.json and by ad-hoc parity probes.
"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_loader  # noqa: E402


def ref_kmer_select(read_strs, ref_strs, k, sc_strs=None):
    """Jellyfish 1.1.11 `count -m k` (no -C) + `dump -c` semantics restated (binary absent;
    SURVEY 8c 'parity unpinned, low risk') feeding the reference's own set algebra
    (sv_processor.py:613-631, executed here literally on dicts)."""
    def count(seqs):
        d = {}
        for s in seqs:
            for i in range(len(s) - k + 1):
                m = s[i:i + k]
                if set(m) <= set("ACGT"):
                    d[m] = d.get(m, 0) + 1
        return d
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    ref = {}
    for r in ref_strs:
        for s in (r, "".join(comp.get(c, "N") for c in reversed(r))):
            for m, c in count([s]).items():
                ref[m] = ref.get(m, 0) + c
    case = count(read_strs)
    case_sc = case if sc_strs is None else count(sc_strs)
    sc_mers = set(case.keys()) & set(case_sc)
    sample_only = list(sc_mers.difference(set(ref.keys())))
    return {m: case[m] for m in sample_only}


def jellyfish_standin(fa_fn, _jellyfish, kmer_size):
    """utils.run_jellyfish (utils.py:151-178) with the absent Jellyfish 1.1.11 binary replaced by a counter that writes the
    file the reference then reads back with ITS OWN load_kmers (utils.py:287-296): `jellyfish count -m k` (no -C: strand
    specific, every position, k-mers with a non-ACGT base skipped) + `dump -c` = one "<mer> <count>" line per k-mer."""
    seqs = []
    with open(fa_fn) as f:
        lines = [ln.rstrip("\n") for ln in f]
    if lines and lines[0].startswith("@"):
        seqs = lines[1::4]
    else:
        cur = []
        for ln in lines:
            if ln.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            else:
                cur.append(ln.strip())
        if cur:
            seqs.append("".join(cur))
    d = {}
    for s_ in seqs:
        for i in range(len(s_) - kmer_size + 1):
            m = s_[i:i + kmer_size]
            if not m.strip("ACGT"):
                d[m] = d.get(m, 0) + 1
    dump_fn = fa_fn + "_%dmers_dump" % kmer_size
    with open(dump_fn, "w") as f:
        for m, c in d.items():
            f.write("%s %d\n" % (m, c))
    return dump_fn



def ref_compare_kmers(read_ids, read_strs, window, k, sc_strs=None, indel_only=None, opts=None):
    """The REAL target.compare_kmers (sv_processor.py:609-645) on files written the way the reference has them at that
    point; run_jellyfish -> jellyfish_standin (binary absent), init_assembly replaced by a recorder (the assembler is
    pinned separately).  Returns (case_only dict as handed to init_assembly, recorded arguments)."""
    import shutil
    import tempfile
    mods = ref_loader.load()
    sp, ut = mods["sv_processor"], mods["utils"]
    base = tempfile.mkdtemp()
    o = dict(DEFAULT_OPTS)
    o.update({"keep_repeat_regions": True, "jellyfish": "jellyfish", "kmer_size": k})
    if opts:
        o.update(opts)
    params = StubParams(o, {}, None)
    params.paths = {"targets": os.path.join(base, "targets"), "ref_data": os.path.join(base, "ref"), "output": os.path.join(base, "output")}
    params.get_kmer_size = lambda: k
    for p_ in params.paths.values():
        os.makedirs(p_, exist_ok=True)
    t = sp.target([("1", 1000, 1000 + len(window) - 400, "TGT", "exon")], params)
    os.makedirs(t.paths['ref_data'], exist_ok=True)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    with open(t.files['target_ref_fn'][0], "w") as f:
        f.write(">TGT\n" + window + "\n")
    with open(t.files['target_ref_fn'][1], "w") as f:
        f.write(">TGT\n" + "".join(comp.get(c, "N") for c in reversed(window)) + "\n")
    t.files['cleaned_fq'] = os.path.join(t.paths['data'], "TGT_cleaned.fastq")
    with open(t.files['cleaned_fq'], "w") as f:
        for rid, s in zip(read_ids, read_strs):
            f.write(rid + "\n" + s + "\n+\n" + "I" * len(s) + "\n")
    t.files['sv_sc_unmapped_fa'] = os.path.join(t.paths['data'], "TGT_sc.fa")
    with open(t.files['sv_sc_unmapped_fa'], "w") as f:
        for i, s in enumerate(sc_strs if sc_strs is not None else read_strs):
            f.write(">s%d\n%s\n" % (i, s))
    fq = OrderedDict()
    for i, (rid, s) in enumerate(zip(read_ids, read_strs)):
        fq.setdefault(s, []).append(ut.fq_read(rid, s, "I" * len(s), bool(indel_only[i]) if indel_only is not None else False))
        t.read_len = max(t.read_len, len(s))
    t.cleaned_read_recs = fq
    cap = {}

    def recorder(mers, fq_recs, kmer_len, rc_thresh, read_len):
        cap.update({"mers": dict(mers), "kmer_len": kmer_len, "rc_thresh": rc_thresh, "read_len": read_len, "n_recs": len(fq_recs)})
        return []
    saved_j, saved_a = sp.run_jellyfish, sp.init_assembly
    sp.run_jellyfish, sp.init_assembly = jellyfish_standin, recorder
    try:
        t.compare_kmers()
    finally:
        sp.run_jellyfish, sp.init_assembly = saved_j, saved_a
        shutil.rmtree(base, ignore_errors=True)
    return cap["mers"], cap


def ref_init_assembly(read_ids, read_strs, mers, k, rc_thresh, indel_only=None):
    """Call the reference's init_assembly (sv_assembly.py:30) with fq_recs in FASTQ order (P4)."""
    mods = ref_loader.load()
    ut, sa = mods["utils"], mods["sv_assembly"]
    fq = OrderedDict()
    read_len = 0
    for i, (rid, s) in enumerate(zip(read_ids, read_strs)):
        io = bool(indel_only[i]) if indel_only is not None else False
        fr = ut.fq_read(rid, s, "I" * len(s), io)
        read_len = max(read_len, len(s))
        fq.setdefault(s, []).append(fr)
    contigs = sa.init_assembly(dict(mers), fq, k, rc_thresh, read_len)
    idx = {rid: i for i, rid in enumerate(read_ids)}
    out = []
    for c in contigs:
        out.append({"seq": c.aseq.seq,
                    "indel_only": list(c.aseq.counts.indel_only),
                    "others": list(c.aseq.counts.others),
                    "kmer_locs": list(c.kmer_locs),
                    "kmers": [t[0] for t in c.kmers],
                    "reads": sorted(idx[r.id] for r in c.reads)})
    return out, contigs


# ---------------------------------------------------------------------------------------------
# reference caller (sv_caller.align_manager) on explicit PSL columns
class StubAnno(object):
    def __init__(self, genes):
        self.genes = genes


class StubParams(object):
    """What sv_caller reads from utils.params (utils.py:535-674)."""

    def __init__(self, opts, genes, repeat_mask=None):
        self.opts = dict(opts)
        self.gene_annotations = StubAnno(genes)
        self.repeat_mask = repeat_mask

    def get_min_segment_length(self, kind):
        return int(self.opts[kind + '_minseg_len'])

    def get_sr_thresh(self, kind):
        if kind == 'min':
            return min(self.get_sr_thresh('trl'), self.get_sr_thresh('rearrangement'), self.get_sr_thresh('indel'))
        return int(self.opts[{'trl': 'trl_sr_thresh', 'rearrangement': 'rearr_sr_thresh', 'indel': 'indel_sr_thresh'}[kind]])


DEFAULT_OPTS = {'indel_size': 15, 'trl_sr_thresh': 2, 'indel_sr_thresh': 5, 'rearr_sr_thresh': 3, 'rearr_minseg_len': 30,
                'trl_minseg_len': 25, 'keep_intron_vars': False, 'var_filter': ['indel', 'rearrangement', 'trl'],
                'keep_repeat_regions': False, 'sample_bam_file': None}


def ref_call(psl_rows, ref_contig, contig_id, query_region, opts, genes, disc_reads, target_repeat_mask=None, all_repeat_mask=None,
             offset=None, tname=None):
    """sv_caller.align_manager(meta_dict).get_result() of the REAL reference on explicit PSL rows
    (21 columns each) for an assembled reference contig object; returns the 13-field row or None."""
    import tempfile
    mods = ref_loader.load()
    sc = mods["sv_caller"]
    params = StubParams(opts, genes, all_repeat_mask)
    with tempfile.NamedTemporaryFile("w", suffix=".psl", delete=False) as f:
        for row in psl_rows:
            f.write("\t".join(str(x) for x in row) + "\n")
        fn = f.name
    meta = {'params': params, 'repeat_mask': target_repeat_mask, 'query_region': query_region, 'query_res_fn': fn,
            'disc_reads': disc_reads,
            'contig_vals': (ref_contig.aseq.seq, ref_contig.aseq.counts, contig_id, ref_contig.reads, len(ref_contig.kmers), ref_contig.kmer_locs),
            'sbam': None}
    if offset is not None:
        meta['offset'] = offset
    if tname is not None:
        meta['tname'] = tname
    try:
        am = sc.align_manager(meta)
        res = am.get_result()
        # state useful for debugging / target_hit
        return res, am
    finally:
        os.unlink(fn)
