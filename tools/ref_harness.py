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
