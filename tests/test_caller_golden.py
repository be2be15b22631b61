"""breakmer_amd.sv_caller (host-side call logic) against rows the REAL reference produced (G5)."""
import json
import os

from breakmer_amd import sv_assembly, sv_caller


class _Anno(object):
    def __init__(self, genes):
        self.genes = genes


class _Params(object):
    def __init__(self, opts, genes, repeat_mask):
        self.opts = dict(opts)
        self.gene_annotations = _Anno(genes)
        self.repeat_mask = repeat_mask

    def get_min_segment_length(self, kind):
        return int(self.opts[kind + '_minseg_len'])

    def get_sr_thresh(self, kind):
        if kind == 'min':
            return min(self.get_sr_thresh(x) for x in ('trl', 'rearrangement', 'indel'))
        return int(self.opts[{'trl': 'trl_sr_thresh', 'rearrangement': 'rearr_sr_thresh', 'indel': 'indel_sr_thresh'}[kind]])


def _tuplify(mask):
    if mask is None:
        return None
    if isinstance(mask, dict):
        return {k: [tuple(x) for x in v] for k, v in mask.items()}
    return [tuple(x) for x in mask]


def run_case(c):
    qr = c["query_region"]
    query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
    cd = c["contig"]
    reads = [sv_assembly.fq_read(i, "", "", False) for i in c["read_ids"]]
    ct = sv_assembly.contig(cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], cd["kmers"], reads, 31)
    d = c["disc_reads"]
    disc = {"disc": {k: [tuple(x) for x in v] for k, v in d["disc"].items()}, "inv": [tuple(x) for x in d["inv"]],
            "td": [tuple(x) for x in d["td"]], "other": [tuple(x) for x in d["other"]]}
    meta = {'params': _Params(c["opts"], c["genes"], _tuplify(c["all_repeat_mask"])), 'repeat_mask': _tuplify(c["target_repeat_mask"]),
            'query_region': query_region, 'psl_records': c["psl_rows"], 'disc_reads': disc,
            'contig_vals': (ct.get_contig_seq(), ct.get_contig_counts(), c["contig_id"], ct.reads, len(ct.kmers), ct.get_kmer_locs()),
            'sbam': None}
    if c["offset"] is not None:
        meta['offset'] = c["offset"]
    if c["tname"] is not None:
        meta['tname'] = c["tname"]
    am = sv_caller.align_manager(meta)
    hit = bool(am.bm.target_hit()) if am.bm.has_blat_results else None
    return am.get_result(), hit


def test_g5_caller_rows(golden_dir):
    with open(os.path.join(golden_dir, "caller.json")) as f:
        d = json.load(f)
    assert len(d["cases"]) >= 30
    called = 0
    for c in d["cases"]:
        got, hit = run_case(c)
        assert got == c["expected"], c["tag"]
        assert hit == c["target_hit"], c["tag"]
        called += got is not None
    assert called >= 15


def test_g5_native_call_tail(golden_dir):
    """The C++ port of the call logic (csrc/bk_call.h, host code of libbreakmer_hip.so; no GPU needed) against
    the same reference rows."""
    from breakmer_amd import call_context as cc, hip_backend as hb
    with open(os.path.join(golden_dir, "caller.json")) as f:
        d = json.load(f)
    called = 0
    for c in d["cases"]:
        qr = c["query_region"]
        query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
        cd = c["contig"]
        tags = set(i.split("/")[1] for i in c["read_ids"])
        lines = [cc.opts_line(c["opts"])] + cc.tables_lines(c["genes"], _tuplify(c["all_repeat_mask"]))
        lines += cc.region_lines(0, query_region, _tuplify(c["target_repeat_mask"]), c["disc_reads"])
        lines += cc.contig_lines(c["contig_id"], cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], len(cd["kmers"]), len(tags) == 1,
                                 c["psl_rows"], c["offset"], c["tname"])
        got, hit = hb.call_text("\n".join(lines) + "\n")
        assert got == c["expected"], c["tag"]
        assert hit == c["target_hit"], c["tag"]
        called += got is not None
    assert called >= 15


def test_g8_realign_contract_evidence(golden_dir):
    """R2 (BLAT parity unpinned): contigs with small indels / mismatch clusters next to the SV junction.  The fixture holds
    (A) the records of this build's realign contract and (B) BLAT-style records built from the known edit script, each with
    the row the REAL reference's caller made of them.  Checked here: the oracle still produces (A); the Python and the
    native call tail reproduce the reference's rows for both; and (A) gives the same call as (B) in every case but the one
    whose inserted bases can be placed one position either way (same breakpoints, CIGAR differs)."""
    from breakmer_amd import call_context as cc, hip_backend as hb
    from oracle import bk_oracle as bo
    from breakmer_amd import synth
    with open(os.path.join(golden_dir, "realign_evidence.json")) as f:
        d = json.load(f)
    r = synth.make_region(**d["region"])
    same = 0
    for c in d["cases"]:
        assert bo.realign(c["contig"]["seq"], [r.window_str]) == c["gapfree_chain"]["records"], c["tag"]
        for label in ("gapfree_chain", "blat_style"):
            genome = c.get(label + "_genome_rows", False)
            case = {"query_region": d["query_region"], "contig": c["contig"], "read_ids": c["read_ids"], "disc_reads": d["disc_reads"],
                    "opts": d["opts"], "genes": d["genes"], "all_repeat_mask": None, "target_repeat_mask": None, "psl_rows": c[label]["psl_rows"],
                    "contig_id": "contig1", "offset": None if genome else d["offset"], "tname": None if genome else d["tname"]}
            got, _hit = run_case(case)
            assert got == c[label]["expected"], (c["tag"], label)
            qr = d["query_region"]
            query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
            cd = c["contig"]
            lines = [cc.opts_line(d["opts"])] + cc.tables_lines(d["genes"], None) + cc.region_lines(0, query_region, None, d["disc_reads"])
            lines += cc.contig_lines("contig1", cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], len(cd["kmers"]),
                                     len(set(i.split("/")[1] for i in c["read_ids"])) == 1, c[label]["psl_rows"], case["offset"], case["tname"])
            assert hb.call_text("\n".join(lines) + "\n")[0] == c[label]["expected"], (c["tag"], label, "native")
        a, b = c["gapfree_chain"]["expected"], c["blat_style"]["expected"]
        assert a is not None and b is not None and a[1] == b[1] and a[6] == b[6] == "indel", c["tag"]       # same breakpoints, same call
        same += a == b
    assert same >= len(d["cases"]) - 1


def test_g8m_multi_mapping_contigs(golden_dir):
    """R2 steps 5-6 (secondary alignments, placement of ambiguous hits): BLAT prints every alignment >= -minScore and the
    reference's caller counts them per query base.  The fixture holds, per case, the records of this build's contract and
    BLAT-style records written down from the construction, each with the row the REAL reference's align_manager made of them:
    both flanks duplicated -> mean_cov 2 -> the indel is filtered (sv_caller.py:631); a repeated partner half -> low uniqueness
    (:430-432) filters a translocation without discordant pairs and shows as mean_cov 6.0 in the row of one with; the partner
    half also present in the target window -> check_previous_add (:55-72) turns the event into an in-target rearrangement;
    25 bases copied from elsewhere into a deletion junction: with one index tile (BLAT's -stepSize=10 -minMatch=2 cannot seed
    it) the contig is ONE record with an insertion and a deletion, with two tiles it is also a record of its own -- and the
    chain of the two flanks passes over it.
    Checked here: the oracle still produces the contract records; the Python and the native call tail reproduce the
    reference's rows for both record sets; both record sets give the same row in every case."""
    from breakmer_amd import call_context as cc, hip_backend as hb
    from oracle import bk_oracle as bo
    with open(os.path.join(golden_dir, "realign_multihit.json")) as f:
        d = json.load(f)
    tags = {c["tag"]: c for c in d["cases"]}
    assert tags["del_both_flanks_dup"]["contract"]["expected"] is None and tags["del_unique"]["contract"]["expected"] is not None
    assert tags["del_left_flank_dup"]["contract"]["expected"] == tags["del_unique"]["contract"]["expected"]
    assert tags["trl_partner_repeat_nodisc"]["contract"]["expected"] is None and tags["trl_unique_nodisc"]["contract"]["expected"] is not None
    assert tags["trl_partner_repeat_disc"]["contract"]["expected"][5].endswith(":6.0")
    assert tags["trl_partner_half_also_in_target"]["contract"]["expected"] is None
    # contract step 8 (BLAT's -minIdentity default of 90 %, -minScore): a diverged copy of the left flank at ~80 % identity -- a
    # +1/-2 segment runs through it -- is not a record (BLAT would not print it: hit_freq stays 1), at ~93 % it is, with its mismatches
    lo, hi = tags["del_left_flank_diverged_copy_80pct"], tags["del_left_flank_diverged_copy_93pct"]
    assert len(lo["contract"]["records"]) == 1 and lo["contract"]["max_hit_freq"] == 1 and "AC" not in lo["tag"]
    assert len(hi["contract"]["records"]) == 2 and hi["contract"]["max_hit_freq"] == 2
    sec = hi["contract"]["records"][1]
    assert sec["mismatches"] >= 8 and 100 * sec["matches"] >= 90 * (sec["matches"] + sec["mismatches"])
    for c in d["cases"]:
        assert bo.realign(c["contig"]["seq"], c["targets"]) == c["contract"]["records"], c["tag"]
        assert c["contract"]["max_hit_freq"] == c["blat_style"]["max_hit_freq"], c["tag"]
        for label in ("contract", "blat_style"):
            e = c[label]
            case = {"query_region": c["query_region"], "contig": c["contig"], "read_ids": c["read_ids"], "disc_reads": c["disc_reads"],
                    "opts": d["opts"], "genes": c["genes"], "all_repeat_mask": None, "target_repeat_mask": None, "psl_rows": e["psl_rows"],
                    "contig_id": "contig1", "offset": e["offset"], "tname": e["tname"]}
            got, hit = run_case(case)
            assert got == e["expected"], (c["tag"], label)
            qr = c["query_region"]
            query_region = (qr[0], qr[1], qr[2], qr[3], [tuple(x) for x in qr[4]])
            cd = c["contig"]
            lines = [cc.opts_line(d["opts"])] + cc.tables_lines(c["genes"], None) + cc.region_lines(0, query_region, None, c["disc_reads"])
            lines += cc.contig_lines("contig1", cd["seq"], cd["indel_only"], cd["others"], cd["kmer_locs"], len(cd["kmers"]),
                                     len(set(i.split("/")[1] for i in c["read_ids"])) == 1, e["psl_rows"], e["offset"], e["tname"])
            assert hb.call_text("\n".join(lines) + "\n")[0] == e["expected"], (c["tag"], label, "native")
        a, b = c["contract"]["expected"], c["blat_style"]["expected"]
        if c["tag"] == "del_templated_insert_unseedable":
            # the hand-written BLAT-style record ends its blocks exactly at the planted edits; the contract's segments run on over
            # chance matches next to them (as any extending aligner does): the same call, breakpoints a few bases apart
            assert a is not None and b is not None and a[6] == b[6] == "indel" and a[0] == b[0]
            pa, pb = int(a[1].split(":")[1].split()[0]), int(b[1].split(":")[1].split()[0])
            assert abs(pa - pb) <= 6 and "I" in a[2] and "D" in a[2] and len(c["contract"]["records"]) == 1, (a[1], b[1])
        else:
            assert a == b, c["tag"]
