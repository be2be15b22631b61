"""Generate tests/golden/*.json from the REAL reference (imported from /root/reference through
oracle/ref_loader.py).  Runs only in the build container; the fixtures (data: inputs or
generator parameters + expected outputs) are committed, the reference never travels.

  PYTHONHASHSEED=0 python tools/make_golden.py [g1 g2 g3 g4 g5 g6]
"""
import hashlib
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from breakmer_amd import synth  # noqa: E402
from oracle import ref_loader  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
META = {"generator": "tools/make_golden.py", "reference": "ccgd-profile/BreaKmer @ /root/reference",
        "patches": ref_loader.PATCH_IDS, "python": sys.version.split()[0]}


def dump(name, obj):
    obj = dict(obj)
    obj["_meta"] = META
    path = os.path.join(GOLD, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


def rs(rnd, n, alphabet="ACGT"):
    return "".join(rnd.choice(alphabet) for _ in range(n))


def g1():
    """G1: olc.nw known-answer tests, all 7 return fields (olc.py:40-107)."""
    olc = ref_loader.load()["olc"]
    rnd = random.Random(20260102)
    cases = []

    def add(a, b, tag):
        cases.append({"tag": tag, "seq1": a, "seq2": b, "out": list(olc.nw(a, b))})
    add("AAAAAAAA", "CCCCCCCC", "no_overlap_Q5")
    add("ACGTACGT", "ACGTACGT", "identical")
    add("ACGTTGCA", "TGCAGGGG", "dovetail")
    add("TTTTACGT", "ACGTACGTACGT", "last_column_tie")
    add("GGGACGTACGTTT", "ACGTACG", "read_inside")
    add("ACGTACG", "GGGACGTACGTTT", "contig_inside")
    add("A", "A", "one")
    add("A", "C", "one_mis")
    base = rs(rnd, 80)
    add(base, base[40:] + rs(rnd, 40), "suffix_prefix_40")
    add(base, base[40:60] + "T" + base[61:] + rs(rnd, 30), "internal_mismatch")
    add(base, base[40:60] + base[62:] + rs(rnd, 30), "gap_in_read")
    add(base, base[40:60] + "GG" + base[60:] + rs(rnd, 30), "gap_in_contig")
    for t in range(160):
        m, n = rnd.randint(1, 70), rnd.randint(1, 70)
        a = rs(rnd, m)
        mode = rnd.random()
        if mode < 0.55:
            ov = rnd.randint(1, min(m, 50))
            b = a[m - ov:] + rs(rnd, max(0, n - ov))
            b = "".join((c if rnd.random() > 0.06 else rnd.choice("ACGT")) for c in b)
        elif mode < 0.75:
            b = rs(rnd, n, "AC")
            a = rs(rnd, m, "AC")
        else:
            b = rs(rnd, n)
        add(a, b, "rand%d" % t)
    for t, (m, n) in enumerate([(150, 150), (298, 150), (150, 298), (600, 250), (250, 600), (900, 250), (31, 900)]):
        a = rs(rnd, m)
        ov = min(m, n) * 2 // 3
        b = a[m - ov:] + rs(rnd, n - ov)
        b = "".join((c if rnd.random() > 0.02 else rnd.choice("ACGT")) for c in b)
        add(a, b, "long%d" % t)
        add(b, a, "long%d_swapped" % t)
    dump("nw_kats.json", {"cases": cases})


class _FakeRead(object):
    def __init__(self, rid, seq, indel_only):
        self.id, self.seq, self.qual, self.used, self.dup, self.indel_only = rid, seq, "I" * len(seq), False, False, indel_only


def g2():
    """G2: contig.check_align branch coverage (sv_assembly.py:449-546) on hand-built contig/read pairs.
    State after the call: match, seq, counts, kmers (grow mode)."""
    sa = ref_loader.load()["sv_assembly"]
    rnd = random.Random(77)
    k = 15
    cases = []
    base = rs(rnd, 400)

    def run(tag, contig_seq, read_seq, mer, mode, nreads=3, indel_only=False, founder_nreads=2, pre=None):
        founder = _FakeRead("@f/1_0", contig_seq, False)
        ct = sa.contig(mer, founder, contig_seq.find(mer), founder_nreads, k)
        skm = set()
        for s in (contig_seq, read_seq):
            for i in range(len(s) - k + 1):
                skm.add(s[i:i + k])
        if pre is not None:       # extra earlier read to de-synchronise counts length from seq length (Q8)
            ct.check_align(_FakeRead("@p/1_0", pre, False), mer, 1, skm, mode)
        if mode == "grow":
            ct.set_kmers(skm)
        rd = _FakeRead("@r/1_0", read_seq, indel_only)
        m = ct.check_align(rd, mer, nreads, skm, mode)
        cases.append({"tag": tag, "k": k, "contig": contig_seq, "read": read_seq, "mer": mer, "mode": mode,
                      "nreads": nreads, "indel_only": indel_only, "founder_nreads": founder_nreads, "pre": pre,
                      "match": bool(m), "seq": ct.aseq.seq, "io": list(ct.aseq.counts.indel_only),
                      "ot": list(ct.aseq.counts.others), "kmers": [list(t) for t in ct.kmers]})
    c = base[100:250]
    mer = c[60:75]
    for mode in ("setup", "grow"):
        run("no_match", c, rs(rnd, 150), mer, mode)
        run("identical_Q9", c, c, mer, mode)
        run("read_right_overhang", c, base[130:280], base[135:150], mode)
        run("read_left_overhang", c, base[70:220], base[135:150], mode)
        run("read_left_overhang_indel_only", c, base[70:220], base[135:150], mode, indel_only=True)
        run("contig_inside_longer_read", c, base[80:270], mer, mode)
        run("read_inside_contig", c, base[120:230], base[135:150], mode)
        run("read_inside_contig_indel_only", c, base[120:230], base[135:150], mode, indel_only=True)
        run("same_len_shift_tiebreak", c, base[101:251], mer, mode)
        mm = list(base[70:220]); mm[100] = "A" if mm[100] != "A" else "C"
        run("left_overhang_with_mismatch", c, "".join(mm), base[135:150], mode)
        gp = base[130:200] + base[202:282]
        run("right_overhang_gap", c, gp, base[135:150], mode)
        # superseq whose aligned span is longer than the old contig (gap on contig side): Q8 shrink
        longer = base[80:160] + "GT" + base[160:270]
        run("superseq_Q8_gap", c, longer, mer, mode)
        run("superseq_after_pre", c, base[60:300], mer, mode, pre=base[90:240])
        # low identity overlap (<0.90) but long
        noisy = "".join((ch if rnd.random() > 0.2 else rnd.choice("ACGT")) for ch in base[130:280])
        run("low_identity", c, noisy, base[135:150], mode)
        # periodic sequence: equal scores both ways, k-mer position tie-break (sv_assembly.py:485-496)
        per = "ACGTTGCAAT" * 15
        run("periodic_tie", per, per[3:] + "ACG", per[20:35], mode)
        run("periodic_tie2", per, "TTG" + per[:147], per[20:35], mode)
    dump("check_align.json", {"cases": cases})


def region_case(tag, k=31, rc_thresh=2, **kw):
    r = synth.make_region(**kw)
    reads = r.read_strs()
    t = time.time()
    mers, cap = rh.ref_compare_kmers(r.read_ids, reads, r.window_str, k, None, r.indel_only)       # the reference's own compare_kmers (+ load_kmers)
    assert cap["kmer_len"] == k and cap["read_len"] == max(len(x) for x in reads)
    assert mers == rh.ref_kmer_select(reads, [r.window_str], k)                                    # the restated set algebra agrees
    contigs, _ = rh.ref_init_assembly(r.read_ids, reads, mers, k, rc_thresh, r.indel_only)
    dt = time.time() - t
    h = hashlib.sha256(("\n".join(reads)).encode()).hexdigest()
    print("  %-28s reads %5d mers %6d contigs %3d  %.1fs" % (tag, len(reads), len(mers), len(contigs), dt))
    msum = hashlib.sha256(("\n".join("%s %d" % (m, mers[m]) for m in sorted(mers))).encode()).hexdigest()
    return {"tag": tag, "k": k, "rc_thresh": rc_thresh, "gen": kw, "reads_sha256": h, "n_mers": len(mers),
            "mers_sha256": msum, "mers": ({m: mers[m] for m in sorted(mers)} if len(mers) <= 200 else None),
            "contigs": contigs, "ref_seconds": round(dt, 2)}


def g3():
    """G3: init_assembly (sv_assembly.py:30-63) end to end on seeded synthetic regions.  Inputs are
    regenerated from `gen` (breakmer_amd.synth.make_region) and checked by sha256."""
    cases = []
    for i, sv in enumerate(synth.SV_TYPES):
        if sv == "trl":
            continue
        cases.append(region_case("d60_" + sv, region_id=10 + i, sv_type=sv, depth=60, W=1500))
    cases.append(region_case("d30_trl", region_id=15, sv_type="trl", depth=30, W=900))
    cases.append(region_case("config1_del_500x", region_id=0, sv_type="del", depth=500, W=3000))
    cases.append(region_case("L250_k41_ins", k=41, region_id=21, sv_type="ins", depth=40, W=1500, L=250))
    cases.append(region_case("L250_k41_noise5", k=41, region_id=22, sv_type="del", depth=24, W=1000, L=250, noise=0.05))
    cases.append(region_case("L250_k41_noise5_del_d40", k=41, region_id=23, sv_type="del", depth=40, W=700, L=250, noise=0.05))      # >= 1 contig at k=41 / 5 %
    cases.append(region_case("L250_k41_noise5_ins_d60", k=41, region_id=24, sv_type="ins", depth=60, W=700, L=250, noise=0.05))
    for j, nz in enumerate((0.01, 0.02, 0.05)):
        cases.append(region_case("noise%d_del" % int(nz * 100), region_id=30 + j, sv_type="del", depth=30, W=900, noise=nz))
    cases.append(region_case("noise2_inv_d60", region_id=34, sv_type="inv", depth=60, W=900, noise=0.02))
    cases.append(region_case("varlen_indelonly_ins", region_id=7, sv_type="ins", depth=60, W=1500, var_len=0.4, indel_only_frac=0.3, noise=0.005))
    cases.append(region_case("varlen_dup_rc3", rc_thresh=3, region_id=8, sv_type="dup", depth=80, W=1500, var_len=0.5, noise=0.01))
    cases.append(region_case("no_sv", region_id=9, sv_type="del", sv_size=0, depth=40, W=900))
    # reads with N calls (kept by the reference, utils.py:203-246; N == N is a match in olc.nw; no k-mer spans an N)
    cases.append(region_case("nreads_del", region_id=40, sv_type="del", depth=60, W=1500, n_frac=0.15))
    cases.append(region_case("nreads_ins_noise_varlen", region_id=41, sv_type="ins", depth=60, W=1200, n_frac=0.3, noise=0.01, var_len=0.3))
    cases.append(region_case("nreads_inv_k21", k=21, region_id=42, sv_type="inv", depth=40, W=1000, n_frac=0.5, noise=0.005))
    dump("assembly.json", {"cases": cases})


def g4():
    """G4: k-mer set algebra incl. a separate soft-clip set, produced by the REAL target.compare_kmers + utils.load_kmers
    (sv_processor.py:609-645, utils.py:287-296); only the absent Jellyfish binary is replaced (rh.jellyfish_standin)."""
    rnd = random.Random(4)
    cases = []
    for t in range(6):
        k = rnd.choice([5, 7, 11])
        ref = rs(rnd, 120)
        reads = [ref[s:s + 40] for s in (rnd.randint(0, 80) for _ in range(25))]
        reads += [rs(rnd, 40) for _ in range(3)] + ["A" * 40] + [reads[0]] * 2
        sc = None if t % 2 == 0 else [r for r in reads if rnd.random() < 0.5]
        d, cap = rh.ref_compare_kmers(["@S:1:1:4:%d/1_0" % i for i in range(len(reads))], reads, ref, k, sc)
        assert cap["rc_thresh"] == 2 and cap["kmer_len"] == k and cap["read_len"] == 40
        assert d == rh.ref_kmer_select(reads, [ref], k, sc)
        cases.append({"k": k, "ref": ref, "reads": reads, "sc": sc, "mers": {m: d[m] for m in sorted(d)}})
    dump("kmer_select.json", {"cases": cases})


def g5():
    """G5: sv_caller.align_manager(meta_dict).get_result() (sv_caller.py:785-833) on explicit PSL rows.
    Contigs come from the reference's own init_assembly; PSL rows are the build's realign records
    (oracle contract) and hand-edited variants; the expected 13-field rows come from the reference."""
    from oracle import bk_oracle as bo
    from breakmer_amd import sv_caller as my
    cases = []

    def add(tag, r, sv_rows_fn=None, opts=None, genes_extra=None, drop_partner_gene=False, disc=None, trm=None, arm=None,
            features='exon', indel_mode=None, noise_seed=None, soft=None):
        reads = r.read_strs()
        mers = rh.ref_kmer_select(reads, [r.window_str], 31)
        cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
        targets = [r.window_str] + [synth.codes_to_str(p[4]) for p in r.partners]
        if soft:                 # a soft-masked (lower-case) stretch of the target window: BLAT -repeats=lower reports matches on it as repMatches (sv_processor.py:843)
            targets[0] = targets[0][:soft[0]] + targets[0][soft[0]:soft[1]].lower() + targets[0][soft[1]:]
        tinfo = [(r.chrom, r.start - 200)] + [(p[0], p[1]) for p in r.partners]
        qr = (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, features)])
        genes = {r.name: ['chr' + r.chrom, r.start, r.end]}
        if not drop_partner_gene:
            for p_ in r.partners:
                genes[p_[3]] = ['chr' + p_[0], p_[1], p_[2]]
        if genes_extra:
            genes.update(genes_extra)
        o = dict(rh.DEFAULT_OPTS)
        if opts:
            o.update(opts)
        d = r.disc_reads if disc is None else disc
        for ci, (cd, co) in enumerate(list(zip(cdicts, cobjs))[:5]):
            recs = bo.realign(cd['seq'], targets)
            if indel_mode:       # the offset-shifted '.mod' path of check_target_blat (sv_processor.py:855-859): window coordinates + offset/tname override
                rows = [my.psl_fields(x, 'contig1', r.name, 0) for x in recs if x['t_index'] == 0]
                offset, tname = r.start - 200, r.chrom
            else:                # whole-genome style rows: genome coordinates and chr names (Q14)
                rows = [my.psl_fields(x, 'contig1', 'chr' + tinfo[x['t_index']][0], tinfo[x['t_index']][1], repeats_lower=False) for x in recs]      # the genome-wide gfClient call has no -repeats=lower
                offset, tname = None, None
            if sv_rows_fn:
                rows = sv_rows_fn(rows)
            res, am = rh.ref_call(rows, co, 'contig%d' % (ci + 1), qr, o, genes, d, trm, arm, offset, tname)
            hit = bool(am.bm.target_hit()) if am.bm.has_blat_results else None
            cases.append({"tag": "%s_c%d" % (tag, ci), "psl_rows": rows, "contig": cd, "read_ids": sorted(x.id for x in co.reads),
                          "contig_id": 'contig%d' % (ci + 1), "query_region": [qr[0], qr[1], qr[2], qr[3], [list(x) for x in qr[4]]],
                          "opts": o, "genes": genes, "disc_reads": {"disc": {k: [list(x) for x in v] for k, v in d["disc"].items()},
                                                                    "inv": [list(x) for x in d["inv"]], "td": [list(x) for x in d["td"]],
                                                                    "other": [list(x) for x in d["other"]]},
                          "target_repeat_mask": trm, "all_repeat_mask": arm, "offset": offset, "tname": tname,
                          "target_hit": hit, "expected": res})
            print("  %-34s rows %d -> %s" % (cases[-1]["tag"], len(rows), (res[:2] + [res[6]]) if res else None))

    mk = lambda i, sv, **kw: synth.make_region(i, sv_type=sv, depth=60, W=1500, **kw)
    add("del", mk(3, "del"))
    add("del_indelmode", mk(3, "del"), indel_mode=True)
    add("ins", mk(3, "ins"))
    add("ins_indelmode", mk(5, "ins"), indel_mode=True)
    add("inv_disc", mk(3, "inv"))
    add("inv_nodisc", mk(3, "inv"), disc={"disc": {}, "inv": [], "td": [], "other": []})
    add("dup", mk(3, "dup"))
    add("trl", mk(3, "trl"))
    add("trl_intergenic_partner", mk(3, "trl"), drop_partner_gene=True)
    add("trl_nodisc", mk(3, "trl"), disc={"disc": {}, "inv": [], "td": [], "other": []})
    add("del_small_indel_size", mk(6, "del", sv_size=10))
    add("del_size_thresh0", mk(6, "del", sv_size=10), opts={"indel_size": 0})
    add("del_intron", mk(3, "del"), features='intron')
    add("del_intron_keep", mk(3, "del"), features='intron', opts={"keep_intron_vars": True})
    add("del_var_filter_trl_only", mk(3, "del"), opts={"var_filter": ["trl"]})
    add("inv_var_filter_indel_only", mk(3, "inv"), opts={"var_filter": ["indel"]})
    add("del_noise2", mk(9, "del", noise=0.02))
    add("ins_noise1", mk(9, "ins", noise=0.01))
    r = mk(3, "del")
    g0 = r.start - 200
    simple = [(r.chrom, g0 + 640, g0 + 660, "(CA)n")]
    alu = [(r.chrom, g0 + 300, g0 + 620, "AluY")]
    add("del_simple_repeat_at_brkpt", r, trm=simple, arm={r.chrom: simple})
    add("del_alu_far", r, trm=alu, arm={r.chrom: alu})
    r = mk(3, "trl")
    p0 = r.partners[0]
    prep = [(p0[0], p0[1] + 700, p0[1] + 1500, "GA_rich")]
    add("trl_partner_simple_repeat", r, trm=[(r.chrom, g0 + 10, g0 + 30, "L1")], arm={r.chrom: [(r.chrom, g0 + 10, g0 + 30, "L1")], p0[0]: prep})
    # '-' strand indel: reverse-complement the contig rows by feeding a window that is the rc (hand edit: flip strand of a del row)
    def flip(rows):
        out = []
        for row in rows:
            row = list(row)
            row[8] = '-' if row[8] == '+' else '+'
            out.append(row)
        return out
    add("del_minus_strand_rows", mk(3, "del"), sv_rows_fn=flip)
    add("no_rows", mk(3, "del"), sv_rows_fn=lambda rows: [])
    # soft-masked windows: rows with repMatches > 0 (sv_caller.py:913; :975-986: such a record is in_repeat and SKIPS the repeat-mask overlap)
    r = mk(3, "del")
    c_ = 750
    g0 = r.start - 200
    simple_r = [(r.chrom, g0 + 640, g0 + 660, "(CA)n")]
    add("del_soft_left_flank_indelmode", r, indel_mode=True, soft=(c_ - 160, c_ - 100))
    add("del_soft_left_flank_rmask_indelmode", r, indel_mode=True, soft=(c_ - 160, c_ - 100), trm=simple_r, arm={r.chrom: simple_r})
    add("del_rmask_indelmode", r, indel_mode=True, trm=simple_r, arm={r.chrom: simple_r})
    add("del_soft_whole_window_indelmode", r, indel_mode=True, soft=(0, 1500))
    add("del_soft_left_flank_genome_rows", r, soft=(c_ - 160, c_ - 100))
    add("ins_soft_at_brkpt_indelmode", mk(5, "ins"), indel_mode=True, soft=(c_ - 20, c_ + 20))
    add("single_partial_hit", mk(3, "del"), sv_rows_fn=lambda rows: [[str(x) for x in ["120", "0", "0", "0", "0", "0", "0", "0", "+", "contig1", "297", "0", "120", "chr4", "1500", "100502", "100622", "1", "120,", "0,", "100502,"]]])
    dump("caller.json", {"cases": cases})


def toy_cutadapt(fq_text, sv_reads_clips):
    """Deterministic stand-in for the adapter-trimming step between extraction and get_fastq_reads: every 5th
    record loses 18 bases at its 3' end, every 7th loses its leading soft-clip exactly (when it has one)."""
    lines = fq_text.split("\n")
    out = []
    for n, i in enumerate(range(0, len(lines) - 3, 4)):
        h, s, q = lines[i], lines[i + 1], lines[i + 3]
        clips = sv_reads_clips.get("_".join(h.lstrip("@").split("_")[:-1]))
        if n % 7 == 3 and clips and clips[0] and s.startswith(clips[0]):
            s, q = s[len(clips[0]):], q[len(clips[0]):]
        elif n % 5 == 2:
            s, q = s[:-18], q[:-18]
        out += [h, s, "+", q]
    return "\n".join(out) + "\n"


def g6():
    """N2: read extraction (sv_processor.py:12-93,422-583) and get_fastq_reads (utils.py:203-246) of the real
    reference, fed by breakmer_amd.samio records through a patched `Samfile`."""
    import logging
    import tempfile
    import types
    from breakmer_amd import samio
    mods = ref_loader.load()
    sp, ut = mods["sv_processor"], mods["utils"]
    cases = []
    tmp = tempfile.mkdtemp()
    for tag, rid, svt, size, npairs, k in [("del120", 3, "del", 120, 180, 15), ("ins60", 5, "ins", 60, 180, 15),
                                           ("del40_k31", 8, "del", 40, 200, 31)]:
        r = synth.make_region(rid, W=1200, L=100, depth=5, sv_type=svt, sv_size=size)
        sam = synth.make_sam(r, npairs)
        fn = os.path.join(tmp, tag + ".sam")
        open(fn, "w").write(sam)

        class Writer(object):
            def write(self, _r): pass
            def close(self): pass
        sp.Samfile = lambda path, mode="rb", **kw: samio.Samfile(path) if mode == "rb" else Writer()
        sp.sort = sp.index = lambda *a, **kw: None
        T = sp.target
        me = types.SimpleNamespace()
        me.params = types.SimpleNamespace(opts={"sample_bam_file": fn}, get_kmer_size=lambda: k)
        me.paths = {"data": tmp}
        me.files = {}
        me.name = tag
        me.chrom, me.start, me.end = r.chrom, r.start, r.end
        me.logger = logging.getLogger("g6")
        for meth in ("setup_read_extraction_files", "check_pair_overlap", "check_overlap"):
            setattr(me, meth, types.MethodType(getattr(T, meth), me))
        T.extract_bam_reads(me)
        fq = open(me.files["sv_fq"]).read()
        fa = open(me.files["sv_sc_unmapped_fa"]).read()
        clips = {q: (v[1]["clipped"] if v[1] else None) for q, v in me.sv_reads.items()}
        sv_meta = {q: [hashlib.sha1(v[0].seq.encode()).hexdigest()[:10], v[1], v[2], bool(v[3])] for q, v in me.sv_reads.items()}
        exp = {"fastq": fq, "sc_fasta": fa, "disc_reads": me.disc_reads, "sv_reads": sv_meta, "sv_order": list(me.sv_reads)}
        for variant, text in (("as_extracted", fq), ("trimmed", toy_cutadapt(fq, clips))):
            cfn = os.path.join(tmp, tag + "_" + variant + ".fastq")
            open(cfn, "w").write(text)
            _ffn, fq_recs, read_len = ut.get_fastq_reads(cfn, me.sv_reads)
            kept = [ln for ln in open(_ffn).read().split("\n")[0::4] if ln]
            flags = {}
            for seq, lst in fq_recs.items():
                for fr in lst:
                    flags[fr.id] = bool(fr.indel_only)
            exp[variant] = {"cleaned": text if variant == "trimmed" else None, "kept": kept, "read_len": read_len, "indel_only": [flags[h] for h in kept]}
        # sv_event.get_brkpt_coverages (sv_caller.py:99-133) on the same alignment file
        sc = mods["sv_caller"]
        sc.Samfile = sp.Samfile
        c0 = r.start - 200 + len(r.window) // 2
        cov = {}
        for tbp in ["chr%s:%d-%d (D120)" % (r.chrom, c0 - 60, c0 + 60), "chr%s:%d" % (r.chrom, c0), "chr%s:%d,chr%s:%d-%d" % (r.chrom, c0 - 300, r.chrom, c0 + 5, c0 + 400),
                    "chr%s:%d" % (r.chrom, r.start - 5000), "chr%d:%d" % (1 + (int(r.chrom) + 4) % 22, 5010)]:
            ev = types.SimpleNamespace(result_values={"target_breakpoints": tbp}, sample_bam=fn)
            cov[tbp] = sc.sv_event.get_brkpt_coverages(ev)
        exp["brkpt_coverages"] = cov
        cases.append({"tag": tag, "region": {"region_id": rid, "W": 1200, "L": 100, "depth": 5, "sv_type": svt, "sv_size": size},
                      "n_pairs": npairs, "kmer": k, "sam_sha1": hashlib.sha1(sam.encode()).hexdigest(), "expected": exp})
        print(tag, len(me.sv_reads), fq.count("\n") // 4, [len(exp[v]["kept"]) for v in ("as_extracted", "trimmed")])
    dump("read_extraction.json", {"cases": cases})


# ---------------------------------------------------------------------------------------------------------------
def _read_tree(base):
    out = {}
    for dp, _dn, fns in os.walk(base):
        for fn in fns:
            full = os.path.join(dp, fn)
            out[os.path.relpath(full, base)] = open(full).read()
    return out


def g7():
    """G7 (K2, R1, N1): the REAL reference's per-target surface driven end to end on synthetic regions --
    target.compare_kmers (sv_processor.py:609-645: run_jellyfish -> load_kmers -> set algebra -> init_assembly),
    target.resolve_sv (:648-665: contig.__init__/setup/write_* :731-782, query_ref :823-859, make_calls :863-866,
    write_result :791-799), get_summary (:708-721), write_results (:668-683) and runner.write_output (:212-234).
    Only what is absent from this image is stood in: the Jellyfish binary (a counter that writes its dump file), the BLAT
    binary (contig.run_blat writes the PSL text of the build's realign records) and pysam (write_bam is skipped).
    Expected: the 13-field rows, the summary lines and every output file's bytes (files whose line order comes from
    CPython set iteration are stored with their records sorted and flagged)."""
    import logging
    import shutil
    import tempfile
    from collections import OrderedDict
    from oracle import bk_oracle as bo
    from breakmer_amd import sv_caller as my
    mods = ref_loader.load()
    sp, ut = mods["sv_processor"], mods["utils"]
    sp.run_jellyfish = rh.jellyfish_standin
    cur = {}

    def run_blat(self, db, name):                      # sv_processor.py:835-851 with the binary replaced
        r = cur["r"]
        targets = [r.window_str] + [synth.codes_to_str(p_[4]) for p_ in r.partners]
        tinfo = [(r.chrom, r.start - 200)] + [(p_[0], p_[1]) for p_ in r.partners]
        recs = bo.realign(self.contig_seq, targets)
        if name == "target":
            rows = [my.psl_fields(x, 'contig1', r.name, 0) for x in recs if x['t_index'] == 0]
        else:
            rows = [my.psl_fields(x, 'contig1', 'chr' + tinfo[x['t_index']][0], tinfo[x['t_index']][1]) for x in recs]
        self.query_res_fn = os.path.join(self.path, 'blat_res.' + name + '.psl')
        with open(self.query_res_fn, "w") as f:
            for row in rows:
                f.write("\t".join(str(x) for x in row) + "\n")
        if not rows:
            self.query_res_fn = self.query_res_fn           # empty file: blat_manager reports no results
    sp.contig.run_blat = run_blat
    sp.contig.write_bam = lambda self, bam_in, path: None
    cases = []
    specs = [("del_nosc", dict(region_id=3, sv_type="del", depth=60, W=1500), 31, None),
             ("ins_sc_half", dict(region_id=5, sv_type="ins", depth=60, W=1500), 31, 2),
             ("inv_disc", dict(region_id=7, sv_type="inv", depth=60, W=1500), 31, None),
             ("dup", dict(region_id=9, sv_type="dup", depth=60, W=1500), 31, 3),
             ("trl", dict(region_id=11, sv_type="trl", depth=40, W=1200), 31, None),
             ("del_k15_noise", dict(region_id=13, sv_type="del", depth=40, W=1000, noise=0.01, var_len=0.3, indel_only_frac=0.2), 15, 2),
             ("nosv", dict(region_id=15, sv_type="del", sv_size=0, depth=30, W=900), 31, None)]
    base = tempfile.mkdtemp()
    opts = dict(rh.DEFAULT_OPTS)
    opts.update({"keep_repeat_regions": True, "analysis_name": "g7", "no_output_header": False, "jellyfish": "jellyfish", "kmer_size": 31,
                 "blat": "blat", "gfclient": "gfClient", "blat_port": 0, "reference_fasta": "", "reference_fasta_dir": ""})
    genes = {}
    regions = []
    for tag, kw, k, sc_mod in specs:
        r = synth.make_region(**kw)
        regions.append((tag, kw, k, sc_mod, r))
        genes[r.name] = ['chr' + r.chrom, r.start, r.end]
        for p_ in r.partners:
            genes[p_[3]] = ['chr' + p_[0], p_[1], p_[2]]
    runs = {}                                                                   # one run (analysis) per k-mer size
    for tag, kw, k, sc_mod, r in sorted(regions, key=lambda x: x[4].name):       # runner.run loops the sorted target names (:175-176)
        params = rh.StubParams(dict(opts, kmer_size=k), genes, None)
        params.paths = {"targets": os.path.join(base, "targets"), "ref_data": os.path.join(base, "ref"), "output": os.path.join(base, "output")}
        params.get_kmer_size = lambda k=k: k
        for p_ in params.paths.values():
            os.makedirs(p_, exist_ok=True)
        t = sp.target([(r.chrom, r.start, r.end, r.name, 'exon')], params)
        cur["r"] = r
        reads = r.read_strs()
        quals = ["".join(chr(33 + 20 + ((i * 7 + j) % 20)) for j in range(len(s_))) for i, s_ in enumerate(reads)]
        sc = None if sc_mod is None else [s_[:60] for i, s_ in enumerate(reads) if i % sc_mod == 0]
        os.makedirs(t.paths['ref_data'], exist_ok=True)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        with open(t.files['target_ref_fn'][0], "w") as f:
            f.write(">" + r.name + "\n" + r.window_str + "\n")
        with open(t.files['target_ref_fn'][1], "w") as f:
            f.write(">" + r.name + "\n" + "".join(comp[c] for c in reversed(r.window_str)) + "\n")
        t.files['cleaned_fq'] = os.path.join(t.paths['data'], r.name + "_sv_reads_cleaned.fastq")
        with open(t.files['cleaned_fq'], "w") as f:
            for rid, s_, q in zip(r.read_ids, reads, quals):
                f.write(rid + "\n" + s_ + "\n+\n" + q + "\n")
        t.files['sv_sc_unmapped_fa'] = os.path.join(t.paths['data'], r.name + "_sv_sc_seqs.fa")
        with open(t.files['sv_sc_unmapped_fa'], "w") as f:
            for i, s_ in enumerate(sc if sc is not None else reads):            # case_sc := case when no separate set is given
                f.write(">s%d\n%s\n" % (i, s_))
        t.files['sv_bam_sorted'] = None
        fq = OrderedDict()
        for i, (rid, s_, q) in enumerate(zip(r.read_ids, reads, quals)):       # utils.get_fastq_reads :239-244 in FASTQ order (P4)
            fq.setdefault(s_, []).append(ut.fq_read(rid, s_, q, bool(r.indel_only[i])))
            t.read_len = max(t.read_len, len(s_))
        t.cleaned_read_recs = fq
        t.disc_reads = r.disc_reads
        t.repeat_mask = None
        t0 = time.time()
        t.compare_kmers()
        t.resolve_sv()
        summary_header, summ = t.get_summary()
        run = runs.setdefault(k, {"results": [], "summary": {}, "header": ""})
        run["summary"][t.name] = summ
        run["header"] = summary_header
        if t.has_results():
            t.write_results()
            run["results"].extend(t.results)
        files = _read_tree(os.path.join(base, "targets", r.name))
        files.update({os.path.join("@output", k2): v for k2, v in _read_tree(t.paths['output']).items()})
        canon = {}
        for fn, text in files.items():
            if fn.endswith("_sample_kmers.out"):                                # list(set) order (sv_processor.py:622)
                canon[fn] = ["sorted_lines", "\n".join(sorted(text.split("\n")))]
            elif fn.endswith(".fq") and "/contigs/" in "/" + fn:                # `for read in self.reads` over a set (:777)
                recs = text.split("\n")
                recs = sorted("\n".join(recs[i:i + 4]) for i in range(0, len(recs) - 3, 4))
                canon[fn] = ["sorted_records", "\n".join(recs)]
            elif fn.endswith("_sample_kmers_merged.out"):                       # third line: read ids of a set (:763)
                ln = text.split("\n")
                ln[2] = ",".join(sorted(ln[2].split(",")))
                canon[fn] = ["sorted_read_ids", "\n".join(ln)]
            elif fn.endswith(".psl") or fn.endswith(".mod") or fn.endswith("_dump") or fn.endswith(".fastq") or fn.endswith("_sc_seqs.fa"):
                continue                                                       # inputs / stand-in artefacts, not the reference's writers
            else:
                canon[fn] = ["bytes", text]
        cases.append({"tag": tag, "gen": kw, "k": k, "sc_mod": sc_mod, "name": r.name, "rows": t.results, "summary": summ,
                      "n_clusters": len(t.kmers['clusters']), "files": canon})
        print("  %-16s k=%d clusters %d rows %d files %d  %.1fs" % (tag, k, len(t.kmers['clusters']), len(t.results), len(canon), time.time() - t0))
    # run level: runner.write_output (sv_processor.py:212-234) on a runner object that only carries what it reads
    run_files = {}
    for k, run in sorted(runs.items()):
        rn = object.__new__(sp.runner)
        rn.params = rh.StubParams(dict(opts, analysis_name="g7k%d" % k), genes, None)
        rn.params.paths = {"output": os.path.join(base, "output")}
        rn.results, rn.summary, rn.summary_header, rn.logger = run["results"], run["summary"], run["header"], logging.getLogger("g7")
        rn.write_output()
        run_files[str(k)] = {fn: text for fn, text in _read_tree(os.path.join(base, "output")).items() if os.sep not in fn and fn.startswith("g7k%d_" % k)}
    dump("surface.json", {"cases": cases, "run_files": run_files, "opts": {k_: v for k_, v in opts.items()},
                          "genes": genes, "note": "quals: chr(33+20+((i*7+j)%20)); sc: reads[i][:60] for i % sc_mod == 0; one analysis (g7k<k>) per k-mer size"})
    shutil.rmtree(base, ignore_errors=True)


def g8():
    """G8 (R2 evidence, SURVEY 8c "BLAT parity unpinned"): what the REAL reference's caller reports when a contig carries
    small indels / clustered mismatches close to the SV junction, from (A) the records of this build's realign contract
    (iterated gap-free segments with score >= 20, chained: a flank shorter than 20 bases between two differences cannot
    anchor and is lost) and from (B) BLAT-style records built from the known edit script (every indel splits a block,
    mismatches stay inside blocks, the alignment runs to the contig ends).  Both row sets come from the reference's
    align_manager; the fixture records them side by side with the fields that differ."""
    import copy
    from oracle import bk_oracle as bo
    from breakmer_amd import sv_caller as my
    r = synth.make_region(3, sv_type="del", depth=60, W=1500)
    reads = r.read_strs()
    mers = rh.ref_kmer_select(reads, [r.window_str], 31)
    cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
    base_seq = cdicts[0]["seq"]
    W = r.window_str
    J = W.find(base_seq[:60]) + 0                       # window position of contig base 0
    # contig = W[J : c-100] + W[c+100 : ...]: junction after `jq` contig bases
    c = len(W) // 2
    jq = (c - 100) - J
    assert base_seq == W[J:c - 100] + W[c + 100:c + 100 + len(base_seq) - jq], "unexpected contig layout"
    qr = (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, 'exon')])
    genes = {r.name: ['chr' + r.chrom, r.start, r.end]}
    o = dict(rh.DEFAULT_OPTS)
    comp = {"A": "C", "C": "G", "G": "T", "T": "A"}

    def variant(tag, edits):
        """edits: list of (contig position, kind, arg) applied right to left on the contig; kinds: 'sub' (arg = count of
        consecutive substitutions every 4th base), 'del' (arg = bases removed from the contig = bases present only in the
        window), 'ins' (arg = string inserted into the contig)."""
        co = copy.deepcopy(cobjs[0])
        seq = list(base_seq)
        io, ot, kl = list(co.aseq.counts.indel_only), list(co.aseq.counts.others), list(co.kmer_locs)
        # truth alignment as (contig index -> window index) built while editing: start from the plain deletion contig
        wpos = [J + i if i < jq else (c + 100) + (i - jq) for i in range(len(seq))]
        for pos, kind, arg in sorted(edits, key=lambda e: -e[0]):
            if kind == 'sub':
                for t in range(arg):
                    q = pos + 4 * t
                    seq[q] = comp[seq[q]]
            elif kind == 'del':
                del seq[pos:pos + arg]; del io[pos:pos + arg]; del ot[pos:pos + arg]; del kl[pos:pos + arg]; del wpos[pos:pos + arg]
            else:
                seq[pos:pos] = list(arg); io[pos:pos] = [io[pos]] * len(arg); ot[pos:pos] = [ot[pos]] * len(arg)
                kl[pos:pos] = [kl[pos]] * len(arg); wpos[pos:pos] = [None] * len(arg)
        cs = "".join(seq)
        co.aseq.seq = cs
        co.aseq.counts.indel_only, co.aseq.counts.others, co.kmer_locs = io, ot, kl
        # (B) BLAT-style record from the truth: maximal runs of consecutive (contig, window) pairs on one diagonal
        blocks = []
        i = 0
        while i < len(cs):
            if wpos[i] is None:
                i += 1
                continue
            j = i
            while j + 1 < len(cs) and wpos[j + 1] is not None and wpos[j + 1] == wpos[j] + 1:
                j += 1
            blocks.append((i, wpos[i], j - i + 1))
            i = j + 1
        mism = sum(1 for (q0, t0, ln) in blocks for z in range(ln) if cs[q0 + z] != W[t0 + z])
        match = sum(ln for _q, _t, ln in blocks) - mism
        qni = sum(1 for a, b in zip(blocks, blocks[1:]) if b[0] > a[0] + a[2]); qbi = sum(b[0] - a[0] - a[2] for a, b in zip(blocks, blocks[1:]))
        tni = sum(1 for a, b in zip(blocks, blocks[1:]) if b[1] > a[1] + a[2]); tbi = sum(b[1] - a[1] - a[2] for a, b in zip(blocks, blocks[1:]))
        rec = {"matches": match, "mismatches": mism, "rep_matches": 0, "n_count": 0, "q_num_insert": qni, "q_base_insert": qbi, "t_num_insert": tni,
               "t_base_insert": tbi, "strand": "+", "q_size": len(cs), "q_start": blocks[0][0], "q_end": blocks[-1][0] + blocks[-1][2], "t_index": 0,
               "t_size": len(W), "t_start": blocks[0][1], "t_end": blocks[-1][1] + blocks[-1][2], "block_sizes": [b[2] for b in blocks],
               "q_starts": [b[0] for b in blocks], "t_starts": [b[1] for b in blocks], "score": match - 2 * mism}
        out = {"tag": tag, "contig": {"seq": cs, "indel_only": io, "others": ot, "kmer_locs": kl, "kmers": cdicts[0]["kmers"], "reads": cdicts[0]["reads"]},
               "read_ids": sorted(x.id for x in co.reads), "edits": [list(e) for e in edits]}
        for label, recs in (("gapfree_chain", bo.realign(cs, [W])), ("blat_style", [rec])):
            rows = [my.psl_fields(x, 'contig1', r.name, 0) for x in recs if x['t_index'] == 0]
            res, am = rh.ref_call(rows, co, 'contig1', qr, o, genes, r.disc_reads, None, None, r.start - 200, r.chrom)
            if res is None and not am.bm.target_hit():          # Q14: the reference would go to the whole genome: genome-coordinate rows
                rows = [my.psl_fields(x, 'contig1', 'chr' + r.chrom, r.start - 200) for x in recs]
                res, am = rh.ref_call(rows, co, 'contig1', qr, o, genes, r.disc_reads, None, None, None, None)
                out[label + "_genome_rows"] = True
            out[label] = {"records": recs, "psl_rows": rows, "expected": res}
        a, b_ = out["gapfree_chain"]["expected"], out["blat_style"]["expected"]
        out["same_row"] = a == b_
        out["differing_fields"] = None if (a is None or b_ is None) else [i for i in range(13) if a[i] != b_[i]]
        print("  %-30s same=%s diff=%s\n      A %s\n      B %s" % (tag, out["same_row"], out["differing_fields"], a and a[:7], b_ and b_[:7]))
        return out

    cases = [variant("plain_deletion", []),
             variant("del2_10bp_left_of_junction", [(jq - 10, 'del', 2)]),
             variant("ins3_12bp_right_of_junction", [(jq + 12, 'ins', "GAT")]),
             variant("mismatch_cluster_at_junction", [(jq + 3, 'sub', 3)]),
             variant("del1_25bp_from_contig_end", [(len(base_seq) - 25, 'del', 1)]),
             variant("ins5_and_mismatches_both_sides", [(jq - 14, 'ins', "ACGTA"), (jq + 8, 'sub', 2)]),
             variant("del4_30bp_left_of_junction", [(jq - 30, 'del', 4)])]
    dump("realign_evidence.json", {"cases": cases, "region": {"region_id": 3, "sv_type": "del", "depth": 60, "W": 1500}, "opts": o, "genes": genes,
                                   "query_region": [qr[0], qr[1], qr[2], qr[3], [list(x) for x in qr[4]]],
                                   "disc_reads": {"disc": {}, "inv": [], "td": [], "other": []}, "offset": r.start - 200, "tname": r.chrom})


def g8m():
    """G8m (R2, multi-mapping contigs): BLAT prints EVERY alignment >= -minScore (sv_processor.py:843) and the caller counts
    them per query base -- hit_freq (sv_caller.py:593-594), mean_cov (:616) in the indel keep test (:631), in the
    repeat_matching field (:224) and in check_uniqueness (:430-432), check_previous_add (:55-72).  Each case holds (A) the
    records of this build's realign contract (steps 1-6 of oracle/bk_oracle.h, incl. secondary alignments and the placement
    of ambiguous hits) and (B) BLAT-style records written down from how the sequences were constructed, with the row the
    REAL reference's align_manager makes of each."""
    import copy
    from oracle import bk_oracle as bo
    from breakmer_amd import sv_caller as my
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda x: "".join(comp[ch] for ch in reversed(x))
    o = dict(rh.DEFAULT_OPTS)
    cases = []

    def block_rec(cs, targets, blocks, strand, tidx):
        """PSL-equivalent record of ungapped blocks (strand-coordinate query start, target start, length) on one target"""
        q = cs if strand == '+' else rc(cs)
        t = targets[tidx]
        mism = sum(1 for (q0, t0, ln) in blocks for z in range(ln) if q[q0 + z] != t[t0 + z])
        match = sum(b[2] for b in blocks) - mism
        qni = sum(1 for a, b in zip(blocks, blocks[1:]) if b[0] > a[0] + a[2]); qbi = sum(b[0] - a[0] - a[2] for a, b in zip(blocks, blocks[1:]))
        tni = sum(1 for a, b in zip(blocks, blocks[1:]) if b[1] > a[1] + a[2]); tbi = sum(b[1] - a[1] - a[2] for a, b in zip(blocks, blocks[1:]))
        sq, eq = blocks[0][0], blocks[-1][0] + blocks[-1][2]
        return {"matches": match, "mismatches": mism, "rep_matches": 0, "n_count": 0, "q_num_insert": qni, "q_base_insert": qbi, "t_num_insert": tni,
                "t_base_insert": tbi, "strand": strand, "q_size": len(cs), "q_start": sq if strand == '+' else len(cs) - eq,
                "q_end": eq if strand == '+' else len(cs) - sq, "t_index": tidx, "t_size": len(t), "t_start": blocks[0][1],
                "t_end": blocks[-1][1] + blocks[-1][2], "block_sizes": [b[2] for b in blocks], "q_starts": [b[0] for b in blocks],
                "t_starts": [b[1] for b in blocks], "score": match - 2 * mism}

    def run(tag, r, cd, co, targets, tinfo, truth, genes, disc, qr=None):
        qr = qr or (r.chrom, r.start, r.end, r.name, [(r.chrom, r.start, r.end, r.name, 'exon')])
        out = {"tag": tag, "contig": cd, "read_ids": sorted(x.id for x in co.reads), "targets": targets, "tinfo": [list(x) for x in tinfo],
               "query_region": [qr[0], qr[1], qr[2], qr[3], [list(x) for x in qr[4]]], "genes": genes,
               "disc_reads": {"disc": {k: [list(x) for x in v] for k, v in disc["disc"].items()}, "inv": [list(x) for x in disc["inv"]],
                              "td": [list(x) for x in disc["td"]], "other": [list(x) for x in disc["other"]]}}
        for label, recs in (("contract", bo.realign(cd["seq"], targets)), ("blat_style", truth)):
            rows = [my.psl_fields(x, 'contig1', r.name, 0) for x in recs if x['t_index'] == 0]
            offset, tname = r.start - 200, r.chrom
            res, am = rh.ref_call(rows, co, 'contig1', qr, o, genes, disc, None, None, offset, tname)
            hit = bool(am.bm.target_hit()) if am.bm.has_blat_results else None
            if not hit:                                         # Q14: the reference goes to the whole genome: every record, genome coordinates
                rows = [my.psl_fields(x, 'contig1', 'chr' + tinfo[x['t_index']][0], tinfo[x['t_index']][1]) for x in recs]
                offset, tname = None, None
                res, am = rh.ref_call(rows, co, 'contig1', qr, o, genes, disc, None, None, None, None)
            br0 = am.bm.blat_results[0][3]
            out[label] = {"records": recs, "psl_rows": rows, "offset": offset, "tname": tname, "expected": res, "target_hit": hit,
                          "n_results": len(am.bm.blat_results), "top_mean_cov": getattr(br0, "mean_cov", None),
                          "max_hit_freq": max(am.bm.hit_freq)}
        a, b_ = out["contract"]["expected"], out["blat_style"]["expected"]
        out["same_row"] = a == b_
        print("  %-34s same=%s recs %d/%d max_hit_freq %d/%d\n      A %s\n      B %s" % (
            tag, out["same_row"], len(out["contract"]["records"]), len(truth), out["contract"]["max_hit_freq"], out["blat_style"]["max_hit_freq"],
            a and a[:8], b_ and b_[:8]))
        cases.append(out)

    # ---- deletions whose flanks also sit elsewhere in the window -------------------------------------------------------------
    L, W = 150, 1500
    # (round 5, contract step 8: BLAT's -minIdentity default of 90 % -- a diverged copy of the left flank at ~80 % is not printed by
    #  BLAT and is dropped by the contract; at ~93 % both print it, with its mismatches)
    for tag, fd, div in (("del_unique", 0, 0), ("del_left_flank_dup", 1, 0), ("del_both_flanks_dup", 3, 0), ("del_right_flank_dup_rc", 6, 0), ("del_right_flank_dup", 2, 0),
                         ("del_left_flank_diverged_copy_80pct", 1, 4), ("del_left_flank_diverged_copy_93pct", 1, 12)):
        r = synth.make_region(3, sv_type="del", depth=60, W=W, flank_dups=fd, flank_div=div)
        reads = r.read_strs()
        mers = rh.ref_kmer_select(reads, [r.window_str], 31)
        cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
        assert len(cdicts) == 1
        cs, Wd = cdicts[0]["seq"], r.window_str
        c, h = W // 2, 100
        jq = cs.find(Wd[c + h:c + h + 40])                     # contig bases left of the junction
        assert cs == Wd[c - h - jq:c - h] + Wd[c + h:c + h + len(cs) - jq]
        truth = [block_rec(cs, [Wd], [(0, c - h - jq, jq), (jq, c + h, len(cs) - jq)], '+', 0)]
        if fd & 1 and div != 4:                                  # (the 80 % copy: below -minIdentity=90, BLAT prints nothing for it)
            truth.append(block_rec(cs, [Wd], [(0, 20 + L - jq, jq)], '+', 0))
        if fd & 2:
            n2 = len(cs) - jq
            truth.append(block_rec(cs, [Wd], [(jq, W - 20 - L, n2)], '+', 0) if not fd & 4 else block_rec(cs, [Wd], [(0, W - 20 - n2, n2)], '-', 0))
        run(tag, r, cdicts[0], cobjs[0], [Wd], [(r.chrom, r.start - 200)], truth, {r.name: ['chr' + r.chrom, r.start, r.end]}, r.disc_reads)

    # ---- a deletion with a TEMPLATED insertion at the junction: 25 bases copied from elsewhere in the window.  BLAT indexes the
    # window by the 11-mers at every 10th base and needs two of them to match on a diagonal (sv_processor.py:843 -stepSize=10
    # -minMatch=2): copied from a position 5 past a tile start the stretch holds ONE tile and BLAT never sees it (one record:
    # flank, 25 inserted bases, 200 deleted, flank); copied from a tile start it holds two, and BLAT prints it as a second line.
    for tag, src in (("del_templated_insert_unseedable", 105), ("del_templated_insert_seedable", 100)):
        r = synth.make_region(3, sv_type="del", depth=60, W=W)
        reads = r.read_strs()
        mers = rh.ref_kmer_select(reads, [r.window_str], 31)
        cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
        cs0, Wd = cdicts[0]["seq"], r.window_str
        c, h = W // 2, 100
        jq = cs0.find(Wd[c + h:c + h + 40])
        # a copy whose ends do not happen to continue either flank (a chance match there would move the block borders by a base in
        # any aligner, BLAT included, and the hand-written records below do not model that)
        while Wd[src] == Wd[c - h] or Wd[src + 24] == Wd[c + h - 1] or Wd[src - 1] == cs0[jq - 1] or Wd[src + 25] == cs0[jq]:
            src += 10
        ins = Wd[src:src + 25]
        co = copy.deepcopy(cobjs[0])
        cs = cs0[:jq] + ins + cs0[jq:]
        io, ot, kl = list(co.aseq.counts.indel_only), list(co.aseq.counts.others), list(co.kmer_locs)
        io[jq:jq] = [io[jq]] * 25; ot[jq:jq] = [ot[jq]] * 25; kl[jq:jq] = [kl[jq]] * 25
        co.aseq.seq = cs
        co.aseq.counts.indel_only, co.aseq.counts.others, co.kmer_locs = io, ot, kl
        cd = {"seq": cs, "indel_only": io, "others": ot, "kmer_locs": kl, "kmers": cdicts[0]["kmers"], "reads": cdicts[0]["reads"]}
        truth = [block_rec(cs, [Wd], [(0, c - h - jq, jq), (jq + 25, c + h, len(cs0) - jq)], '+', 0)]
        if src % 10 == 0:
            truth.append(block_rec(cs, [Wd], [(jq, src, 25)], '+', 0))
        run(tag, r, cd, co, [Wd], [(r.chrom, r.start - 200)], truth, {r.name: ['chr' + r.chrom, r.start, r.end]}, r.disc_reads)

    # ---- translocation whose partner half is a repeat (check_uniqueness, sv_caller.py:430-432; the mean_cov of the row) -----
    for tag, copies, with_disc in (("trl_unique_nodisc", 0, False), ("trl_partner_repeat_nodisc", 5, False), ("trl_partner_repeat_disc", 5, True),
                                   ("trl_partner_repeat4_nodisc", 3, False)):
        r = synth.make_region(3, sv_type="trl", depth=60, W=W, trl_repeat_copies=copies)
        reads = r.read_strs()
        mers = rh.ref_kmer_select(reads, [r.window_str], 31)
        cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
        assert len(cdicts) == 1
        co = copy.deepcopy(cobjs[0])
        for i, rd in enumerate(sorted(co.reads, key=lambda x: x.id)):      # reads of both strands: no read_strand_bias check (sv_caller.py:436-446)
            if i % 2:
                rd.id = rd.id.replace("/1_0", "/2_1")
        cs = cdicts[0]["seq"]
        targets = [r.window_str] + [synth.codes_to_str(p_[4]) for p_ in r.partners]
        tinfo = [(r.chrom, r.start - 200)] + [(p_[0], p_[1]) for p_ in r.partners]
        c = W // 2
        ja = cs.find(targets[1][c:c + 40])
        assert cs == targets[0][c - ja:c] + targets[1][c:c + len(cs) - ja]
        truth = [block_rec(cs, targets, [(0, c - ja, ja)], '+', 0), block_rec(cs, targets, [(ja, c, len(cs) - ja)], '+', 1)]
        for i in range(copies):
            truth.append(block_rec(cs, targets, [(ja, 40 + i * (W - c + 40), len(cs) - ja)], '+', 2))
        genes = {r.name: ['chr' + r.chrom, r.start, r.end]}
        for p_ in r.partners:
            genes[p_[3]] = ['chr' + p_[0], p_[1], p_[2]]
        disc = r.disc_reads if with_disc else {"disc": {}, "inv": [], "td": [], "other": []}
        run(tag, r, cdicts[0], co, targets, tinfo, truth, genes, disc)

    # ---- the partner half also sits in the target window, a little less well (check_previous_add, sv_caller.py:55-72) -----
    r = synth.make_region(3, sv_type="trl", depth=60, W=W)
    reads = r.read_strs()
    mers = rh.ref_kmer_select(reads, [r.window_str], 31)
    cdicts, cobjs = rh.ref_init_assembly(r.read_ids, reads, mers, 31, 2, r.indel_only)
    cs = cdicts[0]["seq"]
    pw = synth.codes_to_str(r.partners[0][4])
    c = W // 2
    ja = cs.find(pw[c:c + 40])
    cp = list(pw[c:c + len(cs) - ja])
    for at in (300, 500):
        cp[at] = comp[cp[at]]
    spacer = synth.codes_to_str(synth.rand_bases(synth.stream_key(1, 3, 9), 50))
    W2 = r.window_str + spacer + "".join(cp)
    targets = [W2, pw]
    tinfo = [(r.chrom, r.start - 200), (r.partners[0][0], r.partners[0][1])]
    end2 = r.start + len(W2) - 400                              # the target interval covers the longer window
    qr = (r.chrom, r.start, end2, r.name, [(r.chrom, r.start, end2, r.name, 'exon')])
    genes = {r.name: ['chr' + r.chrom, r.start, end2], r.partners[0][3]: ['chr' + r.partners[0][0], r.partners[0][1], r.partners[0][2]]}
    truth = [block_rec(cs, targets, [(0, c - ja, ja)], '+', 0), block_rec(cs, targets, [(ja, c, len(cs) - ja)], '+', 1),
             block_rec(cs, targets, [(ja, W + 50, len(cs) - ja)], '+', 0)]
    run("trl_partner_half_also_in_target", r, cdicts[0], cobjs[0], targets, tinfo, truth, genes, r.disc_reads, qr)
    dump("realign_multihit.json", {"cases": cases, "opts": o})


def g9():
    """G9 (N3): the file loaders in front of the caller's filters, executed by the REAL reference on synthetic files --
    utils.anno.add_genes / add_regions / set_gene (utils.py:727-773), setup_rmask_all (:302-316) and setup_rmask (:320-353,
    first call: filter + write <name>_rep_mask.bed; second call: read that file back)."""
    import shutil
    import tempfile
    ut = ref_loader.load()["utils"]
    rnd = random.Random(9)
    base = tempfile.mkdtemp()
    # refGene-like table: header line, columns 2 = chrom, 4 = txStart, 5 = txEnd, 12 = name2; repeated gene ids (the widest wins)
    genes = [("GENEA", "chr1", 1000, 9000), ("GENEA", "chr1", 500, 9500), ("GENEA", "chr1", 2000, 3000), ("GENEB", "chr1", 20000, 30000),
             ("GENEB", "chr1", 21000, 35000), ("GENEC", "chr2", 100, 900), ("GENED", "chrX", 5000, 6000), ("GENEA2", "chr1", 8000, 12000)]
    lines = ["#bin\tname\tchrom\tstrand\ttxStart\ttxEnd\tc6\tc7\tc8\tc9\tc10\tc11\tname2"]
    for n, (g, c, s_, e) in enumerate(genes):
        lines.append("\t".join(["0", "NM_%d" % n, c, "+" if n % 2 else "-", str(s_), str(e)] + ["x"] * 6 + [g]))
    gene_fn = os.path.join(base, "genes.txt")
    open(gene_fn, "w").write("\n".join(lines) + "\n")
    regions = [("chr3", 100, 2000, "REGION1"), ("chr1", 40000, 41000, "GENEB"), ("chr4", 7, 9, "REGION2")]
    reg_fn = os.path.join(base, "other.bed")
    open(reg_fn, "w").write("".join("%s %d %d %s\n" % r_ for r_ in regions))
    an = ut.anno()
    an.add_genes(gene_fn)
    after_genes = {k: list(v) for k, v in an.genes.items()}
    an.add_regions(reg_fn)
    queries = [("1", [2500]), ("chr1", [8500]), ("1", [8500, 25000]), ("1", [15000]), ("2", [100]), ("2", [901]), ("X", [5500, 5600]), ("3", [150]), ("7", [5])]
    set_gene = [[c, pos, an.set_gene(c, pos)] for c, pos in queries]
    # repeat mask: chrom (with chr), start, end, name, extra columns
    rm = []
    for n in range(60):
        c = rnd.choice(["chr1", "chr1", "chr2", "chrX"])
        s_ = rnd.randint(0, 40000)
        rm.append((c, s_, s_ + rnd.randint(5, 400), rnd.choice(["AluY", "(CA)n", "L1PA3", "GA_rich", "MIR"]), rnd.randint(0, 999)))
    rm_fn = os.path.join(base, "rmask.bed")
    open(rm_fn, "w").write("".join("%s\t%d\t%d\t%s\t%d\n" % r_ for r_ in rm))
    all_mask = ut.setup_rmask_all(rm_fn)
    per_target = []
    for chrom, s_, e, name in (("1", 1000, 20000, "GENEA"), ("chr1", 1000, 20000, "GENEA_CHR"), ("2", 0, 50000, "GENEC"), ("9", 0, 100, "NONE")):
        ref_path = os.path.join(base, "ref_" + name)
        os.makedirs(ref_path)
        coords = (chrom, s_, e, name, [(chrom, s_, e, name, "exon")])
        first = ut.setup_rmask(coords, ref_path, rm_fn)
        bed = open(os.path.join(ref_path, name + "_rep_mask.bed")).read()
        second = ut.setup_rmask(coords, ref_path, rm_fn)             # marker file present: reads the bed back
        per_target.append({"coords": [chrom, s_, e, name], "first": [list(x) for x in first], "bed": bed, "second": [list(x) for x in second]})
    dump("loaders.json", {"gene_table": open(gene_fn).read(), "regions_bed": open(reg_fn).read(), "genes_after_add_genes": after_genes,
                          "genes_after_add_regions": {k: list(v) for k, v in an.genes.items()}, "gene_order": list(an.genes.keys()), "set_gene": set_gene,
                          "repeat_mask_bed": open(rm_fn).read(), "rmask_all": {k: [list(x) for x in v] for k, v in all_mask.items()}, "rmask_targets": per_target})
    shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    assert ref_loader.available(), "reference not present"
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g8m", "g9"]
    for w in which:
        print(w)
        globals()[w]()
