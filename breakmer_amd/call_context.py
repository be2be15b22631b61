"""Serialise what the native call tail (csrc/bk_call.h) needs -- options, gene table, repeat masks and,
per region, query_region / discordant pairs / partner placement / read-id classes -- into the line-based
text format `bk_set_call_context` / `bk_call_text` parse."""
from __future__ import annotations


def _simple(name):
    return 1 if (")n" in name or "_rich" in name) else 0            # sv_caller.py:865


def _nochr(c):
    return str(c).replace("chr", "")


def opts_line(opts):
    vf = opts['var_filter']
    mask = (1 if 'indel' in vf else 0) | (2 if 'rearrangement' in vf else 0) | (4 if 'trl' in vf else 0)
    return "opts %d %d %d %d %d %d %d %d" % (int(opts['indel_size']), int(opts['trl_sr_thresh']), int(opts['indel_sr_thresh']),
                                             int(opts['rearr_sr_thresh']), int(opts['rearr_minseg_len']), int(opts['trl_minseg_len']),
                                             1 if opts['keep_intron_vars'] else 0, mask)


def tables_lines(genes, all_repeat_mask):
    out = ["gene %s %s %d %d" % (g, v[0], v[1], v[2]) for g, v in genes.items()]
    if all_repeat_mask:
        out.append("arep_on")
        for c, lst in all_repeat_mask.items():
            for r in lst:
                out.append("arep %s %d %d %d" % (_nochr(r[0]), r[1], r[2], _simple(r[3])))
    return out


def region_lines(idx, query_region, target_repeat_mask, disc_reads, partners=(), read_ids=None):
    chrom, start, end, name, intervals = query_region
    out = ["region %d %s %d %d %s" % (idx, chrom, start, end, name)]
    for iv in intervals:
        out.append("iv %d %d %d" % (iv[1], iv[2], 1 if iv[4] == 'exon' else 0))
    if target_repeat_mask:
        out.append("trep_on")
        for r in target_repeat_mask:
            out.append("trep %s %d %d %d" % (_nochr(r[0]), r[1], r[2], _simple(r[3])))
    for key, tag in (("inv", "inv"), ("td", "td"), ("other", "other")):
        for p in disc_reads.get(key, []):
            out.append("%s %d %d %d %d" % (tag, p[0], p[1], p[2], p[3]))
    for c, lst in disc_reads.get("disc", {}).items():
        for p in lst:
            out.append("disc %s %d %d" % (c, p[0], p[1]))
    for p in partners:
        out.append("partner %s %d" % (p[0], p[1]))
    if read_ids is not None and getattr(read_ids, "uniform_tag", None) is not None:
        out.append("rtags 0")                                        # every id carries the same tag: one class (reads beyond the string read as class 0, bk_api.hip)
    elif read_ids is not None:                                       # classes of read.id.split("/")[1] (sv_caller.py:441)
        cls, tags = {}, []
        for rid in read_ids:
            s = rid.split("/")[1] if "/" in rid else ""
            tags.append("0123456789abcdefghijklmnopqrstuvwxyz"[min(cls.setdefault(s, len(cls)), 35)])
        out.append("rtags " + ("".join(tags) if tags else "0"))
    return out


def contig_lines(contig_id, seq, indel_only, others, kmer_locs, nkmers, same_tag, rows, offset=None, tname=None):
    out = ["contig %s %s %d %d" % (contig_id, seq, nkmers, 1 if same_tag else 0),
           "io " + ",".join(str(x) for x in indel_only), "ot " + ",".join(str(x) for x in others),
           "klocs " + ",".join(str(x) for x in kmer_locs)]
    if offset is not None:
        out.append("offset %d" % offset)
    if tname is not None:
        out.append("tname %s" % tname)
    for r in rows:
        out.append("row " + " ".join(str(x) for x in r))
    return out
