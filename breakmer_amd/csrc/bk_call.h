// bk_call.h -- native (host C++) SV-call tail: PSL-equivalent records + contig vectors -> the 13-field row.
// Same semantics as breakmer_amd/sv_caller.py, which is pinned against the real reference (tests/golden/
// caller.json, G5); this port exists so that the whole-path throughput is not bound by the Python interpreter.
// Reference: sv_caller.py:13-1142 (blat_res, blat_manager, align_manager, sv_event), utils.py:20-94.
// Host code only (no device code here).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

namespace bkcall {

struct Opts { int indel_size = 15, trl_sr = 2, indel_sr = 5, rearr_sr = 3, rearr_minseg = 30, trl_minseg = 25, keep_intron = 0, var_filter = 7; };   // var_filter: 1 indel | 2 rearrangement | 4 trl
struct Interval { int start, end; bool exon; };
struct Gene { std::string name, chrom; int start, end; };
struct Repeat { std::string chrom; int start, end; bool simple; };          // simple: ")n" or "_rich" in the name (sv_caller.py:865)
struct Pair5 { int p1, p2, s1, s2; };
struct Disc { std::string chrom; int p1, p2; };
struct Region {
    std::string chrom, name; int start = 0, end = 0;
    std::vector<Interval> intervals;
    bool has_target_repeats = false; std::vector<Repeat> target_repeats;
    std::vector<Pair5> inv, td, other; std::vector<Disc> disc;
};
struct Tables { std::vector<Gene> genes; bool has_repeats = false; std::vector<Repeat> repeats; };

struct Psl {   // one record, target coordinates already absolute (offset applied), tname without "chr"
    int matches, mis, rep, qni, qbi, tni, tbi; char strand; int qsize, qstart, qend; std::string tname; int tsize, tstart, tend;
    std::vector<int> bs, qs, ts;
};
struct Contig { std::string seq, id; const int *io; const int *ot; int clen; const int *klocs; int nkmers; bool same_read_tag; };

// Python 2.7 round(): correctly rounded on the exact binary value, exact ties away from zero (values >= 0 here)
inline double round2(double x, int n)
{
    static const double P[] = {1, 10, 100, 1000, 10000};
    const double p = P[n], y = x * p, e = std::fma(x, p, -y), f = std::floor(y), d = y - f;
    double r = f;
    if (d > 0.5 || (d == 0.5 && e >= 0)) r = f + 1;
    return r / p;
}
inline std::string pyfloat(double v)          // str(float) of Python 3: shortest round-trip digits, repr layout
{
    if (v == 0) return "0.0";
    char buf[64]; int p = 1;
    for (; p <= 17; p++) { snprintf(buf, sizeof buf, "%.*e", p - 1, v); if (strtod(buf, nullptr) == v) break; }
    std::string m(buf); const bool neg = m[0] == '-'; if (neg) m.erase(0, 1);
    const size_t epos = m.find('e'); const int e10 = atoi(m.c_str() + epos + 1);
    std::string digits; for (size_t i = 0; i < epos; i++) if (m[i] != '.') digits += m[i];
    std::string out;
    if (e10 >= -4 && e10 < 16) {
        const int nd = (int)digits.size();
        if (e10 >= nd - 1) out = digits + std::string((size_t)(e10 - (nd - 1)), '0') + ".0";
        else if (e10 >= 0) out = digits.substr(0, (size_t)e10 + 1) + "." + digits.substr((size_t)e10 + 1);
        else out = "0." + std::string((size_t)(-e10 - 1), '0') + digits;
    } else {
        out = digits.substr(0, 1) + (digits.size() > 1 ? "." + digits.substr(1) : "") + "e" + (e10 < 0 ? "-" : "+");
        const int ae = std::abs(e10); if (ae < 10) out += "0"; out += std::to_string(ae);
    }
    return neg ? "-" + out : out;
}
inline std::string istr(long long v) { return std::to_string(v); }

struct RepMan { bool bp_in_rep[2] = {false, false}; double total = 0, simple = 0; bool in_repeat = false; double overlap = 0; bool edges[2] = {false, false}; };

struct BlatRes {
    Psl p; std::string genes; bool in_target = false, valid = true, in_repeat = false; double repeat_overlap = 0; bool filter_edges[2] = {false, false};
    bool has_repman = false; RepMan rm;
    std::vector<std::vector<int>> breakpts; std::vector<int> query_brkpts; std::vector<std::string> indel_sizes;
    double mean_cov = 0, perc_ident = 0; int seg_overlap[2] = {0, 0}; std::string cigar; int flank[2] = {0, 0};
    int nmatch_total() const { return p.matches + p.rep; }
    int ngap_total() const { return p.tbi + p.qbi; }
    int num_gaps() const { return p.tni + p.qni; }
    bool spans_query() const { return p.qsize == p.qend - p.qstart; }
};

inline int flank_m_sum(const std::string &piece)
{
    int total = 0;
    for (size_t i = 0; i < piece.size(); i++) {
        if (piece[i] != 'M') continue;
        int j = (int)i - 1; std::string digits;
        while (j > -1 && piece[j] >= '0' && piece[j] <= '9') { digits = piece[j] + digits; j--; }
        total += atoi(digits.c_str());
    }
    return total;
}

inline void set_indel_locs(BlatRes &b)                               // sv_caller.py:1044-1085
{
    const Psl &p = b.p; const int n = (int)p.bs.size();
    std::vector<std::string> sizes; int big = 0; char bigt = 0;
    for (int i = 0; i < n - 1; i++) {
        if (i == 0 && p.qs[0] > 0) b.cigar = istr(p.qs[0]) + "S";
        const int qend1 = p.qs[i] + p.bs[i], qstart2 = p.qs[i + 1], tend1 = p.ts[i] + p.bs[i], tstart2 = p.ts[i + 1];
        const int ins = qstart2 - qend1, del = tstart2 - tend1;
        b.cigar += istr(p.bs[i]) + "M";
        auto addq = [&](int v) { if (std::find(b.query_brkpts.begin(), b.query_brkpts.end(), v) == b.query_brkpts.end()) b.query_brkpts.push_back(v); };
        if (ins > 0) { b.breakpts.push_back({tend1}); sizes.push_back("I" + istr(ins)); addq(qend1); addq(qstart2); b.cigar += istr(ins) + "I"; if (ins > big) { big = ins; bigt = 'I'; } }
        if (del > 0) { b.breakpts.push_back({tend1, tstart2}); sizes.push_back("D" + istr(del)); addq(qend1); b.cigar += istr(del) + "D"; if (del > big) { big = del; bigt = 'D'; } }
    }
    b.cigar += istr(p.bs[n - 1]) + "M";
    const int tail = p.qsize - p.qend;
    if (tail > 0) b.cigar += istr(tail) + "S";
    if (big > 0) {
        const std::string key = istr(big) + bigt; std::vector<std::string> parts; size_t pos = 0, f;
        while ((f = b.cigar.find(key, pos)) != std::string::npos) { parts.push_back(b.cigar.substr(pos, f - pos)); pos = f + key.size(); }
        parts.push_back(b.cigar.substr(pos));
        b.flank[0] += flank_m_sum(parts.front()); b.flank[1] += flank_m_sum(parts.back());
    }
    if (!sizes.empty()) { std::string s; for (size_t i = 0; i < sizes.size(); i++) s += (i ? "," : "") + sizes[i]; b.indel_sizes.push_back(s); }
    if (p.strand == '-') for (auto &q : b.query_brkpts) q = p.qsize - q;
}

inline double calc_milli_bad(const BlatRes &b)                        // :954-968
{
    const Psl &p = b.p; const int qali = p.qend - p.qstart, tali = p.tend - p.tstart;
    if (std::min(qali, tali) <= 0) return 0.0;
    const int dif = std::max(qali - tali, 0), total = p.matches + p.rep + p.mis;
    double bad = 0.0;
    if (total != 0) bad = (1000 * (p.mis + p.qni + round2(3 * std::log(1 + (double)dif), 0))) / total;
    return bad * 0.1;
}

inline void repman_setup(RepMan &rm, int start, int end, const std::vector<Repeat> &locs)   // :850-875
{
    const double seg = (double)(end - start); bool hit = false; double total = 0, simple = 0; bool edges[2] = {false, false};
    for (const Repeat &r : locs) {
        if ((r.start >= start && r.start <= end) || (r.end >= start && r.end <= end) || (r.start <= start && r.end >= end)) {
            hit = true; const double ov = (double)(std::min(r.end, end) - std::max(r.start, start)); total += ov;
            if (r.simple) { simple += ov; if (r.start <= start && r.end >= start) edges[0] = true; else if (r.start <= end && r.end >= end) edges[1] = true; }
        }
    }
    rm.total = round2(std::min(total, seg) / seg * 100, 2); rm.simple = round2(std::min(simple, seg) / seg * 100, 2);
    rm.bp_in_rep[0] = edges[0]; rm.bp_in_rep[1] = edges[1]; rm.in_repeat = hit; rm.overlap = rm.total; rm.edges[0] = edges[0]; rm.edges[1] = edges[1];
}

// annotated == false: the check_target_blat pass (no params in the meta dict: no gene/repeat annotation)
inline BlatRes make_blat_res(const Psl &p, bool annotated, const Region &rg, const Tables &tb)
{
    BlatRes b; b.p = p;
    if (annotated) {
        const int lo = rg.start - 200, hi = rg.end + 200;          // set_gene_anno :1119-1142
        if (rg.chrom == p.tname && ((p.tstart >= lo && p.tstart <= hi) || (p.tend >= lo && p.tend <= hi))) { b.in_target = true; b.genes = rg.name; }
        else {
            std::string chrom = p.tname; if (chrom.find("chr") == std::string::npos) chrom = "chr" + chrom;
            bool found = false;
            for (const Gene &g : tb.genes) if (chrom == g.chrom && p.tstart >= g.start && p.tstart <= g.end) { b.genes = g.name; found = true; break; }
            if (!found) { b.genes = "intergenic"; b.valid = false; }
        }
        b.has_repman = true;                                         // set_repeat :973-986
        if (p.rep > 0) b.in_repeat = true;
        if (!b.in_repeat && rg.has_target_repeats && !rg.target_repeats.empty() && tb.has_repeats && !tb.repeats.empty()) {
            std::vector<Repeat> sel; const std::vector<Repeat> *rmask = &rg.target_repeats;
            if (!b.in_target) { for (const Repeat &r : tb.repeats) if (r.chrom == p.tname) sel.push_back(r); rmask = &sel; }
            if (!rmask->empty()) {
                repman_setup(b.rm, p.tstart, p.tend, *rmask);
                b.in_repeat = b.rm.in_repeat; b.repeat_overlap = b.rm.overlap; b.filter_edges[0] = b.rm.edges[0]; b.filter_edges[1] = b.rm.edges[1];
            }
        }
    }
    set_indel_locs(b);
    b.perc_ident = 100.0 - calc_milli_bad(b);
    return b;
}

struct Counts {
    const int *io, *ot; int n;
    // python slice / index semantics of assembly_counts.get_counts (sv_assembly.py:167-176); ok=false on an IndexError
    int at(int p, bool both, bool &ok) const { int q = p < 0 ? p + n : p; if (q < 0 || q >= n) { ok = false; return 0; } return both ? io[q] + ot[q] : ot[q]; }
    std::vector<int> range(int p1, int p2, bool both) const {
        auto clip = [&](int v) { if (v < 0) v += n; return std::max(0, std::min(v, n)); };
        std::vector<int> r; for (int i = clip(p1); i < clip(p2); i++) r.push_back(both ? io[i] + ot[i] : ot[i]); return r;
    }
};

inline void contig_complexity(const std::string &seq, double &avg, std::vector<double> &vec)   // utils.py:20-39
{
    const int L = (int)seq.size(); vec.clear(); double sum = 0;
    for (int i = 0; i < L; i++) {
        const int s = std::max(0, i - 6), e = std::min(L, i + 6);
        std::set<std::string> kinds; for (int j = s; j + 3 <= e; j++) kinds.insert(seq.substr(j, 3));
        const double v = round2((double)kinds.size() / (double)(e - s), 2); vec.push_back(v); sum += v;
    }
    avg = sum / (double)L;
}

inline void filter_by_feature(const std::vector<int> &bps, const Region &rg, bool keep_intron, bool &in_f, bool &span_f)   // utils.py:58-94
{
    in_f = span_f = false;
    if (keep_intron) return;
    bool in_any = false, in_exon = false, sp_any = false, sp_exon = false;
    if (!bps.empty()) {
        const int mx = *std::max_element(bps.begin(), bps.end()), mn = *std::min_element(bps.begin(), bps.end());
        for (int bp : bps) for (const Interval &iv : rg.intervals) {
            if (bp >= iv.start - 20 && bp <= iv.end + 20) { in_any = true; if (iv.exon) in_exon = true; }
            if (iv.end <= mx && iv.start >= mn) { sp_any = true; if (iv.exon) sp_exon = true; }
        }
    }
    in_f = !(in_any && in_exon); span_f = !(sp_any && sp_exon);
}

struct Event {                                                       // sv_event
    std::vector<std::pair<int, int>> blat_res;                      // (qstart, index into brs)
    std::vector<std::pair<int, int>> br_sorted;                     // (index, nmatch)
    int qlen = 0; bool in_target = false, valid = true; std::vector<int> query_cov;
};

struct Caller {
    const Opts &o; const Region &rg; const Tables &tb; const Contig &ct;
    std::vector<BlatRes> brs; std::vector<int> order;               // order: blat_results sorted
    std::vector<int> hit_freq; int qsize = 0;
    std::vector<std::pair<int, int>> clipped;                       // (index into brs, i)
    Event se; bool has_se = false; bool error = false;
    Caller(const Opts &o_, const Region &r_, const Tables &t_, const Contig &c_) : o(o_), rg(r_), tb(t_), ct(c_) {}

    void load(const std::vector<Psl> &rows, bool annotated)
    {
        std::vector<double> score;
        for (const Psl &p : rows) {
            brs.push_back(make_blat_res(p, annotated, rg, tb));
            const BlatRes &b = brs.back();
            score.push_back(b.nmatch_total() + (double)b.nmatch_total() / (double)p.qsize);
            if (!qsize) { qsize = p.qsize; hit_freq.assign(qsize, 0); }
            for (int i = p.qstart; i < p.qend && i < qsize; i++) hit_freq[i]++;
        }
        order.resize(brs.size()); for (size_t i = 0; i < brs.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            if (score[a] != score[b]) return score[a] > score[b];
            if (brs[a].perc_ident != brs[b].perc_ident) return brs[a].perc_ident > brs[b].perc_ident;
            return brs[a].ngap_total() < brs[b].ngap_total(); });
    }
    double query_coverage() const { int n = 0; for (int v : hit_freq) n += v > 0; return round2((double)n / (double)qsize * 100, 2); }
    bool target_hit() const { return brs[order[0]].spans_query() || (brs.size() == 1 && query_coverage() >= 90.0); }
    double mean_cov(int s, int e) const { double t = 0; int n = 0; for (int i = std::max(s, 0); i < e && i < qsize; i++) { t += hit_freq[i]; n++; } return n ? t / n : 0.0; }

    Event new_event(int bi) { Event e; e.query_cov.assign(ct.seq.size(), 0); add(e, bi); return e; }
    void add(Event &e, int bi)
    {
        const BlatRes &b = brs[bi];
        e.blat_res.push_back({b.p.qstart, bi});
        for (int i = b.p.qstart; i < b.p.qend && i < (int)e.query_cov.size(); i++) e.query_cov[i]++;
        e.qlen += b.p.qend - b.p.qstart; e.in_target = e.in_target || b.in_target; e.valid = e.valid && b.valid;
        e.br_sorted.push_back({bi, b.nmatch_total()});
    }
    void check_previous_add(Event &e, int bi)
    {
        const BlatRes &b = brs[bi]; auto &last = e.br_sorted.back(); const BlatRes &prev = brs[last.first];
        if (b.p.qstart == prev.p.qstart && b.p.qend == prev.p.qend) {
            const int n = b.nmatch_total();
            if (std::abs(last.second - n) < 10 && !prev.in_target && b.in_target) { last = {bi, n}; e.blat_res.back() = {b.p.qstart, bi}; e.in_target = true; }
        }
    }

    bool check_blat_indel(int bi)                                    // :621-651
    {
        BlatRes &b = brs[bi];
        if (!(b.spans_query() || (brs.size() == 1 && b.in_target))) return false;
        const bool keep = b.valid && b.mean_cov < 2 && b.in_target && b.ngap_total() >= o.indel_size && !b.rm.bp_in_rep[0] && !b.rm.bp_in_rep[1];
        if (keep) {
            Counts c{ct.io, ct.ot, ct.clen}; bool ok = true; int mn = 1 << 30;
            for (int x : b.query_brkpts) mn = std::min(mn, c.at(x, true, ok));
            if (b.query_brkpts.empty() || !ok) { error = true; return true; }
            const bool low_cov = mn < o.indel_sr;
            bool flanks = true; for (int f : b.flank) if (round2((double)f / (double)b.p.qsize * 100, 2) < 10.0) flanks = false;
            std::vector<int> locs; for (auto &bp : b.breakpts) for (int v : bp) locs.push_back(v);
            bool in_f, sp_f; filter_by_feature(locs, rg, o.keep_intron, in_f, sp_f);
            if (!in_f && !low_cov && flanks) { se = new_event(bi); has_se = true; }
        }
        return true;
    }
    bool check_indels()
    {
        for (size_t i = 0; i < order.size(); i++) {
            BlatRes &b = brs[order[i]]; b.mean_cov = mean_cov(b.p.qstart, b.p.qend);
            if (i == 0 && check_blat_indel(order[i])) return true;
            clipped.push_back({order[i], (int)i});
        }
        return false;
    }
    bool check_add_br(int qs, int qe, int gs, int ge, BlatRes &b)
    {
        const double over = round2((double)(std::min(qe, ge) - std::max(qs, gs)) / (double)(qe - qs) * 100, 0);
        const int ovr = qe > ge ? std::abs(qe - ge) : 0, ovl = qs < gs ? std::abs(qs - gs) : 0;
        b.seg_overlap[0] = ovr; b.seg_overlap[1] = ovl;              // set_segment_overlap(ov_left, ov_right) stores [left=arg2, right=arg1] (:970)
        return over >= 50 && (std::max(ovr, ovl) < 15 || (b.in_target && se.in_target));
    }
    bool check_svs()
    {
        std::vector<std::pair<int, int>> gaps{{0, qsize}};
        if (clipped.size() > 1) {
            // merged_clip of the reference always holds the one sv_event object created at the first clipped hit
            // (later hits mutate it in place), so no copy is kept here
            for (size_t it = 0; it < clipped.size(); it++) {
                const int bi = clipped[it].first; BlatRes &b = brs[bi]; const int qs = b.p.qstart, qe = b.p.qend;
                std::vector<std::pair<int, int>> out; bool hit = false;
                for (auto &g : gaps) {
                    const int gs = g.first, ge = g.second;
                    if ((qs >= gs && qs <= ge) || (qe <= ge && qe >= gs)) {
                        std::vector<std::pair<int, int>> rest;
                        if (qs > gs && (qs - 1 - gs) > 10) rest.push_back({gs, qs - 1});
                        if (qe < ge && (ge - qe + 1) > 10) rest.push_back({qe + 1, ge});
                        if (it == 0) { se = new_event(bi); has_se = true; out.insert(out.end(), rest.begin(), rest.end()); hit = true; }
                        else if (check_add_br(qs, qe, gs, ge, b)) { out.insert(out.end(), rest.begin(), rest.end()); add(se, bi); hit = true; }
                        else out.push_back(g);
                    } else out.push_back(g);
                }
                if (!hit && has_se) check_previous_add(se, bi);
                gaps = out;
            }
        }
        if (has_se && se.blat_res.size() > 1 && se.in_target) {
            int miss = 0; for (int v : se.query_cov) miss += v == 0;
            return miss < o.trl_minseg;
        }
        return false;
    }

    // ---- rows
    std::string brkpt_coverages(const std::string &tbp_in) const
    {
        std::string tbp = tbp_in; if (tbp.find('(') != std::string::npos) tbp = tbp.substr(0, tbp.find_first_of(" \t"));
        int n = 0; size_t pos = 0;
        while (pos <= tbp.size()) { size_t c = tbp.find(',', pos); std::string bp = tbp.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
            size_t col = bp.find(':'); std::string locs = col == std::string::npos ? bp : bp.substr(col + 1); n += locs.find('-') != std::string::npos ? 2 : 1;
            if (c == std::string::npos) break; pos = c + 1; }
        std::string s; for (int i = 0; i < n; i++) s += i ? ",0" : "0"; return s;
    }
    std::vector<std::string> row(const std::string &genes, const std::string &tbp, const std::string &cigar, const std::string &mism, const std::string &strands,
                                 const std::string &rep, std::string svtype, const std::string &split, const std::string &disc) const
    {
        if (svtype == "trl") svtype = "rearrangement";
        return {genes, tbp, cigar, mism, strands, rep, svtype, split, istr(ct.nkmers), disc, brkpt_coverages(tbp), rg.name + "_" + ct.id, ct.seq};
    }
    bool indel_result(std::vector<std::string> &out)
    {
        if (!has_se) return false;
        const BlatRes &b = brs[se.blat_res[0].second];
        std::string tbp;
        if (!b.breakpts.empty()) {
            const size_t n = std::min(b.breakpts.size(), b.indel_sizes.size());
            for (size_t i = 0; i < n; i++) { const auto &bp = b.breakpts[i]; tbp = "chr" + b.p.tname + ":" + (bp.size() > 1 ? istr(bp[0]) + "-" + istr(bp[1]) : istr(bp[0])) + " (" + b.indel_sizes[i] + ")"; }
        }
        Counts c{ct.io, ct.ot, ct.clen}; bool ok = true; std::string split;
        for (size_t i = 0; i < b.query_brkpts.size(); i++) split += (i ? "," : "") + istr(c.at(b.query_brkpts[i], true, ok));
        if (!ok) return false;
        out = row(b.genes, tbp, b.cigar, istr(b.p.mis), std::string(1, b.p.strand), "0.0:" + istr(b.nmatch_total()), "indel", split, "0");
        return true;
    }
    bool svs_result(std::vector<std::string> &out)
    {
        if (!has_se) return false;
        std::vector<std::pair<int, int>> ord = se.blat_res;
        std::stable_sort(ord.begin(), ord.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
        struct QB { int a, b, c; bool cnone; };
        std::vector<QB> q1; int q0[2] = {0, 0};
        std::vector<std::string> chrs, brk_str, genes, cigars, strands, mism, repm; std::vector<int> r; std::vector<std::pair<int, int>> tcoords; std::vector<int> f;   // f: -1 None, 0 False, 1 True
        std::string t_in_name, t_other_name; int t_in_pos = 0, t_other_pos = 0; bool has_in = false, has_other = false;
        bool all_simple = true; double max_repeat = 0;
        for (size_t i = 0; i < ord.size(); i++) {
            const BlatRes &b = brs[ord[i].second]; const bool last = i == ord.size() - 1;
            all_simple = all_simple && (b.rm.simple > 75.0);
            max_repeat = std::max(max_repeat, b.repeat_overlap);
            repm.push_back(pyfloat(b.repeat_overlap) + ":" + istr(b.nmatch_total()) + ":" + pyfloat(round2(b.mean_cov, 3)));
            genes.push_back(b.genes); cigars.push_back(b.cigar); strands.push_back(std::string(1, b.p.strand)); mism.push_back(istr(b.p.mis));
            const int ts = b.p.tstart, te = b.p.tend, qs = b.p.qstart, qe = b.p.qend; const bool minus = b.p.strand == '-';
            chrs.push_back(b.p.tname); tcoords.push_back({ts, te});
            std::vector<int> pts; int rep = -1;
            if (i == 0) { q0[0] = std::max(0, qs - 1); q0[1] = qe; q1.push_back({qe, qe - q0[0], 0, true}); pts = {minus ? ts : te}; rep = b.filter_edges[0]; }
            else if (last) { q1.back().c = qe - q1.back().a; q1.back().cnone = false; q1.push_back({qs, qs - q0[0], qe - qs, false}); pts = {minus ? te : ts}; rep = minus ? b.filter_edges[1] : b.filter_edges[0]; }
            else { q1.back().c = qe - q1.back().b; q1.back().cnone = false; q1.push_back({qs, qs - q0[0], qe - qs, false}); q1.push_back({qe, qe - qs, 0, true}); q0[0] = qs; q0[1] = qe;
                   if (minus) { rep = b.filter_edges[1]; pts = {te, ts}; } else pts = {ts, te}; }
            std::string text = "chr" + b.p.tname + ":"; for (size_t z = 0; z < pts.size(); z++) text += (z ? "-" : "") + istr(pts[z]);
            brk_str.push_back(text); for (int v : pts) r.push_back(v); f.push_back(rep);
            if (b.in_target) { t_in_name = b.p.tname; t_in_pos = pts[0]; has_in = true; } else { t_other_name = b.p.tname; t_other_pos = pts[0]; has_other = true; }
        }
        std::vector<std::pair<int, int>> brs_sorted = se.br_sorted;
        std::stable_sort(brs_sorted.begin(), brs_sorted.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.second < b.second; });
        auto join = [](const std::vector<std::string> &v) { std::string s; for (size_t i = 0; i < v.size(); i++) s += (i ? "," : "") + v[i]; return s; };
        // multiple_genes (:483-492)
        std::set<std::string> gset(genes.begin(), genes.end()); bool mult = true;
        if (gset.size() == 1) mult = false;
        else {
            bool dup = false; for (size_t i = 0; i + 1 < genes.size(); i++) for (size_t j = i + 1; j < genes.size(); j++) if (genes[i].find(genes[j]) != std::string::npos || genes[j].find(genes[i]) != std::string::npos) dup = true;
            std::set<std::string> cs(chrs.begin(), chrs.end());
            if (dup && cs.size() == 1 && (*std::max_element(r.begin(), r.end()) - *std::min_element(r.begin(), r.end())) < 10000) mult = false;
        }
        Counts c{ct.io, ct.ot, ct.clen};
        auto brk_counts = [&](bool both, std::vector<int> &cn, std::vector<int> &cd, std::vector<int> &cb, std::vector<int> &km, bool &rep_filt) -> bool {
            double avg; std::vector<double> vec; contig_complexity(ct.seq, avg, vec); rep_filt = false; bool ok = true;
            for (const QB &qb : q1) {
                if (qb.cnone) return false;
                const int left = qb.a - std::min(qb.b, 5), right = qb.a + std::min(qb.c, 5);
                std::vector<int> v1 = c.range(left, right, both), v2 = c.range(qb.a - 1, qb.a + 1, both);
                if (v1.empty() || v2.empty() || qb.a < 0 || qb.a >= (int)ct.seq.size()) return false;
                cn.push_back(*std::min_element(v1.begin(), v1.end())); cd.push_back(*std::min_element(v2.begin(), v2.end())); cb.push_back(c.at(qb.a, both, ok));
                km.push_back(ct.klocs[qb.a]); rep_filt = rep_filt || (vec[qb.a] < avg / 2);
            }
            for (int v : f) if (v == 1) rep_filt = true;
            return ok;
        };
        if (!mult) {
            std::vector<int> cn, cd, cb, km; bool rep_filt;
            if (!brk_counts(true, cn, cd, cb, km, rep_filt)) { error = true; return false; }
            // define_rearr (:327-367)
            std::string kind = "rearrangement"; int support = 0; bool typed = false;
            auto contained = [](std::pair<int, int> a, std::pair<int, int> b) { return (a.first >= b.first && a.second <= b.second) || (b.first >= a.first && b.second <= a.second); };
            if (strands.size() < 3 && !contained(tcoords[0], tcoords[1])) {
                if (strands[0] != strands[1] && r[0] < r[1]) {
                    typed = true; kind = "inversion";
                    for (const Pair5 &p : rg.inv) { if (p.s1 == 1 && p.s2 == 1) { if (p.p1 <= r[0] && p.p2 <= r[1] && p.p2 >= r[0]) support++; } else if (p.p1 <= r[1] && p.p1 >= r[0] && p.p2 >= r[1]) support++; }
                } else if (strands[0] == "+" && strands[1] == "+" && r[0] > r[1]) { typed = true; kind = "tandem_dup"; }     // support is never counted (Q13)
            }
            if (!typed) { int mx = 0; for (int b : r) { int n = 0; for (const Pair5 &p : rg.other) if (std::abs(p.p1 - b) <= 300 || std::abs(p.p2 - b) <= 300) n++; mx = std::max(mx, n); } support = mx; }
            bool in_f, sp_f; filter_by_feature(r, rg, o.keep_intron, in_f, sp_f);
            const bool drop = (*std::min_element(cn.begin(), cn.end()) < o.rearr_sr) || brs_sorted[0].second < o.rearr_minseg || (in_f && sp_f) || support < 1 || kind == "rearrangement" || *std::min_element(km.begin(), km.end()) == 0;
            if (drop || !(o.var_filter & 2)) return false;
            std::vector<std::string> ug(gset.begin(), gset.end());     // list(set(...)): one element, or sorted
            std::vector<std::string> cbs; for (int v : cb) cbs.push_back(istr(v));
            out = row(join(ug), join(brk_str), join(cigars), join(mism), join(strands), join(repm), "rearrangement_" + kind, join(cbs), istr(support));
            return true;
        }
        int mx_ot = ct.ot[0]; for (int i = 1; i < ct.clen; i++) mx_ot = std::max(mx_ot, ct.ot[i]);
        if (mx_ot < o.trl_sr) return false;
        std::vector<int> cn, cd, cb, km; bool rep_filt;
        if (!brk_counts(false, cn, cd, cb, km, rep_filt)) { error = true; return false; }
        int disc = 0;                                                // check_disc_reads (:507-514)
        if (!has_other || !has_in) { error = !has_other ? true : error; if (!has_other) return false; }
        for (const Disc &d : rg.disc) if (d.chrom == t_other_name && std::abs(d.p1 - t_in_pos) <= 1000 && std::abs(d.p2 - t_other_pos) <= 1000) disc++;
        bool drop = all_simple || (*std::max_element(cd.begin(), cd.end()) < o.trl_sr);   // filter_trl (:383-422)
        if (!drop && disc < 2) {
            const BlatRes &sh = brs[brs_sorted[0].first]; const int short_n = brs_sorted[0].second;
            if (short_n < o.trl_minseg || *std::min_element(cn.begin(), cn.end()) < o.trl_sr || *std::min_element(km.begin(), km.end()) == 0 || rep_filt) drop = true;
            else if (disc == 0) {
                int checks = 0;
                { const std::string seg = ct.seq.substr(std::min<size_t>(sh.p.qstart, ct.seq.size()), std::max(0, sh.p.qend - sh.p.qstart)); std::set<std::string> kinds; for (int i = 0; i + 3 <= (int)seg.size(); i++) kinds.insert(seg.substr(i, 3));
                  if (round2((double)kinds.size() / (double)((int)seg.size() - 2) * 100, 4) < 25.0) checks++; }
                { int miss = 0; for (size_t i = 0; i < se.query_cov.size() && !se.query_cov[i]; i++) miss++; for (size_t i = se.query_cov.size(); i-- > 0 && !se.query_cov[i];) miss++;
                  if (round2((double)miss / (double)ct.seq.size() * 100, 4) > 5.0) checks++; }
                if (short_n <= round2((double)ct.seq.size() / 4.0, 0)) checks++;
                if (std::max(sh.seg_overlap[0], sh.seg_overlap[1]) > 5) checks++;
                if (std::max(sh.p.qni, sh.p.tni) > 0) checks++;
                { bool low = false; for (auto &bs : brs_sorted) { const BlatRes &b = brs[bs.first]; if (b.in_target ? b.mean_cov > 10 : b.mean_cov > 4) low = true; } if (low) checks++; }
                if (ct.same_read_tag) checks++;
                if (std::find(genes.begin(), genes.end(), "intergenic") != genes.end()) checks++;
                if (checks > 1) drop = true;
            }
        }
        if (drop || !(o.var_filter & 4)) return false;
        std::vector<std::string> cbs; for (int v : cb) cbs.push_back(istr(v));
        out = row(join(genes), join(brk_str), join(cigars), join(mism), join(strands), join(repm), "trl", join(cbs), istr(disc));
        return true;
    }
};

// align_manager(meta).get_result() on already-final rows (sv_caller.py:820-832)
inline bool get_result(const Opts &o, const Region &rg, const Tables &tb, const Contig &ct, const std::vector<Psl> &rows, std::vector<std::string> &out)
{
    if (rows.empty()) return false;
    Caller c(o, rg, tb, ct); c.load(rows, true);
    if (c.check_indels()) return c.indel_result(out);
    if (c.check_svs()) return c.svs_result(out);
    return false;
}
inline bool target_hit(const Opts &o, const Region &rg, const Tables &tb, const Contig &ct, const std::vector<Psl> &rows)
{
    if (rows.empty()) return false;
    Caller c(o, rg, tb, ct); c.load(rows, false);
    return c.target_hit();
}


// ------------------------------------------------------------------------------------------------
// text context: options, annotation tables, per-region query_region / repeat mask / discordant pairs
struct Context {
    Opts opts; Tables tables; bool keep_tables = false;      // "keep_tables": the gene / repeat tables of the handle's previous context stay (the same for every batch of a run)
    std::vector<Region> regions; std::vector<std::vector<std::pair<std::string, int>>> partners; std::vector<std::string> rtags;
    // single-contig section (CPU test entry)
    std::string c_id, c_seq; int c_nkmers = 0, c_same = 0; std::vector<int> c_io, c_ot, c_kl; std::vector<Psl> rows; bool has_offset = false; int offset = 0; bool has_tname = false; std::string tname;
};
inline std::vector<std::string> split_ws(const std::string &ln) { std::vector<std::string> t; size_t i = 0; while (i < ln.size()) { while (i < ln.size() && (ln[i] == ' ' || ln[i] == '\t')) i++; size_t j = i; while (j < ln.size() && ln[j] != ' ' && ln[j] != '\t') j++; if (j > i) t.push_back(ln.substr(i, j - i)); i = j; } return t; }
inline std::vector<int> ints_csv(const std::string &s) { std::vector<int> v; size_t i = 0; while (i < s.size()) { size_t j = s.find(',', i); if (j == std::string::npos) j = s.size(); if (j > i) v.push_back(atoi(s.substr(i, j - i).c_str())); i = j + 1; } return v; }
inline std::string strip_chr(std::string s) { size_t f; while ((f = s.find("chr")) != std::string::npos) s.erase(f, 3); return s; }
inline bool parse_context(const char *text, Context &cx, std::string &err)
{
    std::string all(text); size_t pos = 0; Region *cur = nullptr;
    while (pos < all.size()) {
        size_t nl = all.find('\n', pos); if (nl == std::string::npos) nl = all.size();
        std::string ln = all.substr(pos, nl - pos); pos = nl + 1;
        std::vector<std::string> t = split_ws(ln); if (t.empty()) continue;
        const std::string &k = t[0]; auto I = [&](size_t i) { return atoi(t[i].c_str()); };
        if (k == "opts" && t.size() >= 9) { cx.opts.indel_size = I(1); cx.opts.trl_sr = I(2); cx.opts.indel_sr = I(3); cx.opts.rearr_sr = I(4); cx.opts.rearr_minseg = I(5); cx.opts.trl_minseg = I(6); cx.opts.keep_intron = I(7); cx.opts.var_filter = I(8); }
        else if (k == "gene" && t.size() >= 5) cx.tables.genes.push_back({t[1], t[2], I(3), I(4)});
        else if (k == "keep_tables") cx.keep_tables = true;
        else if (k == "arep_on") cx.tables.has_repeats = true;
        else if (k == "arep" && t.size() >= 5) cx.tables.repeats.push_back({t[1], I(2), I(3), I(4) != 0});
        else if (k == "region" && t.size() >= 6) { size_t idx = (size_t)I(1); if (cx.regions.size() <= idx) { cx.regions.resize(idx + 1); cx.partners.resize(idx + 1); cx.rtags.resize(idx + 1); } cur = &cx.regions[idx]; cur->chrom = t[2]; cur->start = I(3); cur->end = I(4); cur->name = t[5]; }
        else if (!cur && (k == "iv" || k == "trep" || k == "trep_on" || k == "inv" || k == "td" || k == "other" || k == "disc" || k == "partner" || k == "rtags")) { err = "context: '" + k + "' before any region"; return false; }
        else if (k == "iv" && t.size() >= 4) cur->intervals.push_back({I(1), I(2), I(3) != 0});
        else if (k == "trep_on") cur->has_target_repeats = true;
        else if (k == "trep" && t.size() >= 5) cur->target_repeats.push_back({t[1], I(2), I(3), I(4) != 0});
        else if ((k == "inv" || k == "td" || k == "other") && t.size() >= 5) { Pair5 p{I(1), I(2), I(3), I(4)}; (k == "inv" ? cur->inv : k == "td" ? cur->td : cur->other).push_back(p); }
        else if (k == "disc" && t.size() >= 4) cur->disc.push_back({t[1], I(2), I(3)});
        else if (k == "partner" && t.size() >= 3) cx.partners[cur - &cx.regions[0]].push_back({t[1], I(2)});
        else if (k == "rtags" && t.size() >= 2) cx.rtags[cur - &cx.regions[0]] = t[1];
        else if (k == "contig" && t.size() >= 5) { cx.c_id = t[1]; cx.c_seq = t[2]; cx.c_nkmers = I(3); cx.c_same = I(4); }
        else if (k == "io" && t.size() >= 2) cx.c_io = ints_csv(t[1]);
        else if (k == "ot" && t.size() >= 2) cx.c_ot = ints_csv(t[1]);
        else if (k == "klocs" && t.size() >= 2) cx.c_kl = ints_csv(t[1]);
        else if (k == "offset" && t.size() >= 2) { cx.has_offset = true; cx.offset = I(1); }
        else if (k == "tname" && t.size() >= 2) { cx.has_tname = true; cx.tname = t[1]; }
        else if (k == "row" && t.size() >= 22) {
            Psl p; p.matches = I(1); p.mis = I(2); p.rep = I(3); p.qni = I(5); p.qbi = I(6); p.tni = I(7); p.tbi = I(8); p.strand = t[9][0]; p.qsize = I(11); p.qstart = I(12); p.qend = I(13);
            p.tname = strip_chr(t[14]); p.tsize = I(15); p.tstart = I(16); p.tend = I(17); p.bs = ints_csv(t[19]); p.qs = ints_csv(t[20]); p.ts = ints_csv(t[21]);
            cx.rows.push_back(p);
        }
    }
    return true;
}
inline std::string join_row(const std::vector<std::string> &r) { std::string s; for (size_t i = 0; i < r.size(); i++) s += (i ? "\t" : "") + r[i]; return s; }

}  // namespace bkcall
