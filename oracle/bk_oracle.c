/* TEST INFRASTRUCTURE ONLY -- see bk_oracle.h.
 * CPU restatement of BreaKmer's hot path (olc.py, sv_assembly.py, k-mer set algebra of
 * sv_processor.py:609-645).  Written from the reference's observable semantics; every
 * function cites the lines it follows.  Canonicalisations P1 (int division), P2 (sorted set
 * iteration) and P4 (fq_recs in FASTQ order) of SURVEY.md 8c are part of the definition.
 */
#define _GNU_SOURCE
#include "bk_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static uint64_t g_cells = 0, g_calls = 0;
uint64_t bko_cells(int reset) { uint64_t v = g_cells; if (reset) g_cells = 0; return v; }
uint64_t bko_nw_calls(int reset) { uint64_t v = g_calls; if (reset) g_calls = 0; return v; }

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "bk_oracle: OOM\n"); abort(); } return p; }
static void *xcalloc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s ? s : 1); if (!p) { fprintf(stderr, "bk_oracle: OOM\n"); abort(); } return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "bk_oracle: OOM\n"); abort(); } return p; }

/* ------------------------------------------------------------------ olc.nw */
/* olc.py:40-107: overlap DP, rows = seq2 (n), cols = seq1 (m); scores olc.py:18-20. */
static int *nw_score = NULL; static unsigned char *nw_ptr = NULL; static size_t nw_cap = 0;

void bko_nw(const char *seq1, int m, const char *seq2, int n, int *out7, char *align1, char *align2)
{
    size_t need = (size_t)(n + 1) * (size_t)(m + 1);
    if (need > nw_cap) {
        nw_cap = need;
        nw_score = (int *)xrealloc(nw_score, nw_cap * sizeof(int));
        nw_ptr = (unsigned char *)xrealloc(nw_ptr, nw_cap);
    }
    const int W = m + 1;
    int *S = nw_score; unsigned char *P = nw_ptr;
    for (int i = 0; i <= n; i++) { S[(size_t)i * W] = 0; P[(size_t)i * W] = 1; }   /* olc.py:48-49,56-57 */
    for (int j = 0; j <= m; j++) { S[j] = 0; P[j] = 2; }                            /* olc.py:51-52,58-59 */
    for (int i = 1; i <= n; i++) {                                                  /* olc.py:62-74 */
        const int *up = S + (size_t)(i - 1) * W; int *cur = S + (size_t)i * W; unsigned char *pc = P + (size_t)i * W;
        const char b = seq2[i - 1];
        for (int j = 1; j <= m; j++) {
            int d = up[j - 1] + (seq1[j - 1] == b ? 1 : -2);     /* match_score olc.py:32-38 */
            int su = cur[j - 1] - 2;                              /* score_up   (pointer 2) */
            int sl = up[j] - 2;                                   /* score_left (pointer 1) */
            int s = sl > su ? sl : su; if (d > s) s = d;
            cur[j] = s;
            pc[j] = (s == d) ? 3 : (s == su) ? 2 : 1;             /* olc.py:69-74 */
        }
    }
    g_cells += (uint64_t)n * (uint64_t)m; g_calls++;
    int max_i = -200, i = 0;                                      /* olc.py:79-83: last row wins (>=) */
    for (int ii = 0; ii <= n; ii++) if (S[(size_t)ii * W + m] >= max_i) { max_i = S[(size_t)ii * W + m]; i = ii; }
    int j = m; const int prei = i, prej = j;
    /* traceback olc.py:88-105, strings built back to front */
    int cap = m + n + 1, a = cap, b2 = cap;
    char *t1 = (char *)xmalloc((size_t)cap + 1), *t2 = (char *)xmalloc((size_t)cap + 1);
    t1[cap] = 0; t2[cap] = 0;
    for (;;) {
        unsigned char p = P[(size_t)i * W + j];
        if (p == 3) { t1[--a] = seq1[j - 1]; t2[--b2] = seq2[i - 1]; i--; j--; }
        else if (p == 2) { t2[--b2] = '-'; t1[--a] = seq1[j - 1]; j--; }
        else { t2[--b2] = seq2[i - 1]; t1[--a] = '-'; i--; }
        if (i == 0 || j == 0) break;
    }
    out7[0] = cap - a; out7[1] = cap - b2; out7[2] = prej; out7[3] = j; out7[4] = prei; out7[5] = i; out7[6] = max_i;
    if (align1) { memcpy(align1, t1 + a, (size_t)(cap - a)); align1[cap - a] = 0; }
    if (align2) { memcpy(align2, t2 + b2, (size_t)(cap - b2)); align2[cap - b2] = 0; }
    free(t1); free(t2);
}

/* ------------------------------------------------------------------ small string-keyed hash map */
typedef struct { const char **keys; int *klen; int *val; size_t cap, n; } smap;
static uint64_t fnv(const char *s, int n) { uint64_t h = 1469598103934665603ULL; for (int i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ULL; } return h ^ (h >> 29); }
static void smap_init(smap *m, size_t expect) { size_t c = 16; while (c < expect * 2 + 2) c <<= 1; m->cap = c; m->n = 0; m->keys = (const char **)xcalloc(c, sizeof(char *)); m->klen = (int *)xcalloc(c, sizeof(int)); m->val = (int *)xcalloc(c, sizeof(int)); }
static void smap_free(smap *m) { free(m->keys); free(m->klen); free(m->val); }
static int *smap_find(smap *m, const char *k, int n) { size_t i = fnv(k, n) & (m->cap - 1); while (m->keys[i]) { if (m->klen[i] == n && memcmp(m->keys[i], k, (size_t)n) == 0) return &m->val[i]; i = (i + 1) & (m->cap - 1); } return NULL; }
static void smap_grow(smap *m);
static int *smap_put(smap *m, const char *k, int n, int v0) {
    if ((m->n + 1) * 2 > m->cap) smap_grow(m);
    size_t i = fnv(k, n) & (m->cap - 1);
    while (m->keys[i]) { if (m->klen[i] == n && memcmp(m->keys[i], k, (size_t)n) == 0) return &m->val[i]; i = (i + 1) & (m->cap - 1); }
    m->keys[i] = k; m->klen[i] = n; m->val[i] = v0; m->n++; return &m->val[i];
}
static void smap_grow(smap *m) {
    smap o = *m; m->cap = o.cap * 2; m->n = 0;
    m->keys = (const char **)xcalloc(m->cap, sizeof(char *)); m->klen = (int *)xcalloc(m->cap, sizeof(int)); m->val = (int *)xcalloc(m->cap, sizeof(int));
    for (size_t i = 0; i < o.cap; i++) if (o.keys[i]) *smap_put(m, o.keys[i], o.klen[i], 0) = o.val[i];
    smap_free(&o);
}

/* ------------------------------------------------------------------ T1 */
/* utils.py:239-244: fq_recs[seq].append(fr); iteration order = first occurrence (P4). */
int bko_group_reads(const char *reads, int stride, const int *lens, int n, int *out_rep, int *out_n)
{
    smap m; smap_init(&m, (size_t)n);
    int U = 0;
    for (int i = 0; i < n; i++) {
        const char *s = reads + (size_t)i * stride;
        int *v = smap_find(&m, s, lens[i]);
        if (v) out_n[*v]++;
        else { *smap_put(&m, s, lens[i], U) = U; out_rep[U] = i; out_n[U] = 1; U++; }
    }
    smap_free(&m);
    return U;
}

/* ------------------------------------------------------------------ K1 + K2 */
static int acgt_only(const char *s, int k) { for (int i = 0; i < k; i++) { char c = s[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return 0; } return 1; }
static void count_kmers(smap *m, const char *s, int len, int k) {                /* jellyfish count, all len-k+1 positions */
    for (int i = 0; i + k <= len; i++) if (acgt_only(s + i, k)) (*smap_put(m, s + i, k, 0))++;
}
static int cmp_merp(const void *a, const void *b, void *kk) { return memcmp(*(const char *const *)a, *(const char *const *)b, (size_t)*(int *)kk); }

int bko_kmer_select(const char *reads, int stride, const int *lens, int n,
                    const char *sc, int sc_stride, const int *sc_lens, int nsc,
                    const char *const *refs, const int *ref_lens, int nref,
                    int k, char *out_mers, int *out_counts, int cap)
{
    smap cs, rf, scm; size_t tot = 0;
    for (int i = 0; i < n; i++) tot += (size_t)(lens[i] >= k ? lens[i] - k + 1 : 0);
    smap_init(&cs, tot / 4 + 16);
    for (int i = 0; i < n; i++) count_kmers(&cs, reads + (size_t)i * stride, lens[i], k);     /* sv_processor.py:617-618 */
    /* reference: forward and reverse-complement window (sv_processor.py:613-615, files :291) */
    size_t rtot = 0; for (int r = 0; r < nref; r++) rtot += (size_t)ref_lens[r];
    smap_init(&rf, rtot * 2 + 16);
    char **rcs = (char **)xcalloc((size_t)nref, sizeof(char *));
    for (int r = 0; r < nref; r++) {
        count_kmers(&rf, refs[r], ref_lens[r], k);
        rcs[r] = (char *)xmalloc((size_t)ref_lens[r] + 1);
        for (int i = 0; i < ref_lens[r]; i++) { char c = refs[r][ref_lens[r] - 1 - i]; rcs[r][i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
        count_kmers(&rf, rcs[r], ref_lens[r], k);
    }
    int have_sc = nsc >= 0;
    if (have_sc) { size_t st = 0; for (int i = 0; i < nsc; i++) st += (size_t)sc_lens[i]; smap_init(&scm, st + 16); for (int i = 0; i < nsc; i++) count_kmers(&scm, sc + (size_t)i * sc_stride, sc_lens[i], k); }
    /* sample_only = (keys(case) & keys(case_sc)) - keys(ref); case_only[mer] = case[mer]  (:621-631) */
    const char **sel = (const char **)xmalloc(cs.n * sizeof(char *)); size_t ns = 0;
    for (size_t i = 0; i < cs.cap; i++) if (cs.keys[i]) {
        if (smap_find(&rf, cs.keys[i], k)) continue;
        if (have_sc && !smap_find(&scm, cs.keys[i], k)) continue;
        sel[ns++] = cs.keys[i];
    }
    qsort_r(sel, ns, sizeof(char *), cmp_merp, &k);
    for (size_t i = 0; i < ns && (int)i < cap; i++) { memcpy(out_mers + i * (size_t)k, sel[i], (size_t)k); out_counts[i] = *smap_find(&cs, sel[i], k); }
    free(sel); smap_free(&cs); smap_free(&rf); if (have_sc) smap_free(&scm);
    for (int r = 0; r < nref; r++) free(rcs[r]);
    free(rcs);
    return (int)ns;
}

/* ------------------------------------------------------------------ assembler */
enum { ORD_FOR = 0, ORD_REV = 1, ORD_MID = 2 };
typedef struct { int mer, pos, lt, dist, order; } ktup;           /* sv_assembly.py:130 tuple */
typedef struct { int u, pos, len, nreads; } rhit;                  /* read_search tuple, sv_assembly.py:106 */

typedef struct {
    char *seq; int len;
    int *io, *ot; int clen;                                        /* assembly_counts :160-221 */
    ktup *kmers; int nk, kcap;
    int setup, serial, in_fifo;
    int *reads; int nr;                                            /* self.reads (set) */
    int *batch; unsigned char *batch_aligned; int nb, bcap;        /* rb.batch_reads */
    int *alt; int *alt_n; int nalt;                                /* rb.alt */
    int *del; int ndel;                                            /* rb.delete */
    int *kmer_locs;
} contig;

typedef struct { int mer, read, nreads, alive; } pending;

struct bko_asm {
    /* inputs */
    const char *useqs; int stride; const int *ulens, *unreads; const uint8_t *uindel; int U;
    const char *mers; const int *counts; int M, k, rc_thresh, read_len;
    /* read state */
    unsigned char *used, *deleted; int *buf_stamp, *reads_stamp, *founder_of;
    /* mer state */
    unsigned char *removed, *usedmer; int *checked_stamp, *mset_stamp, *tmp_stamp; int *order;
    smap mermap; int stamp_ctr;
    /* FIFO buff.contigs */
    pending *pend; int phead, ptail, pcap;
    int serial;
    /* optional index behind find_reads (bko_set_find_index): first occurrence of every sample k-mer in every unique read */
    int *ix_off, *ix_u, *ix_pos;
    int *usedlist; int nusedlist;
    /* results */
    contig **out; int nout, outcap;
};

#define RSEQ(a, u) ((a)->useqs + (size_t)(u) * (a)->stride)
#define MER(a, i) ((a)->mers + (size_t)(i) * (a)->k)

static int mer_lookup(bko_asm *a, const char *s) { int *v = smap_find(&a->mermap, s, a->k); return (v && !a->removed[*v]) ? *v : -1; }   /* membership in akmers.smers_set */

static int cmp_ktup_mid(const void *x, const void *y) {           /* :142 sorted by (x[2], x[3]); stable via pos order of gathering */
    const ktup *p = (const ktup *)x, *q = (const ktup *)y;
    if (p->lt != q->lt) return p->lt - q->lt;
    if (p->dist != q->dist) return p->dist - q->dist;
    return p->pos - q->pos;                                        /* unreachable tie: (lt,dist) determines pos */
}
/* get_read_kmers_ordered, sv_assembly.py:126-143 (P1: integer m; Q1: last k-mer omitted) */
static int read_kmers_ordered(bko_asm *a, const char *seq, int len, int order, ktup **out)
{
    int l = a->k, m = len / 2, n = 0, cap = len > l ? len - l : 0;
    ktup *v = (ktup *)xmalloc((size_t)(cap + 1) * sizeof(ktup));
    for (int x = 0; x < len - l; x++) {
        int idx = mer_lookup(a, seq + x);
        if (idx < 0) continue;
        v[n].mer = idx; v[n].pos = x; v[n].lt = x < m; v[n].dist = abs(x - m); v[n].order = order; n++;
    }
    if (order == ORD_REV) { for (int i = 0; i < n / 2; i++) { ktup t = v[i]; v[i] = v[n - 1 - i]; v[n - 1 - i] = t; } }
    else if (order == ORD_MID) qsort(v, (size_t)n, sizeof(ktup), cmp_ktup_mid);
    *out = v; return n;
}

static int cmp_hit_for(const void *x, const void *y) {            /* :121 key (pos, -len), stable (tie -> fq_recs order = u) */
    const rhit *p = (const rhit *)x, *q = (const rhit *)y;
    if (p->pos != q->pos) return p->pos - q->pos;
    if (p->len != q->len) return q->len - p->len;
    return p->u - q->u;
}
static int cmp_hit_rev(const void *x, const void *y) {            /* :119 key (-pos, -len) */
    const rhit *p = (const rhit *)x, *q = (const rhit *)y;
    if (p->pos != q->pos) return q->pos - p->pos;
    if (p->len != q->len) return q->len - p->len;
    return p->u - q->u;
}
/* The reference's find_reads scans EVERY remaining read with a regex for every k-mer visit: O(visits x reads x L), hours
 * for one region of BASELINE configs[4] (10^5 visits x 24,000 reads).  bko_set_find_index(1) answers the same question
 * -- which reads contain the k-mer, and where first -- from an index built once per region (k-mer -> (read, first
 * position), reads in fq_recs order); the dynamic filters (deleted, buffered) and the sort are applied exactly as in
 * the scan.  Same result by construction, checked against the scan in tests/test_oracle_golden.py; default is the scan. */
static int g_find_index = 0;
void bko_set_find_index(int on) { g_find_index = on; }
static void build_find_index(bko_asm *a)
{
    const int M = a->M, k = a->k;
    int *last = (int *)xmalloc((size_t)(M + 1) * sizeof(int)); for (int i = 0; i < M; i++) last[i] = -1;
    a->ix_off = (int *)xcalloc((size_t)M + 2, sizeof(int));
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < M; i++) last[i] = -1;
        for (int u = 0; u < a->U; u++) {
            const char *s = RSEQ(a, u); const int len = a->ulens[u];
            for (int x = 0; x + k <= len; x++) {
                const int *v = smap_find(&a->mermap, s + x, k);
                if (!v || last[*v] == u) continue;                  /* first occurrence only (re.search) */
                last[*v] = u;
                if (pass == 0) a->ix_off[*v + 1]++;
                else { const int at = a->ix_off[*v]++; a->ix_u[at] = u; a->ix_pos[at] = x; }
            }
        }
        if (pass == 0) {
            for (int i = 0; i < M; i++) a->ix_off[i + 1] += a->ix_off[i];
            a->ix_u = (int *)xmalloc((size_t)(a->ix_off[M] + 1) * sizeof(int)); a->ix_pos = (int *)xmalloc((size_t)(a->ix_off[M] + 1) * sizeof(int));
        } else { for (int i = M; i > 0; i--) a->ix_off[i] = a->ix_off[i - 1]; a->ix_off[0] = 0; }
    }
    free(last);
}
/* find_reads sv_assembly.py:111-122 (+read_search :102-107). filter_serial<0: used_reads = set() */
static int find_reads(bko_asm *a, int mer, int rev, int filter_serial, rhit **out)
{
    int n = 0, cap = 64; rhit *v = (rhit *)xmalloc((size_t)cap * sizeof(rhit));
    const char *ms = MER(a, mer);
    if (a->ix_off) {
        for (int t = a->ix_off[mer]; t < a->ix_off[mer + 1]; t++) {
            const int u = a->ix_u[t];
            if (a->deleted[u]) continue;
            if (filter_serial >= 0 && a->buf_stamp[u] == filter_serial) continue;
            if (n == cap) { cap *= 2; v = (rhit *)xrealloc(v, (size_t)cap * sizeof(rhit)); }
            v[n].u = u; v[n].pos = a->ix_pos[t]; v[n].len = a->ulens[u]; v[n].nreads = a->unreads[u]; n++;
        }
        qsort(v, (size_t)n, sizeof(rhit), rev ? cmp_hit_rev : cmp_hit_for);
        *out = v; return n;
    }
    for (int u = 0; u < a->U; u++) {
        if (a->deleted[u]) continue;
        const char *s = RSEQ(a, u); int len = a->ulens[u];
        const char *hit = (const char *)memmem(s, (size_t)len, ms, (size_t)a->k);   /* re.search: first occurrence */
        if (!hit) continue;
        if (filter_serial >= 0 && a->buf_stamp[u] == filter_serial) continue;       /* :115-116 */
        if (n == cap) { cap *= 2; v = (rhit *)xrealloc(v, (size_t)cap * sizeof(rhit)); }
        v[n].u = u; v[n].pos = (int)(hit - s); v[n].len = len; v[n].nreads = a->unreads[u]; n++;
    }
    qsort(v, (size_t)n, sizeof(rhit), rev ? cmp_hit_rev : cmp_hit_for);
    *out = v; return n;
}

/* ---- assembly_counts (sv_assembly.py:160-221) */
static void set_counts(contig *c, int start, int end, int nreads, int indel_only)   /* :195-199, python slice clipping */
{
    int s = start < c->clen ? start : c->clen, e = end < c->clen ? end : c->clen;
    int *v = indel_only ? c->io : c->ot;
    for (int t = s; t < e; t++) v[t] += nreads;
}
static void extend_counts(contig *c, int l, int nreads, int indel_only, int post)     /* :201-221 */
{
    int nl = c->clen + l;
    int *io = (int *)xmalloc((size_t)nl * sizeof(int)), *ot = (int *)xmalloc((size_t)nl * sizeof(int));
    int off = post ? 0 : l, fill = post ? c->clen : 0;
    memcpy(io + off, c->io, (size_t)c->clen * sizeof(int)); memcpy(ot + off, c->ot, (size_t)c->clen * sizeof(int));
    for (int t = 0; t < l; t++) { io[fill + t] = indel_only ? nreads : 0; ot[fill + t] = indel_only ? 0 : nreads; }
    free(c->io); free(c->ot); c->io = io; c->ot = ot; c->clen = nl;
}
static void counts_superseq(contig *c, int rlen, int nreads, int indel_only, int start, int end)   /* :181-193 incl. Q8 */
{
    int s = start < rlen ? start : rlen, e = end < rlen ? end : rlen; if (e < s) e = s;
    int seg = e - s, z = seg < c->clen ? seg : c->clen;           /* zip() truncation */
    int nl = rlen - seg + z;
    int *io = (int *)xmalloc((size_t)(nl + 1) * sizeof(int)), *ot = (int *)xmalloc((size_t)(nl + 1) * sizeof(int));
    int bi = indel_only ? nreads : 0, bo = indel_only ? 0 : nreads, w = 0;
    for (int t = 0; t < s; t++, w++) { io[w] = bi; ot[w] = bo; }
    for (int t = 0; t < z; t++, w++) { io[w] = bi + c->io[t]; ot[w] = bo + c->ot[t]; }
    for (int t = e; t < rlen; t++, w++) { io[w] = bi; ot[w] = bo; }
    free(c->io); free(c->ot); c->io = io; c->ot = ot; c->clen = nl;
}
static int total_reads(const contig *c) { int a = 0, b = 0; for (int t = 0; t < c->clen; t++) { if (t == 0 || c->io[t] > a) a = c->io[t]; if (t == 0 || c->ot[t] > b) b = c->ot[t]; } return a + b; }   /* :178-179 */

/* ---- contig (sv_assembly.py:416-649) */
static contig *contig_new(bko_asm *a, int mer, int u, int nreads)                     /* :417-426 */
{
    contig *c = (contig *)xcalloc(1, sizeof(contig));
    c->len = a->ulens[u]; c->seq = (char *)xmalloc((size_t)c->len + 1); memcpy(c->seq, RSEQ(a, u), (size_t)c->len);
    c->clen = c->len; c->io = (int *)xcalloc((size_t)c->len, sizeof(int)); c->ot = (int *)xcalloc((size_t)c->len, sizeof(int));
    set_counts(c, 0, c->len, nreads, a->uindel[u]);                                   /* :165 */
    c->serial = ++a->serial;
    a->checked_stamp[mer] = c->serial;                                                /* checked_kmers = [kmer_val] */
    a->buf_stamp[u] = c->serial;                                                      /* buffer = set([read.id]) */
    c->bcap = 16; c->batch = (int *)xmalloc(16 * sizeof(int)); c->batch_aligned = (unsigned char *)xmalloc(16);
    c->batch[0] = u; c->batch_aligned[0] = 1; c->nb = 1;                              /* read_batch :383 */
    c->reads = (int *)xmalloc((size_t)a->U * sizeof(int));
    c->alt = (int *)xmalloc((size_t)a->U * sizeof(int)); c->alt_n = (int *)xmalloc((size_t)a->U * sizeof(int));
    c->del = (int *)xmalloc((size_t)a->U * sizeof(int));
    return c;
}
static void contig_free(contig *c) { if (!c) return; free(c->seq); free(c->io); free(c->ot); free(c->kmers); free(c->reads); free(c->batch); free(c->batch_aligned); free(c->alt); free(c->alt_n); free(c->del); free(c->kmer_locs); free(c); }

static void set_kmers(bko_asm *a, contig *c) { c->setup = 1; free(c->kmers); c->nk = read_kmers_ordered(a, c->seq, c->len, ORD_MID, &c->kmers); c->kcap = c->nk; }   /* :548-550 */
static void kmers_extend(contig *c, ktup *v, int n) { if (c->nk + n > c->kcap) { c->kcap = (c->nk + n) * 2 + 8; c->kmers = (ktup *)xrealloc(c->kmers, (size_t)c->kcap * sizeof(ktup)); } memcpy(c->kmers + c->nk, v, (size_t)n * sizeof(ktup)); c->nk += n; }

static void aseq_set_superseq(bko_asm *a, contig *c, int u, int nreads, int start, int end)    /* :232-236 */
{
    int rl = a->ulens[u];
    counts_superseq(c, rl, nreads, a->uindel[u], start, end);
    free(c->seq); c->seq = (char *)xmalloc((size_t)rl + 1); memcpy(c->seq, RSEQ(a, u), (size_t)rl); c->len = rl;
}
/* contig_overlap_read :506-528 */
static void contig_overlap_read(bko_asm *a, contig *c, const int *aln, int u, int nreads, int grow)
{
    if (aln[2] == c->len && aln[3] == 0) { aseq_set_superseq(a, c, u, nreads, aln[5], aln[4]); if (grow) set_kmers(a, c); return; }
    const char *rs = RSEQ(a, u); int rl = a->ulens[u], pl = rl - aln[4]; if (pl < 0) pl = 0;
    int km1 = a->k - 1, from = c->len - km1; if (from < 0) { from += c->len; if (from < 0) from = 0; }     /* python negative slice start */
    int nl = (c->len - from) + pl; char *nseq = (char *)xmalloc((size_t)nl + 1);
    memcpy(nseq, c->seq + from, (size_t)(c->len - from)); memcpy(nseq + (c->len - from), rs + aln[4], (size_t)pl);
    /* add_postseq :243-250 */
    int oldlen = c->len; (void)oldlen;
    c->seq = (char *)xrealloc(c->seq, (size_t)c->len + pl + 1); memcpy(c->seq + c->len, rs + aln[4], (size_t)pl); c->len += pl;
    set_counts(c, aln[3], aln[2], nreads, a->uindel[u]);
    extend_counts(c, pl, nreads, a->uindel[u], 1);
    if (grow) { ktup *v; int n = read_kmers_ordered(a, nseq, nl, ORD_FOR, &v); kmers_extend(c, v, n); free(v); }
    free(nseq);
}
/* read_overlap_contig :530-546 */
static void read_overlap_contig(bko_asm *a, contig *c, const int *aln, int u, int nreads, int grow)
{
    int rl = a->ulens[u];
    if (aln[2] == rl && aln[3] == 0) { set_counts(c, aln[5], aln[4], nreads, a->uindel[u]); return; }       /* add_subseq :238-240 */
    const char *rs = RSEQ(a, u); int pl = aln[3], km1 = a->k - 1, take = km1 < c->len ? km1 : c->len;
    int nl = pl + take; char *nseq = (char *)xmalloc((size_t)nl + 1);
    memcpy(nseq, rs, (size_t)pl); memcpy(nseq + pl, c->seq, (size_t)take);
    /* add_preseq :255-262 */
    char *ns = (char *)xmalloc((size_t)c->len + pl + 1); memcpy(ns, rs, (size_t)pl); memcpy(ns + pl, c->seq, (size_t)c->len);
    free(c->seq); c->seq = ns; c->len += pl;
    set_counts(c, aln[5], aln[4], nreads, a->uindel[u]);
    extend_counts(c, pl, nreads, a->uindel[u], 0);
    if (grow) { ktup *v; int n = read_kmers_ordered(a, nseq, nl, ORD_REV, &v); kmers_extend(c, v, n); free(v); }
    free(nseq);
}
static int strip_find(const char *al, int n, const char *mer, int k)                  /* x.replace('-','').find(mer) :485-488 */
{
    char *t = (char *)xmalloc((size_t)n + 1); int w = 0;
    for (int i = 0; i < n; i++) if (al[i] != '-') t[w++] = al[i];
    const char *h = (const char *)memmem(t, (size_t)w, mer, (size_t)k);
    int r = h ? (int)(h - t) : -1; free(t); return r;
}
/* check_align :449-504 */
static int check_align(bko_asm *a, contig *c, int u, int mer, int nreads, int grow)
{
    const char *rs = RSEQ(a, u); int rl = a->ulens[u];
    int v1[7], v2[7];
    char *a11 = (char *)xmalloc((size_t)c->len + rl + 2), *a12 = (char *)xmalloc((size_t)c->len + rl + 2);
    char *a21 = (char *)xmalloc((size_t)c->len + rl + 2), *a22 = (char *)xmalloc((size_t)c->len + rl + 2);
    bko_nw(c->seq, c->len, rs, rl, v1, a11, a12);                                     /* :451 */
    bko_nw(rs, rl, c->seq, c->len, v2, a21, a22);                                     /* :452 */
    int match = 0;
    int minlen = c->len < rl ? c->len : rl;
    /* :459-465 in exact integer form: score >= minlen/4.0  <=> 4*score >= minlen;
       round(score/ov, 2) >= 0.90 <=> 200*score >= 179*ov */
    int ok1 = (4 * v1[6] >= minlen) && (200 * v1[6] >= 179 * (v1[2] - v1[3]));
    int ok2 = (4 * v2[6] >= minlen) && (200 * v2[6] >= 179 * (v2[2] - v2[3]));
    if (!ok1 && !ok2) goto done;
    if (v1[6] == v2[6] && v1[3] == 0 && v1[5] == 0 && c->len == rl) { match = 1; goto done; }        /* :466-468 (Q9) */
    if (v1[6] == v2[6]) {
        match = 1;
        if (c->len < rl || (v1[2] == c->len && v1[3] == 0)) {                         /* :471-479 */
            aseq_set_superseq(a, c, u, nreads, v1[5], v1[4]);
            if (grow) set_kmers(a, c);
        } else if (rl < c->len || (v2[2] == rl && v2[3] == 0)) {                      /* :480-482 */
            set_counts(c, v2[5], v2[4], nreads, a->uindel[u]);
        } else {                                                                      /* :483-496 */
            match = 0;
            const char *ms = MER(a, mer);
            int i11 = strip_find(a11, v1[0], ms, a->k), i12 = strip_find(a12, v1[1], ms, a->k);
            int i21 = strip_find(a21, v2[0], ms, a->k), i22 = strip_find(a22, v2[1], ms, a->k);
            if (i11 > -1 && i12 > -1) {
                if ((i21 == -1 && i22 == -1) || (abs(i21 - i22) > abs(i11 - i12))) { match = 1; contig_overlap_read(a, c, v1, u, nreads, grow); }
            } else if (i21 > -1 && i22 > -1) {
                if ((i11 == -1 && i12 == -1) || (abs(i21 - i22) < abs(i11 - i12))) { match = 1; read_overlap_contig(a, c, v2, u, nreads, grow); }
            }
        }
    } else if (v1[6] > v2[6]) { match = 1; contig_overlap_read(a, c, v1, u, nreads, grow); }           /* :497-499 */
    else { match = 1; read_overlap_contig(a, c, v2, u, nreads, grow); }                                /* :500-503 */
done:
    free(a11); free(a12); free(a21); free(a22);
    return match;
}
/* check_read :552-566 (read_batch.check_mer_read :399-411 always appends and returns True, Q7) */
static int check_read(bko_asm *a, contig *c, int mer, int mer_count, int u, int nreads, int grow)
{
    a->buf_stamp[u] = c->serial;
    if (c->nb == c->bcap) { c->bcap *= 2; c->batch = (int *)xrealloc(c->batch, (size_t)c->bcap * sizeof(int)); c->batch_aligned = (unsigned char *)xrealloc(c->batch_aligned, (size_t)c->bcap); }
    c->batch[c->nb] = u; c->batch_aligned[c->nb] = 0; c->nb++;
    int match = check_align(a, c, u, mer, nreads, grow);
    if (match) { a->used[u] = 1; c->batch_aligned[c->nb - 1] = 1; return 1; }
    if (mer_count > 2 && !a->used[u]) { c->alt[c->nalt] = u; c->alt_n[c->nalt] = nreads; c->nalt++; }
    else c->del[c->ndel++] = u;
    return 0;
}
static void fifo_push(bko_asm *a, int mer, int u, int nreads)                         /* buffer.add_contig :337-340 */
{
    if (a->founder_of[u] >= 0 || a->used[u]) return;
    if (a->ptail == a->pcap) { a->pcap = a->pcap * 2 + 16; a->pend = (pending *)xrealloc(a->pend, (size_t)a->pcap * sizeof(pending)); }
    a->pend[a->ptail].mer = mer; a->pend[a->ptail].read = u; a->pend[a->ptail].nreads = nreads; a->pend[a->ptail].alive = 1;
    a->founder_of[u] = a->ptail++; a->used[u] = 1;
}
static void fifo_remove(bko_asm *a, int u) { int p = a->founder_of[u]; if (p >= 0) { a->pend[p].alive = 0; a->founder_of[u] = -1; } }   /* remove_contig :342-344 */

static int cmp_mer_asc_ctx_k; static const char *cmp_mer_asc_ctx_mers;
static int cmp_mer_asc(const void *x, const void *y) { return memcmp(cmp_mer_asc_ctx_mers + (size_t)(*(const int *)x) * cmp_mer_asc_ctx_k, cmp_mer_asc_ctx_mers + (size_t)(*(const int *)y) * cmp_mer_asc_ctx_k, (size_t)cmp_mer_asc_ctx_k); }

/* check_alt_reads :568-582.  set(self.kmers) holds 5-tuples, so it never removes a string (Q-note);
   iteration over x is sorted() (P2). */
static void check_alt_reads(bko_asm *a, contig *c)
{
    int fin = ++a->stamp_ctr;                     /* identifies mer_set of this call */
    int xcap = a->read_len + 8; int *x = (int *)xmalloc((size_t)xcap * sizeof(int));
    for (int t = 0; t < c->nalt; t++) {
        int u = c->alt[t]; const char *s = RSEQ(a, u); int len = a->ulens[u], nx = 0;
        int tmp = ++a->stamp_ctr;
        if (len + 8 > xcap) { xcap = len + 8; x = (int *)xrealloc(x, (size_t)xcap * sizeof(int)); }
        for (int p = 0; p < len - a->k; p++) {                                        /* get_read_kmers :147-155 (Q1) */
            int idx = mer_lookup(a, s + p);
            if (idx < 0 || a->usedmer[idx] || a->mset_stamp[idx] == fin || a->tmp_stamp[idx] == tmp) continue;
            a->tmp_stamp[idx] = tmp; x[nx++] = idx;
        }
        if (nx == 0) continue;                                                        /* :574 */
        cmp_mer_asc_ctx_k = a->k; cmp_mer_asc_ctx_mers = a->mers;
        qsort(x, (size_t)nx, sizeof(int), cmp_mer_asc);                               /* P2: sorted(x) */
        for (int q = 0; q < nx; q++) {
            if (a->counts[x[q]] > 1) {                                                /* :576-581 */
                int seed = x[q];
                for (int z = 0; z < nx; z++) a->mset_stamp[x[z]] = fin;               /* mer_set = mer_set | x */
                /* The reference collects new_contigs and adds them to buff in finalize (:591-592) in alt
                   order; add_contig depends only on read.used / membership, which later alt iterations do
                   not touch, so pushing here is equivalent.  mer_pos is stored but never read. */
                fifo_push(a, seed, u, c->alt_n[t]);
                break;
            }
        }
    }
    free(x);
}
/* finalize :584-599 */
static void finalize(bko_asm *a, contig *c, int setup)
{
    if (setup) set_kmers(a, c);
    check_alt_reads(a, c);
    int last = -1;
    for (int t = 0; t < c->nb; t++) if (c->batch_aligned[t]) {                        /* keep_reads :595-597 */
        int u = c->batch[t]; last = u;
        if (a->reads_stamp[u] != c->serial) { a->reads_stamp[u] = c->serial; c->reads[c->nr++] = u; }
    }
    for (int t = 0; t < c->ndel; t++) a->deleted[c->del[t]] = 1;                      /* rb.clean :390 */
    c->ndel = 0; c->nalt = 0;
    c->batch[0] = last; c->batch_aligned[0] = 1; c->nb = 1;                           /* :396 */
}
static void set_kmer_locs(bko_asm *a, contig *c)                                      /* :434-438 */
{
    free(c->kmer_locs); c->kmer_locs = (int *)xcalloc((size_t)c->len + 1, sizeof(int));
    for (int t = 0; t < c->nk; t++) {
        const char *h = (const char *)memmem(c->seq, (size_t)c->len, MER(a, c->kmers[t].mer), (size_t)a->k);
        int p = h ? (int)(h - c->seq) : -1;
        int s = p, e = p + a->k;
        if (s < 0) { s += c->len; if (s < 0) s = 0; }                                 /* python slice semantics for find()==-1 */
        if (e < 0) { e += c->len; if (e < 0) e = 0; }
        if (s > c->len) s = c->len;
        if (e > c->len) e = c->len;
        for (int q = s; q < e; q++) c->kmer_locs[q]++;
    }
}
/* grow :616-649 */
static void grow(bko_asm *a, contig *c)
{
    if (!c->setup) set_kmers(a, c);
    for (;;) {
        int n = 0; ktup *nk = (ktup *)xmalloc((size_t)(c->nk + 1) * sizeof(ktup));
        for (int t = 0; t < c->nk; t++) if (a->checked_stamp[c->kmers[t].mer] != c->serial) nk[n++] = c->kmers[t];    /* refresh_kmers :601-602 */
        if (n == 0) { free(nk); break; }
        for (int t = 0; t < n; t++) {
            ktup kv = nk[t];
            int rev = 0;                                                              /* get_mer_reads :604-614 */
            if (kv.order == ORD_MID) { if (kv.lt == 0) rev = 1; } else if (kv.order == ORD_FOR) rev = 1;
            rhit *hits; int nh = find_reads(a, kv.mer, rev, c->serial, &hits);
            if (!a->usedmer[kv.mer]) { a->usedmer[kv.mer] = 1; a->usedlist[a->nusedlist++] = kv.mer; }
            for (int q = 0; q < nh; q++) {
                if (check_read(a, c, kv.mer, a->counts[kv.mer], hits[q].u, hits[q].nreads, 1)) fifo_remove(a, hits[q].u);
            }
            free(hits);
            finalize(a, c, 0);
            a->checked_stamp[kv.mer] = c->serial;      /* checked_kmers.append(mer): stamp is per contig */
        }
        free(nk);
    }
    set_kmer_locs(a, c);
}
/* the per-contig checked set must survive other contigs' stamps: stamps are (serial) values, and only one
   contig is active at a time (FIFO pops one, grows it to completion), so a single stamp array suffices. */

static void keep_or_drop(bko_asm *a, contig *c)                                       /* init_assembly :53-59 */
{
    if (total_reads(c) < a->rc_thresh || c->len <= a->read_len) { contig_free(c); return; }
    if (a->nout == a->outcap) { a->outcap = a->outcap * 2 + 4; a->out = (contig **)xrealloc(a->out, (size_t)a->outcap * sizeof(contig *)); }
    a->out[a->nout++] = c;
}
/* setup_contigs :11-26 */
static void setup_contigs(bko_asm *a, int mer)
{
    rhit *hits; int nh = find_reads(a, mer, 0, -1, &hits);
    if (!a->usedmer[mer]) { a->usedmer[mer] = 1; a->usedlist[a->nusedlist++] = mer; }
    contig *ct = NULL;
    for (int q = 0; q < nh; q++) {
        int u = hits[q].u;
        if (!ct) {
            ct = contig_new(a, mer, u, hits[q].nreads);
            if (a->founder_of[u] < 0 && !a->used[u]) { ct->in_fifo = 1; a->used[u] = 1; }    /* buff.add_contig :337-340 */
        } else check_read(a, ct, mer, a->counts[mer], u, hits[q].nreads, 0);
    }
    free(hits);
    if (!ct) return;
    finalize(a, ct, 1);
    if (ct->in_fifo) { grow(a, ct); keep_or_drop(a, ct); }                            /* it is the FIFO head (:50-52) */
    else contig_free(ct);
}

static const char *cmp_order_mers; static const int *cmp_order_counts; static int cmp_order_k;
static int cmp_order(const void *x, const void *y)                                    /* kmers.get_all_kmer_values :281, reverse=True */
{
    int i = *(const int *)x, j = *(const int *)y;
    if (cmp_order_counts[i] != cmp_order_counts[j]) return cmp_order_counts[j] - cmp_order_counts[i];
    return memcmp(cmp_order_mers + (size_t)j * cmp_order_k, cmp_order_mers + (size_t)i * cmp_order_k, (size_t)cmp_order_k);
}

bko_asm *bko_init_assembly(const char *useqs, int stride, const int *ulens, const int *unreads,
                           const uint8_t *uindel, int U,
                           const char *mers, const int *counts, int M,
                           int k, int rc_thresh, int read_len)
{
    bko_asm *a = (bko_asm *)xcalloc(1, sizeof(bko_asm));
    a->useqs = useqs; a->stride = stride; a->ulens = ulens; a->unreads = unreads; a->uindel = uindel; a->U = U;
    a->mers = mers; a->counts = counts; a->M = M; a->k = k; a->rc_thresh = rc_thresh; a->read_len = read_len;
    a->used = (unsigned char *)xcalloc((size_t)U, 1); a->deleted = (unsigned char *)xcalloc((size_t)U, 1);
    a->buf_stamp = (int *)xcalloc((size_t)U, sizeof(int)); a->reads_stamp = (int *)xcalloc((size_t)U, sizeof(int));
    a->founder_of = (int *)xmalloc((size_t)(U + 1) * sizeof(int)); for (int u = 0; u < U; u++) a->founder_of[u] = -1;
    a->removed = (unsigned char *)xcalloc((size_t)M, 1); a->usedmer = (unsigned char *)xcalloc((size_t)M, 1);
    a->checked_stamp = (int *)xcalloc((size_t)M, sizeof(int)); a->mset_stamp = (int *)xcalloc((size_t)M, sizeof(int)); a->tmp_stamp = (int *)xcalloc((size_t)M, sizeof(int));
    a->order = (int *)xmalloc((size_t)(M + 1) * sizeof(int));
    a->usedlist = (int *)xmalloc((size_t)(M + 1) * sizeof(int)); a->nusedlist = 0;
    smap_init(&a->mermap, (size_t)M);
    if (M == 0) return a;                                                             /* :33-34 */
    int no = 0;
    for (int i = 0; i < M; i++) {
        *smap_put(&a->mermap, MER(a, i), k, i) = i;
        int multi = 0; for (int t = 1; t < k; t++) if (MER(a, i)[t] != MER(a, i)[0]) { multi = 1; break; }
        if (multi) a->order[no++] = i; else a->removed[i] = 1;                        /* kmers.add_kmer :276-278 */
    }
    cmp_order_mers = mers; cmp_order_counts = counts; cmp_order_k = k;
    qsort(a->order, (size_t)no, sizeof(int), cmp_order);
    if (g_find_index) build_find_index(a);
    int head = 0;
    for (;;) {                                                                        /* :43-62 */
        while (head < no && a->removed[a->order[head]]) head++;
        if (head == no || counts[a->order[head]] < 2) break;                          /* has_mers :318-322 (sorted desc => head is max) */
        int mer = a->order[head];
        setup_contigs(a, mer);
        while (a->phead < a->ptail) {                                                 /* :50-59 */
            pending p = a->pend[a->phead++];
            if (!p.alive) continue;
            a->founder_of[p.read] = -1;
            contig *c = contig_new(a, p.mer, p.read, p.nreads);
            grow(a, c); keep_or_drop(a, c);
        }
        for (int t = 0; t < a->nusedlist; t++) { const int i = a->usedlist[t]; a->removed[i] = 1; a->usedmer[i] = 0; }      /* remove_kmers :358-360 */
        a->nusedlist = 0;
        /* buff.remove_reads is a no-op (ids vs sequence keys, Q7) */
    }
    return a;
}
int bko_asm_ncontigs(const bko_asm *a) { return a->nout; }
int bko_asm_contig_len(const bko_asm *a, int c) { return a->out[c]->len; }
int bko_asm_contig_clen(const bko_asm *a, int c) { return a->out[c]->clen; }
int bko_asm_contig_nkmers(const bko_asm *a, int c) { return a->out[c]->nk; }
int bko_asm_contig_nreads(const bko_asm *a, int c) { return a->out[c]->nr; }
static int cmp_int(const void *x, const void *y) { return *(const int *)x - *(const int *)y; }
void bko_asm_contig_get(const bko_asm *a, int ci, char *seq, int *indel_only, int *others, int *kmer_locs, int *kmer_idx, int *read_idx)
{
    const contig *c = a->out[ci];
    if (seq) memcpy(seq, c->seq, (size_t)c->len);
    if (indel_only) memcpy(indel_only, c->io, (size_t)c->clen * sizeof(int));
    if (others) memcpy(others, c->ot, (size_t)c->clen * sizeof(int));
    if (kmer_locs) memcpy(kmer_locs, c->kmer_locs, (size_t)c->len * sizeof(int));
    if (kmer_idx) for (int t = 0; t < c->nk; t++) kmer_idx[t] = c->kmers[t].mer;
    if (read_idx) { memcpy(read_idx, c->reads, (size_t)c->nr * sizeof(int)); qsort(read_idx, (size_t)c->nr, sizeof(int), cmp_int); }
}
void bko_asm_read_flags(const bko_asm *a, uint8_t *flags) { for (int u = 0; u < a->U; u++) flags[u] = (uint8_t)((a->used[u] ? 1 : 0) | (a->deleted[u] ? 2 : 0)); }
void bko_asm_free(bko_asm *a)
{
    if (!a) return;
    for (int i = 0; i < a->nout; i++) contig_free(a->out[i]);
    free(a->ix_off); free(a->ix_u); free(a->ix_pos); free(a->usedlist);
    free(a->out); free(a->used); free(a->deleted); free(a->buf_stamp); free(a->reads_stamp); free(a->founder_of);
    free(a->removed); free(a->usedmer); free(a->checked_stamp); free(a->mset_stamp); free(a->tmp_stamp); free(a->order);
    smap_free(&a->mermap); free(a->pend); free(a);
}

/* ------------------------------------------------------------------ G2 test hook
 * Reproduces tools/make_golden.py g2(): a contig founded on `contig_seq` (nreads = founder_nreads),
 * sample k-mer set = every k-mer of contig_seq and read_seq, optional earlier read `pre`
 * (nreads 1), set_kmers() when grow, then ONE check_align(read).  Outputs the state after. */
int bko_check_align_case(const char *contig_seq, int clen, const char *read_seq, int rlen, const char *pre, int plen,
                         const char *mer, int k, int grow, int nreads, int indel_only, int founder_nreads,
                         char *out_seq, int *out_len, int *out_io, int *out_ot, int *out_clen,
                         char *out_kmers /* nk*k chars */, int *out_kmeta /* nk*4: pos,lt,dist,order */, int *out_nk)
{
    int stride = clen; if (rlen > stride) stride = rlen; if (plen > stride) stride = plen;
    int U = pre ? 3 : 2;
    char *useqs = (char *)xcalloc((size_t)U * stride + 1, 1);
    int ulens[3], unreads[3] = { founder_nreads, nreads, 1 }; uint8_t uindel[3] = { 0, (uint8_t)indel_only, 0 };
    memcpy(useqs, contig_seq, (size_t)clen); ulens[0] = clen;
    memcpy(useqs + stride, read_seq, (size_t)rlen); ulens[1] = rlen;
    if (pre) { memcpy(useqs + 2 * (size_t)stride, pre, (size_t)plen); ulens[2] = plen; }
    /* sample k-mer set */
    smap m; smap_init(&m, (size_t)(clen + rlen));
    int M = 0; char *mers = (char *)xmalloc((size_t)(clen + rlen + 2) * k);
    const char *srcs[2] = { contig_seq, read_seq }; int sl[2] = { clen, rlen };
    for (int s = 0; s < 2; s++) for (int i = 0; i + k <= sl[s]; i++) if (!smap_find(&m, srcs[s] + i, k)) { memcpy(mers + (size_t)M * k, srcs[s] + i, (size_t)k); *smap_put(&m, mers + (size_t)M * k, k, M) = M; M++; }
    int *counts = (int *)xmalloc((size_t)(M + 1) * sizeof(int)); for (int i = 0; i < M; i++) counts[i] = 3;
    smap_free(&m);
    /* context without running init_assembly's loop */
    bko_asm *a = (bko_asm *)xcalloc(1, sizeof(bko_asm));
    a->useqs = useqs; a->stride = stride; a->ulens = ulens; a->unreads = unreads; a->uindel = uindel; a->U = U;
    a->mers = mers; a->counts = counts; a->M = M; a->k = k; a->rc_thresh = 2; a->read_len = stride;
    a->used = (unsigned char *)xcalloc((size_t)U, 1); a->deleted = (unsigned char *)xcalloc((size_t)U, 1);
    a->buf_stamp = (int *)xcalloc((size_t)U, sizeof(int)); a->reads_stamp = (int *)xcalloc((size_t)U, sizeof(int));
    a->founder_of = (int *)xmalloc((size_t)(U + 1) * sizeof(int)); for (int u = 0; u < U; u++) a->founder_of[u] = -1;
    a->removed = (unsigned char *)xcalloc((size_t)M + 1, 1); a->usedmer = (unsigned char *)xcalloc((size_t)M + 1, 1);
    a->checked_stamp = (int *)xcalloc((size_t)M + 1, sizeof(int)); a->mset_stamp = (int *)xcalloc((size_t)M + 1, sizeof(int)); a->tmp_stamp = (int *)xcalloc((size_t)M + 1, sizeof(int));
    a->order = (int *)xmalloc((size_t)(M + 1) * sizeof(int));
    a->usedlist = (int *)xmalloc((size_t)(M + 1) * sizeof(int)); a->nusedlist = 0;
    smap_init(&a->mermap, (size_t)M);
    for (int i = 0; i < M; i++) *smap_put(&a->mermap, MER(a, i), k, i) = i;
    int *mi = smap_find(&a->mermap, mer, k); int meridx = mi ? *mi : 0;
    contig *c = contig_new(a, meridx, 0, founder_nreads);
    if (pre) check_align(a, c, 2, meridx, 1, grow);
    if (grow) set_kmers(a, c);
    int match = check_align(a, c, 1, meridx, nreads, grow);
    memcpy(out_seq, c->seq, (size_t)c->len); *out_len = c->len; *out_clen = c->clen;
    memcpy(out_io, c->io, (size_t)c->clen * sizeof(int)); memcpy(out_ot, c->ot, (size_t)c->clen * sizeof(int));
    *out_nk = c->nk;
    for (int t = 0; t < c->nk; t++) { memcpy(out_kmers + (size_t)t * k, MER(a, c->kmers[t].mer), (size_t)k); out_kmeta[4 * t] = c->kmers[t].pos; out_kmeta[4 * t + 1] = c->kmers[t].lt; out_kmeta[4 * t + 2] = c->kmers[t].dist; out_kmeta[4 * t + 3] = c->kmers[t].order; }
    contig_free(c); bko_asm_free(a); free(useqs); free(mers); free(counts);
    return match;
}

/* ------------------------------------------------------------------ R2 realign (see bk_oracle.h) */
static uint64_t g_sw_cells = 0;
uint64_t bko_sw_cells(int reset) { uint64_t v = g_sw_cells; if (reset) g_sw_cells = 0; return v; }
#define SW_MAXHITS 2048      /* the hits tile the query (>= 20 bases each): contigs up to 40,000 bases; the oracle aborts beyond */
#define SW_MAXSEC (1 << 22)
#define SW_EQ(a, b) ((a) == (b) && (a) != 'N')      /* an N (contig or window) matches nothing in the realign stage */
typedef struct { int qs, qe, ts, te, strand, tidx, score, nb; int bs[1], bq[1], bt[1]; int fq; } swhit;   /* q coords are strand coords; fq = forward start */

/* Gap-free local Smith-Waterman (maximal scoring segment): H[a][b] = max(0, H[a-1][b-1] + s(q_a, t_b)).
 * Gaps are NOT opened inside a hit -- like BLAT, hits are ungapped blocks and gaps only appear when
 * collinear hits are chained (a finite linear gap penalty smears long inserts over coincidental
 * matches).  Best end = max score, then smallest query end a, then smallest target
 * end b; the start is where the positive run on that diagonal began. */
static int sw_local(const char *q, int n, const char *t, int m, int *a0, int *a1, int *b0, int *b1)
{
    int *H = (int *)xcalloc((size_t)(m + 1) * 2, sizeof(int)), *R = (int *)xcalloc((size_t)(m + 1) * 2, sizeof(int));
    int best = 0, ba = 0, bb = 0, br = 0;
    for (int a = 1; a <= n; a++) {
        int *hp = H + ((a - 1) & 1) * (m + 1), *hc = H + (a & 1) * (m + 1), *rp = R + ((a - 1) & 1) * (m + 1), *rc_ = R + (a & 1) * (m + 1);
        hc[0] = 0; rc_[0] = 0;
        for (int b = 1; b <= m; b++) {
            int s = hp[b - 1] + (SW_EQ(q[a - 1], t[b - 1]) ? 1 : -2), run = rp[b - 1] + 1;
            if (s <= 0) { s = 0; run = 0; }
            hc[b] = s; rc_[b] = run;
            if (s > best) { best = s; ba = a; bb = b; br = run; }    /* strict: smallest a, then smallest b */
        }
    }
    g_sw_cells += (uint64_t)n * (uint64_t)m;
    *a0 = ba - br; *a1 = ba; *b0 = bb - br; *b1 = bb;
    free(H); free(R);
    return best;
}
static int sw_blocks(const char *q, const char *t, int a0, int a1, int b0, int b1, int score, int *bs, int *bq, int *bt)
{
    (void)q; (void)t; (void)score; (void)b1;
    bs[0] = a1 - a0; bq[0] = a0; bt[0] = b0;
    return 1;
}
static int cmp_hit_fq(const void *x, const void *y) { return ((const swhit *)x)->fq - ((const swhit *)y)->fq; }
typedef struct { int qs, qe, ts, te, strand, tidx, score; } swsec;                  /* strand coordinates */
static int cmp_sec(const void *x, const void *y)
{
    const swsec *a = (const swsec *)x, *b = (const swsec *)y;
    if (a->score != b->score) return b->score - a->score;
    if (a->tidx != b->tidx) return a->tidx - b->tidx;
    if (a->strand != b->strand) return a->strand - b->strand;
    if (a->qe != b->qe) return a->qe - b->qe;
    return a->te - b->te;
}

/* ---- R2 step 4: island fill (bk_oracle.h).  A flank shorter than min_score between two differences cannot anchor a
 * segment of its own; it is recovered here, next to the anchors it belongs to: inside the unaligned rectangle between two
 * chained blocks (or beyond the first / last block) the best gap-free segment on a diagonal within FILL_BAND of a
 * neighbouring block's diagonal becomes a block if it scores >= FILL_MIN; the two rectangles it leaves are filled the
 * same way.  Ties: higher score, then smaller query end, then smaller target end. */
#define FILL_BAND 16
#define FILL_MIN 8
typedef struct { int qs, qe, ts, te, score; } fblk;
static int fill_best(const char *q, const char *t, int qa, int qb, int ta, int tb, int use1, int d1, int use2, int d2, fblk *out)
{
    int best = 0; out->score = 0;
    int dlo = use1 ? d1 - FILL_BAND : d2 - FILL_BAND, dhi = use1 ? d1 + FILL_BAND : d2 + FILL_BAND;
    if (use2) { if (d2 - FILL_BAND < dlo) dlo = d2 - FILL_BAND; if (d2 + FILL_BAND > dhi) dhi = d2 + FILL_BAND; }
    for (int d = dlo; d <= dhi; d++) {
        int in1 = use1 && d >= d1 - FILL_BAND && d <= d1 + FILL_BAND, in2 = use2 && d >= d2 - FILL_BAND && d <= d2 + FILL_BAND;
        if (!in1 && !in2) continue;
        int a0 = qa > ta - d ? qa : ta - d, a1 = qb < tb - d ? qb : tb - d, h = 0, run = 0;
        for (int a = a0; a < a1; a++) {
            h += SW_EQ(q[a], t[a + d]) ? 1 : -2; run++;
            if (h <= 0) { h = 0; run = 0; continue; }
            int qe = a + 1, te = a + 1 + d;
            if (h > best || (h == best && (qe < out->qe || (qe == out->qe && te < out->te)))) { best = h; out->qs = qe - run; out->qe = qe; out->ts = te - run; out->te = te; out->score = h; }
        }
    }
    return best;
}
static void fill_gap(const char *q, const char *t, const fblk *L, const fblk *R, int qlo, int qhi, int tlo, int thi, fblk *v, int *n, int cap)
{
    if (qhi - qlo < FILL_MIN || thi - tlo < FILL_MIN || *n >= cap) return;
    fblk nb;
    int sc = fill_best(q, t, qlo, qhi, tlo, thi, L != NULL, L ? L->te - L->qe : 0, R != NULL, R ? R->ts - R->qs : 0, &nb);
    if (sc < FILL_MIN) return;
    fill_gap(q, t, L, &nb, qlo, nb.qs, tlo, nb.ts, v, n, cap);
    if (*n < cap) v[(*n)++] = nb;
    fill_gap(q, t, &nb, R, nb.qe, qhi, nb.te, thi, v, n, cap);
}

/* BLAT's published seeding rule for the command line the reference uses (sv_processor.py:843: -stepSize=10 -minMatch=2, tile size
 * 11 by default): the target is indexed by the 11-mers that start at every 10th base of it, and an alignment is only ever looked
 * for where TWO of those tiles match the query exactly on one diagonal.  A gap-free segment is seedable iff it holds two such
 * tiles: target positions 10 j .. 10 j + 10 inside the segment, all eleven bases matching. */
#define SEED_TILE 11
#define SEED_STEP 10
#define SEED_MIN 2
static int sw_seedable(const char *q, const char *t, int qs, int ts, int len)
{
    int tiles = 0;
    for (int j = (ts + SEED_STEP - 1) / SEED_STEP * SEED_STEP; j + SEED_TILE <= ts + len; j += SEED_STEP) {
        int ok = 1;
        for (int z = 0; z < SEED_TILE && ok; z++) ok = SW_EQ(q[qs + (j - ts) + z], t[j + z]);
        tiles += ok;
    }
    return tiles >= SEED_MIN;
}

/* ---- R2 step 8 (bk_oracle.h): BLAT's documented output filters, applied to every record before it is reported.
 * The reference's command lines (sv_processor.py:840, 843) give -minScore=20 and leave -minIdentity at its default; BLAT's usage
 * text: "-minScore=N sets minimum score.  This is the matches minus the mismatches minus some sort of gap penalty" and
 * "-minIdentity=N Sets minimum sequence identity (in percent).  Default is 90 for nucleotide searches".
 *   score    = matches + repMatches - misMatches - qNumInsert - tNumInsert          >= min_score
 *   identity = 100 - milliBad / 10 with the milliBad of the PSL record exactly as the reference's caller computes it
 *              (sv_caller.py:954-968, blat_res.calcMilliBad: UCSC's pslCalcMilliBad for DNA)  >= 90
 *            <=> 10 * (misMatches + qNumInsert + round(3 ln(1 + max(0, qAliSize - tAliSize)))) <= matches + repMatches + misMatches
 * (integer exact: every term is an integer; 3 ln n is never a half-integer, so the rounding mode does not matter). */
#define BKO_MIN_IDENTITY 90
int bko_psl_passes(const bko_psl *r, int min_score)
{
    const long total = (long)r->matches + r->rep_matches + r->mismatches;
    if ((long)r->matches + r->rep_matches - r->mismatches - r->q_num_insert - r->t_num_insert < (long)min_score) return 0;
    const int qali = r->q_end - r->q_start, tali = r->t_end - r->t_start;
    const int ali = qali < tali ? qali : tali;
    if (ali <= 0 || total == 0) return 1;                                  /* calcMilliBad returns 0: identity 100 */
    const int dif = qali - tali > 0 ? qali - tali : 0;
    const long bad = (long)r->mismatches + r->q_num_insert + lround(3.0 * log(1.0 + (double)dif));
    return 100L * bad <= (long)(100 - BKO_MIN_IDENTITY) * total;           /* 100 - 100 bad / total >= BKO_MIN_IDENTITY */
}

int bko_realign(const char *contig, int Q, const char *const *targets_in, const int *tlens, int ntargets,
                int min_score, int min_seg, bko_psl *out, int cap)
{
    /* soft-masked (lower-case) window bases are the same bases; a match on one is reported as repMatches (BLAT -repeats=lower,
     * sv_processor.py:843; consumed at sv_caller.py:913, 975-986): upper-cased copies + one flag byte per base */
    char **targets = (char **)xmalloc((size_t)ntargets * sizeof(char *)); unsigned char **soft = (unsigned char **)xmalloc((size_t)ntargets * sizeof(unsigned char *));
    for (int ti = 0; ti < ntargets; ti++) {
        targets[ti] = (char *)xmalloc((size_t)tlens[ti] + 1); soft[ti] = (unsigned char *)xcalloc((size_t)tlens[ti] + 1, 1);
        for (int z = 0; z < tlens[ti]; z++) { char c = targets_in[ti][z]; if (c >= 'a' && c <= 'z') { soft[ti][z] = 1; c = (char)(c - 'a' + 'A'); } targets[ti][z] = c; }
        targets[ti][tlens[ti]] = 0;
    }
    char *rc = (char *)xmalloc((size_t)Q + 1);
    for (int i = 0; i < Q; i++) { char c = contig[Q - 1 - i]; rc[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
    swhit *hits = (swhit *)xmalloc(sizeof(swhit) * SW_MAXHITS); int nh = 0;
    int *stk = (int *)xmalloc(sizeof(int) * 2 * (SW_MAXHITS * 2 + 2)), sp = 0;
    stk[sp++] = 0; stk[sp++] = Q;
    while (sp > 0) {
        int qe = stk[--sp], qs = stk[--sp];
        if (qe - qs < min_seg) continue;
        swhit best; best.score = 0; int have = 0;
        for (int ti = 0; ti < ntargets; ti++) for (int st = 0; st < 2; st++) {
            const char *qq = st == 0 ? contig + qs : rc + (Q - qe);
            int a0, a1, b0, b1, sc = sw_local(qq, qe - qs, targets[ti], tlens[ti], &a0, &a1, &b0, &b1);
            if (sc > best.score) {                                   /* strict: earlier (target, strand) wins ties */
                int off = st == 0 ? qs : Q - qe;
                best.score = sc; best.qs = off + a0; best.qe = off + a1; best.ts = b0; best.te = b1; best.strand = st; best.tidx = ti; have = 1;
            }
        }
        if (!have || best.score < min_score) continue;
        const char *qstr = best.strand == 0 ? contig : rc;
        best.nb = sw_blocks(qstr, targets[best.tidx], best.qs, best.qe, best.ts, best.te, best.score, best.bs, best.bq, best.bt);
        int fs = best.strand == 0 ? best.qs : Q - best.qe, fe = best.strand == 0 ? best.qe : Q - best.qs;   /* forward query interval */
        best.fq = fs;
        if (nh >= SW_MAXHITS) { fprintf(stderr, "bk_oracle: more than %d primary hits\n", SW_MAXHITS); abort(); }
        hits[nh++] = best;
        stk[sp++] = fe; stk[sp++] = qe;                              /* right remainder (processed after the left one) */
        stk[sp++] = qs; stk[sp++] = fs;
    }
    /* ---- step 5 (bk_oracle.h): secondary alignments.  BLAT prints EVERY alignment scoring >= -minScore, so a contig that
     * also matches somewhere else in a window (a duplicated flank, a repeat, a partner window) gets further PSL lines and
     * the caller counts them per query base (hit_freq, sv_caller.py:593-594; mean_cov :616,:631,:676; check_uniqueness
     * :430-432; check_previous_add :55-72).  Here: every diagonal of every (target, strand) is walked over the WHOLE query
     * with H = max(0, H + s); each positive excursion (from a reset to the next reset or the end of the diagonal) yields
     * one segment [start of the excursion, first position of its highest H); it is a secondary alignment iff that peak
     * is >= min_score and the segment does not overlap, on the same target / strand / diagonal, a hit of step 1 (that
     * would be the same alignment).  Each is reported as a one-block record of its own, after the chained records, ordered
     * by (score desc, target index asc, '+' first, query end asc, target end asc). */
    swsec *sec = NULL; int nsec = 0, seccap = 0;
    for (int ti = 0; ti < ntargets; ti++) for (int st = 0; st < 2; st++) {
        const char *qq = st == 0 ? contig : rc; const char *tt = targets[ti]; const int m = tlens[ti];
        for (int off = -(Q - 1); off < m; off++) {
            const int a0 = off < 0 ? -off : 0, a1 = Q < m - off ? Q : m - off;
            int h = 0, run = 0, ph = 0, pa = 0, prun = 0;
            for (int a = a0; a <= a1; a++) {
                int closed = a == a1;
                if (!closed) {
                    h += SW_EQ(qq[a], tt[a + off]) ? 1 : -2; run++;
                    if (h <= 0) closed = 1;
                    else if (h > ph) { ph = h; pa = a + 1; prun = run; }
                }
                if (closed) {
                    if (ph >= min_score) {
                        const int qs = pa - prun, qe = pa; int same = 0;
                        for (int x = 0; x < nh && !same; x++)
                            same = hits[x].tidx == ti && hits[x].strand == st && hits[x].ts - hits[x].qs == off && hits[x].qs < qe && qs < hits[x].qe;
                        if (!same) {
                            if (nsec == seccap) { seccap = seccap ? seccap * 2 : 64; sec = (swsec *)realloc(sec, (size_t)seccap * sizeof(swsec)); if (!sec || seccap > SW_MAXSEC) { fprintf(stderr, "bk_oracle: secondary hits overflow\n"); abort(); } }
                            swsec *e = &sec[nsec++]; e->qs = qs; e->qe = qe; e->ts = qs + off; e->te = qe + off; e->strand = st; e->tidx = ti; e->score = ph;
                        }
                    }
                    h = 0; run = 0; ph = 0; pa = 0; prun = 0;
                }
            }
        }
        g_sw_cells += (uint64_t)Q * (uint64_t)m;
    }
    /* ---- step 2b (bk_oracle.h): only what BLAT could have seeded.  Step-1 hits and secondary alignments without two matching
     * index tiles are dropped (the query bases of a dropped step-1 hit stay unaligned; island fill may still recover them
     * next to a chained anchor, as BLAT's extension does). */
    { int w = 0; for (int x = 0; x < nh; x++) if (sw_seedable(hits[x].strand == 0 ? contig : rc, targets[hits[x].tidx], hits[x].qs, hits[x].ts, hits[x].qe - hits[x].qs)) hits[w++] = hits[x]; nh = w; }
    { int w = 0; for (int x = 0; x < nsec; x++) if (sw_seedable(sec[x].strand == 0 ? contig : rc, targets[sec[x].tidx], sec[x].qs, sec[x].ts, sec[x].qe - sec[x].qs)) sec[w++] = sec[x]; nsec = w; }
    qsort(sec, (size_t)nsec, sizeof(swsec), cmp_sec);
    qsort(hits, (size_t)nh, sizeof(swhit), cmp_hit_fq);
    /* ---- step 6 (bk_oracle.h): placement of ambiguous hits.  When a step-1 hit has equal alternatives -- a secondary
     * alignment that covers its query interval and scores the same over it (a duplicated flank) -- step 1's tie-break
     * (smallest target end) is arbitrary; BLAT reports the chain with the smaller gaps.  The hits in forward query order each
     * choose among {the hit, its alternatives in step-5 order} so that the number of chain breaks between consecutive hits,
     * then the sum of |diagonal shifts| between chained neighbours, is smallest (ties: the earlier candidate).  A chosen
     * alternative becomes the hit (restricted to the hit's query interval); the hit it replaces is listed with the secondary
     * alignments instead, and the secondary it came from is dropped. */
    if (nsec > 0 && nh > 0) {
        typedef struct { int qs, qe, ts, te, strand, tidx, from; } pcand;
        pcand **cd = (pcand **)xcalloc((size_t)nh, sizeof(pcand *)); int *ncd = (int *)xcalloc((size_t)nh, sizeof(int));
        long long **cost = (long long **)xcalloc((size_t)nh, sizeof(long long *)); int **back = (int **)xcalloc((size_t)nh, sizeof(int *));
        for (int x = 0; x < nh; x++) {
            const swhit *hx = &hits[x];
            const int fs = hx->strand == 0 ? hx->qs : Q - hx->qe, fe = hx->strand == 0 ? hx->qe : Q - hx->qs;
            cd[x] = (pcand *)xmalloc((size_t)(nsec + 1) * sizeof(pcand));
            pcand self = { hx->qs, hx->qe, hx->ts, hx->te, hx->strand, hx->tidx, -1 }; cd[x][0] = self; ncd[x] = 1;
            for (int y = 0; y < nsec; y++) {
                const swsec *e = &sec[y];
                const int sfs = e->strand == 0 ? e->qs : Q - e->qe, sfe = e->strand == 0 ? e->qe : Q - e->qs;
                if (sfs > fs || sfe < fe) continue;
                const int cqs = e->strand == 0 ? fs : Q - fe, cqe = e->strand == 0 ? fe : Q - fs, dg = e->ts - e->qs;
                const char *qstr = e->strand == 0 ? contig : rc; const char *tstr = targets[e->tidx];
                int sc = 0; for (int z = cqs; z < cqe; z++) sc += SW_EQ(qstr[z], tstr[z + dg]) ? 1 : -2;
                if (sc != hx->score) continue;
                pcand c = { cqs, cqe, cqs + dg, cqe + dg, e->strand, e->tidx, y }; cd[x][ncd[x]++] = c;
            }
            cost[x] = (long long *)xcalloc((size_t)ncd[x], sizeof(long long)); back[x] = (int *)xcalloc((size_t)ncd[x], sizeof(int));
        }
        for (int x = 1; x < nh; x++) for (int b = 0; b < ncd[x]; b++) {
            long long bestc = -1; int besta = 0;
            for (int a = 0; a < ncd[x - 1]; a++) {
                const pcand *A = &cd[x - 1][a], *B = &cd[x][b]; long long t = 1ll << 24;            /* a chain break */
                if (A->tidx == B->tidx && A->strand == B->strand) {
                    const pcand *first = A->strand == 0 ? A : B, *second = A->strand == 0 ? B : A;
                    const int ov = first->te - second->ts;
                    if (!(ov > 0 && (2 * ov >= first->qe - first->qs || 2 * ov >= second->qe - second->qs)) && second->qs >= first->qe) {
                        t = (long long)(second->ts - second->qs) - (long long)(first->ts - first->qs); if (t < 0) t = -t;
                    }
                }
                if (bestc < 0 || cost[x - 1][a] + t < bestc) { bestc = cost[x - 1][a] + t; besta = a; }
            }
            cost[x][b] = bestc; back[x][b] = besta;
        }
        int pick = 0; for (int b = 1; b < ncd[nh - 1]; b++) if (cost[nh - 1][b] < cost[nh - 1][pick]) pick = b;
        unsigned char *gone = (unsigned char *)xcalloc((size_t)nsec + 1, 1); int nnew = 0; swsec *demoted = (swsec *)xmalloc((size_t)nh * sizeof(swsec));
        for (int x = nh - 1; x >= 0; x--) {
            const pcand *c = &cd[x][pick];
            if (c->from >= 0) {
                swhit *hx = &hits[x];
                swsec dm = { hx->qs, hx->qe, hx->ts, hx->te, hx->strand, hx->tidx, hx->score }; demoted[nnew++] = dm;
                gone[c->from] = 1;
                hx->qs = c->qs; hx->qe = c->qe; hx->ts = c->ts; hx->te = c->te; hx->strand = c->strand; hx->tidx = c->tidx;
                hx->bs[0] = c->qe - c->qs; hx->bq[0] = c->qs; hx->bt[0] = c->ts;                      /* fq (forward start) is unchanged */
            }
            pick = back[x][pick];
        }
        if (nnew > 0) {
            int w = 0; for (int y = 0; y < nsec; y++) if (!gone[y]) sec[w++] = sec[y];
            nsec = w;
            sec = (swsec *)realloc(sec, (size_t)(nsec + nnew + 1) * sizeof(swsec));
            for (int y = 0; y < nnew; y++) sec[nsec++] = demoted[y];
            qsort(sec, (size_t)nsec, sizeof(swsec), cmp_sec);
        }
        for (int x = 0; x < nh; x++) { free(cd[x]); free(cost[x]); free(back[x]); }
        free(cd); free(ncd); free(cost); free(back); free(gone); free(demoted);
    }
    /* chain + emit: hits consecutive in forward query order, same target and strand, collinear on the strand */
    /* A chain starts at the first hit not yet in a record and takes, going on in forward query order, every later free hit that
     * is collinear with it; a hit that is not (another target or strand, out of order on the target) is passed over and left for
     * a record of its own -- as BLAT's chaining does: a stretch copied from elsewhere in the middle of a contig does not
     * split the alignment of its two flanks. */
    int nrec = 0, overflow = 0;
    unsigned char *taken = (unsigned char *)xcalloc(SW_MAXHITS, 1);
    swhit *chain = (swhit *)xmalloc(sizeof(swhit) * (2 * SW_MAXHITS + 1));
    fblk *blocks = (fblk *)xmalloc(sizeof(fblk) * (BKO_MAX_BLOCKS * 2)), *anch = (fblk *)xmalloc(sizeof(fblk) * (2 * SW_MAXHITS + 1));
    for (int i = 0; i < nh; i++) {
        if (taken[i]) continue;
        int head = SW_MAXHITS, tail = SW_MAXHITS;     /* deque [head, tail) in strand order */
        chain[tail++] = hits[i]; taken[i] = 1;
        for (int j = i + 1; j < nh; j++) {
            if (taken[j]) continue;
            swhit h = hits[j];
            if (h.tidx != chain[head].tidx || h.strand != chain[head].strand) continue;
            /* '+': forward order == strand order; '-': the forward-later hit comes first on the strand */
            swhit *first = h.strand == 0 ? &chain[tail - 1] : &h, *second = h.strand == 0 ? &h : &chain[head];
            int ov = first->te - second->ts;
            if (ov > 0) {                                                 /* small target overlap (micro-homology): trim the later hit */
                if (2 * ov >= first->qe - first->qs || 2 * ov >= second->qe - second->qs || second->bs[0] <= ov) continue;
            }
            if (second->qs < first->qe) continue;
            if (ov > 0) { second->bs[0] -= ov; second->bq[0] += ov; second->bt[0] += ov; second->qs += ov; second->ts += ov; }
            if (h.strand == 0) chain[tail++] = h; else chain[--head] = h;
            taken[j] = 1;
        }
        {
            static bko_psl tmpc_; bko_psl *r = nrec < cap ? &out[nrec] : &tmpc_; memset(r, 0, sizeof(*r));      /* (beyond cap: built all the same, the filter needs it; the caller comes back with a larger cap) */
            const swhit *f = &chain[head], *l = &chain[tail - 1];
            const char *qstr = f->strand == 0 ? contig : rc; const char *tstr = targets[f->tidx];
            r->strand = f->strand == 0 ? '+' : '-'; r->q_size = Q; r->t_index = f->tidx; r->t_size = tlens[f->tidx];
            r->t_start = f->ts; r->t_end = l->te;
            const int sq = f->qs, eq = l->qe;                            /* strand coordinates */
            r->q_start = f->strand == 0 ? sq : Q - eq; r->q_end = f->strand == 0 ? eq : Q - sq;
            /* anchors in strand order, islands between them and beyond the ends filled (step 4) */
            int nbk = 0; const int bcap = BKO_MAX_BLOCKS * 2 - 1;      /* more than BKO_MAX_BLOCKS: the call fails (-2) */
            int na = 0;
            for (int c = head; c < tail; c++) { anch[na].qs = chain[c].qs; anch[na].qe = chain[c].qe; anch[na].ts = chain[c].ts; anch[na].te = chain[c].te; anch[na].score = chain[c].score; na++; }
            { const int gq = anch[0].qs; int tlo = anch[0].ts - gq - FILL_BAND; if (tlo < 0) tlo = 0;
              fill_gap(qstr, tstr, NULL, &anch[0], 0, anch[0].qs, tlo, anch[0].ts, blocks, &nbk, bcap); }
            for (int c = 0; c < na; c++) {
                if (nbk < bcap) blocks[nbk++] = anch[c];
                if (c + 1 < na) fill_gap(qstr, tstr, &anch[c], &anch[c + 1], anch[c].qe, anch[c + 1].qs, anch[c].te, anch[c + 1].ts, blocks, &nbk, bcap);
            }
            { const int gq = Q - anch[na - 1].qe; int thi = anch[na - 1].te + gq + FILL_BAND; if (thi > tlens[f->tidx]) thi = tlens[f->tidx];
              fill_gap(qstr, tstr, &anch[na - 1], NULL, anch[na - 1].qe, Q, anch[na - 1].te, thi, blocks, &nbk, bcap); }
            if (nbk > BKO_MAX_BLOCKS) { overflow = 1; nbk = BKO_MAX_BLOCKS; }
            r->t_start = blocks[0].ts; r->t_end = blocks[nbk - 1].te;
            { const int sq2 = blocks[0].qs, eq2 = blocks[nbk - 1].qe; r->q_start = f->strand == 0 ? sq2 : Q - eq2; r->q_end = f->strand == 0 ? eq2 : Q - sq2; }
            int nb = 0, pq = -1, pt = -1;
            for (int c = 0; c < nbk; c++) {
                r->score += blocks[c].score;
                const int bs = blocks[c].qe - blocks[c].qs, bq = blocks[c].qs, bt = blocks[c].ts;
                for (int z = 0; z < bs; z++) { if (SW_EQ(qstr[bq + z], tstr[bt + z])) { if (soft[f->tidx][bt + z]) r->rep_matches++; else r->matches++; } else r->mismatches++; }
                if (pq >= 0) { if (bq > pq) { r->q_num_insert++; r->q_base_insert += bq - pq; } if (bt > pt) { r->t_num_insert++; r->t_base_insert += bt - pt; } }
                r->block_sizes[nb] = bs; r->q_starts[nb] = bq; r->t_starts[nb] = bt; nb++;
                pq = bq + bs; pt = bt + bs;
            }
            r->block_count = nb;
            if (!bko_psl_passes(r, min_score)) continue;                   /* step 8: BLAT would not print it */
        }
        nrec++;
    }
    for (int x = 0; x < nsec; x++) {
        static bko_psl tmp_; bko_psl *r = nrec < cap ? &out[nrec] : &tmp_; memset(r, 0, sizeof(*r));
        const swsec *e = &sec[x]; const char *qstr = e->strand == 0 ? contig : rc; const char *tstr = targets[e->tidx];
        r->strand = e->strand == 0 ? '+' : '-'; r->q_size = Q; r->t_index = e->tidx; r->t_size = tlens[e->tidx];
        r->t_start = e->ts; r->t_end = e->te;
        r->q_start = e->strand == 0 ? e->qs : Q - e->qe; r->q_end = e->strand == 0 ? e->qe : Q - e->qs;
        for (int z = 0; z < e->qe - e->qs; z++) { if (SW_EQ(qstr[e->qs + z], tstr[e->ts + z])) { if (soft[e->tidx][e->ts + z]) r->rep_matches++; else r->matches++; } else r->mismatches++; }
        r->block_count = 1; r->block_sizes[0] = e->qe - e->qs; r->q_starts[0] = e->qs; r->t_starts[0] = e->ts; r->score = e->score;
        if (bko_psl_passes(r, min_score)) nrec++;                          /* step 8 */
    }
    free(sec);
    free(rc); free(hits); free(stk); free(taken); free(chain); free(blocks); free(anch);
    for (int ti = 0; ti < ntargets; ti++) { free(targets[ti]); free(soft[ti]); }
    free(targets); free(soft);
    return overflow ? -2 : nrec;
}
