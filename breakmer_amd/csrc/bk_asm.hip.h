// bk_asm.hip.h -- stage BK_STAGE_ASSEMBLE: sv_assembly.init_assembly (sv_assembly.py:30-63) as a
// device-side state machine, one 256-thread workgroup per target region.
//
// The reference's greedy assembler is order dependent (every accepted read changes the contig the
// next read is aligned to), so within a region the chain is serial; parallelism comes from
//   * many regions in flight (one workgroup each),
//   * the two overlap DPs of check_align (sv_assembly.py:451-452) on two wavefronts,
//   * the lanes of each wavefront along the DP (bk_nw.hip.h),
//   * the data-parallel bookkeeping (k-mer lists, count vectors, candidate sort) over 256 threads.
// Only ONE contig is live at a time: buff.contigs is a FIFO whose head is grown to completion, and
// every other entry is a fresh contig fully described by (seed k-mer, founding read).
//
// All threads execute the control flow redundantly on state kept in LDS (struct BkAsmShared);
// thread 0 commits scalar state between barriers.  Canonicalisations P1/P2/P4 of SURVEY.md 8c.
// This header is included TWICE by bk_api.hip, each time inside its own namespace, with BK_AT = 512 (8 wavefronts, 8
// look-ahead slots, 2 workgroups per CU: shortest time for one batch) and BK_AT = 256 (4 wavefronts, 4 slots, 4 workgroups per
// CU: highest throughput when batches are in flight); BK_ASM_KERNEL names the kernel.  bk_common.h / bk_nw.hip.h are
// included by bk_api.hip before, outside the namespaces.

#ifndef BK_AT
#error "define BK_AT (threads per assembler workgroup: 512 or 256) and BK_ASM_KERNEL before including bk_asm.hip.h"
#endif
#undef BK_SPEC
#undef BK_SPEC_WIDE
#undef BK_WAVES
#define BK_WAVES (BK_AT / 64)
#ifdef BK_PAIR                      // throughput build (bk_api.hip: the 256-thread kernel): a wavefront aligns TWO reads of a round (bk_nw_pair) ...
#define BK_SPEC (2 * BK_WAVES)      // ... so a round has twice as many look-ahead slots as wavefronts
#else
#define BK_SPEC (BK_AT / 64)        // look-ahead slots, both overlap DPs of a slot on one wavefront (bk_nw_dual);
#endif
#define BK_SPEC_WIDE (BK_AT / 128)  // contigs beyond BK_NW_DUAL_COLS or few reads in a round: half as many slots x 2 wavefronts (one DP each)


#define BK_ACC_MAX 32
enum { BK_ORD_FOR = 0, BK_ORD_REV = 1, BK_ORD_MID = 2 };
enum { BK_DEC_NONE = 0, BK_DEC_SAME = 1, BK_DEC_SUPER = 2, BK_DEC_SUB = 3, BK_DEC_POST = 4, BK_DEC_PRE = 5 };

struct BkAsmShared {
    int status;
    // live contig
    int cbase, clen;             // sequence deque in cseq[]
    int nbase, nlen, cbuf;       // count-vector deque (global), active buffer
    int serial, setup, founder, founder_added, in_fifo;
    int nk;                      // len(contig.kmers)
    int nr;                      // len(contig.reads)
    int nalt;
    // current read
    int ru, rlen, rn, rindel;
    // FIFO buff.contigs
    int phead, ptail;
    // misc
    int serial_ctr, stamp_ctr, head, nused;
    int ncand, dec, dstart, dend, tmp0, tmp1, tmp2;
    int n_contigs;
    unsigned long long cells, calls;
    // look-ahead slots (see bk_run_candidates): read q+s aligned against the PREDICTED contig after reads q..q+s-1
    struct Slot { int u, rl, rn, rindel, pos, pb, plen, kind, amt, hasn, vt, rank, fu, first, dec, ds, de; BkNwResult v1, v2; } slot[BK_SPEC];
    int nb, pc, last_dec, dual;
    // the score sweep (bk_nw.hip.h: bk_nw_score_c) runs first while few of its reads need the full sweep after all: reads swept /
    // reads that had to be swept again, in a window that is halved at 64; fast = the plan of the current round uses it
    int dp_n, dp_redo, fast;
    int dp_tot, dp_rtot;             // the same counts over the whole region (reported: BkRegionWork.dp_sweeps / dp_redos)
    // look-ahead across k-mer visits of grow (bk_run_candidates): slots [plan_r, nb) hold reads of LATER visits, already aligned
    // against the predicted contig; plan_upto = last visit (index into the snapshot) whose candidate list the plan knows
    int ncur, plan_r, plan_upto, plan_ok, hit;
    int plan_kind;               // 0: slots of later visits of this grow snapshot; 1: slots of the first round of later SEEDS (vt = -2 - rank)
    int la_planned, la_adopted;  // slots planned for later visits / retired from there, in the current window of 64 planned
    int la_pause, la_backoff;
    int n_rej, n_acc;            // recent reads rejected / accepted by check_align: with most rejected, the prediction is "nothing changes"    // rounds the look-ahead stays off after a window in which most of its slots were wasted (doubles)
    int la_n[BK_WAVES], la_t[BK_WAVES], la_rank[BK_WAVES], la_pc[BK_WAVES];      // one look-ahead list per wavefront
    int qslot;                   // position in the region queue this workgroup is working on
    int kscan;                   // contig k-mer list: the entries before this index are all in checked_kmers (grow snapshots only look at what came after)
    // split regions (bk_comp.hip.h): the component of the current seed, the components its contigs have taken in (same unit or
    // claimed), what bk_kmers_ordered found beyond them
    int seed_rank, emit_seq, t0;
    uint32_t ccomp, acc_n, foreign, foreign_root, dirty;
    uint32_t scan[10];
#ifdef BK_PHASE_STAMPS
    unsigned long long acc[24], last; int ctx;
#endif
};

// The context lives in LDS (not in registers: ~60 uniform pointers would spill the SGPR file) at the
// start of the dynamic LDS block, followed by BkAsmShared and the byte/word buffers (offsets below).
struct BkAsmCtx {
    // LDS offsets in units of 16 bytes (BK_O_*): contig deque (2*MAXC bytes) and the reads of the slots; DP tile boundary; candidate sort
    // keys (max_cand; afterwards int scratch[2*max_cand]); sorted candidate reads (unique-read index)
    uint16_t o_cseq16, o_rseq16, o_bound16, o_cand16, o_candu16;
    uint16_t MAXR, MAXCAND;
    // region data
    BkRegionWork *wk;
    uint8_t *out; unsigned long long *out_top; uint64_t out_cap;
    unsigned long long *n_clist, *clist; uint64_t clist_cap; int region;
    int rc_thresh; uint32_t read_words, max_len;
    const uint32_t *reads; const uint16_t *rlen;
    const uint32_t *nlist; uint32_t n_nlist;   // N calls of the region's reads (read index << 10 | position), sorted
    uint32_t *urep, *unr; uint8_t *ufl; int32_t *ubuf, *ureads, *ufound, *uminpos;
    const uint32_t *ulen;                      // length of unique read u (k-mer stage)
    const uint32_t *tslot; const uint64_t *klo, *khi; const uint32_t *kcnt; uint8_t *kstate; int32_t *kstamp;
    const uint32_t *poff, *post;
    uint32_t U, M, tmask;
    // per-region scratch (arena)
    int32_t *cnt;                // 2 buffers x {io, ot} x 2*MAXC
    uint32_t *klist, *nklist;    // contig.kmers / refresh snapshot (rank | rev << 31)
    uint32_t *pend;              // FIFO: 2 words per entry (rank, u)
    uint32_t *altl, *readl, *usedl;
    int MAXC, k, flags;
    int32_t *myseeds; int n_my;      // split regions, once the components are labelled (own): the seed ranks of this unit's components, ascending (M2 + 64 entries of scratch); else every rank 0 .. n_my-1 is a candidate
    // split regions: this workgroup is unit `unit` of `split`; it owns the components whose info word reads `want`
    uint8_t split, unit, pass, own; uint32_t want, M2;      // own: the components are labelled and this unit only touches its own (off while unit 0 runs the serial prefix)
    const uint32_t *kroot; uint32_t *cinfo; unsigned long long *cidx_key; uint32_t *pairs, *acc_root;
};

// Functions off the DP round trip (emit, alt reads, find_reads, contig k-mer lists) are kept out of line: inlined into
// the state machine they pushed it to 65 spilled VGPRs (176 B of scratch per lane); out of line it has 25, all in the
// candidate loop with its look-ahead (that loop out of line too: none, and 4 % fewer regions/s -- measured, not kept).
#ifndef BK_INLINE_COLD
#define BK_COLD __device__ __noinline__
#else
#define BK_COLD __device__ inline
#endif
#define BK_TID ((int)threadIdx.x)
#undef BK_SRC_ID
#define BK_SRC_ID 5      // barrier sites of this file (bk_common.h: BK_SYNC; the two instances of this header share the site ids)
#ifdef BK_PHASE_STAMPS      // diagnostic build only: where does a region's time go (s_memrealtime, 100 MHz)
#define BK_ACC(i) do { if (BK_TID == 0) { unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); S_->acc[i] += now_ - S_->last; S_->last = now_; } } while (0)
#else
#define BK_ACC(i) do { } while (0)
#endif
#ifdef BK_PHASE_STAMPS
#define BK_CTX(v) do { BK_SYNC(); if (BK_TID == 0) S_->ctx = (v); BK_SYNC(); } while (0)
#else
#define BK_CTX(v) do { } while (0)
#endif

extern __shared__ __attribute__((aligned(16))) uint8_t bk_lds[];
#define BK_O_CSEQ ((int)C_.o_cseq16 << 4)
#define BK_O_RSEQ ((int)C_.o_rseq16 << 4)
#define BK_O_BOUND ((int)C_.o_bound16 << 4)
#define BK_O_CAND ((int)C_.o_cand16 << 4)
#define BK_O_CANDU ((int)C_.o_candu16 << 4)
#define BK_SH_OFF ((int)((sizeof(BkAsmCtx) + 15) / 16 * 16))
#define BK_BUF_OFF ((int)(BK_SH_OFF + (sizeof(BkAsmShared) + 15) / 16 * 16))
#define C_ (*(BkAsmCtx *)bk_lds)
#define S_ ((BkAsmShared *)(bk_lds + BK_SH_OFF))
#define L_CSEQ (bk_lds + BK_O_CSEQ)
#define L_RSEQ (bk_lds + BK_O_BOUND)                                  // the generic read buffer (bk_load_read): not a slot's (slots outlive a visit); shares the DP scratch
#define L_RSEQ_S(s) (bk_lds + BK_O_RSEQ + (s) * (C_.MAXR + 16))
#define BK_SEEDBUF(g) (2 * C_.MAXC + (g) * 3 * (C_.MAXR + 16))         // offset (from L_CSEQ) of the strip of look-ahead seed g: in the DP scratch behind the deque
#define L_BOUND ((int *)(bk_lds + BK_O_BOUND))
#define L_BOUND_W(w) ((int *)(bk_lds + BK_O_BOUND) + (w) * 2 * (C_.MAXR + 2))
#define L_CAND ((unsigned long long *)(bk_lds + BK_O_CAND))
#define L_CANDU ((uint32_t *)(bk_lds + BK_O_CANDU))

__device__ inline int32_t *bk_cnt_io(int buf) { return C_.cnt + (size_t)buf * 4 * C_.MAXC; }
__device__ inline int32_t *bk_cnt_ot(int buf) { return C_.cnt + (size_t)buf * 4 * C_.MAXC + 2 * C_.MAXC; }

// A region gives up (a working cap overflowed: the library runs it again under larger ones).  S->status steers every loop of the
// state machine (`if (S->status) return;`), so the write is bracketed by barriers HERE: every wavefront has finished the reads that
// came before, and none reads the word again before the second barrier.  Every call site is uniform.  (Round 5: bk_retire called the
// bare write of rounds 1-4 with no barrier behind the loop-top test of the retire loop -- a late wavefront saw the status, left
// bk_run_candidates and stood in bk_finalize's barrier while its workgroup stood in bk_retire's.  Only on the contig-overflow path;
// found by running the whole GPU suite through the barrier-check build, tests/test_hip_gpu.py with BK_TEST_VARIANT=checkjit.)
__device__ inline void bk_fail(int st) { BK_SYNC(); if (BK_TID == 0 && S_->status == 0) S_->status = st; BK_SYNC(); }

// k-mer key of LDS bytes s[0..k); false when the window holds an N (code 4): no such k-mer exists (Jellyfish skips them)
// Four bases per step: aligned LDS words funnelled to the byte offset of s (v_alignbyte), the four 2-bit codes of a word
// gathered with shifts -- ~12 instructions per four bases; base by base with a 128-bit rolling key it was ~10 per BASE, and
// the k-mer searches of the planning code (bk_find_kmer_wave) were a tenth of all instructions of a clean region.
__device__ inline bool bk_bytes_kmer(const uint8_t *s, int k, BkKey &key)
{
    const uint32_t off = (uint32_t)(s - bk_lds) & 3u;
    const uint32_t *wp = (const uint32_t *)(s - off);
    const int G = (k + 3) >> 2;
    uint64_t hi = 0, lo = 0; uint32_t acc = 0, any = 0, w0 = wp[0];
    for (int g = 0; g < G; g++) {
        const uint32_t w1 = wp[g + 1];
        uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, off);              // bytes s[4g .. 4g+3], the first base in the low byte
        w0 = w1;
        if (g == G - 1 && (k & 3)) x &= (1u << (8 * (k & 3))) - 1u;        // bases beyond k
        any |= x;
        x &= 0x03030303u;
        acc = (acc << 8) | (((x << 6) | (x >> 4) | (x >> 14) | (x >> 24)) & 0xFFu);      // first base most significant, as key_push
        if ((g & 3) == 3) { hi = (hi << 32) | (lo >> 32); lo = (lo << 32) | acc; acc = 0; }
    }
    if (G & 3) { const int sh = 8 * (G & 3); hi = (hi << sh) | (lo >> (64 - sh)); lo = (lo << sh) | acc; }
    const int drop = 2 * (4 * G - k);                                      // the (zeroed) codes of the last group beyond k
    if (drop) { lo = (lo >> drop) | (hi << (64 - drop)); hi >>= drop; }
    key.hi = hi; key.lo = lo;
    return (any & (0x01010101u * BK_CODE_N)) == 0;
}
// write code 4 over the N calls of read i (region index) that fall into bases [from, from + count) of the copy at dst
__device__ inline void bk_patch_n(uint32_t i, uint8_t *dst, int from, int count)
{
    uint32_t lo, hi; bk_nlist_range(C_.nlist, C_.n_nlist, i, lo, hi);
    for (uint32_t e = lo; e < hi; e++) { const int q = (int)(C_.nlist[e] & 1023u) - from; if (q >= 0 && q < count) dst[q] = BK_CODE_N; }
}

#ifdef BK_CHECK      // diagnostic build: indices that must hold, recorded (first one wins) in wk->stamps[10..13] instead of being followed
#define BK_CHK(cond, code, val) ((cond) ? true : (bk_chk_fail((code), (unsigned long long)(val)), false))
__device__ __noinline__ void bk_chk_fail(int code, unsigned long long val)
{
    unsigned long long *st = (unsigned long long *)C_.wk->stamps;
    if (atomicCAS(&st[10], 0ull, (unsigned long long)code) == 0ull) { st[11] = val; st[12] = ((unsigned long long)C_.unit << 32) | C_.pass; st[13] = ((unsigned long long)S_->serial << 32) | (uint32_t)S_->seed_rank; }
}
#else
#define BK_CHK(cond, code, val) true
#endif
// split regions: does k-mer `rk` belong to a component this unit owns in this pass (always true for an unsplit region)
__device__ inline bool bk_mine(int rk)
{
    if (!C_.own) return true;
    const uint32_t root = C_.kroot[rk];
    if (root == BK_EMPTY32) return false;
    return (__hip_atomic_load(&C_.cinfo[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (0xFFFFu | BK_CI_ABORT)) == C_.want;
}
#define BK_UFL0 ((uint8_t *)C_.pairs + bk_align_up((uint64_t)C_.wk->pairs_cap * 12, 256))      // the read flags as they were when the graph was labelled (bk_comp.hip.h: bk_ufl0)
__device__ inline int bk_seed_at(int i) { if (C_.own && !BK_CHK((uint32_t)i < C_.M2 + 64u, 8, i)) return -1; return C_.own ? C_.myseeds[i] : i; }      // the i-th seed candidate of this unit (rank; -1: none)
// the components of the running seed iteration: the seed's own, those taken in (entries of acc_root), and -- bit 31 set -- those of
// OTHER units it met: their k-mers are left out of the contig's k-mer list and the iteration runs on to its normal end (its component
// is given up and runs again after the merge; nothing of the other unit's state is touched).  bk_acc_has: known either way;
// bk_acc_mine: held by this iteration.
__device__ inline bool bk_acc_has(uint32_t root) { BkAsmShared *S = S_; if (root == S->ccomp) return true; for (uint32_t i = 0; i < S->acc_n; i++) if ((C_.acc_root[i] & 0x7FFFFFFFu) == root) return true; return false; }
__device__ inline bool bk_acc_mine(uint32_t root) { BkAsmShared *S = S_; if (root == S->ccomp) return true; for (uint32_t i = 0; i < S->acc_n; i++) if (C_.acc_root[i] == root) return true; return false; }

// sample k-mer table lookup -> rank or -1 (any state)
__device__ inline int bk_lookup(const BkKey &key)
{
    if (C_.M == 0) return -1;
    uint32_t s = key_hash(key) & C_.tmask;
    for (;;) {
        const uint32_t rk = C_.tslot[s];                   // rank of the k-mer in this slot (the k-mer stage stores it there)
        if (rk == BK_EMPTY32) return -1;
        if (rk != BK_EMPTY32 - 1 && C_.klo[rk] == key.lo && C_.khi[rk] == key.hi) return (int)rk;      // EMPTY-1: tombstone
        s = (s + 1) & C_.tmask;
    }
}

// ... and its state, asked for together with the key words: every table lookup of a noisy region's chain is a string of
// dependent global round trips (slot -> key -> state), and this is one less
__device__ inline int bk_lookup_state(const BkKey &key, uint32_t &state)
{
    state = 0;
    if (C_.M == 0) return -1;
    uint32_t s = key_hash(key) & C_.tmask;
    for (;;) {
        const uint32_t rk = C_.tslot[s];
        if (rk == BK_EMPTY32) return -1;
        if (rk != BK_EMPTY32 - 1) {                         // EMPTY-1: tombstone
            const unsigned long long lo = C_.klo[rk], hi = C_.khi[rk]; const uint32_t st = C_.kstate[rk];
            if (lo == key.lo && hi == key.hi) { state = st; return (int)rk; }
        }
        s = (s + 1) & C_.tmask;
    }
}

__device__ inline uint32_t bk_scan256(uint32_t v, uint32_t *scr, uint32_t *total)
{
    const int lane = BK_TID & 63, wv = BK_TID >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    BK_SYNC();
    if (lane == 63) scr[wv] = inc;
    BK_SYNC();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < BK_AT / 64; i++) { uint32_t t = scr[i]; if (i < wv) base += t; tot += t; }
    *total = tot;
    BK_SYNC();
    return base + inc - v;
}
__device__ inline int bk_max256(int v, uint32_t *scr)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    BK_SYNC();
    if ((BK_TID & 63) == 0) scr[BK_TID >> 6] = (uint32_t)v;
    BK_SYNC();
    int r = (int)scr[0];
    for (int i = 1; i < BK_AT / 64; i++) r = max(r, (int)scr[i]);
    BK_SYNC();
    return r;
}

// ---- unpack unique read u into rseq (LDS bytes) ----------------------------------------------------
__device__ inline void bk_load_read(int u)
{
    if (!BK_CHK((uint32_t)u < C_.U, 14, u)) u = 0;
    const uint32_t i = C_.urep[u];
    const int len = C_.rlen[i];
    const uint32_t *w = C_.reads + (uint64_t)i * C_.read_words;
    BK_SYNC();
    for (int t = BK_TID; t < len; t += BK_AT) L_RSEQ[t] = (uint8_t)seq_base(w, t);
    if (BK_TID == 0) { S_->ru = u; S_->rlen = len; S_->rn = (int)C_.unr[u]; S_->rindel = (C_.ufl[u] & BK_R_INDEL) ? 1 : 0; }
    BK_SYNC();
    if (C_.n_nlist && (C_.ufl[u] & BK_R_HASN)) { if (BK_TID == 0) bk_patch_n(i, L_RSEQ, 0, len); BK_SYNC(); }
}

// ---- assembly_counts (sv_assembly.py:160-221) --------------------------------------------------------
__device__ inline void bk_set_counts(int start, int end, int nreads, int indel)   // :195-199 with python slice clipping
{
    BkAsmShared *S = S_;
    const int e = min(end, S->nlen);
    int32_t *v = (indel ? bk_cnt_io(S->cbuf) : bk_cnt_ot(S->cbuf)) + S->nbase;
    for (int t = start + BK_TID; t < e; t += BK_AT) v[t] += nreads;
    BK_SYNC();
}
__device__ inline void bk_extend_counts(int l, int nreads, int indel, bool post)   // :201-221
{
    BkAsmShared *S = S_;
    int32_t *io = bk_cnt_io(S->cbuf), *ot = bk_cnt_ot(S->cbuf);
    const int at = post ? S->nbase + S->nlen : S->nbase - l;
    if (at < 0 || at + l > 2 * C_.MAXC) { bk_fail(BK_ST_CONTIG); return; }
    for (int t = BK_TID; t < l; t += BK_AT) { io[at + t] = indel ? nreads : 0; ot[at + t] = indel ? 0 : nreads; }
    BK_SYNC();
    if (BK_TID == 0) { if (!post) S->nbase -= l; S->nlen += l; }
    BK_SYNC();
}
// set_superseq :181-193 incl. the zip()-truncating slice assignment (Q8); new vectors go to the other buffer
__device__ inline void bk_counts_superseq(int rlen, int nreads, int indel, int start, int end)
{
    BkAsmShared *S = S_;
    const int s = min(start, rlen), e = max(min(end, rlen), s);
    const int seg = e - s, z = min(seg, S->nlen), nl = rlen - seg + z;
    const int nb = C_.MAXC - nl, ob = S->nbase, obuf = S->cbuf, nbuf = obuf ^ 1;
    const int32_t *oio = bk_cnt_io(obuf) + ob, *oot = bk_cnt_ot(obuf) + ob;
    int32_t *nio = bk_cnt_io(nbuf) + nb, *not_ = bk_cnt_ot(nbuf) + nb;
    const int bi = indel ? nreads : 0, bo = indel ? 0 : nreads;
    for (int w = BK_TID; w < nl; w += BK_AT) {
        int a = bi, b = bo;
        if (w >= s && w < s + z) { a += oio[w - s]; b += oot[w - s]; }
        nio[w] = a; not_[w] = b;
    }
    BK_SYNC();
    if (BK_TID == 0) { S->cbuf = nbuf; S->nbase = nb; S->nlen = nl; }
    BK_SYNC();
}
__device__ inline int bk_total_reads()                                                   // :178-179
{
    BkAsmShared *S = S_;
    const int32_t *io = bk_cnt_io(S->cbuf) + S->nbase, *ot = bk_cnt_ot(S->cbuf) + S->nbase;
    int a = -0x7FFFFFFF, b = -0x7FFFFFFF;
    for (int t = BK_TID; t < S->nlen; t += BK_AT) { a = max(a, io[t]); b = max(b, ot[t]); }
    a = bk_max256(a, S->scan); b = bk_max256(b, S->scan);
    return a + b;
}

// ---- split regions: the contig of the running seed iteration holds a k-mer of component `root`, which is neither the seed's
//      nor one it has taken in.  Thread 0 decides (bk_comp.hip.h):
//   no unit (no seeds)              -> claimed, taken in;
//   this unit's                     -> taken in (the unit walks its seeds in rank order: the other component stands where the
//                                      serial run would have it); noted: the two are one component from now on;
//   anything else                   -> the seed's component is given up, the pair noted for the repair pass.
__device__ inline void bk_note_pair(uint32_t a, uint32_t b, uint32_t kind)
{
    const uint32_t at = atomicAdd(&C_.wk->n_pairs, 1u);
    if (at < C_.wk->pairs_cap) { C_.pairs[3 * at] = a; C_.pairs[3 * at + 1] = b; C_.pairs[3 * at + 2] = kind; }
    if (kind) atomicAdd(&C_.wk->n_conf, 1u);
}
__device__ inline void bk_meet(uint32_t root)
{
    BkAsmShared *S = S_;
    if (!BK_CHK(root < C_.U, 1, root)) { S->status = BK_ST_CONFLICT; S->foreign = 0; S->foreign_root = BK_EMPTY32; return; }
    uint32_t ci = __hip_atomic_load(&C_.cinfo[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((ci & BK_CI_UNIT) == BK_CI_NOUNIT) { const uint32_t old = atomicCAS(&C_.cinfo[root], ci, C_.want); ci = old == ci ? C_.want : old; }
    if ((ci & (0xFFFFu | BK_CI_ABORT)) == C_.want && S->acc_n < BK_ACC_MAX) {
        bk_note_pair(S->ccomp, root, 0u);
        C_.acc_root[S->acc_n] = root; __threadfence_block(); S->acc_n++;
    } else {
        bk_note_pair(S->ccomp, root, 1u);
        if (S->acc_n < BK_ACC_MAX) { C_.acc_root[S->acc_n] = root | 0x80000000u; __threadfence_block(); S->acc_n++; S->dirty = 1; }      // met, not taken in: the iteration goes on without its k-mers
        else S->status = BK_ST_CONFLICT;                                    // (no room to remember it: the iteration is left here, as in the first version)
    }
    S->foreign = 0; S->foreign_root = BK_EMPTY32;
}

BK_COLD void bk_build_myseeds(int fresh);
// order MID replaces contig.kmers (set_kmers :548-550), FOR/REV extend it (:525-527, :543-545).
// P1: m = L // 2 ; Q1: positions range(0, L-k).
BK_COLD void bk_kmers_ordered(int s0, int L, int order)
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const int k = C_.k, np = L - k;                      // number of positions
    int *tmp = (int *)L_CAND;                           // rank per position (or -1)
    const int m = L / 2;
    if (np > 2 * C_.MAXCAND) { bk_fail(BK_ST_KLIST); return; }
    if (np <= 0 && order != BK_ORD_MID) return;         // a one-base extension has no new k-mer (Q1: range(0, L-k) of a window of k bases): nothing to append, six barriers saved
    if (S->status) return;                              // (uniform) a conflict is being unwound
    // Split regions: a k-mer of a component this unit does not hold counts as a meeting WHATEVER its state says -- the other unit
    // may be ahead of this one in seed order, and what it has removed by now may still have been there at this seed's turn in
    // the serial order.  (Homopolymer k-mers are in no component: kroot = BK_EMPTY32.)  Bit 30 of tmp[x]: removed.
    for (int x = BK_TID; x < np; x += BK_AT) {
        BkKey key; uint32_t st = 0; int rk = bk_bytes_kmer(L_CSEQ + s0 + x, k, key) ? bk_lookup_state(key, st) : -1;
        if (C_.own && rk >= 0) { const uint32_t root = C_.kroot[rk]; if (root != BK_EMPTY32 && root != S->ccomp && !bk_acc_has(root)) { S->foreign = 1; atomicMin(&S->foreign_root, root); } }
        if (rk >= 0 && st == BK_K_REMOVED) rk = C_.own ? (rk | 0x40000000) : -1;                     // not in akmers.smers_set
        tmp[x] = rk;
    }
    BK_SYNC();
    if (C_.own) {
        // The contig holds k-mers of components other than the seed's (a k-mer across the seam of two read pieces).  One new
        // component per turn, smallest root first: same unit -> taken in; no unit (it has no seeds) -> claimed, taken in;
        // another unit's -> the current component is given up (bk_comp.hip.h).  Rare: thread 0 decides, everyone re-checks.
        // S->foreign steers the loop and bk_meet (thread 0) resets it: every wavefront reads it, THEN a barrier, then the reset
        // (round 5: without that barrier a late wavefront read the reset word, skipped the loop and its barriers -- the wild
        // indices and hangs of the split path under load).
        bool more = S->foreign != 0;
        while (more) {
            BK_SYNC();
            if (BK_TID == 0) bk_meet(S->foreign_root);
            BK_SYNC();
            if (S->status) return;
            for (int x = BK_TID; x < np; x += BK_AT) { const int rk = tmp[x]; if (rk >= 0) { const uint32_t root = C_.kroot[rk & 0x3FFFFFFF]; if (root != BK_EMPTY32 && root != S->ccomp && !bk_acc_has(root)) { S->foreign = 1; atomicMin(&S->foreign_root, root); } } }
            BK_SYNC();
            more = S->foreign != 0;
        }
    }
    if (C_.own) {
        const bool dirty = S->dirty != 0;                    // (uniform) the iteration has met a component of another unit
        for (int x = BK_TID; x < np; x += BK_AT) {
            const int rk = tmp[x];
            if (rk < 0) continue;
            if (rk & 0x40000000) { tmp[x] = -1; continue; }
            if (dirty) { const uint32_t root = C_.kroot[rk]; if (root != BK_EMPTY32 && !bk_acc_mine(root)) tmp[x] = -1; }
        }
        BK_SYNC();
    }
    const int chunk = (max(np, 0) + BK_AT - 1) / BK_AT, b = BK_TID * chunk, e = min(np, b + chunk);
    uint32_t cnt = 0, T;
    for (int x = b; x < e; x++) cnt += tmp[x] >= 0;
    uint32_t pre = bk_scan256(cnt, S->scan, &T);
    // pre_m = number of valid positions < m
    if (order == BK_ORD_MID) {
        if (BK_TID == 0) S->tmp0 = (int)T;              // default when m >= np
        BK_SYNC();
        if (m >= b && m < e) { uint32_t q = pre; for (int x = b; x < m; x++) q += tmp[x] >= 0; S->tmp0 = (int)q; }
        BK_SYNC();
    }
    const int pre_m = S->tmp0;
    const int base = (order == BK_ORD_MID) ? 0 : S->nk;
    if (base + (int)T > (2 * C_.MAXC)) { bk_fail(BK_ST_KLIST); return; }
    uint32_t q = pre;
    for (int x = b; x < e; x++) {
        int rk = tmp[x];
        if (rk < 0) continue;
        int idx; uint32_t rev;
        if (order == BK_ORD_FOR) { idx = (int)q; rev = 1u; }                            // get_mer_reads :610-611: 'for' -> 'rev'
        else if (order == BK_ORD_REV) { idx = (int)T - 1 - (int)q; rev = 0u; }
        else { if (x >= m) { idx = (int)q - pre_m; rev = 1u; } else { idx = (int)T - 1 - (int)q; rev = 0u; } }   // :142 sorted by (x<m, |x-m|); :607-609
        C_.klist[base + idx] = (uint32_t)rk | (rev << 31);
        q++;
    }
    BK_SYNC();
    if (BK_TID == 0) { S->nk = base + (int)T; if (order == BK_ORD_MID) { S->setup = 1; S->kscan = 0; } }
    BK_SYNC();
    BK_ACC(5);
}

// ---- find_reads (sv_assembly.py:111-122) from the posting list of k-mer `rank` -------------------------
// key (pos, -len) / (-pos, -len); stable sort ties keep fq_recs order = unique index u.
BK_COLD void bk_find_reads(int rank, bool rev, bool filter)
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const uint32_t b = C_.poff[rank], e = C_.poff[rank + 1];
    if (e - b <= 64u) {
        // Short posting list (the rule for sequencing-error k-mers): one wavefront does everything in registers --
        // first occurrence per read, filters, order -- with two global round trips and a single workgroup barrier.
        if ((BK_TID >> 6) == 0) {
            const int lane = BK_TID, np = (int)(e - b);
            const bool have = lane < np;
            const uint32_t en = have ? C_.post[b + lane] : 0u;
            const uint32_t u = en >> 10; const int pos = (int)(en & 1023u);
            uint32_t fl = 0; int bufst = 0; uint32_t len = 0;
            if (have) { fl = C_.ufl[u]; bufst = C_.ubuf[u]; len = C_.ulen[u]; }
            bool drop = false;                                                     // a smaller position of the same read exists
            for (int j = 0; j < np; j++) {
                const uint32_t oen = (uint32_t)__builtin_amdgcn_readlane((int)en, j);
                drop = drop || ((oen >> 10) == u && (int)(oen & 1023u) < pos);
            }
            const bool valid = have && !drop && !(fl & BK_R_DELETED) && !(filter && bufst == S->serial);
            const unsigned long long pk = rev ? (unsigned long long)(0xFFFF - pos) : (unsigned long long)pos;
            const unsigned long long key = valid ? ((pk << 40) | ((0xFFFFull - len) << 24) | u) : ~0ull;
            int idx = 0;                                                           // rank among the valid keys (unique: u is)
            for (int j = 0; j < np; j++) {
                const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                idx += kj < key;
            }
            const unsigned long long vm = __ballot(valid);
            if (valid) L_CANDU[idx] = u | ((uint32_t)pos << 22);
            if (lane == 0) S->ncand = __popcll(vm);
        }
        BK_SYNC();
        // every thread reads ncand BEFORE thread 0 may reset it (the extra barrier is only taken on the failing path; the
        // condition is uniform, so the workgroup's barriers stay aligned -- a late wavefront must not see the reset value)
        const int nc_ = S->ncand;
        if (nc_ > C_.MAXCAND) { bk_fail(BK_ST_CAND); if (BK_TID == 0) S->ncand = 0; BK_SYNC(); }      // (bk_fail starts with a barrier: every thread has read ncand)
        BK_ACC(4);
        return;
    }
    if (BK_TID == 0) S->ncand = 0;
    BK_SYNC();
    // first occurrence of the k-mer in each read (re.search): min pos per read
    for (uint32_t i = b + BK_TID; i < e; i += BK_AT) { uint32_t en = C_.post[i]; atomicMin(&C_.uminpos[en >> 10], (int)(en & 1023u)); }
    BK_SYNC();
    for (uint32_t i = b + BK_TID; i < e; i += BK_AT) {
        uint32_t en = C_.post[i], u = en >> 10; int pos = (int)(en & 1023u);
        if (C_.uminpos[u] != pos) continue;
        if (C_.ufl[u] & BK_R_DELETED) continue;                                  // deleted from fq_recs (rb.clean :390)
        if (filter && C_.ubuf[u] == S->serial) continue;                         // ids - self.buffer (:115-116)
        int idx = atomicAdd(&S->ncand, 1);
        if (idx < C_.MAXCAND) {
            unsigned long long len = C_.ulen[u];
            unsigned long long pk = rev ? (unsigned long long)(0xFFFF - pos) : (unsigned long long)pos;
            L_CAND[idx] = (pk << 40) | ((0xFFFFull - len) << 24) | u;
        }
    }
    BK_SYNC();
    for (uint32_t i = b + BK_TID; i < e; i += BK_AT) C_.uminpos[C_.post[i] >> 10] = 0x7FFFFFFF;
    const int n = S->ncand;
    if (n > C_.MAXCAND) { bk_fail(BK_ST_CAND); if (BK_TID == 0) S->ncand = 0; BK_SYNC(); return; }      // (bk_fail starts with a barrier: every thread has read ncand)
    int npad = 1; while (npad < n) npad <<= 1;
    for (int i = n + BK_TID; i < npad; i += BK_AT) L_CAND[i] = ~0ull;
    BK_SYNC();
    for (int sz = 2; sz <= npad; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = BK_TID; i < npad / 2; i += BK_AT) {
                int lo = (i / st) * (st * 2) + (i % st), hi = lo + st;
                bool up = ((lo & sz) == 0);
                unsigned long long a = L_CAND[lo], bb = L_CAND[hi];
                if ((a > bb) == up) { L_CAND[lo] = bb; L_CAND[hi] = a; }
            }
            BK_SYNC();
        }
    for (int i = BK_TID; i < n; i += BK_AT) {             // u | (k-mer position in the read << 22)
        const unsigned long long key = L_CAND[i]; const uint32_t pk = (uint32_t)(key >> 40) & 0xFFFFu;
        L_CANDU[i] = (uint32_t)(key & 0x3FFFFFull) | ((rev ? 0xFFFFu - pk : pk) << 22);
    }
    BK_SYNC();
    BK_ACC(4);
}

// first occurrence of k-mer `key` in seq[0..n) (str.find), executed by one wavefront: every lane rolls the k-mer at its
// own position out of the LDS bytes and compares keys
__device__ inline int bk_find_kmer_wave(const uint8_t *seq, int n, const BkKey &key, int k)
{
    const int lane = BK_TID & 63;
    for (int b = 0; b + k <= n; b += 64) {
        const int x = b + lane; bool ok = x + k <= n;
        if (ok) { BkKey c; ok = bk_bytes_kmer(seq + x, k, c) && c.lo == key.lo && c.hi == key.hi; }
        const unsigned long long m = __ballot(ok);
        if (m) return b + __ffsll((long long)m) - 1;
    }
    return -1;
}

// ---- contig life cycle ------------------------------------------------------------------------------
__device__ inline void bk_contig_new(int rank, int u, bool in_fifo)                  // contig.__init__ :417-426
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    bk_load_read(u);
    const int len = S->rlen, nreads = S->rn, indel = S->rindel;
    if (len > C_.MAXC) { bk_fail(BK_ST_CONTIG); return; }
    const int base = C_.MAXC - len;
    for (int t = BK_TID; t < len; t += BK_AT) L_CSEQ[base + t] = L_RSEQ[t];
    int32_t *io = bk_cnt_io(0) + base, *ot = bk_cnt_ot(0) + base;
    for (int t = BK_TID; t < len; t += BK_AT) { io[t] = indel ? nreads : 0; ot[t] = indel ? 0 : nreads; }      // :162-165
    BK_SYNC();
    if (BK_TID == 0) {
        S->cbase = base; S->clen = len; S->nbase = base; S->nlen = len; S->cbuf = 0;
        S->serial = ++S->serial_ctr; S->setup = 0; S->founder = u; S->founder_added = 0; S->in_fifo = in_fifo ? 1 : 0;
        S->nk = 0; S->nr = 0; S->nalt = 0; S->kscan = 0;
        C_.kstamp[3 * rank] = S->serial;                 // checked_kmers = [kmer_val]
        C_.ubuf[u] = S->serial;                          // buffer = set([read.id])
    }
    BK_SYNC();
    BK_ACC(9);
}
__device__ inline void bk_fifo_push(int rank, int u)                                 // buffer.add_contig :337-340 (thread 0)
{
    BkAsmShared *S = S_;
    if (C_.ufound[u] >= 0 || (C_.ufl[u] & BK_R_USED)) return;
    C_.pend[2 * S->ptail] = (uint32_t)rank; C_.pend[2 * S->ptail + 1] = (uint32_t)u;
    C_.ufound[u] = S->ptail; S->ptail++; C_.ufl[u] |= BK_R_USED;
}
__device__ inline void bk_add_used_mer(int rank)                                     // thread 0
{
    if (C_.kstate[rank] == BK_K_LIVE) { C_.kstate[rank] = BK_K_USED; C_.usedl[S_->nused++] = (uint32_t)rank; }
}

// check_align's verdict on one read from its two overlap DPs (sv_assembly.py:459-503); tie: the k-mer position rule decides
__device__ inline int bk_decide(const BkNwResult &v1, const BkNwResult &v2, int clen, int rl, int &ds, int &de, bool &tie)
{
    int dec = BK_DEC_NONE; ds = 0; de = 0; tie = false;
    const int minlen = min(clen, rl);
    const bool ok1 = (4 * v1.score >= minlen) && (200 * v1.score >= 179 * (clen - v1.j_start));
    const bool ok2 = (4 * v2.score >= minlen) && (200 * v2.score >= 179 * (rl - v2.j_start));
    if (!ok1 && !ok2) dec = BK_DEC_NONE;
    else if (v1.score == v2.score && v1.j_start == 0 && v1.i_start == 0 && clen == rl) dec = BK_DEC_SAME;
    else if (v1.score == v2.score) {
        if (clen < rl || v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; }
        else if (rl < clen || v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; }
        else tie = true;
    } else if (v1.score > v2.score) {
        if (v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; } else dec = BK_DEC_POST;
    } else {
        if (v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; } else dec = BK_DEC_PRE;
    }
    return dec;
}

// ---- check_align (sv_assembly.py:449-504) + check_read (:552-566) -------------------------------------
// Decision and state update for the read in look-ahead slot `sl`, whose two overlap DPs (v1 = nw(contig, read),
// v2 = nw(read, contig)) were computed against the CURRENT contig.  `rank` = the k-mer that recruited the read.
// Returns (uniform) whether the read matched.
__device__ inline bool bk_retire(int rank, int sl, bool grow)
{
    BkAsmShared *S = S_;
    const int k = C_.k;
    const int u = S->slot[sl].u, rl = S->slot[sl].rl, nreads = S->slot[sl].rn, indel = S->slot[sl].rindel;
    const uint8_t *rseq = L_RSEQ_S(sl);
    const int clen = S->clen;
    const uint8_t *cs = L_CSEQ + S->cbase;
    const int wv = BK_TID >> 6;
    const BkNwResult v1 = S->slot[sl].v1, v2 = S->slot[sl].v2;
    // thread 0 asks now for the per-read words its bookkeeping at the end needs: nobody else writes them meanwhile, and the
    // round trip hides behind the count updates instead of standing between two barriers
    uint32_t pre_fl = 0, pre_kc = 0; int pre_ureads = 0, pre_found = -1;
    if (BK_TID == 0) { pre_fl = C_.ufl[u]; pre_ureads = C_.ureads[u]; pre_kc = C_.kcnt[rank]; if (grow) pre_found = C_.ufound[u]; }
    int ds = 0, de = 0;                             // uniform: computed identically by every thread
    bool tie = false;
    bool synced = false;                            // (uniform) a barrier has been passed since this function was entered
    int dec = bk_decide(v1, v2, clen, rl, ds, de, tie);
    if (tie) {
        synced = true;
        // k-mer position tie-break: x.replace('-','') of the aligned strings are the plain slices
        if (wv == 0) {
            BkKey key; key.hi = C_.khi[rank]; key.lo = C_.klo[rank];
            int i11 = bk_find_kmer_wave(cs + v1.j_start, clen - v1.j_start, key, k);
            int i12 = bk_find_kmer_wave(rseq + v1.i_start, v1.i_end - v1.i_start, key, k);
            int i21 = bk_find_kmer_wave(rseq + v2.j_start, rl - v2.j_start, key, k);
            int i22 = bk_find_kmer_wave(cs + v2.i_start, v2.i_end - v2.i_start, key, k);
            int d = BK_DEC_NONE;
            if (i11 > -1 && i12 > -1) { if ((i21 == -1 && i22 == -1) || (abs(i21 - i22) > abs(i11 - i12))) d = BK_DEC_POST; }
            else if (i21 > -1 && i22 > -1) { if ((i11 == -1 && i12 == -1) || (abs(i21 - i22) < abs(i11 - i12))) d = BK_DEC_PRE; }
            if (BK_TID == 0) S->dec = d;
        }
        BK_SYNC();
        dec = S->dec;
        // contig_overlap_read / read_overlap_contig re-test the containment case (:508, :531)
        if (dec == BK_DEC_POST && v1.j_start == 0) { dec = BK_DEC_SUPER; ds = v1.i_start; de = v1.i_end; }
        if (dec == BK_DEC_PRE && v2.j_start == 0) { dec = BK_DEC_SUB; ds = v2.i_start; de = v2.i_end; }
        BK_SYNC();
    }
    // ---- apply: one fused phase (disjoint index ranges, scalar state committed by thread 0, one barrier) -------
    const bool match = dec != BK_DEC_NONE;
    bool ext = false;                                   // contig was extended (POST / PRE): new k-mers in grow mode
    if (dec == BK_DEC_SUPER) {                                                    // aseq.set_superseq :232-236 (rare: separate phases)
        synced = true;
        bk_counts_superseq(rl, nreads, indel, ds, de);
        const int base = C_.MAXC - rl;
        for (int t = BK_TID; t < rl; t += BK_AT) L_CSEQ[base + t] = rseq[t];
        BK_SYNC();
        if (BK_TID == 0) { S->cbase = base; S->clen = rl; S->pc = S->slot[sl].pos; }
        BK_SYNC();
        if (grow) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);              // set_kmers(skmers) :479/:515
    } else if (dec != BK_DEC_NONE && dec != BK_DEC_SAME) {
        const int cbase = S->cbase, nbase = S->nbase, nlen = S->nlen;
        int32_t *io = bk_cnt_io(S->cbuf), *ot = bk_cnt_ot(S->cbuf);
        int32_t *cv = (indel ? io : ot) + nbase;
        int s0 = ds, s1 = de, pl = 0, at = 0; bool fail = false;
        if (dec == BK_DEC_POST) {                                                 // :520-527, add_postseq :243-250
            pl = max(rl - v1.i_end, 0); s0 = v1.j_start; s1 = clen; at = nbase + nlen;
            fail = cbase + clen + pl > 2 * C_.MAXC || clen + pl > C_.MAXC || at + pl > 2 * C_.MAXC;
            if (!fail) for (int t = BK_TID; t < pl; t += BK_AT) L_CSEQ[cbase + clen + t] = rseq[v1.i_end + t];
        } else if (dec == BK_DEC_PRE) {                                           // :538-545, add_preseq :255-262
            pl = v2.j_start; s0 = v2.i_start; s1 = v2.i_end; at = nbase - pl;
            fail = cbase - pl < 0 || clen + pl > C_.MAXC || at < 0;
            if (!fail) for (int t = BK_TID; t < pl; t += BK_AT) L_CSEQ[cbase - pl + t] = rseq[t];
        }
        if (fail) { bk_fail(BK_ST_CONTIG); synced = true; }
        else {
            for (int t = s0 + BK_TID; t < min(s1, nlen); t += BK_AT) cv[t] += nreads;          // set_counts :195-199 (old coordinates)
            for (int t = BK_TID; t < pl; t += BK_AT) { io[at + t] = indel ? nreads : 0; ot[at + t] = indel ? 0 : nreads; }   // extend_counts :201-221
            // Every wavefront has read the scalars above (cbase, nbase, clen, nlen) before thread 0 replaces them: without
            // this barrier a wavefront that is late into this function (starved by co-resident workgroups) reads the
            // NEW base and updates a range shifted by the prepended length.
            BK_SYNC();
            synced = true;
            if (BK_TID == 0 && dec != BK_DEC_SUB) {
                if (dec == BK_DEC_PRE) { S->cbase = cbase - pl; S->nbase = nbase - pl; S->pc += pl; }
                S->clen = clen + pl; S->nlen = nlen + pl;
            }
            ext = dec != BK_DEC_SUB;
        }
    }
    // check_read bookkeeping (:552-565).  It replaces words the CALLER's control flow reads right before this call, on every wavefront
    // for itself (S->last_dec in bk_retire_checked, the acceptance counters in bk_expect_reject: which retire path is taken, whether the
    // prediction held): a wavefront that is late into this call must have read them before thread 0 writes -- a read that changes
    // nothing (dec NONE / SAME) passes no barrier on its way here, and a late wavefront then took another path than the others,
    // with other barriers (found with a sleeping wavefront behind every barrier, -DBK_JITTER: wrong fixtures at any load; in the
    // field: faults and hangs once several noisy workgroups share a CU).
    if (!synced) BK_SYNC();
    if (BK_TID == 0) {
        S->last_dec = dec; S->hit = match ? 1 : 0;
        if (match) S->n_acc++; else S->n_rej++;
        if (S->n_acc + S->n_rej >= 64) { S->n_acc >>= 1; S->n_rej >>= 1; }
        C_.ubuf[u] = S->serial;                                                    // self.buffer.add(read.id)
        S->cells += 2ull * (unsigned long long)clen * (unsigned long long)rl; S->calls += 2;
        if (match) {
            C_.ufl[u] = (uint8_t)(pre_fl | BK_R_USED);
            if (pre_ureads != S->serial) { C_.ureads[u] = S->serial; C_.readl[S->nr++] = (uint32_t)u; }
            if (grow && pre_found >= 0 && BK_CHK((uint32_t)pre_found <= C_.U, 10, pre_found)) { C_.pend[2 * pre_found] = BK_EMPTY32; C_.ufound[u] = -1; }       // buff.remove_contig :638-639
        } else if (pre_kc > 2 && !(pre_fl & BK_R_USED)) {
            if (S->nalt < C_.MAXCAND) C_.altl[S->nalt++] = (uint32_t)u; else S->status = BK_ST_CAND;
        } else C_.ufl[u] = (uint8_t)(pre_fl | BK_R_DELETED);                       // rb.delete -> rb.clean :390
    }
    BK_SYNC();
    BK_ACC(3);
    if (ext && grow && !S->status) {
        const int k1 = k - 1;
        if (dec == BK_DEC_POST) { const int from = max(clen - k1, 0); bk_kmers_ordered(S->cbase + from, (clen - from) + (S->clen - clen), BK_ORD_FOR); }
        else bk_kmers_ordered(S->cbase, (S->clen - clen) + min(k1, clen), BK_ORD_REV);
    }
    return match;
}

// ---- the read loop of setup_contigs (:16-23) / grow (:634-639) with speculative look-ahead --------------------
// The reference checks the candidate reads strictly one after the other: each accepted read changes the contig
// the next one is aligned to.  The alignment of read q+1 only depends on the contig SEQUENCE after read q, and
// that is predictable from where the recruiting k-mer sits in the read and in the contig (the read sticks out
// `pos - pc` bases to the left, or its tail beyond the contig end).  So each round aligns up to BK_SPEC reads
// at once -- slot s against the contig predicted after slots 0..s-1, two wavefronts per slot -- and then retires
// them in order, for as long as the contig really became what was predicted (same kind of change, same
// geometry => same bytes); the first misprediction discards the later slots, which are redone next round.
// Results are therefore bit-identical to the serial loop; only the DP latency chain gets shorter.
enum { BK_PK_SAME = 0, BK_PK_PRE = 1, BK_PK_POST = 2, BK_PK_STOP = 3 };
// The overlap DPs of one look-ahead round.  Everything it needs is in LDS (slots, contig deque, staged reads); it is kept
// OUT of line so that the dozens of DP variants it dispatches to (one function per column count) have ONE call site
// whose live state is nothing: inlined into the state machine they made the allocator spill around every variant.
__device__ __noinline__ void bk_dp_round()
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6, nb = S->nb;
    if (S->dual && nb > BK_WAVES) {                      // more reads than wavefronts (BK_PAIR builds): one score matrix per read, wavefront w takes slots 2w, 2w+1 (bk_nw_pair)
        const int a = 2 * wv, b = 2 * wv + 1;
        if (a < nb) {
            BkPairArgs A, B;
            A.contig = BK_O_CSEQ + S->slot[a].pb; A.clen = S->slot[a].plen; A.read = BK_O_RSEQ + a * (C_.MAXR + 16); A.n = S->slot[a].rl; A.res = (int)((uint8_t *)&S->slot[a].v1 - bk_lds);
            if (b < nb) { B.contig = BK_O_CSEQ + S->slot[b].pb; B.clen = S->slot[b].plen; B.read = BK_O_RSEQ + b * (C_.MAXR + 16); B.n = S->slot[b].rl; B.res = (int)((uint8_t *)&S->slot[b].v1 - bk_lds); }
            else { B.contig = 0; B.clen = 0; B.read = 0; B.n = 0; B.res = 0; }
            if (S->fast) {
                // the score sweep first: end cells and scores of both calls; the border cells follow without a traceback for overlaps
                // without a mismatch or an indel (bk_nw.hip.h).  A read it cannot settle is swept again in full -- with its partner
                const int nrd = b < nb ? 2 : 1;
                bk_nw_score_pair(A, B);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                bool redo = S->slot[a].v1.j_start == BK_NW_NEEDS_DP || S->slot[a].v2.j_start == BK_NW_NEEDS_DP;
                if (b < nb) redo = redo || S->slot[b].v1.j_start == BK_NW_NEEDS_DP || S->slot[b].v2.j_start == BK_NW_NEEDS_DP;
                redo = __builtin_amdgcn_readfirstlane((int)redo) != 0;
                if ((BK_TID & 63) == 0) { atomicAdd(&S->dp_n, nrd); if (redo) atomicAdd(&S->dp_redo, nrd); }
                if (redo) bk_nw_pair(A, B);
            } else bk_nw_pair(A, B);
        }
    } else if (S->dual) {                                // both DPs of slot wv on this wavefront
        if (wv < nb) {
            const int contig = BK_O_CSEQ + S->slot[wv].pb, clen = S->slot[wv].plen, rd = BK_O_RSEQ + wv * (C_.MAXR + 16), rl = S->slot[wv].rl, res = (int)((uint8_t *)&S->slot[wv].v1 - bk_lds);
            if (S->fast) {
                BkPairArgs A; A.contig = contig; A.clen = clen; A.read = rd; A.n = rl; A.res = res;
                bk_nw_score_one(A);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool redo = __builtin_amdgcn_readfirstlane((int)(S->slot[wv].v1.j_start == BK_NW_NEEDS_DP || S->slot[wv].v2.j_start == BK_NW_NEEDS_DP)) != 0;
                if ((BK_TID & 63) == 0) { atomicAdd(&S->dp_n, 1); if (redo) atomicAdd(&S->dp_redo, 1); }
                if (redo) bk_nw_dual(contig, clen, rd, rl, res);
            } else bk_nw_dual(contig, clen, rd, rl, res);
        }
    } else {                                             // two wavefronts per slot, both with the contig on the tile columns
        const int sl = wv >> 1;
        if (sl < nb) {
            const int cl = S->slot[sl].plen, rl = S->slot[sl].rl;
            if (S->fast) {
                // the score sweep of the whole matrix on ONE of the slot's two wavefronts (any contig length: column tiles); what it
                // cannot settle is swept in full after the round's barrier (bk_dp_redo)
                // (a tile pipeline over BOTH wavefronts of the slot -- alternate column tiles, the second ~128 steps behind on the edge
                //  column in LDS -- was built and measured in round 5: bit-exact, and no faster where long contigs occur: configs[4]
                //  6,367 -> 6,394 ms per batch, configs[3] 945 -> 953; not kept: profiles/r05/score_sweep_ab.txt)
                if ((wv & 1) == 0) {
                    bk_nw_score_long(BK_O_CSEQ + S->slot[sl].pb, cl, BK_O_RSEQ + sl * (C_.MAXR + 16), rl, (int)((uint8_t *)&S->slot[sl].v1 - bk_lds), L_BOUND_W(wv));
                    if ((BK_TID & 63) == 0) atomicAdd(&S->dp_n, 1);
                }
            } else {
                const uint8_t *cs = L_CSEQ + S->slot[sl].pb;
                // waves w and w+4 land on the same SIMD: give it one direct (heavier) and one transposed sweep
                if ((((wv & 1) ^ (wv >> 2)) & 1) == 0) { BkNwResult r = bk_nw_suffix(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v1 = r; }
                else { BkNwResult r = bk_nw_wave<true>(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v2 = r; }
            }
        }
    }
}
// the slots of a two-wavefronts-per-slot round whose score sweep left a border cell open: both overlap DPs in full
__device__ __noinline__ void bk_dp_redo()
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6, sl = wv >> 1;
    if (sl < S->nb && S->slot[sl].dec) {                 // (Slot::dec is free between the staging of a round and its retirement: here it says "sweep again")
        const uint8_t *cs = L_CSEQ + S->slot[sl].pb; const int cl = S->slot[sl].plen, rl = S->slot[sl].rl;
        if ((((wv & 1) ^ (wv >> 2)) & 1) == 0) { BkNwResult r = bk_nw_suffix(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v1 = r; }
        else { BkNwResult r = bk_nw_wave<true>(cs, cl, L_RSEQ_S(sl), rl, L_BOUND_W(wv)); if ((BK_TID & 63) == 0) S->slot[sl].v2 = r; }
    }
}
// noisy reads: check_align has lately rejected three reads out of four (prediction then is "nothing changes", bk_predict)
__device__ inline bool bk_expect_reject() { return S_->n_rej >= 24 && S_->n_rej >= 3 * S_->n_acc; }

// Look-ahead across the k-mer visits of grow.  A visit usually recruits only a handful of reads (clean data) or a single
// one (sequencing noise), far fewer than there are look-ahead slots.  The visits of a snapshot are known in advance
// (nklist), and so is the candidate list of a LATER visit: every read a visit looks at ends up in the contig's buffer
// (check_read :552 buffer.add, matched or not), so the list a later visit will see is its eligible reads now minus the
// reads of the visits in between -- which are exactly the slots planned before it.  A round therefore goes on planning
// into the following visits until the slots are full; their DPs run in the same round against the predicted contig and
// their results wait in the slots.  When such a visit comes up, its real candidate list (the ordinary find_reads) is
// compared with the planned one: equal -> the slots are retired in order under the usual prediction checks, no DP;
// anything else -> the plan is dropped and the visit runs as before.  State only ever changes in bk_retire / finalize,
// in the reference's order.
//
// One wavefront per following visit: its eligible reads (short posting lists only), ordered as find_reads orders them,
// with the per-read fields a slot needs, and the position of its k-mer in the current contig.
#define BK_LA_CH 16                                  // posting lists of up to 64 * BK_LA_CH entries are looked into
#define BK_LA_CU(w) ((uint32_t *)L_CAND + (w) * 64)
#define BK_LA_RL(w) ((int *)L_CAND + (BK_AT / 64 + (w)) * 64)
#define BK_LA_RN(w) ((int *)L_CAND + (2 * (BK_AT / 64) + (w)) * 64)
#define BK_LA_FL(w) ((int *)L_CAND + (3 * (BK_AT / 64) + (w)) * 64)
BK_COLD void bk_lookahead_wave(int w, int vt, int T)
{
    BkAsmShared *S = S_;
    const int lane = BK_TID & 63;
    const int idx = vt + 1 + lane;
    const uint32_t en = idx < T ? C_.nklist[idx] : 0x40000000u;
    unsigned long long m = __ballot(!(en & 0x40000000u));                      // visits that may have candidates, in order
    for (int i = 0; i < w; i++) m &= m - 1;
    int cnt = -1, tt = -1, rank2 = 0, pc2 = -1;
    if (m) {
        const int bit = __ffsll((long long)m) - 1;
        const uint32_t e2 = (uint32_t)__shfl((int)en, bit);
        tt = vt + 1 + bit; rank2 = (int)(e2 & 0x3FFFFFFFu);
        const bool rev = (e2 >> 31) != 0;
        const uint32_t b = C_.poff[rank2], e = C_.poff[rank2 + 1];
        if (e - b <= 64u * BK_LA_CH) {
            // eligible entries of the posting list (a k-mer of a deep region sits in a few hundred reads, nearly all of them
            // in the buffer already): BK_LA_CH entries per lane, loads issued together, compacted into the list
            const int np = (int)(e - b);
            uint32_t pe[BK_LA_CH]; uint32_t fl[BK_LA_CH]; int bs[BK_LA_CH];
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) pe[c] = lane + 64 * c < np ? C_.post[b + lane + 64 * c] : 0u;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) { const bool have = lane + 64 * c < np; fl[c] = have ? C_.ufl[pe[c] >> 10] : 0u; bs[c] = have ? C_.ubuf[pe[c] >> 10] : 0; }
            int nv = 0;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) {
                const bool okc = lane + 64 * c < np && !(fl[c] & BK_R_DELETED) && bs[c] != S->serial;
                const unsigned long long vm = __ballot(okc);
                const int at = nv + __popcll(vm & ((1ull << lane) - 1ull));
                if (okc && at < 64) { BK_LA_RL(w)[at] = (int)pe[c]; BK_LA_FL(w)[at] = (int)fl[c]; }      // staging: overwritten by the ordered list below
                nv += __popcll(vm);
            }
            if (nv <= 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool have = lane < nv;
                const uint32_t mypost = have ? (uint32_t)BK_LA_RL(w)[lane] : 0u; const uint32_t myfl = have ? (uint32_t)BK_LA_FL(w)[lane] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const uint32_t u = mypost >> 10; const int pos = (int)(mypost & 1023u);
                uint32_t len = 0; int rn = 0;
                if (have) { len = C_.ulen[u]; rn = (int)C_.unr[u]; }
                bool drop = false;                                                 // the k-mer twice in one read: first occurrence (re.search)
                for (int j = 0; j < nv; j++) {
                    const uint32_t oen = (uint32_t)__builtin_amdgcn_readlane((int)mypost, j);
                    drop = drop || ((oen >> 10) == u && (int)(oen & 1023u) < pos);
                }
                const bool valid = have && !drop;
                const unsigned long long pk = rev ? (unsigned long long)(0xFFFF - pos) : (unsigned long long)pos;
                const unsigned long long key = valid ? ((pk << 40) | ((0xFFFFull - len) << 24) | u) : ~0ull;
                int ord = 0;
                for (int j = 0; j < nv; j++) {
                    const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                    ord += kj < key;
                }
                if (valid) { BK_LA_CU(w)[ord] = u | ((uint32_t)pos << 22); BK_LA_RL(w)[ord] = (int)len; BK_LA_RN(w)[ord] = rn; BK_LA_FL(w)[ord] = (int)myfl; }
                cnt = __popcll(__ballot(valid));
                BkKey kk; kk.hi = C_.khi[rank2]; kk.lo = C_.klo[rank2];
                if (!bk_expect_reject()) pc2 = bk_find_kmer_wave(L_CSEQ + S->cbase, S->clen, kk, C_.k);      // the geometry is not used while rejections are expected
            }
        }
    }
    if (lane == 0) { S->la_n[w] = cnt; S->la_t[w] = tt; S->la_rank[w] = rank2; S->la_pc[w] = pc2; }
}

// The same for the SEEDS that follow (setup_contigs :11-26).  With sequencing noise most DP rounds are the first round of a
// seed: an error k-mer shared by two or three reads, one of them the founder.  The seeds to come are the next live k-mers in
// (count, mer) order, their candidate lists are find_reads results without a buffer filter (only deleted reads are left out).
// Wavefront w takes the (w+1)-th live k-mer after `rank`: list in find_reads order (entry 0 = founder) with the slot fields.
BK_COLD void bk_seedahead_wave(int w, int rank)
{
    BkAsmShared *S = S_;
    const int lane = BK_TID & 63;
    // (the seeds that follow in THIS unit's list: S->head is where the running seed sits in it)
    const int li = S->head + 1 + lane;
    const int r = li < C_.n_my ? bk_seed_at(li) : -1;
    const bool live = r >= 0 && C_.kstate[r] == BK_K_LIVE && C_.kcnt[r] >= 2;
    unsigned long long m = __ballot(live);
    for (int i = 0; i < w; i++) m &= m - 1;
    int cnt = -1, rank2 = 0;
    (void)rank;
    if (m) {
        rank2 = __shfl(r, __ffsll((long long)m) - 1);
        const uint32_t b = C_.poff[rank2], e = C_.poff[rank2 + 1];
        if (e - b <= 64u * BK_LA_CH) {
            const int np = (int)(e - b);
            uint32_t pe[BK_LA_CH]; uint32_t fl[BK_LA_CH];
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) pe[c] = lane + 64 * c < np ? C_.post[b + lane + 64 * c] : 0u;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) fl[c] = lane + 64 * c < np ? C_.ufl[pe[c] >> 10] : 0u;
            int nv = 0;
#pragma unroll
            for (int c = 0; c < BK_LA_CH; c++) {
                const bool okc = lane + 64 * c < np && !(fl[c] & BK_R_DELETED);             // used_reads = set(): no buffer filter
                const unsigned long long vm = __ballot(okc);
                const int at = nv + __popcll(vm & ((1ull << lane) - 1ull));
                if (okc && at < 64) { BK_LA_RL(w)[at] = (int)pe[c]; BK_LA_FL(w)[at] = (int)fl[c]; }
                nv += __popcll(vm);
            }
            if (nv <= 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const bool have = lane < nv;
                const uint32_t mypost = have ? (uint32_t)BK_LA_RL(w)[lane] : 0u; const uint32_t myfl = have ? (uint32_t)BK_LA_FL(w)[lane] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
                const uint32_t u = mypost >> 10; const int pos = (int)(mypost & 1023u);
                uint32_t len = 0; int rn = 0;
                if (have) { len = C_.ulen[u]; rn = (int)C_.unr[u]; }
                bool drop = false;
                for (int j = 0; j < nv; j++) {
                    const uint32_t oen = (uint32_t)__builtin_amdgcn_readlane((int)mypost, j);
                    drop = drop || ((oen >> 10) == u && (int)(oen & 1023u) < pos);
                }
                const bool valid = have && !drop;
                const unsigned long long key = valid ? (((unsigned long long)pos << 40) | ((0xFFFFull - len) << 24) | u) : ~0ull;
                int ord = 0;
                for (int j = 0; j < nv; j++) {
                    const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                    ord += kj < key;
                }
                if (valid) { BK_LA_CU(w)[ord] = u | ((uint32_t)pos << 22); BK_LA_RL(w)[ord] = (int)len; BK_LA_RN(w)[ord] = rn; BK_LA_FL(w)[ord] = (int)myfl; }
                cnt = __popcll(__ballot(valid));
            }
        }
    }
    if (lane == 0) { S->la_n[w] = cnt; S->la_t[w] = -2 - rank2; S->la_rank[w] = rank2; S->la_pc[w] = 0; }
}

// one step of the prediction chain (thread 0): slot t is aligned against the contig [pb, pb+plen) in which its k-mer sits
// at ppc; what the read is predicted to do to it.  Returns false when nothing can be predicted past this slot.
__device__ inline bool bk_predict(BkAsmShared::Slot &t, int &pb, int &plen, int &ppc, int lo, int hi)
{
    const int pos = t.pos, rl = t.rl;
    t.pb = pb; t.plen = plen;
    // noisy reads: where check_align has lately rejected three reads out of four, the best guess for the next one is that
    // it is rejected too and the contig stays as it is (a rejection after a predicted extension would void the later slots)
    if (bk_expect_reject()) { t.kind = BK_PK_SAME; t.amt = 0; return true; }
    const int left = pos - ppc, right = (rl - pos) - (plen - ppc);
    if (ppc < 0 || (left > 0 && right > 0)) { t.kind = BK_PK_STOP; t.amt = 0; return false; }
    if (left > 0) { t.kind = BK_PK_PRE; t.amt = left; if (pb - left < lo || plen + left > C_.MAXC) { t.kind = BK_PK_STOP; return false; } pb -= left; plen += left; ppc += left; }
    else if (right > 0) { t.kind = BK_PK_POST; t.amt = right; if (pb + plen + right > hi || plen + right > C_.MAXC) { t.kind = BK_PK_STOP; return false; } plen += right; }
    else { t.kind = BK_PK_SAME; t.amt = 0; }
    return true;
}

// Run retire (BK_F_NO_RUN_RETIRE switches it off).  The slots of a round are retired together for as long as every one of
// them does what was predicted: the decisions of slots sl .. s1-1 are taken at once (a lane each, against the contig length
// each was aligned to), thread 0 walks the prediction chain over them -- a rejected, identical or contained read leaves the
// sequence alone, an extension (contig_overlap_read :520-527, read_overlap_contig :538-545) must be the predicted kind and
// length, which makes the bytes the predicted ones -- and the run is applied in ONE pass: the appended / prepended bases, the
// count entries of the new bases (assigned, extend_counts :201-221), then one summed count update per base over the ranges
// of all reads of the run in the coordinates each range was taken in (set_counts :195-199; old coordinates before a
// prepend), one lane per read for the per-read words, appends in slot order.  Left to the one-read path (bk_retire): a read
// that replaces the contig (superseq), a tie of the two scores (k-mer position rule), the first read of a seed planned
// ahead, an extension that fails a bound, and in grow mode an extension by two or more bases (it adds contig k-mers, in
// order, bk_kmers_ordered; an extension by one base adds none, Q1).  Four barriers per run instead of two to three per read.
// The number retired is left in S->tmp0 (uniform after the function's last barrier).
__device__ __forceinline__ void bk_retire_run(int sl, int s1, bool grow)
{
    BkAsmShared *S = S_;
    const int nbt = s1 - sl;
    const int cbase0 = S->cbase, clen0 = S->clen, nbase0 = S->nbase, nlen0 = S->nlen;      // read by every thread before thread 0 replaces them (two barriers on)
    // wavefront 0, a lane per slot: the decision, whether the slot does what it may do here, whether the contig it was aligned
    // against is what its predecessor leaves behind; the run = the slots before the first lane that says no.  The per-read
    // words the bookkeeping at the end needs are asked for now (nobody writes them in between).
    int my_u = 0, my_dec = BK_DEC_NONE, my_found = -1, my_ur = 0; uint32_t my_fl = 0, my_kc = 0; unsigned long long my_cells = 0;
    if ((BK_TID >> 6) == 0) {
        const int j = BK_TID;
        const bool have = j < nbt;
        bool ok = false; int dec = BK_DEC_NONE, pl = 0, a0 = 0, a1 = 0, cb_after = 0, cl_after = 0, pb = 0, plen = 0;
        if (have) {
            BkAsmShared::Slot &t = S->slot[sl + j];
            my_u = t.u; my_fl = C_.ufl[my_u]; my_ur = C_.ureads[my_u]; my_kc = C_.kcnt[t.rank]; if (grow) my_found = C_.ufound[my_u];
            pb = t.pb; plen = t.plen;
            int ds, de; bool tie;
            dec = bk_decide(t.v1, t.v2, plen, t.rl, ds, de, tie);
            if (dec == BK_DEC_POST) { pl = max(t.rl - t.v1.i_end, 0); ds = t.v1.j_start; de = plen; }
            else if (dec == BK_DEC_PRE) { pl = t.v2.j_start; ds = t.v2.i_start; de = t.v2.i_end; }
            const int nb_ = nbase0 - (cbase0 - pb), nl_ = nlen0 + (plen - clen0);          // the count vectors move with the contig
            ok = !t.first && !tie && dec != BK_DEC_SUPER && !(grow && pl >= 2);
            if (dec == BK_DEC_POST) ok = ok && !(pb + plen + pl > 2 * C_.MAXC || plen + pl > C_.MAXC || nb_ + nl_ + pl > 2 * C_.MAXC);
            else if (dec == BK_DEC_PRE) ok = ok && !(pb - pl < 0 || plen + pl > C_.MAXC || nb_ - pl < 0);
            if (dec != BK_DEC_NONE && dec != BK_DEC_SAME) { a0 = nb_ + ds; a1 = max(nb_ + min(de, nl_), a0); }      // absolute span of the count update
            cb_after = dec == BK_DEC_PRE ? pb - pl : pb; cl_after = (dec == BK_DEC_PRE || dec == BK_DEC_POST) ? plen + pl : plen;
            my_dec = dec; my_cells = 2ull * (unsigned long long)plen * (unsigned long long)t.rl;
            if (ok) { t.dec = dec; t.ds = a0; t.de = a1; t.hasn = pl; }      // hasn: only read when the slot is staged; from here on the length of the extension
        }
        // the contig this slot was aligned against = what the slot before it leaves (slot sl: the contig as it is)
        int pcb = __shfl_up(cb_after, 1), pcl = __shfl_up(cl_after, 1), pdec = __shfl_up(dec, 1);
        if (j == 0) { pcb = cbase0; pcl = clen0; pdec = S->last_dec; }
        if (have) {
            bool chain = pcb == pb && pcl == plen;
            if (sl + j > 0) {
                const int pk = S->slot[sl + j - 1].kind;
                chain = chain && ((pk == BK_PK_PRE && pdec == BK_DEC_PRE) || (pk == BK_PK_POST && pdec == BK_DEC_POST) ||
                                  (pk == BK_PK_SAME && (pdec == BK_DEC_NONE || pdec == BK_DEC_SAME || pdec == BK_DEC_SUB)));
            }
            ok = ok && chain;
        }
        const unsigned long long bad = ~__ballot(ok);
        const int m = bad ? __ffsll((long long)bad) - 1 : 64;              // lanes >= nbt say no: m <= nbt
        // span of all count updates of the run, geometry after its last slot
        int lo = (j < m && a1 > a0) ? a0 : 0x7FFFFFFF, hi = (j < m && a1 > a0) ? a1 : 0;
        for (int o = 1; o < 16; o <<= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }      // BK_SPEC <= 16 slots
        if (j == 0) { S->tmp0 = m; S->tmp1 = hi > 0 ? lo : 0; S->tmp2 = hi; }
        if (m > 0 && j == m - 1) { S->dstart = cb_after; S->dend = cl_after; }
    }
    BK_SYNC();
    const int m = S->tmp0;
    if (m == 0) return;
    const int lo = S->tmp1, hi = S->tmp2, cbase1 = S->dstart, clen1 = S->dend;
    int32_t *io = bk_cnt_io(S->cbuf), *ot = bk_cnt_ot(S->cbuf);
    // new bases and their count entries: a wavefront per slot (the bases of all slots but the round's last were pre-written by the staging; the same bytes)
    for (int j = BK_TID >> 6; j < m; j += BK_AT / 64) {
        const BkAsmShared::Slot &q = S->slot[sl + j];
        const int pl = q.hasn;
        if ((q.dec != BK_DEC_POST && q.dec != BK_DEC_PRE) || pl == 0) continue;
        const int nb_ = nbase0 - (cbase0 - q.pb), nl_ = nlen0 + (q.plen - clen0);
        const uint8_t *rs = L_RSEQ_S(sl + j);
        const int cat_io = q.rindel ? q.rn : 0, cat_ot = q.rindel ? 0 : q.rn;
        if (q.dec == BK_DEC_POST) {
            for (int t = BK_TID & 63; t < pl; t += 64) { L_CSEQ[q.pb + q.plen + t] = rs[q.rl - pl + t]; io[nb_ + nl_ + t] = cat_io; ot[nb_ + nl_ + t] = cat_ot; }
        } else {
            for (int t = BK_TID & 63; t < pl; t += 64) { L_CSEQ[q.pb - pl + t] = rs[t]; io[nb_ - pl + t] = cat_io; ot[nb_ - pl + t] = cat_ot; }
        }
    }
    BK_SYNC();
    for (int t = lo + BK_TID; t < hi; t += BK_AT) {
        int a = 0, b = 0;
        for (int j = 0; j < m; j++) {
            const BkAsmShared::Slot &q = S->slot[sl + j];
            if (t >= q.ds && t < q.de) { if (q.rindel) a += q.rn; else b += q.rn; }      // empty span for a read that changes no count
        }
        if (a) io[t] += a;
        if (b) ot[t] += b;
    }
    if ((BK_TID >> 6) == 0) {
        const int j = BK_TID;
        const bool have = j < m;
        const int u = my_u, dec = my_dec;
        const bool match = have && dec != BK_DEC_NONE;
        const bool to_list = match && my_ur != S->serial;
        const bool to_alt = have && !match && my_kc > 2 && !(my_fl & BK_R_USED);
        const unsigned long long lm = __ballot(to_list), am = __ballot(to_alt), mm = __ballot(match), below = (1ull << j) - 1ull;
        if (have) {
            C_.ubuf[u] = S->serial;
            if (match) {
                C_.ufl[u] = (uint8_t)(my_fl | BK_R_USED);
                if (to_list) { C_.ureads[u] = S->serial; C_.readl[S->nr + __popcll(lm & below)] = (uint32_t)u; }
                if (grow && my_found >= 0 && BK_CHK((uint32_t)my_found <= C_.U, 11, my_found)) { C_.pend[2 * my_found] = BK_EMPTY32; C_.ufound[u] = -1; }
            } else if (to_alt) {
                const int at = S->nalt + __popcll(am & below);
                if (at < C_.MAXCAND) C_.altl[at] = (uint32_t)u; else S->status = BK_ST_CAND;
            } else C_.ufl[u] = (uint8_t)(my_fl | BK_R_DELETED);
        }
        unsigned long long cells = have ? my_cells : 0ull;
        for (int o = 1; o < 16; o <<= 1) cells += __shfl_xor(cells, o);
        if (BK_TID == 0) {
            S->nr += __popcll(lm); S->nalt = min(S->nalt + __popcll(am), C_.MAXCAND);
            int na = S->n_acc, nr_ = S->n_rej;
            for (int i = 0; i < m; i++) { if ((mm >> i) & 1ull) na++; else nr_++; if (na + nr_ >= 64) { na >>= 1; nr_ >>= 1; } }
            S->n_acc = na; S->n_rej = nr_;
            S->cells += cells; S->calls += 2 * m;
            const bool lastm = ((mm >> (m - 1)) & 1ull) != 0;
            S->last_dec = S->slot[sl + m - 1].dec; S->hit = lastm ? 1 : 0;
            S->pc += cbase0 - cbase1;                                       // the recruiting k-mer moves with what was prepended
            S->nbase = nbase0 - (cbase0 - cbase1); S->nlen = nlen0 + (clen1 - clen0); S->cbase = cbase1; S->clen = clen1;
#ifdef BK_PHASE_STAMPS
            S->acc[17] += m;
#endif
        }
    }
    BK_SYNC();
}

// retire slot sl if the contig is what the slot was aligned against; uniform result: 1 retired, 0 prediction failed
__device__ __forceinline__ int bk_retire_checked(int sl, bool grow)
{
    BkAsmShared *S = S_;
    if (S->slot[sl].first) {                        // first read of a seed planned ahead: the contig is its founder, nothing came before
        if (S->cbase != S->slot[sl].pb || S->clen != S->slot[sl].plen) return 0;
    } else if (sl > 0) {
        const int pk = S->slot[sl - 1].kind, ld = S->last_dec;
        const bool kind_ok = (pk == BK_PK_PRE && ld == BK_DEC_PRE) || (pk == BK_PK_POST && ld == BK_DEC_POST) ||
                             (pk == BK_PK_SAME && (ld == BK_DEC_NONE || ld == BK_DEC_SAME || ld == BK_DEC_SUB));
        if (!kind_ok || S->cbase != S->slot[sl].pb || S->clen != S->slot[sl].plen) return 0;
    }
    (void)bk_retire(S->slot[sl].rank, sl, grow);    // ends with a barrier; the FIFO entry of a matched read is dropped in its bookkeeping
#ifdef BK_PHASE_STAMPS
    if (BK_TID == 0) S->acc[17] += 1;
#endif
    return 1;
}

// the plan of one round (thread 0): prediction chain over this visit's slots, then over the following visits' lists.  Out of
// line: its loops over slots and lists would otherwise sit in the register budget of the state machine's hot loop.
BK_COLD void bk_plan_round(int q, int n, int nbmax, int cap, int vt, int la, int la_on)
{
    BkAsmShared *S = S_;
    if (BK_TID != 0) return;
    int pb = S->cbase, plen = S->clen, ppc = S->pc, nb = 0;
    bool go = true;
    // both DPs of a slot run on one wavefront while the (predicted) contig fits its columns; else two wavefronts per slot
    for (int sl = 0; sl < nbmax && go; sl++) {
        if (sl >= BK_SPEC_WIDE && plen > BK_NW_DUAL_COLS) { go = false; break; }
        nb = sl + 1;
        go = bk_predict(S->slot[sl], pb, plen, ppc, 0, 2 * C_.MAXC);
    }
    const int ncur = nb;
    int upto = vt;
#ifdef BK_PHASE_STAMPS
    if (la_on) S->acc[19] += 1ull;                                               // rounds with free slots
    if (la_on && !(la && go && q + ncur == n)) S->acc[19] += 1ull << 16;          // ... not looked ahead (paused / STOP / round does not finish the visit)
    else if (la_on && S->la_n[0] < 0) S->acc[19] += 1ull << 32;                  // ... next visit: none in the window or posting list too long
    else if (la_on && S->la_n[0] == 0) S->acc[19] += 1ull << 48;                 // ... next visit has no eligible read
#endif
    if (la && vt >= 0 && go && q + ncur == n) {
        for (int w = 0; w < BK_AT / 64 && go; w++) {
            const int cn = S->la_n[w];
            if (cn < 0) break;
            // the visit's list once the reads planned before it are in the buffer
            int keep = 0;
            for (int i = 0; i < cn; i++) {
                const int u = (int)(BK_LA_CU(w)[i] & 0x3FFFFFu); bool inflight = false;
                for (int s2 = 0; s2 < nb; s2++) inflight = inflight || S->slot[s2].u == u;
                keep += !inflight;
                if (nb + keep > cap) break;                         // does not fit: no need to look at the rest of the list
            }
            if (nb + keep > cap) break;
            const int pc2 = S->la_pc[w];
            if (keep > 0 && pc2 < 0 && !bk_expect_reject()) break;
            ppc = pc2 + (S->cbase - pb);                    // the predicted contig starts cbase - pb bases before the current one
            const int nb0 = nb;
            for (int i = 0; i < cn && go; i++) {
                const uint32_t cu = BK_LA_CU(w)[i]; const int u = (int)(cu & 0x3FFFFFu); bool inflight = false;
                for (int s2 = 0; s2 < nb0; s2++) inflight = inflight || S->slot[s2].u == u;
                if (inflight) continue;
                if (nb >= BK_SPEC_WIDE && plen > BK_NW_DUAL_COLS) { go = false; break; }
                BkAsmShared::Slot &t = S->slot[nb];
                const int fl = BK_LA_FL(w)[i];
                t.u = u; t.pos = (int)(cu >> 22); t.rl = BK_LA_RL(w)[i]; t.rn = BK_LA_RN(w)[i]; t.rindel = (fl & BK_R_INDEL) ? 1 : 0;
                t.hasn = (C_.n_nlist && (fl & BK_R_HASN)) ? 1 : 0;
                t.vt = S->la_t[w]; t.rank = S->la_rank[w]; t.fu = -1; t.first = 0;
                nb++;
                go = bk_predict(t, pb, plen, ppc, 0, 2 * C_.MAXC);
            }
            if (nb - nb0 != keep) { nb = nb0; break; }      // a visit is planned whole or not at all
            upto = S->la_t[w];
        }
    }
    int kind = 0;
    if (la && vt < 0 && q + ncur == n) {
        // the first round of the seeds that follow: a group per seed, aligned against its founder in a strip of its own
        kind = 1;
        int g = 0;
        for (int w = 0; w < BK_AT / 64; w++) {
            const int cn = S->la_n[w];
            if (cn < 0) break;
            if (cn < 2) continue;                                   // a seed with its founder only has no DP
            if (nb + cn - 1 > cap) break;
            const uint32_t fcu = BK_LA_CU(w)[0];
            const int base = BK_SEEDBUF(g) + C_.MAXR + 16, lo = BK_SEEDBUF(g), hi = BK_SEEDBUF(g) + 3 * (C_.MAXR + 16);
            int pb2 = base, plen2 = BK_LA_RL(w)[0], ppc2 = (int)(fcu >> 22);          // the contig IS the founder; the k-mer sits where it sits in that read
            const int nb0 = nb; bool whole = true;
            for (int i = 1; i < cn; i++) {
                const uint32_t cu = BK_LA_CU(w)[i];
                if (nb >= BK_SPEC_WIDE && plen2 > BK_NW_DUAL_COLS) { whole = false; break; }
                BkAsmShared::Slot &t = S->slot[nb];
                const int fl = BK_LA_FL(w)[i];
                t.u = (int)(cu & 0x3FFFFFu); t.pos = (int)(cu >> 22); t.rl = BK_LA_RL(w)[i]; t.rn = BK_LA_RN(w)[i]; t.rindel = (fl & BK_R_INDEL) ? 1 : 0;
                t.hasn = (C_.n_nlist && (fl & BK_R_HASN)) ? 1 : 0;
                t.vt = S->la_t[w]; t.rank = S->la_rank[w]; t.fu = (int)(fcu & 0x3FFFFFu); t.first = (i == 1) ? 1 : 0;
                nb++;
                if (!bk_predict(t, pb2, plen2, ppc2, lo, hi) && i + 1 < cn) { whole = false; break; }
            }
            if (!whole) { nb = nb0; continue; }                     // a seed is planned whole or not at all
            g++;
        }
    }
    int mx = 0;
    for (int sl = 0; sl < nb; sl++) mx = max(mx, S->slot[sl].plen);
    if (kind == 1 && mx > BK_NW_TILE_COLS) { nb = ncur; mx = 0; for (int sl = 0; sl < nb; sl++) mx = max(mx, S->slot[sl].plen); }      // a multi-tile DP would use the scratch the strips sit in
    // both DPs of a slot on one wavefront when more than BK_SPEC_WIDE reads are in the round; with fewer, the idle
    // wavefronts take the second DP (two 64-lane sweeps finish sooner than one half-wave pair)
    S->dual = mx <= BK_NW_DUAL_COLS && !(C_.flags & BK_F_NO_DUAL) && (nb > BK_SPEC_WIDE || (C_.flags & BK_F_DUAL_ALWAYS));
    if (S->dp_n >= 64) { S->dp_tot += S->dp_n - (S->dp_n >> 1); S->dp_rtot += S->dp_redo - (S->dp_redo >> 1); S->dp_n >>= 1; S->dp_redo >>= 1; }
    S->fast = !(C_.flags & BK_F_NO_SCORE_SWEEP) && 4 * S->dp_redo <= S->dp_n + 8;      // (a quarter of the reads swept twice: the score sweep costs half a full one, so it still pays)
    int nc = ncur;
    if (!S->dual && nb > BK_SPEC_WIDE) { nc = min(nc, BK_SPEC_WIDE); nb = nc; upto = vt; }      // two wavefronts per slot: this visit's reads only
    S->nb = nb; S->ncur = nc;
    S->plan_r = nc; S->plan_upto = upto; S->plan_ok = (nb > nc || upto > vt) ? 1 : 0; S->plan_kind = kind;
    S->la_planned += nb - nc;
    if (la_on && S->la_pause > 0) S->la_pause--;
    if (S->la_planned >= 64) {                      // one window: did the slots planned for later visits get used?
        if (4 * S->la_adopted < S->la_planned) { S->la_pause = S->la_backoff; S->la_backoff = min(2 * S->la_backoff, 4096); }
        else S->la_backoff = 32;
        S->la_planned = 0; S->la_adopted = 0;
    }
}

// vt / T: index of this visit in the snapshot and the snapshot's length (grow); vt < 0: setup_contigs
__device__ __forceinline__ void bk_run_candidates(int rank, int first, int n, bool grow, int vt, int T)
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6;
    int q = first;
    // 0. a plan of an earlier round covers this visit: adopt its slots if it predicted exactly this candidate list
    //    (uniform: every thread evaluates the same LDS words)
    bool adopt = false;
    int r0 = 0;
    if (vt >= 0 && S->plan_ok && S->plan_kind == 0) {
        r0 = S->plan_r;
        adopt = vt <= S->plan_upto;
        int g = 0;
        if (adopt) { while (r0 + g < S->nb && S->slot[r0 + g].vt == vt) g++; adopt = g == n; }
        for (int i = 0; adopt && i < n; i++) { const uint32_t cu = L_CANDU[i]; adopt = S->slot[r0 + i].u == (int)(cu & 0x3FFFFFu) && S->slot[r0 + i].pos == (int)(cu >> 22); }
        if (!adopt || n == 0) {
            BK_SYNC();
            if (BK_TID == 0) { if (!adopt) S->plan_ok = 0; else if (r0 >= S->nb && vt >= S->plan_upto) S->plan_ok = 0; }
            BK_SYNC();
            adopt = false;
        }
    }
    if (vt == -1 && S->plan_ok && S->plan_kind == 1) {
        // a setup round planned the first round of this seed ahead: same founder, same candidates -> adopt
        const int nbp = S->nb;
        int r = S->plan_r;
        while (r < nbp && -2 - S->slot[r].vt < rank) r++;          // seeds that never came up (their k-mer was used up meanwhile)
        int g = 0; bool ad = false;
        if (r < nbp && -2 - S->slot[r].vt == rank) {
            while (r + g < nbp && S->slot[r + g].vt == S->slot[r].vt) g++;
            ad = n >= 2 && g == n - 1 && S->slot[r].first && S->slot[r].fu == (int)(L_CANDU[0] & 0x3FFFFFu);
            for (int i = 0; ad && i < g; i++) { const uint32_t cu = L_CANDU[1 + i]; ad = S->slot[r + i].u == (int)(cu & 0x3FFFFFu) && S->slot[r + i].pos == (int)(cu >> 22); }
        }
        BK_SYNC();
        if (BK_TID == 0) {
            if (ad) { const int delta = S->cbase - S->slot[r].pb; for (int i = 0; i < g; i++) S->slot[r + i].pb += delta; S->plan_r = r; }      // strip -> contig deque coordinates
            else { S->plan_r = r + g; if (S->plan_r >= nbp) S->plan_ok = 0; }
        }
        BK_SYNC();
        if (ad) { adopt = true; r0 = r; }
    }
    // one loop for both kinds of pass, so that the decision/apply step (bk_retire) is inlined once: a pass either retires
    // the adopted slots of this visit or plans, aligns and retires a fresh round
    while (adopt || q < n) {
        int s0 = r0, s1 = r0 + (n - q);             // the adopted slots of this visit / seed ...
        if (!adopt) {                               // ... or a fresh round:
        if (S->status) return;
        BK_ACC(S_->ctx);
        // slots of a round: one per wavefront while both DPs of a slot fit one wavefront (contig <= BK_NW_DUAL_COLS), else half
        const int cap = ((C_.flags & (BK_F_NO_DUAL | BK_F_SPEC4)) || S->clen > BK_NW_DUAL_COLS) ? BK_SPEC_WIDE : BK_SPEC;
        const int nbmax = min((C_.flags & (BK_F_NO_DUAL | BK_F_SPEC4)) ? BK_SPEC_WIDE : BK_SPEC, n - q);
        // look into the following visits when this one leaves slots free (needs the scratch for BK_AT/64 lists of 64 reads)
        // (split regions, bk_comp.hip.h: until round 4 the look-ahead was off inside them -- with it the assembler faulted about once
        //  in 25 runs of a 64-region noisy batch.  The causes were two missing barriers (bk_retire's bookkeeping, round 4; the
        //  S->foreign loop of bk_kmers_ordered, round 5), not the plans; BK_F_SPLIT_NO_LOOKAHEAD is the round-4 setting.)
        const bool la_on = n - q < cap && !(C_.flags & (vt >= 0 ? BK_F_NO_XVISIT : BK_F_NO_XSEED)) && 2 * C_.MAXCAND >= 4 * BK_AT && !(C_.split && (C_.flags & BK_F_SPLIT_NO_LOOKAHEAD));
        const bool la = la_on && S->la_pause == 0;
        // 1. stage the reads of this round (one lane per slot fetches the read's metadata), plan the predictions
        BK_SYNC();
        if (BK_TID < nbmax) {
            const uint32_t cu = L_CANDU[q + BK_TID]; const int u = (int)(cu & 0x3FFFFFu);
            const uint32_t ri = C_.urep[u];
            BkAsmShared::Slot &t = S->slot[BK_TID];
            t.u = u; t.pos = (int)(cu >> 22); t.rl = C_.rlen[ri]; t.rn = (int)C_.unr[u]; t.rindel = (C_.ufl[u] & BK_R_INDEL) ? 1 : 0;
            t.hasn = (C_.n_nlist && (C_.ufl[u] & BK_R_HASN)) ? 1 : 0;
            t.vt = vt; t.rank = rank; t.fu = -1; t.first = 0;
        }
        if (la) { if (vt >= 0) bk_lookahead_wave(wv, vt, T); else bk_seedahead_wave(wv, rank); }
        BK_SYNC();
        bk_plan_round(q, n, nbmax, cap, vt, la ? 1 : 0, la_on ? 1 : 0);
        BK_SYNC();
        const int nb = S->nb, ncur = S->ncur;
#ifdef BK_PHASE_STAMPS
        if (BK_TID == 0) { S->acc[16] += nb; S->acc[18] += 1; }
#endif
        {   // unpack the reads; pre-write the bytes slot sl is predicted to add.  One wavefront per slot: the global loads
            // of all slots are in flight together (slot after slot they were nb dependent round trips per round)
            const int ln = BK_TID & 63;
            for (int sl = BK_TID >> 6; sl < nb; sl += BK_WAVES) {
                const int rl = S->slot[sl].rl; uint8_t *rs = L_RSEQ_S(sl);
                const uint32_t *w = C_.reads + (uint64_t)C_.urep[S->slot[sl].u] * C_.read_words;
                for (int t = ln; t < rl; t += 64) rs[t] = (uint8_t)seq_base(w, t);
                if (S->slot[sl].first) {                 // founder of a seed planned ahead: the contig its reads are aligned against
                    const uint32_t fri = C_.urep[S->slot[sl].fu];
                    const uint32_t *fw = C_.reads + (uint64_t)fri * C_.read_words;
                    uint8_t *fs = L_CSEQ + S->slot[sl].pb; const int fl2 = S->slot[sl].plen;
                    for (int t = ln; t < fl2; t += 64) fs[t] = (uint8_t)seq_base(fw, t);
                    if (C_.n_nlist && (C_.ufl[S->slot[sl].fu] & BK_R_HASN)) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                        if (ln == 0) bk_patch_n(fri, fs, 0, fl2);
                    }
                }
                if (sl + 1 < nb) {
                    const int amt = S->slot[sl].amt, pb = S->slot[sl].pb, plen = S->slot[sl].plen;
                    if (S->slot[sl].kind == BK_PK_PRE) for (int t = ln; t < amt; t += 64) L_CSEQ[pb - amt + t] = (uint8_t)seq_base(w, t);
                    else if (S->slot[sl].kind == BK_PK_POST) for (int t = ln; t < amt; t += 64) L_CSEQ[pb + plen + t] = (uint8_t)seq_base(w, rl - amt + t);
                }
            }
        }
        BK_SYNC();
        if (C_.n_nlist) {                                 // reads with N calls (rare): code 4 over the unpacked bytes and over the predicted contig bytes
            bool any = false;
            for (int sl = 0; sl < nb; sl++) any = any || S->slot[sl].hasn;
            if (any) {
                if (BK_TID < nb && S->slot[BK_TID].hasn) {
                    const BkAsmShared::Slot &t = S->slot[BK_TID]; const uint32_t ri = C_.urep[t.u];
                    bk_patch_n(ri, L_RSEQ_S(BK_TID), 0, t.rl);
                    if (BK_TID + 1 < nb) {
                        if (t.kind == BK_PK_PRE) bk_patch_n(ri, L_CSEQ + t.pb - t.amt, 0, t.amt);
                        else if (t.kind == BK_PK_POST) bk_patch_n(ri, L_CSEQ + t.pb + t.plen, t.rl - t.amt, t.amt);
                    }
                }
                BK_SYNC();
            }
        }
        BK_ACC(1);
        // 2. the overlap DPs (:451-452) of every slot of this round
        bk_dp_round();
        BK_SYNC();
        if (!S->dual && S->fast) {                      // (uniform) two wavefronts per slot: did the score sweep leave a border cell open?
            bool any = false;
            for (int sl = 0; sl < nb; sl++) any = any || S->slot[sl].v1.j_start == BK_NW_NEEDS_DP || S->slot[sl].v2.j_start == BK_NW_NEEDS_DP;
            if (any) {
                BK_SYNC();                              // every wavefront has looked at the result words before they change
                if (BK_TID < nb) {
                    const int f = (S->slot[BK_TID].v1.j_start == BK_NW_NEEDS_DP || S->slot[BK_TID].v2.j_start == BK_NW_NEEDS_DP) ? 1 : 0;
                    S->slot[BK_TID].dec = f;
                    if (f) atomicAdd(&S->dp_redo, 1);
                }
                BK_SYNC();
                bk_dp_redo();
                BK_SYNC();
            }
        }
        BK_ACC(2);
        s0 = 0; s1 = ncur;
        }
        // 3. retire this visit's slots in order while the predictions hold; the slots of later visits wait for their turn
        int sl = s0;
        while (sl < s1) {
            if (S->status) return;
            // the reads that change nothing, together -- where there are runs of them: on clean data nearly every read
            // extends the contig by a base and the attempt only costs its two barriers (same-box A/B on the headline:
            // 1.82 ms with it always on, 1.70 ms without), so it waits until check_align has lately rejected most reads
            if (!(C_.flags & BK_F_NO_RUN_RETIRE) && (s1 - sl >= 2 || bk_expect_reject())) {
                bk_retire_run(sl, s1, grow);
                const int m = S->tmp0;
                sl += m; q += m;
                if (sl >= s1 || S->status) break;
            }
            if (!bk_retire_checked(sl, grow)) break;        // the read that extends / replaces the contig (or fails the prediction)
            sl++; q++;
        }
        if (adopt) {
            BK_SYNC();
            if (BK_TID == 0) {
                S->la_adopted += sl - s0;
                if (sl < s1) S->plan_ok = 0;
                else { S->plan_r = s1; if (S->plan_r >= S->nb && vt >= S->plan_upto) S->plan_ok = 0; }
            }
            BK_SYNC();
            adopt = false;
        } else if (sl < s1 && S->plan_ok) { BK_SYNC(); if (BK_TID == 0) S->plan_ok = 0; BK_SYNC(); }      // a prediction failed: what was planned behind it is void
    }
}

// ---- check_alt_reads (sv_assembly.py:568-582) + the adds of finalize (:590-592) -------------------------
BK_COLD void bk_check_alt_reads()
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const int k = C_.k;
    const int nalt = S->nalt;
    if (nalt == 0) return;
    BK_SYNC();
    if (BK_TID == 0) S->tmp2 = ++S->stamp_ctr;          // identifies mer_set of this call
    BK_SYNC();
    const int fin = S->tmp2;
    int *tmp = (int *)L_CAND;
    for (int a = 0; a < nalt; a++) {
        const int u = (int)C_.altl[a];
        bk_load_read(u);
        const int len = S->rlen, np = len - k;
        // x = get_read_kmers(read) - used_mers - mer_set   (set(self.kmers) holds tuples: removes nothing)
        BkKey best; best.hi = ~0ull; best.lo = ~0ull; int bestrk = -1; int anyx = 0;
        for (int x = BK_TID; x < np; x += BK_AT) {
            BkKey key; uint32_t st = 0; int rk = bk_bytes_kmer(L_RSEQ + x, k, key) ? bk_lookup_state(key, st) : -1;
            if (rk >= 0 && (st != BK_K_LIVE || C_.kstamp[3 * rk + 1] == fin)) rk = -1;
            tmp[x] = rk;
            if (rk >= 0) { anyx = 1; if (C_.kcnt[rk] > 1 && key_lt(key, best)) { best = key; bestrk = rk; } }   // sorted(x) (P2): smallest mer with count > 1
        }
        // reduce the minimum key over the block
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long oh = __shfl_xor(best.hi, o), ol = __shfl_xor(best.lo, o); int ork = __shfl_xor(bestrk, o);
            BkKey ob; ob.hi = oh; ob.lo = ol;
            if (ork >= 0 && (bestrk < 0 || key_lt(ob, best))) { best = ob; bestrk = ork; }
        }
        unsigned long long *red = (unsigned long long *)(tmp + 2 * C_.MAXCAND - 64);   // tail of the scratch: (hi, lo, rk) per wavefront (8 x 3 x 8 B <= 256 B)
        BK_SYNC();
        if ((BK_TID & 63) == 0) { int w = BK_TID >> 6; red[3 * w] = best.hi; red[3 * w + 1] = best.lo; red[3 * w + 2] = (unsigned long long)(long long)bestrk; }
        BK_SYNC();
        { best.hi = red[0]; best.lo = red[1]; bestrk = (int)(long long)red[2];
          for (int w = 1; w < BK_AT / 64; w++) { BkKey ob; ob.hi = red[3 * w]; ob.lo = red[3 * w + 1]; int ork = (int)(long long)red[3 * w + 2];
              if (ork >= 0 && (bestrk < 0 || key_lt(ob, best))) { best = ob; bestrk = ork; } } }
        BK_SYNC();
        (void)anyx;
        if (bestrk >= 0) {
            for (int x = BK_TID; x < np; x += BK_AT) if (tmp[x] >= 0) C_.kstamp[3 * tmp[x] + 1] = fin;       // mer_set = mer_set | x
            if (BK_TID == 0) bk_fifo_push(bestrk, u);
        }
        BK_SYNC();
    }
    BK_ACC(6);
}

// ---- finalize (sv_assembly.py:584-599) ----------------------------------------------------------------
__device__ inline void bk_finalize(bool setup)
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    if (setup) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);                 // set_kmers(akmers.smers_set)
    bk_check_alt_reads();
    BK_SYNC();
    if (BK_TID == 0) {
        if (!S->founder_added) {                                                  // batch_reads[0] = founder, aligned (:383)
            S->founder_added = 1; int u = S->founder;
            if (C_.ureads[u] != S->serial) { C_.ureads[u] = S->serial; C_.readl[S->nr++] = (uint32_t)u; }
        }
        S->nalt = 0;
    }
    BK_SYNC();
    BK_ACC(10);
}

// ---- contig.grow (sv_assembly.py:616-649) --------------------------------------------------------------
__device__ __forceinline__ void bk_grow()
{
    BK_ACC(S_->ctx); BK_CTX(15);
    BkAsmShared *S = S_;
    if (!S->setup) bk_kmers_ordered(S->cbase, S->clen, BK_ORD_MID);
    for (;;) {
        if (S->status) return;
        // refresh_kmers :601-602 -> snapshot list
        const int nk = S->nk, k0 = S->kscan;
        uint32_t T;
        // Every k-mer of a snapshot is in checked_kmers when its visits are over, and the list only grows at its end (a
        // replacement, set_kmers, starts it anew): the next snapshot can only hold what was appended since.  With sequencing
        // noise that is a handful of k-mers after every extension, a snapshot per round -- one wavefront takes them, the
        // candidate-less ones marked in the same pass (below), one barrier instead of seven.
        const bool small = nk - k0 <= 64;
        if (small) {
            if ((BK_TID >> 6) == 0) {
                const int idx = k0 + BK_TID;
                uint32_t en = idx < nk ? C_.klist[idx] : 0u;
                const bool unchecked = idx < nk && C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial;
                const unsigned long long bm = __ballot(unchecked);
                if (unchecked) {
                    const int rank = (int)(en & 0x3FFFFFFFu);
                    const uint32_t pb = C_.poff[rank], pe = C_.poff[rank + 1];
                    bool has = pe - pb > 16u;
                    for (uint32_t i = pb; !has && i < pe; i++) { const uint32_t u = C_.post[i] >> 10; has = !(C_.ufl[u] & BK_R_DELETED) && C_.ubuf[u] != S->serial; }
                    if (!has) { C_.kstamp[3 * rank] = S->serial; en |= 0x40000000u; }
                    C_.nklist[__popcll(bm & ((1ull << BK_TID) - 1ull))] = en;
                }
                if (BK_TID == 0) {
#ifdef BK_PHASE_STAMPS
                    S->acc[20] += 1; S->acc[22] += (unsigned long long)(nk - k0);
#endif
                    S->tmp0 = __popcll(bm); S->kscan = nk;
                    if (bm) {
                        if (!S->founder_added) { S->founder_added = 1; const int fu = S->founder; if (C_.ureads[fu] != S->serial) { C_.ureads[fu] = S->serial; C_.readl[S->nr++] = (uint32_t)fu; } }
                        if (S->plan_kind == 0) S->plan_ok = 0;
                    }
                }
            }
            BK_SYNC();
            T = (uint32_t)S->tmp0;
            BK_ACC(13);
            if (T == 0) break;
        } else {
        const int chunk = (nk + BK_AT - 1) / BK_AT, b = BK_TID * chunk, e = min(nk, b + chunk);
        uint32_t cnt = 0;
        if (chunk == 1) {                                   // the usual case: an entry per thread, looked at once
            uint32_t en = 0;
            if (b < e) { en = C_.klist[b]; cnt = C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial; }
            const uint32_t pre = bk_scan256(cnt, S->scan, &T);
            if (cnt) C_.nklist[pre] = en;
        } else {
        for (int t = b; t < e; t++) cnt += C_.kstamp[3 * (C_.klist[t] & 0x7FFFFFFFu)] != S->serial;
        uint32_t pre = bk_scan256(cnt, S->scan, &T);
        for (int t = b; t < e; t++) { uint32_t en = C_.klist[t]; if (C_.kstamp[3 * (en & 0x7FFFFFFFu)] != S->serial) C_.nklist[pre++] = en; }
        }
#ifdef BK_PHASE_STAMPS
        if (BK_TID == 0) { S->acc[21] += 1; S->acc[23] += (unsigned long long)nk; }
#endif
        if (BK_TID == 0) S->kscan = nk;
        BK_SYNC();
        BK_ACC(13);
        if (T == 0) break;
        // Visits without any candidate read.  Within one contig the candidate set of a k-mer only shrinks (reads get
        // buffered or deleted, never the reverse), so a k-mer whose short posting list holds no eligible read now has
        // none when its turn comes, and such a visit does nothing but mark the k-mer checked and, AT ITS TURN (the
        // used set is read by check_alt_reads of the visits before it), used (get_mer_reads :604-614 returns []).
        // With sequencing noise that is the majority of all visits: they are found here for the whole snapshot at once
        // (bit 30 of the entry), and the loop below retires whole runs of them with one wavefront.
        // The founder read joins the read list first, as the first finalize would do (:383).
        if (BK_TID == 0 && !S->founder_added) {
            S->founder_added = 1; const int fu = S->founder;
            if (C_.ureads[fu] != S->serial) { C_.ureads[fu] = S->serial; C_.readl[S->nr++] = (uint32_t)fu; }
        }
        for (uint32_t t = BK_TID; t < T; t += BK_AT) {
            const uint32_t en = C_.nklist[t]; int rank = (int)(en & 0x3FFFFFFFu);
            if (!BK_CHK((uint32_t)rank < C_.M, 7, ((unsigned long long)t << 32) | en)) rank = 0;
            const uint32_t pb = C_.poff[rank], pe = C_.poff[rank + 1];
            bool has = pe - pb > 16u;                                             // long lists take the ordinary visit
            for (uint32_t i = pb; !has && i < pe; i++) { const uint32_t u = C_.post[i] >> 10; has = !(C_.ufl[u] & BK_R_DELETED) && C_.ubuf[u] != S->serial; }
            if (!has) { C_.kstamp[3 * rank] = S->serial; C_.nklist[t] = en | 0x40000000u; }       // checked_kmers is only read by the next snapshot
        }
        BK_SYNC();
        if (BK_TID == 0 && S->plan_kind == 0) S->plan_ok = 0;                      // a plan of visits refers to one snapshot (and one contig)
        BK_SYNC();
        }
        uint32_t t = 0, en_next = C_.nklist[0];
        while (t < T) {
            if (S->status) return;
            const uint32_t en = en_next; int rank = (int)(en & 0x3FFFFFFFu); const bool rev = (en >> 31) != 0;
            if (!BK_CHK((uint32_t)rank < C_.M, 6, ((unsigned long long)t << 32) | en)) { S->status = S->status ? S->status : BK_ST_UNSPLIT; BK_SYNC(); return; }
            if (en & 0x40000000u) {
                // a run of candidate-less visits, in order: used_mers.add(mer) for each k-mer not yet in it
                if ((BK_TID >> 6) == 0) {
                    const uint32_t idx = t + BK_TID;
                    const uint32_t e2 = idx < T ? C_.nklist[idx] : 0u;
                    const unsigned long long bm = __ballot((e2 & 0x40000000u) != 0);
                    const int run = ~bm ? __ffsll((long long)~bm) - 1 : 64;                  // entries t .. t+run-1 are candidate-less
                    const bool in_run = BK_TID < run;
                    const int rk = (int)(e2 & 0x3FFFFFFFu);
                    bool ap = in_run && C_.kstate[in_run ? rk : 0] == BK_K_LIVE;
                    for (int j = 0; j < run; j++) { const int rj = __builtin_amdgcn_readlane(rk, j); if (j < BK_TID && rj == rk) ap = false; }   // listed twice: once
                    const unsigned long long am = __ballot(ap);
                    if (ap) { C_.kstate[rk] = BK_K_USED; C_.usedl[S->nused + __popcll(am & ((1ull << BK_TID) - 1ull))] = (uint32_t)rk; }
                    if (BK_TID == 0) { S->nused += __popcll(am); S->tmp1 = run; }
                }
                BK_SYNC();
                t += (uint32_t)S->tmp1;
                if (t < T) en_next = C_.nklist[t];
                BK_SYNC();
                continue;
            }
            const int vt = (int)t;
            t++;
            if (t < T) en_next = C_.nklist[t];                                     // fetched a whole visit ahead of its use
            bk_find_reads(rank, rev, true);                                    // get_mer_reads :604-614
            if (BK_TID == 0) bk_add_used_mer(rank);
            BK_SYNC();
            // position of this k-mer in the contig (prediction seed); -1 disables the look-ahead
            if ((BK_TID >> 6) == 0) {
                int pc = -1;
                // (a single candidate needs no prediction unless the round may go on into the following visits)
                if (!bk_expect_reject() && (S->ncand >= 2 || (S->ncand == 1 && S->la_pause == 0 && !(C_.flags & BK_F_NO_XVISIT)))) { BkKey key; key.hi = C_.khi[rank]; key.lo = C_.klo[rank]; pc = bk_find_kmer_wave(L_CSEQ + S->cbase, S->clen, key, C_.k); }
                if (BK_TID == 0) S->pc = pc;
            }
            BK_SYNC();
            BK_ACC(14);
            bk_run_candidates(rank, 0, S->ncand, true, vt, (int)T);
            bk_finalize(false);
            if (BK_TID == 0) C_.kstamp[3 * rank] = S->serial;                       // checked_kmers.append(mer): read by the next snapshot only
        }
        BK_SYNC();                                                                  // the stamps above are visible before the next snapshot reads them
    }
    BK_ACC(15); BK_CTX(0);
}

// ---- init_assembly keeps a contig iff support >= rc_thresh and len > read_len (sv_assembly.py:53-59);
//      set_kmer_locs (:434-438) and the record the host reads back ------------------------------------------
BK_COLD void bk_emit_contig()
{
    BK_ACC(S_->ctx);
    BkAsmShared *S = S_;
    const int total = bk_total_reads();
    if (total < C_.rc_thresh || S->clen <= (int)C_.max_len) return;
    const int k = C_.k, len = S->clen, nlen = S->nlen, nk = S->nk, nr = S->nr;
    const uint32_t o_seq = (uint32_t)sizeof(BkContigRec), o_io = (uint32_t)bk_align_up(o_seq + len, 8), o_ot = o_io + 4u * nlen,
                   o_kl = o_ot + 4u * nlen, o_km = (uint32_t)bk_align_up(o_kl + 4u * len, 8), o_rd = o_km + 16u * nk, size = (uint32_t)bk_align_up(o_rd + 4u * nr, 8);
    BK_SYNC();
    if (BK_TID == 0) {
        uint64_t need = bk_align_up(size, 256);
        uint64_t off = atomicAdd(C_.out_top, (unsigned long long)need);
        if (off + need > C_.out_cap) { off = 0; S->status = BK_ST_OUT; }
        S->scan[8] = (uint32_t)off; S->scan[9] = (uint32_t)(off >> 32);
    }
    BK_SYNC();
    const uint64_t off = ((uint64_t)S->scan[9] << 32) | S->scan[8];
    if (off == 0) return;                                 // offset 0 is reserved (out_top starts at 256)
    uint8_t *rec = C_.out + off;
    BkContigRec *h = (BkContigRec *)rec;
    char *oseq = (char *)(rec + o_seq); int32_t *oio = (int32_t *)(rec + o_io), *oot = (int32_t *)(rec + o_ot), *okl = (int32_t *)(rec + o_kl);
    uint64_t *okm = (uint64_t *)(rec + o_km); uint32_t *ord_ = (uint32_t *)(rec + o_rd);
    const uint8_t *cs = L_CSEQ + S->cbase;
    const int32_t *io = bk_cnt_io(S->cbuf) + S->nbase, *ot = bk_cnt_ot(S->cbuf) + S->nbase;
    for (int t = BK_TID; t < len; t += BK_AT) { oseq[t] = "ACGTN"[cs[t]]; okl[t] = 0; }
    for (int t = BK_TID; t < nlen; t += BK_AT) { oio[t] = io[t]; oot[t] = ot[t]; }
    for (int t = BK_TID; t < nk; t += BK_AT) { uint32_t rk = C_.klist[t] & 0x7FFFFFFFu; if (!BK_CHK(rk < C_.M, 4, ((unsigned long long)t << 32) | C_.klist[t])) rk = 0; okm[2 * t] = C_.klo[rk]; okm[2 * t + 1] = C_.khi[rk]; }
    for (int t = BK_TID; t < nr; t += BK_AT) ord_[t] = C_.urep[C_.readl[t]];
    // first occurrence of every sample k-mer in the contig (str.find over all len-k+1 positions)
    // (split regions: the stamps of other units' k-mers are left alone -- only the contig's own k-mers are read back below, and
    // those all belong to components this unit holds)
    for (int x = BK_TID; x + k <= len; x += BK_AT) { BkKey key; int rk = bk_bytes_kmer(cs + x, k, key) ? bk_lookup(key) : -1; if (rk >= 0 && (!C_.own || (C_.kroot[rk] != BK_EMPTY32 && bk_acc_mine(C_.kroot[rk])))) atomicMin(&C_.kstamp[3 * rk + 2], x); }
    BK_SYNC();
    for (int t = BK_TID; t < nk; t += BK_AT) {
        if (!BK_CHK((C_.klist[t] & 0x7FFFFFFFu) < C_.M, 5, C_.klist[t])) continue;
        int pos = C_.kstamp[3 * (C_.klist[t] & 0x7FFFFFFFu) + 2];
        if (pos == 0x7FFFFFFF) continue;                 // find() == -1: the python slice [-1:k-1] is empty for len >= k
        for (int q = pos; q < min(pos + k, len); q++) atomicAdd(&okl[q], 1);
    }
    BK_SYNC();
    for (int x = BK_TID; x + k <= len; x += BK_AT) { BkKey key; int rk = bk_bytes_kmer(cs + x, k, key) ? bk_lookup(key) : -1; if (rk >= 0 && (!C_.own || (C_.kroot[rk] != BK_EMPTY32 && bk_acc_mine(C_.kroot[rk])))) C_.kstamp[3 * rk + 2] = 0x7FFFFFFF; }
    if (BK_TID == 0) {
        h->root = S->ccomp; h->pass = C_.want;
        h->next = 0; h->hits_off = 0; h->seq_len = len; h->counts_len = nlen; h->n_kmers = nk; h->n_reads = nr; h->total_reads = total; h->n_hits = 0;
        h->o_seq = o_seq; h->o_io = o_io; h->o_ot = o_ot; h->o_klocs = o_kl; h->o_kmers = o_km; h->o_reads = o_rd; h->n_sec = 0; h->size = size;
        if (C_.split) {       // the units of a region emit side by side: (order key, record) pairs, ordered and linked by bk_link_kernel
            const uint32_t at = atomicAdd(&C_.wk->n_cidx, 1u);
            if (at < C_.wk->cidx_cap) { C_.cidx_key[at] = ((unsigned long long)(uint32_t)S->seed_rank << 20) | (unsigned long long)(uint32_t)min(S->emit_seq, 0xFFFFF); C_.cidx_key[C_.wk->cidx_cap + at] = off; }
            S->emit_seq++;
        } else {
            if (C_.wk->o_first_contig == 0) C_.wk->o_first_contig = off; else ((BkContigRec *)(C_.out + C_.wk->o_last_contig))->next = off;
            C_.wk->o_last_contig = off;
        }
        S->n_contigs++;
        const unsigned long long ci = atomicAdd(C_.n_clist, 1ull);              // work list of the realign stage (one workgroup per contig)
        if (ci < C_.clist_cap) C_.clist[ci] = off | ((unsigned long long)C_.region << 40);
    }
    BK_SYNC();
    BK_ACC(7);
}

// ---- setup_contigs (sv_assembly.py:11-26) -----------------------------------------------------------------
__device__ inline void bk_setup_contigs(int rank)
{
    BK_ACC(S_->ctx); BK_CTX(8);
    BkAsmShared *S = S_;
    bk_find_reads(rank, false, false);                                         // used_reads = set()
    if (BK_TID == 0) bk_add_used_mer(rank);
    BK_SYNC();
    const int n = S->ncand;
    if (n == 0 || S->status) return;
    // the candidate list must survive the check_read calls below: candu is not touched by them
    const int u0 = (int)(L_CANDU[0] & 0x3FFFFFu);
    const bool in_fifo = C_.ufound[u0] < 0 && !(C_.ufl[u0] & BK_R_USED);            // buff.add_contig :337-340
    BK_SYNC();
    bk_contig_new(rank, u0, in_fifo);
    if (BK_TID == 0 && in_fifo) C_.ufl[u0] |= BK_R_USED;
    BK_SYNC();
    if (BK_TID == 0) S->pc = (int)(L_CANDU[0] >> 22);      // the contig IS the first read: the k-mer sits where it sits in that read
    BK_SYNC();
    bk_run_candidates(rank, 1, n, false, -1, 0);
    bk_finalize(true);
    if (in_fifo) { bk_grow(); if (!S->status) bk_emit_contig(); }                // it is the FIFO head (:50-52)
    BK_ACC(8); BK_CTX(0);
}

// The seed ranks of this unit's components, ascending: the seed scan and the look-ahead into the next seeds walk this list
// (a sixteenth of the ranks; looking each rank's owner up costs two dependent loads, and the owner words must be read past
// the caches once other units change them).  fresh: read the owner words with device-scope loads (after this unit lost a
// component to unit 0; otherwise the words are as the labelling / the resolve kernel wrote them).  A wavefront per block of
// ranks: count, prefix over the wavefronts, write in place.
BK_COLD void bk_build_myseeds(int fresh)
{
    BkAsmShared *S = S_;
    const int wv = BK_TID >> 6, lane = BK_TID & 63, M2 = (int)C_.M2;
    const int B = (((M2 + BK_WAVES - 1) / BK_WAVES) + 63) / 64 * 64, lo = wv * B, hi = min(M2, lo + B);
    auto mine = [&](int j) -> bool {
        if (j >= hi || C_.kstate[j] == BK_K_REMOVED) return false;
        const uint32_t root = C_.kroot[j];
        if (root == BK_EMPTY32) return false;
        const uint32_t ci = fresh ? __hip_atomic_load(&C_.cinfo[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : C_.cinfo[root];
        return (ci & (0xFFFFu | BK_CI_ABORT)) == C_.want;
    };
    int cnt = 0;
    for (int b = lo; b < hi; b += 64) cnt += __popcll(__ballot(mine(b + lane)));
    BK_SYNC();
    if (lane == 0) S->scan[wv] = (uint32_t)cnt;
    BK_SYNC();
    int base = 0, tot = 0;
    for (int i = 0; i < BK_WAVES; i++) { const int t = (int)S->scan[i]; if (i < wv) base += t; tot += t; }
    BK_SYNC();
    int off = base;
    const int end = base + cnt;
    for (int b = lo; b < hi; b += 64) {
        const bool m = mine(b + lane);
        const unsigned long long bal = __ballot(m);
        const int at = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (m && at < end) C_.myseeds[at] = b + lane;
        off += __popcll(bal);
    }
    for (int i = min(off, end) + lane; i < end; i += 64) C_.myseeds[i] = -1;          // (owner words changed between the two sweeps: fewer than counted)
    __threadfence_block();
    BK_SYNC();
    if (BK_TID == 0) { C_.n_my = tot; S->head = 0; }
    BK_SYNC();
}

// unit 0, after the serial prefix (the seeds with a count >= BK_SPLIT_HI, run alone and in order): the connected components of what
// is LEFT of the read / k-mer graph -- the k-mers that are still live -- dealt to the units (bk_comp.hip.h).  The read flags of
// this moment are kept: a component that runs again starts from here.
BK_COLD void bk_label_live()
{
    BkAsmShared *S = S_;
    uint32_t *rroot = (uint32_t *)((uint8_t *)C_.cinfo - bk_align_up((uint64_t)C_.U * 4, 256));
    uint32_t *kroot = const_cast<uint32_t *>(C_.kroot), *cinfo = C_.cinfo, *csz = C_.readl;      // readl: U + 1 words, free between two contigs
    uint8_t *ufl0 = BK_UFL0;
    const uint32_t U = C_.U, M = C_.M, M2 = C_.M2;
    BK_SYNC();
    if (BK_TID == 0) { const int now = (int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull); C_.wk->dbg_us[0] = (uint32_t)((now - S->t0) & 0x7FFFFFFF) / 100u; S->t0 = now; }
    for (uint32_t u = BK_TID; u < U; u += BK_AT) { rroot[u] = u; csz[u] = 0; ufl0[u] = C_.ufl[u]; }
    __threadfence(); BK_SYNC();
    for (uint32_t j = BK_TID; j < M; j += BK_AT) {
        if (C_.kstate[j] != BK_K_LIVE) continue;
        const uint32_t b = C_.poff[j], e = C_.poff[j + 1];
        uint32_t first = BK_EMPTY32, last = BK_EMPTY32;
        for (uint32_t i = b; i < e; i++) {
            const uint32_t v = C_.post[i] >> 10;
            if (v == last) continue;                 // (deleted reads stay nodes: every live k-mer then has a component, and only one unit ever touches its state)
            if (first == BK_EMPTY32) first = v; else bk_uf_union(rroot, first, v);
            last = v;
        }
    }
    __threadfence(); BK_SYNC();
    for (uint32_t u = BK_TID; u < U; u += BK_AT) atomicMin(&rroot[u], bk_uf_find(rroot, u));
    __threadfence(); BK_SYNC();
    uint32_t seeds = 0;
    for (uint32_t j = BK_TID; j < M; j += BK_AT) {
        uint32_t r = BK_EMPTY32;
        if (C_.kstate[j] == BK_K_LIVE && C_.poff[j + 1] > C_.poff[j]) r = bk_ld_agent(&rroot[C_.post[C_.poff[j]] >> 10]);
        kroot[j] = r;
        if (j < M2 && r != BK_EMPTY32 && C_.kcnt[j] >= 2) { atomicAdd(&csz[r], 1u); seeds++; }
    }
    uint32_t total;
    (void)bk_scan256(seeds, S->scan, &total);
    __threadfence(); BK_SYNC();
    int big = 0;
    for (uint32_t u = BK_TID; u < U; u += BK_AT) big = max(big, (int)bk_ld_agent(&csz[u]));
    big = bk_max256(big, S->scan);
    // one component with most of the seeds (the graph has percolated: 1 % noise and beyond): everything stays with unit 0
    const bool deal = 10ull * (unsigned long long)big <= 7ull * total || (C_.flags & BK_F_SPLIT_ALWAYS);
    for (uint32_t u = BK_TID; u < U; u += BK_AT) {
        uint32_t ci = BK_CI_NOUNIT;
        if (bk_ld_agent(&rroot[u]) == u && bk_ld_agent(&csz[u])) ci = (deal ? (uint32_t)(mix64(0x9E3779B97F4A7C15ull ^ u) % (uint32_t)C_.split) : 0u) | BK_CI_ACTIVE;
        cinfo[u] = ci;
    }
    __threadfence(); BK_SYNC();
    // Dealt by size, largest first to the unit with the least so far (a unit's time follows its seed k-mers; by a hash of the root
    // the fullest unit had 1.8x the mean).  The components with seeds are listed and ordered in the candidate scratch (LDS);
    // more of them than fit there keep the hash.
    if (deal) {
        unsigned long long *L = L_CAND;
        if (BK_TID == 0) S->tmp0 = 0;
        BK_SYNC();
        for (uint32_t u = BK_TID; u < U; u += BK_AT) {
            const uint32_t c = bk_ld_agent(&csz[u]);
            if (bk_ld_agent(&rroot[u]) != u || !c) continue;
            const int at = atomicAdd(&S->tmp0, 1);
            if (at < (int)C_.MAXCAND) L[at] = ((unsigned long long)(0xFFFFFFFFu - c) << 32) | u;          // ascending key = size descending, then root ascending: deterministic
        }
        BK_SYNC();
        const int n = S->tmp0;
        if (n <= (int)C_.MAXCAND) {
            int npad = 1; while (npad < n) npad <<= 1;
            for (int i = n + BK_TID; i < npad; i += BK_AT) L[i] = ~0ull;
            BK_SYNC();
            for (int sz = 2; sz <= npad; sz <<= 1)
                for (int st = sz >> 1; st > 0; st >>= 1) {
                    for (int i = BK_TID; i < npad / 2; i += BK_AT) {
                        const int lo = (i / st) * (st * 2) + (i % st), hi = lo + st;
                        const bool up = ((lo & sz) == 0);
                        const unsigned long long a = L[lo], bb = L[hi];
                        if ((a > bb) == up) { L[lo] = bb; L[hi] = a; }
                    }
                    BK_SYNC();
                }
            if (BK_TID == 0) {
                // (the units' loads in LDS -- the candidate list is free between two seeds.  NOT a local array: indexed at run time it
                //  lives in scratch memory, and that made this out-of-line function fault at random, 3 runs in 20 of a 32-region
                //  batch -- the second lesson of this kind after the out-of-line return values of round 2)
                uint32_t *load = L_CANDU;
                const int G_ = (int)C_.split;                                                      // units of this region (bk_sched_kernel: 2 .. BK_SPLIT_G)
                for (int g = 0; g < G_; g++) load[g] = 0;
                for (int i = 0; i < n; i++) {
                    const uint32_t root = (uint32_t)L[i], c = 0xFFFFFFFFu - (uint32_t)(L[i] >> 32);
                    int best = 0;
                    for (int g = 1; g < G_; g++) if (load[g] < load[best]) best = g;
                    load[best] += c + 4u;                                                             // (+ what an iteration costs whatever its size)
                    cinfo[root] = (uint32_t)best | BK_CI_ACTIVE;
                }
            }
            __threadfence(); BK_SYNC();
        }
    }
    if (BK_TID == 0) {
        C_.wk->serial_base = (uint32_t)S->serial_ctr; C_.wk->stamp_base = (uint32_t)S->stamp_ctr;
        __threadfence();
        __hip_atomic_store(&C_.wk->phase, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        C_.own = 1;
        { const int now = (int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull); C_.wk->dbg_us[1] = (uint32_t)((now - S->t0) & 0x7FFFFFFF) / 100u; S->t0 = now; }
    }
    BK_SYNC();
    bk_build_myseeds(0);
    if (BK_TID == 0) { const int now = (int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull); C_.wk->dbg_us[2] = (uint32_t)((now - S->t0) & 0x7FFFFFFF) / 100u; S->t0 = now; }
}

__device__ __forceinline__ void bk_asm_region(const BkParams &p, const int r, const int unit)
{
    BkAsmShared *S = S_;
    BkRegionWork *wk = &p.work[r];
#ifdef BK_DIAG
    if (p.poison) {      // diagnostic: everything but the queue slot word of the shared state
        const int keep = S->qslot;
        BK_SYNC();
        const uint32_t v = (p.poison & 0xFFu) * 0x01010101u;
        for (uint32_t i = BK_TID; i < p.asm_lds_bytes / 4; i += BK_AT) ((uint32_t *)bk_lds)[i] = v;
        BK_SYNC();
        if (BK_TID == 0) S->qslot = keep;
        BK_SYNC();
    }
    if (p.asm_lds_pad) {      // diagnostic: the guard band behind the block
        BK_SYNC();
        for (uint32_t i = BK_TID; i < p.asm_lds_pad / 4; i += BK_AT) ((uint32_t *)(bk_lds + p.asm_lds_bytes - p.asm_lds_pad))[i] = 0xA5A5A5A5u;
        BK_SYNC();
    }
#endif
    if (__hip_atomic_load(&wk->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != BK_ST_OK && !wk->split) return;              // k-mer stage failed for this region (a split region: another unit may have failed meanwhile; this one still reports in below)
    if (BK_TID == 0) {
        const BkRegionDesc d = p.desc[r];
        BkAsmCtx &c = C_;
        c.wk = wk; c.out = p.out; c.out_top = p.out_top; c.out_cap = p.out_cap; c.rc_thresh = p.rc_thresh;
        c.n_clist = p.n_clist; c.clist = p.clist; c.clist_cap = p.clist_cap; c.region = r;
        c.read_words = d.read_words; c.max_len = d.max_len;
        c.flags = p.flags; c.MAXC = p.max_contig; c.MAXR = (uint16_t)p.max_read; c.MAXCAND = (uint16_t)p.max_cand; c.k = p.k; c.myseeds = nullptr; c.n_my = (int)wk->M2;
        int o = BK_BUF_OFF;
        // 40 KB at the 256-thread size (four workgroups per CU).  The DP tile-boundary scratch directly behind the contig deque
        // doubles as (a) the [read | founder | read] strips of look-ahead seeds, which only live through the DPs of the round
        // that plans them, all of them single-tile (bk_plan_round), and (b) the generic read buffer of bk_load_read, which
        // is never used while a DP runs.
        // (every block is a multiple of 16 bytes: the caps are multiples of 8, the read buffers of 16)
        c.o_cand16 = (uint16_t)(o >> 4); o += (c.MAXCAND * 8 + 15) & ~15;
        c.o_candu16 = (uint16_t)(o >> 4); o += (c.MAXCAND * 4 + 15) & ~15;
        c.o_cseq16 = (uint16_t)(o >> 4); o += (2 * c.MAXC + 15) & ~15;
        c.o_bound16 = (uint16_t)(o >> 4); o += ((BK_AT / 64) * 2 * (c.MAXR + 2) * 4 + 15) & ~15;
        c.o_rseq16 = (uint16_t)(o >> 4); o += BK_SPEC * (c.MAXR + 16);
        c.reads = p.reads + d.reads_word_off; c.rlen = p.read_len + d.read_meta_off;
        c.nlist = p.nlist + d.nlist_off; c.n_nlist = d.n_nlist;
        const uint64_t mo = d.read_meta_off;
        c.urep = p.urep + mo; c.unr = p.unreads + mo; c.ufl = p.uflag + mo; c.ubuf = p.ubuf + mo; c.ureads = p.ureads + mo; c.ufound = p.ufound + mo; c.uminpos = p.uminpos + mo; c.ulen = p.dd_rep + d.dedup_off;
        c.U = wk->U; c.M = wk->M; c.tmask = wk->tcap - 1; c.M2 = wk->M2;
        c.split = (uint8_t)wk->split; c.unit = (uint8_t)unit; c.pass = (uint8_t)wk->pass; c.want = (uint32_t)unit | ((uint32_t)wk->pass << 8);
        c.own = (wk->split && !(wk->pass == 0 && unit == 0 && wk->phase == 0)) ? 1 : 0;      // unit 0 of the first pass starts with the serial prefix
        if (c.split) {
            c.kroot = (const uint32_t *)(p.arena + wk->o_kroot); c.cinfo = (uint32_t *)(p.arena + wk->o_cinfo);
            c.cidx_key = (unsigned long long *)(p.arena + wk->o_cidx); c.pairs = (uint32_t *)(p.arena + wk->o_pairs);
        }
        S->ccomp = 0; S->acc_n = 0; S->dirty = 0; S->foreign = 0; S->foreign_root = BK_EMPTY32; S->seed_rank = 0; S->emit_seq = 0; S->t0 = (int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull);
        c.tslot = (const uint32_t *)(p.arena + wk->o_tslot);
        c.klo = (const uint64_t *)(p.arena + wk->o_key_lo); c.khi = (const uint64_t *)(p.arena + wk->o_key_hi);
        c.kcnt = (const uint32_t *)(p.arena + wk->o_kcnt); c.kstate = p.arena + wk->o_kstate; c.kstamp = (int32_t *)(p.arena + wk->o_kstamp);
        c.poff = (const uint32_t *)(p.arena + wk->o_poff); c.post = (const uint32_t *)(p.arena + wk->o_post);
        S->status = 0; S->serial_ctr = 0; S->stamp_ctr = 0; S->head = 0; S->nused = 0; S->phead = 0; S->ptail = 0; S->n_contigs = 0; S->cells = 0; S->calls = 0;
        S->nalt = 0; S->nk = 0; S->nr = 0; S->ncand = 0; S->plan_ok = 0; S->plan_kind = 0; S->la_planned = 0; S->la_adopted = 0; S->la_pause = 0; S->la_backoff = 32; S->n_rej = 0; S->n_acc = 0; S->dp_n = 0; S->dp_redo = 0; S->fast = 0; S->dp_tot = 0; S->dp_rtot = 0;
#ifdef BK_PHASE_STAMPS
        for (int i = 0; i < 24; i++) S->acc[i] = 0; S->ctx = 0;
        S->last = __builtin_amdgcn_s_memrealtime();
#endif
        // per-region scratch from the arena
        const uint64_t b_cnt = (uint64_t)8 * c.MAXC * 4, b_kl = (uint64_t)2 * c.MAXC * 4, b_pend = (uint64_t)2 * (c.U + 1) * 4, b_alt = (uint64_t)c.MAXCAND * 4,
                       b_rd = (uint64_t)(c.U + 1) * 4, b_used = (uint64_t)(c.M + 1) * 4;
        uint64_t need = bk_align_up(b_cnt + 2 * b_kl + b_pend + b_alt + b_rd + b_used + 64 + BK_ACC_MAX * 4 + (wk->split ? ((uint64_t)wk->M2 + 64) * 4 : 0), 256);
        uint64_t off = atomicAdd(p.arena_top, (unsigned long long)need);
        if (off + need > p.arena_cap) S->status = BK_ST_ARENA;
        else {
            uint8_t *sp = p.arena + off;
            c.cnt = (int32_t *)sp; sp += b_cnt; c.klist = (uint32_t *)sp; sp += b_kl; c.nklist = (uint32_t *)sp; sp += b_kl;
            c.pend = (uint32_t *)sp; sp += b_pend; c.altl = (uint32_t *)sp; sp += b_alt; c.readl = (uint32_t *)sp; sp += b_rd; c.usedl = (uint32_t *)sp; sp += b_used; c.acc_root = (uint32_t *)sp;
            c.myseeds = (int32_t *)(sp + BK_ACC_MAX * 4);
        }
    }
    BK_SYNC();
    if (C_.split && C_.own) {
        // the other units start once unit 0 has run the serial prefix and labelled the graph: since round 6 it appends their queue
        // entries at that moment, so the word is set when they get here; with BK_F_PREQUEUE_UNITS (the round-5 queue) they wait for it --
        // unit 0 holds a workgroup by then, its queue entry was handed out before theirs.  Contig serials and stamps go on beyond unit
        // 0's, so nothing it left behind looks current
        if (BK_TID == 0) {
            int ok = 0;
            for (int spin = 0; spin < 4000000; spin++) {
                if (__hip_atomic_load(&wk->phase, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) { ok = 1; break; }
                if (__hip_atomic_load(&wk->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != BK_ST_OK) break;
#ifdef BK_SYNC_CHECK
                if (__hip_atomic_load(&bk_sync_report[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;      // (unit 0 diverged and ended: bk_common.h)
#endif
                __builtin_amdgcn_s_sleep(100);
            }
            if (!ok && !S->status) S->status = BK_ST_UNSPLIT;
            S->serial_ctr = (int)wk->serial_base; S->stamp_ctr = (int)wk->stamp_base;
        }
        BK_SYNC();
        if (BK_TID == 0) S->t0 = (int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull);
        if (!S->status) bk_build_myseeds(0);
    }
    if (C_.M == 0 && !C_.split) { if (BK_TID == 0) wk->n_contigs = 0; return; }                 // init_assembly :33-34
    if (S->status && !C_.split) { if (BK_TID == 0) wk->status = S->status; return; }
    // ---- init_assembly main loop (:43-62) --------------------------------------------------------------
#ifdef BK_DIAG
    uint32_t iters_done = 0;
#endif
    while (!S->status) {
        // first k-mer still in akmers.mers in (count, mer) descending order; has_mers (:318-322) <=> its count > 1
        // (the seed-capable k-mers are ranks 0 .. M2-1; a unit of a split region takes those of its own components)
        int head = S->head, found = -1, fidx = -1;
        while (head < C_.n_my) {
            const int ci_ = head + BK_TID;
            const int cand_rk = ci_ < C_.n_my ? bk_seed_at(ci_) : -1;
            int mine = (cand_rk >= 0 && C_.kstate[cand_rk] != BK_K_REMOVED) ? ci_ : 0x7FFFFFFF;
            int mn = -bk_max256(-mine, S->scan);
            if (mn != 0x7FFFFFFF) { fidx = mn; found = bk_seed_at(mn); break; }
            head += BK_AT;
        }
        if (C_.split && !C_.own && (found < 0 || C_.kcnt[found] < BK_SPLIT_HI)) {      // the serial prefix is over: label what is left, the other units start
            bk_label_live();
            if (!(p.flags & BK_F_PREQUEUE_UNITS)) {
                // ... as queue entries this unit appends now (round 6; bk_sched.hip.h): nobody held a workgroup slot waiting for this
                // prefix.  No room (cannot happen while the host sizes `order` for it): the region reports in for the units that never
                // started and is run again as one unit by the host.
                if (BK_TID == 0) {
                    const unsigned long long cap_ = __hip_atomic_load(p.queue_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), need = (unsigned long long)C_.split - 1ull;
                    int ok = 0;
                    for (;;) {
                        const unsigned long long old = __hip_atomic_load(p.n_queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old + need > cap_) break;
                        if (atomicCAS(p.n_queue, old, old + need) == old) {
                            for (unsigned long long i = 0; i < need; i++) __hip_atomic_store(&p.order[old + i], (uint32_t)r | ((uint32_t)(i + 1ull) << BK_QUEUE_UNIT_SHIFT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            ok = 1; break;
                        }
                    }
                    if (!ok) { __threadfence(); atomicAdd(&wk->units_done, (uint32_t)need); }
                    S->tmp0 = ok;
                }
                BK_SYNC();
                const int ok = S->tmp0;
                BK_SYNC();
                if (!ok) bk_fail(BK_ST_UNSPLIT);
            }
            continue;
        }
        if (found >= 0 && !BK_CHK((uint32_t)found < C_.M, 9, found)) break;
        if (found < 0 || C_.kcnt[found] < 2) break;
#ifdef BK_DIAG
        if (p.dbg_iters && iters_done++ >= p.dbg_iters) break;      // diagnostic build (uniform): stop after BK_DBG_ITERS seed iterations
#endif
        BK_SYNC();
        if (BK_TID == 0) {
            S->head = fidx; S->seed_rank = found; S->emit_seq = 0; S->acc_n = 0; S->dirty = 0; S->ccomp = C_.own ? C_.kroot[found] : BK_EMPTY32;
            if (C_.split) C_.wk->unit_iters[C_.unit]++;
        }
        BK_SYNC();
        BK_ACC(11);
        bk_setup_contigs(found);
        while (!S->status && S->phead < S->ptail) {                                // :50-59
            const int ph = S->phead;
            uint32_t prk = C_.pend[2 * ph], pu = C_.pend[2 * ph + 1];
            if (prk != BK_EMPTY32 && !(BK_CHK(prk < C_.M, 12, prk) && BK_CHK(pu < C_.U, 13, pu))) prk = BK_EMPTY32;
            BK_SYNC();
            if (BK_TID == 0) { S->phead = ph + 1; if (prk != BK_EMPTY32) C_.ufound[pu] = -1; }
            BK_SYNC();
            if (prk == BK_EMPTY32) continue;
            bk_contig_new((int)prk, (int)pu, true);
            bk_grow();
            if (!S->status) bk_emit_contig();
        }
        if (S->status == BK_ST_CONFLICT || (S->dirty && !S->status)) {
            // The iteration met a component of another unit: nothing of that unit's state was touched.  The seed's component and
            // what its contigs had taken in are given up for this pass (bk_resolve_kernel merges them with what they met and
            // they run again); the unit goes on with its other components.
            BK_SYNC();
            if (BK_TID == 0) {
                if (BK_CHK(S->ccomp < C_.U, 2, S->ccomp)) atomicOr(&C_.cinfo[S->ccomp], BK_CI_ABORT);
                for (uint32_t i = 0; i < S->acc_n; i++) if (!(C_.acc_root[i] & 0x80000000u) && BK_CHK(C_.acc_root[i] < C_.U, 3, C_.acc_root[i])) atomicOr(&C_.cinfo[C_.acc_root[i]], BK_CI_ABORT);      // (not the other units' components it met)
                S->status = 0; S->dirty = 0; S->phead = S->ptail; S->nused = 0; S->nalt = 0; S->plan_ok = 0;
            }
            BK_SYNC();
            bk_build_myseeds(1);                                 // (what was given up leaves the unit's seed list)
            continue;
        }
        if (S->status) break;
        BK_SYNC();
        const int nu = S->nused;
        for (int i = BK_TID; i < nu; i += BK_AT) C_.kstate[C_.usedl[i]] = BK_K_REMOVED;   // buff.remove_kmers :358-360
        BK_SYNC();
        BK_ACC(12);
        if (BK_TID == 0) S->nused = 0;
        BK_SYNC();
    }
    BK_SYNC();
    BK_ACC(S_->ctx);
    if (BK_TID == 0) {
        int action = 0;                                  // split regions, the last unit to report in: 1 merge + re-queue, 2 link, 3 leave it to the host
        atomicAdd((unsigned long long *)&C_.wk->dp_sweeps, (unsigned long long)(S->dp_tot + S->dp_n)); atomicAdd((unsigned long long *)&C_.wk->dp_redos, (unsigned long long)(S->dp_rtot + S->dp_redo));
        if (!C_.split) { C_.wk->n_contigs = (uint32_t)S->n_contigs; C_.wk->nw_cells = S->cells; C_.wk->nw_calls = S->calls; if (S->status) C_.wk->status = S->status; }
        else {
            // a unit reports in; the last one of the region decides what the host sees: a failed unit fails the region (the library
            // runs it again as one unit under larger caps), components that met across units mean another pass
            atomicAdd((unsigned long long *)&C_.wk->nw_cells, S->cells); atomicAdd((unsigned long long *)&C_.wk->nw_calls, S->calls);
            C_.wk->unit_us[C_.unit] = (uint32_t)((((int)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFFull)) - S->t0) & 0x7FFFFFFF) / 100u;
            if (S->status) atomicCAS((int *)&C_.wk->status, BK_ST_OK, S->status);
            __threadfence();
            // Unit 0 leaves before it has labelled the graph (the scratch arena was exhausted -- the ordinary first launch of a noisy
            // batch: the host grows the arena and runs the batch again --, or a cap overflowed in the prefix): the other units were
            // never appended to the queue, so it reports in for them too.  Without this the region never settles, *pending never
            // reaches 0 and every workgroup of the launch waits for ever.
            if (!C_.own && !(p.flags & BK_F_PREQUEUE_UNITS)) atomicAdd(&C_.wk->units_done, (uint32_t)C_.split - 1u);
            const uint32_t done = atomicAdd(&C_.wk->units_done, 1u) + 1u;
            if (done == (uint32_t)C_.split) {
                // The last unit of the region to report in settles it, here and now (round 5; until round 4 the host did, after the
                // LAST region of the batch had finished its pass): nothing met across units -> the contigs are put into the
                // reference's order (bk_link_region); components met -> they are merged, reset and dealt again (bk_resolve_region)
                // and the units of the next pass appended to the queue, where the workgroups that have run out of work pick them up
                // while the stragglers of the batch still run.  No room in the queue / bookkeeping overflow / a failed unit: the
                // host takes over as before (BK_ST_REDO / BK_ST_UNSPLIT / the unit's status).
                __threadfence();
                const uint32_t np_ = __hip_atomic_load(&C_.wk->n_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nc_ = __hip_atomic_load(&C_.wk->n_conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               nx_ = __hip_atomic_load(&C_.wk->n_cidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int st_ = __hip_atomic_load(&C_.wk->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st_ != BK_ST_OK) action = 3;
                else if (np_ > C_.wk->pairs_cap || nx_ > C_.wk->cidx_cap) { atomicCAS((int *)&C_.wk->status, BK_ST_OK, BK_ST_UNSPLIT); action = 3; }
                else if (nc_ > 0) {
                    const unsigned long long cap_ = __hip_atomic_load(p.queue_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        const unsigned long long old = __hip_atomic_load(p.n_queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old + (unsigned long long)C_.split > cap_) { atomicCAS((int *)&C_.wk->status, BK_ST_OK, BK_ST_REDO); action = 3; break; }
                        if (atomicCAS(p.n_queue, old, old + (unsigned long long)C_.split) == old) { S->tmp1 = (int)old; action = 1; break; }
                    }
                } else action = 2;
            }
        }
        S->tmp0 = action;
    }
    BK_SYNC();
    {
        const int action = S->tmp0, qat = S->tmp1, nunits = (int)C_.split;
        BK_SYNC();                                       // (every wavefront has read the two words)
        if (action == 1) {
            bk_resolve_region(p, (uint32_t)r, (uint32_t)BK_TID, (uint32_t)BK_AT);
            if (BK_TID < nunits) {
                const bool ok = __hip_atomic_load(&wk->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == BK_ST_OK;          // (too many passes: the host runs the region again as one unit)
                __hip_atomic_store(&p.order[qat + BK_TID], ok ? ((uint32_t)r | ((uint32_t)BK_TID << BK_QUEUE_UNIT_SHIFT)) : BK_QUEUE_NOP, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (!ok && BK_TID == 0) { __threadfence(); atomicSub(p.pending, 1ull); }
            }
        } else if (action == 2) {
            bk_link_region(p, (uint32_t)r, (uint32_t)BK_TID, (uint32_t)BK_AT, S->scan);
            if (BK_TID == 0) { __threadfence(); atomicSub(p.pending, 1ull); }
        } else if (action == 3) {
            if (BK_TID == 0) { __threadfence(); atomicSub(p.pending, 1ull); }
        }
    }
#ifdef BK_DIAG
    if (p.asm_lds_pad) {
        BK_SYNC();
        for (uint32_t i = BK_TID; i < p.asm_lds_pad / 4; i += BK_AT) {
            const uint32_t v = ((uint32_t *)(bk_lds + p.asm_lds_bytes - p.asm_lds_pad))[i];
            if (v != 0xA5A5A5A5u) { unsigned long long *st = (unsigned long long *)wk->stamps; if (atomicCAS(&st[14], 0ull, (unsigned long long)(4 * i + 1)) == 0ull) st[15] = v; }
        }
        BK_SYNC();
    }
#endif
#ifdef BK_PHASE_STAMPS
    if (BK_TID == 0) for (int i = 0; i < 20; i++) { if (C_.split) atomicAdd((unsigned long long *)&C_.wk->stamps[i], (unsigned long long)S->acc[i]); else C_.wk->stamps[i] = S->acc[i]; }      // split regions: summed over the units (and passes)
#ifdef BK_SNAP_COUNT      // one-off: snapshots taken by one wavefront / by the workgroup, entries they looked at (in place of slots / retired / rounds / look-ahead counters)
    if (BK_TID == 0) for (int i = 0; i < 4; i++) C_.wk->stamps[16 + i] = S->acc[20 + i];
#endif
#endif
}

// Persistent workgroups: each pulls the next region of the cost-ordered queue (bk_sched.hip.h) until it is empty, so a
// batch is not bound by whichever heavy region happened to be launched last, and a batch may hold many more regions
// than workgroups fit on the chip.
#ifndef BK_ASM_MINB
#define BK_ASM_MINB 4
#endif

extern "C" __global__ void __launch_bounds__(BK_AT, BK_ASM_MINB) BK_ASM_KERNEL(BkParams p)
{
    bool first = true;
    for (;;) {
        BK_SYNC();                                       // the previous region's LDS state is dead
        // A batch WITHOUT split regions: the first entry a workgroup takes is the one of its own index -- the grid is sized for the most
        // units the batch can have (a noisy region is split into up to BK_SPLIT_G on the device), and with one unit per region the
        // workgroups that find work must be the FIRST ones launched -- one per CU -- not whichever of two on a CU wins a race for the
        // queue head.  A batch WITH split regions has *n_queue0 = 0 (bk_sched.hip.h): every entry is handed out through the head, so no
        // entry is tied to a workgroup that may not be resident (two handles' kernels could otherwise wait for each other).
        // The queue is dynamic: *asm_head is the next entry to hand out, *n_queue the entries allocated (split regions
        // whose components met append the units of their next pass, bk_asm_region), an entry is valid once written; a workgroup
        // that finds the queue empty leaves only when no split region can append any more (*pending == 0: at once for a batch
        // without split regions).
        if (BK_TID == 0) {
            int q = -1; uint32_t e = BK_QUEUE_NOP;
            if (first && (unsigned long long)blockIdx.x < __hip_atomic_load(p.n_queue0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) q = (int)blockIdx.x;
            else for (;;) {
                const unsigned long long n = __hip_atomic_load(p.n_queue, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hd = __hip_atomic_load(p.asm_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hd < n) { if (atomicCAS(p.asm_head, hd, hd + 1ull) == hd) { q = (int)hd; break; } continue; }
                if (__hip_atomic_load(p.pending, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0ull) break;      // (what was appended before the last region settled has been handed out: it could not settle otherwise)
#ifdef BK_SYNC_CHECK
                // barrier-check build: a workgroup that diverged has ended without reporting its unit in (bk_common.h) -- *pending will never
                // reach 0; everybody leaves so that bk_sync can report the divergence instead of hanging
                if (__hip_atomic_load(&bk_sync_report[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
#endif
                __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
            }
            if (q >= 0) {
                while ((e = __hip_atomic_load(&p.order[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == BK_EMPTY32) {      // allocated, not yet written
#ifdef BK_SYNC_CHECK
                    if (__hip_atomic_load(&bk_sync_report[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { e = BK_QUEUE_NOP; break; }
#endif
                    __builtin_amdgcn_s_sleep(20);
                }
            }
            S_->qslot = q; S_->tmp2 = (int)e;
        }
        BK_SYNC();
        first = false;
        const int q = S_->qslot; const uint32_t e = (uint32_t)S_->tmp2;
        if (q < 0) break;
        if (e == BK_QUEUE_NOP) continue;
        bk_asm_region(p, (int)(e & ((1u << BK_QUEUE_UNIT_SHIFT) - 1u)), (int)(e >> BK_QUEUE_UNIT_SHIFT));
    }
}

#ifdef BK_WITH_NW_BATCH
// ---- stand-alone batched olc.nw (known-answer tests G1, DP micro-benchmark) -----------------------------------
extern "C" __global__ void __launch_bounds__(64) bk_nw_batch_kernel(const uint8_t *codes, const uint32_t *off1, const uint32_t *len1,
                                                                     const uint32_t *off2, const uint32_t *len2, int32_t *out, int reps, int transposed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t l[];
    const int b = blockIdx.x, m = (int)len1[b], n = (int)len2[b];
    uint8_t *s1 = l, *s2 = l + ((m + 15) & ~15);
    int *bound = (int *)(l + ((m + 15) & ~15) + ((n + 15) & ~15));   // 2*(max(m,n)+2) ints
    for (int t = threadIdx.x; t < m; t += 64) s1[t] = codes[off1[b] + t];
    for (int t = threadIdx.x; t < n; t += 64) s2[t] = codes[off2[b] + t];
    BK_SYNC();
    if (transposed == 18 || transposed == 19 || transposed == 20) {      // the score sweep for any contig length (column tiles) + the full sweeps for what it flags: 18 -> nw(seq1, seq2), 19 -> nw(seq2, seq1); 20: the sweep alone (timing)
        int *res = bound + 2 * (max(m, n) + 2);          // (8 ints behind the scratch: bk_nw_batch allocates them)
        for (int i = 0; i < reps; i++) bk_nw_score_long((int)(s1 - l), m, (int)(s2 - l), n, (int)((uint8_t *)res - l), bound, transposed == 20 ? 0 : 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        int o[4];
        const int sel = transposed == 19 ? 4 : 0;
        for (int q = 0; q < 4; q++) o[q] = res[sel + q];
        if (transposed != 20 && __builtin_amdgcn_readfirstlane((int)(o[0] == BK_NW_NEEDS_DP))) {
            const BkNwResult r2 = transposed == 18 ? bk_nw_suffix(s1, m, s2, n, bound) : bk_nw_wave<true>(s1, m, s2, n, bound);
            o[0] = r2.j_start; o[1] = r2.i_end; o[2] = r2.i_start; o[3] = r2.score;
        }
        if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = o[q];
        return;
    }
    if (transposed >= 5) {                             // bk_nw_pair: half A = pair b, half B = pair b+1 (cyclic); 5 / 6 -> A's nw(seq1, seq2) / nw(seq2, seq1), 7 / 8 -> B's
        const int b2 = (b + 1) % (int)gridDim.x, m2 = (int)len1[b2], n2 = (int)len2[b2];
        const int mp = (int)((m + 15) & ~15), np_ = (int)((n + 15) & ~15), mp2 = (int)((m2 + 15) & ~15), np2 = (int)((n2 + 15) & ~15);
        uint8_t *t1 = l + mp + np_, *t2 = t1 + mp2;
        int *res = (int *)(l + mp + np_ + mp2 + np2);
        for (int t = threadIdx.x; t < m2; t += 64) t1[t] = codes[off1[b2] + t];
        for (int t = threadIdx.x; t < n2; t += 64) t2[t] = codes[off2[b2] + t];
        BkPairArgs A, B;
        A.contig = 0; A.clen = m; A.read = mp; A.n = n; A.res = (int)((uint8_t *)res - l);
        B.contig = mp + np_; B.clen = m2; B.read = mp + np_ + mp2; B.n = n2; B.res = (int)((uint8_t *)(res + 8) - l);
        BK_SYNC();
        if (transposed >= 9) {
            // the score sweep (bk_nw_score_c).  9..12: two reads per wavefront, every origin wanted, flagged reads swept again by
            // bk_nw_pair -- A's nw(seq1, seq2) / nw(seq2, seq1), B's two; 13 / 14: one read per wavefront + bk_nw_dual;
            // 15: the score sweep ALONE as the assembler calls it (origins only where check_align can look at them): A's
            // (v1.j_start or -1 = needs the full sweep, v1.score, v2.j_start or -1, v2.score); 16 / 17: timing of the sweep alone
            if (m > BK_NW_DUAL_COLS || m2 > BK_NW_DUAL_COLS) { if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = -1; return; }
            const bool one = transposed == 13 || transposed == 14 || transposed == 17, timing = transposed >= 16;
            for (int i = 0; i < reps; i++) { if (one) bk_nw_score_one(A, timing || transposed == 15 ? 0 : 1); else bk_nw_score_pair(A, B, timing || transposed == 15 ? 0 : 1); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (transposed == 15 || timing) { if (threadIdx.x == 0) { out[4 * b] = res[0]; out[4 * b + 1] = res[3]; out[4 * b + 2] = res[4]; out[4 * b + 3] = res[7]; } return; }
            const bool fa = res[0] == BK_NW_NEEDS_DP || res[4] == BK_NW_NEEDS_DP, fb = !one && (res[8] == BK_NW_NEEDS_DP || res[12] == BK_NW_NEEDS_DP);
            if (__builtin_amdgcn_readfirstlane((int)(fa || fb))) {
                if (one) bk_nw_dual(A.contig, m, A.read, n, A.res); else bk_nw_pair(A, B);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            const int sel = transposed <= 12 ? (transposed - 9) * 4 : (transposed - 13) * 4;
            if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = res[sel + q];
            return;
        }
        if (m <= BK_NW_DUAL_COLS && m2 <= BK_NW_DUAL_COLS) {
            for (int i = 0; i < reps; i++) bk_nw_pair(A, B);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = res[(transposed - 5) * 4 + q];
        } else if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = -1;
        return;
    }
    if (transposed >= 3) {                             // both DPs of check_align on one wavefront: 3 -> nw(seq1, seq2), 4 -> nw(seq2, seq1)
        int *res = bound;
        if (m <= BK_NW_DUAL_COLS) {
            for (int i = 0; i < reps; i++) bk_nw_dual((int)(s1 - l), m, (int)(s2 - l), n, (int)((uint8_t *)res - l));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = res[(transposed == 4 ? 4 : 0) + q];
        } else if (threadIdx.x == 0) for (int q = 0; q < 4; q++) out[4 * b + q] = -1;
        return;
    }
    BkNwResult r{};
    for (int i = 0; i < reps; i++) r = transposed == 2 ? bk_nw_suffix(s1, m, s2, n, bound) : transposed ? bk_nw_wave<true>(s2, n, s1, m, bound) : bk_nw_wave<false>(s1, m, s2, n, bound);
    if (threadIdx.x == 0) { out[4 * b] = r.j_start; out[4 * b + 1] = r.i_end; out[4 * b + 2] = r.i_start; out[4 * b + 3] = r.score; }
}
#endif
