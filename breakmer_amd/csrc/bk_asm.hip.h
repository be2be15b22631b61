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
// The state machine is spread over this header and six it includes (round 6; one file of 2,200 lines until then), in dependency order:
//   bk_asm.hip.h         shared state (BkAsmShared, BkAsmCtx), LDS layout, table look-ups, count vectors; bk_asm_region (init_assembly's main
//                        loop, the settling of split regions), the persistent-workgroup kernel, the stand-alone nw batch kernel
//   bk_asm_kmers.hip.h   meetings of components (split regions), contig k-mer lists, find_reads
//   bk_asm_apply.hip.h   contig life cycle, check_align's verdict (bk_decide) and its application for one read (bk_retire)
//   bk_asm_plan.hip.h    the DP dispatch of a round, look-ahead lists of the following visits / seeds, the prediction step
//   bk_asm_round.hip.h   run retire, the plan of a round, the candidate loop with its speculative look-ahead (bk_run_candidates)
//   bk_asm_grow.hip.h    check_alt_reads, finalize, grow, the contig record (emit), setup_contigs
//   bk_asm_units.hip.h   units of a split region: seed lists, unit 0's labelling of the graph

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
    int dp_off;                      // rounds planned since the sweep went off: after BK_SWEEP_RETRY of them the window starts afresh (bk_plan_round)
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

#include "bk_asm_kmers.hip.h"
#include "bk_asm_apply.hip.h"
#include "bk_asm_plan.hip.h"
#include "bk_asm_round.hip.h"
#include "bk_asm_grow.hip.h"
#include "bk_asm_units.hip.h"
#undef BK_SRC_ID
#define BK_SRC_ID 5

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
        S->nalt = 0; S->nk = 0; S->nr = 0; S->ncand = 0; S->plan_ok = 0; S->plan_kind = 0; S->la_planned = 0; S->la_adopted = 0; S->la_pause = 0; S->la_backoff = 32; S->n_rej = 0; S->n_acc = 0; S->dp_n = 0; S->dp_redo = 0; S->fast = 0; S->dp_tot = 0; S->dp_rtot = 0; S->dp_off = 0;
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
