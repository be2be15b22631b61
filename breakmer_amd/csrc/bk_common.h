// bk_common.h -- shared device/host definitions for libbreakmer_hip.so (gfx950 only).
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   * sequences are 2 bit/base, 16 bases per 32-bit word, FIRST base in the MOST significant bits
//     (so that a k-mer read as an integer compares like the string: A<C<G<T), zero padded;
//   * every read of a region starts on its own word boundary, fixed stride `read_words`;
//   * k-mer keys are (hi,lo) 128-bit integers: sum(code[i] << 2*(k-1-i));
//   * N calls (the reference keeps reads with N: utils.py:203-246; Jellyfish skips k-mers that contain one; olc.nw
//     compares characters, so N matches N and nothing else): packed as code 0, and listed per region as sorted
//     (read index << 10 | position) words; reads that have any carry BK_RF_HASN.  Unpacked byte code of N = 4.
//   * N in a reference / partner window: packed as code 0 too, positions listed per window (BkParams.wnlist): no window k-mer
//     spans one (bk_kmer.hip.h: BkRefTabT::nbits), the realigner masks it (bk_sw.hip.h: bk_sw_tn).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BK_WAVE 64
#define BK_EMPTY32 0xFFFFFFFFu
#define BK_EMPTY64 0xFFFFFFFFFFFFFFFFull

// ---- BK_SYNC(): THE workgroup barrier of all device code of this library (no kernel calls __syncthreads() directly) ----------
// Every kernel here runs its control flow redundantly on all threads over state in LDS that one thread commits between barriers.
// The discipline that keeps that correct: a word that steers control flow is read by every wavefront BEFORE a barrier and
// written only AFTER it.  Three defects of that class (a late wavefront reads the new value, takes another branch, and meets
// its workgroup at a different barrier: wild indices, hangs) were found by accident under load in rounds 2-4.  Two diagnostic
// builds make the discipline testable on every kernel:
//   -DBK_SYNC_CHECK   every barrier verifies that ALL wavefronts of the workgroup arrived at the SAME barrier site: the site id
//                     (source file << 16 | line) is published per wavefront, compared after the barrier; the first disagreement
//                     is recorded in bk_sync_report[] (read and reported by bk_sync as BK_E_HIP) and the workgroup ends cleanly
//                     -- a divergence is a deterministic red test instead of a rare fault or hang;
//   -DBK_JITTER       behind every barrier a pseudo-random subset of the wavefronts (hash of site, wavefront, workgroup and
//                     the run-time seed bk_jitter_seed: BK_JITTER_SEED in the environment of a -DBK_DIAG build) sleeps for a
//                     few microseconds, so a wavefront that is "late after the barrier" happens at every site at any load.
// The product build defines neither: BK_SYNC() is __syncthreads().
// Source file ids of the barrier sites (the check build reports "file <id> line <n>"): 0 bk_common.h, 1 bk_kmer.hip.h, 2 bk_sched.hip.h,
// 3 bk_nw.hip.h, 4 bk_comp.hip.h, 5 bk_asm.hip.h, 6 bk_sw.hip.h, 7 bk_asm_kmers, 8 bk_asm_apply, 9 bk_asm_plan, 10 bk_asm_round,
// 11 bk_asm_grow, 12 bk_asm_units (.hip.h).
#ifndef BK_SRC_ID
#define BK_SRC_ID 0
#endif
#if defined(BK_SYNC_CHECK) || defined(BK_JITTER)
__device__ unsigned long long bk_sync_report[4];      // [0] site of the reporting wavefront + 1 (0: none), [1] the other site, [2] workgroup, [3] wavefront
__device__ uint32_t bk_jitter_seed;
__device__ __forceinline__ void bk_sync_diag(const uint32_t site)
{
    const uint32_t wv = threadIdx.x >> 6;
#ifdef BK_SYNC_CHECK
    __shared__ uint32_t bk_sync_site[16];
    if ((threadIdx.x & 63u) == 0) bk_sync_site[wv] = site;
    __syncthreads();
    const uint32_t nw = (blockDim.x + 63u) >> 6;
    uint32_t other = site;
    for (uint32_t w = 0; w < nw; w++) { const uint32_t s = bk_sync_site[w]; if (s != site) other = s; }
    __syncthreads();                                   // every wavefront has compared before the next barrier's sites are written
    if (other != site) {                               // (all wavefronts see the same table: the whole workgroup leaves here)
        if ((threadIdx.x & 63u) == 0 && atomicCAS(&bk_sync_report[0], 0ull, (unsigned long long)site + 1ull) == 0ull) { bk_sync_report[1] = other; bk_sync_report[2] = blockIdx.x; bk_sync_report[3] = wv; }
        __threadfence();
        __builtin_amdgcn_endpgm();
    }
#else
    __syncthreads();
#endif
#ifdef BK_JITTER
    uint32_t x = site * 0x9E3779B1u ^ (wv + 1u) * 0x85EBCA77u ^ bk_jitter_seed * 0xC2B2AE3Du ^ blockIdx.x * 0x27D4EB2Fu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    if ((x & 3u) == 0u) __builtin_amdgcn_s_sleep(100);             // ~64 x 100 cycles: a few microseconds
#endif
}
#define BK_SYNC() bk_sync_diag(((uint32_t)BK_SRC_ID << 16) | (uint32_t)__LINE__)
#else
#define BK_SYNC() __syncthreads()
#endif

// per-region status codes written by the kernels (0 = ok)
enum {
    BK_ST_OK = 0,
    BK_ST_ARENA = 1,          // device arena exhausted (host grows it and reruns)
    BK_ST_WINDOW = 2,         // reference window too long for the LDS k-mer table
    BK_ST_CONTIG = 3,         // contig longer than max_contig_len
    BK_ST_CAND = 4,           // more candidate reads for one k-mer than max_candidates
    BK_ST_KLIST = 5,          // contig k-mer list overflow
    BK_ST_READLEN = 6,        // read longer than max_read_len
    BK_ST_OUT = 7,            // output arena exhausted (host grows it and reruns)
    BK_ST_HITS = 8,           // (ABI 2: realign stage, hit lists full.  No longer produced: the step-1 list is sized by the contig cap, secondary alignments spill to the result arena)
    BK_ST_BLOCKS = 9,         // (ABI 2: a chained record with more than BK_MAX_BLOCKS blocks.  No longer a region status: bk_call has no block limit; bk_get_hits alone returns BK_E_LIMIT for such a contig)
    // internal (never seen by a caller: bk_sync resolves them before it returns)
    BK_ST_REDO = 20,          // split region: components touched each other across units; they are merged and run again (bk_comp.hip.h)
    BK_ST_UNSPLIT = 21,       // split region: bookkeeping of the split overflowed; the region is run again as one unit
    BK_ST_CONFLICT = 22       // assembler-internal: the current seed iteration reached a k-mer of another unit's component
};

#ifndef BK_SPLIT_G
#define BK_SPLIT_G 16                 // units of a split region (bk_comp.hip.h)
#endif
struct BkKey { uint64_t hi, lo; };

__host__ __device__ inline bool key_eq(const BkKey &a, const BkKey &b) { return a.lo == b.lo && a.hi == b.hi; }
__host__ __device__ inline bool key_lt(const BkKey &a, const BkKey &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__host__ __device__ inline uint32_t key_hash(const BkKey &k) { return (uint32_t)(mix64(k.lo ^ (k.hi * 0x9E3779B97F4A7C15ull)) >> 20); }
// append one base (rolling k-mer), k <= 64
__host__ __device__ inline void key_push(BkKey &key, uint32_t c, int k) {
    key.hi = (key.hi << 2) | (key.lo >> 62);
    key.lo = (key.lo << 2) | c;
    if (k <= 32) { key.hi = 0; if (k < 32) key.lo &= ((1ull << (2 * k)) - 1ull); }
    else if (k < 64) key.hi &= ((1ull << (2 * (k - 32))) - 1ull);
}
// base i of a packed sequence
__host__ __device__ inline uint32_t seq_base(const uint32_t *w, int i) { return (w[i >> 4] >> (30 - 2 * (i & 15))) & 3u; }
// k-mer starting at base `pos` of a packed sequence (words may be read up to (pos+k+15)/16)
__host__ __device__ inline BkKey seq_kmer(const uint32_t *w, int pos, int k) {
    BkKey key; key.hi = 0; key.lo = 0;
    int i = pos, end = pos + k;
    // head: unaligned bases until a word boundary
    while (i < end && (i & 15)) { key_push(key, seq_base(w, i), 64); i++; }
    while (i + 16 <= end) { uint32_t x = w[i >> 4]; key.hi = (key.hi << 32) | (key.lo >> 32); key.lo = (key.lo << 32) | x; i += 16; }
    while (i < end) { key_push(key, seq_base(w, i), 64); i++; }
    if (k <= 32) { key.hi = 0; if (k < 32) key.lo &= ((1ull << (2 * k)) - 1ull); }
    else if (k < 64) key.hi &= ((1ull << (2 * (k - 32))) - 1ull);
    return key;
}
// same k-mer, extracted with two funnel shifts instead of a per-base loop; words at index >= nw read as 0
__host__ __device__ inline BkKey seq_kmer_fast(const uint32_t *w, int nw, int pos, int k) {
    const int wi = pos >> 4, sh = 2 * (pos & 15);
    const uint64_t w0 = wi < nw ? w[wi] : 0u, w1 = wi + 1 < nw ? w[wi + 1] : 0u, w2 = wi + 2 < nw ? w[wi + 2] : 0u, w3 = wi + 3 < nw ? w[wi + 3] : 0u;
    const uint64_t a = (w0 << 32) | w1, b = (w2 << 32) | w3;
    const uint64_t X0 = sh ? (a << sh) | (b >> (64 - sh)) : a;           // bases pos .. pos+31
    BkKey key;
    if (k <= 32) { key.hi = 0; key.lo = X0 >> (64 - 2 * k); }
    else {
        const uint64_t c = (wi + 4 < nw ? (uint64_t)w[wi + 4] : 0ull) << 32;
        const uint64_t X1 = sh ? (b << sh) | (c >> (64 - sh)) : b;       // bases pos+32 .. pos+63
        const int r = 2 * (k - 32);
        key.hi = r == 64 ? X0 : X0 >> (64 - r);
        key.lo = r == 64 ? X1 : (X0 << r) | (X1 >> (64 - r));
    }
    return key;
}
__host__ __device__ inline bool key_homopolymer(const BkKey &key, int k) {       // len(set(mer)) == 1  (sv_assembly.py:277)
    uint32_t c = (uint32_t)(key.lo & 3u);
    BkKey h; h.hi = 0; h.lo = 0;
    const uint64_t rep = 0x5555555555555555ull * c;
    if (k <= 32) { h.lo = (k < 32) ? (rep & ((1ull << (2 * k)) - 1ull)) : rep; }
    else { h.lo = rep; h.hi = (k < 64) ? (rep & ((1ull << (2 * (k - 32))) - 1ull)) : rep; }
    return key_eq(h, key);
}

// ---- host-filled region descriptor (one per region, device-resident) ----------------------
struct BkRegionDesc {
    uint64_t reads_word_off;     // into packed reads
    uint64_t read_meta_off;      // base index of this region in all per-read arrays
    uint64_t win_word_off;       // packed forward target window
    uint64_t sc_word_off, sc_meta_off;
    uint64_t dedup_off;          // base slot of this region in the read-grouping table
    uint64_t part_desc_off;      // first partner window descriptor
    uint32_t n_reads, read_words;
    uint32_t win_len;
    int32_t n_sc; uint32_t sc_words;
    uint32_t dedup_cap;          // power of two >= n_reads / 0.7
    uint32_t n_partners;
    uint32_t max_len;            // read_len = max cleaned read length (utils.py:240)
    uint32_t big;                // window does not fit the LDS k-mer set: handled by bk_kmer_kernel_g
    uint32_t n_nlist;            // N calls in this region's reads
    uint64_t nlist_off;          // first entry of this region in BkParams.nlist
    uint32_t n_win_n, pad_;      // N positions in the target window (an assembly gap near the target): sorted, in BkParams.wnlist
    uint64_t win_n_off;
};
struct BkPartnerDesc { uint64_t word_off; uint32_t len, n_n; uint64_t n_off; };      // n_n / n_off: its N positions in BkParams.wnlist

// ---- device-written per-region work record (k-mer stage -> assembler -> realign) -----------
struct BkRegionWork {
    int32_t status;              // BK_ST_*
    uint32_t U;                  // unique read sequences (fq_recs keys)
    uint32_t T;                  // non-reference k-mer occurrences in unique reads
    uint32_t M;                  // sample-only k-mers (incl. homopolymers, which start REMOVED)
    uint32_t M2;                 // ... of which can seed a contig (count >= 2): ranks 0 .. M2-1, ordered by (count, mer) descending
    uint32_t tcap;               // k-mer table capacity (power of two)
    uint32_t n_contigs;
    uint64_t nw_cells, nw_calls; // algorithmic DP work
    uint64_t dp_sweeps, dp_redos;   // reads whose overlap DPs went through the score sweep / of those, swept again in full (bk_nw.hip.h; bk_get_stat 30, 31)
    uint64_t sw_cells;
    // arena offsets (bytes)
    uint64_t o_trip_ent;         // uint32[T]   (u << 10 | pos)
    uint64_t o_trip_slot;        // uint32[T]
    uint64_t o_tslot;            // uint32[tcap] claimant triple index while the table is built, then the k-mer rank
    uint64_t o_tcnt;             // uint32[tcap] occurrence count (sum of nreads)
    uint64_t o_trank;            // uint32[tcap] rank of slot (BK_EMPTY32 = dropped)
    uint64_t o_key_lo, o_key_hi; // uint64[M] sorted (count desc, mer desc)
    uint64_t o_kcnt;             // uint32[M]
    uint64_t o_kstate;           // uint8[M]
    uint64_t o_kstamp;           // int32[3*M]: checked, mset, firstpos
    uint64_t o_poff;             // uint32[M+1]
    uint64_t o_post;             // uint32[T]
    uint64_t o_first_contig;     // `out` offset of first contig record (linked list), 0 = none
    uint64_t o_last_contig;
    // split regions (bk_comp.hip.h): the read / k-mer graph of a noisy region falls into components that never meet; G units
    // (workgroups) assemble disjoint sets of them side by side.  split = G (0: the region is one unit)
    uint32_t split, pass;        // pass: 0 = first execution; k = k-th repair pass (components that met across units, merged, run again)
    uint32_t dbg_us[4];          // diagnostic (BK_DEBUG_SPLIT): unit 0's serial prefix, labelling, seed list (microseconds)
    uint32_t phase, serial_base, stamp_base, pad2_;      // phase 1: unit 0 has run the high-count seeds (the SV's own k-mers) alone and labelled what is left; the other units start then, their contig serials / stamps beyond unit 0's
    uint32_t units_done, n_cidx, n_pairs, n_conf, cidx_cap, pairs_cap;
    uint64_t o_rroot;            // uint32[U]  component root (a read index) of every unique read
    uint64_t o_kroot;            // uint32[M]  component root of every sample k-mer (BK_EMPTY32: homopolymer, never in the graph)
    uint64_t o_cinfo;            // uint32[U]  per component, at its root: unit | pass << 8 | flags (BK_CI_*)
    uint64_t o_cidx;             // 2 x uint64[cidx_cap]  contigs as emitted: (seed rank << 20 | sequence in the seed iteration, `out` offset)
    uint32_t unit_us[BK_SPLIT_G], unit_iters[BK_SPLIT_G];      // diagnostic (BK_DEBUG_SPLIT): time and seed iterations of each unit in the last pass
    uint64_t o_pairs;            // uint32[3 * pairs_cap]  (root a, root b, kind): components that met (kind 1: across units = conflict)
    uint64_t stamps[20];         // diagnostic builds only (-DBK_PHASE_STAMPS): s_memrealtime at phase boundaries
};

// k-mer states (akmers.mers membership, sv_assembly.py:301-326; buffer.used_mers :333)
enum { BK_K_LIVE = 0, BK_K_USED = 1, BK_K_REMOVED = 2 };
// unique-read flags
enum { BK_R_USED = 1, BK_R_DELETED = 2, BK_R_INDEL = 4, BK_R_HASN = 8 };
// per-read input flags (read_flag)
enum { BK_RF_INDEL = 1, BK_RF_HASN = 2 };
#define BK_CODE_N 4

// contig record in the `out` arena (o_* relative to the record start, 8-byte aligned; k-mers are
// stored as (lo, hi) key pairs so the record is self-contained)
struct BkHit { int32_t qs, qe, ts, te, strand, tidx, score, fq; };   // qs/qe in strand coordinates, fq = forward query start
struct BkContigRec {
    uint64_t next;               // `out` offset of the next contig record of the region (0 = end)
    uint32_t root, pass;         // split regions: the component the contig belongs to (its root when it was emitted) and the pass that made it
    uint64_t hits_off;           // `out` offset of BkHit[n_hits + n_sec] (realign stage: step-1 hits, then secondary alignments in no particular order), 0 = none
    int32_t seq_len, counts_len, n_kmers, n_reads, total_reads, n_hits;
    uint32_t o_seq, o_io, o_ot, o_klocs, o_kmers, o_reads, n_sec, size;
};

struct BkParams {
    const BkRegionDesc *desc; BkRegionWork *work; const BkPartnerDesc *partners;
    const uint32_t *reads; const uint16_t *read_len; const uint8_t *read_flag;
    const uint32_t *nlist;                                             // N calls: (read index in region << 10 | position), sorted per region
    const uint32_t *wnlist;                                            // N positions of the windows (target and partner), sorted per window
    const uint32_t *sc; const uint16_t *sc_len;
    const uint32_t *windows;
    // read grouping (sized by total reads / total dedup slots)
    unsigned long long *dd_slot; uint32_t *dd_rep, *dd_cnt;            // dd_rep holds ulen[] (unique-read lengths, indexed from dedup_off)
    uint32_t *grp_slot;
    // unique-read arrays (indexed read_meta_off + u)
    uint32_t *urep, *unreads; uint8_t *uflag; int32_t *ubuf, *ureads, *ufound, *uminpos;
    // arena
    uint8_t *arena; unsigned long long *arena_top; uint64_t arena_cap;     // scratch: k-mer tables, assembler state
    uint8_t *out; unsigned long long *out_top; uint64_t out_cap;           // results: contig records, hits (copied to the host)
    // scheduling (bk_sched.hip.h): regions in descending order of estimated assembler cost, pulled by persistent
    // workgroups; every emitted contig is appended to `clist` (its `out` offset | region << 40), pulled by the realigner
    uint32_t *order; unsigned long long *asm_head, *sw_head, *n_clist, *n_queue; unsigned long long *clist; uint64_t clist_cap;
    // the unit queue is DYNAMIC since round 5 (bk_asm.hip.h): order[0 .. *n_queue0) are the entries of the launch (bk_sched_kernel / the
    // host), a split region whose components met across units appends the units of its next pass itself (entries up to order_cap; *n_queue
    // = entries allocated, an entry is valid once its word is not BK_EMPTY32), *asm_head = next entry to hand out, *pending = split
    // regions that may still append (persistent workgroups that find the queue empty wait for it to reach 0)
    unsigned long long *n_queue0, *pending, *queue_cap; uint32_t order_cap, pad_q;      // *queue_cap: entries of `order` that may be used by this launch (marked empty beyond *n_queue0)
    uint32_t asm_lds_pad, dbg_iters;      // dbg_iters (BK_DBG_ITERS=k): diagnostic -- every region stops after k seed iterations (0: no limit)
    //      // diagnostic (BK_LDS_PAD=<bytes>): a guard band behind the assembler's LDS block, filled before and checked after every region (BkRegionWork.stamps[14..15])
    uint32_t asm_lds_bytes, poison;      // diagnostic (BK_POISON_LDS=<byte>): the assembler's LDS block is filled with this byte before every region (an uninitialised read then behaves the same whatever ran on the CU before)
    unsigned long long *sw_long, *n_sw_long, *sw_long_head;      // contigs too long for the realigner's SHORT tier (bk_sw.hip.h): same entries as clist, same capacity      // order: (region | unit << 24) entries, *n_queue of them
    int32_t k, rc_thresh, max_contig, max_read, max_cand, sw_min_score;
    int32_t n_regions;           // regions of this launch (= length of the assembler's queue `order`)
    int32_t flags;               // BK_F_*
    const uint32_t *rmap;        // launch over a subset of the batch (re-run of regions that overflowed a cap): workgroup b of the k-mer kernels takes region rmap[b]; nullptr = region b
};
// BkParams.flags = bk_config.flags (include/breakmer_hip.h: BK_CFG_*, same values; every bit is another way to the same results)
enum {
    BK_F_NO_DUAL = 1,                // diagnostic: one overlap DP per wavefront even for short contigs
    BK_F_SPEC4 = 2,                  // diagnostic: at most 4 look-ahead slots
    BK_F_DUAL_ALWAYS = 4,            // diagnostic: both DPs of a slot on one wavefront whatever the round holds
    BK_F_NO_XVISIT = 8,              // look-ahead within one k-mer visit only
    BK_F_NO_XSEED = 16,              // no look-ahead into the next seeds
    BK_F_BUCKET_SORT = 32,           // the k-mer stage orders the seed k-mers with the bucket sort of large regions whatever their number
    BK_F_NO_RUN_RETIRE = 64,         // every read retired on its own
    BK_F_NO_SPLIT = 128,             // every region is one unit (no component split)
    BK_F_SPLIT_ALWAYS = 256,         // split whatever the size (the split path on small fixtures)
    BK_F_SPLIT_NO_LOOKAHEAD = 512,   // diagnostic: no look-ahead inside split regions (the round-4 setting)
    BK_F_SPLIT = 1024,               // accepted, no effect (it switched the split on while it was experimental)
    BK_F_NO_CALL_SHORTCUT = 2048,    // bk_call takes every contig through the full caller
    BK_F_HOST_REPAIR = 4096,         // split regions whose components met are repaired by host-driven passes (the fallback of the in-kernel repair, kept testable)
    BK_F_NO_SCORE_SWEEP = 8192,      // every overlap DP is the full sweep with origins (the round-4 DP rounds)
    BK_F_FORCE_REDO = 32768,         // diagnostic: the score sweep of a long-contig round flags EVERY read, so that every slot goes through the full overlap DPs afterwards (bk_dp_redo, all its passes)
    BK_F_PREQUEUE_UNITS = 16384      // the units of a split region are queue entries of the launch and wait for unit 0 (the round-5 queue); default since round 6: unit 0 appends them once the graph is labelled
};

// component info word (BkRegionWork.o_cinfo, at the component's root read)
#define BK_CI_UNIT 0xFFu
#define BK_CI_NOUNIT 0xFFu            // no unit owns it: a component without seed k-mers, until a unit's contig reaches it and claims it
#define BK_CI_ACTIVE 0x10000u         // has seed k-mers (count >= 2)
#define BK_CI_REDO 0x20000u           // resolve kernel: merged with a component it met across units; runs again in the next pass
#define BK_CI_ABORT 0x40000u          // its unit gave it up in this pass (it met another unit's component)
#define BK_SPLIT_HI 8                 // seeds with a count of at least this are run by unit 0 alone, in order, before the graph is labelled (bk_comp.hip.h)
#define BK_QUEUE_UNIT_SHIFT 24
#define BK_QUEUE_NOP 0xFFFFFFFEu      // a reserved queue entry that turned out to have no work (its region failed while the entries were reserved)
#define BK_REQUEUE_PASSES 12          // room in the unit queue for this many in-kernel repair passes of every split region (beyond it: the host drives the pass)

__device__ inline uint64_t bk_align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// entries [lo, hi) of a region's N list that belong to read i (binary search; only reads flagged HASN get here)
__device__ inline void bk_nlist_range(const uint32_t *nl, uint32_t n, uint32_t i, uint32_t &lo, uint32_t &hi)
{
    uint32_t a = 0, b = n;
    while (a < b) { const uint32_t m = (a + b) >> 1; if ((nl[m] >> 10) < i) a = m + 1; else b = m; }
    lo = a; b = n;
    while (a < b) { const uint32_t m = (a + b) >> 1; if ((nl[m] >> 10) <= i) a = m + 1; else b = m; }
    hi = a;
}
