// bk_kmer.hip.h -- stage BK_STAGE_KMER: read grouping (T1) + sample-only k-mer selection (K1/K2)
// + k-mer -> read posting lists (replaces the regex scan of find_reads, sv_assembly.py:111-122).
//
// One 512-thread workgroup per target region.  Reference behaviour reproduced:
//   utils.py:239-244      fq_recs = dict seq -> [fq_read...]   (grouping, first-occurrence order)
//   utils.py:151-178      jellyfish count -m k (no -C: strand specific), dump -c
//   utils.py:287-296      load_kmers: counts summed over files
//   sv_processor.py:613-631  sample_only = (keys(case) & keys(case_sc)) - keys(ref fwd+rc)
//   sv_assembly.py:272-285   drop homopolymers, order by (count, mer) descending
// Bound: HBM streaming of the packed reads (one pass for grouping, one per k-mer pass over the
// unique reads; the re-reads hit L2/MALL).  No MFMA: integer/byte work.
#pragma once
#include <type_traits>
#include "bk_common.h"
#undef BK_SRC_ID
#define BK_SRC_ID 1      // barrier sites of this file (bk_common.h: BK_SYNC)

#define BK_KT 512             // threads per workgroup when batches are in flight (8 waves: co-resides with the assembler workgroups of other batches)
#define BK_KT_MAX 1024        // ... when one batch runs at a time (every phase is a chain of dependent accesses of ONE workgroup: 0.30 -> 0.21 ms per launch)

// Reference k-mer set: window forward + reverse complement.  GLB = false: packed windows and table in LDS, entries
// (tag:14 | idx:18); GLB = true (windows that do not fit the LDS, e.g. whole-gene targets): windows and table in
// global memory (scratch arena), entries (tag:32 | idx:32).
template <bool GLB>
struct BkRefTabT {
    typedef typename std::conditional<GLB, unsigned long long, uint32_t>::type E;
    const uint32_t *win_f, *win_r;   // packed
    const uint32_t *nbits = nullptr; // windows with N (packed as code 0): bit i set = the k-mer at index i (forward 0..wk-1, reverse wk..2wk-1) covers an N: it
                                     // does not exist (Jellyfish skips it, utils.py:151-178), is not in the table, and no shortcut may step onto it
    __device__ inline bool dead(int idx) const { return nbits && ((nbits[idx >> 5] >> (idx & 31)) & 1u); }
    E *tab;
    uint32_t cap_mask; int wk;       // wk = W-k+1 k-mers per strand
    int k, nww;                      // nww = words of each packed window copy
    static constexpr E EMPTY = (E)~(E)0;
    __device__ inline BkKey key_at(uint32_t idx) const {
        return idx < (uint32_t)wk ? seq_kmer_fast(win_f, nww, (int)idx, k) : seq_kmer_fast(win_r, nww, (int)idx - wk, k);
    }
    __device__ static inline void split(const BkKey &key, uint32_t cap_mask, uint32_t &slot, uint32_t &tag) {
        if constexpr (GLB) { const uint64_t h = mix64(key.lo ^ (key.hi * 0x9E3779B97F4A7C15ull)); slot = (uint32_t)(h >> 20) & cap_mask; tag = (uint32_t)(h >> 32) ^ (uint32_t)h; }
        else { const uint32_t h = key_hash(key); slot = h & cap_mask; tag = (h >> 18) & 0x3FFFu; }
    }
    __device__ static inline E pack(uint32_t tag, uint32_t idx) { if constexpr (GLB) return ((E)tag << 32) | idx; else return (tag << 18) | idx; }
    __device__ static inline uint32_t e_tag(E e) { if constexpr (GLB) return (uint32_t)(e >> 32); else return e >> 18; }
    __device__ static inline uint32_t e_idx(E e) { if constexpr (GLB) return (uint32_t)e; else return e & 0x3FFFFu; }
    __device__ inline int find(const BkKey &key) const {
        uint32_t i, tag; split(key, cap_mask, i, tag);
        for (;;) {
            const E e = tab[i];
            if (e == EMPTY) return -1;
            if (e_tag(e) == tag) { const uint32_t idx = e_idx(e); if (key_eq(key_at(idx), key)) return (int)idx; }
            i = (i + 1) & cap_mask;
        }
    }
    __device__ inline void insert(uint32_t idx) {
        const BkKey key = key_at(idx);
        uint32_t i, tag; split(key, cap_mask, i, tag);
        const E mine = pack(tag, idx);
        for (;;) {
            const E e = atomicCAS(&tab[i], EMPTY, mine);
            if (e == EMPTY) return;
            if (e_tag(e) == tag && key_eq(key_at(e_idx(e)), key)) return;   // duplicate k-mer
            i = (i + 1) & cap_mask;
        }
    }
    // next k-mer on the same diagonal as the previous hit (seed-and-extend: skips the hash probe)
    __device__ inline int extend(int ridx, uint32_t c) const {
        if (ridx < 0) return -1;
        if (ridx < wk) { int cand = ridx + 1; return (cand < wk && seq_base(win_f, cand + k - 1) == c && !dead(cand)) ? cand : -1; }
        int cand = ridx - wk + 1; return (cand < wk && seq_base(win_r, cand + k - 1) == c && !dead(wk + cand)) ? wk + cand : -1;
    }
};

// walk one packed sequence, call f(pos, key) for every k-mer that is NOT a reference k-mer
template <class RT, class F>
__device__ inline void bk_scan_nonref(const uint32_t *w, int len, const RT &rt, F &&f)
{
    const int k = rt.k;
    BkKey key; key.hi = 0; key.lo = 0; int ridx = -1; uint32_t word = 0;
    for (int i = 0; i < len; i++) {
        if ((i & 15) == 0) word = w[i >> 4];
        uint32_t c = word >> 30; word <<= 2;
        key_push(key, c, k);
        if (i >= k - 1) {
            int nidx = rt.extend(ridx, c);
            if (nidx < 0) nidx = rt.find(key);
            ridx = nidx;
            if (nidx < 0) f(i - k + 1, key);
        }
    }
}

// Reads up to 256 bp: all packed words are fetched up front with independent loads (a thread that walks its read
// word by word pays one dependent global-load latency per 16 bases), then the k-mers are rolled out of registers.
#define BK_RW_MAX 16
__device__ inline void bk_load_words(const uint32_t *gw, uint32_t nw, uint32_t (&wb)[BK_RW_MAX])
{
#pragma unroll
    for (int t = 0; t < BK_RW_MAX; t++) wb[t] = (uint32_t)t < nw ? gw[t] : 0u;
}
// Word-level seed-and-extend: a read that keeps matching the reference on the diagonal of its previous k-mer hit is
// advanced 16 bases at a time (one funnel-shifted compare against the packed window in LDS); only the stretches
// that leave the diagonal (sequencing errors, the SV junction) are walked base by base with hash probes.
template <class RT, class F>
__device__ inline void bk_scan_nonref_regs(const uint32_t (&wb)[BK_RW_MAX], int len, const RT &rt, F &&f)
{
    const int k = rt.k;
    BkKey key; key.hi = 0; key.lo = 0; int ridx = -1;
#pragma unroll
    for (int wi = 0; wi < BK_RW_MAX; wi++) {
        if (wi * 16 < len) {
            uint32_t word = wb[wi];
            const int e = min(16, len - wi * 16);
            bool fast = false;
            if (e == 16 && ridx >= 0 && wi * 16 >= k && !rt.nbits) {
                const bool fw = ridx < rt.wk; const int loc = fw ? ridx : ridx - rt.wk;
                if (loc + 16 < rt.wk) {                                    // the 16 next k-mers exist on this strand
                    const uint32_t *W = fw ? rt.win_f : rt.win_r;
                    const int off = loc + k, wq = off >> 4, sh = 2 * (off & 15);
                    const uint32_t v = sh ? (W[wq] << sh) | (W[wq + 1] >> (32 - sh)) : W[wq];
                    fast = v == word;
                }
            }
            if (fast) {
                key.hi = (key.hi << 32) | (key.lo >> 32); key.lo = (key.lo << 32) | word;
                if (k <= 32) { key.hi = 0; if (k < 32) key.lo &= ((1ull << (2 * k)) - 1ull); }
                else if (k < 64) key.hi &= ((1ull << (2 * (k - 32))) - 1ull);
                ridx += 16;
            } else {
                for (int b = 0; b < e; b++) {
                    const int i = wi * 16 + b;
                    uint32_t c = word >> 30; word <<= 2;
                    key_push(key, c, k);
                    if (i >= k - 1) {
                        int nidx = rt.extend(ridx, c);
                        if (nidx < 0) nidx = rt.find(key);
                        ridx = nidx;
                        if (nidx < 0) f(i - k + 1, key);
                    }
                }
            }
        }
    }
}

// Phase A of the k-mer passes: does this read simply match the window?  (true for the bulk of the reads.)  One hash
// probe for its first k-mer gives a window position and strand; the read is "clean" iff it equals the window there over
// its whole length (then every k-mer of it is a reference k-mer): word compares against the packed window in LDS.
// Anything else -- no hit, a repeat that put the probe on another copy, the window end -- returns false and the read is
// left to the full scan.  Separating the two populations matters on a SIMT machine: in a mixed wavefront every lane pays
// for the slow path of one lane.  (The first version walked the first k bases one by one: 75 of the kernel's 460 us.)
template <class RT>
__device__ inline bool bk_read_is_clean(const uint32_t (&wb)[BK_RW_MAX], int len, const RT &rt)
{
    const int k = rt.k;
    if (len < k) return true;                               // no k-mers at all
    const int ridx = rt.find(seq_kmer_fast(wb, BK_RW_MAX, 0, k));
    if (ridx < 0) return false;
    const bool fw = ridx < rt.wk; const int loc = fw ? ridx : ridx - rt.wk;
    if (loc + len > rt.wk + k - 1) return false;            // runs off the window
    if (rt.nbits) for (int x = ridx; x <= ridx + len - k; x++) if (rt.dead(x)) return false;      // the window has an N under this read
    const uint32_t *W = fw ? rt.win_f : rt.win_r;
    bool ok = true;
#pragma unroll
    for (int wi = 0; wi < BK_RW_MAX; wi++) {
        if (wi * 16 < len) {
            const int off = loc + wi * 16, wq = off >> 4, sh = 2 * (off & 15);
            const uint32_t v = sh ? (W[wq] << sh) | (W[wq + 1] >> (32 - sh)) : W[wq];
            const int nb = min(16, len - wi * 16);
            const uint32_t mask = nb == 16 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (2 * nb));
            ok = ok && ((v ^ wb[wi]) & mask) == 0;
        }
    }
    return ok;
}

__device__ inline uint32_t bk_block_sum(uint32_t v, uint32_t *scratch /* >= 17 words */)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    BK_SYNC();
    if ((threadIdx.x & 63) == 0) scratch[wv] = v;
    BK_SYNC();
    if (threadIdx.x == 0) { uint32_t s = 0; for (int i = 0; i < nw; i++) s += scratch[i]; scratch[16] = s; }
    BK_SYNC();
    return scratch[16];
}
// exclusive scan of one value per thread across the block; returns (prefix, total via *total)
__device__ inline uint32_t bk_block_excl_scan(uint32_t v, uint32_t *scratch /* >= 18 words */, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    BK_SYNC();
    if (lane == 63) scratch[wv] = inc;
    BK_SYNC();
    if (threadIdx.x == 0) { uint32_t s = 0; for (int i = 0; i < nw; i++) { uint32_t t = scratch[i]; scratch[i] = s; s += t; } scratch[17] = s; }
    BK_SYNC();
    *total = scratch[17];
    return scratch[wv] + inc - v;
}

__device__ inline uint64_t bk_arena_alloc(const BkParams &p, uint64_t bytes, uint32_t *bcast /* LDS 2 words */)
{
    BK_SYNC();
    if (threadIdx.x == 0) {
        uint64_t need = bk_align_up(bytes, 256);
        uint64_t off = atomicAdd(p.arena_top, (unsigned long long)need);
        if (off + need > p.arena_cap) off = ~0ull;
        bcast[0] = (uint32_t)off; bcast[1] = (uint32_t)(off >> 32);
    }
    BK_SYNC();
    return ((uint64_t)bcast[1] << 32) | bcast[0];
}

// comparator of the akmers order: (count, mer) descending (sv_assembly.py:281)
__device__ inline bool bk_kmer_before(uint32_t ca, const BkKey &ka, uint32_t cb, const BkKey &kb)
{
    if (ca != cb) return ca > cb;
    return key_lt(kb, ka);
}

// ---- order of the seed-capable k-mers when they do not fit the LDS (regions with heavy sequencing noise: millions) ----
// One workgroup cannot bitonic-sort millions of keys in global memory in reasonable time (0.2 s for 3.4 M).  The order is
// (count, mer) descending and the k-mers are spread evenly over the mer space, so: histogram over the counts (<= 254
// exact, the rest in one class), every class split by as many leading mer bits as make its buckets ~128 k-mers, one
// counting pass and one scattering pass through global counters, then each bucket is sorted by ONE wavefront in LDS on
// a 64-bit proxy (count and leading mer bits; ties settled on the full key).  A bucket that does not fit (more than
// 512: skewed mers, many counts >= 255) makes the caller fall back to the global bitonic sort.
#define BK_BS_CAP 512
__device__ inline uint64_t bk_key_top64(const BkKey &key, int k)
{
    const int B = 2 * k;
    return B <= 64 ? key.lo << (64 - B) : (B == 128 ? key.hi : ((key.hi << (128 - B)) | (key.lo >> (B - 64))));
}
// returns (uniform) whether perm[0..M2) is in order now; L: LDS scratch of LC words; bst/bcur: global scratch of `bcap` words each
__device__ inline bool bk_bucket_sort(uint32_t *perm, uint32_t M2, const uint64_t *klo, const uint64_t *khi, const uint32_t *kcnt, int k,
                                      uint32_t *L, uint32_t LC, uint32_t *bst, uint32_t *bcur, uint32_t bcap, uint32_t *scr)
{
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6;
    if (LC < 1024 + 1536) return false;
    uint32_t *hc = L, *sb = L + 256, *bb = L + 512;            // per count class: size, split bits, first bucket
    for (uint32_t c = tid; c < 256; c += nt) hc[c] = 0;
    BK_SYNC();
    for (uint32_t j = tid; j < M2; j += nt) atomicAdd(&hc[min(kcnt[j], 255u)], 1u);
    BK_SYNC();
    if (tid == 0) {
        uint32_t nbk = 0;
        for (int c = 255; c >= 0; c--) {                         // descending count
            uint32_t s = 0;
            if (c < 255) while ((hc[c] >> s) > 128u && s < 24u) s++;
            sb[c] = s; bb[c] = nbk; nbk += hc[c] ? (1u << s) : 0u;
        }
        scr[26] = nbk;
    }
    BK_SYNC();
    const uint32_t NB = scr[26];
    if (NB + 1 > bcap) return false;
    for (uint32_t b = tid; b <= NB; b += nt) bst[b] = 0;
    BK_SYNC();
    auto bucket_of = [&](uint32_t j) -> uint32_t {
        const uint32_t c = min(kcnt[j], 255u), s = sb[c];
        const BkKey key{khi[j], klo[j]};
        const uint32_t top = s ? (uint32_t)(bk_key_top64(key, k) >> (64 - s)) : 0u;
        return bb[c] + ((1u << s) - 1u - top);                   // mer descending inside the class
    };
    for (uint32_t j = tid; j < M2; j += nt) atomicAdd(&bst[bucket_of(j)], 1u);
    BK_SYNC();
    {   // exclusive prefix over the buckets -> first position of each; a copy serves as the scatter cursor
        const uint32_t chunk = (NB + nt - 1) / nt, b0 = tid * chunk, b1 = min(NB, b0 + chunk);
        uint32_t c = 0, tot = 0, big = 0;
        for (uint32_t b = b0; b < b1; b++) { c += bst[b]; big |= bst[b] > BK_BS_CAP; }
        uint32_t pre = bk_block_excl_scan(c, scr, &tot);
        for (uint32_t b = b0; b < b1; b++) { const uint32_t n = bst[b]; bst[b] = pre; bcur[b] = pre; pre += n; }
        if (tid == 0) { bst[NB] = tot; scr[27] = 0; }
        BK_SYNC();
        if (big) scr[27] = 1;
        BK_SYNC();
        if (scr[27]) return false;                               // some bucket is larger than a wavefront sorts
    }
    for (uint32_t j = tid; j < M2; j += nt) perm[atomicAdd(&bcur[bucket_of(j)], 1u)] = j;
    BK_SYNC();
    // one wavefront per bucket; (proxy, index) pairs in LDS
    const uint32_t stride = BK_BS_CAP * 3, nwv = min(nt >> 6, LC / stride);
    if (wv < nwv) {
        unsigned long long *K = (unsigned long long *)(L + wv * stride); uint32_t *P = L + wv * stride + 2 * BK_BS_CAP;
        for (uint32_t b = wv; b < NB; b += nwv) {
            const uint32_t start = bst[b], n = bst[b + 1] - start;
            if (n < 2) continue;
            uint32_t np2 = 2; while (np2 < n) np2 <<= 1;
            for (uint32_t e = lane; e < np2; e += 64) {
                if (e < n) {
                    const uint32_t j = perm[start + e]; const BkKey key{khi[j], klo[j]};
                    K[e] = ((unsigned long long)min(kcnt[j], 0xFFFFFu) << 44) | (bk_key_top64(key, k) >> 20); P[e] = j;
                } else { K[e] = 0; P[e] = BK_EMPTY32; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            for (uint32_t sz = 2; sz <= np2; sz <<= 1)
                for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
                    for (uint32_t i = lane; i < np2 / 2; i += 64) {
                        const uint32_t lo = (i / st) * (st * 2) + (i % st), hi2 = lo + st;
                        const bool up = (lo & sz) == 0;
                        const unsigned long long ka = K[lo], kb = K[hi2]; const uint32_t a = P[lo], b2 = P[hi2];
                        bool a_first;
                        if (a == BK_EMPTY32) a_first = false; else if (b2 == BK_EMPTY32) a_first = true;
                        else if (ka != kb && ((ka & kb) >> 44) != 0xFFFFFull) a_first = ka > kb;      // both counts saturate the proxy (>= 2^20): only the full comparison orders them
                        else { const BkKey fa{khi[a], klo[a]}, fb{khi[b2], klo[b2]}; a_first = bk_kmer_before(kcnt[a], fa, kcnt[b2], fb); }
                        if (a_first != up) { K[lo] = kb; K[hi2] = ka; P[lo] = b2; P[hi2] = a; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
            for (uint32_t e = lane; e < n; e += 64) perm[start + e] = P[e];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    BK_SYNC();
    return true;
}

#ifdef BK_PHASE_STAMPS
#define BK_STAMP(i) do { BK_SYNC(); if (threadIdx.x == 0) wk->stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BK_STAMP(i) do { } while (0)
#endif

// ---- P1 + P2: group identical reads (utils.py:239-244), unique reads in first-occurrence (FASTQ) order -----------
// Open-addressing table, slot = (hash tag | smallest read index seen so far): probing reads it with a plain load, only
// the first read of a sequence does a CAS and only a smaller index an atomicMin, so a duplicate costs one atomicAdd of
// the count.  LG = true: the table lives in LDS (32-bit slots tag:18|index:14; the copy counts are a second pass that
// reuses the same words as a histogram once the representatives are known) -- no table initialisation or atomics in
// global memory; needs n_reads < 16383 and dedup_cap words of LDS, which the reference k-mer set only takes over
// afterwards.  LG = false: 64-bit slots and 32-bit counts in global memory.
template <bool LG>
__device__ inline uint32_t bk_group_reads(const BkParams &p, const BkRegionDesc &d, uint32_t *scr, uint32_t *lslot)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const uint32_t N = d.n_reads, RW = d.read_words;
    const uint32_t *reads = p.reads + d.reads_word_off;
    const uint16_t *rlen = p.read_len + d.read_meta_off;
    unsigned long long *dslot = p.dd_slot + d.dedup_off;
    uint32_t *dcnt = p.dd_cnt + d.dedup_off;
    uint32_t *gslot = p.grp_slot + d.read_meta_off;
    if constexpr (LG) { for (uint32_t i = tid; i < d.dedup_cap; i += nt) lslot[i] = BK_EMPTY32; }
    else for (uint32_t i = tid; i < d.dedup_cap; i += nt) { dslot[i] = BK_EMPTY64; dcnt[i] = 0; }
    BK_SYNC();
    const uint32_t dmask = d.dedup_cap - 1;
    const uint8_t *rfl = p.read_flag + d.read_meta_off;
    const uint32_t *nl = p.nlist + d.nlist_off; const uint32_t nnl = d.n_nlist;
    for (uint32_t i = tid; i < N; i += nt) {
        const uint32_t *w = reads + (uint64_t)i * RW; const uint32_t len = rlen[i], nw = (len + 15) / 16;
        uint32_t wb[BK_RW_MAX]; bk_load_words(w, min(nw, (uint32_t)BK_RW_MAX), wb);
        // two 32-bit multiplicative lanes per word and one 64-bit mix at the end: a 64 x 64 multiply is eight quarter-rate
        // instructions on this machine and the pass spent a third of its time in eleven splitmix rounds per read (the hash
        // only places the read in the table: equality is decided by comparing the strings)
        uint32_t h1 = 0x9E3779B9u ^ len, h2 = 0x85EBCA6Bu + len;
        auto absorb = [&](uint32_t x) { h1 = (h1 ^ x) * 0x9E3779B1u; h1 = (h1 << 15) | (h1 >> 17); h2 = (h2 + x) * 0xC2B2AE35u; h2 ^= h2 >> 15; };
        const bool hasn = nnl && (rfl[i] & BK_RF_HASN);
        uint32_t nlo = 0, nhi = 0;
        if (hasn) { bk_nlist_range(nl, nnl, i, nlo, nhi); for (uint32_t e = nlo; e < nhi; e++) absorb(0x4E00u | (nl[e] & 1023u)); }   // the N calls are part of the string
        if (nw <= BK_RW_MAX) {
#pragma unroll
            for (int t = 0; t < BK_RW_MAX; t++) if ((uint32_t)t < nw) absorb(wb[t]);
        } else for (uint32_t t = 0; t < nw; t++) absorb(w[t]);
        uint64_t h = mix64(((uint64_t)h1 << 32) | h2);
        const uint32_t tag = LG ? (uint32_t)(h >> 32) & 0x3FFFFu : (uint32_t)(h >> 32); uint32_t s = (uint32_t)h & dmask;
        const unsigned long long mine = LG ? (unsigned long long)((tag << 14) | i) : (((unsigned long long)tag << 32) | i);
        for (;;) {
            unsigned long long cur; bool empty;
            if constexpr (LG) { uint32_t c32 = *(volatile uint32_t *)&lslot[s]; if (c32 == BK_EMPTY32) c32 = atomicCAS(&lslot[s], BK_EMPTY32, (uint32_t)mine); empty = c32 == BK_EMPTY32; cur = c32; }
            else { cur = __atomic_load_n(&dslot[s], __ATOMIC_RELAXED); if (cur == BK_EMPTY64) cur = atomicCAS(&dslot[s], BK_EMPTY64, mine); empty = cur == BK_EMPTY64; }
            if (empty) break;
            if ((LG ? (uint32_t)cur >> 14 : (uint32_t)(cur >> 32)) == tag) {
                const uint32_t j = LG ? (uint32_t)cur & 0x3FFFu : (uint32_t)cur; bool same = rlen[j] == len;
                if (nnl && same) {                                    // same N calls (packed words hold code 0 under an N)
                    const bool jn = (rfl[j] & BK_RF_HASN) != 0;
                    same = jn == hasn;
                    if (same && hasn) {
                        uint32_t jlo, jhi; bk_nlist_range(nl, nnl, j, jlo, jhi);
                        same = jhi - jlo == nhi - nlo;
                        for (uint32_t e = 0; same && e < nhi - nlo; e++) same = (nl[nlo + e] & 1023u) == (nl[jlo + e] & 1023u);
                    }
                }
                const uint32_t *wj = reads + (uint64_t)j * RW;
                if (same) {
                    if (nw <= BK_RW_MAX) {
                        uint32_t wo[BK_RW_MAX]; bk_load_words(wj, nw, wo);
#pragma unroll
                        for (int t = 0; t < BK_RW_MAX; t++) same = same && wo[t] == wb[t];
                    } else for (uint32_t t = 0; same && t < nw; t++) same = wj[t] == w[t];
                }
                if (same) { if (i < j) { if constexpr (LG) atomicMin(&lslot[s], (uint32_t)mine); else atomicMin(&dslot[s], mine); } break; }
            }
            s = (s + 1) & dmask;
        }
        gslot[i] = s;
        if constexpr (!LG) atomicAdd(&dcnt[s], 1u);
    }
    BK_SYNC();
#ifdef BK_PHASE_STAMPS
    if (threadIdx.x == 0) p.work[p.rmap ? p.rmap[blockIdx.x] : blockIdx.x].stamps[2] = __builtin_amdgcn_s_memrealtime();
#endif
    auto rep_of = [&](uint32_t sl) -> uint32_t { if constexpr (LG) return lslot[sl] & 0x3FFFu; else return (uint32_t)dslot[sl]; };
    auto cnt_of = [&](uint32_t sl) -> uint32_t { if constexpr (LG) return lslot[sl]; else return dcnt[sl]; };
    uint32_t *urep = p.urep + d.read_meta_off, *unr = p.unreads + d.read_meta_off;
    uint8_t *ufl = p.uflag + d.read_meta_off;
    uint32_t *ulen = p.dd_rep + d.dedup_off;                    // length per unique read, for the assembler's find_reads
    const uint8_t *rflag = p.read_flag + d.read_meta_off;
    uint32_t U = 0;
    const uint32_t chunk = (N + nt - 1) / nt, b = tid * chunk, e = min(N, b + chunk);
    uint32_t c = 0;
    if (chunk <= 32) {
        // all loads of the chunk issued back to back (two dependent rounds instead of 2*chunk)
        uint32_t g[32]; uint32_t isrep = 0;
#pragma unroll
        for (int t = 0; t < 32; t++) g[t] = b + t < e ? gslot[b + t] : 0u;
#pragma unroll
        for (int t = 0; t < 32; t++) if (b + t < e && rep_of(g[t]) == b + t) isrep |= 1u << t;
        c = __popc(isrep);
        if constexpr (LG) {                                     // copy counts: the slot words become a histogram over grp_slot
            BK_SYNC();
            for (uint32_t i = tid; i < d.dedup_cap; i += nt) lslot[i] = 0;
            BK_SYNC();
#pragma unroll
            for (int t = 0; t < 32; t++) if (b + t < e) atomicAdd(&lslot[g[t]], 1u);
            BK_SYNC();
        }
        uint32_t pre = bk_block_excl_scan(c, scr, &U);
#pragma unroll
        for (int t = 0; t < 32; t++) if (isrep & (1u << t)) {
            urep[pre++] = b + t;
        }
    } else {
        for (uint32_t i = b; i < e; i++) c += rep_of(gslot[i]) == i;
        uint32_t pre = bk_block_excl_scan(c, scr, &U);
        for (uint32_t i = b; i < e; i++) if (rep_of(gslot[i]) == i) urep[pre++] = i;
    }
    BK_SYNC();
    // only the representative list is written from the per-thread chunks (scattered); the other per-unique-read arrays
    // are filled by index so that a wave writes whole lines
    for (uint32_t j = tid; j < U; j += nt) {
        const uint32_t i = urep[j], sl = gslot[i]; const uint8_t f = rflag[i];
        unr[j] = cnt_of(sl); ufl[j] = ((f & BK_RF_INDEL) ? BK_R_INDEL : 0) | ((f & BK_RF_HASN) ? BK_R_HASN : 0); ulen[j] = rlen[i];
        p.ubuf[d.read_meta_off + j] = 0; p.ureads[d.read_meta_off + j] = 0; p.ufound[d.read_meta_off + j] = -1; p.uminpos[d.read_meta_off + j] = 0x7FFFFFFF;
    }
    BK_SYNC();
    return U;
}

#include "bk_comp.hip.h"
#undef BK_SRC_ID
#define BK_SRC_ID 1

#define BK_K_PERM_G 16384      // words of LDS sort permutation in the global-table variant

// GLB = false: regions whose window fits the LDS (d.big == 0); GLB = true: the others.  Both kernels are launched
// over all regions and return at once for regions of the other kind.
template <bool GLB>
__device__ inline void bk_kmer_body(const BkParams &p, uint32_t ref_cap, uint32_t win_words_cap, uint32_t lds_words, uint32_t *lds)
{
    const int r = p.rmap ? (int)p.rmap[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const BkRegionDesc d = p.desc[r];
    if ((d.big != 0) != GLB) return;
    BkRegionWork *wk = &p.work[r];
    const int k = p.k;
    uint32_t *scr = lds;                       // 32 words scratch
    uint32_t *stage = lds + 32;                // 16 words per wavefront: the read a wavefront scans cooperatively

    if (tid == 0) { wk->status = BK_ST_OK; wk->U = 0; wk->T = 0; wk->M = 0; wk->M2 = 0; wk->split = 0; wk->pass = 0; wk->phase = 0; wk->units_done = 0; wk->n_cidx = 0; wk->n_pairs = 0; wk->n_conf = 0; wk->tcap = 0; wk->n_contigs = 0; wk->nw_cells = 0; wk->nw_calls = 0; wk->sw_cells = 0; wk->dp_sweeps = 0; wk->dp_redos = 0; wk->o_first_contig = 0; wk->o_last_contig = 0; for (int q = 0; q < 4; q++) wk->stamps[q] = 0; }
    const int W = (int)d.win_len, WK = W >= k ? W - k + 1 : 0;
    const uint32_t *gw = p.windows + d.win_word_off;
    const int ww = (W + 15) / 16;
    BK_STAMP(1);
    // ---- P1, P2: read grouping, before the LDS is taken by the reference k-mer set -----------------
    const uint32_t N = d.n_reads, RW = d.read_words;
    const uint32_t *reads = p.reads + d.reads_word_off;
    const uint16_t *rlen = p.read_len + d.read_meta_off;
    uint32_t *gslot = p.grp_slot + d.read_meta_off;
    uint32_t *urep = p.urep + d.read_meta_off, *unr = p.unreads + d.read_meta_off;
    const bool lds_group = N < 16383u && 32u + 16u * 16u + d.dedup_cap <= lds_words;      // N <= 32 per thread: the register-chunk branch of P2
    const uint32_t U = lds_group ? bk_group_reads<true>(p, d, scr, lds + 32 + 16 * 16) : bk_group_reads<false>(p, d, scr, nullptr);
    BK_STAMP(0);
    // ---- P0: reference k-mer set (sv_processor.py:613-615: forward and reverse file) -------------
    BkRefTabT<GLB> rt; rt.cap_mask = 0; rt.wk = WK; rt.k = k;
    uint32_t *perm_lds; uint32_t perm_cap;
    if constexpr (!GLB) {
        uint32_t *win_f = lds + 32 + 16 * 16;
        uint32_t *win_r = win_f + win_words_cap;
        uint32_t *tab = win_r + win_words_cap;     // ref_cap words; later reused as sort permutation
        if ((uint32_t)(2 * WK) * 2u > ref_cap || (uint32_t)(ww + 2) > win_words_cap || 2 * WK >= (1 << 18)) {
            if (tid == 0) wk->status = BK_ST_WINDOW;
            return;
        }
        for (int i = tid; i < (int)win_words_cap; i += nt) { win_f[i] = i < ww ? gw[i] : 0u; win_r[i] = 0u; }
        for (uint32_t i = tid; i < ref_cap; i += nt) tab[i] = BK_EMPTY32;
        rt.win_f = win_f; rt.win_r = win_r; rt.tab = tab; rt.cap_mask = ref_cap - 1; rt.nww = (int)win_words_cap;
        perm_lds = tab; perm_cap = ref_cap;
    } else {
        uint32_t gcap = 1024; while (gcap < 4u * (uint32_t)WK && gcap < (1u << 30)) gcap <<= 1;     // load factor <= 0.5
        const uint64_t o_wr = bk_arena_alloc(p, (uint64_t)(ww + 2) * 4 + 256 + (uint64_t)gcap * 8, scr + 20);
        if (o_wr == ~0ull) { if (tid == 0) wk->status = BK_ST_ARENA; return; }
        if ((uint64_t)4 * WK > gcap) { if (tid == 0) wk->status = BK_ST_WINDOW; return; }
        uint32_t *win_r = (uint32_t *)(p.arena + o_wr);
        unsigned long long *tab = (unsigned long long *)(p.arena + bk_align_up(o_wr + (uint64_t)(ww + 2) * 4, 256));
        for (uint32_t i = tid; i < gcap; i += nt) tab[i] = ~0ull;
        if (tid < 2) win_r[ww + tid] = 0u;
        rt.win_f = gw; rt.win_r = win_r; rt.tab = tab; rt.cap_mask = gcap - 1; rt.nww = ww + 2;     // the host pads every packed window with 2 zero words
        perm_lds = lds + 32 + 16 * 16; perm_cap = BK_K_PERM_G;
    }
    BK_SYNC();
    // reverse complement, packed: base j of rc = 3 - base (W-1-j) of forward
    {
        uint32_t *win_r = const_cast<uint32_t *>(rt.win_r);
        for (int wi = tid; wi < ww; wi += nt) {
            uint32_t x = 0;
            for (int t = 0; t < 16; t++) { int j = wi * 16 + t; uint32_t c = j < W ? 3u - seq_base(rt.win_f, W - 1 - j) : 0u; x = (x << 2) | c; }
            win_r[wi] = x;
        }
    }
    BK_SYNC();
    if (d.n_win_n) {                                         // a window with N (rare): which window k-mers do not exist
        const uint32_t nbw = (uint32_t)(2 * WK + 31) / 32 + 1;
        const uint64_t o_nb = bk_arena_alloc(p, (uint64_t)nbw * 4, scr + 20);
        if (o_nb == ~0ull) { if (tid == 0) wk->status = BK_ST_ARENA; return; }
        uint32_t *nb = (uint32_t *)(p.arena + o_nb);
        for (uint32_t i = tid; i < nbw; i += nt) nb[i] = 0;
        BK_SYNC();
        const uint32_t *wn = p.wnlist + d.win_n_off;
        for (uint32_t e = tid; e < d.n_win_n; e += nt) {
            const int pf = (int)wn[e], pr = W - 1 - pf;      // position in the forward / in the reverse-complement window
            for (int x = max(pf - k + 1, 0); x <= min(pf, WK - 1); x++) atomicOr(&nb[x >> 5], 1u << (x & 31));
            for (int x = max(pr - k + 1, 0); x <= min(pr, WK - 1); x++) atomicOr(&nb[(WK + x) >> 5], 1u << ((WK + x) & 31));
        }
        __threadfence();
        BK_SYNC();
        rt.nbits = nb;
    }
    for (int i = tid; i < 2 * WK; i += nt) if (!rt.dead(i)) rt.insert((uint32_t)i);
    BK_SYNC();

    BK_STAMP(3);
    // ---- P3a: which unique reads have non-reference k-mers at all -------------------------------------
    // reads that simply match the window are recognised with word compares and dropped; the rest (sequencing
    // errors, SV junctions) go on a list (grp_slot is free after P2) and get the full scan
    uint32_t *slow = gslot;
    if (tid == 0) scr[25] = 0;
    BK_SYNC();
    uint32_t myk = 0;                                            // k-mer positions of the listed reads: upper bound of T
    for (uint32_t u = tid; u < U; u += nt) {
        const uint32_t i = urep[u]; const int len = rlen[i];
        bool clean = false;
        if (len <= 16 * BK_RW_MAX && !(p.uflag[d.read_meta_off + u] & BK_R_HASN)) { uint32_t wb[BK_RW_MAX]; bk_load_words(reads + (uint64_t)i * RW, (len + 15) / 16, wb); clean = bk_read_is_clean(wb, len, rt); }
        if (!clean) { slow[atomicAdd(&scr[25], 1u)] = u; myk += (uint32_t)max(len - k + 1, 0); }
    }
    BK_SYNC();
    const uint32_t nslow = scr[25];
    BK_STAMP(8);
    const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6;
    uint32_t *wst = stage + wv * 16;
    BK_STAMP(9);
#ifdef BK_PHASE_STAMPS
    if (tid == 0) wk->stamps[10] = nslow;
#endif
    const uint32_t Tmax = bk_block_sum(myk, scr);
    BK_STAMP(4);
    // ---- P3b: record the (u,pos) occurrences in ONE scan (buffer sized by the upper bound), then size and fill
    //      the sample k-mer table.  One WAVEFRONT per listed read, one k-mer position per lane (plain probes):
    //      walking them one read per lane serialises the rare expensive events of 64 different reads.
    uint64_t a0 = bk_arena_alloc(p, (uint64_t)Tmax * 8 + 1024, scr + 20);
    if (a0 == ~0ull) {
        // The arena is too small for this batch (a fresh handle sizes it for clean reads; the host grows it and runs the batch again).  The
        // bump pointer is the host's only measure of the demand, and a region that stops here has asked for a seventh of what it needs:
        // it adds an estimate of the rest -- sample k-mer table (<= 3 Tmax slots of 8 bytes), compact arrays and posting lists (~45 bytes per
        // distinct k-mer <= Tmax), and for a region of the size that is split into units their scratch over two or three passes -- so that
        // ONE growth step suffices (round 6; x4 at a time a noisy batch ran four times: 103 -> 495 -> 1,982 -> 7,931 MB).
        if (tid == 0) { wk->status = BK_ST_ARENA; atomicAdd(p.arena_top, (unsigned long long)Tmax * 24ull + (Tmax > 65536u ? ((p.flags & BK_F_NO_SPLIT) ? (6ull << 20) : (40ull << 20)) : (1ull << 20))); }
        return;
    }
    const uint64_t o_ent = a0, o_tsl = bk_align_up(o_ent + (uint64_t)Tmax * 4, 256);
    uint32_t *t_ent = (uint32_t *)(p.arena + o_ent), *t_sl = (uint32_t *)(p.arena + o_tsl);
    if (tid == 0) scr[24] = 0;
    BK_SYNC();
    // Each wavefront takes the listed reads wv, wv + 8, ...: the metadata of up to 64 of them is fetched by the 64 lanes
    // at once and the packed words of read j+1 are requested before read j is scanned, so the scan of a read does not
    // wait for three dependent global accesses of its own (they were 3 of the 4 us a read took).
    for (uint32_t q0 = wv; q0 < nslow; q0 += (uint32_t)nwv * 64u) {
        const uint32_t qm = q0 + (uint32_t)lane * (uint32_t)nwv;
        uint32_t mu = 0, mi = 0; int mlen = 0;
        if (qm < nslow) { mu = slow[qm]; mi = urep[mu]; mlen = (int)rlen[mi]; }
        const int cnt = (int)min(64u, (nslow - q0 + (uint32_t)nwv - 1u) / (uint32_t)nwv);
        uint32_t wnext = 0;
        { const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)mi, 0); const int l0 = __builtin_amdgcn_readlane(mlen, 0);
          if (lane < 16 && l0 <= 16 * BK_RW_MAX) wnext = lane < (l0 + 15) / 16 ? reads[(uint64_t)i0 * RW + lane] : 0u; }
        for (int j = 0; j < cnt; j++) {
            const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)mu, j), i = (uint32_t)__builtin_amdgcn_readlane((int)mi, j); const int len = __builtin_amdgcn_readlane(mlen, j);
            uint32_t nlo = 0, nhi = 0;                                  // N calls of this read (wave-uniform): Jellyfish skips the k-mers that contain one
            if (d.n_nlist && (p.uflag[d.read_meta_off + u] & BK_R_HASN)) bk_nlist_range(p.nlist + d.nlist_off, d.n_nlist, i, nlo, nhi);
            const uint32_t *nlp = p.nlist + d.nlist_off;
            auto has_n = [&](int pos) { bool b = false; for (uint32_t e = nlo; e < nhi; e++) { const int q = (int)(nlp[e] & 1023u); b = b || (q >= pos && q < pos + k); } return b; };
            auto rec = [&](int pos, const BkKey &) { if (has_n(pos)) return; uint32_t idx = atomicAdd(&scr[24], 1u); t_ent[idx] = (u << 10) | (uint32_t)pos; };
            const uint32_t wcur = wnext;
            if (j + 1 < cnt) {
                const uint32_t i2 = (uint32_t)__builtin_amdgcn_readlane((int)mi, j + 1); const int l2 = __builtin_amdgcn_readlane(mlen, j + 1);
                if (lane < 16 && l2 <= 16 * BK_RW_MAX) wnext = lane < (l2 + 15) / 16 ? reads[(uint64_t)i2 * RW + lane] : 0u;
            }
            if (len <= 16 * BK_RW_MAX) {
                if (lane < 16) wst[lane] = wcur;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                for (int p0 = 0; p0 + k <= len; p0 += 64) {
                    const int pp = p0 + lane;
                    const bool nonref = pp + k <= len && !has_n(pp) && rt.find(seq_kmer_fast(wst, 16, pp, k)) < 0;
                    const unsigned long long nm = __ballot(nonref);                 // one counter update per wavefront, not per lane
                    if (nm) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&scr[24], (uint32_t)__popcll(nm));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (nonref) t_ent[base + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull))] = (u << 10) | (uint32_t)pp;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            } else if (lane == 0) bk_scan_nonref(reads + (uint64_t)i * RW, len, rt, rec);
        }
    }
    BK_SYNC();
    const uint32_t T = scr[24];
    // sized by the OCCURRENCES (their number is all that is known here), of which the distinct k-mers are a part (3/4 at 5 % noise):
    // 1.5 T slots keep the load below 2/3 even if every occurrence were distinct; a power of two >= 2 T (round 2) doubled the
    // largest array of a noisy region (configs[4]: 134 MB of its ~340 MB) for nothing
    uint32_t tcap = 256; while (tcap < T + T / 2) tcap <<= 1;
    // The table is filled in LDS when it fits and the reference set is no longer needed (no separate soft-clip set to
    // scan): one CAS and one add per occurrence are device-scope atomics otherwise, each 64 B of write traffic.  In LDS
    // it also STAYS there: it is sized by the occurrences (their number is all that is known up front) but holds M
    // distinct k-mers -- 30 of 16,384 slots for a clean deletion -- so what the assembler gets is a compact copy sized by
    // M, built at the end; the full-size table never crosses HBM.
    const bool lds_tab = !GLB && d.n_sc < 0 && 2u * tcap + 32u + 16u * 16u <= lds_words;
    uint64_t o_tslot = 0, o_tcnt = 0;
    uint32_t *tslot, *tcnt;                                  // slot -> claimant occurrence, later the k-mer rank; slot -> count
    if (lds_tab) { tslot = lds + 32 + 16 * 16; tcnt = tslot + tcap; }
    else {
        uint64_t a0b = bk_arena_alloc(p, (uint64_t)tcap * 8 + 256, scr + 20);
        if (a0b == ~0ull) { if (tid == 0) wk->status = BK_ST_ARENA; return; }
        o_tslot = a0b; o_tcnt = o_tslot + (uint64_t)tcap * 4;
        tslot = (uint32_t *)(p.arena + o_tslot); tcnt = (uint32_t *)(p.arena + o_tcnt);
    }
    uint32_t *wslot = tslot, *wcnt = tcnt;
    for (uint32_t i = tid; i < tcap; i += nt) { wslot[i] = BK_EMPTY32; wcnt[i] = 0; }
    BK_SYNC();
    BK_STAMP(11);
    const uint32_t tmask = tcap - 1;
    // Four occurrences per thread at a time (round 6).  One occurrence is a chain of dependent accesses -- occurrence -> representative
    // read -> its words -> slot (CAS) and, where the slot is taken, its claimant's occurrence -> read -> words --, and a noisy region has
    // 10^5 (configs[4]: 4.4 M) of them on ONE workgroup: the first four links of four occurrences are in flight together, only the
    // claimant checks of the ones that met a taken slot follow one after the other.
    for (uint32_t idx0 = tid; idx0 < T; idx0 += 4 * (uint32_t)nt) {
        uint32_t ee[4], ri[4], ss[4], cur[4], cn[4]; BkKey kk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t idx = idx0 + (uint32_t)q * nt; ee[q] = idx < T ? t_ent[idx] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; q++) { const bool on = idx0 + (uint32_t)q * nt < T; ri[q] = on ? urep[ee[q] >> 10] : 0u; cn[q] = on ? unr[ee[q] >> 10] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; q++) { kk[q] = seq_kmer_fast(reads + (uint64_t)ri[q] * RW, (int)RW, (int)(ee[q] & 1023u), k); ss[q] = key_hash(kk[q]) & tmask; }
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t idx = idx0 + (uint32_t)q * nt; cur[q] = idx < T ? atomicCAS(&wslot[ss[q]], BK_EMPTY32, idx) : BK_EMPTY32; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t idx = idx0 + (uint32_t)q * nt;
            if (idx >= T) continue;
            uint32_t s = ss[q], c = cur[q];
            while (c != BK_EMPTY32) {
                const uint32_t e2 = t_ent[c];
                if (key_eq(seq_kmer_fast(reads + (uint64_t)urep[e2 >> 10] * RW, (int)RW, (int)(e2 & 1023u), k), kk[q])) break;
                s = (s + 1) & tmask;
                c = atomicCAS(&wslot[s], BK_EMPTY32, idx);
            }
            t_sl[idx] = s; atomicAdd(&wcnt[s], cn[q]);         // case[mer] counts every read (duplicates included)
        }
    }
    BK_SYNC();
    // soft-clip set: keep only k-mers also present in case_sc (sv_processor.py:619-621)
    if (d.n_sc >= 0) {
        const uint32_t *sc = p.sc + d.sc_word_off; const uint16_t *sl = p.sc_len + d.sc_meta_off;
        for (uint32_t i = tid; i < (uint32_t)d.n_sc; i += nt) {
            bk_scan_nonref(sc + (uint64_t)i * d.sc_words, sl[i], rt, [&](int, const BkKey &key) {
                uint32_t s = key_hash(key) & tmask;
                for (;;) {
                    uint32_t cur = tslot[s];
                    if (cur == BK_EMPTY32) break;
                    uint32_t e2 = t_ent[cur];
                    if (key_eq(seq_kmer_fast(reads + (uint64_t)urep[e2 >> 10] * RW, (int)RW, (int)(e2 & 1023u), k), key)) { atomicOr(&tcnt[s], 0x80000000u); break; }   // seen in case_sc
                    s = (s + 1) & tmask;
                }
            });
        }
        BK_SYNC();
        for (uint32_t i = tid; i < tcap; i += nt) { if (tslot[i] != BK_EMPTY32 && !(tcnt[i] & 0x80000000u)) tslot[i] = BK_EMPTY32 - 1; tcnt[i] &= 0x7FFFFFFFu; }   // tombstone: probe chains stay intact
        BK_SYNC();
    }
    BK_STAMP(5);
    // ---- P4: compact -> M sample-only k-mers.  Those that can seed a contig (count >= 2, init_assembly :46) come
    //      first, ordered by (count, mer) descending; the count-1 k-mers follow in table order: they are looked up and
    //      recruit reads like the others, but nothing depends on their order, so they are not sorted (with sequencing
    //      noise they are > 95 % of the set).
    // Round 6: the table is walked by WAVEFRONTS -- wavefront w owns the slots [w R, (w + 1) R), lane l of it slot base + l of every
    // block of 64 -- instead of a private chunk of consecutive slots per thread.  Same compaction order (ascending slot index inside
    // each class), but a wavefront reads whole lines (a thread's chunk put the 64 lanes of every load on 64 different lines, each
    // fetched sixteen times over: configs[4]'s 8 M slots took 68 of the stage's 163 ms; in LDS the same walk hit one bank 16- to 32-fold).
    uint32_t M = 0, M2 = 0;
    const uint32_t R4 = (((tcap + (uint32_t)nwv - 1) / (uint32_t)nwv) + 63u) / 64u * 64u, lo4 = min(tcap, (uint32_t)wv * R4), hi4 = min(tcap, lo4 + R4);
    uint32_t pre2, pre1;
    {
        uint32_t c2 = 0, c1 = 0, tot1 = 0;
        for (uint32_t bs = lo4; bs < hi4; bs += 64) {
            const uint32_t i = bs + (uint32_t)lane; const bool occ = i < hi4 && tslot[i] < BK_EMPTY32 - 1, two = occ && tcnt[i] >= 2;
            c2 += (uint32_t)__popcll(__ballot(two)); c1 += (uint32_t)__popcll(__ballot(occ && !two));
        }
        pre2 = bk_block_excl_scan(lane == 0 ? c2 : 0u, scr, &M2);      // lane 0 of a wavefront: the sum over the wavefronts before it
        pre1 = bk_block_excl_scan(lane == 0 ? c1 : 0u, scr, &tot1);
        pre2 = (uint32_t)__shfl((int)pre2, 0); pre1 = (uint32_t)__shfl((int)pre1, 0);
        M = M2 + tot1;
    }
    uint32_t npad = 1; while (npad < M2) npad <<= 1;
    if (lds_tab) { perm_lds = tcnt; perm_cap = tcap; }         // the count words are free once the counts are materialised (npad <= tcap)
    const bool use_bucket = npad > perm_cap || ((p.flags & BK_F_BUCKET_SORT) && M2 >= 2);      // large regions: bk_bucket_sort
    const bool perm_in_lds = !use_bucket;                      // else the permutation lives in global memory
    const uint32_t bcap = use_bucket ? npad / 32 + 4096 : 0;   // buckets of bk_bucket_sort: <= M2 / 64 + 256
    const uint32_t tcap2 = [&] { uint32_t c = 64; while (c < 2 * M) c <<= 1; return c; }();      // compact table for the assembler (lds_tab)
    const uint64_t b2 = (uint64_t)M * (8 + 8 + 4 + 1 + 12 + 4 + 4) + (uint64_t)T * 4 + 4096 + (perm_in_lds ? 0 : (uint64_t)npad * 4 + (uint64_t)bcap * 8 + 64) + (lds_tab ? (uint64_t)tcap2 * 4 + 256 : 0);
    uint64_t a1 = bk_arena_alloc(p, b2, scr + 20);
    if (a1 == ~0ull) { if (tid == 0) wk->status = BK_ST_ARENA; return; }
    const uint64_t o_klo = a1, o_khi = o_klo + (uint64_t)M * 8, o_kcnt = o_khi + (uint64_t)M * 8, o_kstamp = bk_align_up(o_kcnt + (uint64_t)M * 4, 16),
                   o_poff = o_kstamp + (uint64_t)M * 12, o_tmp = bk_align_up(o_poff + (uint64_t)(M + 1) * 4, 16), o_post = bk_align_up(o_tmp + (uint64_t)M * 4, 16),
                   o_kstate = o_post + (uint64_t)T * 4, o_perm = bk_align_up(o_kstate + M, 16), o_bst = bk_align_up(o_perm + (perm_in_lds ? 0 : (uint64_t)npad * 4), 16), o_tab2 = bk_align_up(o_bst + (uint64_t)bcap * 8, 256);
    uint64_t *klo = (uint64_t *)(p.arena + o_klo), *khi = (uint64_t *)(p.arena + o_khi);
    uint32_t *kcnt = (uint32_t *)(p.arena + o_kcnt), *poff = (uint32_t *)(p.arena + o_poff), *ptmp = (uint32_t *)(p.arena + o_tmp), *post = (uint32_t *)(p.arena + o_post);
    int32_t *kstamp = (int32_t *)(p.arena + o_kstamp);
    uint8_t *kstate = (uint8_t *)(p.arena + o_kstate);
    // permutation sort: perm in LDS when it fits, else in global memory
    uint32_t *perm = perm_in_lds ? perm_lds : (uint32_t *)(p.arena + o_perm);
    // keys and counts at the compaction index (count >= 2 first); ptmp: slot of the compaction index
    for (uint32_t bs = lo4; bs < hi4; bs += 64) {
        const uint32_t i = bs + (uint32_t)lane; const bool occ = i < hi4 && tslot[i] < BK_EMPTY32 - 1; const uint32_t cn = occ ? tcnt[i] : 0u; const bool two = occ && cn >= 2;
        const unsigned long long m2 = __ballot(two), m1 = __ballot(occ && !two), below = (1ull << lane) - 1ull;
        if (occ) {
            const uint32_t e2 = t_ent[tslot[i]]; const BkKey key = seq_kmer_fast(reads + (uint64_t)urep[e2 >> 10] * RW, (int)RW, (int)(e2 & 1023u), k);
            const uint32_t j = two ? pre2 + (uint32_t)__popcll(m2 & below) : M2 + pre1 + (uint32_t)__popcll(m1 & below);
            klo[j] = key.lo; khi[j] = key.hi; kcnt[j] = cn; ptmp[j] = i;
        }
        pre2 += (uint32_t)__popcll(m2); pre1 += (uint32_t)__popcll(m1);
    }
    BK_SYNC();
    bool sorted = false;
    if (use_bucket) sorted = bk_bucket_sort(perm, M2, klo, khi, kcnt, k, perm_lds, perm_cap, (uint32_t *)(p.arena + o_bst), (uint32_t *)(p.arena + o_bst) + bcap, bcap, scr);
    if (!sorted) {
        for (uint32_t i = tid; i < npad; i += nt) perm[i] = i < M2 ? i : BK_EMPTY32;
        BK_SYNC();
        for (uint32_t sz = 2; sz <= npad; sz <<= 1)
            for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
                for (uint32_t i = tid; i < npad / 2; i += nt) {
                    uint32_t lo = (i / st) * (st * 2) + (i % st), hi2 = lo + st;
                    bool up = ((lo & sz) == 0);
                    uint32_t a = perm[lo], b = perm[hi2];
                    bool a_first;   // should a come before b in the final order?
                    if (a == BK_EMPTY32) a_first = false; else if (b == BK_EMPTY32) a_first = true;
                    else { BkKey ka{khi[a], klo[a]}, kb{khi[b], klo[b]}; a_first = bk_kmer_before(kcnt[a], ka, kcnt[b], kb); }
                    if (a_first != up) { perm[lo] = b; perm[hi2] = a; }
                }
                BK_SYNC();
            }
    }
    // apply the permutation: arrays indexed by compaction index -> arrays indexed by rank, staged through the
    // (still unused) stamp and posting-offset areas so that no element is overwritten before it is read
    {
        uint64_t *s64 = (uint64_t *)kstamp;                     // M*8 of the M*12 stamp bytes
        uint32_t *s32 = poff;                                   // (M+1)*4
        for (uint32_t j = tid; j < M; j += nt) { uint32_t a = j < M2 ? perm[j] : j; s64[j] = klo[a]; s32[j] = kcnt[a]; }
        BK_SYNC();
        for (uint32_t j = tid; j < M; j += nt) { klo[j] = s64[j]; kcnt[j] = s32[j]; }
        BK_SYNC();
        for (uint32_t j = tid; j < M; j += nt) { uint32_t a = j < M2 ? perm[j] : j; s64[j] = khi[a]; s32[j] = ptmp[a]; }
        BK_SYNC();
        // the assembler's lookups go table slot -> rank -> key: the slot itself now holds the rank (the claimant is not needed any more)
        for (uint32_t j = tid; j < M; j += nt) { khi[j] = s64[j]; tslot[s32[j]] = j; ptmp[j] = 0; }
        BK_SYNC();
    }
    for (uint32_t j = tid; j < M; j += nt) {
        BkKey key{khi[j], klo[j]};
        kstate[j] = key_homopolymer(key, k) ? BK_K_REMOVED : BK_K_LIVE;      // kmers.add_kmer (sv_assembly.py:277)
        kstamp[3 * j] = 0; kstamp[3 * j + 1] = 0; kstamp[3 * j + 2] = 0x7FFFFFFF;
    }
    BK_SYNC();
    BK_STAMP(6);
    // ---- P5: posting lists k-mer rank -> (u, pos) -------------------------------------------------
    // list lengths and fill cursors in LDS when they fit (the LDS holds nothing that is still needed), else in ptmp
    // (LDS table: in the count words, which are free by now; else the start of the LDS, which holds nothing that is still needed)
    uint32_t *pcur = lds_tab ? tcnt : (M + 32u + 16u * 16u <= lds_words ? lds + 32 + 16 * 16 : ptmp);
    if (pcur != ptmp) { for (uint32_t j = tid; j < M; j += nt) pcur[j] = 0; BK_SYNC(); }
    // (four occurrences per thread in flight, as in the table fill above: occurrence -> slot -> rank -> counter is a chain of dependent accesses)
    for (uint32_t idx0 = tid; idx0 < T; idx0 += 4 * (uint32_t)nt) {
        uint32_t sl[4], rk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t idx = idx0 + (uint32_t)q * nt; sl[q] = idx < T ? t_sl[idx] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; q++) rk[q] = idx0 + (uint32_t)q * nt < T ? tslot[sl[q]] : BK_EMPTY32;
#pragma unroll
        for (int q = 0; q < 4; q++) if (rk[q] < BK_EMPTY32 - 1) atomicAdd(&pcur[rk[q]], 1u);
    }
    BK_SYNC();
    {
        const uint32_t chunk = (M + nt - 1) / nt, b = tid * chunk, e = min(M, b + chunk);
        uint32_t c = 0, tot;
        for (uint32_t j = b; j < e; j++) c += pcur[j];
        uint32_t pre = bk_block_excl_scan(c, scr, &tot);
        for (uint32_t j = b; j < e; j++) { uint32_t n = pcur[j]; poff[j] = pre; pcur[j] = pre; pre += n; }
        if (tid == 0) poff[M] = tot;
    }
    BK_SYNC();
    for (uint32_t idx0 = tid; idx0 < T; idx0 += 4 * (uint32_t)nt) {
        uint32_t sl[4], rk[4], en[4], at[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t idx = idx0 + (uint32_t)q * nt; sl[q] = idx < T ? t_sl[idx] : 0u; en[q] = idx < T ? t_ent[idx] : 0u; }
#pragma unroll
        for (int q = 0; q < 4; q++) rk[q] = idx0 + (uint32_t)q * nt < T ? tslot[sl[q]] : BK_EMPTY32;
#pragma unroll
        for (int q = 0; q < 4; q++) at[q] = rk[q] < BK_EMPTY32 - 1 ? atomicAdd(&pcur[rk[q]], 1u) : 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) if (rk[q] < BK_EMPTY32 - 1) post[at[q]] = en[q];
    }
    BK_SYNC();
    uint32_t out_tcap = tcap;
    if (lds_tab) {
        // compact table slot -> rank for the assembler's lookups (bk_lookup): tcap2 >= 2 M slots, built in the count words, copied out once
        uint32_t *tab2 = tcnt, *gtab2 = (uint32_t *)(p.arena + o_tab2);
        for (uint32_t i = tid; i < tcap2; i += nt) tab2[i] = BK_EMPTY32;
        BK_SYNC();
        for (uint32_t j = tid; j < M; j += nt) {
            const BkKey key{khi[j], klo[j]};
            uint32_t s2 = key_hash(key) & (tcap2 - 1);
            while (atomicCAS(&tab2[s2], BK_EMPTY32, j) != BK_EMPTY32) s2 = (s2 + 1) & (tcap2 - 1);
        }
        BK_SYNC();
        for (uint32_t i = tid; i < tcap2; i += nt) gtab2[i] = tab2[i];
        o_tslot = o_tab2; out_tcap = tcap2;
    }
    BK_STAMP(7);
    BK_SYNC();
    bk_split_prepare(p, wk, U, M, M2, kcnt, scr);      // noisy regions: several assembler workgroups per region (bk_comp.hip.h)
    if (tid == 0) {
        wk->U = U; wk->T = T; wk->M = M; wk->M2 = M2; wk->tcap = out_tcap;
        wk->o_trip_ent = o_ent; wk->o_trip_slot = o_tsl; wk->o_tslot = o_tslot; wk->o_tcnt = o_tcnt; wk->o_trank = 0;
        wk->o_key_lo = o_klo; wk->o_key_hi = o_khi; wk->o_kcnt = o_kcnt; wk->o_kstate = o_kstate; wk->o_kstamp = o_kstamp;
        wk->o_poff = o_poff; wk->o_post = o_post;
    }
}

extern "C" __global__ void __launch_bounds__(BK_KT_MAX) bk_kmer_kernel(BkParams p, uint32_t ref_cap, uint32_t win_words_cap, uint32_t lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    bk_kmer_body<false>(p, ref_cap, win_words_cap, lds_words, lds);
}
// regions with a window beyond the LDS budget: reference set in global memory (LDS: scratch + BK_K_PERM_G words)
extern "C" __global__ void __launch_bounds__(BK_KT_MAX) bk_kmer_kernel_g(BkParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    bk_kmer_body<true>(p, 0, 0, 32 + 16 * 16 + BK_K_PERM_G, lds);
}
