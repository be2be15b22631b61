// bk_asm_units.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// units of a split region (bk_comp.hip.h): the seed list of a unit, and unit 0's labelling of what is left of the read / k-mer graph after the serial prefix.
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 12      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

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
