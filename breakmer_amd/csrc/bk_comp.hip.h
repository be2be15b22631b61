// bk_comp.hip.h -- splitting a noisy region over several assembler workgroups ("units").
//
// init_assembly (sv_assembly.py:30-63) is a serial chain over the seed k-mers: with sequencing noise a region has
// thousands of seed iterations (an error shared by two or three reads each) and ONE workgroup walks them one after the
// other -- 0.2 s per region at 0.5 % substitutions whatever else the chip does.  But an iteration only ever touches the
// reads that hold its seed k-mer, the k-mers of the contig those reads build, the reads that hold those, and so on: the
// CONNECTED COMPONENT of its seed in the graph {unique reads} -- {sample k-mers} (edge = the read holds the k-mer;
// homopolymer k-mers, which the assembler never uses, left out).  Iterations in different components read and write
// disjoint state and commute; the order of the contigs in the result follows from their seed ranks.  Up to ~0.5 % noise
// that graph has hundreds to a thousand components with seeds (the SV's own is the largest, < 10 % of the seed k-mers; at 1 % it
// percolates), so BK_SPLIT_G units take disjoint sets of components and each walks its seeds in rank order -- exactly the
// serial order restricted to what it owns.
//
// Two things the static graph of the k-mer stage does not show:
//  * The SV's own component is a third of all seed iterations, and it would have to run on one unit.  But its k-mers are the
//    ones with the high counts: unit 0 first runs the seeds with a count >= BK_SPLIT_HI alone and in order (a handful of
//    iterations: the SV's contigs), and the graph is labelled AFTERWARDS, inside the assembler, on what is still live
//    (bk_label_live) -- with its junction k-mers used up the SV's component falls apart into error clusters like everything
//    else (largest component 8 % of the seed k-mers before, 1 % after).  The components are dealt to the units by size.
//  * A contig is built from PIECES of reads, and a k-mer across the seam of two pieces may belong to a read of another component
//    (measured with the oracle at 0.5 %: ~35 such meetings per region).  So the assembler checks every contig k-mer
//    (bk_kmers_ordered) whatever its state says (the other unit may be ahead in seed order): a k-mer of
//      - a component of the same unit          -> fine (the unit IS the serial order over everything it owns); noted as a merge;
//      - a component without seeds (no unit)   -> claimed by this unit (atomic), then as above;
//      - a component of another unit           -> CONFLICT: the unit gives the current component up (nothing of the other
//                                                 unit's state has been touched), notes the pair and goes on with its others.
//    After the pass, bk_resolve_kernel merges the components that met, resets every merged set that holds a conflict to the
//    state of the labelling (reads, k-mers; its contigs are dropped) and they run again in the next pass -- a few per cent of
//    the work, itself spread over the units.  Passes repeat until none is left (components only ever merge).
// bk_link_kernel finally orders the surviving contigs by (seed rank, emission order) = the order init_assembly returns them in.
// Results are bit-identical to the one-unit run (tests: every noisy fixture with BK_F_NO_SPLIT on and off).
#pragma once
#include "bk_common.h"
#undef BK_SRC_ID
#define BK_SRC_ID 4      // barrier sites of this file (bk_common.h: BK_SYNC)

#define BK_SPLIT_MIN_SEEDS 1024        // regions with fewer seed k-mers stay one unit (a clean SV has 30)

__device__ inline uint32_t bk_ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lock-free union-find on read indices: roots point to themselves, a root is only ever hooked under a SMALLER root
__device__ inline uint32_t bk_uf_find(uint32_t *par, uint32_t x)
{
    for (;;) {
        const uint32_t p = bk_ld_agent(&par[x]);
        if (p == x) return x;
        const uint32_t gp = bk_ld_agent(&par[p]);
        if (gp != p) atomicMin(&par[x], gp);                     // path halving (an ancestor: never wrong)
        x = p;
    }
}
__device__ inline void bk_uf_union(uint32_t *par, uint32_t a, uint32_t b)
{
    for (;;) {
        a = bk_uf_find(par, a); b = bk_uf_find(par, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }       // a > b: hook a under b
        if (atomicCAS(&par[a], a, b) == a) return;
    }
}

// Called by every thread of the k-mer workgroup at the end of bk_kmer_body: decides whether the region is split and allocates what
// the split needs.  The components themselves are labelled later, inside the assembler (bk_label_live), AFTER unit 0 has run
// the seeds with a count >= BK_SPLIT_HI alone and in order: those are the SV's own k-mers, and the component that holds them
// is a third of all seed iterations -- once its junction k-mers are used up, what is left of it falls apart into error clusters
// like everything else (largest component 8 % of the seed k-mers before, 1 % after).
// M2 = number of seed-capable k-mers (ranks 0 .. M2-1, count >= 2).
__device__ inline void bk_split_prepare(const BkParams &p, BkRegionWork *wk, uint32_t U, uint32_t M, uint32_t M2, const uint32_t *kcnt, uint32_t *scr)
{
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { wk->split = 0; wk->pass = 0; wk->phase = 0; wk->serial_base = 0; wk->stamp_base = 0; wk->units_done = 0; wk->n_cidx = 0; wk->n_pairs = 0; wk->n_conf = 0; }
    if (tid < BK_SPLIT_G) { wk->unit_us[tid] = 0; wk->unit_iters[tid] = 0; }
    if (tid >= 12 && tid < 20) wk->stamps[tid] = 0;      // (check / guard-band words of the diagnostic builds; [0..11] hold the k-mer kernel's phase stamps in a stamps build)
    if ((p.flags & BK_F_NO_SPLIT) || p.rmap || U < 4 || (M2 < BK_SPLIT_MIN_SEEDS && !(p.flags & BK_F_SPLIT_ALWAYS)) || M2 < 2) return;      // uniform
    // the serial prefix must be short: the seeds are ordered by count, so the first rank below BK_SPLIT_HI says how many it has (a
    // deep noisy region -- 2,000x at 5 %: most error k-mers are seen eight times -- would run serially anyway while fifteen
    // workgroups wait for it)
    { uint32_t lo = 0, hi = M2; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kcnt[mid] >= BK_SPLIT_HI) lo = mid + 1; else hi = mid; }
      if (20ull * lo > M2 && !(p.flags & BK_F_SPLIT_ALWAYS)) return; }                 // uniform: every thread does the same search
    uint32_t cidx_cap = 64; while (cidx_cap < 2 * U + 64) cidx_cap <<= 1;              // power of two: sorted in place by bk_link_kernel
    const uint32_t pairs_cap = 4096;
    const uint64_t b_r = bk_align_up((uint64_t)U * 4, 256), b_k = bk_align_up((uint64_t)M * 4, 256), b_x = (uint64_t)cidx_cap * 16, b_p = bk_align_up((uint64_t)pairs_cap * 12, 256), b_f = bk_align_up((uint64_t)U, 256);
    const uint64_t a0 = bk_arena_alloc(p, 2 * b_r + b_k + b_x + b_p + b_f, scr + 20);
    if (a0 == ~0ull) { if (tid == 0) wk->status = BK_ST_ARENA; return; }
    if (tid == 0) {
        wk->o_rroot = a0; wk->o_cinfo = a0 + b_r; wk->o_kroot = a0 + 2 * b_r; wk->o_cidx = a0 + 2 * b_r + b_k; wk->o_pairs = a0 + 2 * b_r + b_k + b_x;
        wk->cidx_cap = cidx_cap; wk->pairs_cap = pairs_cap; wk->split = BK_SPLIT_G;      // "may be split": bk_sched_kernel sets the number of units (2 .. BK_SPLIT_G) or 0 (the batch fills the chip as it is)
    }
}
// the snapshot of the read flags at the moment the graph was labelled (behind the pairs and the inbox): what a component is reset to
__device__ inline uint8_t *bk_ufl0(const BkParams &p, const BkRegionWork *wk) { return p.arena + wk->o_pairs + bk_align_up((uint64_t)wk->pairs_cap * 12, 256); }

// ---- after a pass in which components met across units: merge what met, reset and re-deal the sets that hold a conflict ----
// Executed by ONE workgroup of nt threads (every thread calls it) for region r: by the last unit of the region to report in, inside
// the assembler (round 5: a region repairs itself while the stragglers of the batch still run), or by bk_resolve_kernel (the
// host-driven fallback when the unit queue has no room for another pass).
__device__ inline void bk_resolve_region(const BkParams &p, const uint32_t r, const uint32_t tid, const uint32_t nt)
{
    BkRegionWork *wk = &p.work[r];
    const BkRegionDesc d = p.desc[r];
    const uint32_t U = wk->U, M = wk->M;
    uint32_t *rroot = (uint32_t *)(p.arena + wk->o_rroot), *cinfo = (uint32_t *)(p.arena + wk->o_cinfo), *kroot = (uint32_t *)(p.arena + wk->o_kroot);
    const uint32_t *pairs = (const uint32_t *)(p.arena + wk->o_pairs);
    const uint32_t np = min(bk_ld_agent(&wk->n_pairs), wk->pairs_cap), pass = wk->pass + 1;
    // 1. everything that met becomes one component (merges inside a unit included: their contigs mixed their reads)
    for (uint32_t i = tid; i < np; i += nt) bk_uf_union(rroot, pairs[3 * i], pairs[3 * i + 1]);
    __threadfence(); BK_SYNC();
    // 2. a merged set that holds a conflict runs again (so does a component its unit gave up: it is always in one)
    for (uint32_t i = tid; i < np; i += nt) if (pairs[3 * i + 2]) atomicOr(&cinfo[bk_uf_find(rroot, pairs[3 * i])], BK_CI_REDO);
    for (uint32_t u = tid; u < U; u += nt) if (bk_ld_agent(&cinfo[u]) & BK_CI_ABORT) atomicOr(&cinfo[bk_uf_find(rroot, u)], BK_CI_REDO);
    __threadfence(); BK_SYNC();
    for (uint32_t u = tid; u < U; u += nt) atomicMin(&rroot[u], bk_uf_find(rroot, u));
    __threadfence(); BK_SYNC();
    for (uint32_t j = tid; j < M; j += nt) { const uint32_t k0 = kroot[j]; if (k0 != BK_EMPTY32) kroot[j] = bk_ld_agent(&rroot[k0]); }
    __threadfence(); BK_SYNC();
    // 3. reset the state the assembler keeps per read and per k-mer (as the k-mer stage left it) for what runs again
    uint8_t *ufl = p.uflag + d.read_meta_off; int32_t *ubuf = p.ubuf + d.read_meta_off, *ureads = p.ureads + d.read_meta_off, *ufound = p.ufound + d.read_meta_off, *uminpos = p.uminpos + d.read_meta_off;
    uint8_t *kstate = p.arena + wk->o_kstate; int32_t *kstamp = (int32_t *)(p.arena + wk->o_kstamp);
    const uint8_t *ufl0 = bk_ufl0(p, wk);
    for (uint32_t u = tid; u < U; u += nt) {
        if (!(bk_ld_agent(&cinfo[bk_ld_agent(&rroot[u])]) & BK_CI_REDO)) continue;
        ufl[u] = ufl0[u]; ubuf[u] = 0; ureads[u] = 0; ufound[u] = -1; uminpos[u] = 0x7FFFFFFF;          // as unit 0 left them when it labelled the graph
    }
    for (uint32_t j = tid; j < M; j += nt) {
        const uint32_t k0 = kroot[j];
        if (k0 == BK_EMPTY32 || !(bk_ld_agent(&cinfo[k0]) & BK_CI_REDO)) continue;
        kstate[j] = BK_K_LIVE; kstamp[3 * j] = 0; kstamp[3 * j + 1] = 0; kstamp[3 * j + 2] = 0x7FFFFFFF;
    }
    __threadfence(); BK_SYNC();
    // 4. deal them to the units of the next pass
    for (uint32_t u = tid; u < U; u += nt) {
        if (bk_ld_agent(&rroot[u]) != u) { cinfo[u] = BK_CI_NOUNIT; continue; }            // no longer a root: its word means nothing (and must not look given up next time)
        const uint32_t ci = bk_ld_agent(&cinfo[u]);
        if (ci & BK_CI_REDO) cinfo[u] = (uint32_t)(mix64(0xD1B54A32D192ED03ull * (pass + 1) ^ u) % wk->split) | (pass << 8) | BK_CI_ACTIVE;
    }
    __threadfence(); BK_SYNC();
    if (tid < BK_SPLIT_G) { wk->unit_us[tid] = 0; wk->unit_iters[tid] = 0; }
    if (tid == 0) { wk->pass = pass; wk->n_pairs = 0; wk->n_conf = 0; wk->units_done = 0; wk->status = pass >= 200 ? BK_ST_UNSPLIT : BK_ST_OK; }
    __threadfence(); BK_SYNC();
}
#define BK_RESOLVE_T 512
extern "C" __global__ void __launch_bounds__(BK_RESOLVE_T) bk_resolve_kernel(BkParams p, const uint32_t *list)
{
    bk_resolve_region(p, list[blockIdx.x], threadIdx.x, BK_RESOLVE_T);
}

// ---- the contigs of a split region in the order init_assembly returns them: by (seed rank, emission order); contigs of
//      components that ran again are dropped (their pass is not the component's last).  One workgroup of nt threads; red: >= 16 words
//      of LDS.  Called by the last unit of a region that settles inside the assembler, or by bk_link_kernel (host-driven passes).
__device__ inline void bk_link_region(const BkParams &p, const uint32_t r, const uint32_t tid, const uint32_t nt, uint32_t *red)
{
    BkRegionWork *wk = &p.work[r];
    const uint32_t *rroot = (const uint32_t *)(p.arena + wk->o_rroot), *cinfo = (const uint32_t *)(p.arena + wk->o_cinfo);
    unsigned long long *key = (unsigned long long *)(p.arena + wk->o_cidx), *off = key + wk->cidx_cap;
    const uint32_t n = bk_ld_agent(&wk->n_cidx);
    if (n > wk->cidx_cap) { if (tid == 0) wk->status = BK_ST_UNSPLIT; return; }             // uniform
    uint32_t npad = 1; while (npad < n) npad <<= 1;
    // dead contigs sort last
    for (uint32_t i = tid; i < npad; i += nt) {
        if (i >= n || key[i] == ~0ull) { key[i] = ~0ull; continue; }      // padding, or dead since an earlier pass (the sort of that pass mixed both behind the live ones: such an entry may hold no record offset at all)
        const BkContigRec *c = (const BkContigRec *)(p.out + off[i]);
        if (c->root != BK_EMPTY32 && (cinfo[rroot[c->root]] & 0xFFFFu) != c->pass) key[i] = ~0ull;            // made by the unit and pass that hold the component now (no component: unit 0's serial prefix)
    }
    __threadfence(); BK_SYNC();
    for (uint32_t sz = 2; sz <= npad; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < npad / 2; i += nt) {
                const uint32_t lo = (i / st) * (st * 2) + (i % st), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; const unsigned long long oa = off[lo]; off[lo] = off[hi]; off[hi] = oa; }
            }
            __threadfence(); BK_SYNC();
        }
    uint32_t live = 0;
    for (uint32_t i = tid; i < n; i += nt) {
        if (key[i] == ~0ull) continue;
        live++;
        BkContigRec *c = (BkContigRec *)(p.out + off[i]);
        c->next = (i + 1 < n && key[i + 1] != ~0ull) ? off[i + 1] : 0ull;
    }
    for (int o = 32; o > 0; o >>= 1) live += __shfl_xor(live, o);
    BK_SYNC();
    if ((tid & 63) == 0) red[tid >> 6] = live;
    BK_SYNC();
    if (tid == 0) {
        uint32_t tot = 0; for (uint32_t w = 0; w < (nt + 63) / 64; w++) tot += red[w];
        wk->n_contigs = tot; wk->o_first_contig = tot ? off[0] : 0ull; wk->o_last_contig = tot ? off[tot - 1] : 0ull;
    }
    BK_SYNC();
}
#define BK_LINK_T 256
extern "C" __global__ void __launch_bounds__(BK_LINK_T) bk_link_kernel(BkParams p, int n_regions)
{
    __shared__ uint32_t red[BK_LINK_T / 64];
    for (int r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const BkRegionWork *wk = &p.work[r];
        if (!wk->split || wk->status != BK_ST_OK) continue;                            // uniform
        bk_link_region(p, (uint32_t)r, threadIdx.x, BK_LINK_T, red);
    }
}
