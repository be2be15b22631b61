// bk_sched.hip.h -- work distribution between the stages.
//
// The reference loops over its targets one after the other (sv_processor.py:185-201); here a batch of regions is in
// flight at once and regions differ in cost by three orders of magnitude (a clean 200 bp deletion: 30 sample k-mers,
// ~10 M DP cells; a translocation whose partner half is all non-reference: 1,500 k-mers, 2 G cells; a noisy region:
// 10^5 k-mers).  One workgroup per region in launch order leaves a batch waiting for whichever heavy region started
// last.  So the k-mer stage's own figures (BkRegionWork.T = non-reference k-mer occurrences in unique reads, M = sample
// k-mers) rank the regions, heaviest first (longest-processing-time-first), and the assembler's workgroups are
// persistent: each pulls the next region of that order from a device-side queue head until the queue is empty.
#pragma once
#include "bk_common.h"
#undef BK_SRC_ID
#define BK_SRC_ID 2      // barrier sites of this file (bk_common.h: BK_SYNC)

#define BK_SCHED_T 1024

// one workgroup: order[] = region ids by (cost descending, id ascending) -- deterministic
extern "C" __global__ void __launch_bounds__(BK_SCHED_T) bk_sched_kernel(BkParams p, unsigned long long *keys /* npad words of scratch */, uint32_t npad, uint32_t asm_grid /* workgroups of the assembler launch that follows */, uint32_t resident /* assembler workgroups the chip holds at once */)
{
    const int tid = threadIdx.x, n = p.n_regions;
    __shared__ uint32_t nsplit_s;
    if (tid == 0) nsplit_s = 0;
    for (uint32_t i = tid; i < npad; i += BK_SCHED_T) {
        unsigned long long key = ~0ull;                                   // padding sorts last
        if ((int)i < n) {
            const BkRegionWork &w = p.work[i];
            // assembler cost grows with the number of read recruitments (T) and of k-mer visits (M); a failed or empty
            // region costs nothing
            unsigned long long cost = w.status == BK_ST_OK ? (unsigned long long)w.T + 4ull * w.M : 0ull;
            if (cost > 0xFFFFFFFFull) cost = 0xFFFFFFFFull;
            key = ((0xFFFFFFFFull - cost) << 32) | i;
        }
        keys[i] = key;
    }
    BK_SYNC();
    for (uint32_t sz = 2; sz <= npad; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < npad / 2; i += BK_SCHED_T) {
                const uint32_t lo = (i / st) * (st * 2) + (i % st), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            BK_SYNC();
        }
    // the queue: one entry per unit -- (region | unit << 24).  First every region once (unit 0 of a split region, bk_comp.hip.h: it
    // runs the serial prefix the other units wait for, so all prefixes of a batch start at once however few workgroups are
    // resident -- with the units of a region behind each other, 64 split regions on 512 resident workgroups ran their prefixes in
    // two waves), then the other units of the split regions, region by region in the same order.
    // How many units a split region gets: BK_SPLIT_G, whatever the batch.  Measured in round 5 (profiles/r05/noisy_units_per_region.txt,
    // regions of the configs[1] shape at 0.5 % noise, 512 resident workgroups): giving a region only as many units as the batch leaves
    // workgroup slots empty (8 / 4 / 2 units for 64 / 128 / 256 noisy regions) was SLOWER than 16 units each with the queue
    // oversubscribed (108 vs 90 ms, 163 vs 134 ms, 420 vs 324 ms per launch; one unit per region: 438 ms for 256): what bounds a
    // noisy batch is the critical path of its slowest regions and the balance of the dealing, not the workgroup time summed up.
    // (`resident` is what the measurement used; the units of a region are a per-region number all the same: wk->split.)
    __shared__ uint32_t wsum[BK_SCHED_T / 64], base_s;
    for (int i = tid; i < n; i += BK_SCHED_T) { p.order[i] = (uint32_t)keys[i]; if (p.work[i].split) atomicAdd(&nsplit_s, 1u); }
    if (tid == 0) base_s = (uint32_t)n;
    BK_SYNC();
    const uint32_t nwant = nsplit_s;
    const uint32_t G = nwant ? (uint32_t)BK_SPLIT_G : 0u; (void)resident;
    // Round 6: the units of a split region beyond the first are NOT entries of the launch any more: unit 0 appends them itself once it
    // has run the serial prefix and labelled the graph (bk_asm.hip.h).  Queued at launch they took workgroup slots and slept until their
    // region's prefix was over -- 2.7 to 36 ms at 0.5 % noise -- while the units of regions that were ready waited behind them in the queue
    // (256 noisy regions: 247 ms of assembler for ~96 ms of work per slot).  BK_F_PREQUEUE_UNITS is the round-5 queue (tests hold both).
    const bool preq = (p.flags & BK_F_PREQUEUE_UNITS) != 0;
    for (int i = tid; i < n; i += BK_SCHED_T) if (p.work[i].split) p.work[i].split = G;
    __threadfence(); BK_SYNC();
    if (tid == 0) nsplit_s = G ? nwant : 0u;
    BK_SYNC();
    for (int c0 = 0; c0 < n; c0 += BK_SCHED_T) {
        const int i = c0 + tid;
        uint32_t rid = 0, g = 0;                                             // g: units beyond the first
        if (i < n) { rid = (uint32_t)keys[i]; const uint32_t sp = p.work[rid].split; g = (sp && preq) ? sp - 1u : 0u; }
        uint32_t inc = g;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((tid & 63) >= o) inc += t; }
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        BK_SYNC();
        uint32_t pre = base_s, tot = 0;
        for (int w = 0; w < BK_SCHED_T / 64; w++) { const uint32_t t = wsum[w]; if (w < (tid >> 6)) pre += t; tot += t; }
        const uint32_t at = pre + inc - g;
        for (uint32_t u = 0; u < g; u++) p.order[at + u] = rid | ((u + 1u) << BK_QUEUE_UNIT_SHIFT);
        BK_SYNC();
        if (tid == 0) base_s += tot;
        BK_SYNC();
    }
    // the queue is dynamic (bk_asm.hip.h): room behind the launch's entries for the units split regions append -- those of the first pass
    // once unit 0 has labelled the graph, those of a repair pass when components met across units (marked empty: an entry is valid once
    // written); a batch without split regions takes its first asm_grid entries by block index
    BK_SYNC();
    const uint32_t nq = base_s, nsp = nsplit_s;
    const unsigned long long room = (preq ? 0ull : (unsigned long long)nsp * (BK_SPLIT_G - 1)) + ((p.flags & BK_F_HOST_REPAIR) ? 0ull : (unsigned long long)nsp * BK_SPLIT_G * BK_REQUEUE_PASSES);
    const uint32_t ext_end = (uint32_t)min((unsigned long long)p.order_cap, (unsigned long long)nq + room);
    for (uint32_t i = nq + tid; i < ext_end; i += BK_SCHED_T) p.order[i] = BK_EMPTY32;
    // A batch WITH split regions hands out every entry through *asm_head, the first one of a workgroup included (n_queue0 = 0): a workgroup
    // that finds the queue empty waits for *pending, and with entries reserved by block index it could wait for a workgroup that is not
    // resident yet -- two assembler kernels of two handles, each partly resident, then wait for each other for ever.  Any
    // resident workgroup can take any entry now; what a taken entry waits for (unit 0's labelling) is held by a workgroup that is running.
    if (tid == 0) { *p.n_queue = nq; *p.n_queue0 = nsp ? 0u : nq; *p.asm_head = nsp ? 0u : min(asm_grid, nq); *p.pending = nsp; *p.queue_cap = ext_end; }
}
