// bk_asm_kmers.hip.h -- part of the assembler state machine (bk_asm.hip.h includes it, once per workgroup size, inside that build's namespace):
// split regions: what a contig meets across components (bk_comp.hip.h); the contig k-mer lists (get_read_kmers_ordered, sv_assembly.py:126-143); find_reads from the posting lists (:111-122).
// No include guard: like bk_asm.hip.h it is compiled twice (BK_AT = 512 and 256).
#undef BK_SRC_ID
#define BK_SRC_ID 7      // barrier sites of this file (bk_common.h: BK_SYNC; both instances share the site ids)

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
