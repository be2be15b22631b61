// bk_index.hip.h -- the genome-wide seed look-up (N4; included by bk_api.hip): bk_index_create / _probe / _set_loci / _find / _destroy and their
// two kernels.  The step behind the reference's whole-genome gfServer fallback (sv_processor.py:829-831, utils.py:620-657); a stand-in, unpinned.

// ---- genome-wide seed lookup: sorted sampled k-mer codes in HBM, one binary search pair per query k-mer (N4; include/breakmer_hip.h)
struct bk_index { int dev = 0; hipStream_t stream = nullptr; DevBuf d_codes, d_q, d_lo, d_hi; uint64_t n = 0; hipEvent_t ev[2] = {};
                  DevBuf d_seqno, d_pos, d_ok, d_keys, d_runs, d_out, d_meta; bool have_loci = false; };
// A thread per query: lower and upper bound in the sorted codes.  The first ~12 levels of every search touch the same few
// hundred lines (L2-resident); the rest are one dependent HBM access each -- bound by the latency of ~2 x 16 of them per query,
// hidden by the other queries in flight.
extern "C" __global__ void __launch_bounds__(256) bk_index_probe_kernel(const uint32_t *codes, uint64_t n, const uint32_t *q, uint64_t nq, uint32_t *lo, uint32_t *hi)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const uint32_t key = q[i];
    uint64_t a = 0, b = n;
    while (a < b) { const uint64_t m = (a + b) >> 1; if (codes[m] < key) a = m + 1; else b = m; }
    const uint64_t l = a;
    b = n;
    while (a < b) { const uint64_t m = (a + b) >> 1; if (codes[m] <= key) a = m + 1; else b = m; }
    lo[i] = (uint32_t)l; hi[i] = (uint32_t)a;
}
// ---- loci of a query sequence in the genome, on the device (N4: the clustering the host did in numpy until round 4).  ONE workgroup per
// call: (A) the index range of every query k-mer (two binary searches, as bk_index_probe_kernel); k-mers that occur more than
// max_occ times are repeats and do not count; (B) prefix sum of the range lengths; (C) one 64-bit key per index hit:
// [sequence number : 16 | diagonal (index position - query position, biased) : 33 | query position : 15] -- ascending keys = hits by
// (sequence, diagonal, position), what the reference-side host code sorted tuples by; (D) bitonic sort of the keys in global memory;
// (E) a locus starts where the sequence changes or the diagonal jumps by more than `band`; (F) loci with >= min_hits hits
// (BLAT's -minMatch=2 for min_hits = 2), in key order, with the first and last index position they cover.
struct BkLocus { uint32_t hits, seqno, start, end; };
#define BK_FIND_T 1024
extern "C" __global__ void __launch_bounds__(BK_FIND_T) bk_index_find_kernel(const uint32_t *codes, const uint16_t *seqno, const uint32_t *pos, uint64_t n,
        const uint32_t *q, const uint8_t *ok, uint32_t nq, uint32_t max_occ, uint32_t band, uint32_t min_hits,
        uint32_t *lo, uint32_t *cnt, unsigned long long *keys, uint32_t key_cap, uint32_t *runs, BkLocus *out, uint32_t out_cap, uint32_t *meta /* [0] hits, [1] loci, [2] 1 = key_cap too small */)
{
    __shared__ uint32_t scr[24];
    const uint32_t tid = threadIdx.x, nt = BK_FIND_T;
    const uint32_t chunk = (nq + nt - 1) / nt, b0 = min(nq, tid * chunk), e0 = min(nq, b0 + chunk);
    uint32_t mine = 0;
    for (uint32_t i = b0; i < e0; i++) {
        const uint32_t key = q[i];
        uint64_t a = 0, b = n;
        while (a < b) { const uint64_t m = (a + b) >> 1; if (codes[m] < key) a = m + 1; else b = m; }
        const uint64_t l = a;
        b = n;
        while (a < b) { const uint64_t m = (a + b) >> 1; if (codes[m] <= key) a = m + 1; else b = m; }
        const uint32_t c = (ok[i] && a > l && a - l <= (uint64_t)max_occ) ? (uint32_t)(a - l) : 0u;
        lo[i] = (uint32_t)l; cnt[i] = c; mine += c;
    }
    uint32_t H;
    uint32_t w = bk_block_excl_scan(mine, scr, &H);
    if (tid == 0) { meta[0] = H; meta[1] = 0; meta[2] = H > key_cap ? 1u : 0u; }
    if (H > key_cap || H == 0) return;                                   // uniform
    uint32_t npad = 1; while (npad < H) npad <<= 1;
    for (uint32_t i = b0; i < e0; i++)
        for (uint32_t j = 0; j < cnt[i]; j++) {
            const uint32_t en = lo[i] + j;
            const unsigned long long dg = (unsigned long long)pos[en] + 32768ull - (unsigned long long)i;
            keys[w++] = ((unsigned long long)seqno[en] << 48) | (dg << 15) | (unsigned long long)i;
        }
    for (uint32_t i = H + tid; i < npad; i += nt) keys[i] = ~0ull;
    __threadfence(); BK_SYNC();
    for (uint32_t sz = 2; sz <= npad; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < npad / 2; i += nt) {
                const uint32_t l2 = (i / st) * (st * 2) + (i % st), h2 = l2 + st;
                const bool up = (l2 & sz) == 0;
                const unsigned long long a = keys[l2], b = keys[h2];
                if ((a > b) == up) { keys[l2] = b; keys[h2] = a; }
            }
            __threadfence(); BK_SYNC();
        }
    // (E) run starts, compacted in order
    auto is_start = [&](uint32_t i) -> bool {
        if (i == 0) return true;
        const unsigned long long a = keys[i - 1], b = keys[i];
        return (a >> 48) != (b >> 48) || ((b >> 15) & 0x1FFFFFFFFull) - ((a >> 15) & 0x1FFFFFFFFull) > (unsigned long long)band;
    };
    const uint32_t hch = (H + nt - 1) / nt, hb = min(H, tid * hch), he = min(H, hb + hch);
    uint32_t ns = 0;
    for (uint32_t i = hb; i < he; i++) ns += is_start(i) ? 1u : 0u;
    uint32_t nruns;
    uint32_t rw = bk_block_excl_scan(ns, scr, &nruns);
    for (uint32_t i = hb; i < he; i++) if (is_start(i)) runs[rw++] = i;
    if (tid == 0) runs[nruns] = H;
    __threadfence(); BK_SYNC();
    // (F) the loci with enough hits, in key order
    const uint32_t rch = (nruns + nt - 1) / nt, rb = min(nruns, tid * rch), re = min(nruns, rb + rch);
    uint32_t nk = 0;
    for (uint32_t r = rb; r < re; r++) nk += (runs[r + 1] - runs[r] >= min_hits) ? 1u : 0u;
    uint32_t nloci;
    uint32_t ow = bk_block_excl_scan(nk, scr, &nloci);
    for (uint32_t r = rb; r < re; r++) {
        const uint32_t a = runs[r], e = runs[r + 1];
        if (e - a < min_hits) continue;
        uint32_t pmin = 0xFFFFFFFFu, pmax = 0;
        for (uint32_t i = a; i < e; i++) { const unsigned long long k = keys[i]; const uint32_t ps = (uint32_t)(((k >> 15) & 0x1FFFFFFFFull) + (k & 0x7FFFull) - 32768ull); pmin = min(pmin, ps); pmax = max(pmax, ps); }
        if (ow < out_cap) { BkLocus L; L.hits = e - a; L.seqno = (uint32_t)(keys[a] >> 48); L.start = pmin; L.end = pmax; out[ow] = L; }
        ow++;
    }
    if (tid == 0) meta[1] = nloci;
}
extern "C" int bk_index_set_loci(bk_index *ix, const uint16_t *seqno, const uint32_t *pos)
{
    if (!ix || (ix->n && (!seqno || !pos))) return BK_E_ARG;
    if (hipSetDevice(ix->dev) != hipSuccess) return BK_E_HIP;
    if (ix->d_seqno.ensure(std::max<uint64_t>(ix->n, 1) * 2) != hipSuccess || ix->d_pos.ensure(std::max<uint64_t>(ix->n, 1) * 4) != hipSuccess) return BK_E_NOMEM;
    if (ix->n && (hipMemcpy(ix->d_seqno.p, seqno, ix->n * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(ix->d_pos.p, pos, ix->n * 4, hipMemcpyHostToDevice) != hipSuccess)) return BK_E_HIP;
    ix->have_loci = true;
    return BK_OK;
}
extern "C" int bk_index_find(bk_index *ix, const uint32_t *queries, const uint8_t *ok, uint32_t n_queries, uint32_t max_occ, uint32_t band, uint32_t min_hits,
                             bk_locus *loci, uint32_t cap, uint32_t *n_loci, float *kernel_ms)
{
    static_assert(sizeof(bk_locus) == sizeof(BkLocus), "bk_locus layout");
    if (!ix || !n_loci || (n_queries && (!queries || !ok)) || (cap && !loci)) return BK_E_ARG;
    if (!ix->have_loci) return BK_E_STATE;
    if (n_queries > 32768u) return BK_E_LIMIT;                            // the query position takes 15 bits of the sort key
    *n_loci = 0;
    if (kernel_ms) *kernel_ms = 0.0f;
    if (!n_queries || !ix->n) return BK_OK;
    if (hipSetDevice(ix->dev) != hipSuccess) return BK_E_HIP;
    if (ix->d_q.ensure((size_t)n_queries * 4) != hipSuccess || ix->d_ok.ensure(n_queries) != hipSuccess || ix->d_lo.ensure((size_t)n_queries * 4) != hipSuccess || ix->d_hi.ensure((size_t)n_queries * 4) != hipSuccess ||
        ix->d_meta.ensure(64) != hipSuccess) return BK_E_NOMEM;
    if (hipMemcpyAsync(ix->d_q.p, queries, (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess || hipMemcpyAsync(ix->d_ok.p, ok, n_queries, hipMemcpyHostToDevice, ix->stream) != hipSuccess) return BK_E_HIP;
    uint32_t key_cap = 1; while (key_cap < std::max<uint32_t>(4096u, 4u * n_queries)) key_cap <<= 1;      // a power of two: the sort pads to one
    uint32_t meta[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 3; attempt++) {
        const uint32_t out_cap = min_hits >= 2 ? key_cap / 2 + 1 : key_cap + 1;      // a locus has >= min_hits of the <= key_cap hits (min_hits < 2: every hit may be one)
        if (ix->d_keys.ensure((size_t)key_cap * 8) != hipSuccess || ix->d_runs.ensure(((size_t)key_cap + 1) * 4) != hipSuccess || ix->d_out.ensure((size_t)out_cap * sizeof(BkLocus)) != hipSuccess) return BK_E_NOMEM;
        (void)hipEventRecord(ix->ev[0], ix->stream);
        hipLaunchKernelGGL(bk_index_find_kernel, dim3(1), dim3(BK_FIND_T), 0, ix->stream, (const uint32_t *)ix->d_codes.p, (const uint16_t *)ix->d_seqno.p, (const uint32_t *)ix->d_pos.p, ix->n,
                           (const uint32_t *)ix->d_q.p, (const uint8_t *)ix->d_ok.p, n_queries, max_occ, band, min_hits, (uint32_t *)ix->d_lo.p, (uint32_t *)ix->d_hi.p,
                           (unsigned long long *)ix->d_keys.p, key_cap, (uint32_t *)ix->d_runs.p, (BkLocus *)ix->d_out.p, out_cap, (uint32_t *)ix->d_meta.p);
        if (hipGetLastError() != hipSuccess) return BK_E_HIP;
        (void)hipEventRecord(ix->ev[1], ix->stream);
        if (hipMemcpyAsync(meta, ix->d_meta.p, sizeof(meta), hipMemcpyDeviceToHost, ix->stream) != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess) return BK_E_HIP;
        if (!meta[2]) {
            if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ix->ev[0], ix->ev[1]);
            *n_loci = meta[1];
            const uint32_t take = std::min(std::min(meta[1], cap), out_cap);
            if (take && hipMemcpy(loci, ix->d_out.p, (size_t)take * sizeof(BkLocus), hipMemcpyDeviceToHost) != hipSuccess) return BK_E_HIP;
            return BK_OK;
        }
        while (key_cap < meta[0]) key_cap <<= 1;                          // more index hits than keys: as many as it takes (max_occ x queries at most)
    }
    return BK_E_NOMEM;
}
extern "C" int bk_index_create(int device_id, const uint32_t *sorted_codes, uint64_t n, bk_index **out)
{
    if (!out || (n && !sorted_codes) || n >= (1ull << 32)) return fail(nullptr, BK_E_ARG, "bk_index_create: bad argument (at most 2^32 - 1 entries)");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return fail(nullptr, BK_E_NOGPU, "bk_index_create: no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= nd) return fail(nullptr, BK_E_ARG, "bk_index_create: bad device id");
    bk_index *ix = new bk_index(); ix->dev = device_id; ix->n = n;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ix->ev[0]) != hipSuccess || hipEventCreate(&ix->ev[1]) != hipSuccess ||
        ix->d_codes.ensure(std::max<uint64_t>(n, 1) * 4) != hipSuccess || (n && hipMemcpy(ix->d_codes.p, sorted_codes, n * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)bk_index_destroy(ix); return fail(nullptr, BK_E_HIP, "bk_index_create: device allocation / copy failed");
    }
    *out = ix;
    return BK_OK;
}
extern "C" int bk_index_probe(bk_index *ix, const uint32_t *queries, uint64_t nq, uint32_t *lo, uint32_t *hi, float *kernel_ms)
{
    if (!ix || (nq && (!queries || !lo || !hi))) return BK_E_ARG;
    if (!nq) return BK_OK;
    if (hipSetDevice(ix->dev) != hipSuccess) return BK_E_HIP;
    if (ix->d_q.ensure(nq * 4) != hipSuccess || ix->d_lo.ensure(nq * 4) != hipSuccess || ix->d_hi.ensure(nq * 4) != hipSuccess) return BK_E_NOMEM;
    if (hipMemcpyAsync(ix->d_q.p, queries, nq * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess) return BK_E_HIP;
    (void)hipEventRecord(ix->ev[0], ix->stream);
    hipLaunchKernelGGL(bk_index_probe_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ix->stream, (const uint32_t *)ix->d_codes.p, ix->n, (const uint32_t *)ix->d_q.p, nq, (uint32_t *)ix->d_lo.p, (uint32_t *)ix->d_hi.p);
    if (hipGetLastError() != hipSuccess) return BK_E_HIP;
    (void)hipEventRecord(ix->ev[1], ix->stream);
    if (hipMemcpyAsync(lo, ix->d_lo.p, nq * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess || hipMemcpyAsync(hi, ix->d_hi.p, nq * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) return BK_E_HIP;
    if (hipStreamSynchronize(ix->stream) != hipSuccess) return BK_E_HIP;
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ix->ev[0], ix->ev[1]);
    return BK_OK;
}
extern "C" int bk_index_destroy(bk_index *ix)
{
    if (!ix) return BK_OK;
    (void)hipSetDevice(ix->dev);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    ix->d_codes.release(); ix->d_q.release(); ix->d_lo.release(); ix->d_hi.release();
    ix->d_seqno.release(); ix->d_pos.release(); ix->d_ok.release(); ix->d_keys.release(); ix->d_runs.release(); ix->d_out.release(); ix->d_meta.release();
    for (auto &e : ix->ev) if (e) (void)hipEventDestroy(e);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
    return BK_OK;
}
