// bk_submit.hip.h -- host side of bk_submit_regions[_ex] (included by bk_api.hip, which owns bk_handle): 2-bit packing of the caller's
// sequences (table and SSSE3 paths), the team of helper threads that fills the pinned staging buffer, the chunked host-to-device copies,
// the kernel that lays the rows out with their pad word, and the region descriptors.  Replaces the FASTA / FASTQ text the reference
// writes for Jellyfish and the assembler (sv_processor.py:584-606, utils.py:355-381).  No include guard needed: included once.

// 2 bit/base, first base in the most significant bits (bk_common.h).  nlist != nullptr: 'N' is accepted (packed as code 0)
// and its position appended as (tag << 10 | position); any other character fails.
// One table look-up per base: code in bits 0-1, bit 2 = N, bit 3 = invalid; 16 bases are folded into a word without a
// branch and the flag bits of the whole word are tested once.
struct BkPackLut { uint8_t v[256]; BkPackLut() { for (int i = 0; i < 256; i++) v[i] = 8; v[(int)'A'] = 0; v[(int)'C'] = 1; v[(int)'G'] = 2; v[(int)'T'] = 3; v[(int)'N'] = 4; } };
static const BkPackLut g_pack_lut;
// reference / partner windows: soft-masked (lower-case) bases are the same bases (BLAT's -repeats=lower only reports matches on them separately)
struct BkWinLut { uint8_t v[256]; BkWinLut() { for (int i = 0; i < 256; i++) v[i] = 8; const char *u = "ACGTN", *l = "acgtn"; for (int i = 0; i < 5; i++) { v[(int)u[i]] = (uint8_t)i; v[(int)l[i]] = (uint8_t)i; } } };
static const BkWinLut g_win_lut;
struct BkCodeLut { uint8_t v[256]; BkCodeLut() { for (int i = 0; i < 256; i++) v[i] = 8; for (int i = 0; i < 5; i++) v[i] = (uint8_t)i; } };   // bytes are base codes 0..3, 4 = N
static const BkCodeLut g_code_lut;
// 16 bases -> one word with SSSE3 (two multiply-adds fold 16 two-bit codes, one byte shuffle orders them); returns false
// when the block holds anything but A/C/G/T (codes 0..3): the caller then takes the table path for that block.
#include <immintrin.h>
__attribute__((target("ssse3"))) static inline bool pack16_ssse3(const unsigned char *u, bool codes, uint32_t *out)
{
    __m128i v = _mm_loadu_si128((const __m128i *)u);
    if (codes) {
        const __m128i three = _mm_set1_epi8(3);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_max_epu8(v, three), three)) != 0xFFFF) return false;
    } else {
        const __m128i c = _mm_cmpeq_epi8(v, _mm_set1_epi8('C')), g = _mm_cmpeq_epi8(v, _mm_set1_epi8('G')), t = _mm_cmpeq_epi8(v, _mm_set1_epi8('T'));
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('A')), c), _mm_or_si128(g, t));
        if (_mm_movemask_epi8(ok) != 0xFFFF) return false;
        v = _mm_or_si128(_mm_and_si128(c, _mm_set1_epi8(1)), _mm_or_si128(_mm_and_si128(g, _mm_set1_epi8(2)), _mm_and_si128(t, _mm_set1_epi8(3))));
    }
    const __m128i p = _mm_maddubs_epi16(v, _mm_set1_epi16(0x0104));                 // b[2i]*4 + b[2i+1]
    const __m128i q = _mm_madd_epi16(p, _mm_set1_epi32(0x00010010));                // p[2j]*16 + p[2j+1]: 4 bases per 32-bit lane, first base on top
    const __m128i r = _mm_shuffle_epi8(q, _mm_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    *out = (uint32_t)_mm_cvtsi128_si32(r);
    return true;
}
static const bool g_have_ssse3 = __builtin_cpu_supports("ssse3");

// a row of packed words (2.5 million of ~40 bytes per batch: a library memcpy call per row costs more than the copy)
// (the source is the caller's row: bk_region.read_stride is in BYTES and promises no alignment, so it is read as bytes)
static inline void copy_words(uint32_t *dst, const unsigned char *src, uint32_t nw)
{
    uint32_t w = 0;
    for (; w + 4 <= nw; w += 4) _mm_storeu_si128((__m128i *)(dst + w), _mm_loadu_si128((const __m128i *)(src + 4 * (size_t)w)));
    if (w + 2 <= nw) { uint64_t v; memcpy(&v, src + 4 * (size_t)w, 8); memcpy(dst + w, &v, 8); w += 2; }
    if (w < nw) memcpy(dst + w, src + 4 * (size_t)w, 4);
}

static bool pack_seq(const char *s, int len, uint32_t *w, int nwords, std::vector<uint32_t> *nlist = nullptr, uint32_t tag = 0, bool codes = false, bool window = false)
{
    const uint8_t *lut = codes ? g_code_lut.v : window ? g_win_lut.v : g_pack_lut.v; const unsigned char *u = (const unsigned char *)s;
    int i = 0, wi = 0;
    for (; i + 16 <= len; i += 16, wi++) {
        if (g_have_ssse3 && pack16_ssse3(u + i, codes, w + wi)) continue;
        uint32_t x = 0, fl = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) { const uint32_t c = lut[u[i + t]]; fl |= c; x = (x << 2) | (c & 3u); }
        w[wi] = x;
        if (fl & 12u) {                                  // an N or an invalid character among these 16
            if ((fl & 8u) || !nlist) return false;
            for (int t = 0; t < 16; t++) if (lut[u[i + t]] & 4u) nlist->push_back((tag << 10) | (uint32_t)(i + t));
        }
    }
    if (i < len) {
        uint32_t x = 0;
        for (int t = 0; t < 16; t++) {
            uint32_t c = 0;
            if (i + t < len) { c = lut[u[i + t]]; if ((c & 8u) || ((c & 4u) && !nlist)) return false; if (c & 4u) nlist->push_back((tag << 10) | (uint32_t)(i + t)); }
            x = (x << 2) | (c & 3u);
        }
        w[wi++] = x;
    }
    for (; wi < nwords; wi++) w[wi] = 0;
    return true;
}

extern "C" int bk_pack_sequence(const char *seq, int32_t len, uint32_t flags, uint32_t *words, int32_t n_words, uint32_t *n_pos, int32_t cap, int32_t *n_n)
{
    if (!seq || len < 0 || !words || n_words < (len + 15) / 16 || cap < 0 || (cap > 0 && !n_pos)) return BK_E_ARG;
    std::vector<uint32_t> nl;
    if (!pack_seq(seq, len, words, n_words, &nl, 0, (flags & BK_SUBMIT_READ_CODES) != 0)) return BK_E_ARG;
    for (size_t i = 0; i < nl.size() && (int32_t)i < cap; i++) n_pos[i] = nl[i];                       // tag 0: the entry is the position
    if (n_n) *n_n = (int32_t)nl.size();
    return BK_OK;
}

static BkTarget make_target(const char *s, int len);
template <class T> static hipError_t upload(bk_handle *h, DevBuf &b, const std::vector<T> &v)
{
    hipError_t e = b.ensure(std::max<size_t>(v.size() * sizeof(T), 256));
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream);
}

static hipError_t upload_raw(bk_handle *h, DevBuf &b, const void *src, size_t bytes)
{
    hipError_t e = b.ensure(std::max<size_t>(bytes, 256));
    if (e != hipSuccess || !bytes) return e;
    return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream);
}

#define BK_SUBMIT_THREADS 8       // threads filling the staging buffer of one submit unless bk_config.submit_threads says otherwise (a driver has two or three submits in flight on 16 cores: 16 each measured no faster than 8, tools/probes/with_submit_probe.py)
// The helper threads of one submit: spawned once, then handed one job after the other (spawning sixteen threads per phase cost as
// much as a phase).  run(): the helpers and the caller execute the job; start() / wait(): the helpers alone, the caller does
// something else meanwhile (issues the copies of the chunks they finish).
class BkTeam {
    std::vector<std::thread> th_; std::mutex m_; std::condition_variable cv_, done_cv_;
    const std::function<void()> *job_ = nullptr; uint64_t serial_ = 0; int busy_ = 0; bool stop_ = false;
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()> *job;
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return stop_ || serial_ != seen; }); if (stop_) return; seen = serial_; job = job_; }
            (*job)();
            { std::lock_guard<std::mutex> lk(m_); if (--busy_ == 0) done_cv_.notify_all(); }
        }
    }
public:
    explicit BkTeam(int helpers) { for (int i = 0; i < helpers; i++) th_.emplace_back([this] { loop(); }); }
    ~BkTeam() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } cv_.notify_all(); for (auto &t : th_) t.join(); }
    void start(const std::function<void()> &fn) { if (th_.empty()) return; { std::lock_guard<std::mutex> lk(m_); job_ = &fn; busy_ = (int)th_.size(); serial_++; } cv_.notify_all(); }
    void wait() { if (th_.empty()) return; std::unique_lock<std::mutex> lk(m_); done_cv_.wait(lk, [&] { return busy_ == 0; }); }
    void run(const std::function<void()> &fn) { start(fn); fn(); wait(); }
};

static hipError_t ensure_mirrors(bk_handle *h);
// The rows of a submit cross the bus WITHOUT the pad word the kernels want behind every row (k-mer extraction may touch one word past
// the end: 11 instead of 10 words per 150-base read, 9 % of the bytes of a batch); this kernel lays them out with it.  One workgroup
// column per region (blockIdx.x), blockIdx.y strides over its words.
extern "C" __global__ void __launch_bounds__(256) bk_expand_rows_kernel(const BkRegionDesc *desc, const unsigned long long *cwoff, const uint32_t *in, uint32_t *out)
{
    const BkRegionDesc d = desc[blockIdx.x];
    const uint32_t rw = d.read_words, wc = rw - 1;
    const uint32_t *src = in + cwoff[blockIdx.x]; uint32_t *dst = out + d.reads_word_off;
    const uint32_t n = d.n_reads * rw;                       // < 2^22 reads x <= 65 words: fits 32 bits (32-bit divisions below)
    for (uint32_t o = blockIdx.y * blockDim.x + threadIdx.x; o < n; o += gridDim.y * blockDim.x) {
        const uint32_t i = o / rw, w = o - i * rw;
        dst[o] = w < wc ? src[(size_t)i * wc + w] : 0u;
    }
}
static int submit_regions(bk_handle *h, const bk_region *regions, int32_t n_regions, uint32_t flags);
extern "C" int bk_submit_regions(bk_handle *h, const bk_region *regions, int32_t n_regions) { (void)join_pending(h); return submit_regions(h, regions, n_regions, 0); }
extern "C" int bk_submit_regions_ex(bk_handle *h, const bk_region *regions, int32_t n_regions, uint32_t flags)
{
    (void)join_pending(h);                               // an unfinished earlier submit is superseded; its error no longer matters
    if (!(flags & BK_SUBMIT_ASYNC)) return submit_regions(h, regions, n_regions, flags);
    if (!h || !regions || n_regions <= 0) return fail(h, BK_E_ARG, "bk_submit_regions: bad argument");
    // the bk_region array is copied; the sequences it points to stay with the caller until the next call on this handle returns
    h->worker_regions.assign(regions, regions + n_regions);
    h->submitted = false; h->ran = false; h->fetched = false; h->synced = false; h->calls_valid = false;
    h->worker_rc = BK_OK; h->has_worker = true;
    h->worker = std::thread([h, n_regions, flags]() { tl_is_worker = true; h->worker_rc = submit_regions(h, h->worker_regions.data(), n_regions, flags & ~(uint32_t)BK_SUBMIT_ASYNC); });
    return BK_OK;
}

static int submit_regions(bk_handle *h, const bk_region *regions, int32_t n_regions, uint32_t flags)
{
    if (!h || !regions || n_regions <= 0) return fail(h, BK_E_ARG, "bk_submit_regions: bad argument");
    const bool read_codes = (flags & BK_SUBMIT_READ_CODES) != 0, packed = (flags & BK_SUBMIT_PACKED) != 0;
    HIPCHK(h, hipSetDevice(h->dev));
    const int k = h->cfg.kmer_size;
    // a failed submit leaves the handle without a batch (never the old device results paired with new host mirrors):
    // the new host mirrors are built in locals and swapped in on success only
    h->submitted = false; h->ran = false; h->fetched = false; h->synced = false; h->hold_snapshot = false; h->have_ctx = false;
    std::vector<uint32_t> sc, win, wnlist; std::vector<uint16_t> sclen;
    std::vector<BkRegionDesc> n_desc(n_regions, BkRegionDesc{}); std::vector<BkPartnerDesc> n_part; std::vector<std::vector<BkTarget>> n_targets(n_regions);
    uint32_t n_max_win = 0; uint64_t n_alg_bytes = 0;
    uint64_t dd_total = 0; uint32_t max_w = 0;
    size_t tot_reads = 0, tot_words = 0, tot_cwords = 0, tot_sc = 0, tot_scw = 0, tot_win = 0;      // tot_words: rows with their pad word (device layout); tot_cwords: without (staging, transfer)
    const int want_th = h->cfg.submit_threads > 0 ? std::min(h->cfg.submit_threads, 64) : packed ? BK_SUBMIT_THREADS : 2 * BK_SUBMIT_THREADS;      // (2-bit packing of ASCII is four times the bytes and real work per byte)
    const int nth = std::max(1, std::min<int>({want_th, (int)std::thread::hardware_concurrency(), n_regions}));
    BkTeam team(nth - 1);                                // the helper threads of this submit, spawned once (this thread is the nth)
    for (int r = 0; r < n_regions; r++) {
        const bk_region &g = regions[r];
        if (g.n_reads < 0 || !g.window || g.window_len <= 0 || (g.n_reads > 0 && (!g.reads || !g.read_lens))) return fail(h, BK_E_ARG, "bk_submit_regions: region " + std::to_string(r) + ": missing reads/window", r);
        if (packed && read_codes) return fail(h, BK_E_ARG, "bk_submit_regions: BK_SUBMIT_PACKED and BK_SUBMIT_READ_CODES exclude each other");
        if (g.n_reads >= (1 << 22)) return fail(h, BK_E_LIMIT, "bk_submit_regions: more than 4M reads in one region");
        if (g.n_partners > 15) return fail(h, BK_E_LIMIT, "bk_submit_regions: more than 15 partner windows");
    }
    // per region: longest read, number of bases (one pass over the lengths, regions in parallel)
    std::vector<uint32_t> r_maxl(n_regions, 0); std::vector<uint64_t> r_bases(n_regions, 0);
    {
        std::atomic<int> next{0};
        team.run([&]() {
            for (;;) {
                const int r = next.fetch_add(1);
                if (r >= n_regions) break;
                const bk_region &g = regions[r];
                uint32_t maxl = 0; uint64_t bases = 0;
                for (int i = 0; i < g.n_reads; i++) { const uint32_t l = g.read_lens[i]; maxl = std::max(maxl, l); bases += l; }
                r_maxl[r] = maxl; r_bases[r] = bases;
            }
        });
    }
    for (int r = 0; r < n_regions; r++) {
        const bk_region &g = regions[r];
        if ((int)r_maxl[r] > h->cfg.max_read_len) return fail(h, BK_E_LIMIT, "bk_submit_regions: read longer than max_read_len");
        tot_reads += g.n_reads; tot_words += (size_t)g.n_reads * ((r_maxl[r] + 15) / 16 + 1); tot_cwords += (size_t)g.n_reads * ((r_maxl[r] + 15) / 16);
        if (g.n_sc > 0) { uint32_t ms = 0; for (int i = 0; i < g.n_sc; i++) ms = std::max<uint32_t>(ms, g.sc_lens[i]); tot_sc += g.n_sc; tot_scw += (size_t)g.n_sc * ((ms + 15) / 16 + 1); }
        tot_win += (g.window_len + 15) / 16 + 2; for (int q = 0; q < g.n_partners; q++) tot_win += (g.partner_lens[q] + 15) / 16 + 2;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));          // the staging buffers may still feed the copies of the previous submit
    HIPCHK(h, h->hs_reads.resize(std::max<size_t>(tot_cwords, 1) * 4)); HIPCHK(h, h->hs_rlen.resize(std::max<size_t>(tot_reads, 1) * 2)); HIPCHK(h, h->hs_rflag.resize(std::max<size_t>(tot_reads, 1)));
    uint32_t *reads = (uint32_t *)h->hs_reads.data(); uint16_t *rlen = (uint16_t *)h->hs_rlen.data(); uint8_t *rflag = h->hs_rflag.data();
    if (!tot_cwords) reads[0] = 0;
    if (!tot_reads) { rlen[0] = 0; rflag[0] = 0; }
    sc.reserve(tot_scw); sclen.reserve(tot_sc); win.assign(tot_win, 0);
    // Layout of the batch (offsets only; the sequences are packed below, regions in parallel)
    size_t reads_top = 0, creads_top = 0, meta_top = 0, win_top = 0;
    std::vector<unsigned long long> cwoff(n_regions, 0);          // where region r's rows start in the staging buffer (words)
    for (int r = 0; r < n_regions; r++) {
        const bk_region &g = regions[r]; BkRegionDesc &d = n_desc[r];
        const uint32_t maxl = r_maxl[r];
        d.n_reads = g.n_reads; d.read_words = (maxl + 15) / 16 + 1;          // +1: k-mer extraction may touch one word past the end
        d.max_len = maxl;
        d.reads_word_off = reads_top; d.read_meta_off = meta_top; cwoff[r] = creads_top;
        reads_top += (size_t)d.n_reads * d.read_words; creads_top += (size_t)d.n_reads * (d.read_words - 1); meta_top += d.n_reads;
        d.n_sc = g.n_sc < 0 ? -1 : g.n_sc; d.sc_word_off = sc.size(); d.sc_meta_off = sclen.size(); d.sc_words = 1;
        if (g.n_sc > 0) {
            uint32_t ms = 0; for (int i = 0; i < g.n_sc; i++) ms = std::max<uint32_t>(ms, g.sc_lens[i]);
            d.sc_words = (ms + 15) / 16 + 1; sc.resize(sc.size() + (size_t)g.n_sc * d.sc_words);
            for (int i = 0; i < g.n_sc; i++) {
                if (!pack_seq(g.sc_seqs + (size_t)i * g.sc_stride, g.sc_lens[i], sc.data() + d.sc_word_off + (size_t)i * d.sc_words, d.sc_words)) return fail(h, BK_E_ARG, "bk_submit_regions: non-ACGT base in soft-clip sequence");
                sclen.push_back(g.sc_lens[i]);
            }
        }
        d.win_len = g.window_len; d.win_word_off = win_top; win_top += (g.window_len + 15) / 16 + 2;
        max_w = std::max<uint32_t>(max_w, g.window_len);
        n_max_win = std::max<uint32_t>(n_max_win, g.window_len);
        d.n_partners = g.n_partners; d.part_desc_off = n_part.size();
        for (int q = 0; q < g.n_partners; q++) {
            BkPartnerDesc pd; pd.word_off = win_top; pd.len = g.partner_lens[q]; pd.n_n = 0; pd.n_off = 0;
            win_top += (pd.len + 15) / 16 + 2;
            n_part.push_back(pd);
            n_max_win = std::max<uint32_t>(n_max_win, pd.len);
        }
        n_targets[r].resize(1 + (size_t)std::max(g.n_partners, 0));
        uint32_t cap = 64; while ((uint64_t)cap * 7 < (uint64_t)std::max(g.n_reads, 1) * 10) cap <<= 1;      // load factor <= 0.7 even if every read is unique
        d.dedup_cap = cap; d.dedup_off = dd_total; dd_total += cap;
        // algorithmic HBM bytes per region: 2-bit reads + 4 B/read offsets + window fwd+rc + ~2 KB of output
        n_alg_bytes += (r_bases[r] + 3) / 4 + 4ull * g.n_reads + 2ull * ((g.window_len + 3) / 4) + 2048;
    }
    const auto t_pack0 = std::chrono::steady_clock::now();
    // The bulk of a submit: the reads into the pinned staging buffer (2-bit packing of 0.4 GB of ASCII, or row copies of 0.1 GB of
    // packed rows, per 256 regions), the windows packed, the host's copies of the windows made -- regions are independent and go
    // over the helper threads in index order.  The staging buffer goes to the device in CHUNKS of regions as they are finished
    // (this thread issues the copies), so the transfer of the first chunk runs while the last ones are still being filled.
    HIPCHK(h, h->d_reads.ensure(std::max<size_t>(tot_words * 4, 256))); HIPCHK(h, h->d_reads_in.ensure(std::max<size_t>(tot_cwords * 4, 256)));
    std::vector<std::vector<uint32_t>> region_nl(n_regions), region_wn(n_regions);
    std::vector<std::vector<uint32_t>> region_wn_cnt(n_regions);          // N positions per window of the region (window, then its partners): lengths of the pieces of region_wn
    const int n_chunks = std::max(1, std::min(8, n_regions / 8));
    std::vector<int> chunk_end(n_chunks, n_regions);
    { size_t acc = 0; int c = 0; for (int r = 0; r < n_regions && c < n_chunks - 1; r++) { acc += (size_t)n_desc[r].n_reads * (n_desc[r].read_words - 1); if (acc * n_chunks >= tot_cwords * (size_t)(c + 1)) chunk_end[c++] = r + 1; } }
    std::vector<std::atomic<int>> chunk_done(n_chunks);
    for (auto &c : chunk_done) c.store(0);
    std::vector<int> chunk_of(n_regions, 0);
    { int c = 0; for (int r = 0; r < n_regions; r++) { while (r >= chunk_end[c]) c++; chunk_of[r] = c; } }
    // the first (smallest region index) offender of each kind: region << 32 | read, region << 1 | (partner window)
    std::atomic<int> next{0}; std::atomic<uint64_t> bad_read{UINT64_MAX}, bad_window{UINT64_MAX};
    auto note_min = [](std::atomic<uint64_t> &a, uint64_t v) { uint64_t cur = a.load(); while (v < cur && !a.compare_exchange_weak(cur, v)) {} };
    const std::function<void()> pack_region = [&]() {
        for (;;) {
            const int r = next.fetch_add(1);
            if (r >= n_regions) break;
            const bk_region &g = regions[r]; const BkRegionDesc &d = n_desc[r];
            const uint32_t wc = d.read_words - 1;            // words per row in the staging buffer (the device adds the pad word: bk_expand_rows_kernel)
            for (int i = 0; i < g.n_reads; i++) { rlen[d.read_meta_off + i] = g.read_lens[i]; rflag[d.read_meta_off + i] = g.indel_only && g.indel_only[i] ? BK_RF_INDEL : 0; }
            if (packed) {        // rows are 2 bit/base already: copied into the library's row stride, the tail zeroed
                for (int i = 0; i < g.n_reads; i++) {
                    const uint32_t nw = ((uint32_t)g.read_lens[i] + 15u) / 16u; uint32_t *dst = reads + cwoff[r] + (size_t)i * wc;
                    copy_words(dst, (const unsigned char *)g.reads + (size_t)i * g.read_stride, nw);
                    if (g.read_lens[i] & 15) dst[nw - 1] &= 0xFFFFFFFFu << (2 * (16 - (g.read_lens[i] & 15)));      // bases beyond the length must read as A (the kernels compare whole words)
                    for (uint32_t w = nw; w < wc; w++) dst[w] = 0;
                }
                if (g.read_n && g.n_read_n > 0) {
                    region_nl[r].assign(g.read_n, g.read_n + g.n_read_n);
                    for (int e = 0; e < g.n_read_n; e++) {
                        const uint32_t v = g.read_n[e], ri = v >> 10, pos = v & 1023u;
                        if ((int)ri >= g.n_reads || pos >= g.read_lens[ri] || (e && g.read_n[e - 1] >= v)) { note_min(bad_read, ((uint64_t)r << 32) | ri); break; }
                        uint32_t *dst = reads + cwoff[r] + (size_t)ri * wc;
                        dst[pos >> 4] &= ~(3u << (30 - 2 * (pos & 15)));                              // an N is packed as A
                    }
                }
            } else {
                for (int i = 0; i < g.n_reads; i++)
                    if (!pack_seq(g.reads + (size_t)i * g.read_stride, g.read_lens[i], reads + cwoff[r] + (size_t)i * wc, (int)wc, &region_nl[r], (uint32_t)i, read_codes)) { note_min(bad_read, ((uint64_t)r << 32) | (uint32_t)i); break; }
            }
            chunk_done[chunk_of[r]].fetch_add(1, std::memory_order_release);          // the rows of this region are in the staging buffer
            // the windows: an N (an assembly gap near the target) is packed as code 0 and listed: its k-mers do not exist, it matches nothing
            for (int q = -1; q < g.n_partners; q++) {
                const char *ws = q < 0 ? g.window : g.partners[q]; const int wl = q < 0 ? g.window_len : (int)g.partner_lens[q];
                const uint64_t off = q < 0 ? d.win_word_off : n_part[d.part_desc_off + (size_t)q].word_off;
                const size_t before = region_wn[r].size();
                if (!pack_seq(ws, wl, win.data() + off, (wl + 15) / 16 + 2, &region_wn[r], 0, false, true)) { note_min(bad_window, ((uint64_t)r << 1) | (q < 0 ? 0u : 1u)); break; }
                region_wn_cnt[r].push_back((uint32_t)(region_wn[r].size() - before));
                n_targets[r][(size_t)(q + 1)] = make_target(ws, wl);
            }
        }
    };
    if (nth == 1) {
        pack_region();
        if (tot_cwords) HIPCHK(h, hipMemcpyAsync(h->d_reads_in.p, reads, tot_cwords * 4, hipMemcpyHostToDevice, h->stream));
    } else {
        team.start(pack_region);
        hipError_t cerr = hipSuccess; size_t sent = 0;      // words of the staging buffer handed to the copy engine so far
        for (int c = 0; c < n_chunks; c++) {
            const int first = c ? chunk_end[c - 1] : 0, want = chunk_end[c] - first;
            while (chunk_done[c].load(std::memory_order_acquire) < want) std::this_thread::sleep_for(std::chrono::microseconds(20));
            const size_t upto = chunk_end[c] < n_regions ? (size_t)cwoff[chunk_end[c]] : tot_cwords;
            if (cerr == hipSuccess && upto > sent) cerr = hipMemcpyAsync((uint8_t *)h->d_reads_in.p + sent * 4, reads + sent, (upto - sent) * 4, hipMemcpyHostToDevice, h->stream);
            sent = upto;
        }
        team.wait();
        HIPCHK(h, cerr);
    }
    if (bad_window.load() != UINT64_MAX) return fail(h, BK_E_ARG, "bk_submit_regions: region " + std::to_string(bad_window.load() >> 1) + ((bad_window.load() & 1) ? ": character other than A/C/G/T/N in a partner window" : ": character other than A/C/G/T/N in the reference window"), (int32_t)(bad_window.load() >> 1));
    if (bad_read.load() != UINT64_MAX) return fail(h, BK_E_ARG, "bk_submit_regions: region " + std::to_string(bad_read.load() >> 32) + " read " + std::to_string(bad_read.load() & 0xFFFFFFFFu) + (packed ? ": N list not ascending or out of range" : ": base other than A/C/G/T/N"), (int32_t)(bad_read.load() >> 32));
    // N calls: one sorted list per region (reads are packed in order, positions ascending), flag on the reads that have any;
    // the N positions of the windows, window by window
    std::vector<uint32_t> nlist;
    for (int r = 0; r < n_regions; r++) {
        BkRegionDesc &d = n_desc[r];
        d.nlist_off = nlist.size(); d.n_nlist = (uint32_t)region_nl[r].size();
        for (uint32_t e : region_nl[r]) rflag[d.read_meta_off + (e >> 10)] |= BK_RF_HASN;
        nlist.insert(nlist.end(), region_nl[r].begin(), region_nl[r].end());
        d.win_n_off = wnlist.size(); d.n_win_n = region_wn_cnt[r][0];
        size_t o = d.n_win_n;
        for (uint32_t q = 0; q < d.n_partners; q++) { BkPartnerDesc &pd = n_part[d.part_desc_off + q]; pd.n_off = wnlist.size() + o; pd.n_n = region_wn_cnt[r][q + 1]; o += pd.n_n; }
        wnlist.insert(wnlist.end(), region_wn[r].begin(), region_wn[r].end());
    }
    if (nlist.empty()) nlist.push_back(0);
    h->total_reads = tot_reads; h->n_regions = n_regions;
    { uint32_t mx = 0; for (auto &d : n_desc) mx = std::max(mx, d.max_len); h->eff_max_read = (int)std::min<uint32_t>((uint32_t)h->cfg.max_read_len, std::max<uint32_t>(64, (mx + 31) / 32 * 32)); }
    // reference k-mer table geometry (LDS): load factor <= 0.5.  Windows beyond the LDS budget (whole-gene targets)
    // are flagged `big` and go through bk_kmer_kernel_g (table in the scratch arena).
    auto lds_need = [&](uint32_t w, uint32_t &cap, uint32_t &words) {
        const uint32_t wk2 = w >= (uint32_t)k ? 2 * (w - k + 1) : 0;
        cap = 1024; while (cap < 2 * wk2) cap <<= 1;
        words = ((w + 15) / 16 + 2 + 3) & ~3u;
        return wk2 < (1u << 18) ? (32 + 256 + 2 * (size_t)words + cap) * 4 : (size_t)1 << 30;
    };
    uint32_t max_small = 0; h->n_big = 0; uint64_t big_bytes = 0;
    for (auto &d : n_desc) {
        uint32_t cap, words;
        d.big = lds_need(d.win_len, cap, words) > BK_LDS_MAX ? 1u : 0u;
        if (d.big) { h->n_big++; uint64_t gc = 1024; while (gc < 4ull * d.win_len) gc <<= 1; big_bytes += gc * 8 + d.win_len / 4 + 4096; if (d.win_len >= (1u << 28)) return fail(h, BK_E_LIMIT, "bk_submit_regions: reference window longer than 256 Mb"); }
        else max_small = std::max(max_small, d.win_len);
    }
    { uint32_t cap, words; lds_need(max_small, cap, words); h->ref_cap = cap; h->win_words_cap = words; }
    h->group_words = 0;
    for (auto &d : n_desc) {
        const uint32_t need = 32 + 256 + d.dedup_cap;
        if (!d.big && d.n_reads < 16383u && need <= 36864u) h->group_words = std::max(h->group_words, need);
    }
    h->big_bytes = big_bytes;
    if (sc.empty()) sc.push_back(0);
    if (sclen.empty()) sclen.push_back(0);
    if (n_part.empty()) n_part.push_back(BkPartnerDesc{0, 0, 0, 0});
    if (wnlist.empty()) wnlist.push_back(0);
    const auto t_h2d0 = std::chrono::steady_clock::now();
    HIPCHK(h, upload(h, h->d_desc, n_desc)); HIPCHK(h, upload(h, h->d_part, n_part)); HIPCHK(h, upload(h, h->d_cwoff, cwoff));
    // (same stream as the chunks above and the descriptors: the rows are complete and described when this runs)
    hipLaunchKernelGGL(bk_expand_rows_kernel, dim3(n_regions, 16), dim3(256), 0, h->stream, (const BkRegionDesc *)h->d_desc.p, (const unsigned long long *)h->d_cwoff.p, (const uint32_t *)h->d_reads_in.p, (uint32_t *)h->d_reads.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, upload_raw(h, h->d_rlen, rlen, std::max<size_t>(tot_reads, 1) * 2));      // (the reads went over in chunks above)
    HIPCHK(h, upload_raw(h, h->d_rflag, rflag, std::max<size_t>(tot_reads, 1)));
    HIPCHK(h, upload(h, h->d_sc, sc)); HIPCHK(h, upload(h, h->d_sclen, sclen)); HIPCHK(h, upload(h, h->d_win, win));
    HIPCHK(h, upload(h, h->d_nlist, nlist)); HIPCHK(h, upload(h, h->d_wnlist, wnlist));
    const size_t nr = std::max<size_t>(h->total_reads, 1), nd = std::max<uint64_t>(dd_total, 1);
    HIPCHK(h, h->d_work.ensure(sizeof(BkRegionWork) * n_regions));
    HIPCHK(h, h->d_ddslot.ensure(nd * 8)); HIPCHK(h, h->d_ddrep.ensure(nd * 4)); HIPCHK(h, h->d_ddcnt.ensure(nd * 4));
    HIPCHK(h, h->d_grp.ensure(nr * 4)); HIPCHK(h, h->d_urep.ensure(nr * 4)); HIPCHK(h, h->d_unr.ensure(nr * 4)); HIPCHK(h, h->d_ufl.ensure(nr));
    HIPCHK(h, h->d_ubuf.ensure(nr * 4)); HIPCHK(h, h->d_ureads.ensure(nr * 4)); HIPCHK(h, h->d_ufound.ensure(nr * 4)); HIPCHK(h, h->d_uminpos.ensure(nr * 4));
    HIPCHK(h, h->d_tops.ensure(256));
    if (h->arena_cap == 0) {
        uint64_t want = h->cfg.arena_bytes > 0 ? (uint64_t)h->cfg.arena_bytes : std::max<uint64_t>(64ull << 20, (uint64_t)n_regions * (1ull << 20) + h->total_reads * 64ull);
        h->arena_cap = want;
    }
    if (h->arena_cap < h->big_bytes + (64ull << 20) && h->big_bytes) {
        h->arena_cap = h->big_bytes + std::max<uint64_t>(64ull << 20, (uint64_t)n_regions * (1ull << 20) + h->total_reads * 64ull);
    }
    if (h->out_cap == 0) h->out_cap = h->cfg.out_kbytes > 0 ? (uint64_t)h->cfg.out_kbytes << 10 : std::max<uint64_t>(8ull << 20, (uint64_t)n_regions * (64ull << 10));
    HIPCHK(h, h->d_arena.ensure(h->arena_cap)); HIPCHK(h, h->d_out.ensure(h->out_cap));
    HIPCHK(h, ensure_mirrors(h));                        // (here, not at the first launch: pinned allocations take milliseconds and wait for the device)
    HIPCHK(h, hipStreamSynchronize(h->stream));          // host staging vectors go out of scope
    { const auto t1 = std::chrono::steady_clock::now();
      h->submit_pack_ms = std::chrono::duration<double, std::milli>(t_h2d0 - t_pack0).count(); h->submit_h2d_ms = std::chrono::duration<double, std::milli>(t1 - t_h2d0).count(); }
    h->h_desc.swap(n_desc); h->h_part.swap(n_part); h->h_targets.swap(n_targets); h->max_win = n_max_win; h->alg_bytes = n_alg_bytes;
    h->submitted = true; h->ran = false; h->fetched = false; h->calls_valid = false;
    return BK_OK;
}
