// bk_api.hip -- host side of libbreakmer_hip.so: the C-ABI of include/breakmer_hip.h.
// gfx950 only; there is no CPU fallback anywhere in this library.
//
// Builds.  The PRODUCT build (breakmer_amd/build.py, no -D) reads nothing from the environment: its behaviour is a function of
// its arguments.  Every diagnostic switch lives behind -DBK_DIAG (build.py variant "diag"; implied by the barrier-check and
// jitter builds, bk_common.h): environment variables BK_SPLIT_OFF, BK_SW_T, BK_POISON_ARENA, BK_POISON_LDS, BK_LDS_PAD, BK_DBG_ITERS,
// BK_DEBUG_SPLIT, BK_JITTER_SEED -- read through bk_diag_env(), which is a constant nullptr in the product build.
#if (defined(BK_SYNC_CHECK) || defined(BK_JITTER)) && !defined(BK_DIAG)
#define BK_DIAG 1
#endif
#include "../../include/breakmer_hip.h"
#include "bk_common.h"
#include "bk_kmer.hip.h"
#include "bk_sched.hip.h"
#include "bk_nw.hip.h"
#define BK_AT 512
#define BK_ASM_KERNEL bk_asm_kernel
#define BK_WITH_NW_BATCH
namespace at512 {
#include "bk_asm.hip.h"
static const size_t ctx_shared_bytes = (sizeof(BkAsmCtx) + 15) / 16 * 16 + (sizeof(BkAsmShared) + 15) / 16 * 16;
}
#undef BK_AT
#undef BK_ASM_KERNEL
#undef BK_WITH_NW_BATCH
#define BK_AT 256
#define BK_ASM_KERNEL bk_asm_kernel_w4
#define BK_PAIR 1
namespace at256 {
#include "bk_asm.hip.h"
static const size_t ctx_shared_bytes = (sizeof(BkAsmCtx) + 15) / 16 * 16 + (sizeof(BkAsmShared) + 15) / 16 * 16;
}
#undef BK_AT
#undef BK_ASM_KERNEL
#undef BK_PAIR
using at512::bk_nw_batch_kernel;
#include "bk_sw.hip.h"
#include "bk_call.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <thread>
#include <vector>

static std::string g_create_err;
#ifdef BK_DIAG
static inline const char *bk_diag_env(const char *name) { return getenv(name); }
#else
static inline constexpr const char *bk_diag_env(const char *) { return nullptr; }
#endif
#define BK_MAX_CONTIG_ABS 32704      // the realign kernel's 64-bit hit key holds 15 bits of query position
#ifdef BK_SYNC_CHECK
#define BK_LDS_MAX (160 * 1024 - 256)      // (barrier-check build: the site table of bk_sync_diag is static LDS of every kernel)
#else
#define BK_LDS_MAX (160 * 1024)      // LDS of a gfx950 CU = the most one workgroup can have
#endif

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n ? n : 256);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct HostVec {            // pinned host mirror of the result arena (faster D2H than pageable memory)
    uint8_t *p = nullptr; size_t cap = 0, n = 0;
    hipError_t resize(size_t m) { if (m > cap) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; hipError_t e = hipHostMalloc((void **)&p, m + (m >> 2) + 4096, hipHostMallocDefault); if (e != hipSuccess) return e; cap = m + (m >> 2) + 4096; } n = m; return hipSuccess; }
    uint8_t *data() { return p; } const uint8_t *data() const { return p; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
};

// One target window as the host keeps it for the chaining: the bases upper-cased, and where the caller's FASTA was soft-masked
// (lower case: BLAT's -repeats=lower, sv_processor.py:843) a 0/1 byte per base -- a match on such a base counts as repMatches.
struct BkTarget { std::string seq; std::string soft; bool masked(int i) const { return !soft.empty() && soft[(size_t)i] != 0; } size_t size() const { return seq.size(); } char operator[](size_t i) const { return seq[i]; } const char *c_str() const { return seq.c_str(); } };

struct bk_handle {
    int dev = 0; hipStream_t stream = nullptr; hipEvent_t ev[6] = {};
    bk_config cfg{}; std::string err; int32_t err_region = -1;      // err_region: the region the last failed submit names (bk_last_error_region)
    int n_regions = 0; bool submitted = false, ran = false, fetched = false, synced = false;
    bool hold_snapshot = false;      // bk_fetch() done: bk_call() uses the host copy even if a newer run is in flight
    uint32_t ran_mask = 0;
    // device
    DevBuf d_desc, d_work, d_part, d_reads, d_rlen, d_rflag, d_sc, d_sclen, d_win;
    DevBuf d_reads_in, d_cwoff;      // a submit's reads as they cross the bus (rows without the pad word) and where each region's rows start in them
    DevBuf d_ddslot, d_ddrep, d_ddcnt, d_grp, d_urep, d_unr, d_ufl, d_ubuf, d_ureads, d_ufound, d_uminpos;
    DevBuf d_arena, d_out, d_tops, d_order, d_skeys, d_clist, d_nlist, d_wnlist;
    int n_cu = 256, asm_wg_per_cu = 0, sw_wg_per_cu = 0, asm_threads = 512;
    uint64_t arena_cap = 0, out_cap = 0;
    uint32_t ref_cap = 0, win_words_cap = 0;
    int n_big = 0; uint64_t big_bytes = 0;      // regions whose window needs the global-memory k-mer set
    uint32_t group_words = 0;                   // LDS words wanted by the in-LDS read-grouping table (largest region that qualifies)
    // host mirrors
    std::vector<BkRegionDesc> h_desc; std::vector<BkRegionWork> h_work; HostVec h_out;
    // What the host reads of EVERY batch -- the work records, the bump pointers, the contig records -- is written into pinned host
    // memory by the batch's last kernel (bk_mirror_kernel) instead of being fetched by small copies after it: a copy engine that is
    // in the middle of another handle's 100 MB submit made each of those wait its turn (1.3-1.5 ms per batch with submits in flight).
    // m_out and h_out swap roles at bk_fetch (a run in flight never writes into the copy bk_call is still reading).
    HostVec m_work, m_tops, m_out; bool mirror_fresh = false; uint64_t m_out_cap = 0;
    HostVec hs_reads, hs_rlen, hs_rflag;        // pinned staging of a submit (packed reads, lengths, flags): kept and reused, so a
                                                // submit neither page-faults a fresh 100 MB vector nor copies from pageable memory
    std::vector<BkPartnerDesc> h_part;
    std::vector<std::vector<BkTarget>> h_targets;         // per region: target window + partner windows (upper case + soft-mask flags), for PSL assembly
    uint32_t max_win = 0;
    int eff_max_read = 64;          // batch maximum read length rounded up to 32: sizes the assembler's LDS buffers (occupancy)
    uint64_t total_reads = 0, alg_bytes = 0;
    float ms[4] = {0, 0, 0, 0};
    double submit_pack_ms = 0, submit_h2d_ms = 0;   // host 2-bit packing / host-to-device copies of the last bk_submit_regions
    int n_failed = 0;                                // regions of the last run that hit a device limit (status per region)
    int n_escalated = 0;                             // regions of the last run that were run again with larger assembler caps
    int n_repair_passes = 0;                         // extra passes over split regions (components that met across units)
    DevBuf d_rmap;                                   // their indices (k-mer kernels of the re-run)
    BkParams params{};
    bkcall::Context call_ctx; bool have_ctx = false, have_tables = false; std::string calls_blob;
    bool calls_valid = false;        // calls_blob holds the calls of the last run under the current context (bk_call_async made them already)
    // BK_SUBMIT_ASYNC: the submit runs on this thread; every later call on the handle joins it first (and reports its error)
    std::thread worker; bool has_worker = false; int worker_rc = 0; std::vector<bk_region> worker_regions;
};

// A handle's worker thread runs call_impl / submit_regions, which pass through entry-point code that joins the worker: the
// worker recognises ITSELF by a thread-local flag it sets first thing (not through h->worker, which the launching thread may
// still be move-assigning when the new thread gets here: get_id() of a half-assigned std::thread, then join() on a
// non-joinable one -> std::terminate.  ADVICE round 4).
static thread_local bool tl_is_worker = false;
static int join_pending(bk_handle *h)
{
    if (!h || tl_is_worker || !h->has_worker) return 0;
    h->worker.join(); h->has_worker = false;
    return h->worker_rc;
}
#define BK_JOIN(h) do { const int jr_ = join_pending(h); if (jr_ != BK_OK) return jr_; } while (0)

#define HIPCHK(h, call)                                                                              \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return BK_E_HIP; } } while (0)

// Launch-time queries that do not change between launches of the same shape (a driver launches thousands of batches per second;
// each of these calls costs the host thread microseconds): the occupancy of (kernel, workgroup size, LDS bytes) per device, and
// the dynamic-LDS attribute a kernel was last given.
static int cached_occupancy(int dev, const void *fn, int threads, size_t lds)
{
    static std::mutex mu; static std::map<std::tuple<int, const void *, int, size_t>, int> memo;
    std::lock_guard<std::mutex> g(mu);
    auto key = std::make_tuple(dev, fn, threads, lds);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    memo[key] = per_cu;
    return per_cu;
}
static hipError_t set_max_dyn_lds(int dev, const void *fn, int bytes)
{
    static std::mutex mu; static std::map<std::pair<int, const void *>, int> last;
    std::lock_guard<std::mutex> g(mu);
    auto key = std::make_pair(dev, fn);
    auto it = last.find(key);
    if (it != last.end() && it->second >= bytes) return hipSuccess;          // (the attribute is a maximum: a smaller request is covered)
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) last[key] = bytes;
    return e;
}

static int fail(bk_handle *h, int code, const std::string &msg, int32_t region = -1) { if (h) { h->err = msg; h->err_region = region; } else g_create_err = msg; return code; }

extern "C" int bk_abi_version(void) { return BK_ABI_VERSION; }
extern "C" const char *bk_last_error(const bk_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }
extern "C" int32_t bk_last_error_region(const bk_handle *h) { return h ? h->err_region : -1; }

static size_t asm_lds_bytes(const bk_handle *h, int threads, int max_cand, int max_contig);
extern "C" int bk_create(int device_id, const bk_config *cfg, bk_handle **out)
{
    if (!cfg || !out) return fail(nullptr, BK_E_ARG, "bk_create: null argument");
    if (cfg->abi_version != BK_ABI_VERSION) return fail(nullptr, BK_E_ARG, "bk_create: ABI version mismatch");
    // the arguments first (a bad configuration is BK_E_ARG on any machine), then the device
    if (cfg->kmer_size < 2 || cfg->kmer_size > 64) return fail(nullptr, BK_E_ARG, "bk_create: kmer_size must be in [2, 64]");
    // ABI 5: what steers the library has a name; nothing rides in `reserved`
    static_assert(sizeof(bk_config) == 64 && offsetof(bk_config, flags) == 40 && offsetof(bk_config, asm_wg_threads) == 44 && offsetof(bk_config, no_escalation) == 48 &&
                  offsetof(bk_config, submit_threads) == 52 && offsetof(bk_config, reserved) == 56, "bk_config layout (include/breakmer_hip.h; hip_backend.py and tests/test_abi_cpu.py mirror it)");
    if (cfg->reserved[0] || cfg->reserved[1]) return fail(nullptr, BK_E_ARG, "bk_create: bk_config.reserved must be 0 (ABI 5: flags, asm_wg_threads, no_escalation and submit_threads are fields of their own)");
    if (cfg->flags & ~(uint32_t)BK_CFG_KNOWN_MASK) return fail(nullptr, BK_E_ARG, "bk_create: unknown bit in bk_config.flags");
#ifndef BK_DIAG
    if (cfg->flags & (uint32_t)BK_CFG_DIAG_MASK) return fail(nullptr, BK_E_ARG, "bk_create: bk_config.flags holds a diagnostic-only bit (BK_CFG_DIAG_MASK): those are accepted by the -DBK_DIAG builds of the library only");
#endif
    if (cfg->asm_wg_threads != 0 && cfg->asm_wg_threads != 256 && cfg->asm_wg_threads != 512) return fail(nullptr, BK_E_ARG, "bk_create: asm_wg_threads must be 0 (the library chooses), 256 or 512");
    if ((cfg->no_escalation != 0 && cfg->no_escalation != 1) || cfg->submit_threads < 0 || cfg->submit_threads > 64) return fail(nullptr, BK_E_ARG, "bk_create: no_escalation must be 0 or 1, submit_threads in [0, 64]");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, BK_E_NOGPU, "bk_create: no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, BK_E_ARG, "bk_create: bad device id");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, BK_E_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(nullptr, BK_E_NOGPU, std::string("bk_create: device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only");
    bk_handle *h = new bk_handle();
    h->dev = device_id; h->cfg = *cfg; h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    { const char *e = bk_diag_env("BK_SPLIT_OFF"); if (e && atoi(e) > 0) h->cfg.flags |= BK_F_NO_SPLIT; }      // diagnostic build: every region one unit, for every handle of the process
    if (h->cfg.max_contig_len <= 0) h->cfg.max_contig_len = 4096;
    if (h->cfg.max_read_len <= 0) h->cfg.max_read_len = 1024;
    if (h->cfg.max_candidates <= 0) h->cfg.max_candidates = 2048;
    if (h->cfg.sw_min_score <= 0) h->cfg.sw_min_score = 20;
    if (h->cfg.rc_thresh <= 0) h->cfg.rc_thresh = 2;
    if (h->cfg.max_read_len > 1024 || h->cfg.max_contig_len > BK_MAX_CONTIG_ABS || h->cfg.max_contig_len > 2 * h->cfg.max_candidates) {
        delete h; return fail(nullptr, BK_E_ARG, "bk_create: limits: max_read_len <= 1024, max_contig_len <= 32,704 and <= 2*max_candidates");
    }
    {   // what the caps cost in LDS: the assembler's block at the 512-thread size with the shortest read buffers, the realigner's LONG tier
        h->eff_max_read = 64;
        const size_t a = asm_lds_bytes(h, 512, h->cfg.max_candidates, h->cfg.max_contig_len);
        const size_t w = bk_sw_layout(h->cfg.max_contig_len, 2 * (uint32_t)h->cfg.max_contig_len + 4096, 256, 1024, h->cfg.sw_min_score).total;
        if (a > BK_LDS_MAX || w > BK_LDS_MAX) {
            const std::string msg = "bk_create: max_contig_len " + std::to_string(h->cfg.max_contig_len) + " / max_candidates " + std::to_string(h->cfg.max_candidates) + " need " + std::to_string(a) + " B (assembler) and " + std::to_string(w) +
                                    " B (realigner) of LDS per workgroup; a CU has " + std::to_string(BK_LDS_MAX) + " (the library raises working caps by itself for regions that overflow them)";
            delete h; return fail(nullptr, BK_E_ARG, msg);
        }
    }
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(nullptr, BK_E_HIP, "stream creation failed"); }
    for (auto &e : h->ev) if (hipEventCreate(&e) != hipSuccess) { (void)bk_destroy(h); return fail(nullptr, BK_E_HIP, "event creation failed"); }
#if defined(BK_JITTER)
    { const char *e = bk_diag_env("BK_JITTER_SEED"); const uint32_t seed = e ? (uint32_t)strtoul(e, nullptr, 0) : 1u; (void)hipMemcpyToSymbol(HIP_SYMBOL(bk_jitter_seed), &seed, sizeof(seed)); }
#endif
    *out = h;
    return BK_OK;
}

extern "C" int bk_destroy(bk_handle *h)
{
    if (!h) return BK_OK;
    (void)join_pending(h);
    (void)hipSetDevice(h->dev);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->d_reads_in, &h->d_cwoff, &h->d_desc, &h->d_work, &h->d_part, &h->d_reads, &h->d_rlen, &h->d_rflag, &h->d_sc, &h->d_sclen, &h->d_win, &h->d_ddslot, &h->d_ddrep, &h->d_ddcnt,
                      &h->d_grp, &h->d_urep, &h->d_unr, &h->d_ufl, &h->d_ubuf, &h->d_ureads, &h->d_ufound, &h->d_uminpos, &h->d_arena, &h->d_out, &h->d_tops, &h->d_order, &h->d_skeys, &h->d_clist, &h->d_nlist, &h->d_wnlist, &h->d_rmap};
    for (auto b : bufs) b->release();
    h->h_out.release(); h->hs_reads.release(); h->hs_rlen.release(); h->hs_rflag.release(); h->m_work.release(); h->m_tops.release(); h->m_out.release();
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return BK_OK;
}

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

static void fill_params(bk_handle *h)
{
    BkParams &p = h->params;
    p.desc = (const BkRegionDesc *)h->d_desc.p; p.work = (BkRegionWork *)h->d_work.p; p.partners = (const BkPartnerDesc *)h->d_part.p;
    p.reads = (const uint32_t *)h->d_reads.p; p.read_len = (const uint16_t *)h->d_rlen.p; p.read_flag = (const uint8_t *)h->d_rflag.p;
    p.nlist = (const uint32_t *)h->d_nlist.p; p.wnlist = (const uint32_t *)h->d_wnlist.p;
    p.sc = (const uint32_t *)h->d_sc.p; p.sc_len = (const uint16_t *)h->d_sclen.p; p.windows = (const uint32_t *)h->d_win.p;
    p.dd_slot = (unsigned long long *)h->d_ddslot.p; p.dd_rep = (uint32_t *)h->d_ddrep.p; p.dd_cnt = (uint32_t *)h->d_ddcnt.p;
    p.grp_slot = (uint32_t *)h->d_grp.p; p.urep = (uint32_t *)h->d_urep.p; p.unreads = (uint32_t *)h->d_unr.p; p.uflag = (uint8_t *)h->d_ufl.p;
    p.ubuf = (int32_t *)h->d_ubuf.p; p.ureads = (int32_t *)h->d_ureads.p; p.ufound = (int32_t *)h->d_ufound.p; p.uminpos = (int32_t *)h->d_uminpos.p;
    p.arena = (uint8_t *)h->d_arena.p; p.arena_top = (unsigned long long *)h->d_tops.p; p.arena_cap = h->arena_cap;
    p.out = (uint8_t *)h->d_out.p; p.out_top = (unsigned long long *)h->d_tops.p + 1; p.out_cap = h->out_cap;
    p.n_clist = (unsigned long long *)h->d_tops.p + 2; p.asm_head = (unsigned long long *)h->d_tops.p + 3; p.sw_head = (unsigned long long *)h->d_tops.p + 4; p.n_queue = (unsigned long long *)h->d_tops.p + 5;
    p.order = (uint32_t *)h->d_order.p; p.clist = (unsigned long long *)h->d_clist.p; p.clist_cap = h->d_clist.bytes / 16;      // first half: the contig list; second half: the realigner's long-contig list
    p.sw_long = p.clist + p.clist_cap; p.n_sw_long = (unsigned long long *)h->d_tops.p + 6; p.sw_long_head = (unsigned long long *)h->d_tops.p + 7;
    p.n_queue0 = (unsigned long long *)h->d_tops.p + 8; p.pending = (unsigned long long *)h->d_tops.p + 9; p.queue_cap = (unsigned long long *)h->d_tops.p + 10;      // the dynamic unit queue (bk_asm.hip.h)
    p.order_cap = (uint32_t)std::min<uint64_t>(h->d_order.bytes / 4, 0xFFFFFFF0u); p.pad_q = 0;
    p.k = h->cfg.kmer_size; p.rc_thresh = h->cfg.rc_thresh; p.max_contig = h->cfg.max_contig_len; p.max_read = h->eff_max_read;
    p.max_cand = h->cfg.max_candidates; p.sw_min_score = h->cfg.sw_min_score; p.n_regions = h->n_regions; p.flags = (int32_t)h->cfg.flags;
    p.rmap = nullptr;
}

static size_t asm_lds_bytes(const bk_handle *h, int threads, int max_cand, int max_contig)
{
    const size_t waves = threads / 64, slots = threads == 512 ? waves : 2 * waves;      // the 256-thread build aligns two reads per wavefront (BK_PAIR)
    size_t o = threads == 512 ? at512::ctx_shared_bytes : at256::ctx_shared_bytes;
    o += (size_t)max_cand * 8 + waves * 2 * (h->eff_max_read + 2) * 4 + (size_t)max_cand * 4 + (size_t)2 * max_contig + slots * (h->eff_max_read + 16);
    return (o + 15) / 16 * 16;
}
// The realign stage (bk_sw.hip.h), two tiers: SHORT -- contigs up to BK_SW_SHORT bases on small workgroups with a small LDS block
// (many per CU) --, LONG -- whatever is longer, on 512-thread workgroups with the block sized for max_contig (its list is
// filled by the SHORT tier; idle as a rule).  One tier of 512 threads when the contig cap is short anyway.
#define BK_SW_SHORT 1024
static BkSwTier sw_tier(const bk_handle *h, int contig_cap, int max_contig_for_window, bool small)
{
    BkSwTier T; T.contig_cap = contig_cap; T.mode = 0;
    T.sec_lds = small ? 64 : 256; T.n_flags = small ? 256 : 1024;
    // target staging buffer (packed, 4 bases per byte): the whole window when it fits, else chunks of diagonals
    uint32_t tw_cap = std::min<uint32_t>(h->max_win + 2 * (uint32_t)contig_cap + 16, small ? 16384u : std::max<uint32_t>(131072, 4 * (uint32_t)contig_cap));
    while (bk_sw_layout(contig_cap, tw_cap, T.sec_lds, T.n_flags, h->cfg.sw_min_score).total > BK_LDS_MAX && tw_cap > 2 * (uint32_t)contig_cap + 4096) tw_cap -= 4096;      // long contig caps: shorter chunks of a long window
    T.tw_cap = tw_cap; (void)max_contig_for_window;
    return T;
}
static int launch_sw(bk_handle *h, int max_contig, bool note_occupancy)
{
    static const int env_t = [] { const char *e = bk_diag_env("BK_SW_T"); return e ? atoi(e) : 0; }();              // diagnostic build: workgroup size of the SHORT tier (64 .. 512); 1 = one tier as before round 4
    const bool two = max_contig > BK_SW_SHORT && env_t != 1;
    const int threads = !two ? BK_ST_TMAX : (env_t >= 64 && env_t <= 512 ? env_t : 128);
    BkSwTier T = sw_tier(h, two ? BK_SW_SHORT : max_contig, max_contig, two);
    HIPCHK(h, hipMemsetAsync((unsigned long long *)h->d_tops.p + 6, 0, 16, h->stream));
    size_t lds = bk_sw_layout(T.contig_cap, T.tw_cap, T.sec_lds, T.n_flags, h->cfg.sw_min_score).total;
    HIPCHK(h, set_max_dyn_lds(h->dev, (const void *)bk_sw_kernel, (int)BK_LDS_MAX));
    // one workgroup per contig, pulled from the list the assembler appended to; the number of contigs is only known
    // on the device, so a resident-sized grid of persistent workgroups is launched (idle ones exit at once)
    const int per_cu = cached_occupancy(h->dev, (const void *)bk_sw_kernel, threads, lds);
    if (note_occupancy) h->sw_wg_per_cu = per_cu;
    hipLaunchKernelGGL(bk_sw_kernel, dim3(per_cu * h->n_cu), dim3(threads), lds, h->stream, h->params, T);
    HIPCHK(h, hipGetLastError());
    if (two) {
        BkSwTier TL = sw_tier(h, max_contig, max_contig, false); TL.mode = 1;
        lds = bk_sw_layout(TL.contig_cap, TL.tw_cap, TL.sec_lds, TL.n_flags, h->cfg.sw_min_score).total;
        hipLaunchKernelGGL(bk_sw_kernel, dim3(h->n_cu), dim3(BK_ST_TMAX), lds, h->stream, h->params, TL);
        HIPCHK(h, hipGetLastError());
    }
    return BK_OK;
}
// The caps of the re-run of regions that overflowed one (bk_get_region_status): 4x the configured ones as far as one 512-thread
// workgroup's LDS (the whole CU's) holds them.  false: nothing larger fits (very long reads), the regions fail as they are.
static bool escalated_caps(const bk_handle *h, int &max_cand, int &max_contig)
{
    max_cand = 4 * h->cfg.max_candidates; max_contig = std::min(4 * h->cfg.max_contig_len, BK_MAX_CONTIG_ABS);
    while (max_cand > h->cfg.max_candidates && asm_lds_bytes(h, 512, max_cand, std::min(max_contig, 2 * max_cand)) > BK_LDS_MAX) max_cand /= 2;
    max_contig = std::min(max_contig, 2 * max_cand);
    return max_cand > h->cfg.max_candidates || max_contig > h->cfg.max_contig_len;
}

// The assembler launch of a batch, of a re-run subset and of a repair pass: LDS size, kernel attribute, diagnostic parameters and
// grid in ONE place (they used to be recomputed, differently, by launch_repair).  `units`: entries in the queue at most.
struct BkAsmShape { size_t lds; int per_cu, grid; };
static int asm_shape(bk_handle *h, int threads, int max_cand, int max_contig, long long units, BkAsmShape &sh)
{
    const void *kfn = threads == 512 ? (const void *)at512::bk_asm_kernel : (const void *)at256::bk_asm_kernel_w4;
    size_t lds = asm_lds_bytes(h, threads, max_cand, max_contig);
    { const char *e = bk_diag_env("BK_LDS_PAD"); if (e) lds += (size_t)(atoi(e) & ~15); }      // diagnostic build: guard band behind the block
    if (lds > BK_LDS_MAX) return fail(h, BK_E_LIMIT, "assembler: max_candidates / max_contig_len / read length of this batch need " + std::to_string(lds) + " B of LDS per workgroup, a CU has " + std::to_string(BK_LDS_MAX));
    HIPCHK(h, set_max_dyn_lds(h->dev, kfn, (int)lds));
    // persistent workgroups: as many as are resident at once (a surplus one would only find the queue empty)
    sh.lds = lds; sh.per_cu = cached_occupancy(h->dev, kfn, threads, lds);
    sh.grid = (int)std::max<long long>(1, std::min<long long>(units, (long long)sh.per_cu * h->n_cu));
    return BK_OK;
}
static int launch_asm(bk_handle *h, int threads, int max_cand, int max_contig, long long units, bool note)
{
    BkAsmShape sh; { const int rc = asm_shape(h, threads, max_cand, max_contig, units, sh); if (rc != BK_OK) return rc; }
    const size_t lds = sh.lds; const int grid = sh.grid;
    h->params.asm_lds_pad = 0; h->params.dbg_iters = 0; h->params.poison = 0;
    { const char *e = bk_diag_env("BK_LDS_PAD"); if (e) h->params.asm_lds_pad = (uint32_t)(atoi(e) & ~15); }
    { const char *e = bk_diag_env("BK_DBG_ITERS"); if (e) h->params.dbg_iters = (uint32_t)atoi(e); }
    { const char *e = bk_diag_env("BK_POISON_LDS"); if (e) h->params.poison = 0x100u | (uint32_t)(atoi(e) & 0xFF); }
    h->params.asm_lds_bytes = (uint32_t)lds;
    if (note) { h->asm_wg_per_cu = sh.per_cu; h->asm_threads = threads; }
    if (threads == 512) hipLaunchKernelGGL(at512::bk_asm_kernel, dim3(grid), dim3(512), lds, h->stream, h->params);
    else hipLaunchKernelGGL(at256::bk_asm_kernel_w4, dim3(grid), dim3(256), lds, h->stream, h->params);
    HIPCHK(h, hipGetLastError());
    return BK_OK;
}

// subset == nullptr: the whole batch.  Else: only these regions, from the k-mer stage on (it resets their state), each as ONE unit
// (no component split), with the escalated caps if `escalate`, on top of what the batch's run left in the arenas (bump
// pointers and the contig list go on).
// The pinned host buffers bk_mirror_kernel writes into: the work records, the bump pointers, and BOTH buffers that take turns as
// the host copy of the contig records (up to 8 MB of them: a batch with more -- tens of thousands of contigs -- is fetched by a copy).
static hipError_t ensure_mirrors(bk_handle *h)
{
    const uint64_t want = std::min<uint64_t>(h->out_cap & ~15ull, 8ull << 20);
    hipError_t e = hipSuccess;
    if (h->m_out.cap < want) e = h->m_out.resize(want);
    if (e == hipSuccess && h->h_out.cap < want) { e = h->h_out.resize(want); h->h_out.n = 0; }
    if (e == hipSuccess) e = h->m_work.resize(sizeof(BkRegionWork) * (size_t)std::max(h->n_regions, 1));
    if (e == hipSuccess) e = h->m_tops.resize(11 * sizeof(unsigned long long));
    h->m_out_cap = std::min<uint64_t>(want, std::min(h->m_out.cap, h->h_out.cap) & ~(size_t)15);
    return e;
}

// Last kernel of a batch: the work records, the bump pointers and the contig records (when they fit the host buffer) go to pinned
// host memory over the fabric, written by the shader -- no copy engine, no host-side wait behind another handle's transfers.
extern "C" __global__ void __launch_bounds__(256) bk_mirror_kernel(const uint32_t *work, uint32_t work_words, const unsigned long long *tops, const uint4 *out,
                                                                   uint32_t *m_work, unsigned long long *m_tops, uint4 *m_out, unsigned long long m_out_cap)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < work_words; i += nth) m_work[i] = work[i];
    if (tid < 11) m_tops[tid] = tops[tid];
    const unsigned long long nb = tops[1];
    if (nb <= m_out_cap) for (size_t i = tid; i < (size_t)((nb + 15) / 16); i += nth) m_out[i] = out[i];
}

static int launch(bk_handle *h, uint32_t mask, const std::vector<uint32_t> *subset = nullptr, bool escalate = true)
{
    h->mirror_fresh = false;
    // every contig record takes >= 256 B of the result arena, so out_cap / 256 list entries can never overflow
    HIPCHK(h, h->d_clist.ensure(std::max<uint64_t>(h->out_cap / 256, 1024) * 16));
    uint32_t npad = 1; while ((int)npad < h->n_regions) npad <<= 1;
    HIPCHK(h, h->d_order.ensure((size_t)h->n_regions * 4 * BK_SPLIT_G * (1 + BK_REQUEUE_PASSES))); HIPCHK(h, h->d_skeys.ensure((size_t)npad * 8));      // queue: up to BK_SPLIT_G units per region, and room for the passes split regions append themselves
    fill_params(h);
    const int n_launch = subset ? (int)subset->size() : h->n_regions;
    int max_cand = h->cfg.max_candidates, max_contig = h->cfg.max_contig_len;
    if (subset) {
        if (escalate) escalated_caps(h, max_cand, max_contig);
        HIPCHK(h, h->d_rmap.ensure(subset->size() * 4));
        HIPCHK(h, hipMemcpy(h->d_rmap.p, subset->data(), subset->size() * 4, hipMemcpyHostToDevice));       // (the stream is idle: bk_sync has just waited for it)
        HIPCHK(h, hipMemcpy(h->d_order.p, subset->data(), subset->size() * 4, hipMemcpyHostToDevice));      // the assembler's queue: these regions, in index order
        unsigned long long tops[11];
        HIPCHK(h, hipMemcpy(tops, h->d_tops.p, sizeof(tops), hipMemcpyDeviceToHost));
        BkAsmShape sh; { const int rc = asm_shape(h, 512, max_cand, max_contig, n_launch, sh); if (rc != BK_OK) return rc; }
        // unit queue: these regions, one unit each (no split: nothing is appended); the realigner goes on behind the contigs it has seen
        tops[3] = (unsigned long long)std::min(sh.grid, n_launch); tops[4] = std::min<unsigned long long>(tops[2], h->d_clist.bytes / 16); tops[5] = (unsigned long long)n_launch;
        tops[8] = (unsigned long long)n_launch; tops[9] = 0; tops[10] = (unsigned long long)n_launch;
        HIPCHK(h, hipMemcpy((unsigned long long *)h->d_tops.p + 3, tops + 3, 8 * sizeof(unsigned long long), hipMemcpyHostToDevice));
        h->params.rmap = (const uint32_t *)h->d_rmap.p; h->params.n_regions = n_launch; h->params.max_cand = max_cand; h->params.max_contig = max_contig;
    } else {
        static const unsigned long long tops[11] = {256, 256, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // arena top, out top, contigs listed, unit queue head, contig queue head, units queued, (realigner's long list: 2), initial units, split regions pending, queue capacity (bk_sched_kernel)
        HIPCHK(h, hipMemcpyAsync(h->d_tops.p, tops, sizeof(tops), hipMemcpyHostToDevice, h->stream));
    }
    { const char *e = bk_diag_env("BK_POISON_ARENA"); if (e && h->d_arena.p) HIPCHK(h, hipMemsetAsync(h->d_arena.p, atoi(e) & 0xFF, h->d_arena.bytes, h->stream)); }      // diagnostic: what an uninitialised read of the scratch arena sees
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    // workgroup sizes: latency mode (one batch at a time) or throughput mode (batches in flight / a batch that fills the chip)
    const int asm_threads = subset ? 512 : h->cfg.asm_wg_threads == 256 ? 256 : h->cfg.asm_wg_threads == 512 ? 512 : (h->n_regions > 2 * h->n_cu ? 256 : 512);      // more regions than 512-thread workgroups can be resident at once (2 per CU): the smaller ones keep them all in flight
    const int kmer_threads = asm_threads == 512 ? BK_KT_MAX : BK_KT;
    if (mask & BK_STAGE_KMER) {
        if (h->n_big < h->n_regions) {
            // the LDS first holds the read-grouping table (one word per slot, when it fits), then the reference k-mer set
            const uint32_t lds_words = std::max<uint32_t>(32 + 256 + 2 * h->win_words_cap + h->ref_cap, h->group_words);
            const size_t lds = (size_t)lds_words * 4;
            HIPCHK(h, set_max_dyn_lds(h->dev, (const void *)bk_kmer_kernel, (int)lds));
            hipLaunchKernelGGL(bk_kmer_kernel, dim3(n_launch), dim3(kmer_threads), lds, h->stream, h->params, h->ref_cap, h->win_words_cap, lds_words);
            HIPCHK(h, hipGetLastError());
        }
        if (h->n_big > 0) {
            const size_t lds = (32 + 256 + (size_t)BK_K_PERM_G) * 4;
            HIPCHK(h, set_max_dyn_lds(h->dev, (const void *)bk_kmer_kernel_g, (int)lds));
            hipLaunchKernelGGL(bk_kmer_kernel_g, dim3(n_launch), dim3(kmer_threads), lds, h->stream, h->params);
            HIPCHK(h, hipGetLastError());
        }
    }
    const bool dbg = bk_diag_env("BK_DEBUG_SPLIT") != nullptr;
    if (dbg) {
        const std::pair<const char *, const DevBuf *> bufs[] = {{"desc", &h->d_desc}, {"work", &h->d_work}, {"part", &h->d_part}, {"reads", &h->d_reads}, {"rlen", &h->d_rlen}, {"rflag", &h->d_rflag}, {"sc", &h->d_sc}, {"sclen", &h->d_sclen}, {"win", &h->d_win},
            {"ddslot", &h->d_ddslot}, {"ddrep", &h->d_ddrep}, {"ddcnt", &h->d_ddcnt}, {"grp", &h->d_grp}, {"urep", &h->d_urep}, {"unr", &h->d_unr}, {"ufl", &h->d_ufl}, {"ubuf", &h->d_ubuf}, {"ureads", &h->d_ureads}, {"ufound", &h->d_ufound}, {"uminpos", &h->d_uminpos},
            {"arena", &h->d_arena}, {"out", &h->d_out}, {"tops", &h->d_tops}, {"order", &h->d_order}, {"skeys", &h->d_skeys}, {"clist", &h->d_clist}, {"nlist", &h->d_nlist}, {"wnlist", &h->d_wnlist}, {"rmap", &h->d_rmap}};
        fprintf(stderr, "[bk launch] buffers:");
        for (const auto &b : bufs) fprintf(stderr, " %s %p+%zu", b.first, b.second->p, b.second->bytes);
        fprintf(stderr, "\n");
    }
    if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk launch] k-mer stage done (%d regions%s, arena %.1f MB)\n", n_launch, subset ? ", subset" : "", h->arena_cap / 1048576.0); }
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    if (mask & BK_STAGE_ASSEMBLE) {
        // regions ordered by estimated cost, heaviest first (part of the assembler's measured time: ev[1]..ev[2])
        const bool may_split = !subset && !(h->cfg.flags & BK_F_NO_SPLIT);
        if (!subset) {
            BkAsmShape sh; { const int rc = asm_shape(h, asm_threads, max_cand, max_contig, (long long)n_launch * (may_split ? BK_SPLIT_G : 1), sh); if (rc != BK_OK) return rc; }
            hipLaunchKernelGGL(bk_sched_kernel, dim3(1), dim3(BK_SCHED_T), 0, h->stream, h->params, (unsigned long long *)h->d_skeys.p, npad, (uint32_t)sh.grid, (uint32_t)(sh.per_cu * h->n_cu));
            HIPCHK(h, hipGetLastError());
            if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk launch] sched done\n"); }
        }
        // Workgroup size: 512 threads (8 wavefronts, 8 look-ahead slots, 2 per CU) finish ONE batch soonest; 256 threads (4
        // wavefronts, 8 slots, 4 per CU) give more regions per CU whose serial phases overlap: +14 % regions/s once the
        // chip is full.  bk_config.asm_wg_threads = 256 / 512 chooses; 0 = 512 unless the batch alone fills the chip twice.
        // (a noisy region is split into up to BK_SPLIT_G units on the device, bk_comp.hip.h: the host only knows the bound)
        // (split regions settle inside the assembler: merge + re-queue when components met, link when none did -- no link kernel here)
        { const int rc = launch_asm(h, asm_threads, max_cand, max_contig, (long long)n_launch * (may_split ? BK_SPLIT_G : 1), !subset); if (rc != BK_OK) return rc; }
        if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk launch] assembler done\n"); }
    }
    HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    if (mask & BK_STAGE_REALIGN) { const int rc = launch_sw(h, max_contig, !subset); if (rc != BK_OK) return rc; }
    HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    if (!subset) {
        static_assert(sizeof(BkRegionWork) % 4 == 0, "BkRegionWork is copied word by word");
        HIPCHK(h, ensure_mirrors(h));                    // (a no-op unless the result arena grew since the submit)
        const uint32_t words = (uint32_t)(sizeof(BkRegionWork) / 4 * (size_t)h->n_regions);
        hipLaunchKernelGGL(bk_mirror_kernel, dim3(32), dim3(256), 0, h->stream, (const uint32_t *)h->d_work.p, words, (const unsigned long long *)h->d_tops.p, (const uint4 *)h->d_out.p,
                           (uint32_t *)h->m_work.data(), (unsigned long long *)h->m_tops.data(), (uint4 *)h->m_out.data(), (unsigned long long)h->m_out_cap);
        HIPCHK(h, hipGetLastError());
        h->mirror_fresh = true;
    }
    return BK_OK;
}

// Another pass over split regions whose components met across units (status BK_ST_REDO): merge + reset on the device
// (bk_resolve_kernel), every unit of those regions again (it takes what was re-dealt to it), contigs re-linked, the new ones
// realigned.  Same kernels and caps as the batch's run.
static int launch_repair(bk_handle *h, uint32_t mask, const std::vector<uint32_t> &redo)
{
    h->mirror_fresh = false;
    fill_params(h);
    const int n = (int)redo.size();
    HIPCHK(h, h->d_rmap.ensure(redo.size() * 4));
    HIPCHK(h, hipMemcpy(h->d_rmap.p, redo.data(), redo.size() * 4, hipMemcpyHostToDevice));             // (the stream is idle: bk_sync has just waited for it)
    std::vector<uint32_t> q; q.reserve((size_t)n * BK_SPLIT_G);
    for (uint32_t r : redo) for (uint32_t g = 0; g < h->h_work[r].split; g++) q.push_back(r | (g << BK_QUEUE_UNIT_SHIFT));
    HIPCHK(h, hipMemcpy(h->d_order.p, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    unsigned long long tops[11];
    HIPCHK(h, hipMemcpy(tops, h->d_tops.p, sizeof(tops), hipMemcpyDeviceToHost));
    BkAsmShape sh; { const int rc = asm_shape(h, h->asm_threads, h->cfg.max_candidates, h->cfg.max_contig_len, (long long)q.size(), sh); if (rc != BK_OK) return rc; }
    // (a host-driven pass: the queue holds exactly the units of this pass -- queue_cap leaves no room to append, so a region whose
    //  components meet again comes back with BK_ST_REDO; every region of the pass is pending until its last unit has reported in)
    // (every entry through the queue head, none by block index: the workgroups of this pass wait for *pending, bk_sched.hip.h)
    tops[3] = 0; tops[4] = std::min<unsigned long long>(tops[2], h->d_clist.bytes / 16); tops[5] = (unsigned long long)q.size();
    tops[8] = 0; tops[9] = (unsigned long long)n; tops[10] = (unsigned long long)q.size();
    HIPCHK(h, hipMemcpy((unsigned long long *)h->d_tops.p + 3, tops + 3, 8 * sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    const bool dbg = bk_diag_env("BK_DEBUG_SPLIT") != nullptr;
    hipLaunchKernelGGL(bk_resolve_kernel, dim3(n), dim3(BK_RESOLVE_T), 0, h->stream, h->params, (const uint32_t *)h->d_rmap.p);
    HIPCHK(h, hipGetLastError());
    if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk split]   resolve done\n"); }
    { const int rc = launch_asm(h, h->asm_threads, h->cfg.max_candidates, h->cfg.max_contig_len, (long long)q.size(), false); if (rc != BK_OK) return rc; }
    if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk split]   assembler pass done\n"); }
    hipLaunchKernelGGL(bk_link_kernel, dim3(std::min(h->n_regions, h->n_cu)), dim3(BK_LINK_T), 0, h->stream, h->params, h->n_regions);
    HIPCHK(h, hipGetLastError());
    if (dbg) { HIPCHK(h, hipStreamSynchronize(h->stream)); fprintf(stderr, "[bk split]   link done\n"); }
    HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    if (mask & BK_STAGE_REALIGN) { const int rc = launch_sw(h, h->cfg.max_contig_len, false); if (rc != BK_OK) return rc; }
    HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    return BK_OK;
}

extern "C" int bk_run(bk_handle *h, uint32_t stage_mask)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (!h->submitted) return fail(h, BK_E_STATE, "bk_run: no regions submitted");
    if ((stage_mask & BK_STAGE_ASSEMBLE) && !(stage_mask & BK_STAGE_KMER)) return fail(h, BK_E_ARG, "bk_run: BK_STAGE_ASSEMBLE needs BK_STAGE_KMER in the same run");
    if ((stage_mask & BK_STAGE_REALIGN) && !(stage_mask & BK_STAGE_ASSEMBLE)) return fail(h, BK_E_ARG, "bk_run: BK_STAGE_REALIGN needs BK_STAGE_ASSEMBLE in the same run");
    HIPCHK(h, hipSetDevice(h->dev));
    h->ran_mask = stage_mask; h->fetched = false; h->synced = false; h->calls_valid = false;
    int rc = launch(h, stage_mask);
    if (rc == BK_OK) h->ran = true;
    return rc;
}

static const char *st_name(int s)
{
    switch (s) {
    case BK_ST_ARENA: return "device scratch arena exhausted"; case BK_ST_WINDOW: return "reference window too long";
    case BK_ST_CONTIG: return "contig longer than max_contig_len"; case BK_ST_CAND: return "more candidate reads for one k-mer than max_candidates";
    case BK_ST_KLIST: return "contig k-mer list overflow"; case BK_ST_READLEN: return "read longer than max_read_len";
    case BK_ST_OUT: return "output arena exhausted"; case BK_ST_HITS: return "realign stage: step-1 hit list overflow"; default: return "unknown";
    }
}

// wait, read back the work records and the result arena; grow arenas and rerun when they overflowed
static int sync_impl(bk_handle *h)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (!h->ran) return fail(h, BK_E_STATE, "bk_sync: nothing was run");
    if (h->synced) return BK_OK;
    h->hold_snapshot = false;                       // the work records below replace the fetched copy's
    HIPCHK(h, hipSetDevice(h->dev));
    bool escalated = false; float ms_first[4] = {0, 0, 0, 0};
    h->n_escalated = 0; h->n_repair_passes = 0;
    for (int attempt = 0; attempt < 16; attempt++) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
#if defined(BK_SYNC_CHECK)
        {   // barrier-check build (bk_common.h): did any workgroup's wavefronts meet at different barrier sites?
            unsigned long long rep_[4] = {0, 0, 0, 0};
            HIPCHK(h, hipMemcpyFromSymbol(rep_, HIP_SYMBOL(bk_sync_report), sizeof(rep_)));
            if (rep_[0]) {
                const unsigned long long zero[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(bk_sync_report), zero, sizeof(zero));
                const unsigned a = (unsigned)(rep_[0] - 1), b = (unsigned)rep_[1];
                return fail(h, BK_E_HIP, "BARRIER DIVERGENCE: wavefront " + std::to_string(rep_[3]) + " of workgroup " + std::to_string(rep_[2]) + " stood at barrier site file " + std::to_string(a >> 16) + " line " + std::to_string(a & 0xFFFF) +
                            " while another wavefront of the workgroup stood at file " + std::to_string(b >> 16) + " line " + std::to_string(b & 0xFFFF) + " (file ids: bk_common.h BK_SRC_ID)");
            }
        }
#endif
        h->h_work.resize(h->n_regions);
        if (h->mirror_fresh) memcpy((void *)h->h_work.data(), h->m_work.data(), sizeof(BkRegionWork) * (size_t)h->n_regions);      // written by bk_mirror_kernel behind the batch's kernels
        else HIPCHK(h, hipMemcpy(h->h_work.data(), h->d_work.p, sizeof(BkRegionWork) * h->n_regions, hipMemcpyDeviceToHost));
        // A region that overflowed an assembler cap (candidates per k-mer visit, contig length, k-mer list) is run again, with
        // the others that did, under caps 4x larger (below).  One that overflows those too fails ALONE: its status is kept
        // (bk_get_region_status), it reports no contigs, the other regions of the batch are unaffected.
        bool grow_arena = false, grow_out = false; int bad = 0;
        std::vector<uint32_t> over, redo, unsplit;
        for (int r = 0; r < h->n_regions; r++) {
            int s = h->h_work[r].status;
            if (s == BK_ST_ARENA) grow_arena = true; else if (s == BK_ST_OUT) grow_out = true;
            else if (s == BK_ST_REDO) redo.push_back((uint32_t)r); else if (s == BK_ST_UNSPLIT) unsplit.push_back((uint32_t)r);
            else if (s != BK_ST_OK) { bad++; if (s == BK_ST_CAND || s == BK_ST_CONTIG || s == BK_ST_KLIST) over.push_back((uint32_t)r); }
        }
        h->n_failed = bad;
        if (!grow_arena && !grow_out) {
            float ms[4];
            for (int i = 0; i < 3; i++) { ms[i + 1] = 0; (void)hipEventElapsedTime(&ms[i + 1], h->ev[i], h->ev[i + 1]); }
            ms[0] = 0; (void)hipEventElapsedTime(&ms[0], h->ev[0], h->ev[3]);
            if (bk_diag_env("BK_DEBUG_SPLIT")) {
                unsigned long long np_ = 0, nc_ = 0, nx_ = 0, ns_ = 0; for (int r = 0; r < h->n_regions; r++) { const BkRegionWork &w = h->h_work[r]; if (w.split) { ns_++; np_ += w.n_pairs; nc_ += w.n_conf; nx_ += w.n_cidx; } }
                fprintf(stderr, "[bk split] pass %d: %zu region(s) to repair, %zu to unsplit; split regions %llu, meetings noted %llu (conflicts %llu), contigs emitted so far %llu; kernels %.2f / %.2f / %.2f ms\n",
                        h->n_repair_passes, redo.size(), unsplit.size(), ns_, np_, nc_, nx_, ms[1], ms[2], ms[3]);
                for (int r = 0; r < h->n_regions && r < 4; r++) { const BkRegionWork &w = h->h_work[r]; if (!w.split) continue;
                    fprintf(stderr, "   region %d  U %u M %u M2 %u  open conflicts %u; unit 0: prefix %u us, labelling %u us, seed list %u us; unit us/iterations:", r, w.U, w.M, w.M2, w.n_conf, w.dbg_us[0], w.dbg_us[1], w.dbg_us[2]); for (int g = 0; g < BK_SPLIT_G; g++) fprintf(stderr, " %u/%u", w.unit_us[g], w.unit_iters[g]); fprintf(stderr, "\n"); }
                for (int r = 0; r < h->n_regions; r++) { const BkRegionWork &w = h->h_work[r]; if (w.stamps[14]) fprintf(stderr, "   LDS GUARD region %d: first byte written behind the block at +%llu (value 0x%llx)\n", r, (unsigned long long)w.stamps[14] - 1, (unsigned long long)w.stamps[15]); }
                for (int r = 0; r < h->n_regions; r++) { const BkRegionWork &w = h->h_work[r]; if (w.split && w.stamps[10]) fprintf(stderr, "   CHECK region %d: code %llu value 0x%llx unit/pass 0x%llx serial/seed 0x%llx (U %u M %u M2 %u)\n", r, (unsigned long long)w.stamps[10], (unsigned long long)w.stamps[11], (unsigned long long)w.stamps[12], (unsigned long long)w.stamps[13], w.U, w.M, w.M2); }
                fprintf(stderr, "   all split regions, prefix+labelling+slowest unit (ms):");
                for (int r = 0; r < h->n_regions; r++) { const BkRegionWork &w = h->h_work[r]; if (!w.split) continue; uint32_t mx = 0; for (int g = 0; g < BK_SPLIT_G; g++) mx = std::max(mx, w.unit_us[g]);
                    fprintf(stderr, " %.1f+%.1f+%.1f", w.dbg_us[0] / 1000.0, (w.dbg_us[1] + w.dbg_us[2]) / 1000.0, mx / 1000.0); }
                fprintf(stderr, "\n");
            }
            if (!redo.empty() || !unsplit.empty()) {
                // split regions (bk_comp.hip.h): components that met across units are merged and run again (a few per cent of the
                // region's work per pass); a region whose split bookkeeping overflowed is run again as one unit
                for (int i = 0; i < 4; i++) ms_first[i] += ms[i];
                h->n_repair_passes++;
                int rc = !redo.empty() ? launch_repair(h, h->ran_mask, redo) : launch(h, h->ran_mask, &unsplit, false);
                if (rc != BK_OK) return rc;
                attempt--;                                   // bounded by the passes themselves (components only ever merge)
                if (h->n_repair_passes > 400) return fail(h, BK_E_HIP, "bk_sync: split regions did not settle");
                continue;
            }
            int mc, ml;
            if (!over.empty() && !escalated && !h->cfg.no_escalation && (h->ran_mask & BK_STAGE_ASSEMBLE) && escalated_caps(h, mc, ml)) {
                for (int i = 0; i < 4; i++) ms_first[i] += ms[i];
                escalated = true; h->n_escalated = (int)over.size();
                int rc = launch(h, h->ran_mask, &over);
                if (rc != BK_OK) return rc;
                continue;
            }
            for (int r = 0; r < h->n_regions; r++) if (h->h_work[r].status != BK_ST_OK) { h->h_work[r].n_contigs = 0; h->h_work[r].o_first_contig = 0; }
            for (int i = 0; i < 4; i++) h->ms[i] = ms[i] + ms_first[i];          // the re-run's kernels count
            h->synced = true;
            return BK_OK;
        }
        // the bump pointers keep counting past the capacity: what the regions that got that far asked for is a lower
        // bound of the demand, so grow to that (plus slack) when it is more than the usual factor
        unsigned long long tops[2] = {0, 0};
        HIPCHK(h, hipMemcpy(tops, h->d_tops.p, sizeof(tops), hipMemcpyDeviceToHost));
        // (x4 while the arena is small, x1.5 beyond 4 GB or what was asked for + 25 %: a 256-region configs[4] batch needs ~90 GB
        // and must not jump from 42 to 166)
        if (grow_arena) { h->arena_cap = std::max<uint64_t>(h->arena_cap < (4ull << 30) ? h->arena_cap * 4 : h->arena_cap + h->arena_cap / 2, tops[0] + tops[0] / 4); HIPCHK(h, h->d_arena.ensure(h->arena_cap)); }
        if (grow_out) { h->out_cap = std::max<uint64_t>(h->out_cap * 4, tops[1] + tops[1] / 2); HIPCHK(h, h->d_out.ensure(h->out_cap)); }
        escalated = false; h->n_escalated = 0; h->n_repair_passes = 0; for (float &m : ms_first) m = 0;      // the whole batch again: regions that overflow a cap will be re-run again
        int rc = launch(h, h->ran_mask);
        if (rc != BK_OK) return rc;
    }
    return fail(h, BK_E_NOMEM, "bk_sync: arenas still too small after 16 growth steps");
}

// Public: like the internal wait, but says so when regions of the batch hit a cap: BK_W_REGIONS_FAILED (> 0; every other
// result of the batch is valid, bk_get_region_status names the regions and the caps, bk_get_stat(22) counts them).
extern "C" int bk_sync(bk_handle *h)
{
    const int rc = sync_impl(h);
    if (rc != BK_OK) return rc;
    return h->n_failed > 0 ? BK_W_REGIONS_FAILED : BK_OK;
}

static int fetch(bk_handle *h)
{
    if (h->hold_snapshot) return BK_OK;             // explicit bk_fetch(): keep working on that copy
    HIPCHK(h, hipSetDevice(h->dev));                // entry points may be called from any host thread
    int rc = sync_impl(h);
    if (rc != BK_OK) return rc;
    if (h->fetched) return BK_OK;
    if (h->mirror_fresh && ((const unsigned long long *)h->m_tops.data())[1] <= h->m_out_cap) {      // the records are on the host already: the buffers change roles
        std::swap(h->h_out, h->m_out); h->h_out.n = (size_t)((const unsigned long long *)h->m_tops.data())[1];
        h->mirror_fresh = false;                  // (the work records were taken by sync_impl above; a second fetch of this run finds h->fetched)
        h->fetched = true;
        return BK_OK;
    }
    unsigned long long tops[2];
    HIPCHK(h, hipMemcpy(tops, h->d_tops.p, sizeof(tops), hipMemcpyDeviceToHost));
    HIPCHK(h, h->h_out.resize(tops[1]));
    HIPCHK(h, hipMemcpyAsync(h->h_out.data(), h->d_out.p, tops[1], hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->fetched = true;
    return BK_OK;
}

extern "C" int bk_last_kernel_ms(bk_handle *h, int which, float *ms)
{
    BK_JOIN(h);
    if (!h || !ms || which < 0 || which > 3) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    *ms = h->ms[which]; return BK_OK;
}

static void key_to_str(uint64_t lo, uint64_t hi, int k, char *out)
{
    for (int t = k - 1; t >= 0; t--) { out[t] = "ACGT"[lo & 3u]; lo = (lo >> 2) | (hi << 62); hi >>= 2; }
}

extern "C" int bk_get_kmer_count(bk_handle *h, int32_t region, int32_t *n_mers, int32_t *n_unique)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    if (n_mers) *n_mers = (int32_t)h->h_work[region].M;
    if (n_unique) *n_unique = (int32_t)h->h_work[region].U;
    return BK_OK;
}

extern "C" int bk_get_kmers(bk_handle *h, int32_t region, char *mers, int32_t *counts, int32_t cap)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    const BkRegionWork &w = h->h_work[region];
    const int M = (int)w.M, n = std::min<int>(M, cap), k = h->cfg.kmer_size;
    if (n <= 0) return BK_OK;
    std::vector<uint64_t> lo(M), hi(M); std::vector<uint32_t> c(M);
    HIPCHK(h, hipMemcpy(lo.data(), (uint8_t *)h->d_arena.p + w.o_key_lo, (size_t)M * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(hi.data(), (uint8_t *)h->d_arena.p + w.o_key_hi, (size_t)M * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(c.data(), (uint8_t *)h->d_arena.p + w.o_kcnt, (size_t)M * 4, hipMemcpyDeviceToHost));
    // the device orders the k-mers that can seed a contig (count >= 2) by (count, mer) descending and leaves the
    // count-1 k-mers behind them in table order (nothing on the device depends on their order); the API returns
    // the whole list in the order init_assembly visits it (sv_assembly.py:281)
    std::vector<int> ord(M);
    for (int i = 0; i < M; i++) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { if (c[a] != c[b]) return c[a] > c[b]; if (hi[a] != hi[b]) return hi[a] > hi[b]; return lo[a] > lo[b]; });
    for (int i = 0; i < n; i++) { const int j = ord[i]; if (mers) key_to_str(lo[j], hi[j], k, mers + (size_t)i * k); if (counts) counts[i] = (int32_t)c[j]; }
    return BK_OK;
}

static const BkContigRec *find_contig(bk_handle *h, int region, int contig)
{
    uint64_t off = h->h_work[region].o_first_contig;
    for (int i = 0; off && i < contig; i++) off = ((const BkContigRec *)(h->h_out.data() + off))->next;
    return off ? (const BkContigRec *)(h->h_out.data() + off) : nullptr;
}

extern "C" int bk_get_region_status(bk_handle *h, int32_t region, int32_t *status, const char **text)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    int s = h->h_work[region].status;
    if (status) *status = s;
    if (text) *text = s == BK_ST_OK ? "ok" : st_name(s);
    return BK_OK;
}

extern "C" int bk_get_contig_count(bk_handle *h, int32_t region, int32_t *n)
{
    BK_JOIN(h);
    if (!h || !n || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    *n = (int32_t)h->h_work[region].n_contigs; return BK_OK;
}

// the contig counts of all regions of the batch in one call (a driver that only needs the counts: one call per batch, not per target)
extern "C" int bk_get_contig_counts(bk_handle *h, int32_t *n, int32_t cap)
{
    BK_JOIN(h);
    if (!h || !n || cap < h->n_regions) return BK_E_ARG;
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    for (int r = 0; r < h->n_regions; r++) n[r] = (int32_t)h->h_work[r].n_contigs;
    return BK_OK;
}

extern "C" int bk_get_contig_info(bk_handle *h, int32_t region, int32_t contig, bk_contig_info *info)
{
    BK_JOIN(h);
    if (!h || !info || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = fetch(h); if (rc != BK_OK) return rc;
    const BkContigRec *c = find_contig(h, region, contig);
    if (!c) return fail(h, BK_E_ARG, "bk_get_contig_info: no such contig");
    info->seq_len = c->seq_len; info->counts_len = c->counts_len; info->n_kmers = c->n_kmers; info->n_reads = c->n_reads; info->total_reads = c->total_reads; info->n_hits = c->n_hits;
    return BK_OK;
}

extern "C" int bk_get_contig(bk_handle *h, int32_t region, int32_t contig, char *seq, int32_t *indel_only, int32_t *others, int32_t *kmer_locs, char *kmers, int32_t *reads)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions) return BK_E_ARG;
    int rc = fetch(h); if (rc != BK_OK) return rc;
    const BkContigRec *c = find_contig(h, region, contig);
    if (!c) return fail(h, BK_E_ARG, "bk_get_contig: no such contig");
    const uint8_t *b = (const uint8_t *)c; const int k = h->cfg.kmer_size;
    if (seq) memcpy(seq, b + c->o_seq, c->seq_len);
    if (indel_only) memcpy(indel_only, b + c->o_io, (size_t)c->counts_len * 4);
    if (others) memcpy(others, b + c->o_ot, (size_t)c->counts_len * 4);
    if (kmer_locs) memcpy(kmer_locs, b + c->o_klocs, (size_t)c->seq_len * 4);
    if (kmers) { const uint64_t *kk = (const uint64_t *)(b + c->o_kmers); for (int i = 0; i < c->n_kmers; i++) key_to_str(kk[2 * i], kk[2 * i + 1], k, kmers + (size_t)i * k); }
    if (reads) { memcpy(reads, b + c->o_reads, (size_t)c->n_reads * 4); std::sort(reads, reads + c->n_reads); }
    return BK_OK;
}

// ---- realign contract step 4 (oracle/bk_oracle.h): island fill.  Host code like the chaining it refines: the islands are the
// unaligned rectangles next to chained anchors, a few hundred cells per contig, and they only exist once the chain is known.
static const int BK_FILL_BAND = 16, BK_FILL_MIN = 8;
struct BkFillBlk { int qs, qe, ts, te, score; };
static int fill_best(const char *q, const char *t, int qa, int qb, int ta, int tb, bool use1, int d1, bool use2, int d2, BkFillBlk &out)
{
    int best = 0; out.score = 0; out.qe = out.te = 0;
    int dlo = use1 ? d1 - BK_FILL_BAND : d2 - BK_FILL_BAND, dhi = use1 ? d1 + BK_FILL_BAND : d2 + BK_FILL_BAND;
    if (use2) { dlo = std::min(dlo, d2 - BK_FILL_BAND); dhi = std::max(dhi, d2 + BK_FILL_BAND); }
    for (int d = dlo; d <= dhi; d++) {
        const bool in1 = use1 && d >= d1 - BK_FILL_BAND && d <= d1 + BK_FILL_BAND, in2 = use2 && d >= d2 - BK_FILL_BAND && d <= d2 + BK_FILL_BAND;
        if (!in1 && !in2) continue;
        const int a0 = std::max(qa, ta - d), a1 = std::min(qb, tb - d);
        int h = 0, run = 0;
        for (int a = a0; a < a1; a++) {
            h += (q[a] == t[a + d] && q[a] != 'N') ? 1 : -2; run++;
            if (h <= 0) { h = 0; run = 0; continue; }
            const int qe = a + 1, te = a + 1 + d;
            if (h > best || (h == best && (qe < out.qe || (qe == out.qe && te < out.te)))) { best = h; out.qs = qe - run; out.qe = qe; out.ts = te - run; out.te = te; out.score = h; }
        }
    }
    return best;
}
static void fill_gap(const char *q, const char *t, const BkFillBlk *L, const BkFillBlk *R, int qlo, int qhi, int tlo, int thi, std::vector<BkFillBlk> &v, size_t cap)
{
    if (qhi - qlo < BK_FILL_MIN || thi - tlo < BK_FILL_MIN || v.size() >= cap) return;
    BkFillBlk nb;
    if (fill_best(q, t, qlo, qhi, tlo, thi, L != nullptr, L ? L->te - L->qe : 0, R != nullptr, R ? R->ts - R->qs : 0, nb) < BK_FILL_MIN) return;
    fill_gap(q, t, L, &nb, qlo, nb.qs, tlo, nb.ts, v, cap);
    if (v.size() < cap) v.push_back(nb);
    fill_gap(q, t, &nb, R, nb.qe, qhi, nb.te, thi, v, cap);
}

// Chain the raw hits of one contig into PSL-equivalent records (contract: oracle/bk_oracle.h R2 steps 3 and 4).
// Secondary alignments (step 5) follow as one-block records, ordered by (score desc, target asc, '+' first, query end asc,
// target end asc).  Returns the number of records, or -1 when a chained record needs more than BK_MAX_BLOCKS blocks.
// A record as chain_hits makes it: the scalar fields of bk_psl, blocks without a limit (bk_psl holds BK_MAX_BLOCKS of them).
struct BkPslV {
    int matches = 0, mismatches = 0, rep_matches = 0, q_num_insert = 0, q_base_insert = 0, t_num_insert = 0, t_base_insert = 0;
    int strand = '+', q_size = 0, q_start = 0, q_end = 0, t_index = 0, t_size = 0, t_start = 0, t_end = 0, score = 0;
    std::vector<int> block_sizes, q_starts, t_starts;
};
// Realign contract step 8 (oracle/bk_oracle.h: bko_psl_passes): BLAT's documented output filters -- -minScore ("matches minus the
// mismatches minus some sort of gap penalty", sv_processor.py:840, 843: 20) and -minIdentity (default 90 for nucleotide searches),
// the identity from the milliBad the reference's caller computes for a record (sv_caller.py:954-968).  Integer exact.
#define BK_MIN_IDENTITY 90
static bool psl_passes(const BkPslV &r, int min_score)
{
    const long total = (long)r.matches + r.rep_matches + r.mismatches;
    if ((long)r.matches + r.rep_matches - r.mismatches - r.q_num_insert - r.t_num_insert < (long)min_score) return false;
    const int qali = r.q_end - r.q_start, tali = r.t_end - r.t_start;
    if (std::min(qali, tali) <= 0 || total == 0) return true;
    const int dif = std::max(0, qali - tali);
    const long bad = (long)r.mismatches + r.q_num_insert + lround(3.0 * log(1.0 + (double)dif));
    return 100L * bad <= (long)(100 - BK_MIN_IDENTITY) * total;
}
static void chain_hits(const char *contig, int Q, const std::vector<BkTarget> &targets, std::vector<BkHit> hits, std::vector<BkHit> sec, std::vector<BkPslV> &out, int min_score)
{
    out.clear();
    std::string rc(Q, 'N');
    for (int i = 0; i < Q; i++) { char c = contig[Q - 1 - i]; rc[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
    // ---- step 2b: BLAT's published seeding rule (sv_processor.py:843: -stepSize=10 -minMatch=2, tile 11): a hit is kept only if it
    // holds two index tiles (target positions 10 j .. 10 j + 10 inside it, all eleven bases matching)
    auto seedable = [&](const BkHit &e) {
        const char *q = e.strand == 0 ? contig : rc.c_str(); const BkTarget &t = targets[e.tidx];
        int tiles = 0;
        for (int j = (e.ts + 9) / 10 * 10; j + 11 <= e.te; j += 10) {
            bool ok = true;
            for (int z = 0; z < 11 && ok; z++) { const char a = q[e.qs + (j - e.ts) + z]; ok = a == t[j + z] && a != 'N'; }
            tiles += ok;
        }
        return tiles >= 2; };
    hits.erase(std::remove_if(hits.begin(), hits.end(), [&](const BkHit &e) { return !seedable(e); }), hits.end());
    sec.erase(std::remove_if(sec.begin(), sec.end(), [&](const BkHit &e) { return !seedable(e); }), sec.end());
    std::stable_sort(hits.begin(), hits.end(), [](const BkHit &a, const BkHit &b) { return a.fq < b.fq; });
    const int nh = (int)hits.size();
    auto sec_less = [](const BkHit &a, const BkHit &b) {
        if (a.score != b.score) return a.score > b.score;
        if (a.tidx != b.tidx) return a.tidx < b.tidx;
        if (a.strand != b.strand) return a.strand < b.strand;
        if (a.qe != b.qe) return a.qe < b.qe;
        return a.te < b.te; };
    std::sort(sec.begin(), sec.end(), sec_less);
    // ---- step 6: placement of ambiguous hits (oracle/bk_oracle.h).  A step-1 hit with equal alternatives (a secondary alignment
    // covering its query interval with the same score over it) chooses, together with its neighbours, the placement with the
    // fewest chain breaks, then the smallest sum of diagonal shifts (dynamic programme over the hits in forward query order).
    if (!sec.empty() && nh > 0) {
        struct PCand { int qs, qe, ts, te, strand, tidx, from; };
        std::vector<std::vector<PCand>> cd(nh); std::vector<std::vector<long long>> cost(nh); std::vector<std::vector<int>> back(nh);
        for (int x = 0; x < nh; x++) {
            const BkHit &hx = hits[x];
            const int fs = hx.strand == 0 ? hx.qs : Q - hx.qe, fe = hx.strand == 0 ? hx.qe : Q - hx.qs;
            cd[x].push_back(PCand{hx.qs, hx.qe, hx.ts, hx.te, hx.strand, hx.tidx, -1});
            for (int y = 0; y < (int)sec.size(); y++) {
                const BkHit &e = sec[y];
                const int sfs = e.strand == 0 ? e.qs : Q - e.qe, sfe = e.strand == 0 ? e.qe : Q - e.qs;
                if (sfs > fs || sfe < fe) continue;
                const int cqs = e.strand == 0 ? fs : Q - fe, cqe = e.strand == 0 ? fe : Q - fs, dg = e.ts - e.qs;
                const char *qstr = e.strand == 0 ? contig : rc.c_str(); const BkTarget &t = targets[e.tidx];
                int sc = 0; for (int z = cqs; z < cqe; z++) sc += (qstr[z] == t[z + dg] && qstr[z] != 'N') ? 1 : -2;
                if (sc != hx.score) continue;
                cd[x].push_back(PCand{cqs, cqe, cqs + dg, cqe + dg, e.strand, e.tidx, y});
            }
            cost[x].assign(cd[x].size(), 0); back[x].assign(cd[x].size(), 0);
        }
        for (int x = 1; x < nh; x++) for (size_t b = 0; b < cd[x].size(); b++) {
            long long bestc = -1; int besta = 0;
            for (size_t a = 0; a < cd[x - 1].size(); a++) {
                const PCand &A = cd[x - 1][a], &B = cd[x][b]; long long t = 1ll << 24;                 // a chain break
                if (A.tidx == B.tidx && A.strand == B.strand) {
                    const PCand &first = A.strand == 0 ? A : B, &second = A.strand == 0 ? B : A;
                    const int ov = first.te - second.ts;
                    if (!(ov > 0 && (2 * ov >= first.qe - first.qs || 2 * ov >= second.qe - second.qs)) && second.qs >= first.qe) {
                        t = (long long)(second.ts - second.qs) - (long long)(first.ts - first.qs); if (t < 0) t = -t;
                    }
                }
                if (bestc < 0 || cost[x - 1][a] + t < bestc) { bestc = cost[x - 1][a] + t; besta = (int)a; }
            }
            cost[x][b] = bestc; back[x][b] = besta;
        }
        int pick = 0; for (size_t b = 1; b < cd[nh - 1].size(); b++) if (cost[nh - 1][b] < cost[nh - 1][pick]) pick = (int)b;
        std::vector<char> gone(sec.size(), 0); std::vector<BkHit> demoted;
        for (int x = nh - 1; x >= 0; x--) {
            const PCand c = cd[x][pick];
            if (c.from >= 0) {
                BkHit &hx = hits[x];
                demoted.push_back(hx); gone[c.from] = 1;
                hx.qs = c.qs; hx.qe = c.qe; hx.ts = c.ts; hx.te = c.te; hx.strand = c.strand; hx.tidx = c.tidx;      // fq (forward start) is unchanged
            }
            pick = back[x][pick];
        }
        if (!demoted.empty()) {
            std::vector<BkHit> keep; for (size_t y = 0; y < sec.size(); y++) if (!gone[y]) keep.push_back(sec[y]);
            keep.insert(keep.end(), demoted.begin(), demoted.end());
            sec.swap(keep); std::sort(sec.begin(), sec.end(), sec_less);
        }
    }
    // a chain takes every later free hit that is collinear with it; one that is not is passed over and gets a record of its own
    std::vector<char> taken(nh, 0);
    for (int i = 0; i < nh; i++) {
        if (taken[i]) continue;
        std::vector<BkHit> chain; chain.push_back(hits[i]); taken[i] = 1;           // strand order
        for (int j = i + 1; j < nh; j++) {
            if (taken[j]) continue;
            BkHit h = hits[j];
            if (h.tidx != chain.front().tidx || h.strand != chain.front().strand) continue;
            BkHit *first = h.strand == 0 ? &chain.back() : &h, *second = h.strand == 0 ? &h : &chain.front();
            const int ov = first->te - second->ts;
            if (ov > 0 && (2 * ov >= first->qe - first->qs || 2 * ov >= second->qe - second->qs)) continue;
            if (second->qs < first->qe) continue;
            if (ov > 0) { second->qs += ov; second->ts += ov; }         // trim micro-homology from the later hit
            if (h.strand == 0) chain.push_back(h); else chain.insert(chain.begin(), h);
            taken[j] = 1;
        }
        out.emplace_back(); BkPslV *r = &out.back();
        const BkHit &f = chain.front();
        const char *qstr = f.strand == 0 ? contig : rc.c_str(); const BkTarget &t = targets[f.tidx];
        r->strand = f.strand == 0 ? '+' : '-'; r->q_size = Q; r->t_index = f.tidx; r->t_size = (int32_t)t.size();
        std::vector<BkFillBlk> anch, blocks;
        for (const BkHit &c : chain) anch.push_back(BkFillBlk{c.qs, c.qe, c.ts, c.te, c.score});
        const size_t bcap = (size_t)Q + 2; const int T = (int)t.size(), na = (int)anch.size();      // blocks are disjoint on the query: never reached
        fill_gap(qstr, t.c_str(), nullptr, &anch[0], 0, anch[0].qs, std::max(0, anch[0].ts - anch[0].qs - BK_FILL_BAND), anch[0].ts, blocks, bcap);
        for (int c = 0; c < na; c++) {
            blocks.push_back(anch[c]);
            if (c + 1 < na) fill_gap(qstr, t.c_str(), &anch[c], &anch[c + 1], anch[c].qe, anch[c + 1].qs, anch[c].te, anch[c + 1].ts, blocks, bcap);
        }
        fill_gap(qstr, t.c_str(), &anch[na - 1], nullptr, anch[na - 1].qe, Q, anch[na - 1].te, std::min(T, anch[na - 1].te + (Q - anch[na - 1].qe) + BK_FILL_BAND), blocks, bcap);
        r->t_start = blocks.front().ts; r->t_end = blocks.back().te;
        r->q_start = f.strand == 0 ? blocks.front().qs : Q - blocks.back().qe; r->q_end = f.strand == 0 ? blocks.back().qe : Q - blocks.front().qs;
        int pq = -1, pt = -1;
        for (const BkFillBlk &c : blocks) {
            r->score += c.score;
            const int bs = c.qe - c.qs;
            for (int z = 0; z < bs; z++) { if (qstr[c.qs + z] == t[c.ts + z] && qstr[c.qs + z] != 'N') { if (t.masked(c.ts + z)) r->rep_matches++; else r->matches++; } else r->mismatches++; }
            if (pq >= 0) { if (c.qs > pq) { r->q_num_insert++; r->q_base_insert += c.qs - pq; } if (c.ts > pt) { r->t_num_insert++; r->t_base_insert += c.ts - pt; } }
            r->block_sizes.push_back(bs); r->q_starts.push_back(c.qs); r->t_starts.push_back(c.ts);
            pq = c.qe; pt = c.te;
        }
        if (!psl_passes(*r, min_score)) out.pop_back();                      // step 8: BLAT would not print it
    }
    for (const BkHit &e : sec) {
        out.emplace_back(); BkPslV *r = &out.back();
        const char *qstr = e.strand == 0 ? contig : rc.c_str(); const BkTarget &t = targets[e.tidx];
        r->strand = e.strand == 0 ? '+' : '-'; r->q_size = Q; r->t_index = e.tidx; r->t_size = (int32_t)t.size();
        r->t_start = e.ts; r->t_end = e.te; r->q_start = e.strand == 0 ? e.qs : Q - e.qe; r->q_end = e.strand == 0 ? e.qe : Q - e.qs;
        for (int z = 0; z < e.qe - e.qs; z++) { if (qstr[e.qs + z] == t[e.ts + z] && qstr[e.qs + z] != 'N') { if (t.masked(e.ts + z)) r->rep_matches++; else r->matches++; } else r->mismatches++; }
        r->block_sizes.push_back(e.qe - e.qs); r->q_starts.push_back(e.qs); r->t_starts.push_back(e.ts); r->score = e.score;
        if (!psl_passes(*r, min_score)) out.pop_back();                      // step 8
    }
}
static BkTarget make_target(const char *s, int len)
{
    BkTarget t; t.seq.assign(s, (size_t)len);
    bool any = false;
    for (int i = 0; i < len; i++) if (t.seq[(size_t)i] >= 'a' && t.seq[(size_t)i] <= 'z') { any = true; break; }
    if (any) { t.soft.assign((size_t)len, 0); for (int i = 0; i < len; i++) { char &c = t.seq[(size_t)i]; if (c >= 'a' && c <= 'z') { t.soft[(size_t)i] = 1; c = (char)(c - 'a' + 'A'); } } }
    return t;
}
static void raw_hits_of(const bk_handle *h, const BkContigRec *c, std::vector<BkHit> &prim, std::vector<BkHit> &sec)
{
    prim.clear(); sec.clear();
    if (c->n_hits > 0 && c->hits_off) { const BkHit *p = (const BkHit *)(h->h_out.data() + c->hits_off); prim.assign(p, p + c->n_hits); sec.assign(p + c->n_hits, p + c->n_hits + c->n_sec); }
}

extern "C" int bk_get_hits(bk_handle *h, int32_t region, int32_t contig, bk_psl *hits, int32_t cap)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions || !hits) return BK_E_ARG;
    int rc = fetch(h); if (rc != BK_OK) return rc;
    const BkContigRec *c = find_contig(h, region, contig);
    if (!c) return fail(h, BK_E_ARG, "bk_get_hits: no such contig");
    std::vector<BkHit> raw, sec; raw_hits_of(h, c, raw, sec);
    const char *seq = (const char *)c + c->o_seq;
    std::string s(seq, c->seq_len);
    std::vector<BkPslV> recs;
    chain_hits(s.c_str(), c->seq_len, h->h_targets[region], raw, sec, recs, h->cfg.sw_min_score);
    for (const BkPslV &x : recs) if (x.block_sizes.size() > (size_t)BK_MAX_BLOCKS) return fail(h, BK_E_LIMIT, "bk_get_hits: a chained record of this contig has more than BK_MAX_BLOCKS blocks (bk_psl cannot hold it; bk_call has no such limit)");
    for (size_t i = 0; i < recs.size() && (int32_t)i < cap; i++) {
        const BkPslV &x = recs[i]; bk_psl *r = &hits[i]; memset(r, 0, sizeof(*r));
        r->matches = x.matches; r->mismatches = x.mismatches; r->rep_matches = x.rep_matches; r->q_num_insert = x.q_num_insert; r->q_base_insert = x.q_base_insert;
        r->t_num_insert = x.t_num_insert; r->t_base_insert = x.t_base_insert; r->strand = x.strand; r->q_size = x.q_size; r->q_start = x.q_start; r->q_end = x.q_end;
        r->t_index = x.t_index; r->t_size = x.t_size; r->t_start = x.t_start; r->t_end = x.t_end; r->score = x.score; r->block_count = (int32_t)x.block_sizes.size();
        for (size_t b = 0; b < x.block_sizes.size(); b++) { r->block_sizes[b] = x.block_sizes[b]; r->q_starts[b] = x.q_starts[b]; r->t_starts[b] = x.t_starts[b]; }
    }
    return (int)recs.size();                                                 // the number of records (>= 0, may exceed cap)
}

// The same records without a block limit, as a flat int32 stream: per record BK_PSL_FLAT_HEAD scalars (matches, mismatches,
// rep_matches, n_count, q_num_insert, q_base_insert, t_num_insert, t_base_insert, strand, q_size, q_start, q_end, t_index, t_size,
// t_start, t_end, score, block_count) followed by block_count block sizes, q starts and t starts.
extern "C" int bk_get_hits_flat(bk_handle *h, int32_t region, int32_t contig, int32_t *buf, size_t cap, size_t *needed)
{
    BK_JOIN(h);
    if (!h || region < 0 || region >= h->n_regions || !needed || (cap > 0 && !buf)) return BK_E_ARG;
    int rc = fetch(h); if (rc != BK_OK) return rc;
    const BkContigRec *c = find_contig(h, region, contig);
    if (!c) return fail(h, BK_E_ARG, "bk_get_hits_flat: no such contig");
    std::vector<BkHit> raw, sec; raw_hits_of(h, c, raw, sec);
    std::string s((const char *)c + c->o_seq, c->seq_len);
    std::vector<BkPslV> recs;
    chain_hits(s.c_str(), c->seq_len, h->h_targets[region], raw, sec, recs, h->cfg.sw_min_score);
    size_t need = 0; for (const BkPslV &x : recs) need += BK_PSL_FLAT_HEAD + 3 * x.block_sizes.size();
    *needed = need;
    if (need <= cap) {
        int32_t *o = buf;
        for (const BkPslV &x : recs) {
            const int32_t head[BK_PSL_FLAT_HEAD] = {x.matches, x.mismatches, x.rep_matches, 0, x.q_num_insert, x.q_base_insert, x.t_num_insert, x.t_base_insert, x.strand, x.q_size, x.q_start, x.q_end,
                                                    x.t_index, x.t_size, x.t_start, x.t_end, x.score, (int32_t)x.block_sizes.size()};
            memcpy(o, head, sizeof(head)); o += BK_PSL_FLAT_HEAD;
            for (int v : x.block_sizes) *o++ = v; for (int v : x.q_starts) *o++ = v; for (int v : x.t_starts) *o++ = v;
        }
    }
    return (int)recs.size();
}

extern "C" int bk_get_stat(bk_handle *h, int which, uint64_t *value)
{
    BK_JOIN(h);
    if (!h || !value) return BK_E_ARG;
    if (which == 20 || which == 21) { *value = (uint64_t)((which == 20 ? h->submit_pack_ms : h->submit_h2d_ms) * 1000.0); return BK_OK; }
    int rc = sync_impl(h); if (rc != BK_OK) return rc;
    uint64_t v = 0;
    for (int r = 0; r < h->n_regions; r++) {
        const BkRegionWork &w = h->h_work[r];
        switch (which) { case 0: v += w.nw_cells; break; case 1: v += w.nw_calls; break; case 2: v += w.sw_cells; break; case 4: v += w.U; break; case 5: v += w.M; break; case 6: v += w.n_contigs; break; case 7: v += w.T; break; case 8: v += w.tcap; break; case 30: v += w.dp_sweeps; break; case 31: v += w.dp_redos; break; default: break; }
    }
    if (which == 3) v = h->alg_bytes;
    if (which == 20) v = (uint64_t)(h->submit_pack_ms * 1000.0);              // microseconds
    if (which == 21) v = (uint64_t)(h->submit_h2d_ms * 1000.0);
    if (which == 22) v = (uint64_t)h->n_failed;
    if (which == 26) v = (uint64_t)h->n_escalated;
    if (which == 27) { v = 0; for (int r = 0; r < h->n_regions; r++) v = std::max<uint64_t>(v, h->h_work[r].pass); }      // repair passes of split regions (the most any region needed; in-kernel and host-driven ones)
    if (which == 29) v = (uint64_t)h->n_repair_passes;                                                                          // ... of which the host drove
    if (which == 28) { v = 0; for (int r = 0; r < h->n_regions; r++) v += h->h_work[r].split ? 1 : 0; }          // regions that were split into units
    if (which == 23) v = (uint64_t)h->asm_wg_per_cu;                        // resident assembler / realign workgroups per CU (occupancy query)
    if (which == 24) v = (uint64_t)h->sw_wg_per_cu;
    if (which == 25) v = (uint64_t)h->asm_threads;
    if (which >= 100 && which < 120) { v = 0; for (int r = 0; r < h->n_regions; r++) v += h->h_work[r].stamps[which - 100]; }      // diagnostic counters, summed over regions
    *value = v; return BK_OK;
}

// ---- batched olc.nw (G1 known-answer tests and the DP micro-benchmark) -------------------------------------
// seqs: ASCII, pairs (off1,len1,off2,len2) into `seqs`; out: 4 ints per pair (j_start, i_end, i_start, score);
// reps > 1 repeats each DP (timing); *ms receives the kernel time.
extern "C" int bk_nw_batch(bk_handle *h, const char *seqs, size_t seq_bytes, const uint32_t *off1, const uint32_t *len1,
                           const uint32_t *off2, const uint32_t *len2, int32_t n_pairs, int32_t reps, int32_t transposed, int32_t *out, float *ms)
{
    BK_JOIN(h);
    if (!h || !seqs || n_pairs <= 0 || !out) return BK_E_ARG;
    HIPCHK(h, hipSetDevice(h->dev));
    std::vector<uint8_t> codes(seq_bytes);
    for (size_t i = 0; i < seq_bytes; i++) { char ch = seqs[i]; codes[i] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }
    uint32_t maxm = 0, maxn = 0;
    for (int i = 0; i < n_pairs; i++) { maxm = std::max(maxm, len1[i]); maxn = std::max(maxn, len2[i]); if (len1[i] == 0 || len2[i] == 0) return fail(h, BK_E_ARG, "bk_nw_batch: empty sequence"); }
    if (maxm > 4095 || maxn > 4095) return fail(h, BK_E_LIMIT, "bk_nw_batch: sequence longer than 4095");
    DevBuf dc, d1, d2, d3, d4, dout;
    size_t nb = (size_t)n_pairs * 4;
    HIPCHK(h, dc.ensure(seq_bytes)); HIPCHK(h, d1.ensure(nb)); HIPCHK(h, d2.ensure(nb)); HIPCHK(h, d3.ensure(nb)); HIPCHK(h, d4.ensure(nb)); HIPCHK(h, dout.ensure(nb * 4));
    HIPCHK(h, hipMemcpy(dc.p, codes.data(), seq_bytes, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d1.p, off1, nb, hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(d2.p, len1, nb, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d3.p, off2, nb, hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(d4.p, len2, nb, hipMemcpyHostToDevice));
    const size_t lds = (transposed >= 5 && transposed <= 17) ? (size_t)2 * (((maxm + 15) & ~15u) + ((maxn + 15) & ~15u)) + 256      // bk_nw_pair / the score sweep on pairs: two pairs of sequences, parameters, results
                                       : ((maxm + 15) & ~15u) + ((maxn + 15) & ~15u) + (size_t)2 * (std::max(maxm, maxn) + 2) * 4 + 64;      // (+ 8 result ints of modes 18-20)
    HIPCHK(h, hipFuncSetAttribute((const void *)bk_nw_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    hipLaunchKernelGGL(bk_nw_batch_kernel, dim3(n_pairs), dim3(64), lds, h->stream, (const uint8_t *)dc.p, (const uint32_t *)d1.p, (const uint32_t *)d2.p,
                       (const uint32_t *)d3.p, (const uint32_t *)d4.p, (int32_t *)dout.p, reps < 1 ? 1 : reps, transposed);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev[5], h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ms) (void)hipEventElapsedTime(ms, h->ev[4], h->ev[5]);
    HIPCHK(h, hipMemcpy(out, dout.p, nb * 4, hipMemcpyDeviceToHost));
    dc.release(); d1.release(); d2.release(); d3.release(); d4.release(); dout.release();
    return BK_OK;
}

// ---- native SV-call tail (C1-C3; same semantics as breakmer_amd/sv_caller.py, pinned by tests/golden/caller.json) ----
// bk_call_text: CPU-only entry used by the G5 parity test: one contig described entirely by `text`
// (format: bk_call.h parse_context).  Writes the tab-joined 13-field row (or "") and the target_hit flag.
extern "C" int bk_call_text(const char *text, char *out, size_t cap, int *target_hit)
{
    bkcall::Context cx; std::string err;
    if (!text || !out || !bkcall::parse_context(text, cx, err) || cx.regions.empty()) { g_create_err = "bk_call_text: " + err; return BK_E_ARG; }
    bkcall::Contig ct; ct.seq = cx.c_seq; ct.id = cx.c_id; ct.io = cx.c_io.data(); ct.ot = cx.c_ot.data(); ct.clen = (int)cx.c_ot.size(); ct.klocs = cx.c_kl.data(); ct.nkmers = cx.c_nkmers; ct.same_read_tag = cx.c_same != 0;
    std::vector<bkcall::Psl> rows = cx.rows;
    for (auto &p : rows) {
        if (cx.has_tname) p.tname = cx.tname;
        if (cx.has_offset) { p.tstart += cx.offset; p.tend += cx.offset; for (auto &v : p.ts) v += cx.offset; }
    }
    if (target_hit) *target_hit = rows.empty() ? -1 : (bkcall::target_hit(cx.opts, cx.regions[0], cx.tables, ct, rows) ? 1 : 0);
    std::vector<std::string> row; std::string s;
    if (bkcall::get_result(cx.opts, cx.regions[0], cx.tables, ct, rows, row)) s = bkcall::join_row(row);
    if (s.size() + 1 > cap) return BK_E_LIMIT;
    memcpy(out, s.c_str(), s.size() + 1);
    return BK_OK;
}

extern "C" int bk_fetch(bk_handle *h)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (!h->ran) return fail(h, BK_E_STATE, "bk_fetch: nothing was run");
    h->hold_snapshot = false;
    int rc = fetch(h);
    if (rc == BK_OK) h->hold_snapshot = true;
    return rc;
}

// batch: annotation/query_region context for the submitted regions (text, bk_call.h), then calls for every contig
extern "C" int bk_set_call_context(bk_handle *h, const char *text)
{
    BK_JOIN(h);
    if (!h || !text) return BK_E_ARG;
    bkcall::Tables prev = std::move(h->call_ctx.tables); const bool had = h->have_tables;
    h->call_ctx = bkcall::Context(); h->have_ctx = false; h->have_tables = false; h->calls_valid = false; std::string err;
    if (!bkcall::parse_context(text, h->call_ctx, err)) return fail(h, BK_E_ARG, "bk_set_call_context: " + err);
    if (h->call_ctx.keep_tables) {
        if (!had) return fail(h, BK_E_STATE, "bk_set_call_context: keep_tables without an earlier context on this handle");
        h->call_ctx.tables = std::move(prev);
    }
    h->have_tables = true;
    if ((int)h->call_ctx.regions.size() != h->n_regions) return fail(h, BK_E_ARG, "bk_set_call_context: region count differs from the submitted batch");
    h->have_ctx = true;
    return BK_OK;
}

// replaces, per contig: contig.query_ref + check_target_blat + make_calls (sv_processor.py:823-866).  Result: text,
// one line per called contig: "<region>\t<contig index>\t<13 tab-separated fields>".
static int call_impl(bk_handle *h)
{
    int rc = fetch(h); if (rc != BK_OK) return rc;
    h->calls_blob.clear();
    const bkcall::Context &cx = h->call_ctx;
    // regions are independent: a few host threads, results concatenated in region order
    // (a thread per 32 regions, or per 512 contigs where the regions are few and heavy: 64 noisy regions are 76,000 contigs)
    uint64_t n_contigs = 0; for (int r = 0; r < h->n_regions; r++) n_contigs += h->h_work[r].n_contigs;
    const int want = std::max<int>((h->n_regions + 31) / 32, (int)std::min<uint64_t>(n_contigs / 512, 16));
    const int nthreads = std::max(1, std::min<int>({16, (int)std::thread::hardware_concurrency(), h->n_regions, want}));
    std::vector<std::string> parts(h->n_regions);
    auto work = [&](int t) {
        for (int r = t; r < h->n_regions; r += nthreads) {
            const bkcall::Region &rg = cx.regions[r];
            std::string &blob = parts[r];
            uint64_t off = h->h_work[r].o_first_contig; int ci = 0;
            for (; off; ci++) {
                const BkContigRec *c = (const BkContigRec *)(h->h_out.data() + off); off = c->next;
                const uint8_t *b = (const uint8_t *)c;
                // The bulk of a noisy region's contigs are a few reads that share a sequencing error: ONE gap-free alignment over the
                // whole contig on the target window, nothing else.  Its single record spans the query and has no gap bases, so
                // check_blat_indel (sv_caller.py:621-651) keeps no indel (ngap_total() = 0 < indel_size), has no other hit to make
                // an event from, and the contig has no row -- (or BLAT's seeding rule drops the hit and there is no record at all):
                // no call either way, decided from the raw hit without building the records.
                if (c->n_hits == 1 && c->n_sec == 0 && c->hits_off && cx.opts.indel_size > 0 && !(h->cfg.flags & BK_F_NO_CALL_SHORTCUT)) {
                    const BkHit *h1 = (const BkHit *)(h->h_out.data() + c->hits_off);
                    if (h1->tidx == 0 && h1->qs == 0 && h1->qe == c->seq_len) continue;
                }
                bkcall::Contig ct; ct.seq.assign((const char *)b + c->o_seq, c->seq_len); ct.id = "contig" + std::to_string(ci + 1);
                ct.io = (const int *)(b + c->o_io); ct.ot = (const int *)(b + c->o_ot); ct.clen = c->counts_len; ct.klocs = (const int *)(b + c->o_klocs); ct.nkmers = c->n_kmers;
                { const uint32_t *rd = (const uint32_t *)(b + c->o_reads); const std::string &tg = cx.rtags[r]; bool same = true; char first = 0;
                  for (int i = 0; i < c->n_reads; i++) { char tch = rd[i] < tg.size() ? tg[rd[i]] : '0'; if (i == 0) first = tch; else if (tch != first) same = false; } ct.same_read_tag = same && c->n_reads > 0; }
                std::vector<BkHit> raw, sec; raw_hits_of(h, c, raw, sec);
                std::vector<BkPslV> recs;
                chain_hits(ct.seq.c_str(), c->seq_len, h->h_targets[r], raw, sec, recs, h->cfg.sw_min_score);
                const int n = (int)recs.size();
                // repeats_lower: the BLAT call against the target window has -repeats=lower (sv_processor.py:843), the genome-wide gfClient
                // call (:840) has not -- there a match on a soft-masked base is a plain match
                auto to_psl = [&](const BkPslV &x, const std::string &tname, int offset, bool repeats_lower) {
                    bkcall::Psl p; p.matches = repeats_lower ? x.matches : x.matches + x.rep_matches; p.mis = x.mismatches; p.rep = repeats_lower ? x.rep_matches : 0; p.qni = x.q_num_insert; p.qbi = x.q_base_insert; p.tni = x.t_num_insert; p.tbi = x.t_base_insert;
                    p.strand = (char)x.strand; p.qsize = x.q_size; p.qstart = x.q_start; p.qend = x.q_end; p.tname = bkcall::strip_chr(tname); p.tsize = x.t_size; p.tstart = x.t_start + offset; p.tend = x.t_end + offset;
                    for (size_t i = 0; i < x.block_sizes.size(); i++) { p.bs.push_back(x.block_sizes[i]); p.qs.push_back(x.q_starts[i]); p.ts.push_back(x.t_starts[i] + offset); }
                    return p; };
                std::vector<bkcall::Psl> own, rows;
                for (int i = 0; i < n; i++) if (recs[i].t_index == 0) own.push_back(to_psl(recs[i], rg.chrom, rg.start - 200, true));
                if (!own.empty() && bkcall::target_hit(cx.opts, rg, cx.tables, ct, own)) rows = own;          // the '.mod' rows (Q14)
                else for (int i = 0; i < n; i++) {
                    const int ti = recs[i].t_index;
                    if (ti == 0) rows.push_back(to_psl(recs[i], rg.chrom, rg.start - 200, false));
                    else if (ti - 1 < (int)cx.partners[r].size()) rows.push_back(to_psl(recs[i], cx.partners[r][ti - 1].first, cx.partners[r][ti - 1].second, false));
                }
                std::vector<std::string> row;
                if (bkcall::get_result(cx.opts, rg, cx.tables, ct, rows, row)) { blob += std::to_string(r) + "\t" + std::to_string(ci) + "\t" + bkcall::join_row(row) + "\n"; }
            }
        }
    };
    if (nthreads == 1) work(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
    for (auto &s2 : parts) h->calls_blob += s2;
    h->hold_snapshot = false;                       // the next getter refers to the newest run again
    h->calls_valid = true;
    return BK_OK;
}
extern "C" int bk_call(bk_handle *h)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (!h->have_ctx) return fail(h, BK_E_STATE, "bk_call: bk_set_call_context first");
    if (h->calls_valid && !h->hold_snapshot) return BK_OK;      // bk_call_async has made them
    return call_impl(h);
}
// The same on the handle's own thread: returns at once; the wait for the run, the copy of its records and the call tail happen
// while the caller prepares its next batch.  Every later call on the handle joins that thread first (and reports its error);
// bk_call() / bk_get_calls() then find the calls made.
extern "C" int bk_call_async(bk_handle *h)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (!h->have_ctx) return fail(h, BK_E_STATE, "bk_call_async: bk_set_call_context first");
    if (!h->ran) return fail(h, BK_E_STATE, "bk_call_async: nothing was run");
    h->worker_rc = BK_OK; h->has_worker = true;
    h->worker = std::thread([h]() { tl_is_worker = true; h->worker_rc = call_impl(h); });
    return BK_OK;
}

// Give the large device buffers back (a handle kept in a pool between runs would otherwise hold on to the arena its largest
// batch grew: ~90 GB after one BASELINE configs[4] batch).  Buffers come back, sized by the next batch, with the next submit.
extern "C" int bk_trim(bk_handle *h, uint64_t keep_bytes)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_arena.bytes > keep_bytes) { h->d_arena.release(); h->arena_cap = 0; }
    if (h->d_out.bytes > keep_bytes) { h->d_out.release(); h->out_cap = 0; h->d_clist.release(); }
    DevBuf *bufs[] = {&h->d_reads_in, &h->d_reads, &h->d_ddslot, &h->d_ddrep, &h->d_ddcnt, &h->d_grp, &h->d_urep, &h->d_unr, &h->d_ufl, &h->d_ubuf, &h->d_ureads, &h->d_ufound, &h->d_uminpos};
    for (auto b : bufs) if (b->bytes > keep_bytes) b->release();
    if (h->h_out.cap > keep_bytes) h->h_out.release();
    if (h->hs_reads.cap > keep_bytes) { h->hs_reads.release(); }
    h->submitted = false; h->ran = false; h->fetched = false; h->synced = false; h->hold_snapshot = false;      // the batch is gone with its buffers
    return BK_OK;
}

extern "C" int bk_get_calls(bk_handle *h, char *buf, size_t cap, size_t *needed)
{
    BK_JOIN(h);
    if (!h) return BK_E_ARG;
    if (needed) *needed = h->calls_blob.size() + 1;
    if (buf && cap >= h->calls_blob.size() + 1) memcpy(buf, h->calls_blob.c_str(), h->calls_blob.size() + 1);
    return BK_OK;
}


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
