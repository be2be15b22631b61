// bk_api.hip -- host side of libbreakmer_hip.so: the C-ABI of include/breakmer_hip.h.
// gfx950 only; there is no CPU fallback anywhere in this library.
// This file: handle, bk_create / bk_destroy, launches (bk_run), bk_sync with its re-run logic, the result getters, the realign records'
// chaining (R2 host steps), bk_call.  bk_submit.hip.h (the submit path) and bk_index.hip.h (genome-wide seed look-up) are included below.
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
    std::vector<void *> old;      // outgrown allocations, freed by release(): hipFree waits for EVERY stream of the device -- a handle whose arena grows
                                  // (its first noisy batch) drained the kernels of every other handle in flight; the new block is allocated first and the
                                  // old one kept until the handle is trimmed or destroyed (round 6).  When memory is short the old blocks go first.
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, n ? n : 256);
        if (e != hipSuccess) {                      // not enough room beside what was outgrown: give that back (this waits for the device) and try again
            (void)hipGetLastError();
            for (void *o : old) (void)hipFree(o);
            old.clear();
            if (p) (void)hipFree(p);
            p = nullptr; bytes = 0;
            e = hipMalloc(&np, n ? n : 256);
            if (e != hipSuccess) return e;
        }
        if (p) old.push_back(p);
        p = np; bytes = n;
        return hipSuccess;
    }
    void release() { for (void *o : old) (void)hipFree(o); old.clear(); if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
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

#include "bk_submit.hip.h"      // bk_submit_regions[_ex]: packing, staging team, copies, descriptors

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
        // (round 6: the pointers end at -- an estimate of -- the batch's whole demand (bk_kmer.hip.h, bk_emit_contig), so the slack on top of them is
        //  small at scale: a quarter, but at most 4 GB + 3 %; a 768-region configs[4] batch asks for ~210 GB of the 288, and a quarter on top of
        //  that left no room for its records.  The result arena's demand is exact once a pass has run through: x4 only while it is not.)
        if (grow_arena) { h->arena_cap = std::max<uint64_t>(h->arena_cap < (4ull << 30) ? h->arena_cap * 4 : h->arena_cap + h->arena_cap / 8, tops[0] + std::min<uint64_t>(tops[0] / 4, (4ull << 30) + tops[0] / 32)); HIPCHK(h, h->d_arena.ensure(h->arena_cap)); }
        if (grow_out) { h->out_cap = std::max<uint64_t>(grow_arena ? h->out_cap * 4 : h->out_cap + h->out_cap / 8, tops[1] + std::min<uint64_t>(tops[1] / 4, (1ull << 30) + tops[1] / 16)); HIPCHK(h, h->d_out.ensure(h->out_cap)); }
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


#include "bk_index.hip.h"       // bk_index_*: genome-wide seed look-up (N4)
