// TEST INFRASTRUCTURE: drives the host half of the library (bk_api.hip built --cuda-host-only against tests/sanitize/hip_stub.cpp)
// through its C-ABI under a sanitizer: packing (SIMD and table path), synchronous and asynchronous submits on reused staging
// buffers, two handles from two threads, the getters on an empty result arena, the call tail (bk_call on a batch, bk_call_text
// on the G5/G8m fixture texts the Python test writes), bk_trim, error paths.  Prints the rows of bk_call_text for the Python
// side to compare with the reference's rows; exit code 0 = every call behaved.
//   host_driver <dir>      <dir>/ctx_<n>.txt: call context of an n-region batch; <dir>/case_<i>.txt: one described contig
#include "../../include/breakmer_hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_driver: CHECK failed at line %d: %s\n", __LINE__, #c); exit(2); } } while (0)

static std::string slurp(const std::string &fn) { std::ifstream f(fn); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }

struct Batch {
    std::vector<std::vector<char>> reads; std::vector<std::vector<uint16_t>> lens; std::vector<std::vector<uint8_t>> io;
    std::vector<std::string> win; std::vector<std::vector<std::string>> partners; std::vector<std::vector<const char *>> pptr; std::vector<std::vector<int32_t>> plen;
    std::vector<std::vector<char>> sc; std::vector<std::vector<uint16_t>> sclen;
    std::vector<bk_region> g;
};
static void make_batch(Batch &b, int n, int nreads, int L, unsigned seed, bool codes)
{
    std::mt19937 rnd(seed);
    b.reads.resize(n); b.lens.resize(n); b.io.resize(n); b.win.resize(n); b.partners.resize(n); b.pptr.resize(n); b.plen.resize(n); b.sc.resize(n); b.sclen.resize(n); b.g.assign(n, bk_region{});
    for (int r = 0; r < n; r++) {
        const int nr = r == 1 ? 0 : nreads + (int)(rnd() % 7);                 // one region without reads
        b.reads[r].assign((size_t)nr * L + 16, codes ? 0 : 'A'); b.lens[r].resize(nr + 1); b.io[r].resize(nr + 1);
        for (int i = 0; i < nr; i++) {
            const int len = L - (int)(rnd() % 40);                             // ragged
            b.lens[r][i] = (uint16_t)len; b.io[r][i] = rnd() % 5 == 0;
            for (int t = 0; t < len; t++) { const unsigned c = rnd() % 4; b.reads[r][(size_t)i * L + t] = codes ? (char)c : "ACGT"[c]; }
            if (rnd() % 9 == 0) b.reads[r][(size_t)i * L + rnd() % len] = codes ? 4 : 'N';       // N calls take the table path
        }
        b.win[r].resize(600 + rnd() % 200); for (auto &ch : b.win[r]) ch = "ACGT"[rnd() % 4];
        if (r % 3 == 0) { b.partners[r].push_back(std::string(300, 'A')); for (auto &ch : b.partners[r][0]) ch = "ACGT"[rnd() % 4]; }
        for (auto &p : b.partners[r]) { b.pptr[r].push_back(p.c_str()); b.plen[r].push_back((int32_t)p.size()); }
        const int nsc = r % 2 ? 3 : -1;
        if (nsc > 0) { b.sc[r].assign((size_t)nsc * 64, 'C'); b.sclen[r].assign(nsc, 50); for (auto &ch : b.sc[r]) ch = "ACGT"[rnd() % 4]; }
        bk_region &g = b.g[r];
        g.reads = b.reads[r].data(); g.read_lens = b.lens[r].data(); g.indel_only = b.io[r].data(); g.n_reads = nr; g.read_stride = L;
        g.sc_seqs = nsc > 0 ? b.sc[r].data() : nullptr; g.sc_lens = nsc > 0 ? b.sclen[r].data() : nullptr; g.n_sc = nsc; g.sc_stride = 64;
        g.window = b.win[r].c_str(); g.window_len = (int32_t)b.win[r].size();
        g.n_partners = (int32_t)b.partners[r].size(); g.partners = b.pptr[r].empty() ? nullptr : b.pptr[r].data(); g.partner_lens = b.plen[r].empty() ? nullptr : b.plen[r].data();
    }
}

// the same context with the annotation tables replaced by "keep_tables" (what the driver sends from the second batch of a run on)
static std::string keep_variant(const std::string &ctx)
{
    std::string out; size_t pos = 0; bool said = false;
    while (pos < ctx.size()) {
        size_t nl = ctx.find('\n', pos); if (nl == std::string::npos) nl = ctx.size();
        const std::string ln = ctx.substr(pos, nl - pos); pos = nl + 1;
        if (ln.rfind("gene ", 0) == 0 || ln.rfind("arep", 0) == 0) continue;
        out += ln; out += '\n';
        if (!said && ln.rfind("opts ", 0) == 0) { out += "keep_tables\n"; said = true; }
    }
    return out;
}

static void one_handle(const std::string &dir, unsigned seed)
{
    bk_config cfg{}; cfg.abi_version = BK_ABI_VERSION; cfg.kmer_size = 31; cfg.rc_thresh = 2;
    bk_handle *h = nullptr;
    CHECK(bk_create(0, &cfg, &h) == BK_OK);
    const std::string ctx8 = slurp(dir + "/ctx_8.txt"), ctx5 = slurp(dir + "/ctx_5.txt");
    CHECK(!ctx8.empty() && !ctx5.empty());
    for (int round = 0; round < 6; round++) {
        const int n = round % 2 ? 5 : 8;
        Batch b; make_batch(b, n, round == 3 ? 900 : 120, round == 3 ? 250 : 150, seed + round, round % 3 == 2);
        const uint32_t flags = (round % 3 == 2 ? BK_SUBMIT_READ_CODES : 0u) | (round >= 2 ? BK_SUBMIT_ASYNC : 0u);
        CHECK(bk_submit_regions_ex(h, b.g.data(), n, flags) == BK_OK);
        CHECK(bk_run(h, BK_STAGE_ALL) == BK_OK);                       // joins the asynchronous submit
        CHECK(bk_sync(h) == BK_OK);
        for (int r = 0; r < n; r++) {
            int32_t st = -1, nm = -1, nu = -1, nc = -1; const char *tx = nullptr;
            CHECK(bk_get_region_status(h, r, &st, &tx) == BK_OK && st == 0 && tx);
            CHECK(bk_get_kmer_count(h, r, &nm, &nu) == BK_OK && nm == 0);
            CHECK(bk_get_contig_count(h, r, &nc) == BK_OK && nc == 0);
            bk_contig_info info; CHECK(bk_get_contig_info(h, r, 0, &info) != BK_OK);       // no such contig: an error, not a crash
        }
        if (round == 0) CHECK(bk_set_call_context(h, keep_variant(ctx8).c_str()) == BK_E_STATE);     // keep_tables before the handle has seen any tables
        CHECK(bk_set_call_context(h, (n == 8 ? ctx8 : ctx5).c_str()) == BK_OK);
        CHECK(bk_set_call_context(h, (n == 8 ? ctx5 : ctx8).c_str()) != BK_OK);            // region count mismatch is refused
        if (round >= 1) CHECK(bk_set_call_context(h, keep_variant(n == 8 ? ctx8 : ctx5).c_str()) == BK_OK);      // the tables of the context before stay
        CHECK(bk_set_call_context(h, (n == 8 ? ctx8 : ctx5).c_str()) == BK_OK);
        CHECK(bk_fetch(h) == BK_OK);
        CHECK(bk_call(h) == BK_OK);
        size_t need = 0; CHECK(bk_get_calls(h, nullptr, 0, &need) == BK_OK && need == 1);
        uint64_t v = 0; CHECK(bk_get_stat(h, 22, &v) == BK_OK && v == 0);
        float ms = -1; CHECK(bk_last_kernel_ms(h, 0, &ms) == BK_OK);
        if (round == 3) CHECK(bk_trim(h, 1 << 20) == BK_OK);                                // give the big buffers back, go on with small batches
        if (round == 4) {                                                                    // a bad batch leaves the handle without one
            Batch bad; make_batch(bad, 2, 10, 150, seed + 99, false); bad.reads[0][3] = 'x';
            CHECK(bk_submit_regions(h, bad.g.data(), 2) == BK_E_ARG);
            CHECK(bk_run(h, BK_STAGE_ALL) == BK_E_STATE);
            CHECK(bk_submit_regions_ex(h, bad.g.data(), 2, BK_SUBMIT_ASYNC) == BK_OK);        // asynchronous: the error surfaces at the next call
            CHECK(bk_run(h, BK_STAGE_ALL) == BK_E_ARG);
        }
    }
    CHECK(bk_destroy(h) == BK_OK);
}

// The submit path at the size where it works in chunks (>= 16 regions: the reads go to the device chunk by chunk while the helper
// threads fill the rest), with ASCII reads and with 2-bit packed rows (BK_SUBMIT_PACKED), waited for and asynchronous; and what it
// says when a window, a partner window or an N list is bad: the FIRST offending region, by name of the fault.
static void submit_paths(unsigned seed)
{
    bk_config cfg{}; cfg.abi_version = BK_ABI_VERSION; cfg.kmer_size = 31; cfg.rc_thresh = 2;
    bk_handle *h = nullptr;
    CHECK(bk_create(0, &cfg, &h) == BK_OK);
    const int n = 44, L = 150;
    Batch b; make_batch(b, n, 200, L, seed, false);
    // packed copy of the same batch: rows of W words, N calls listed (row << 10 | position, ascending)
    const int W = (L + 15) / 16;
    std::vector<std::vector<uint32_t>> rows(n), nls(n); std::vector<bk_region> pg = b.g;
    for (int r = 0; r < n; r++) {
        const int nr = b.g[r].n_reads;
        rows[r].assign((size_t)std::max(nr, 1) * W, 0);
        for (int i = 0; i < nr; i++) {
            std::vector<uint32_t> w(W + 1), np(L + 1); int32_t nn = 0;
            CHECK(bk_pack_sequence(b.reads[r].data() + (size_t)i * L, b.lens[r][i], 0, w.data(), (int32_t)w.size(), np.data(), L, &nn) == BK_OK);
            memcpy(rows[r].data() + (size_t)i * W, w.data(), (size_t)W * 4);
            for (int e = 0; e < nn; e++) nls[r].push_back(((uint32_t)i << 10) | np[e]);
        }
        pg[r].reads = (const char *)rows[r].data(); pg[r].read_stride = W * 4; pg[r].read_n = nls[r].empty() ? nullptr : nls[r].data(); pg[r].n_read_n = (int32_t)nls[r].size();
        pg[r].sc_seqs = nullptr; pg[r].sc_lens = nullptr; pg[r].n_sc = -1;
    }
    for (int round = 0; round < 4; round++) {
        const bool packed = round & 1, async = round >= 2;
        CHECK(bk_submit_regions_ex(h, packed ? pg.data() : b.g.data(), n, (packed ? BK_SUBMIT_PACKED : 0u) | (async ? BK_SUBMIT_ASYNC : 0u)) == BK_OK);
        CHECK(bk_run(h, BK_STAGE_ALL) == BK_OK);
        CHECK(bk_sync(h) == BK_OK);
        CHECK(bk_fetch(h) == BK_OK);
    }
    auto says = [&](const char *a, const char *c) { const char *e = bk_last_error(h); return e && strstr(e, a) && strstr(e, c); };
    {   // a foreign character in the windows of regions 30 and 9: region 9 is named
        std::string w30 = b.win[30], w9 = b.win[9]; w30[5] = 'R'; w9[100] = '-';
        std::vector<bk_region> g = b.g; g[30].window = w30.c_str(); g[9].window = w9.c_str();
        CHECK(bk_submit_regions(h, g.data(), n) == BK_E_ARG && says("region 9:", "in the reference window"));
        CHECK(bk_run(h, BK_STAGE_ALL) == BK_E_STATE);
        CHECK(bk_submit_regions_ex(h, g.data(), n, BK_SUBMIT_ASYNC) == BK_OK);
        CHECK(bk_run(h, BK_STAGE_ALL) == BK_E_ARG && says("region 9:", "in the reference window"));
        // ... and in a partner window (regions 0, 3, 6, ... carry one); a window fault anywhere is reported before a read fault
        std::string p12 = b.partners[12][0]; p12[7] = 'X'; const char *pp[1] = {p12.c_str()};
        g = b.g; g[12].partners = pp; std::vector<char> rd = b.reads[2]; rd[3] = 'x'; g[2].reads = rd.data();
        CHECK(bk_submit_regions(h, g.data(), n) == BK_E_ARG && says("region 12:", "in a partner window"));
        g[12].partners = b.g[12].partners;
        CHECK(bk_submit_regions(h, g.data(), n) == BK_E_ARG && says("region 2 read 0", "base other than"));
    }
    {   // packed rows: an N list that is not ascending, one that points behind a read
        std::vector<bk_region> g = pg; std::vector<uint32_t> bad = {(5u << 10) | 7u, (5u << 10) | 7u};
        g[20].read_n = bad.data(); g[20].n_read_n = 2;
        CHECK(bk_submit_regions_ex(h, g.data(), n, BK_SUBMIT_PACKED) == BK_E_ARG && says("region 20 read 5", "N list"));
        bad = {(3u << 10) | 1000u}; g[20].read_n = bad.data(); g[20].n_read_n = 1;
        CHECK(bk_submit_regions_ex(h, g.data(), n, BK_SUBMIT_PACKED) == BK_E_ARG && says("region 20 read 3", "N list"));
        CHECK(bk_submit_regions_ex(h, pg.data(), n, BK_SUBMIT_PACKED | BK_SUBMIT_READ_CODES) == BK_E_ARG);
    }
    CHECK(bk_submit_regions_ex(h, pg.data(), n, BK_SUBMIT_PACKED) == BK_OK);                 // and the handle takes a good batch again
    CHECK(bk_run(h, BK_STAGE_ALL) == BK_OK && bk_sync(h) == BK_OK);
    CHECK(bk_destroy(h) == BK_OK);
}

int main(int argc, char **argv)
{
    CHECK(argc == 2);
    const std::string dir = argv[1];
    CHECK(bk_abi_version() == BK_ABI_VERSION);
    {   // 2-bit packing: SSSE3 path == table path (an N forces the table path for its 16-base block)
        std::mt19937 rnd(7);
        for (int t = 0; t < 200; t++) {
            const int len = 1 + (int)(rnd() % 300);
            std::string s(len, 'A'); for (auto &ch : s) ch = "ACGT"[rnd() % 4];
            std::vector<uint32_t> w((len + 15) / 16 + 1), w2(w.size()), np(len + 1); int32_t nn = -1;
            CHECK(bk_pack_sequence(s.c_str(), len, 0, w.data(), (int32_t)w.size(), np.data(), len, &nn) == BK_OK && nn == 0);
            for (int i = 0; i < len; i++) CHECK(((w[i >> 4] >> (30 - 2 * (i & 15))) & 3u) == (unsigned)(strchr("ACGT", s[i]) - "ACGT"));
            std::string s2 = s; const int at = (int)(rnd() % len); s2[at] = 'N';
            CHECK(bk_pack_sequence(s2.c_str(), len, 0, w2.data(), (int32_t)w2.size(), np.data(), len, &nn) == BK_OK && nn == 1 && (int)np[0] == at);
            s2[at] = 'n'; CHECK(bk_pack_sequence(s2.c_str(), len, 0, w2.data(), (int32_t)w2.size(), np.data(), len, &nn) == BK_E_ARG);
        }
    }
    {   // the call tail on fully described contigs (G5 / G8m texts written by the Python test): rows go to stdout
        for (int i = 0;; i++) {
            const std::string text = slurp(dir + "/case_" + std::to_string(i) + ".txt");
            if (text.empty()) { CHECK(i > 0); break; }
            std::vector<char> out(1 << 16); int hit = -2;
            CHECK(bk_call_text(text.c_str(), out.data(), out.size(), &hit) == BK_OK);
            printf("ROW\t%d\t%d\t%s\n", i, hit, out.data());
        }
        char small[4]; int hit; CHECK(bk_call_text("garbage", small, sizeof(small), &hit) != BK_OK);
    }
    one_handle(dir, 100);
    submit_paths(400);
    std::thread t1(one_handle, dir, 200u), t2(one_handle, dir, 300u);        // handles are independent: two threads, one handle each
    t1.join(); t2.join();
    printf("DONE\n");
    return 0;
}
