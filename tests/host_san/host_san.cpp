// host_san.cpp -- drives the host-only code of the library under sanitizers (tests/test_host_san.py builds this file
// with g++ -fsanitize=address,undefined and, separately, -fsanitize=thread; no HIP, no GPU).
//
//   host_san files   <fixture.idx> <tmpdir>   v2 + native reader/writer on good, truncated, corrupted and bit-flipped files
//   host_san builder                          statistics, tail fit, Huber line, UpperLayers::build on worker threads
//   host_san threads                          UpperLayers::build on 8 threads + parallel_for's exception path (for TSAN)
//
// Every malformed input must end in a C++ exception (what the C ABI turns into an error code), never in a
// sanitizer report.  Exit code 0 = all good; the checks print what they covered.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "../../rabitq-ann-search_amd/csrc/search_coalescer.h"
#include "../../rabitq-ann-search_amd/csrc/builder_host.h"
#include "../../rabitq-ann-search_amd/csrc/host_index.h"
#include "../../rabitq-ann-search_amd/csrc/host_parallel.h"
#include "../../rabitq-ann-search_amd/csrc/native_file.h"

using namespace cph;

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const std::vector<uint8_t>& d, size_t len = (size_t)-1) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char*>(d.data()), (std::streamsize)std::min(len, d.size()));
}
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

struct Params { uint32_t D, bw, dim; };
static Params peek(const std::vector<uint8_t>& d) {
    Params p;
    std::memcpy(&p.D, &d[12], 4); std::memcpy(&p.bw, &d[20], 4); std::memcpy(&p.dim, &d[24], 4);
    return p;
}

// 0 = loaded, 1 = rejected with an exception
template <class F>
static int attempt(F&& f) {
    try { f(); return 0; } catch (const std::exception&) { return 1; }
}

static int run_files(const std::string& fixture, const std::string& tmp) {
    const std::vector<uint8_t> good = slurp(fixture);
    const Params pr = peek(good);
    // ---- v2: round trip, byte identical ----
    HostIndex hi;
    hi.load(fixture, pr.D, pr.bw, pr.dim);
    hi.save(tmp + "/copy.idx");
    REQUIRE(slurp(tmp + "/copy.idx") == good);
    hi.save(tmp + "/copy.idx");                              // over an existing file (atomic replace)
    REQUIRE(slurp(tmp + "/copy.idx") == good);
    int rejected = 0, loaded = 0;
    // ---- v2: every kind of truncation ----
    for (size_t len : {(size_t)0, (size_t)7, (size_t)12, (size_t)67, (size_t)68, (size_t)400, good.size() / 3, good.size() / 2,
                       good.size() - 5000, good.size() - 1}) {
        spit(tmp + "/t.idx", good, len);
        HostIndex t;
        REQUIRE(attempt([&] { t.load(tmp + "/t.idx", pr.D, pr.bw, pr.dim); }) == 1);
        ++rejected;
    }
    // ---- v2: hostile header fields ----
    auto patched = [&](size_t off, uint64_t val, size_t bytes) {
        std::vector<uint8_t> d = good;
        std::memcpy(&d[off], &val, bytes);
        spit(tmp + "/t.idx", d);
        HostIndex t;
        return attempt([&] { t.load(tmp + "/t.idx", pr.D, pr.bw, pr.dim); });
    };
    REQUIRE(patched(28, 0xFFFFFFFFFFFFull, 8) == 1);          // n far beyond the file
    REQUIRE(patched(28, hi.n + 1, 8) == 1);
    REQUIRE(patched(40, (uint32_t)hi.n + 7, 4) == 1);         // entry point out of range
    REQUIRE(patched(8, 3, 4) == 1);                           // version
    REQUIRE(patched(60, 43, 8) == 1);                         // rotation seed
    {   // neighbour count > 32 and id out of range in vertex 3
        const size_t base = 68 + 248 + 72 + pr.dim * 4 + hi.n * 8 + hi.n * hi.D * 4 + 3 * hi.RL.vertex_bytes + hi.RL.nb_off;
        REQUIRE(patched(base + hi.RL.count, 33, 4) == 1);
        REQUIRE(patched(base + hi.RL.ids, hi.n, 4) == 1);
        const size_t tail = 68 + 248 + 72 + pr.dim * 4 + hi.n * 8 + hi.n * hi.D * 4 + hi.n * hi.RL.vertex_bytes;
        REQUIRE(patched(tail, 0x7FFFFFFF, 4) == 1);           // number of upper layers
        if (!hi.upper.empty()) REQUIRE(patched(tail + 4, 0x7FFFFFFF, 4) == 1);   // size of the first one
    }
    // ---- v2: seeded bit flips anywhere in the file: load or throw, nothing else ----
    std::mt19937_64 rng(12345);
    for (int it = 0; it < 200; ++it) {
        std::vector<uint8_t> d = good;
        const int flips = 1 + (int)(rng() % 4);
        for (int k = 0; k < flips; ++k) {
            // half of the flips go to the first 400 bytes (header, calibration) and the tail (upper layers)
            size_t pos;
            const uint64_t r = rng();
            if (r % 4 == 0) pos = (size_t)(rng() % 400);
            else if (r % 4 == 1) pos = good.size() - 1 - (size_t)(rng() % std::min<size_t>(good.size(), 3000));
            else pos = (size_t)(rng() % good.size());
            d[pos] ^= (uint8_t)(1u << (rng() % 8));
        }
        spit(tmp + "/t.idx", d);
        HostIndex t;
        (attempt([&] { t.load(tmp + "/t.idx", pr.D, pr.bw, pr.dim); }) ? rejected : loaded)++;
    }
    // ---- native file: build the device blocks on the host, write, read back ----
    const DevLayout DL = make_dev_layout(pr.D, pr.bw);
    const size_t own_stride = hi.RL.nb_off;
    std::vector<uint8_t> blocks(hi.n * DL.stride), own(hi.n * own_stride);
    for (size_t v = 0; v < hi.n; ++v) {
        repack_ref_to_dev(hi.nb(v), hi.RL, DL, &blocks[v * DL.stride]);
        std::memcpy(&own[v * own_stride], &hi.search_data[v * hi.RL.vertex_bytes], own_stride);
        std::vector<uint8_t> back(hi.RL.nb_bytes);
        repack_dev_to_ref(&blocks[v * DL.stride], DL, hi.RL, back.data());
        // (slots >= count may hold stale ids in the file; everything the search reads must round-trip)
        uint32_t cnt;
        std::memcpy(&cnt, hi.nb(v) + hi.RL.count, 4);
        REQUIRE(std::memcmp(back.data() + hi.RL.codes, hi.nb(v) + hi.RL.codes, hi.RL.ids - hi.RL.codes) == 0);
        REQUIRE(std::memcmp(back.data() + hi.RL.ids, hi.nb(v) + hi.RL.ids, cnt * 4) == 0);
    }
    const std::string np = tmp + "/n.cphn";
    write_native(np, hi, DL.stride, own.data(), (uint32_t)own_stride, blocks.data());
    const std::vector<uint8_t> ngood = slurp(np);
    {
        HostIndex t;
        NativeMapping map;
        const NativeHeader nh = read_native(np, pr.D, pr.bw, pr.dim, t, map);
        REQUIRE(t.n == hi.n && t.entry == hi.entry && t.max_level == hi.max_level && t.levels == hi.levels && t.norm_sq == hi.norm_sq);
        REQUIRE(std::memcmp(t.calib, hi.calib, 248) == 0 && t.upper.size() == hi.upper.size());
        REQUIRE(std::memcmp(t.vec(0), hi.vec(0), hi.n * hi.D * 4) == 0);
        REQUIRE(std::memcmp(static_cast<const uint8_t*>(map.base) + nh.blocks_off, blocks.data(), blocks.size()) == 0);
        REQUIRE(t.has_dup_neighbors == hi.has_dup_neighbors);
        // saving over the file the mapping comes from: the mapping must stay readable (old inode), the new file complete
        write_native(np, t, DL.stride, static_cast<const uint8_t*>(map.base) + nh.own_off, (uint32_t)own_stride,
                     static_cast<const uint8_t*>(map.base) + nh.blocks_off);
        REQUIRE(slurp(np) == ngood);
        REQUIRE(std::memcmp(t.vec(0), hi.vec(0), hi.n * hi.D * 4) == 0);
    }
    auto native_attempt = [&](const std::vector<uint8_t>& d, size_t len = (size_t)-1) {
        spit(tmp + "/t.cphn", d, len);
        HostIndex t;
        NativeMapping map;
        return attempt([&] { read_native(tmp + "/t.cphn", pr.D, pr.bw, pr.dim, t, map); });
    };
    for (size_t len : {(size_t)0, (size_t)50, sizeof(NativeHeader), sizeof(NativeHeader) + 100, ngood.size() / 2, ngood.size() - 1}) {
        REQUIRE(native_attempt(ngood, len) == 1);
        ++rejected;
    }
    NativeHeader nh0;
    std::memcpy(&nh0, ngood.data(), sizeof(nh0));
    auto native_patched = [&](size_t off, uint64_t val, size_t bytes) {
        std::vector<uint8_t> d = ngood;
        std::memcpy(&d[off], &val, bytes);
        return native_attempt(d);
    };
    REQUIRE(native_patched(offsetof(NativeHeader, n), hi.n * 1000, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, n), 0xFFFFFFFFFFFFFFFFull, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, small_bytes), 0xFFFFFFFFFFFFull, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, small_bytes), 16, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, own_off), 8, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, raw_off), nh0.blocks_off + 4096, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, raw_off), 0xFFFFFFFFFFFFFF00ull, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, blocks_off), nh0.file_bytes, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, file_bytes), nh0.file_bytes + 4096, 8) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, stride), nh0.stride + 64, 4) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, own_stride), nh0.own_stride + 64, 4) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, n_layers), 0x7FFFFFFF, 4) == 1);
    REQUIRE(native_patched(offsetof(NativeHeader, entry), (uint32_t)hi.n, 4) == 1);
    REQUIRE(native_patched(nh0.blocks_off + 5 * (size_t)DL.stride + DL.count_off, 40, 4) == 1);      // count > 32
    REQUIRE(native_patched(nh0.blocks_off + 5 * (size_t)DL.stride + DL.ids_off, hi.n + 3, 4) == 1);  // id out of range
    {   // has_dup is recomputed, not believed
        std::vector<uint8_t> d = ngood;
        uint32_t id1;
        std::memcpy(&id1, &d[nh0.blocks_off + DL.ids_off + 4], 4);
        std::memcpy(&d[nh0.blocks_off + DL.ids_off], &id1, 4);        // vertex 0: slot 0 repeats slot 1
        spit(tmp + "/t.cphn", d);
        HostIndex t;
        NativeMapping map;
        read_native(tmp + "/t.cphn", pr.D, pr.bw, pr.dim, t, map);
        REQUIRE(t.has_dup_neighbors);
    }
    for (int it = 0; it < 200; ++it) {
        std::vector<uint8_t> d = ngood;
        const int flips = 1 + (int)(rng() % 4);
        for (int k = 0; k < flips; ++k) {
            const uint64_t r = rng();
            const size_t pos = (r % 2 == 0) ? (size_t)(rng() % (sizeof(NativeHeader) + 400)) : (size_t)(rng() % ngood.size());
            d[pos] ^= (uint8_t)(1u << (rng() % 8));
        }
        (native_attempt(d) ? rejected : loaded)++;
    }
    // ---- the query encoder's host mirror ----
    for (size_t D : {(size_t)16, (size_t)128, (size_t)1024}) {
        Rotation rot;
        rot.init(D, 42);
        std::vector<float> q(D);
        for (auto& x : q) x = (float)((double)(rng() % 2001) / 1000.0 - 1.0);
        EncodedQuery eq;
        encode_query(rot, q.data(), eq);
        std::vector<uint32_t> masks((D >= 32 ? D / 32 : 1) * 4);
        std::vector<uint8_t> lut(D / 4 * 16), qu(D);
        qu_to_masks(eq.qu.data(), D, masks.data());
        qu_to_lut(eq.qu.data(), D, lut.data());
        lut_to_qu(lut.data(), D, qu.data());
        REQUIRE(qu == eq.qu);
    }
    std::printf("files: ok (%d malformed inputs rejected, %d bit-flipped inputs still loadable)\n", rejected, loaded);
    return 0;
}

// a small clustered point set with geometric levels, as build_graph hands it to UpperLayers
struct UpperCase {
    std::vector<float> x;
    std::vector<int32_t> levels;
    size_t n, dim;
    int max_level = 0;
    uint32_t entry = 0;
};
static UpperCase make_upper_case(size_t n, size_t dim, uint64_t seed) {
    UpperCase c;
    c.n = n; c.dim = dim;
    std::mt19937_64 rng(seed);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    c.x.resize(n * dim);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < dim; ++j) c.x[i * dim + j] = nd(rng) + (float)(i % 7) * 2.0f;
    c.levels.assign(n, 0);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (size_t i = 0; i < n; ++i) {
        int l = 0;
        while (u(rng) < 0.25 && l < 4) ++l;
        c.levels[i] = l;
        if (l > c.max_level) { c.max_level = l; c.entry = (uint32_t)i; }
    }
    return c;
}
static void check_upper(const UpperCase& c, build::UpperLayers& ul) {
    std::vector<uint32_t> renumber(c.n);
    for (size_t i = 0; i < c.n; ++i) renumber[i] = (uint32_t)i;
    const auto layers = ul.export_layers(renumber);
    REQUIRE((int)layers.size() == c.max_level);
    size_t edges = 0;
    for (int l = 1; l <= c.max_level; ++l)
        for (const auto& e : layers[l - 1]) {
            REQUIRE(c.levels[e.node] >= l && e.nbrs.size() <= ul.M);
            for (uint32_t w : e.nbrs) { REQUIRE(w < c.n && c.levels[w] >= l && w != e.node); ++edges; }
        }
    REQUIRE(edges > 0);
}

static int run_builder() {
    using namespace build;
    std::mt19937_64 rng(7);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    // statistics on ordinary, tiny and degenerate samples
    for (size_t n : {(size_t)1, (size_t)2, (size_t)3, (size_t)50, (size_t)4000}) {
        std::vector<float> v(n);
        for (auto& x : v) x = nd(rng);
        const float med = median_of(v);
        (void)mad_sigma(v, med);
        std::vector<float> s = v;
        std::sort(s.begin(), s.end());
        (void)quantile_sorted(s, 99, 100);
        (void)quantile_sorted(s, 1, 1);
    }
    REQUIRE(median_of({}) == 0.0f);
    // tail fits: exponential, heavy, constant, too short
    for (int kind = 0; kind < 4; ++kind)
        for (size_t n : {(size_t)10, (size_t)300, (size_t)20000}) {
            std::vector<float> r(n);
            std::exponential_distribution<float> ex(1.0f);
            for (auto& x : r) x = kind == 0 ? ex(rng) : kind == 1 ? std::pow(ex(rng), 3.0f) : kind == 2 ? 1.0f : std::fabs(nd(rng));
            std::sort(r.begin(), r.end());
            for (size_t min_tail : {(size_t)5, (size_t)40}) {
                const TailModel a = fit_tail(r, min_tail, 0.90f, 0.99f);
                const TailModel b = fit_tail_at(r, 0.95f, min_tail);
                const TailModel c = fit_tail(r, min_tail, 0.95f, 0.95f);
                for (const TailModel& m : {a, b, c})
                    for (float alpha : {0.5f, 0.05f, 1e-3f, 1e-6f, 0.0f}) { const float q = tail_quantile(alpha, m); REQUIRE(q == q); }
            }
        }
    {
        double xi, beta;
        std::vector<double> y = {1.0};
        (void)gpd_mle(y, xi, beta);
        y.assign(100, 0.0);
        REQUIRE(!gpd_mle(y, xi, beta));
    }
    // Huber line: clean, with outliers, constant x
    for (int kind = 0; kind < 3; ++kind) {
        std::vector<float> x(500), y(500);
        for (size_t i = 0; i < x.size(); ++i) {
            x[i] = kind == 2 ? 1.0f : nd(rng);
            y[i] = 0.9f * x[i] + 0.05f + 0.01f * nd(rng) + ((kind == 1 && i % 17 == 0) ? 30.0f : 0.0f);
        }
        double a, b;
        robust_line(x, y, a, b);
        REQUIRE(a == a && b == b);
        if (kind < 2) REQUIRE(std::fabs(a - 0.9) < 0.05);
    }
    // upper layers: sequential seed only (few members), and with worker threads
    for (size_t n : {(size_t)300, (size_t)6000}) {
        UpperCase c = make_upper_case(n, 24, 11 + n);
        UpperLayers ul(c.x.data(), c.dim, c.n, c.levels, c.max_level, c.entry, 18, 32);
        ul.build();
        check_upper(c, ul);
    }
    std::printf("builder: ok\n");
    return 0;
}

static int run_threads_mode() {
    setenv("CPH_BUILD_THREADS", "8", 1);
    UpperCase c = make_upper_case(12000, 16, 99);
    build::UpperLayers ul(c.x.data(), c.dim, c.n, c.levels, c.max_level, c.entry, 18, 32);
    ul.build();
    check_upper(c, ul);
    // parallel_for: disjoint writes, then a worker that throws (the exception must arrive here, once, after the join)
    std::vector<uint32_t> out(100000, 0);
    parallel_for(out.size(), 64, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) out[i] = (uint32_t)i * 3u; });
    for (size_t i = 0; i < out.size(); ++i) REQUIRE(out[i] == (uint32_t)i * 3u);
    bool caught = false;
    try {
        parallel_for(out.size(), 64, [&](size_t lo, size_t) { if (lo >= 5000) throw std::runtime_error("worker failed"); });
    } catch (const std::runtime_error& e) {
        caught = std::string(e.what()) == "worker failed";
    }
    REQUIRE(caught);
    std::printf("threads: ok\n");
    return 0;
}

// ---- the coalescing policy of concurrent cph_search callers (csrc/search_coalescer.h) with a stand-in launch -----
// T threads x calls.  The stand-in "launch" gives every member of its group a completion time of its own (so that callers
// pile up and members finish at different times, as queries of different length do) and fails on purpose for k == 7;
// the stand-in "wait" sleeps until the member's time and answers from the member's own query value.  Every caller
// must be answered exactly once, with ITS answer, errors must reach every member of the failing group, a group never
// mixes k, never exceeds the cap, no more launches than slots are in flight, and a slot is not reused before the last
// member of its launch has its answer.
static int run_coalescer_mode() {
    using namespace cph;
    SearchCoalescer co;
    co.n_slots = 3;
    co.gather_us = 100;
    std::atomic<int> launches{0}, max_group{0}, answered{0}, mixed{0}, overlap{0};
    std::atomic<int> slot_members[kLeaderSlots];
    for (auto& x : slot_members) x = 0;
    struct Flight { std::chrono::steady_clock::time_point ready[kLeaderGroup]; };
    Flight flights[kLeaderSlots];
    auto launch = [&](int slot, const std::vector<SearchReq*>& g) {
        ++launches;
        int pg = max_group.load();
        while ((int)g.size() > pg && !max_group.compare_exchange_weak(pg, (int)g.size())) {}
        REQUIRE(slot >= 0 && slot < co.n_slots);
        REQUIRE(!g.empty() && g.size() <= kLeaderGroup);
        for (SearchReq* r : g) if (r->k != g[0]->k) ++mixed;
        if (slot_members[slot].exchange((int)g.size()) != 0) ++overlap;      // the previous launch of this slot still had waiters
        std::this_thread::sleep_for(std::chrono::microseconds(40));          // "enqueue"
        if (g[0]->k == 7) { slot_members[slot] = 0; throw std::runtime_error("stand-in launch failed"); }
        const auto now = std::chrono::steady_clock::now();
        for (size_t i = 0; i < g.size(); ++i)
            flights[slot].ready[i] = now + std::chrono::microseconds(150 + 37 * ((size_t)(g[i]->query[0]) % 7));
    };
    auto wait = [&](int slot, uint32_t index, SearchReq& r) {
        std::this_thread::sleep_until(flights[slot].ready[index]);
        r.ids[0] = (int64_t)(r.query[0] * 2.0f);      // "the answer" = a function of the caller's own query
        r.dist[0] = r.query[0] + (float)r.k;
        *r.m = 1;
        --slot_members[slot];
    };
    const int T = 12, calls = 150;
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        for (int c = 0; c < calls; ++c) {
            float q = (float)(t * 1000 + c);
            int64_t id = -1;
            float d = -1.0f;
            uint64_t m = 0;
            SearchReq r;
            r.query = &q; r.k = (uint64_t)((t % 4 == 3) ? 7 : (t % 3 == 0 ? 1 : 10)); r.ids = &id; r.dist = &d; r.m = &m;
            co.submit(r, launch, wait);
            ++answered;
            if (r.k == 7) { if (r.rc != 2 || r.err != "stand-in launch failed") ++bad; }
            else if (r.rc != 0 || m != 1 || id != (int64_t)(q * 2.0f) || d != q + (float)r.k) ++bad;
        }
    });
    for (auto& x : th) x.join();
    REQUIRE(answered.load() == T * calls);
    REQUIRE(bad.load() == 0);
    REQUIRE(mixed.load() == 0);
    REQUIRE(overlap.load() == 0);
    REQUIRE(launches.load() < T * calls);          // callers WERE gathered
    REQUIRE(max_group.load() > 1);
    REQUIRE(co.waiting.empty() && co.gathering == 0);
    for (int i = 0; i < co.n_slots; ++i) REQUIRE(!co.slots[i].busy && co.slots[i].pending == 0);
    // a lone caller is its own leader and never waits for the window
    {
        float q = 5.0f; int64_t id = 0; float d = 0; uint64_t m = 0;
        SearchReq r; r.query = &q; r.k = 10; r.ids = &id; r.dist = &d; r.m = &m;
        for (auto& sl : co.slots) sl.last_group = 1;
        const auto t0 = std::chrono::steady_clock::now();
        co.gather_us = 200000;
        co.submit(r, launch, wait);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        REQUIRE(r.rc == 0 && id == 10 && ms < 100.0);
    }
    // an error in one member's wait is that member's alone
    {
        float q1 = 1.0f, q2 = 2.0f; int64_t id1 = 0, id2 = 0; float d1 = 0, d2 = 0; uint64_t m1 = 0, m2 = 0;
        SearchReq a, b;
        a.query = &q1; a.k = 10; a.ids = &id1; a.dist = &d1; a.m = &m1;
        b.query = &q2; b.k = 10; b.ids = &id2; b.dist = &d2; b.m = &m2;
        co.gather_us = 0;
        auto wait2 = [&](int slot, uint32_t index, SearchReq& r) {
            if (r.query[0] == 1.0f) { --slot_members[slot]; throw std::runtime_error("this member only"); }
            wait(slot, index, r);
        };
        std::thread t1([&] { co.submit(a, launch, wait2); });
        std::thread t2([&] { co.submit(b, launch, wait2); });
        t1.join(); t2.join();
        REQUIRE(a.rc == 2 && a.err == "this member only" && b.rc == 0 && id2 == 4);
        for (int i = 0; i < co.n_slots; ++i) REQUIRE(!co.slots[i].busy && co.slots[i].pending == 0);
    }
    std::printf("coalescer: ok (%d callers in %d launches, largest group %d)\n", T * calls, launches.load(), max_group.load());
    return 0;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "coalescer") return run_coalescer_mode();
    if (mode == "files" && argc == 4) return run_files(argv[2], argv[3]);
    if (mode == "builder") return run_builder();
    if (mode == "threads") return run_threads_mode();
    std::fprintf(stderr, "usage: host_san files <fixture.idx> <tmpdir> | builder | threads\n");
    return 2;
}
