// builder.h — index construction: what build() + finalize() of the drop-in do.
//
// Designed for the GPU, not translated from the reference's host pipeline.  The vectors go to HBM once
// and everything that touches n x D data runs there (kernels in device_knn.h / device_build.h):
//
//   1. exact 32-NN lists of all vertices on the matrix cores          (reference: NNDescent on the host,
//      graph/graph_refinement.hpp:71-263, 455-515)
//   2. reverse-edge CSR + neighbour selection, one wave per vertex   (:386-429, neighbor_selection.hpp:21-88)
//   3. hub + BFS renumbering on the host (a queue walk over n x 32 ids) (graph/rabitq_graph.hpp:208-340)
//   4. rows gathered into their final order; per-edge codes encoded straight into the device block layout
//      the search kernel reads; the vertices' own codes for the file format
//      (encoder/rabitq_encoder.hpp; bit-identical to the reference's encoder)
//   5. upper layers (the ~5 % of the vertices with level >= 1): incremental insertion on the host, many
//      vertices at a time under per-vertex locks (reference: one vertex at a time, api/hnsw_index.hpp:505-716).
//      Insertion order is what makes these layers navigable -- early vertices keep long links -- which a
//      batch k-NN construction on the GPU does not reproduce (measured: recall of the graph-quality test
//      0.68 with per-level exact k-NN + selection, 0.72 with insertion, reference 0.71)
//   6. calibration: sample evaluation on the GPU (calib_kernel), robust statistics, affine fit and
//      extreme-value tail fit on the host                              (api/hnsw_index.hpp:718-1139, core/evt_crc.hpp)
//
// The device arrays produced in step 4 ARE the searchable index: they are handed to the handle as they
// are; the reference-layout image needed by save() is derived from them.
//
// Parity with the reference's builder is statistical (its own output depends on its OpenMP thread count,
// SURVEY F6), except the edge encoder, which is bit-exact.  An index built here loads in the compiled
// reference, and the reference's CPU search on it equals our GPU search bit for bit (tests/test_gpu_builder.py).
//
// Finding F9: the reference cannot calibrate any index with n > ~230k -- its tail fit demands sqrt(n)
// exceedances but its sample only ever yields sqrt(480 sqrt(n)) of them (hnsw_index.hpp:1046-1056,
// adaptive_defaults.hpp:45-46) and it throws "EVT-CRC fit did not converge".  Here the demanded tail size
// is capped at half of what the residual sample can supply.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>
#include <random>
#include <stdexcept>
#include <thread>
#include <vector>

#include "cph_core.h"
#include "device_buf.h"
#include "device_build.h"
#include "device_knn.h"
#include "host_index.h"

namespace cph {
namespace build {

inline void parallel_for(size_t n, size_t min_chunk, const std::function<void(size_t, size_t)>& fn) {
    // at most 64 host threads per process (one process per GPU, eight per node)
    unsigned hw = std::min(64u, std::thread::hardware_concurrency());
    size_t nt = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n / std::max<size_t>(min_chunk, 1)));
    if (const char* e = getenv("CPH_BUILD_THREADS")) nt = std::max(1, atoi(e));
    if (nt <= 1) { fn(0, n); return; }
    std::atomic<size_t> next{0};
    const size_t chunk = std::max<size_t>(min_chunk, n / (nt * 16));
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (;;) {
                const size_t lo = next.fetch_add(chunk);
                if (lo >= n) break;
                fn(lo, std::min(n, lo + chunk));
            }
        });
    for (auto& x : th) x.join();
}

struct StageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[build] %-34s %.2f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

inline uint32_t grid_for(uint64_t items, uint32_t block) { return (uint32_t)((items + block - 1) / block); }

// ---- small robust statistics ---------------------------------------------------------------------
inline float median_of(std::vector<float> v) {      // upper median, as nth element of the sorted sample
    if (v.empty()) return 0.0f;
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v[v.size() / 2];
}
inline float mad_sigma(const std::vector<float>& v, float med) {   // 1.4826 * median absolute deviation
    std::vector<float> a(v.size());
    for (size_t i = 0; i < v.size(); ++i) a[i] = std::fabs(v[i] - med);
    return 1.4826f * median_of(std::move(a));
}
inline float quantile_sorted(const std::vector<float>& s, size_t num, size_t den) { return s[std::min(s.size() - 1, s.size() * num / den)]; }

// ---- exact kNN on the matrix cores (device pointers; D a multiple of 32) ----------------------------
inline void knn_device(const float* d_q, const float* d_qnorm, size_t nq, const float* d_b, const float* d_bnorm,
                       size_t nb, size_t D, bool exclude_self, int num_cus, uint32_t* d_ids, float* d_dist) {
    if (D % kKnnKC != 0 || nq == 0 || nb == 0) throw std::invalid_argument("knn_device: D must be a multiple of 32");
    // slices of row blocks, so that no single launch runs for minutes
    const uint32_t rows_per_launch = (uint32_t)num_cus * 8u * kKnnTile;
    for (size_t rb = 0; rb < nq; rb += rows_per_launch) {
        KnnArgs a{d_q, d_b, d_bnorm, d_qnorm, (uint32_t)nq, (uint32_t)nb, (uint32_t)D, (uint32_t)rb,
                  (uint32_t)std::min<size_t>(nq, rb + rows_per_launch), exclude_self ? 1u : 0u, d_ids, d_dist};
        const uint32_t grid = (a.row_end - a.row_begin + kKnnTile - 1) / kKnnTile;
        hipLaunchKernelGGL(knn_mfma_kernel, dim3(grid), dim3(256), 0, nullptr, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
}

// rows of `src` (host, `width` floats each) into a zeroed device image with `ld` floats per row
inline void upload_padded(float* dst, size_t ld, const float* src, size_t width, size_t rows) {
    if (ld == width) { HIP_CHECK(hipMemcpy(dst, src, rows * width * 4, hipMemcpyHostToDevice)); return; }
    HIP_CHECK(hipMemset(dst, 0, rows * ld * 4));
    HIP_CHECK(hipMemcpy2D(dst, ld * 4, src, width * 4, width * 4, rows, hipMemcpyHostToDevice));
}

// Host-pointer convenience for the C-ABI hook: x[n][D] against itself (self excluded) or q[nq][D] against it.
inline void gpu_knn(const float* q, const float* qnorm, size_t nq, const float* x, const float* norm_sq, size_t n,
                    size_t D, int num_cus, uint32_t* out_ids, float* out_dist) {
    const bool self = (q == nullptr);
    const size_t Dk = (D + kKnnKC - 1) / kKnnKC * kKnnKC;      // D = 16 -> 32: zero columns
    DevBuf<float> d_x(n * Dk), d_norm(n), d_od((self ? n : nq) * kKnnK);
    DevBuf<uint32_t> d_oi((self ? n : nq) * kKnnK);
    upload_padded(d_x.p, Dk, x, D, n);
    HIP_CHECK(hipMemcpy(d_norm.p, norm_sq, n * 4, hipMemcpyHostToDevice));
    if (self) {
        knn_device(d_x.p, d_norm.p, n, d_x.p, d_norm.p, n, Dk, true, num_cus, d_oi.p, d_od.p);
        nq = n;
    } else {
        DevBuf<float> d_q(nq * Dk), d_qn(nq);
        upload_padded(d_q.p, Dk, q, D, nq);
        HIP_CHECK(hipMemcpy(d_qn.p, qnorm, nq * 4, hipMemcpyHostToDevice));
        knn_device(d_q.p, d_qn.p, nq, d_x.p, d_norm.p, n, Dk, false, num_cus, d_oi.p, d_od.p);
    }
    HIP_CHECK(hipMemcpy(out_ids, d_oi.p, nq * kKnnK * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_dist, d_od.p, nq * kKnnK * 4, hipMemcpyDeviceToHost));
}

// ---- one graph layer on the GPU: kNN lists -> reverse CSR -> selection -------------------------------
// rows = the layer's vertices (row_ids == nullptr: every vertex, row == id).  d_sub = their vectors [rows][Dk]
// with norms, for the kNN; d_x = all vectors [n][D] for the exact distances of the selection.
struct LayerParams { uint32_t R; float alpha, tau, alpha_max; const float* d_err; };

inline void reverse_csr(const uint32_t* d_knn, size_t rows, DevBuf<uint64_t>& d_off, DevBuf<uint32_t>& d_rev) {
    DevBuf<uint32_t> d_deg(rows), d_cur(rows);
    HIP_CHECK(hipMemset(d_deg.p, 0, rows * 4));
    HIP_CHECK(hipMemset(d_cur.p, 0, rows * 4));
    const uint64_t ne = (uint64_t)rows * kKnnK;
    hipLaunchKernelGGL(reverse_count_kernel, dim3(grid_for(ne, 256)), dim3(256), 0, nullptr, d_knn, ne, d_deg.p);
    HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> deg(rows);
    HIP_CHECK(hipMemcpy(deg.data(), d_deg.p, rows * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off(rows + 1);
    off[0] = 0;
    for (size_t i = 0; i < rows; ++i) off[i + 1] = off[i] + deg[i];
    d_off.alloc(rows + 1);
    d_rev.alloc(off[rows] + 1);
    HIP_CHECK(hipMemcpy(d_off.p, off.data(), (rows + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(reverse_fill_kernel, dim3(grid_for(ne, 256)), dim3(256), 0, nullptr, d_knn, ne,
                       (uint32_t)kKnnK, d_off.p, d_cur.p, d_rev.p);
    HIP_CHECK(hipGetLastError());
}

inline void select_layer(const float* d_x, size_t D, uint32_t* d_knn /* row indices; translated in place */,
                         size_t rows, const uint32_t* d_row_ids, const LayerParams& lp, int num_cus,
                         uint32_t* d_out, uint32_t* d_out_cnt) {
    DevBuf<uint64_t> d_off;
    DevBuf<uint32_t> d_rev;
    reverse_csr(d_knn, rows, d_off, d_rev);
    if (d_row_ids) {
        hipLaunchKernelGGL(remap_ids_kernel, dim3(grid_for(rows * kKnnK, 256)), dim3(256), 0, nullptr, d_knn,
                           (uint64_t)rows * kKnnK, d_row_ids);
        HIP_CHECK(hipGetLastError());
    }
    SelectArgs a{};
    a.x = d_x; a.fwd = d_knn; a.rev_off = d_off.p; a.rev = d_rev.p; a.row_ids = d_row_ids; a.err = lp.d_err;
    a.rows = rows; a.D = (uint32_t)D; a.R = lp.R; a.alpha = lp.alpha; a.tau = lp.tau; a.alpha_max = lp.alpha_max;
    a.out = d_out; a.out_cnt = d_out_cnt;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rows, (uint64_t)num_cus * 32);
    hipLaunchKernelGGL(select_kernel, dim3(grid), dim3(64), select_lds(a.D), nullptr, a);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
}

// ---- upper layers: concurrent incremental insertion ---------------------------------------------------------
// Flat adjacency per level (slot = position of the vertex among the level's members, M + 1 entries each), a
// spin lock per member, vertices inserted by all host threads at once after a sequential seed.  A vertex
// descends greedily from the entry through the levels above its own, then on each of its levels runs a
// best-first search of width ef, keeps a diverse subset of what it found (the occlusion rule of
// select_kernel, on the host here) and links both ways; a list that overflows is re-selected.
struct UpperLayers {
    const float* vecs; size_t dim, n;
    const std::vector<int32_t>& levels;
    int max_level; uint32_t entry; size_t M, R;
    float tau = 0.0f, alpha = 1.2f;
    struct Level {
        std::vector<uint32_t> members;            // ascending vertex ids
        std::vector<uint32_t> adj;                // [members][M + 1]
        std::vector<uint8_t> deg;
        std::vector<std::atomic_flag> lock;
    };
    std::vector<Level> lv;                        // lv[l - 1] = level l
    std::vector<uint32_t> slot_of;                // vertex -> slot on level 1 (higher levels: binary search of `members`)

    UpperLayers(const float* v, size_t d, size_t n_, const std::vector<int32_t>& lev, int ml, uint32_t e, size_t M_, size_t R_)
        : vecs(v), dim(d), n(n_), levels(lev), max_level(ml), entry(e), M(M_), R(R_) {}

    float dist(uint32_t a, uint32_t b) const {
        const float* x = vecs + (size_t)a * dim;
        const float* y = vecs + (size_t)b * dim;
        float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t i = 0;
        for (; i + 8 <= dim; i += 8)
            for (int j = 0; j < 8; ++j) { const float t = x[i + j] - y[i + j]; c[j] += t * t; }
        float s = ((c[0] + c[4]) + (c[1] + c[5])) + ((c[2] + c[6]) + (c[3] + c[7]));
        for (; i < dim; ++i) { const float t = x[i] - y[i]; s += t * t; }
        return s;
    }
    size_t slot(int level, uint32_t v) const {
        if (level == 1) return slot_of[v];
        const auto& m = lv[level - 1].members;
        return (size_t)(std::lower_bound(m.begin(), m.end(), v) - m.begin());
    }
    struct Guard {
        std::atomic_flag& f;
        explicit Guard(std::atomic_flag& x) : f(x) { while (f.test_and_set(std::memory_order_acquire)) {} }
        ~Guard() { f.clear(std::memory_order_release); }
    };
    size_t copy_list(int level, uint32_t v, uint32_t* out) {
        Level& L = lv[level - 1];
        const size_t s = slot(level, v);
        Guard g(L.lock[s]);
        const size_t d = L.deg[s];
        std::memcpy(out, &L.adj[s * (M + 1)], d * 4);
        return d;
    }
    struct Near {
        float d; uint32_t id;
        bool operator<(const Near& o) const { return d < o.d || (d == o.d && id < o.id); }
        bool operator>(const Near& o) const { return o < *this; }
    };

    // keep at most `cap` of `c` (any order in, nearest first out): a candidate is dropped when a kept one
    // is closer to it than  la * its own distance - (la - 1) tau,  dropped ones refill an underfull list
    void diversify(std::vector<Near>& c, size_t cap) const {
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end(), [](const Near& a, const Near& b) { return a.id == b.id; }), c.end());
        if (c.size() <= cap) return;
        const float la = std::clamp(alpha * std::sqrt((float)c.size() / (float)cap), 1.0f, 2.0f * alpha);
        std::vector<Near> keep, rest;
        for (const Near& x : c) {
            bool hidden = false;
            if (keep.size() < cap)
                for (const Near& k : keep)
                    if (dist(x.id, k.id) < la * x.d - (la - 1.0f) * tau) { hidden = true; break; }
            if (!hidden && keep.size() < cap) keep.push_back(x); else rest.push_back(x);
        }
        for (size_t i = 0; i < rest.size() && keep.size() < cap; ++i) keep.push_back(rest[i]);
        c.swap(keep);
    }

    void insert(uint32_t v, std::vector<uint32_t>& stamp, uint32_t& epoch) {
        const int top = levels[v];
        uint32_t ep = entry;
        float epd = dist(v, ep);
        std::vector<uint32_t> nb(M + 1);
        for (int l = max_level; l > top; --l) {            // greedy descent above the vertex' own levels
            for (bool moved = true; moved;) {
                moved = false;
                const size_t d = copy_list(l, ep, nb.data());
                for (size_t i = 0; i < d; ++i) {
                    const float t = dist(v, nb[i]);
                    if (t < epd) { epd = t; ep = nb[i]; moved = true; }
                }
            }
        }
        const size_t n_upper = lv[0].members.size();
        for (int l = std::min(top, max_level); l >= 1; --l) {
            const float scale = 1.0f + (float)l * std::log((float)std::max<size_t>(n_upper, 2)) / std::log((float)std::max<size_t>(n, 2));
            const size_t ef = std::clamp((size_t)((float)R * scale), R, 4 * R);
            // best-first search of width ef from ep
            ++epoch;
            std::priority_queue<Near> best;                                      // farthest on top
            std::priority_queue<Near, std::vector<Near>, std::greater<Near>> open;  // nearest on top (needs operator>)
            stamp[ep] = epoch;
            best.push({dist(v, ep), ep});
            open.push(best.top());
            while (!open.empty()) {
                const Near cur = open.top();
                open.pop();
                if (best.size() >= ef && cur.d > best.top().d) break;
                const size_t d = copy_list(l, cur.id, nb.data());
                for (size_t i = 0; i < d; ++i) {
                    const uint32_t w = nb[i];
                    if (w == v || stamp[w] == epoch) continue;
                    stamp[w] = epoch;
                    const float t = dist(v, w);
                    if (best.size() < ef || t < best.top().d) {
                        best.push({t, w});
                        open.push({t, w});
                        if (best.size() > ef) best.pop();
                    }
                }
            }
            std::vector<Near> cand;
            while (!best.empty()) { if (best.top().id != v) cand.push_back(best.top()); best.pop(); }
            diversify(cand, M);
            Level& L = lv[l - 1];
            {
                const size_t s = slot(l, v);
                Guard g(L.lock[s]);
                L.deg[s] = (uint8_t)cand.size();
                for (size_t i = 0; i < cand.size(); ++i) L.adj[s * (M + 1) + i] = cand[i].id;
            }
            for (const Near& c : cand) {                    // back links; an overfull list is re-selected
                const size_t s = slot(l, c.id);
                Guard g(L.lock[s]);
                uint32_t* a = &L.adj[s * (M + 1)];
                size_t d = L.deg[s];
                bool have = false;
                for (size_t i = 0; i < d; ++i) have |= a[i] == v;
                if (have) continue;
                a[d++] = v;
                if (d > M) {
                    std::vector<Near> all(d);
                    for (size_t i = 0; i < d; ++i) all[i] = {dist(c.id, a[i]), a[i]};
                    diversify(all, M);
                    d = all.size();
                    for (size_t i = 0; i < d; ++i) a[i] = all[i].id;
                }
                L.deg[s] = (uint8_t)d;
            }
            if (!cand.empty()) ep = cand[0].id;
        }
    }

    void build() {
        lv = std::vector<Level>(max_level);
        for (int l = 1; l <= max_level; ++l) {
            Level& L = lv[l - 1];
            for (size_t v = 0; v < n; ++v)
                if (levels[v] >= l) L.members.push_back((uint32_t)v);
            L.adj.assign(L.members.size() * (M + 1), kInvalidNode);
            L.deg.assign(L.members.size(), 0);
            L.lock = std::vector<std::atomic_flag>(L.members.size());
            for (auto& f : L.lock) f.clear();
        }
        if (max_level == 0) return;
        slot_of.assign(n, kInvalidNode);
        for (size_t s = 0; s < lv[0].members.size(); ++s) slot_of[lv[0].members[s]] = (uint32_t)s;
        // insertion order: highest level first (the entry is the first vertex that reached the top level)
        std::vector<uint32_t> order = lv[0].members;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return levels[a] > levels[b]; });
        {
            auto it = std::find(order.begin(), order.end(), entry);
            std::rotate(order.begin(), it, it + 1);
        }
        // pruning parameters from the nearest-neighbour distances inside a sample of upper vertices
        {
            const size_t m = order.size();
            const size_t take = std::min(m, (size_t)(10.0 * std::sqrt((double)m)) + 1), pool = std::min(m, 2 * take);
            std::vector<float> nn;
            for (size_t i = 0; i < take; ++i) {
                float best = 3.402823466e+38f;
                for (size_t j = 0; j < pool; ++j)
                    if (j != i) best = std::min(best, dist(order[i], order[j]));
                if (best < 3.0e38f) nn.push_back(best);
            }
            if (!nn.empty()) {
                tau = mad_sigma(nn, median_of(nn));
                double mu = 0, var = 0;
                for (float d : nn) mu += d;
                mu /= nn.size();
                for (float d : nn) var += (d - mu) * (d - mu);
                var /= nn.size();
                alpha = 1.0f + (mu > kEpsSmall ? (float)(std::sqrt(var) / mu) : 0.2f);
            }
        }
        const size_t seed = std::min<size_t>(order.size(), 512);
        {
            std::vector<uint32_t> stamp(n, 0);
            uint32_t epoch = 0;
            for (size_t i = 1; i < seed; ++i) insert(order[i], stamp, epoch);       // order[0] = entry: nothing to link yet
        }
        if (order.size() > seed) {
            unsigned hw = std::min(64u, std::thread::hardware_concurrency());
            size_t nt = std::max<size_t>(1, hw ? hw : 4);
            if (const char* e = getenv("CPH_BUILD_THREADS")) nt = std::max(1, atoi(e));
            std::atomic<size_t> next{seed};
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    std::vector<uint32_t> stamp(n, 0);
                    uint32_t epoch = 0;
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= order.size()) break;
                        insert(order[i], stamp, epoch);
                    }
                });
            for (auto& x : th) x.join();
        }
    }

    std::vector<std::vector<UpperEdge>> export_layers(const std::vector<uint32_t>& renumber) const {
        std::vector<std::vector<UpperEdge>> out(max_level);
        for (int l = 1; l <= max_level; ++l) {
            const Level& L = lv[l - 1];
            auto& o = out[l - 1];
            o.resize(L.members.size());
            for (size_t s = 0; s < L.members.size(); ++s) {
                o[s].node = renumber[L.members[s]];
                for (size_t i = 0; i < L.deg[s]; ++i) o[s].nbrs.push_back(renumber[L.adj[s * (M + 1) + i]]);
            }
            std::sort(o.begin(), o.end(), [](const UpperEdge& a, const UpperEdge& b) { return a.node < b.node; });
        }
        return out;
    }
};

// ---- extreme-value tail model (core/evt_crc.hpp's EVTState is the file format) ---------------------
struct TailModel {          // 56 bytes, the layout the reference serialises
    float u = 0, p_u = 0, xi = 0, beta = 0;
    uint32_t n_tail = 0;
    bool fitted = false, use_empirical = false;
    float empirical[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static_assert(sizeof(TailModel) == 56, "EVTState layout");
constexpr float kTailAlphas[8] = {0.5f, 0.1f, 0.05f, 0.01f, 0.005f, 0.001f, 5e-4f, 1e-4f};

// Quantile of the residual distribution at exceedance probability alpha.
inline float tail_quantile(float alpha, const TailModel& m) {
    alpha = std::clamp(alpha, 1e-12f, 0.5f);
    if (alpha >= m.p_u) return m.u;
    if (m.use_empirical) {
        // piecewise linear in alpha between the stored checkpoints, log-linear beyond the last one
        for (int j = 0; j + 1 < 8; ++j)
            if (alpha >= kTailAlphas[j + 1]) {
                const float w = (alpha - kTailAlphas[j + 1]) / (kTailAlphas[j] - kTailAlphas[j + 1]);
                return m.empirical[j + 1] + w * (m.empirical[j] - m.empirical[j + 1]);
            }
        const float span = std::log(kTailAlphas[6] / kTailAlphas[7]);
        const float slope = span > kEpsSmall ? (m.empirical[7] - m.empirical[6]) / span : 0.0f;
        return m.empirical[7] + slope * std::log(kTailAlphas[7] / alpha);
    }
    const float ratio = m.p_u / alpha;     // generalised Pareto: u + beta/xi ((p_u/alpha)^xi - 1)
    return std::fabs(m.xi) < 1e-6f ? m.u + m.beta * std::log(ratio) : m.u + (m.beta / m.xi) * (std::pow(ratio, m.xi) - 1.0f);
}

// Maximum-likelihood generalised Pareto fit to the exceedances y > 0 through the one-parameter profile
// likelihood (theta = xi / beta):  xi(theta) = mean log(1 + theta y),  l*(theta) = -m [log(xi/theta) + xi + 1].
// Grid scan of theta * mean(y), then golden-section refinement of the best bracket.
inline bool gpd_mle(const std::vector<double>& y, double& xi, double& beta) {
    const size_t m = y.size();
    double mean = 0.0, ymax = 0.0;
    for (double v : y) { mean += v; ymax = std::max(ymax, v); }
    mean /= (double)m;
    if (!(mean > 0.0)) return false;
    auto profile = [&](double theta, double& xi_out) {
        if (std::fabs(theta) < 1e-12 / mean) { xi_out = 0.0; return -(double)m * (std::log(mean) + 1.0); }
        double s = 0.0;
        for (double v : y) s += std::log1p(theta * v);
        xi_out = s / (double)m;
        if (!(xi_out / theta > 0.0)) return -1e300;
        return -(double)m * (std::log(xi_out / theta) + xi_out + 1.0);
    };
    const double lo = -0.98 / ymax, hi = 6.0 / mean;
    const int G = 96;
    double best_t = 0.0, best_l = -1e300, dummy;
    std::vector<double> ts(G + 1);
    for (int g = 0; g <= G; ++g) {
        ts[g] = lo + (hi - lo) * (double)g / G;
        const double l = profile(ts[g], dummy);
        if (l > best_l) { best_l = l; best_t = ts[g]; }
    }
    double a = std::max(lo, best_t - (hi - lo) / G), b = std::min(hi, best_t + (hi - lo) / G);
    const double gr = 0.6180339887498949;
    for (int it = 0; it < 60; ++it) {
        const double c = b - gr * (b - a), d = a + gr * (b - a);
        if (profile(c, dummy) > profile(d, dummy)) b = d; else a = c;
    }
    const double theta = 0.5 * (a + b);
    double x;
    if (profile(theta, x) < -1e299) return false;
    xi = x;
    beta = std::fabs(theta) < 1e-12 / mean ? mean : x / theta;
    return beta > 0.0 && std::isfinite(beta) && std::isfinite(xi);
}

// Fit at one threshold (a quantile of the sorted residuals r).
inline TailModel fit_tail_at(const std::vector<float>& r, float thr_q, size_t min_tail) {
    TailModel t;
    const size_t n = r.size();
    if (n < 2 * min_tail) return t;
    const size_t cut = std::min((size_t)((float)n * thr_q), n - 1);
    t.u = r[cut];
    std::vector<double> y;
    for (size_t i = cut + 1; i < n; ++i)
        if (r[i] > t.u) y.push_back((double)r[i] - (double)t.u);
    t.n_tail = (uint32_t)y.size();
    t.p_u = (float)y.size() / (float)n;
    if (y.size() < min_tail) return t;
    double xi, beta;
    if (!gpd_mle(y, xi, beta)) {          // method of moments as the fallback
        double m1 = 0, m2 = 0;
        for (double v : y) { m1 += v; m2 += v * v; }
        m1 /= y.size(); m2 = m2 / y.size() - m1 * m1;
        xi = m2 > kEpsTiny ? 0.5 * (1.0 - m1 * m1 / m2) : 0.0;
        beta = std::max(m1 * (1.0 - xi), 1e-8);
    }
    t.xi = std::clamp((float)xi, -0.2f, 0.5f);        // the range the search-side quantile code expects
    t.beta = std::max((float)beta, 1e-8f);
    t.fitted = true;
    return t;
}

// Threshold choice by parameter stability: fits on a ladder of thresholds, the one that differs least from
// its neighbours wins; a Kolmogorov-Smirnov check of the winner decides between the parametric tail and
// stored empirical checkpoints.
inline TailModel fit_tail(const std::vector<float>& r, size_t min_tail, float q_lo, float q_hi) {
    if (q_hi <= q_lo) return fit_tail_at(r, q_lo, min_tail);
    const size_t steps = std::clamp((size_t)std::ceil(std::sqrt(std::log2(std::max((float)r.size(), 64.0f)))), (size_t)3, (size_t)8);
    std::vector<TailModel> fits(steps);
    std::vector<float> qs(steps);
    std::vector<size_t> ok;
    for (size_t k = 0; k < steps; ++k) {
        qs[k] = q_lo + (q_hi - q_lo) * (float)k / (float)(steps - 1);
        fits[k] = fit_tail_at(r, qs[k], min_tail);
        if (fits[k].fitted) ok.push_back(k);
    }
    if (ok.empty()) return TailModel{};
    size_t win = ok[0];
    if (ok.size() >= 2) {
        auto gap = [&](size_t a, size_t b) {
            const float dx = fits[a].xi - fits[b].xi;
            const float db = (fits[a].beta - fits[b].beta) / std::max(0.5f * (fits[a].beta + fits[b].beta), 1e-8f);
            return dx * dx + db * db;
        };
        float best = 3.402823466e+38f;
        for (size_t i = 0; i < ok.size(); ++i) {
            float s = 0.0f;
            int cnt = 0;
            if (i > 0) { s += gap(ok[i], ok[i - 1]); ++cnt; }
            if (i + 1 < ok.size()) { s += gap(ok[i], ok[i + 1]); ++cnt; }
            if (cnt && s / cnt < best) { best = s / cnt; win = ok[i]; }
        }
    }
    TailModel m = fits[win];
    if (m.n_tail >= 20) {
        const size_t cut = std::min((size_t)((float)r.size() * qs[win]), r.size() - 1);
        std::vector<float> tail;
        for (size_t i = cut + 1; i < r.size(); ++i)
            if (r[i] > m.u) tail.push_back(r[i] - m.u);           // ascending: r is sorted
        float ks = 0.0f;
        for (size_t i = 0; i < tail.size(); ++i) {
            const float emp = (float)(i + 1) / (float)tail.size();
            float cdf;
            if (std::fabs(m.xi) < 1e-6f) cdf = 1.0f - std::exp(-tail[i] / m.beta);
            else {
                const float z = 1.0f + m.xi * tail[i] / m.beta;
                cdf = z > 0.0f ? 1.0f - std::pow(z, -1.0f / m.xi) : 1.0f;
            }
            ks = std::max(ks, std::fabs(emp - cdf));
        }
        if (!tail.empty() && ks > 1.25f * 1.358f / std::sqrt((float)tail.size())) {
            m.use_empirical = true;
            for (int j = 0; j < 8; ++j) {
                const float q = std::clamp(1.0f - kTailAlphas[j] / m.p_u, 0.0f, 1.0f);
                m.empirical[j] = m.u + tail[std::min((size_t)(q * (float)tail.size()), tail.size() - 1)];
            }
        }
    }
    return m;
}

// ---- file-format records --------------------------------------------------------------------------
struct CalibrationRecord {      // CalibrationSnapshot, api/hnsw_index.hpp:33-58 (248 bytes in the file)
    float affine_a, affine_b, ip_qo_floor, median_nn_dist_sq, min_slack_sq, median_nop;
    TailModel evt;
    float gamma_min, gamma_max, gamma_beta;
    size_t gamma_warmup;
    int slack_levels;
    float search_ip_slack_levels[32];
    int search_num_slack_levels;
    float search_gamma;
};
static_assert(sizeof(CalibrationRecord) == 248, "CalibrationSnapshot layout");
struct GraphStatsRecord { float avg_degree, alpha, tau, alpha_max; };
struct ProfileRecord {          // IndexProfile (72 bytes in the file)
    size_t n = 0, D = 0, R = 0, bits = 0, evt_min_tail = 0, min_calib_samples = 0;
    int slack_levels = 0;
    GraphStatsRecord graph_stats{};
};
static_assert(sizeof(ProfileRecord) == 72, "IndexProfile layout");

// ---- Huber-weighted straight line y = a x + b ---------------------------------------------------------
inline void robust_line(const std::vector<float>& x, const std::vector<float>& y, double& a, double& b) {
    const size_t n = x.size();
    auto wls = [&](const std::vector<double>* w, double& aa, double& bb) {
        double sw = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;
        for (size_t i = 0; i < n; ++i) {
            const double wi = w ? (*w)[i] : 1.0;
            sw += wi; sx += wi * x[i]; sy += wi * y[i]; sxx += wi * x[i] * x[i]; sxy += wi * x[i] * y[i];
        }
        const double mx = sx / sw, my = sy / sw, vx = sxx / sw - mx * mx, cxy = sxy / sw - mx * my;
        if (vx <= kEpsSmall) return false;
        aa = cxy / vx;
        bb = my - aa * mx;
        return true;
    };
    a = 1.0; b = 0.0;
    wls(nullptr, a, b);
    std::vector<double> w(n);
    std::vector<float> res(n);
    for (int round = 0; round < 10; ++round) {
        for (size_t i = 0; i < n; ++i) res[i] = std::fabs(y[i] - (float)(a * x[i] + b));
        const float cut = 1.345f * 1.4826f * median_of(res);      // Huber's k on a MAD scale
        if (cut < kEpsSmall) break;
        for (size_t i = 0; i < n; ++i) w[i] = res[i] <= cut ? 1.0 : (double)(cut / res[i]);
        double na = a, nb = b;
        if (!wls(&w, na, nb)) break;
        const bool done = std::fabs(na - a) + std::fabs(nb - b) < 1e-6;
        a = na; b = nb;
        if (done) break;
    }
}

}  // namespace build
}  // namespace cph
