// builder_pipeline.h — finalize(): layers, upper layers, GPU kNN, pruning, edge encoding, BFS
// reorder, calibration.  See builder.h for scope and the deliberate differences.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>

#include "builder.h"
#include "device_knn.h"

namespace cph {
namespace build {

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

inline size_t isqrt_sz(size_t n) {
    if (n < 2) return n;
    size_t x = n, y = (x + 1) / 2;
    while (y < x) { x = y; y = (x + n / x) / 2; }
    return x;
}

struct UpperBuilder {  // api/hnsw_index.hpp:476-716 on the pre-reorder ids
    const float* raw; size_t D, n, M;
    std::vector<int32_t>& levels;
    std::vector<std::vector<UpperEdge>>& layers;
    int max_level; uint32_t entry;
    float tau = 0.0f, alpha = 1.2f;
    std::vector<uint64_t> visit; uint64_t epoch = 0;

    const float* vec(uint32_t i) const { return raw + (size_t)i * D; }
    UpperEdge* find(int level, uint32_t node) {
        auto& L = layers[level - 1];
        auto it = std::lower_bound(L.begin(), L.end(), node, [](const UpperEdge& e, uint32_t v) { return e.node < v; });
        return (it != L.end() && it->node == node) ? &*it : nullptr;
    }
    UpperEdge& get_or_create(int level, uint32_t node) {
        auto& L = layers[level - 1];
        auto it = std::lower_bound(L.begin(), L.end(), node, [](const UpperEdge& e, uint32_t v) { return e.node < v; });
        if (it != L.end() && it->node == node) return *it;
        return *L.insert(it, UpperEdge{node, {}});
    }
    uint32_t greedy(const float* q, uint32_t ep, int level) {
        float best = l2sq(D, q, vec(ep));
        uint32_t bid = ep;
        bool improved = true;
        while (improved) {
            improved = false;
            UpperEdge* e = find(level, bid);
            if (!e) break;
            for (uint32_t x : e->nbrs) {
                float d = l2sq(D, q, vec(x));
                if (d < best) { best = d; bid = x; improved = true; }
            }
        }
        return bid;
    }
    std::vector<Cand> search_layer(const float* q, uint32_t ep, int level, size_t ef) {
        auto gt = [](const Cand& a, const Cand& b) { return a.dist > b.dist; };
        std::priority_queue<Cand, std::vector<Cand>, decltype(gt)> cands(gt);
        std::priority_queue<Cand> nearest;
        float epd = l2sq(D, q, vec(ep));
        cands.push({ep, epd});
        nearest.push({ep, epd});
        ++epoch;
        visit[ep] = epoch;
        while (!cands.empty()) {
            Cand cur = cands.top();
            cands.pop();
            if (nearest.size() >= ef && cur.dist > nearest.top().dist) break;
            UpperEdge* e = find(level, cur.id);
            if (!e) continue;
            for (uint32_t x : e->nbrs) {
                if (visit[x] == epoch) continue;
                visit[x] = epoch;
                float d = l2sq(D, q, vec(x));
                if (nearest.size() < ef || d < nearest.top().dist) {
                    cands.push({x, d});
                    nearest.push({x, d});
                    if (nearest.size() > ef) nearest.pop();
                }
            }
        }
        std::vector<Cand> res;
        while (!nearest.empty()) { res.push_back(nearest.top()); nearest.pop(); }
        std::sort(res.begin(), res.end());
        return res;
    }
    void prune(uint32_t node, int level) {
        auto& nb = get_or_create(level, node).nbrs;
        if (nb.size() <= M) return;
        std::vector<Cand> c;
        for (uint32_t id : nb) c.push_back({id, l2sq(D, vec(node), vec(id))});
        auto sel = select_alpha_cng(std::move(c), M, [&](uint32_t a, uint32_t b) { return l2sq(D, vec(a), vec(b)); },
                                    [](uint32_t) { return 0.0f; }, alpha, tau);
        auto& nb2 = get_or_create(level, node).nbrs;
        nb2.clear();
        for (auto& s : sel) nb2.push_back(s.id);
    }
    void run() {
        const size_t R = 32;
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return levels[a] > levels[b]; });
        size_t n_upper = 0;
        for (size_t i = 0; i < n; ++i) { if (levels[order[i]] > 0) ++n_upper; else break; }
        visit.assign(n, 0);
        size_t dist_samples = std::min(static_cast<size_t>(std::sqrt(static_cast<float>(n_upper)) * 10.0f), n_upper);
        size_t nn_limit = std::min(dist_samples * 2, n_upper);
        std::vector<float> nnd;
        for (size_t idx = 0; idx < n && nnd.size() < dist_samples; ++idx) {
            uint32_t node = order[idx];
            if (levels[node] == 0) break;
            float best = 3.402823466e+38f;
            for (size_t j = 0; j < n && j < nn_limit; ++j) {
                uint32_t o = order[j];
                if (o == node) continue;
                if (levels[o] == 0) break;
                best = std::min(best, l2sq(D, vec(node), vec(o)));
            }
            if (best < 3.402823466e+38f) nnd.push_back(best);
        }
        if (!nnd.empty()) {
            std::sort(nnd.begin(), nnd.end());
            float med = nnd[nnd.size() / 2];
            std::vector<float> ad(nnd.size());
            for (size_t i = 0; i < nnd.size(); ++i) ad[i] = std::fabs(nnd[i] - med);
            std::sort(ad.begin(), ad.end());
            tau = 1.4826f * ad[ad.size() / 2];
            float mean = 0;
            for (float d : nnd) mean += d;
            mean /= nnd.size();
            float var = 0;
            for (float d : nnd) var += (d - mean) * (d - mean);
            var /= nnd.size();
            alpha = 1.0f + (mean > kEpsSmall ? std::sqrt(var) / mean : 0.2f);
        }
        for (size_t idx = 0; idx < n; ++idx) {
            uint32_t node = order[idx];
            int nl = levels[node];
            if (nl == 0) break;
            uint32_t ep = entry;
            for (int level = max_level; level > nl; --level) ep = greedy(vec(node), ep, level);
            for (int level = std::min(nl, max_level); level >= 1; --level) {
                size_t ef = std::clamp(
                    static_cast<size_t>(static_cast<float>(R) *
                                        (1.0f + static_cast<float>(level) *
                                                    std::log(static_cast<float>(std::max(n_upper, size_t(2)))) /
                                                    std::log(static_cast<float>(std::max(n, size_t(2)))))),
                    R, R * 4);
                auto cands = search_layer(vec(node), ep, level, ef);
                auto sel = select_alpha_cng(std::move(cands), M,
                                            [&](uint32_t a, uint32_t b) { return l2sq(D, vec(a), vec(b)); },
                                            [](uint32_t) { return 0.0f; }, alpha, tau);
                auto& mine = get_or_create(level, node).nbrs;
                mine.clear();
                for (auto& s : sel) mine.push_back(s.id);
                for (auto& s : sel) {
                    auto& nb = get_or_create(level, s.id).nbrs;
                    nb.push_back(node);
                    if (nb.size() > M) prune(s.id, level);
                }
                if (!sel.empty()) ep = sel[0].id;
            }
        }
    }
};

// RAII device buffer for the construction kernels.
template <class T>
struct BuildBuf {
    T* p = nullptr;
    explicit BuildBuf(size_t count) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            if (e == hipErrorOutOfMemory) throw std::bad_alloc();
            throw std::runtime_error(std::string("HIP error in index construction: ") + hipGetErrorString(e));
        }
    }
    ~BuildBuf() { if (p) (void)hipFree(p); }
    BuildBuf(const BuildBuf&) = delete;
    BuildBuf& operator=(const BuildBuf&) = delete;
};
inline void build_ck(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in index construction: ") + hipGetErrorString(e));
}

// Exact 32-NN of query rows [0, nq) of d_q against base rows d_b, all device pointers; ascending by
// distance.  Launched in slices of row blocks so that no single launch runs for minutes.
// D must be a multiple of 32 (the kernel's K stage); narrower rows are zero-padded by the callers.
inline void knn_device(const float* d_q, const float* d_qnorm, size_t nq, const float* d_b, const float* d_bnorm,
                       size_t nb, size_t D, bool exclude_self, int num_cus, uint32_t* d_ids, float* d_dist) {
    if (D % kKnnKC != 0 || nq == 0 || nb == 0) throw std::invalid_argument("knn_device: D must be a multiple of 32");
    const uint32_t rows_per_launch = (uint32_t)num_cus * 8u * kKnnTile;
    for (size_t rb = 0; rb < nq; rb += rows_per_launch) {
        KnnArgs a{d_q, d_b, d_bnorm, d_qnorm, (uint32_t)nq, (uint32_t)nb, (uint32_t)D, (uint32_t)rb,
                  (uint32_t)std::min<size_t>(nq, rb + rows_per_launch), exclude_self ? 1u : 0u, d_ids, d_dist};
        const uint32_t grid = (a.row_end - a.row_begin + kKnnTile - 1) / kKnnTile;
        hipLaunchKernelGGL(knn_mfma_kernel, dim3(grid), dim3(256), 0, nullptr, a);
        build_ck(hipGetLastError());
        build_ck(hipDeviceSynchronize());
    }
}

// Host-pointer convenience: x[n][D] against itself (self excluded) or q[nq][D] against x[n][D].
inline void gpu_knn(const float* q, const float* qnorm, size_t nq, const float* x, const float* norm_sq, size_t n,
                    size_t D, int num_cus, uint32_t* out_ids, float* out_dist) {
    const bool self = (q == nullptr);
    const size_t Dk = (D + kKnnKC - 1) / kKnnKC * kKnnKC;      // D = 16 -> 32: zero columns
    BuildBuf<float> d_x(n * Dk), d_norm(n), d_od((self ? n : nq) * kKnnK);
    BuildBuf<uint32_t> d_oi((self ? n : nq) * kKnnK);
    auto upload = [&](float* dst, const float* src, size_t rows) {
        if (Dk == D) { build_ck(hipMemcpy(dst, src, rows * D * 4, hipMemcpyHostToDevice)); return; }
        build_ck(hipMemset(dst, 0, rows * Dk * 4));
        build_ck(hipMemcpy2D(dst, Dk * 4, src, D * 4, D * 4, rows, hipMemcpyHostToDevice));
    };
    upload(d_x.p, x, n);
    build_ck(hipMemcpy(d_norm.p, norm_sq, n * 4, hipMemcpyHostToDevice));
    if (self) {
        knn_device(d_x.p, d_norm.p, n, d_x.p, d_norm.p, n, Dk, true, num_cus, d_oi.p, d_od.p);
        nq = n;
    } else {
        BuildBuf<float> d_q(nq * Dk), d_qn(nq);
        upload(d_q.p, q, nq);
        build_ck(hipMemcpy(d_qn.p, qnorm, nq * 4, hipMemcpyHostToDevice));
        knn_device(d_q.p, d_qn.p, nq, d_x.p, d_norm.p, n, Dk, false, num_cus, d_oi.p, d_od.p);
    }
    build_ck(hipMemcpy(out_ids, d_oi.p, nq * kKnnK * 4, hipMemcpyDeviceToHost));
    build_ck(hipMemcpy(out_dist, d_od.p, nq * kKnnK * 4, hipMemcpyDeviceToHost));
}

// The whole finalize.  `vecs` = n x dim input rows.  Fills `hi` (reference-layout host index).
inline void finalize_index(HostIndex& hi, const float* vecs, size_t n, size_t dim, size_t D, size_t bw,
                           int num_cus, bool verbose) {
    Timer tm;
    auto note = [&](const char* what) { if (verbose) fprintf(stderr, "[build] %-28s %.2f s\n", what, tm.lap()); };
    const size_t R = 32;
    hi = HostIndex();
    hi.D = D; hi.bw = bw; hi.dim = dim; hi.n = n; hi.seed = 42;
    hi.RL = make_ref_layout(D, bw);
    hi.rot.init(D, 42);
    const size_t M_UPPER = R / 2 + std::min(isqrt_sz(D) / 4, R / 4);
    hi.mL = 1.0 / std::log(static_cast<double>(M_UPPER));
    IndexProfile prof;
    prof.n = n; prof.D = D; prof.R = R; prof.bits = bw;
    prof.evt_min_tail = std::max<size_t>(64, static_cast<size_t>(std::sqrt(static_cast<double>(n))));
    prof.min_calib_samples = std::clamp(static_cast<size_t>(10.0 * std::sqrt(static_cast<double>(n))), size_t(200), n);
    {
        float log_n = std::log2(static_cast<float>(std::max(n, size_t(64))));
        prof.slack_levels = std::clamp(static_cast<int>(std::ceil(std::log2(std::max(10.0f * log_n, 4.0f)))), 4, 32);
    }

    // ---- vectors, norms, centroid, own codes (graph/rabitq_graph.hpp:73-92; encoder :42-71,225-262,326-352)
    std::vector<float> raw(n * D, 0.0f), norm_sq(n);
    parallel_for(n, 1024, [&](size_t lo, size_t hi_) {
        for (size_t i = lo; i < hi_; ++i) {
            std::memcpy(&raw[i * D], vecs + i * dim, dim * 4);
            float s = 0.0f;
            for (size_t j = 0; j < dim; ++j) s = std::fmaf(raw[i * D + j], raw[i * D + j], s);
            norm_sq[i] = s;
        }
    });
    std::vector<float> centroid(dim, 0.0f);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < dim; ++j) centroid[j] += vecs[i * dim + j];
    for (size_t j = 0; j < dim; ++j) centroid[j] *= 1.0f / static_cast<float>(n);
    DataEncoder enc;
    enc.init(&hi.rot, D, dim, bw);
    std::vector<uint8_t> search(n * hi.RL.vertex_bytes, 0);
    std::vector<float> own_nop(n);
    const size_t words = (D + 63) / 64;
    const size_t code_meta = round_up(bw * words * 8, 64);
    parallel_for(n, 256, [&](size_t lo, size_t hi_) {
        std::vector<float> c(D), rot(D);
        std::vector<int> codes;
        EdgeCode e;
        for (size_t i = lo; i < hi_; ++i) {
            float ns = 0.0f;
            for (size_t j = 0; j < dim; ++j) { c[j] = vecs[i * dim + j] - centroid[j]; ns += c[j] * c[j]; }
            for (size_t j = dim; j < D; ++j) c[j] = 0.0f;
            const float nrm = std::sqrt(ns);
            own_nop[i] = nrm;
            uint8_t* v = &search[i * hi.RL.vertex_bytes];
            float ipqo = 0.0f;
            if (!(nrm < 1e-8f / static_cast<float>(D))) {
                for (size_t j = 0; j < dim; ++j) c[j] *= 1.0f / nrm;
                enc.rotate_scaled(c.data(), rot.data());
                enc.quantize(rot.data(), nullptr, e, codes);
                ipqo = e.ip_qo;
                for (size_t b = 0; b < bw; ++b)
                    for (size_t d = 0; d < D; ++d)
                        if ((e.u[d] >> (bw - 1 - b)) & 1) v[(b * words + d / 64) * 8 + (d % 64) / 8] |= (uint8_t)(1u << (d % 8));
            }
            std::memcpy(v + code_meta, &nrm, 4);
            std::memcpy(v + code_meta + 4, &ipqo, 4);
        }
    });
    note("vectors + own codes");

    // ---- layers (api/hnsw_index.hpp:484-503) + upper layers (:505-615)
    std::vector<int32_t> levels(n);
    int max_level = 0;
    uint32_t entry = kInvalidNode;
    {
        std::mt19937_64 rng(42);
        std::uniform_real_distribution<double> dist(0.0, 1.0);
        for (size_t i = 0; i < n; ++i) {
            double r = dist(rng);
            if (r < 1e-15) r = 1e-15;
            int level = static_cast<int>(-std::log(r) * hi.mL);
            levels[i] = level;
            if (entry == kInvalidNode || level > max_level) { max_level = level; entry = (uint32_t)i; }
        }
    }
    std::vector<std::vector<UpperEdge>> layers(max_level);
    UpperBuilder ub{raw.data(), D, n, M_UPPER, levels, layers, max_level, entry, 0.0f, 1.2f, {}, 0};
    ub.run();
    note("upper layers");

    // ---- exact 32-NN working lists on the GPU (replaces NNDescent) ------------------------
    std::vector<uint32_t> knn_ids(n * kKnnK);
    std::vector<float> knn_d(n * kKnnK);
    gpu_knn(nullptr, nullptr, 0, raw.data(), norm_sq.data(), n, D, num_cus, knn_ids.data(), knn_d.data());
    note("GPU exact 32-NN");

    auto vec = [&](uint32_t i) { return &raw[(size_t)i * D]; };
    // ---- graph statistics (graph_refinement.hpp:266-383) on the working lists ---------------
    GraphStats gs{};
    {
        size_t sample = std::min(static_cast<size_t>(std::sqrt(static_cast<double>(n))), n);
        std::mt19937 rng(43);
        std::vector<size_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::shuffle(idx.begin(), idx.end(), rng);
        idx.resize(sample);
        std::vector<float> nd, ind, nnd;
        float total_deg = 0.0f;
        for (size_t i = 0; i < n; ++i) {
            uint32_t c = 0;
            for (int s = 0; s < kKnnK; ++s) c += knn_ids[i * kKnnK + s] != kInvalidNode;
            total_deg += (float)c;
        }
        gs.avg_degree = total_deg / static_cast<float>(std::max(n, size_t(1)));
        const size_t inter_limit = std::clamp(static_cast<size_t>(2.0 * std::sqrt(static_cast<double>(R))), size_t(4), R);
        for (size_t i : idx) {
            const uint32_t* w = &knn_ids[i * kKnnK];
            const float* wd = &knn_d[i * kKnnK];
            size_t cnt = 0;
            while (cnt < (size_t)kKnnK && w[cnt] != kInvalidNode) ++cnt;
            for (size_t s = 0; s < cnt; ++s) nd.push_back(wd[s]);
            if (cnt) nnd.push_back(wd[0]);
            const size_t il = std::min(cnt, inter_limit);
            for (size_t j = 0; j < il; ++j)
                for (size_t k = j + 1; k < il; ++k) ind.push_back(l2sq(D, vec(w[j]), vec(w[k])));
        }
        if (nd.empty() || ind.empty() || nnd.empty()) { gs.alpha = 1.0f; gs.tau = 0.0f; gs.alpha_max = 4.0f; }
        else {
            std::sort(nd.begin(), nd.end()); std::sort(ind.begin(), ind.end()); std::sort(nnd.begin(), nnd.end());
            const float neps = 1e-8f / static_cast<float>(D);
            const float med = nd[nd.size() / 2], q1 = nd[nd.size() / 4], q3 = nd[3 * nd.size() / 4];
            const float q3q1 = q1 > neps ? q3 / q1 : 2.0f;
            float mean = 0; for (float d : nd) mean += d; mean /= nd.size();
            float var = 0; for (float d : nd) var += (d - mean) * (d - mean); var /= nd.size();
            const float cv = mean > neps ? std::sqrt(var) / mean : 0.2f;
            const float nnm = nnd[nnd.size() / 2];
            std::vector<float> ad(nnd.size());
            for (size_t i = 0; i < nnd.size(); ++i) ad[i] = std::fabs(nnd[i] - nnm);
            std::sort(ad.begin(), ad.end());
            const float d_inter = ind[ind.size() / 4];
            gs.alpha = d_inter < neps ? 1.0f + cv : med / d_inter;
            gs.alpha_max = std::min(q3q1, 5.0f);
            gs.alpha = std::clamp(gs.alpha, 1.0f, gs.alpha_max);
            gs.alpha_max = std::max(gs.alpha_max, 2.0f * gs.alpha);
            gs.tau = 1.4826f * ad[ad.size() / 2];
        }
    }
    prof.graph_stats = gs;

    // ---- first pass keeps the working lists (<= R candidates: no pruning, :535-536), then the
    // reverse-edge pass (:386-429) prunes own + reverse candidates with alpha-CNG ------------
    const float err_tol = 1.0f / std::sqrt(static_cast<float>(D));
    // exact edge lengths in the reference's summation order (the GPU distances served only to rank)
    parallel_for(n, 256, [&](size_t lo, size_t hi_) {
        for (size_t u = lo; u < hi_; ++u)
            for (int s = 0; s < kKnnK; ++s) {
                const uint32_t v = knn_ids[u * kKnnK + s];
                if (v != kInvalidNode) knn_d[u * kKnnK + s] = l2sq(D, vec((uint32_t)u), vec(v));
            }
    });
    std::vector<std::vector<Cand>> rev(n);
    {
        std::vector<uint32_t> indeg(n, 0);
        for (size_t e = 0; e < n * (size_t)kKnnK; ++e)
            if (knn_ids[e] != kInvalidNode) ++indeg[knn_ids[e]];
        for (size_t v = 0; v < n; ++v) rev[v].reserve(indeg[v]);
    }
    for (size_t u = 0; u < n; ++u)
        for (int s = 0; s < kKnnK; ++s) {
            const uint32_t v = knn_ids[u * kKnnK + s];
            if (v == kInvalidNode) continue;
            rev[v].push_back({(uint32_t)u, knn_d[u * kKnnK + s]});
        }
    std::vector<uint32_t> nbr(n * R, kInvalidNode);
    std::vector<uint8_t> nbr_cnt(n, 0);
    parallel_for(n, 64, [&](size_t lo, size_t hi_) {
        for (size_t i = lo; i < hi_; ++i) {
            std::vector<Cand> all;
            for (int s = 0; s < kKnnK; ++s) {
                const uint32_t w = knn_ids[i * kKnnK + s];
                if (w == kInvalidNode) continue;
                all.push_back({w, knn_d[i * kKnnK + s]});
            }
            std::vector<Cand> sel;
            if (rev[i].empty()) {
                sel = all;  // untouched by the reverse pass: the first-pass list stands
                std::sort(sel.begin(), sel.end());
            } else {
                for (const auto& c : rev[i]) if (c.id != i) all.push_back(c);
                sel = select_alpha_cng(std::move(all), R,
                                       [&](uint32_t a, uint32_t b) { return l2sq(D, vec(a), vec(b)); },
                                       [&](uint32_t x) { return err_tol * own_nop[x]; }, gs.alpha, gs.tau, gs.alpha_max);
            }
            nbr_cnt[i] = (uint8_t)std::min<size_t>(sel.size(), R);
            for (size_t j = 0; j < nbr_cnt[i]; ++j) nbr[i * R + j] = sel[j].id;
        }
    });
    { std::vector<std::vector<Cand>>().swap(rev); }
    note("stats + reverse pass + prune");

    // ---- encode every edge into the reference-layout blocks (prune_and_write, :30-68) ---------
    parallel_for(n, 64, [&](size_t lo, size_t hi_) {
        std::vector<float> rp(D), tmp;
        std::vector<int> codes;
        EdgeCode e;
        for (size_t i = lo; i < hi_; ++i) {
            uint8_t* nb = &search[i * hi.RL.vertex_bytes + hi.RL.nb_off];
            std::memset(nb + hi.RL.ids, 0xFF, 128);
            enc.rotate_scaled(vec((uint32_t)i), rp.data());
            const uint32_t cnt = nbr_cnt[i];
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint32_t v = nbr[i * R + j];
                enc.encode_edge(vec((uint32_t)i), vec(v), rp.data(), e, tmp, codes);
                write_slot(nb, hi.RL, D, bw, j, v, e);
            }
            std::memcpy(nb + hi.RL.count, &cnt, 4);
        }
    });
    note("edge encoding");

    // ---- hub entry (rabitq_graph.hpp:303-340) and BFS reorder (:208-278) --------------------
    std::vector<double> cen(dim, 0.0);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < dim; ++j) cen[j] += raw[i * D + j];
    for (size_t j = 0; j < dim; ++j) cen[j] *= 1.0 / static_cast<double>(n);
    uint32_t hub = 0;
    {
        struct CD { uint32_t id; double d; };
        std::vector<CD> cd(n);
        for (size_t i = 0; i < n; ++i) {
            double s = 0;
            for (size_t j = 0; j < dim; ++j) { double t = raw[i * D + j] - cen[j]; s += t * t; }
            cd[i] = {(uint32_t)i, s};
        }
        size_t top = std::max<size_t>(1, static_cast<size_t>(std::sqrt(static_cast<double>(n))));
        if (top < n) std::partial_sort(cd.begin(), cd.begin() + top, cd.end(), [](const CD& a, const CD& b) { return a.d < b.d; });
        uint32_t best = kInvalidNode;
        size_t bdeg = 0;
        for (size_t i = 0; i < top && i < n; ++i)
            if (best == kInvalidNode || nbr_cnt[cd[i].id] > bdeg) { bdeg = nbr_cnt[cd[i].id]; best = cd[i].id; }
        hub = best;
    }
    std::vector<uint32_t> old_to_new(n, kInvalidNode), new_to_old(n);
    {
        std::vector<uint8_t> seen(n, 0);
        std::queue<uint32_t> q;
        uint32_t next = 0;
        auto bfs = [&](uint32_t s) {
            if (s >= n || seen[s]) return;
            q.push(s); seen[s] = 1;
            while (!q.empty()) {
                uint32_t c = q.front(); q.pop();
                old_to_new[c] = next; new_to_old[next] = c; ++next;
                for (uint32_t j = 0; j < nbr_cnt[c]; ++j) {
                    uint32_t v = nbr[c * R + j];
                    if (v != kInvalidNode && v < n && !seen[v]) { seen[v] = 1; q.push(v); }
                }
            }
        };
        bfs(hub);
        for (size_t i = 0; i < n; ++i) if (!seen[i]) bfs((uint32_t)i);
    }
    hi.raw.resize(n * D); hi.norm_sq.resize(n); hi.search_data.resize(n * hi.RL.vertex_bytes); hi.levels.resize(n);
    parallel_for(n, 1024, [&](size_t lo, size_t hi_) {
        for (size_t nw = lo; nw < hi_; ++nw) {
            const uint32_t od = new_to_old[nw];
            std::memcpy(&hi.raw[nw * D], &raw[(size_t)od * D], D * 4);
            hi.norm_sq[nw] = norm_sq[od];
            hi.levels[nw] = levels[od];
            uint8_t* dst = &hi.search_data[nw * hi.RL.vertex_bytes];
            std::memcpy(dst, &search[(size_t)od * hi.RL.vertex_bytes], hi.RL.vertex_bytes);
            uint32_t* ids = reinterpret_cast<uint32_t*>(dst + hi.RL.nb_off + hi.RL.ids);
            for (int j = 0; j < 32; ++j)
                if (ids[j] != kInvalidNode && ids[j] < n) ids[j] = old_to_new[ids[j]];
        }
    });
    for (auto& layer : layers) {
        for (auto& e : layer) {
            e.node = old_to_new[e.node];
            for (auto& x : e.nbrs) x = old_to_new[x];
        }
        std::sort(layer.begin(), layer.end(), [](const UpperEdge& a, const UpperEdge& b) { return a.node < b.node; });
    }
    hi.upper = std::move(layers);
    hi.max_level = max_level;
    hi.entry = old_to_new[entry];
    hi.upper_tau = ub.tau;
    hi.upper_alpha = ub.alpha;
    hi.centroid = centroid;
    { std::vector<float>().swap(raw); std::vector<uint8_t>().swap(search); }
    note("hub + BFS reorder");

    // ---- estimator calibration (api/hnsw_index.hpp:718-1139) ----------------------------------
    CalibrationSnapshot cal{};
    {
        const size_t num_samples = std::min(prof.min_calib_samples, n);
        if (n < 50) throw std::runtime_error("Calibration requires at least 50 nodes.");
        std::vector<uint32_t> sid(n);
        std::iota(sid.begin(), sid.end(), 0u);
        std::mt19937 rng(static_cast<uint32_t>(42 + 99999));
        std::shuffle(sid.begin(), sid.end(), rng);
        const size_t n_db = std::min(num_samples, n), n_synth = std::min(num_samples / 2, n);
        std::vector<float> dim_var(D, 0.0f), dim_mean(D, 0.0f);
        const size_t var_sample = std::min(n, num_samples / 4);
        for (size_t i = 0; i < var_sample; ++i) {
            const float* v = hi.vec(sid[i]);
            for (size_t d = 0; d < D; ++d) { dim_var[d] += v[d] * v[d]; dim_mean[d] += v[d]; }
        }
        for (size_t d = 0; d < D; ++d) {
            dim_mean[d] /= static_cast<float>(var_sample);
            dim_var[d] = dim_var[d] / static_cast<float>(var_sample) - dim_mean[d] * dim_mean[d];
            if (dim_var[d] < kEpsSmall) dim_var[d] = kEpsSmall;
        }
        struct CS { float nop, ipc, ipq, dqp; uint32_t nbr; size_t qi; };
        std::vector<float> ipqo_vals, ps_ipc, ps_ipq, truths, nn_d, nops;
        std::vector<CS> cs;
        std::vector<std::vector<float>> qbuf;
        size_t cursor = 0;
        std::vector<float> work(D);
        EncodedQuery eq;
        auto process = [&](const float* q, size_t qi) {
            uint32_t parent = sid[cursor % n];
            ++cursor;
            float best = l2sq(D, q, hi.vec(parent));
            {
                const uint8_t* nb = hi.nb(parent);
                uint32_t cnt; std::memcpy(&cnt, nb + hi.RL.count, 4);
                const uint32_t* ids = reinterpret_cast<const uint32_t*>(nb + hi.RL.ids);
                const uint32_t p0 = parent;
                for (uint32_t i = 0; i < cnt; ++i) {
                    if (ids[i] == kInvalidNode) break;
                    float d = l2sq(D, q, hi.vec(ids[i]));
                    if (d < best) { best = d; parent = ids[i]; }
                }
                (void)p0;
            }
            nn_d.push_back(best);
            const uint8_t* pnb = hi.nb(parent);
            uint32_t pcnt; std::memcpy(&pcnt, pnb + hi.RL.count, 4);
            std::memcpy(work.data(), q, D * 4);
            encode_query(hi.rot, work.data(), eq);
            const float dqp = l2sq(D, q, hi.vec(parent));
            uint32_t sums[32];
            host_block_sums(pnb, hi.RL, D, bw, eq.qu.data(), sums);
            const uint32_t* ids = reinterpret_cast<const uint32_t*>(pnb + hi.RL.ids);
            const float* nopa = reinterpret_cast<const float*>(pnb + hi.RL.nop);
            const float* ipqa = reinterpret_cast<const float*>(pnb + hi.RL.ip_qo);
            const float* ipca = reinterpret_cast<const float*>(pnb + hi.RL.ip_cp);
            const uint16_t* popa = reinterpret_cast<const uint16_t*>(pnb + hi.RL.pop);
            const uint16_t* wpopa = bw > 1 ? reinterpret_cast<const uint16_t*>(pnb + hi.RL.wpop) : nullptr;
            const float K = static_cast<float>((1u << bw) - 1), invK = 1.0f / K;
            for (uint32_t j = 0; j < pcnt && j < 32; ++j) {
                const uint32_t nbid = ids[j];
                if (nbid == kInvalidNode) break;
                const float ipqo = ipqa[j];
                ipqo_vals.push_back(ipqo);
                const float nop = std::max(nopa[j], kEpsSmall);
                nops.push_back(nop);
                float ipa;
                if (bw == 1) ipa = eq.A * static_cast<float>(sums[j]) + eq.B * static_cast<float>(popa[j]) + eq.C;
                else ipa = eq.A * invK * static_cast<float>(sums[j]) + eq.B * invK * static_cast<float>(wpopa[j]) + eq.C;
                const float ipc = ipa - ipca[j];
                const float ipq = std::max(std::fabs(ipqo), kEpsMedium);
                const float* pv = hi.vec(parent);
                const float* ov = hi.vec(nbid);
                float tip = 0.0f;
                for (size_t d = 0; d < D; ++d) tip += (q[d] - pv[d]) * (ov[d] - pv[d]);
                tip /= nop;
                ps_ipc.push_back(ipc); ps_ipq.push_back(ipq); truths.push_back(tip);
                cs.push_back({nop, ipc, ipq, dqp, nbid, qi});
            }
        };
        for (size_t i = 0; i < n_db; ++i) {
            qbuf.emplace_back(hi.vec(sid[i]), hi.vec(sid[i]) + D);
            process(qbuf.back().data(), qbuf.size() - 1);
        }
        std::normal_distribution<float> nd(0.0f, 1.0f);
        for (size_t i = 0; i < n_synth; ++i) {
            const float* base = hi.vec(sid[i % n]);
            std::vector<float> sq(D);
            for (size_t d = 0; d < D; ++d) sq[d] = base[d] + nd(rng) * std::sqrt(dim_var[d]);
            qbuf.push_back(std::move(sq));
            process(qbuf.back().data(), qbuf.size() - 1);
        }
        if (ipqo_vals.empty()) throw std::runtime_error("Calibration failed: no ip_qo samples.");
        std::sort(ipqo_vals.begin(), ipqo_vals.end());
        {
            const float med = ipqo_vals[ipqo_vals.size() / 2];
            std::vector<float> ad(ipqo_vals.size());
            for (size_t i = 0; i < ad.size(); ++i) ad[i] = std::fabs(ipqo_vals[i] - med);
            std::sort(ad.begin(), ad.end());
            cal.ip_qo_floor = std::max(med - 3.0f * 1.4826f * ad[ad.size() / 2], kEpsMedium);
        }
        std::vector<float> fe(ps_ipc.size());
        for (size_t i = 0; i < fe.size(); ++i) fe[i] = ps_ipc[i] / std::max(ps_ipq[i], cal.ip_qo_floor);
        if (fe.size() < 20) throw std::runtime_error("Calibration failed: too few estimator/target pairs.");
        const size_t np = fe.size();
        double se = 0, stt = 0, see = 0, set_ = 0;
        for (size_t i = 0; i < np; ++i) { double e = fe[i], t = truths[i]; se += e; stt += t; see += e * e; set_ += e * t; }
        const double me = se / np, mt = stt / np, ve = see / np - me * me, cov = set_ / np - me * mt;
        double a = 1.0, b = 0.0;
        if (ve > kEpsSmall) { a = cov / ve; b = mt - a * me; }
        std::vector<float> ar(np);
        for (int iter = 0; iter < 10; ++iter) {  // Huber IRLS, :946-985
            for (size_t i = 0; i < np; ++i) ar[i] = std::fabs(truths[i] - static_cast<float>(a * fe[i] + b));
            std::sort(ar.begin(), ar.end());
            const float hd = 1.345f * 1.4826f * ar[np / 2];
            if (hd < kEpsSmall) break;
            double w0 = 0, we = 0, wt = 0, wee = 0, wet = 0;
            for (size_t i = 0; i < np; ++i) {
                const float r = std::fabs(truths[i] - static_cast<float>(a * fe[i] + b));
                const double w = (r <= hd) ? 1.0 : (double)(hd / r);
                const double e = fe[i], t = truths[i];
                w0 += w; we += w * e; wt += w * t; wee += w * e * e; wet += w * e * t;
            }
            const double wme = we / w0, wmt = wt / w0, wv = wee / w0 - wme * wme, wc = wet / w0 - wme * wmt;
            if (wv > kEpsSmall) {
                const double an = wc / wv, bn = wmt - an * wme;
                const bool done = std::fabs(an - a) + std::fabs(bn - b) < 1e-6;
                a = an; b = bn;
                if (done) break;
            }
        }
        double ssr = 0, sst = 0;
        for (size_t i = 0; i < np; ++i) {
            const double res = truths[i] - (a * fe[i] + b);
            ssr += res * res;
            sst += (truths[i] - mt) * (truths[i] - mt);
        }
        const float r2 = sst > kEpsSmall ? static_cast<float>(1.0 - ssr / sst) : 0.0f;
        const double sxx = ve * static_cast<double>(np);
        float max_lev = 0.0f;
        if (sxx > kEpsSmall)
            for (size_t i = 0; i < np; ++i)
                max_lev = std::max(max_lev, static_cast<float>(1.0 / np + (fe[i] - me) * (fe[i] - me) / sxx));
        if (r2 < 0.1f || max_lev > 4.0f / static_cast<float>(std::max(np, size_t(1)))) { a = 1.0; b = 0.0; }
        cal.affine_a = static_cast<float>(a);
        cal.affine_b = static_cast<float>(b);
        std::sort(nn_d.begin(), nn_d.end());
        cal.median_nn_dist_sq = nn_d[nn_d.size() / 2];
        cal.min_slack_sq = std::max(kEpsSmall, cal.median_nn_dist_sq * 1e-4f);
        std::vector<float> resid;
        resid.reserve(cs.size());
        for (const auto& s : cs) {
            const float fq = std::max(s.ipq, cal.ip_qo_floor);
            float ie = fq > kEpsMedium ? s.ipc / fq : 0.0f;
            ie = cal.affine_a * ie + cal.affine_b;
            const float ed = std::max(s.nop * s.nop + s.dqp - 2.0f * s.nop * ie, 0.0f);
            resid.push_back(std::fabs(ed - l2sq(D, qbuf[s.qi].data(), hi.vec(s.nbr))));
        }
        std::sort(resid.begin(), resid.end());
        const size_t nr = resid.size();
        // the reference's min tail (sqrt(n)) is unreachable for n > ~230k; cap it (see header)
        size_t min_tail = prof.evt_min_tail;
        const size_t supply = static_cast<size_t>(std::sqrt(static_cast<double>(std::max(nr, size_t(4)))));
        if (min_tail > supply / 2) min_tail = std::max<size_t>(64, supply / 2);
        const float tmin = std::max(1.0f - 1.0f / std::sqrt(static_cast<float>(std::max(nr, size_t(4)))), 0.5f);
        const float tmax = 1.0f - static_cast<float>(min_tail) / static_cast<float>(std::max(nr, size_t(1)));
        cal.evt = fit_gpd_stable(resid.data(), nr, min_tail, tmin, tmax);
        std::sort(nops.begin(), nops.end());
        cal.median_nop = nops[nops.size() / 2];
        if (!cal.evt.fitted || cal.median_nop <= 0.0f)
            throw std::runtime_error("Calibration failed: EVT-CRC fit did not converge.");
        const float ref = std::sqrt(std::max(cal.median_nn_dist_sq, cal.min_slack_sq));
        const float q1 = resid[nr / 4] / ref, med = resid[nr / 2] / ref, q3 = resid[3 * nr / 4] / ref, iqr = q3 - q1;
        cal.gamma_min = std::max(1.0f + resid[std::max(size_t(1), nr / 100)] / ref, 1.0f + 1.0f / std::sqrt(static_cast<float>(D)));
        cal.gamma_max = std::max(1.0f + q3 + 1.5f * iqr, cal.gamma_min + std::max(iqr, med));
        double rm = 0; for (float r : resid) rm += r; rm /= nr;
        double rv = 0; for (float r : resid) rv += (r - rm) * (r - rm); rv /= nr;
        const float cv = static_cast<float>(std::sqrt(rv) / std::max(rm, (double)kEpsSmall));
        cal.gamma_beta = 1.0f / std::max(cv, 1.0f / std::sqrt(2.0f * static_cast<float>(std::max(nr, size_t(2)) - 1)));
        cal.gamma_warmup = std::max(size_t(4), static_cast<size_t>(std::ceil(std::sqrt(static_cast<float>(cal.evt.n_tail)))));
        cal.slack_levels = prof.slack_levels;
        const int L = std::clamp(cal.slack_levels, 1, 32);
        cal.search_num_slack_levels = L;
        const float basel = 6.0f / (3.14159265358979f * 3.14159265358979f);
        for (int i = 1; i <= L; ++i)
            cal.search_ip_slack_levels[i - 1] =
                evt_quantile(0.5e-4f * basel / (static_cast<float>(i) * static_cast<float>(i)), cal.evt) / (2.0f * cal.median_nop);
        cal.search_gamma = std::clamp(1.0f + evt_quantile(0.5e-4f, cal.evt) / ref, cal.gamma_min, cal.gamma_max);
    }
    std::memcpy(hi.calib, &cal, 248);
    std::memcpy(hi.profile, &prof, 72);
    note("calibration");
    hi.validate();
}

}  // namespace build
}  // namespace cph
