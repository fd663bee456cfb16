// builder_host.h -- the host-only parts of index construction: robust statistics, the
// concurrent upper-layer insertion, the extreme-value tail fit and the file-format records of the calibration.
// No HIP here: this header also compiles with plain g++, which is how tests/host_san runs it under the address,
// undefined-behaviour and thread sanitizers.  Orchestration and everything that touches n x D data: builder.h,
// builder_pipeline.h.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <vector>

#include "cph_core.h"
#include "host_index.h"
#include "host_parallel.h"

namespace cph {
namespace build {

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

    // `stamp` has one entry per level-1 member (every vertex an upper-layer search can meet is one)
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
            stamp[slot_of[ep]] = epoch;
            best.push({dist(v, ep), ep});
            open.push(best.top());
            while (!open.empty()) {
                const Near cur = open.top();
                open.pop();
                if (best.size() >= ef && cur.d > best.top().d) break;
                const size_t d = copy_list(l, cur.id, nb.data());
                for (size_t i = 0; i < d; ++i) {
                    const uint32_t w = nb[i];
                    if (w == v || stamp[slot_of[w]] == epoch) continue;
                    stamp[slot_of[w]] = epoch;
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
        const size_t n_upper = lv[0].members.size();
        {
            std::vector<uint32_t> stamp(n_upper, 0);
            uint32_t epoch = 0;
            for (size_t i = 1; i < seed; ++i) insert(order[i], stamp, epoch);       // order[0] = entry: nothing to link yet
        }
        if (order.size() > seed) {
            std::atomic<size_t> next{seed};
            run_threads(host_threads(), [&](size_t) {
                std::vector<uint32_t> stamp(n_upper, 0);
                uint32_t epoch = 0;
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= order.size()) break;
                    insert(order[i], stamp, epoch);
                }
            });
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
