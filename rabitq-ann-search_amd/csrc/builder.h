// builder.h — index construction: build() + finalize() for the drop-in API.
//
// Host-side restatement of the reference's construction pipeline with the O(n²) part moved to
// the GPU.  Parity with the reference here is *statistical*, not bit-level: the reference's own
// build depends on the OpenMP thread count.  What is kept exactly are the formulas
// that define the file contents the query path consumes — the per-edge codes and aux values
// (encoder/rabitq_encoder.hpp:138-181, 287-323, 371-467), the neighbour selection rule
// (graph/neighbor_selection.hpp:21-88), the layer assignment and upper-layer construction
// (api/hnsw_index.hpp:484-615, 640-716), the BFS reorder (graph/rabitq_graph.hpp:208-278) and
// the estimator calibration (api/hnsw_index.hpp:718-1139, core/evt_crc.hpp:34-354).
//
// Deliberate differences:
//  * the working 32-NN lists come from an exact brute-force kNN on the GPU (device_knn.h)
//    instead of NNDescent (graph_refinement.hpp:71-263,455-515);
//  * the reference cannot calibrate any index with n > ~230k: its EVT tail fit needs
//    sqrt(n) exceedances but only ever gets sqrt(480*sqrt(n)) of them (hnsw_index.hpp:1046-1056,
//    adaptive_defaults.hpp:45-46) and throws "EVT-CRC fit did not converge"; here the minimum
//    tail size is capped at half of what the residual sample can supply, so SIFT1M-class
//    indexes can be built.
#pragma once
#include <algorithm>
#include <atomic>
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
#include "host_index.h"

namespace cph {
namespace build {

struct Cand {
    uint32_t id;
    float dist;
    bool operator<(const Cand& o) const { return dist < o.dist; }
};

inline void parallel_for(size_t n, size_t min_chunk, const std::function<void(size_t, size_t)>& fn) {
    // default: at most 64 host threads per process (one process per GPU, eight per node)
    unsigned hw = std::min(64u, std::thread::hardware_concurrency());
    size_t nt = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n / std::max<size_t>(min_chunk, 1)));
    if (const char* e = getenv("CPH_BUILD_THREADS")) nt = std::max(1, atoi(e));
    if (nt <= 1) { fn(0, n); return; }
    // dynamic chunks: per-node cost varies
    std::atomic<size_t> next{0};
    const size_t chunk = std::max<size_t>(min_chunk, n / (nt * 16));
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (;;) {
                size_t lo = next.fetch_add(chunk);
                if (lo >= n) break;
                fn(lo, std::min(n, lo + chunk));
            }
        });
    for (auto& x : th) x.join();
}

// l2_distance_simd summation order (core/memory.hpp:65-79)
inline float l2sq(size_t D, const float* a, const float* b) {
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < D; i += 8)
        for (int j = 0; j < 8; ++j) {
            float d = a[i + j] - b[i + j];
            c[j] = std::fmaf(d, d, c[j]);
        }
    float s0 = c[0] + c[4], s1 = c[1] + c[5], s2 = c[2] + c[6], s3 = c[3] + c[7];
    return (s0 + s1) + (s2 + s3);
}

// graph/neighbor_selection.hpp:21-88
template <class DistFn, class ErrFn>
std::vector<Cand> select_alpha_cng(std::vector<Cand> c, size_t R, DistFn dist_fn, ErrFn err_fn,
                                   float alpha, float tau, float alpha_max = 0.0f) {
    std::sort(c.begin(), c.end(), [](const Cand& a, const Cand& b) {
        return a.id < b.id || (a.id == b.id && a.dist < b.dist);
    });
    c.erase(std::unique(c.begin(), c.end(), [](const Cand& a, const Cand& b) { return a.id == b.id; }),
            c.end());
    std::sort(c.begin(), c.end());
    if (c.size() <= R) return c;
    if (alpha_max <= 0.0f) alpha_max = 2.0f * alpha;
    float local_alpha = alpha * std::sqrt(static_cast<float>(c.size()) / static_cast<float>(R));
    local_alpha = std::clamp(local_alpha, 1.0f, alpha_max);
    std::vector<Cand> sel;
    sel.reserve(R);
    for (size_t i = 0; i < c.size() && sel.size() < R; ++i) {
        bool add = true;
        const float err_c = err_fn(c[i].id);
        const float dist_cq = c[i].dist;
        for (const auto& ex : sel) {
            const float dist_ce = dist_fn(c[i].id, ex.id);
            const float margin = err_c + err_fn(ex.id);
            const float thr = local_alpha * dist_cq + margin - (local_alpha - 1.0f) * tau;
            if (dist_ce < thr) { add = false; break; }
        }
        if (add) sel.push_back(c[i]);
    }
    if (sel.size() < R) {
        for (size_t i = 0; i < c.size() && sel.size() < R; ++i) {
            bool already = false;
            for (const auto& s : sel)
                if (s.id == c[i].id) { already = true; break; }
            if (!already) sel.push_back(c[i]);
        }
    }
    return sel;
}

// ---- data-side encoder ------------------------------------------------------------------
struct EdgeCode {
    std::vector<uint8_t> u;     // code value per dimension, 0 .. 2^BW-1   [D]
    float nop = 0, ip_qo = 0, ip_cp = 0;
    uint32_t msb_pop = 0, weighted_pop = 0;
    bool degenerate = false;    // nop < norm_epsilon: all-zero code, zero aux
};

struct DataEncoder {
    const Rotation* rot;
    size_t D, dim, bw;
    float norm_factor, inv_sqrt_d;

    void init(const Rotation* r, size_t D_, size_t dim_, size_t bw_) {
        rot = r; D = D_; dim = dim_; bw = bw_;
        const float d = static_cast<float>(D);
        norm_factor = 1.0f / (d * std::sqrt(d));
        inv_sqrt_d = 1.0f / std::sqrt(d);
    }
    void rotate_scaled(const float* padded, float* out) const {  // rotate_raw_vector, :81-86
        std::memcpy(out, padded, D * sizeof(float));
        rot->apply(out);
        for (size_t i = 0; i < D; ++i) out[i] *= norm_factor;
    }
    // caq_quantize (:371-467) for BW >= 2, sign quantisation (:171-178) for BW == 1
    void quantize(const float* rotated, const float* rotated_parent, EdgeCode& e,
                  std::vector<int>& codes) const {
        e.u.assign(D, 0);
        e.msb_pop = e.weighted_pop = 0;
        if (bw == 1) {
            float l1 = 0.0f, ipcp = 0.0f;
            for (size_t i = 0; i < D; ++i) {
                const bool pos = rotated[i] >= 0.0f;
                e.u[i] = pos ? 1 : 0;
                l1 += std::fabs(rotated[i]);
                if (rotated_parent) ipcp += (pos ? 1.0f : -1.0f) * rotated_parent[i];
                e.msb_pop += pos ? 1u : 0u;
            }
            e.weighted_pop = e.msb_pop;
            e.ip_qo = l1 * inv_sqrt_d;
            e.ip_cp = ipcp * inv_sqrt_d;
            return;
        }
        const int Ki = (1 << bw) - 1;
        const float K = static_cast<float>(Ki);
        float mn = rotated[0], mx = rotated[0];
        for (size_t i = 1; i < D; ++i) {
            if (rotated[i] < mn) mn = rotated[i];
            if (rotated[i] > mx) mx = rotated[i];
        }
        float delta = (mx - mn) / K;
        const float ceps = 1e-10f / static_cast<float>(D);  // coordinate_epsilon
        if (delta < ceps) delta = ceps;
        const float inv_delta = 1.0f / delta;
        codes.resize(D);
        float dot_co = 0.0f, norm_c = 0.0f;
        for (size_t i = 0; i < D; ++i) {
            int u = static_cast<int>((rotated[i] - mn) * inv_delta + 0.5f);
            u = u < 0 ? 0 : (u > Ki ? Ki : u);
            codes[i] = u;
            const float c = (2.0f * u - K) / K;
            dot_co += c * rotated[i];
            norm_c += c * c;
        }
        float prev_cos = 0.0f;
        for (size_t iter = 0; iter < 10; ++iter) {
            bool changed = false;
            for (size_t i = 0; i < D; ++i) {
                const int old_u = codes[i];
                const float old_c = (2.0f * old_u - K) / K;
                const float dot_wo = dot_co - old_c * rotated[i];
                const float norm_wo = norm_c - old_c * old_c;
                int best_u = old_u;
                float best_dot = dot_co, best_norm = norm_c;
                auto consider = [&](int ut) {
                    const float c = (2.0f * ut - K) / K;
                    const float nd = dot_wo + c * rotated[i];
                    const float nn = norm_wo + c * c;
                    if (nd * nd * best_norm > best_dot * best_dot * nn) { best_u = ut; best_dot = nd; best_norm = nn; }
                };
                if (bw >= 4) {
                    if (old_u - 1 >= 0) consider(old_u - 1);
                    if (old_u + 1 <= Ki) consider(old_u + 1);
                } else {
                    for (int ut = 0; ut <= Ki; ++ut)
                        if (ut != old_u) consider(ut);
                }
                if (best_u != old_u) {
                    const float nc = (2.0f * best_u - K) / K;
                    dot_co = dot_wo + nc * rotated[i];
                    norm_c = norm_wo + nc * nc;
                    codes[i] = best_u;
                    changed = true;
                }
            }
            if (!changed) break;
            const float cos_sq = norm_c > 0.0f ? dot_co * dot_co / norm_c : 0.0f;
            if (iter > 0 && (cos_sq - prev_cos) < 1e-4f) break;
            prev_cos = cos_sq;
        }
        float ipqo = 0.0f, ipcp = 0.0f;
        for (size_t i = 0; i < D; ++i) {
            const int u = codes[i];
            e.u[i] = static_cast<uint8_t>(u);
            const float c = (2.0f * u - K) / K;
            ipqo += c * rotated[i];
            if (rotated_parent) ipcp += c * rotated_parent[i];
            e.weighted_pop += static_cast<uint32_t>(u);
            e.msb_pop += static_cast<uint32_t>((u >> (bw - 1)) & 1);
        }
        e.ip_qo = ipqo * inv_sqrt_d;
        e.ip_cp = ipcp * inv_sqrt_d;
    }
    // compute_neighbor_aux / compute_neighbor_aux_nbit (:138-181, :287-323)
    void encode_edge(const float* parent, const float* nbr, const float* rotated_parent, EdgeCode& e,
                     std::vector<float>& tmp, std::vector<int>& codes) const {
        tmp.resize(2 * D);
        float* diff = tmp.data();
        float* rotated = tmp.data() + D;
        float nsq = 0.0f;
        for (size_t i = 0; i < dim; ++i) { diff[i] = nbr[i] - parent[i]; nsq += diff[i] * diff[i]; }
        for (size_t i = dim; i < D; ++i) diff[i] = 0.0f;
        const float nop = std::sqrt(nsq);
        e.nop = nop;
        e.degenerate = nop < 1e-8f / static_cast<float>(D);  // norm_epsilon
        if (e.degenerate) {
            e.u.assign(D, 0);
            e.ip_qo = e.ip_cp = 0.0f;
            e.msb_pop = e.weighted_pop = 0;
            return;
        }
        const float inv = 1.0f / nop;
        for (size_t i = 0; i < D; ++i) diff[i] *= inv;
        rotate_scaled(diff, rotated);
        quantize(rotated, rotated_parent, e, codes);
    }
};

// writes neighbour `slot` of a reference-layout neighbour block
inline void write_slot(uint8_t* nb, const RefLayout& RL, size_t D, size_t bw, uint32_t slot, uint32_t id,
                       const EdgeCode& e) {
    const size_t plane_stride = round_up(RL.plane_bytes, 64);
    const size_t bytes_nb = (D + 7) / 8;
    for (size_t b = 0; b < bw; ++b) {
        uint8_t* plane = nb + RL.codes + b * plane_stride;
        for (size_t sp = 0; sp < bytes_nb; ++sp) {
            uint8_t v = 0;
            for (size_t t = 0; t < 8 && 8 * sp + t < D; ++t)
                if ((e.u[8 * sp + t] >> (bw - 1 - b)) & 1) v |= (uint8_t)(1u << t);
            plane[sp * 32 + slot] = v;
        }
    }
    std::memcpy(nb + RL.nop + 4 * slot, &e.nop, 4);
    std::memcpy(nb + RL.ip_qo + 4 * slot, &e.ip_qo, 4);
    std::memcpy(nb + RL.ip_cp + 4 * slot, &e.ip_cp, 4);
    const uint16_t p = (uint16_t)e.msb_pop, wp = (uint16_t)e.weighted_pop;
    std::memcpy(nb + RL.pop + 2 * slot, &p, 2);
    if (bw > 1) std::memcpy(nb + RL.wpop + 2 * slot, &wp, 2);
    std::memcpy(nb + RL.ids + 4 * slot, &id, 4);
}

// ---- host FastScan of one reference-layout block (calibration only) -----------------------
inline void host_block_sums(const uint8_t* nb, const RefLayout& RL, size_t D, size_t bw,
                            const uint8_t* qu, uint32_t* out /*[32] weighted N-bit (or 1-bit) sum*/) {
    const size_t plane_stride = round_up(RL.plane_bytes, 64);
    for (int i = 0; i < 32; ++i) out[i] = 0;
    for (size_t b = 0; b < bw; ++b) {
        const uint8_t* plane = nb + RL.codes + b * plane_stride;
        const uint32_t w = 1u << (bw - 1 - b);
        for (size_t sp = 0; sp < (D + 7) / 8; ++sp)
            for (int i = 0; i < 32; ++i) {
                const uint8_t c = plane[sp * 32 + i];
                uint32_t s = 0;
                for (int t = 0; t < 8; ++t)
                    if ((c >> t) & 1) s += qu[8 * sp + t];
                out[i] += w * s;
            }
    }
}

// ---- EVT / GPD (core/evt_crc.hpp) ------------------------------------------------------------
struct EVTState {  // layout = reference EVTState (56 bytes)
    float u = 0, p_u = 0, xi = 0, beta = 0;
    uint32_t n_tail = 0;
    bool fitted = false, use_empirical = false;
    float empirical[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
static_assert(sizeof(EVTState) == 56, "EVTState layout");
constexpr float kCheckpointAlphas[8] = {0.5f, 0.1f, 0.05f, 0.01f, 0.005f, 0.001f, 5e-4f, 1e-4f};

inline float evt_quantile(float alpha, const EVTState& e) {  // evt_crc.hpp:34-71
    alpha = std::clamp(alpha, 1e-12f, 0.5f);
    if (alpha >= e.p_u) return e.u;
    if (e.use_empirical) {
        const float* A = kCheckpointAlphas;
        const float* Q = e.empirical;
        for (int j = 0; j < 7; ++j)
            if (alpha >= A[j + 1]) {
                const float t = (alpha - A[j + 1]) / (A[j] - A[j + 1]);
                return Q[j + 1] * (1.0f - t) + Q[j] * t;
            }
        const float lr = std::log(A[6] / A[7]);
        const float slope = lr > kEpsSmall ? (Q[7] - Q[6]) / lr : 0.0f;
        return Q[7] + slope * std::log(A[7] / alpha);
    }
    const float ratio = e.p_u / alpha;
    if (std::fabs(e.xi) < 1e-6f) return e.u + e.beta * std::log(ratio);
    return e.u + (e.beta / e.xi) * (std::pow(ratio, e.xi) - 1.0f);
}

inline EVTState fit_gpd(const float* r, size_t n, float thr_q, size_t min_tail) {  // :74-188
    EVTState st;
    if (n < min_tail * 2) return st;
    size_t u_idx = std::min(static_cast<size_t>(static_cast<float>(n) * thr_q), n - 1);
    st.u = r[u_idx];
    std::vector<double> y;
    y.reserve(n - u_idx);
    for (size_t i = u_idx + 1; i < n; ++i) {
        const double yi = r[i] - st.u;
        if (yi > 0.0) y.push_back(yi);
    }
    const uint32_t m = static_cast<uint32_t>(y.size());
    st.n_tail = m;
    st.p_u = static_cast<float>(m) / static_cast<float>(n);
    if (m < min_tail) return st;
    double sum_y = 0, sum_y2 = 0;
    for (double v : y) { sum_y += v; sum_y2 += v * v; }
    const double mean_y = sum_y / m, var_y = sum_y2 / m - mean_y * mean_y;
    double xi_mom, beta_mom;
    if (var_y < kEpsTiny) { xi_mom = 0.0; beta_mom = std::max(mean_y, 1e-8); }
    else { xi_mom = 0.5 * (1.0 - mean_y * mean_y / var_y); beta_mom = mean_y * (1.0 - xi_mom); }
    double xi = xi_mom, beta = std::max(beta_mom, 1e-8);
    bool conv = false;
    for (int iter = 0; iter < 50; ++iter) {
        if (std::fabs(xi) < 1e-6) { beta = mean_y; xi = 0.0; conv = true; break; }
        bool feas = true;
        for (double v : y) if (1.0 + xi * v / beta <= 0.0) { feas = false; break; }
        if (!feas) break;
        double bn = beta;
        for (int j = 0; j < 5; ++j) {
            double s = 0;
            bool ok = true;
            for (double v : y) {
                const double z = 1.0 + xi * v / bn;
                if (z <= 0.0) { ok = false; break; }
                s += v / z;
            }
            if (!ok) break;
            bn = std::max((1.0 + xi) * s / m, 1e-8);
        }
        beta = bn;
        double score = 0, info = 0;
        for (double v : y) {
            const double z = 1.0 + xi * v / beta;
            if (z <= 0.0) { score = 0; break; }
            const double lz = std::log(z), w = v / (beta * z);
            score += -lz / (xi * xi) + (1.0 + 1.0 / xi) * w;
            info += 2.0 * lz / (xi * xi * xi) - 2.0 * w / (xi * xi) - (1.0 + 1.0 / xi) * w * w;
        }
        if (std::fabs(info) < kEpsTiny) break;
        double xn = std::min(std::max(xi - score / info, -0.2), 0.5);
        if (std::fabs(xn - xi) < 1e-6) { xi = xn; conv = true; break; }
        xi = xn;
    }
    if (!conv) { xi = xi_mom; beta = beta_mom; }
    st.xi = std::clamp(static_cast<float>(xi), -0.2f, 0.5f);
    st.beta = std::max(static_cast<float>(beta), 1e-8f);
    st.fitted = true;
    return st;
}

inline EVTState fit_gpd_stable(const float* r, size_t n, size_t min_tail, float thr_min, float thr_max) {
    if (thr_max <= thr_min) return fit_gpd(r, n, thr_min, min_tail);   // :229-232
    const size_t nt = std::clamp(static_cast<size_t>(std::ceil(std::sqrt(std::log2(std::max((float)n, 64.0f))))),
                                 size_t(3), size_t(8));
    float thr[8];
    EVTState fits[8];
    bool valid[8] = {};
    size_t nv = 0;
    for (size_t t = 0; t < nt; ++t) {
        thr[t] = thr_min + (thr_max - thr_min) * static_cast<float>(t) / static_cast<float>(nt - 1);
        fits[t] = fit_gpd(r, n, thr[t], min_tail);
        if (fits[t].fitted) { valid[t] = true; ++nv; }
    }
    if (nv < 2) {
        for (size_t t = 0; t < nt; ++t) if (valid[t]) return fits[t];
        return EVTState{};
    }
    auto diff = [&](size_t a, size_t b) {
        const float dxi = fits[a].xi - fits[b].xi;
        const float bavg = 0.5f * (fits[a].beta + fits[b].beta);
        const float db = (fits[a].beta - fits[b].beta) / std::max(bavg, 1e-8f);
        return dxi * dxi + db * db;
    };
    size_t best = 0;
    float best_score = 3.402823466e+38f;
    bool found = false;
    for (size_t t = 0; t < nt; ++t) {
        if (!valid[t]) continue;
        float score = 0;
        int nbrs = 0;
        for (size_t p = t; p > 0; --p) if (valid[p - 1]) { score += diff(t, p - 1); ++nbrs; break; }
        for (size_t x = t + 1; x < nt; ++x) if (valid[x]) { score += diff(t, x); ++nbrs; break; }
        if (nbrs > 0) {
            score /= static_cast<float>(nbrs);
            if (score < best_score) { best_score = score; best = t; found = true; }
        }
    }
    if (!found) {
        for (size_t t = 0; t < nt; ++t) if (valid[t]) { best = t; found = true; break; }
        if (!found) return EVTState{};
    }
    EVTState& b = fits[best];
    if (b.fitted && b.n_tail >= 20) {  // KS validation, :321-351
        size_t u_idx = std::min(static_cast<size_t>(static_cast<float>(n) * thr[best]), n - 1);
        std::vector<float> tail;
        for (size_t i = u_idx + 1; i < n; ++i) {
            const float yi = r[i] - b.u;
            if (yi > 0.0f) tail.push_back(yi);
        }
        std::sort(tail.begin(), tail.end());
        if (!tail.empty()) {
            float ks = 0;
            for (size_t i = 0; i < tail.size(); ++i) {
                const float Fe = static_cast<float>(i + 1) / static_cast<float>(tail.size());
                float Fg;
                if (std::fabs(b.xi) < 1e-6f) Fg = 1.0f - std::exp(-tail[i] / b.beta);
                else {
                    const float z = 1.0f + b.xi * tail[i] / b.beta;
                    Fg = z > 0.0f ? 1.0f - std::pow(z, -1.0f / b.xi) : 1.0f;
                }
                ks = std::max(ks, std::fabs(Fe - Fg));
            }
            const float crit = 1.25f * 1.358f / std::sqrt(static_cast<float>(tail.size()));
            if (ks > crit) {
                b.use_empirical = true;
                for (int j = 0; j < 8; ++j) {
                    float tq = std::clamp(1.0f - kCheckpointAlphas[j] / b.p_u, 0.0f, 1.0f);
                    size_t idx = std::min(static_cast<size_t>(tq * static_cast<float>(tail.size())), tail.size() - 1);
                    b.empirical[j] = b.u + tail[idx];
                }
            }
        }
    }
    return fits[best];
}

// CalibrationSnapshot / IndexProfile as the reference lays them out (checked by static_assert)
struct CalibrationSnapshot {
    float affine_a, affine_b, ip_qo_floor, median_nn_dist_sq, min_slack_sq, median_nop;
    EVTState evt;
    float gamma_min, gamma_max, gamma_beta;
    size_t gamma_warmup;
    int slack_levels;
    float search_ip_slack_levels[32];
    int search_num_slack_levels;
    float search_gamma;
};
static_assert(sizeof(CalibrationSnapshot) == 248, "CalibrationSnapshot layout");
struct GraphStats { float avg_degree, alpha, tau, alpha_max; };
struct IndexProfile {
    size_t n = 0, D = 0, R = 0, bits = 0, evt_min_tail = 0, min_calib_samples = 0;
    int slack_levels = 0;
    GraphStats graph_stats{};
};
static_assert(sizeof(IndexProfile) == 72, "IndexProfile layout");

}  // namespace build
}  // namespace cph
