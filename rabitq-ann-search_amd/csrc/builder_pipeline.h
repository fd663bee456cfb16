// builder_pipeline.h — finalize(): the construction pipeline of builder.h, stage by stage, and the
// calibration that follows it.  See builder.h for the design and the reference citations.
#pragma once
#include <exception>
#include <thread>
#include <type_traits>
#include "builder.h"

namespace cph {
namespace build {

struct BuiltDevice {            // the searchable index as the pipeline leaves it in HBM
    DevBuf<uint8_t> blocks;     // [n][stride] device blocks
    DevBuf<float> raw;          // [n][D] vectors, final order
    DevBuf<float> norm;         // [n]
    std::vector<uint8_t> own_host;   // [n][nb_off]: every vertex' own code header (reference layout), for save()
};

inline size_t isqrt_sz(size_t n) { return (size_t)std::floor(std::sqrt((double)n)); }

template <int BW, bool G>
inline void launch_encode(const EncodeArgsB& a, uint32_t grid, size_t lds) {
    if (lds > 48 * 1024)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_edges_kernel<BW, G>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((encode_edges_kernel<BW, G>), dim3(grid), dim3(64), lds, nullptr, a);
    HIP_CHECK(hipGetLastError());
}
// Edge (a.nbr set) or own-code (a.nbr null) encoding of `units` workgroup units.  Up to D = 256 all 32 edges of a vertex
// fit the LDS budget of a workgroup; beyond that the rows live in a transposed HBM scratch (device_build.h, LaneRow):
// 32 live lanes per wave instead of 8, and 12 workgroups per CU instead of 3.
inline void run_encode(size_t bw, EncodeArgsB a, uint64_t units, int num_cus) {
    a.epb = encode_edges_epb(a.D);
    const bool global_rows = a.epb < 32;
    DevBuf<float> scratch_f;
    DevBuf<uint8_t> scratch_u;
    size_t lds;
    uint32_t per_cu;
    if (global_rows) {
        lds = encode_edges_lds_global(a.D);
        per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(12, (150 * 1024) / lds));
    } else {
        lds = encode_edges_lds(a.D, a.epb);
        per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (150 * 1024) / lds));
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(units, (uint64_t)num_cus * per_cu);
    if (global_rows) {
        scratch_f.alloc((size_t)grid * a.D * 32);
        scratch_u.alloc((size_t)grid * a.D * 32);
        a.scratch_f = scratch_f.p;
        a.scratch_u = scratch_u.p;
    }
    auto go = [&](auto bw_t) {
        constexpr int BW = decltype(bw_t)::value;
        if (global_rows) launch_encode<BW, true>(a, grid, lds);
        else launch_encode<BW, false>(a, grid, lds);
    };
    if (bw == 1) go(std::integral_constant<int, 1>());
    else if (bw == 2) go(std::integral_constant<int, 2>());
    else go(std::integral_constant<int, 4>());
    HIP_CHECK(hipDeviceSynchronize());     // (the scratch is released on return)
}

// Graph, codes, upper layers.  `vecs` = n x dim input rows (host).  Fills `hi` (everything but the
// calibration record) and `dev`.
inline void build_graph(HostIndex& hi, BuiltDevice& dev, const float* vecs, size_t n, size_t dim, size_t D, size_t bw,
                        int num_cus, bool verbose) {
    StageTimer tm{verbose};
    const uint32_t R = 32;
    hi = HostIndex();
    hi.D = D; hi.bw = bw; hi.dim = dim; hi.n = n; hi.seed = 42;
    hi.RL = make_ref_layout(D, bw);
    hi.rot.init(D, 42);
    const DevLayout L = make_dev_layout((uint32_t)D, (uint32_t)bw);
    const size_t M_UPPER = R / 2 + std::min(isqrt_sz(D) / 4, (size_t)R / 4);      // api/hnsw_index.hpp:85
    hi.mL = 1.0 / std::log((double)M_UPPER);
    const size_t Dk = (D + kKnnKC - 1) / kKnnKC * kKnnKC;       // the kNN kernel's K stage (D = 16 -> 32)

    // ---- vectors to HBM, norms, centroid ---------------------------------------------------------
    DevBuf<float> d_x(n * D), d_norm(n);
    upload_padded(d_x.p, D, vecs, dim, n);
    hipLaunchKernelGGL(row_norms_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, d_x.p, (uint64_t)n, (uint32_t)D,
                       (uint32_t)dim, d_norm.p);
    HIP_CHECK(hipGetLastError());
    std::vector<float> centroid(dim);
    DevBuf<float> d_centroid(D);
    {
        DevBuf<double> d_cs(dim);
        HIP_CHECK(hipMemset(d_cs.p, 0, dim * 8));
        hipLaunchKernelGGL(column_sums_kernel, dim3((uint32_t)std::min<size_t>(1024, std::max<size_t>(1, n / 256))), dim3(256), 0,
                           nullptr, d_x.p, (uint64_t)n, (uint32_t)D, (uint32_t)dim, d_cs.p);
        HIP_CHECK(hipGetLastError());
        std::vector<double> cs(dim);
        HIP_CHECK(hipMemcpy(cs.data(), d_cs.p, dim * 8, hipMemcpyDeviceToHost));
        std::vector<float> padded(D, 0.0f);
        for (size_t j = 0; j < dim; ++j) padded[j] = centroid[j] = (float)(cs[j] / (double)n);
        HIP_CHECK(hipMemcpy(d_centroid.p, padded.data(), D * 4, hipMemcpyHostToDevice));
    }
    tm.lap("upload, norms, centroid");

    // ---- layer of every vertex: floor(-ln(U) mL), one mt19937_64(42) stream (api/hnsw_index.hpp:484-503) ----
    std::vector<int32_t> levels(n);
    int max_level = 0;
    uint32_t entry_old = kInvalidNode;
    {
        std::mt19937_64 rng(42);
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        for (size_t i = 0; i < n; ++i) {
            const double r = std::max(uni(rng), 1e-15);
            const int lv = (int)(-std::log(r) * hi.mL);
            levels[i] = lv;
            if (entry_old == kInvalidNode || lv > max_level) { max_level = lv; entry_old = (uint32_t)i; }
        }
    }

    // ---- upper layers: concurrent incremental insertion on the host (5 % of the vertices).  They need the vectors and
    // the levels only, so the host threads build them while the GPU computes the kNN lists below ------------------
    UpperLayers ul(vecs, dim, n, levels, max_level, entry_old, M_UPPER, R);
    struct Background {                      // joins on every path out of this function
        std::thread th;
        std::exception_ptr err;
        ~Background() { if (th.joinable()) th.join(); }
        void wait() { if (th.joinable()) th.join(); if (err) std::rethrow_exception(err); }
    } upper_job;
    upper_job.th = std::thread([&ul, &upper_job] {
        try { ul.build(); } catch (...) { upper_job.err = std::current_exception(); }
    });

    // ---- exact 32-NN lists on the matrix cores ----------------------------------------------------------
    DevBuf<uint32_t> d_knn(n * kKnnK);
    {
        DevBuf<float> d_kd(n * kKnnK);
        if (Dk == D) {
            knn_self_device(d_x.p, d_norm.p, n, D, num_cus, d_knn.p, d_kd.p, verbose);
        } else {
            DevBuf<float> d_xk(n * Dk);
            HIP_CHECK(hipMemset(d_xk.p, 0, n * Dk * 4));
            HIP_CHECK(hipMemcpy2D(d_xk.p, Dk * 4, d_x.p, D * 4, D * 4, n, hipMemcpyDeviceToDevice));
            knn_self_device(d_xk.p, d_norm.p, n, Dk, num_cus, d_knn.p, d_kd.p, verbose);
        }
    }
    tm.lap("exact 32-NN (MFMA)");

    // ---- pruning parameters from a sample of the lists (graph_refinement.hpp:266-383's statistics) ---------
    GraphStatsRecord gs{};
    {
        const size_t sample = std::max<size_t>(1, std::min(isqrt_sz(n), n));
        std::mt19937 rng(43);
        std::vector<size_t> pick(n);
        std::iota(pick.begin(), pick.end(), 0);
        std::shuffle(pick.begin(), pick.end(), rng);
        pick.resize(sample);
        auto dist = [&](uint32_t a, uint32_t b) {
            double s = 0;
            for (size_t j = 0; j < dim; ++j) { const double t = (double)vecs[(size_t)a * dim + j] - vecs[(size_t)b * dim + j]; s += t * t; }
            return (float)s;
        };
        const size_t pairs_of = std::clamp((size_t)(2.0 * std::sqrt((double)R)), (size_t)4, (size_t)R);
        std::vector<float> edge_len, first_len, between;
        std::vector<uint32_t> row(kKnnK);
        double deg = 0;
        for (size_t v : pick) {
            HIP_CHECK(hipMemcpy(row.data(), d_knn.p + v * kKnnK, kKnnK * 4, hipMemcpyDeviceToHost));
            size_t c = 0;
            while (c < (size_t)kKnnK && row[c] != kInvalidNode) ++c;
            deg += (double)c;
            for (size_t s = 0; s < c; ++s) edge_len.push_back(dist((uint32_t)v, row[s]));
            if (c) first_len.push_back(edge_len[edge_len.size() - c]);
            const size_t m = std::min(c, pairs_of);
            for (size_t i = 0; i < m; ++i)
                for (size_t j = i + 1; j < m; ++j) between.push_back(dist(row[i], row[j]));
        }
        gs.avg_degree = (float)(deg / (double)pick.size());
        if (edge_len.empty() || between.empty() || first_len.empty()) {
            gs.alpha = 1.0f; gs.tau = 0.0f; gs.alpha_max = 4.0f;
        } else {
            std::sort(edge_len.begin(), edge_len.end());
            std::sort(between.begin(), between.end());
            const float tiny = 1e-8f / (float)D;
            const float med = quantile_sorted(edge_len, 1, 2), q1 = quantile_sorted(edge_len, 1, 4), q3 = quantile_sorted(edge_len, 3, 4);
            double mu = 0, var = 0;
            for (float d : edge_len) mu += d;
            mu /= edge_len.size();
            for (float d : edge_len) var += (d - mu) * (d - mu);
            var /= edge_len.size();
            const float cv = mu > tiny ? (float)(std::sqrt(var) / mu) : 0.2f;
            const float inter = quantile_sorted(between, 1, 4);
            // alpha: typical edge length over the lower-quartile distance BETWEEN a vertex' neighbours; its
            // ceiling from the spread of the edge lengths; tau: robust scale of the nearest-neighbour distances
            gs.alpha = inter < tiny ? 1.0f + cv : med / inter;
            gs.alpha_max = std::min(q1 > tiny ? q3 / q1 : 2.0f, 5.0f);
            gs.alpha = std::clamp(gs.alpha, 1.0f, gs.alpha_max);
            gs.alpha_max = std::max(gs.alpha_max, 2.0f * gs.alpha);
            gs.tau = mad_sigma(first_len, median_of(first_len));
        }
    }

    // ---- layer 0: reverse edges + selection on the GPU --------------------------------------------------
    DevBuf<uint32_t> d_nbr(n * 32), d_cnt(n);
    {
        DevBuf<float> d_err(n);
        const float err_tol = 1.0f / std::sqrt((float)D);
        hipLaunchKernelGGL(centered_norm_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, d_x.p, d_centroid.p, (uint64_t)n,
                           (uint32_t)D, (uint32_t)dim, err_tol, d_err.p);
        HIP_CHECK(hipGetLastError());
        LayerParams lp{R, gs.alpha, gs.tau, gs.alpha_max, d_err.p};
        select_layer(d_x.p, D, d_knn.p, n, nullptr, lp, num_cus, d_nbr.p, d_cnt.p);
    }
    d_knn.release();
    tm.lap("reverse edges + selection");

    // ---- hub and BFS renumbering (host: a queue walk) -----------------------------------------------------
    std::vector<uint32_t> nbr(n * 32), cnt(n);
    HIP_CHECK(hipMemcpy(nbr.data(), d_nbr.p, n * 32 * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(cnt.data(), d_cnt.p, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> old_to_new(n, kInvalidNode), new_to_old(n);
    {
        // hub: among the sqrt(n) vertices nearest the centroid, the one with the most edges
        std::vector<std::pair<double, uint32_t>> near(n);
        parallel_for(n, 4096, [&](size_t lo, size_t hi_) {
            for (size_t i = lo; i < hi_; ++i) {
                double s = 0;
                for (size_t j = 0; j < dim; ++j) { const double t = (double)vecs[i * dim + j] - centroid[j]; s += t * t; }
                near[i] = {s, (uint32_t)i};
            }
        });
        const size_t top = std::min(n, std::max<size_t>(1, isqrt_sz(n)));
        std::partial_sort(near.begin(), near.begin() + top, near.end());
        uint32_t hub = near[0].second;
        for (size_t i = 1; i < top; ++i)
            if (cnt[near[i].second] > cnt[hub]) hub = near[i].second;
        std::vector<uint8_t> seen(n, 0);
        std::vector<uint32_t> queue;
        queue.reserve(n);
        uint32_t next = 0;
        auto walk = [&](uint32_t start) {
            if (seen[start]) return;
            seen[start] = 1;
            queue.push_back(start);
            for (size_t head = queue.size() - 1; head < queue.size(); ++head) {
                const uint32_t v = queue[head];
                old_to_new[v] = next;
                new_to_old[next++] = v;
                for (uint32_t j = 0; j < cnt[v]; ++j) {
                    const uint32_t w = nbr[(size_t)v * 32 + j];
                    if (w < n && !seen[w]) { seen[w] = 1; queue.push_back(w); }
                }
            }
        };
        walk(hub);
        for (size_t i = 0; i < n; ++i) walk((uint32_t)i);
    }
    tm.lap("hub + BFS order");

    // ---- final order in HBM: vectors, norms, neighbour lists ------------------------------------------------
    DevBuf<uint32_t> d_o2n(n), d_n2o(n);
    HIP_CHECK(hipMemcpy(d_o2n.p, old_to_new.data(), n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_n2o.p, new_to_old.data(), n * 4, hipMemcpyHostToDevice));
    dev.raw.alloc(n * D);
    dev.norm.alloc(n);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)std::min<uint64_t>(n, (uint64_t)num_cus * 64)), dim3(64), 0, nullptr,
                       d_x.p, d_n2o.p, (uint64_t)n, (uint32_t)D, dev.raw.p);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, d_norm.p, d_n2o.p, (uint64_t)n, dev.norm.p);
    HIP_CHECK(hipGetLastError());
    DevBuf<uint32_t> d_nbr_new(n * 32);
    hipLaunchKernelGGL(permute_lists_kernel, dim3(grid_for(n * 32, 256)), dim3(256), 0, nullptr, d_nbr.p, d_n2o.p, d_o2n.p,
                       (uint64_t)n, d_nbr_new.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    d_x.release(); d_norm.release(); d_nbr.release(); d_cnt.release();

    // ---- codes: every edge into the device blocks, every vertex' own code for the file ----------------------
    DevBuf<float> d_signs(3 * D);
    HIP_CHECK(hipMemcpy(d_signs.p, hi.rot.signs.data(), 3 * D * 4, hipMemcpyHostToDevice));
    const float df = (float)D;
    const float norm_factor = 1.0f / (df * std::sqrt(df)), inv_sqrt_d = 1.0f / std::sqrt(df);
    dev.blocks.alloc(n * L.stride + 64);
    HIP_CHECK(hipMemset(dev.blocks.p, 0, n * L.stride + 64));
    const size_t words = (D + 63) / 64;
    const uint32_t own_meta = (uint32_t)round_up(bw * words * 8, 64);
    const uint32_t own_stride = (uint32_t)hi.RL.nb_off;            // the vertex header in front of the neighbour block
    DevBuf<uint8_t> d_own(n * own_stride);
    HIP_CHECK(hipMemset(d_own.p, 0, n * own_stride));
    {
        EncodeArgsB a{};
        a.x = dev.raw.p; a.nbr = d_nbr_new.p; a.centroid = d_centroid.p; a.n = n; a.dim = (uint32_t)dim; a.D = (uint32_t)D;
        a.signs = d_signs.p; a.norm_factor = norm_factor; a.inv_sqrt_d = inv_sqrt_d; a.L = L; a.blocks = dev.blocks.p;
        a.own = d_own.p; a.own_stride = own_stride; a.own_meta = own_meta;
        run_encode(bw, a, n, num_cus);
        a.nbr = nullptr;
        run_encode(bw, a, (n + 31) / 32, num_cus);
    }
    tm.lap("gather + edge / own codes");

    // ---- upper layers: started before the kNN (above) --------------------------------------------------------------
    std::vector<int32_t> levels_new(n);
    for (size_t nw = 0; nw < n; ++nw) levels_new[nw] = levels[new_to_old[nw]];
    upper_job.wait();
    std::vector<std::vector<UpperEdge>> upper = ul.export_layers(old_to_new);
    const float upper_tau = ul.tau, upper_alpha = ul.alpha;
    tm.lap("upper layers (host threads, built behind the kNN): wait + export");

    // ---- host side: vectors, norms, own-code headers (the reference-layout block image waits for save()) ----------
    hi.raw.resize(n * D);
    hi.norm_sq.resize(n);
    HIP_CHECK(hipMemcpy(hi.raw.data(), dev.raw.p, n * D * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hi.norm_sq.data(), dev.norm.p, n * 4, hipMemcpyDeviceToHost));
    hi.levels = std::move(levels_new);
    // The reference-layout image of the neighbour blocks (hi.search_data, 2.9 KB per vertex at D = 128, 17.7 KB at
    // D = 1024) is only what save() writes: it is derived from the device blocks when a save asks for it
    // (materialize_search_data), not here.  The own-code headers are small and come down now.
    hi.search_data.clear();
    dev.own_host.resize(n * (size_t)own_stride);
    HIP_CHECK(hipMemcpy(dev.own_host.data(), d_own.p, n * (size_t)own_stride, hipMemcpyDeviceToHost));
    hi.upper = std::move(upper);
    hi.max_level = max_level;
    hi.entry = old_to_new[entry_old];
    hi.upper_tau = upper_tau;
    hi.upper_alpha = upper_alpha;
    hi.centroid = centroid;
    ProfileRecord prof;
    prof.n = n; prof.D = D; prof.R = R; prof.bits = bw;
    prof.evt_min_tail = std::max<size_t>(64, isqrt_sz(n));                                       // adaptive_defaults.hpp:45-46
    prof.min_calib_samples = std::clamp((size_t)(10.0 * std::sqrt((double)n)), (size_t)200, n);   // :48-52
    prof.slack_levels = std::clamp((int)std::ceil(std::log2(std::max(10.0f * std::log2((float)std::max(n, (size_t)64)), 4.0f))), 4, 32);
    prof.graph_stats = gs;
    std::memcpy(hi.profile, &prof, 72);
    std::memset(hi.calib, 0, sizeof(hi.calib));
    tm.lap("vectors, norms, own codes to the host");
}

// What the calibration needs from the device-resident index.
struct DeviceIndexView {
    const uint8_t* blocks;
    const float* raw;
    const float* signs;
    DevLayout L;
    float norm_factor, inv_sqrt_d;
};

// Estimator calibration: samples evaluated on the GPU, statistics on the host; writes hi.calib.
inline void calibrate(HostIndex& hi, const DeviceIndexView& dv, int num_cus, bool verbose) {
    StageTimer tm{verbose};
    const size_t n = hi.n, D = hi.D, dim = hi.dim, bw = hi.bw;
    if (n < 50) throw std::runtime_error("Calibration requires at least 50 nodes.");
    ProfileRecord prof;
    std::memcpy(&prof, hi.profile, 72);
    // ---- sample queries: database vectors (start = themselves) and perturbed ones (start = elsewhere) ----
    const size_t n_db = std::min(prof.min_calib_samples, n), n_syn = std::min(prof.min_calib_samples / 2, n);
    const size_t ns = n_db + n_syn;
    std::mt19937 rng(42 + 99999);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::shuffle(order.begin(), order.end(), rng);
    std::vector<float> sigma(dim, 0.0f);
    {
        const size_t vs = std::max<size_t>(2, std::min(n, prof.min_calib_samples / 4));
        std::vector<double> s1(dim, 0.0), s2(dim, 0.0);
        for (size_t i = 0; i < vs; ++i) {
            const float* v = hi.vec(order[i]);
            for (size_t d = 0; d < dim; ++d) { s1[d] += v[d]; s2[d] += (double)v[d] * v[d]; }
        }
        for (size_t d = 0; d < dim; ++d) {
            const double mu = s1[d] / vs;
            sigma[d] = (float)std::sqrt(std::max(s2[d] / vs - mu * mu, (double)kEpsSmall));
        }
    }
    std::vector<float> q(ns * dim);
    std::vector<uint32_t> start(ns);
    for (size_t i = 0; i < n_db; ++i) {
        std::memcpy(&q[i * dim], hi.vec(order[i]), dim * 4);
        start[i] = order[i];
    }
    std::normal_distribution<float> gauss(0.0f, 1.0f);
    for (size_t i = 0; i < n_syn; ++i) {
        const float* base = hi.vec(order[i % n]);
        for (size_t d = 0; d < dim; ++d) q[(n_db + i) * dim + d] = base[d] + gauss(rng) * sigma[d];
        start[n_db + i] = order[(n_db + i) % n];
    }
    // ---- on the GPU: encode, hop, estimate vs exact ---------------------------------------------------------
    const uint32_t PW = dv.L.PW;
    DevBuf<float> d_q(ns * dim), d_qp(ns * D), d_rec(ns * 32 * 6), d_dqp(ns), d_ed(ns);
    DevBuf<uint4> d_masks(ns * PW);
    DevBuf<QueryHeader> d_hdr(ns);
    DevBuf<uint32_t> d_start(ns), d_rc(ns);
    HIP_CHECK(hipMemcpy(d_q.p, q.data(), ns * dim * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_start.p, start.data(), ns * 4, hipMemcpyHostToDevice));
    {
        EncodeArgs e{};
        e.queries_raw = d_q.p; e.nq = (uint32_t)ns; e.dim = (uint32_t)dim; e.D = (uint32_t)D; e.PW = PW;
        e.signs = dv.signs; e.norm_factor = dv.norm_factor; e.inv_sqrt_d = dv.inv_sqrt_d;
        e.raw = dv.raw; e.n = n; e.entry = 0; e.max_level = 0;
        e.queries_padded = d_qp.p; e.qmasks = d_masks.p; e.qhdr = d_hdr.p; e.entry_dist = d_ed.p;
        hipLaunchKernelGGL(encode_kernel, dim3((uint32_t)std::min<uint64_t>(ns, (uint64_t)num_cus * 32)), dim3(64),
                           encode_lds_bytes((uint32_t)D), nullptr, e);
        HIP_CHECK(hipGetLastError());
        CalibArgs c{};
        c.blocks = dv.blocks; c.raw = dv.raw; c.L = dv.L; c.n = n; c.queries = d_qp.p; c.qmasks = d_masks.p; c.qhdr = d_hdr.p;
        c.start = d_start.p; c.ns = (uint32_t)ns; c.rec = d_rec.p; c.rec_cnt = d_rc.p; c.dqp_out = d_dqp.p;
        const size_t lds = (size_t)PW * 16 + (size_t)D * 8;
        const dim3 grid((uint32_t)std::min<uint64_t>(ns, (uint64_t)num_cus * 16));
        if (bw == 1) hipLaunchKernelGGL(calib_kernel<1>, grid, dim3(64), lds, nullptr, c);
        else if (bw == 2) hipLaunchKernelGGL(calib_kernel<2>, grid, dim3(64), lds, nullptr, c);
        else hipLaunchKernelGGL(calib_kernel<4>, grid, dim3(64), lds, nullptr, c);
        HIP_CHECK(hipGetLastError());
    }
    std::vector<float> rec(ns * 32 * 6), dqp(ns);
    std::vector<uint32_t> rc(ns);
    HIP_CHECK(hipMemcpy(rec.data(), d_rec.p, rec.size() * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(dqp.data(), d_dqp.p, ns * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(rc.data(), d_rc.p, ns * 4, hipMemcpyDeviceToHost));
    tm.lap("calibration samples (GPU)");

    // ---- statistics ---------------------------------------------------------------------------------------
    CalibrationRecord cal{};
    std::vector<float> ipqo, est_ratio, truth, nops;
    struct Pair { float nop, ipc, ipq, dqp, exact; };
    std::vector<Pair> pairs;
    for (size_t s = 0; s < ns; ++s)
        for (uint32_t j = 0; j < rc[s] && j < 32; ++j) {
            const float* r = &rec[(s * 32 + j) * 6];
            ipqo.push_back(r[5]);
            nops.push_back(r[0]);
            truth.push_back(r[3]);
            pairs.push_back({r[0], r[1], r[2], dqp[s], r[4]});
        }
    if (ipqo.empty()) throw std::runtime_error("Calibration failed: no ip_qo samples.");
    if (pairs.size() < 20) throw std::runtime_error("Calibration failed: too few estimator/target pairs.");
    {   // floor of the estimator's denominator: a low robust quantile of the observed ip_qo
        const float med = median_of(ipqo);
        cal.ip_qo_floor = std::max(med - 3.0f * mad_sigma(ipqo, med), kEpsMedium);
    }
    est_ratio.resize(pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) est_ratio[i] = pairs[i].ipc / std::max(pairs[i].ipq, cal.ip_qo_floor);
    {   // affine correction truth ~ a * estimate + b, kept only if it explains the data and no point dominates it
        double a, b;
        robust_line(est_ratio, truth, a, b);
        const size_t np = truth.size();
        double mt = 0, me = 0;
        for (size_t i = 0; i < np; ++i) { mt += truth[i]; me += est_ratio[i]; }
        mt /= np; me /= np;
        double ssr = 0, sst = 0, sxx = 0;
        for (size_t i = 0; i < np; ++i) {
            const double res = truth[i] - (a * est_ratio[i] + b);
            ssr += res * res;
            sst += (truth[i] - mt) * (truth[i] - mt);
            sxx += (est_ratio[i] - me) * (est_ratio[i] - me);
        }
        const double r2 = sst > kEpsSmall ? 1.0 - ssr / sst : 0.0;
        double lev = 0.0;
        if (sxx > kEpsSmall)
            for (size_t i = 0; i < np; ++i) lev = std::max(lev, 1.0 / np + (est_ratio[i] - me) * (est_ratio[i] - me) / sxx);
        if (r2 < 0.1 || lev > 4.0 / (double)np) { a = 1.0; b = 0.0; }
        cal.affine_a = (float)a;
        cal.affine_b = (float)b;
    }
    cal.median_nn_dist_sq = median_of(dqp);
    cal.min_slack_sq = std::max(kEpsSmall, cal.median_nn_dist_sq * 1e-4f);
    cal.median_nop = median_of(nops);
    // residuals of the calibrated distance estimate
    std::vector<float> resid(pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) {
        const Pair& p = pairs[i];
        const float den = std::max(p.ipq, cal.ip_qo_floor);
        const float ip = cal.affine_a * (den > kEpsMedium ? p.ipc / den : 0.0f) + cal.affine_b;
        const float est = std::max(p.nop * p.nop + p.dqp - 2.0f * p.nop * ip, 0.0f);
        resid[i] = std::fabs(est - p.exact);
    }
    std::sort(resid.begin(), resid.end());
    const size_t nr = resid.size();
    // (F9) the reference's minimum tail is unreachable for large n: cap it at half of what the sample supplies
    size_t min_tail = prof.evt_min_tail;
    const size_t supply = isqrt_sz(std::max(nr, (size_t)4));
    if (min_tail > supply / 2) min_tail = std::max<size_t>(64, supply / 2);
    const float q_lo = std::max(1.0f - 1.0f / std::sqrt((float)std::max(nr, (size_t)4)), 0.5f);
    const float q_hi = 1.0f - (float)min_tail / (float)std::max(nr, (size_t)1);
    cal.evt = fit_tail(resid, min_tail, q_lo, q_hi);
    if (!cal.evt.fitted || cal.median_nop <= 0.0f) throw std::runtime_error("Calibration failed: EVT-CRC fit did not converge.");
    const float ref = std::sqrt(std::max(cal.median_nn_dist_sq, cal.min_slack_sq));
    {
        const float q1 = resid[nr / 4] / ref, med = resid[nr / 2] / ref, q3 = resid[3 * nr / 4] / ref, iqr = q3 - q1;
        cal.gamma_min = std::max(1.0f + resid[std::max<size_t>(1, nr / 100)] / ref, 1.0f + 1.0f / std::sqrt((float)D));
        cal.gamma_max = std::max(1.0f + q3 + 1.5f * iqr, cal.gamma_min + std::max(iqr, med));
        double mu = 0, var = 0;
        for (float r : resid) mu += r;
        mu /= nr;
        for (float r : resid) var += (r - mu) * (r - mu);
        var /= nr;
        const float cv = (float)(std::sqrt(var) / std::max(mu, (double)kEpsSmall));
        cal.gamma_beta = 1.0f / std::max(cv, 1.0f / std::sqrt(2.0f * (float)(std::max(nr, (size_t)2) - 1)));
        cal.gamma_warmup = std::max<size_t>(4, (size_t)std::ceil(std::sqrt((float)cal.evt.n_tail)));
    }
    cal.slack_levels = prof.slack_levels;
    const int levels = std::clamp(cal.slack_levels, 1, 32);
    cal.search_num_slack_levels = levels;
    const float basel = 6.0f / (3.14159265358979f * 3.14159265358979f);    // sum 1/i^2 budget split
    for (int i = 1; i <= levels; ++i)
        cal.search_ip_slack_levels[i - 1] = tail_quantile(0.5e-4f * basel / ((float)i * (float)i), cal.evt) / (2.0f * cal.median_nop);
    cal.search_gamma = std::clamp(1.0f + tail_quantile(0.5e-4f, cal.evt) / ref, cal.gamma_min, cal.gamma_max);
    std::memcpy(hi.calib, &cal, 248);
    tm.lap("calibration statistics");
}

}  // namespace build
}  // namespace cph
