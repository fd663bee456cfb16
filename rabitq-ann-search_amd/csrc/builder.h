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
#include "device_knn_sym.h"
#include "host_index.h"
#include "host_parallel.h"
#include "builder_host.h"

namespace cph {
namespace build {

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

// ---- exact kNN on the matrix cores (device pointers; D a multiple of 32) ----------------------------
// exclude_self: query row i never lists column i -- or column q_ids[i] when the query rows are a gathered subset of the base
inline void knn_device(const float* d_q, const float* d_qnorm, size_t nq, const float* d_b, const float* d_bnorm,
                       size_t nb, size_t D, bool exclude_self, int num_cus, uint32_t* d_ids, float* d_dist,
                       const uint32_t* d_q_ids = nullptr) {
    if (D % kKnnKC != 0 || nq == 0 || nb == 0) throw std::invalid_argument("knn_device: D must be a multiple of 32");
    // slices of row blocks, so that no single launch runs for minutes
    const uint32_t rows_per_launch = (uint32_t)num_cus * 8u * kKnnTile;
    for (size_t rb = 0; rb < nq; rb += rows_per_launch) {
        KnnArgs a{d_q, d_b, d_bnorm, d_qnorm, (uint32_t)nq, (uint32_t)nb, (uint32_t)D, (uint32_t)rb,
                  (uint32_t)std::min<size_t>(nq, rb + rows_per_launch), exclude_self ? 1u : 0u, d_ids, d_dist, d_q_ids};
        const uint32_t grid = (a.row_end - a.row_begin + kKnnTile - 1) / kKnnTile;
        hipLaunchKernelGGL(knn_mfma_kernel, dim3(grid), dim3(256), 0, nullptr, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
}

// ---- the self-join: every tile of the distance matrix once (device_knn_sym.h) -----------------------------------------
// d_x [n][D] against itself, a row never lists itself.  Small inputs, D = 32 (one chunk per tile: the staging protocol of
// the column norms needs two) and CPH_KNN_SYM=0 take knn_mfma_kernel with every pair computed twice.
inline bool knn_sym_wanted(size_t n, size_t D) {
    static const int env = getenv("CPH_KNN_SYM") ? atoi(getenv("CPH_KNN_SYM")) : -1;
    if (env == 0) return false;
    const size_t min_n = env > 1 ? (size_t)env : 32768;      // CPH_KNN_SYM=<n>: lower the size limit (tests)
    return n >= min_n && n / kKnnSymSample >= 64 && D >= 2 * kKnnKC && n < (1ull << 31);
}
inline void knn_self_device(const float* d_x, const float* d_norm, size_t n, size_t D, int num_cus, uint32_t* d_ids, float* d_dist,
                            bool verbose = false) {
    if (!knn_sym_wanted(n, D)) {
        knn_device(d_x, d_norm, n, d_x, d_norm, n, D, true, num_cus, d_ids, d_dist);
        return;
    }
    StageTimer tm{verbose};
    // 1. thresholds from a sample of the rows
    const uint32_t ns = (uint32_t)((n + kKnnSymSample - 1) / kKnnSymSample);
    DevBuf<float> d_tau(n);
    {
        DevBuf<uint32_t> d_sid(ns), d_soi(n * kKnnK);
        DevBuf<float> d_sx((size_t)ns * D), d_sn(ns), d_sod(n * kKnnK);
        hipLaunchKernelGGL(knn_sym_iota_kernel, dim3(grid_for(ns, 256)), dim3(256), 0, nullptr, d_sid.p, ns, kKnnSymSample);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)std::min<size_t>(ns, (size_t)num_cus * 64)), dim3(64), 0, nullptr, d_x,
                           d_sid.p, (uint64_t)ns, (uint32_t)D, d_sx.p);
        hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(ns, 256)), dim3(256), 0, nullptr, d_norm, d_sid.p, (uint64_t)ns, d_sn.p);
        HIP_CHECK(hipGetLastError());
        knn_device(d_x, d_norm, n, d_sx.p, d_sn.p, ns, D, false, num_cus, d_soi.p, d_sod.p);
        hipLaunchKernelGGL(knn_sym_tau_kernel, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, d_sod.p, (uint32_t)n, d_tau.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
    tm.lap("kNN: thresholds from a 1/16 sample");
    // 2. the join: every tile once, both sides
    DevBuf<uint32_t> d_cnt(n), d_cid(n * kKnnSymCap), d_redo(n), d_nredo(1);
    DevBuf<float> d_cd(n * kKnnSymCap);
    HIP_CHECK(hipMemset(d_cnt.p, 0, n * 4));
    HIP_CHECK(hipMemset(d_nredo.p, 0, 4));
    const uint32_t nblk = (uint32_t)((n + kKnnTile - 1) / kKnnTile);
    const uint32_t nwg = (nblk + 1) / 2;
    const uint32_t wg_per_launch = (uint32_t)num_cus * 4u;     // slices, so that no single launch runs for minutes
    for (uint32_t w0 = 0; w0 < nwg; w0 += wg_per_launch) {
        KnnSymArgs a{d_x, d_norm, d_tau.p, (uint32_t)n, (uint32_t)D, nblk, w0, d_cnt.p, d_cid.p, d_cd.p};
        hipLaunchKernelGGL(knn_sym_kernel, dim3(std::min(wg_per_launch, nwg - w0)), dim3(256), 0, nullptr, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
    tm.lap("kNN: symmetric join");
    // 3. the best 32 of every row's candidates; rows without enough of them go on the list
    hipLaunchKernelGGL(knn_sym_select_kernel, dim3((uint32_t)std::min<size_t>(n, (size_t)num_cus * 64)), dim3(64), 0, nullptr, d_cnt.p,
                       d_cid.p, d_cd.p, (uint32_t)n, d_ids, d_dist, d_redo.p, d_nredo.p);
    HIP_CHECK(hipGetLastError());
    uint32_t nredo = 0;
    HIP_CHECK(hipMemcpy(&nredo, d_nredo.p, 4, hipMemcpyDeviceToHost));
    tm.lap("kNN: selection");
    // 4. ... and are answered exactly against every row
    if (nredo) {
        DevBuf<float> d_rx((size_t)nredo * D), d_rn(nredo), d_rd((size_t)nredo * kKnnK);
        DevBuf<uint32_t> d_ri((size_t)nredo * kKnnK);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)std::min<size_t>(nredo, (size_t)num_cus * 64)), dim3(64), 0, nullptr, d_x,
                           d_redo.p, (uint64_t)nredo, (uint32_t)D, d_rx.p);
        hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(nredo, 256)), dim3(256), 0, nullptr, d_norm, d_redo.p, (uint64_t)nredo, d_rn.p);
        HIP_CHECK(hipGetLastError());
        knn_device(d_rx.p, d_rn.p, nredo, d_x, d_norm, n, D, true, num_cus, d_ri.p, d_rd.p, d_redo.p);
        hipLaunchKernelGGL(knn_sym_scatter_kernel, dim3(grid_for((size_t)nredo * kKnnK, 256)), dim3(256), 0, nullptr, d_redo.p, nredo,
                           d_ri.p, d_rd.p, d_ids, d_dist);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    }
    if (verbose) fprintf(stderr, "[build] kNN: %u of %zu rows took the fallback\n", nredo, n);
    tm.lap("kNN: fallback rows");
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
        knn_self_device(d_x.p, d_norm.p, n, Dk, num_cus, d_oi.p, d_od.p, getenv("CPH_BUILD_VERBOSE") != nullptr);
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

}  // namespace build
}  // namespace cph
