// device_knn.h — exact k-nearest-neighbour graph by tiled brute force (index construction).
//
// Replaces the reference's CPU NNDescent (graph/graph_refinement.hpp:71-263, 455-515), whose
// product is a working list of the R=32 nearest neighbours per node.  On MI355X the exact
// answer is cheaper than the approximation: n² · D FMAs in fp32 (2.6e14 flop at n = 1M,
// D = 128) with a threshold-filtered top-K kept in LDS, so nothing but the K results per node
// is ever written to HBM.
//
// One workgroup = 64 query rows x all base rows, 64x64 distance tiles, 4x4 register micro-tile
// per thread, K dimension staged through LDS in chunks of 32.  dist = |q|² + |b|² − 2 q·b.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cph {

constexpr int kKnnK = 32;       // neighbours kept per row (R)
constexpr int kKnnCap = 96;     // per-row candidate buffer (K + up to 64 new per tile)

struct KnnArgs {
    const float* x;        // [n][D]
    const float* norm;     // [n] squared norms (fp32)
    uint32_t n, D;
    uint32_t row_begin, row_end;   // query rows handled by this launch
    uint32_t* out_ids;     // [n][K] ascending by distance
    float* out_dist;       // [n][K]
};

__global__ __launch_bounds__(256) void knn_bruteforce_kernel(KnnArgs a) {
    __shared__ float Qs[32][64 + 4];     // [k][row]  (transposed: conflict-free column reads)
    __shared__ float Bs[32][64 + 4];     // [k][col]
    __shared__ float cand_d[64][kKnnCap];
    __shared__ uint32_t cand_i[64][kKnnCap];
    __shared__ uint32_t cnt[64];
    __shared__ float tau[64];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;         // 16 x 16 threads, 4x4 outputs each
    const uint32_t row0 = a.row_begin + blockIdx.x * 64;
    if (row0 >= a.row_end) return;
    if (tid < 64) { cnt[tid] = 0; tau[tid] = 3.402823466e+38f; }
    __syncthreads();

    for (uint32_t col0 = 0; col0 < a.n; col0 += 64) {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
        for (uint32_t k0 = 0; k0 < a.D; k0 += 32) {
            // stage 64 rows x 32 k of Q and B (transposed into [k][row])
            for (int e = tid; e < 64 * 32; e += 256) {
                const int r = e >> 5, k = e & 31;
                const uint32_t qr = row0 + r, br = col0 + r;
                Qs[k][r] = (qr < a.n && k0 + k < a.D) ? a.x[(size_t)qr * a.D + k0 + k] : 0.0f;
                Bs[k][r] = (br < a.n && k0 + k < a.D) ? a.x[(size_t)br * a.D + k0 + k] : 0.0f;
            }
            __syncthreads();
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                float q[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = Qs[k][ty * 4 + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(q[i], b[j], acc[i][j]);
            }
            __syncthreads();
        }
        // threshold filter into the per-row candidate buffers
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty * 4 + i;
            const uint32_t qr = row0 + r;
            if (qr >= a.n || qr >= a.row_end) continue;
            const float qn = a.norm[qr];
            const float t = tau[r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bc = col0 + tx * 4 + j;
                if (bc >= a.n || bc == qr) continue;
                float d = (qn + a.norm[bc]) - 2.0f * acc[i][j];
                d = d > 0.0f ? d : 0.0f;
                if (d < t) {
                    const uint32_t pos = atomicAdd(&cnt[r], 1u);
                    if (pos < (uint32_t)kKnnCap) { cand_d[r][pos] = d; cand_i[r][pos] = bc; }
                }
            }
        }
        __syncthreads();
        // compaction: rows whose buffer could overflow on the next tile keep their K best
        const int wave = tid >> 6, lane = tid & 63;
        const bool last = col0 + 64 >= a.n;
        for (int r = wave; r < 64; r += 4) {
            uint32_t c = cnt[r];
            if (c > (uint32_t)kKnnCap) c = kKnnCap;   // cannot happen: <= K + 64 by construction
            if (!(c > (uint32_t)(kKnnCap - 64) || (last && c > 0))) continue;
            // selection of the min(K, c) smallest by repeated wave-min extraction;
            // entries lane and lane+64; ties broken by smaller id for determinism
            float d0 = lane < (int)c ? cand_d[r][lane] : 3.402823466e+38f;
            float d1 = lane + 64 < (int)c ? cand_d[r][lane + 64] : 3.402823466e+38f;
            uint32_t i0 = lane < (int)c ? cand_i[r][lane] : 0xFFFFFFFFu;
            uint32_t i1 = lane + 64 < (int)c ? cand_i[r][lane + 64] : 0xFFFFFFFFu;
            const uint32_t keep = c < (uint32_t)kKnnK ? c : (uint32_t)kKnnK;
            float kth = 0.0f;
            for (uint32_t s = 0; s < keep; ++s) {
                float md = d0; uint32_t mi = i0; int which = 0;
                if (d1 < md || (d1 == md && i1 < mi)) { md = d1; mi = i1; which = 1; }
                float bd = md; uint32_t bi = mi;
                for (int o = 1; o < 64; o <<= 1) {
                    const float od = __shfl_xor(bd, o);
                    const uint32_t oi = __shfl_xor(bi, o);
                    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
                }
                if (mi == bi && md == bd) {   // this lane owns the extracted element
                    if (which == 0) { d0 = 3.402823466e+38f; i0 = 0xFFFFFFFFu; }
                    else { d1 = 3.402823466e+38f; i1 = 0xFFFFFFFFu; }
                }
                if (lane == 0) { cand_d[r][s] = bd; cand_i[r][s] = bi; }
                kth = bd;
                // LDS writes by lane 0 alias entries other lanes already hold in registers: safe
            }
            if (lane == 0) {
                cnt[r] = keep;
                if (keep == (uint32_t)kKnnK) tau[r] = kth;
            }
        }
        __syncthreads();
    }
    // rows are now sorted ascending (extraction order); write out, padding short rows
    for (int e = tid; e < 64 * kKnnK; e += 256) {
        const int r = e / kKnnK, s = e % kKnnK;
        const uint32_t qr = row0 + r;
        if (qr >= a.n || qr >= a.row_end) continue;
        const bool have = (uint32_t)s < cnt[r];
        a.out_ids[(size_t)qr * kKnnK + s] = have ? cand_i[r][s] : 0xFFFFFFFFu;
        a.out_dist[(size_t)qr * kKnnK + s] = have ? cand_d[r][s] : 3.402823466e+38f;
    }
}

}  // namespace cph
