// device_encode.h — per-query feeders on the GPU: query encoder + upper-layer greedy descent.
//
// GPU counterparts of
//   RandomHadamardRotation::apply_copy    encoder/rotation.hpp:34-51, transform/fht.hpp:23-57
//   encode_query_raw / build_lut          encoder/rabitq_encoder.hpp:73-79,98-136,197-209
//   Index::search prologue + greedy_search_layer  api/hnsw_index.hpp:174-202,468-474,617-638
// One wave per query.  Every butterfly output is a single add/sub, the scale is a single
// multiply and the quantiser is one explicit fma, so any schedule is bit-exact as long as the
// stage order and the reference's sign convention are kept; min/max carry the index of their
// first occurrence, reproducing the reference's sequential strict-compare scans.
#pragma once
#include <hip/hip_runtime.h>

#include "cph_core.h"
#include "device_fastscan.h"

namespace cph {

struct UpperLayerDev {       // CSR of one upper layer, nodes sorted ascending (find_edge)
    const uint32_t* nodes;   // [n_nodes]
    const uint32_t* offsets; // [n_nodes + 1]
    const uint32_t* nbrs;
    const uint32_t* row_of;  // [n] vertex -> first edge | degree << 26 (kInvalidNode if absent), or null:
                             // binary search over `nodes` + `offsets`
    uint32_t n_nodes;
};

constexpr int kMaxUpperLayers = 24;

struct EncodeArgs {
    const float* queries_raw;  // [nq][dim]
    uint32_t nq, dim, D, PW;
    const float* signs;        // [3][D]
    float norm_factor, inv_sqrt_d;
    // index (upper-layer descent)
    const float* raw;          // [n][D]
    uint64_t n;
    uint32_t entry;
    int32_t max_level;         // number of usable upper layers (0 = none)
    UpperLayerDev layers[kMaxUpperLayers];
    // outputs
    float* queries_padded;     // [nq][D]
    uint4* qmasks;             // [nq][PW]
    QueryHeader* qhdr;         // [nq]
    float* entry_dist;         // [nq] squared distance to the layer-0 entry (scheduling key), or null
};

__device__ __forceinline__ void wave_fht(float* x, uint32_t D, int lane) {
    for (uint32_t h = 1; h < D; h <<= 1) {
        for (uint32_t b = lane; b < D / 2; b += 64) {
            const uint32_t i = (b / h) * 2 * h + (b % h);
            const float lo = x[i], hi = x[i + h];
            x[i] = lo + hi;
            x[i + h] = (h < 8) ? (hi - lo) : (lo - hi);
        }
        __syncthreads();
    }
}

// LDS: raw padded query [D] | work buffer [D]
__host__ __device__ inline size_t encode_lds_bytes(uint32_t D) { return (size_t)D * 8; }

__global__ __launch_bounds__(64) void encode_kernel(EncodeArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* qv = reinterpret_cast<float*>(smem);
    float* x = qv + a.D;
    const int lane = threadIdx.x;
    const uint32_t D = a.D;
    for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
        for (uint32_t d = lane; d < D; d += 64) {
            float v = d < a.dim ? a.queries_raw[(size_t)qi * a.dim + d] : 0.0f;
            qv[d] = v;
            x[d] = v;
            a.queries_padded[(size_t)qi * D + d] = v;
        }
        __syncthreads();
        // ---- rotation: 3 x (diag, FHT), then the deferred D^-1.5 normalisation ---------
        for (int l = 0; l < 3; ++l) {
            for (uint32_t d = lane; d < D; d += 64) x[d] = x[d] * a.signs[l * D + d];
            __syncthreads();
            wave_fht(x, D, lane);
        }
        for (uint32_t d = lane; d < D; d += 64) x[d] = x[d] * a.norm_factor;
        __syncthreads();
        // ---- min / max with first-occurrence semantics (rabitq_encoder.hpp:99-103) -------
        float vmin = x[lane < (int)D ? lane : 0], vmx = vmin;
        uint32_t imin = lane < (int)D ? lane : 0, imax = imin;
        for (uint32_t d = lane + 64; d < D; d += 64) {
            const float v = x[d];
            if (v < vmin) { vmin = v; imin = d; }
            if (v > vmx) { vmx = v; imax = d; }
        }
        for (int o = 1; o < 64; o <<= 1) {
            const float ov = __shfl_xor(vmin, o);
            const uint32_t oi = __shfl_xor(imin, o);
            if (ov < vmin || (ov == vmin && oi < imin)) { vmin = ov; imin = oi; }
            const float ow = __shfl_xor(vmx, o);
            const uint32_t oj = __shfl_xor(imax, o);
            if (ow > vmx || (ow == vmx && oj < imax)) { vmx = ow; imax = oj; }
        }
        const float vl = x[imin];     // exact bits of the first minimum (keeps the sign of zero)
        const float vh = x[imax];
        float delta = (vh - vl) / 15.0f;
        if (delta < kEpsTiny) delta = kEpsTiny;
        const float inv_delta = 1.0f / delta;
        // ---- 4-bit scalars, their sum, and the bit-sliced masks ---------------------------
        uint32_t usum = 0;
        for (uint32_t base = 0; base < D; base += 64) {
            const uint32_t d = base + lane;
            int u = 0;
            if (d < D) {
                u = (int)__fmaf_rn(x[d] - vl, inv_delta, 0.5f);
                u = u < 0 ? 0 : (u > 15 ? 15 : u);
            }
            usum += (uint32_t)u;
            const unsigned long long b0 = __ballot(u & 1), b1 = __ballot(u & 2), b2 = __ballot(u & 4),
                                     b3 = __ballot(u & 8);
            if (lane == 0) {
                const uint32_t w = base / 32;
                a.qmasks[(size_t)qi * a.PW + w] =
                    make_uint4((uint32_t)b0, (uint32_t)b1, (uint32_t)b2, (uint32_t)b3);
                if (w + 1 < a.PW)
                    a.qmasks[(size_t)qi * a.PW + w + 1] = make_uint4(
                        (uint32_t)(b0 >> 32), (uint32_t)(b1 >> 32), (uint32_t)(b2 >> 32), (uint32_t)(b3 >> 32));
            }
        }
        for (int o = 1; o < 64; o <<= 1) usum += __shfl_xor(usum, o);
        const float sum_qu = (float)usum;  // integer-valued partial sums: exact in any order
        const float df = (float)D;
        QueryHeader hd;
        hd.A = (2.0f * delta) * a.inv_sqrt_d;
        hd.B = (2.0f * vl) * a.inv_sqrt_d;
        hd.C = -__fmaf_rn(df, vl, delta * sum_qu) * a.inv_sqrt_d;

        // ---- upper-layer greedy descent (api/hnsw_index.hpp:196-202,617-638) ---------------
        uint32_t ep = a.entry;
        float ep_dist = 0.0f;
        if (a.max_level > 0 && ep < a.n) {
            for (int level = a.max_level; level >= 1; --level) {
                const UpperLayerDev& Ly = a.layers[level - 1];
                // distance to the layer's entry: the previous layer's winner, whose distance is already
                // known (same query, same vector, same arithmetic => same float as recomputing it)
                float best = (level == a.max_level) ? group_l2sq8(qv, a.raw + (size_t)ep * D, D, lane & 7) : ep_dist;
                uint32_t best_id = ep;
                bool improved = true;
                while (improved) {
                    improved = false;
                    float cand = 3.402823466e+38f;
                    uint32_t cand_id = kInvalidNode;
                    uint32_t cand_pos = 0xFFFFFFFFu;
                    if (Ly.row_of) {
                        // dense vertex -> (first edge, degree) map, then the whole neighbour list in one
                        // load: two dependent round trips per hop ahead of the vectors instead of
                        // log2(n_nodes) + 1 + one per pass
                        const uint32_t info = Ly.row_of[best_id];
                        if (info == kInvalidNode) break;
                        const uint32_t beg = info & 0x03FFFFFFu, cnt = info >> 26;
                        const uint32_t mine = (uint32_t)lane < cnt ? Ly.nbrs[beg + lane] : best_id;
                        // (issuing the loads of all passes together was tried: 150 VGPRs, 3 waves per
                        // SIMD, and the kernel got slower -- occupancy x latency stays the same product)
                        for (uint32_t base = 0; base < cnt; base += 8) {
                            const uint32_t pos = base + (lane >> 3);
                            const bool have = pos < cnt;
                            const uint32_t nb = (uint32_t)__shfl((int)mine, (int)(have ? pos : 0));
                            const float d = group_l2sq8(qv, a.raw + (size_t)(have ? nb : best_id) * D, D, lane & 7);
                            if (have && (d < cand || (d == cand && pos < cand_pos))) {
                                cand = d; cand_id = nb; cand_pos = pos;
                            }
                        }
                    } else {
                        // find_edge: lower_bound over the sorted node list (uniform across lanes)
                        uint32_t lo = 0, hi = Ly.n_nodes;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (Ly.nodes[mid] < best_id) lo = mid + 1; else hi = mid;
                        }
                        if (lo >= Ly.n_nodes || Ly.nodes[lo] != best_id) break;
                        const uint32_t beg = Ly.offsets[lo], end = Ly.offsets[lo + 1];
                        // neighbours in stored order; strict improvement => the first minimum wins
                        for (uint32_t base = beg; base < end; base += 8) {
                            const uint32_t pos = base + (lane >> 3);
                            const bool have = pos < end;
                            const uint32_t nb = have ? Ly.nbrs[pos] : best_id;
                            const float d = group_l2sq8(qv, a.raw + (size_t)nb * D, D, lane & 7);
                            if (have && (d < cand || (d == cand && pos < cand_pos))) {
                                cand = d; cand_id = nb; cand_pos = pos;
                            }
                        }
                    }
                    for (int o = 8; o < 64; o <<= 1) {
                        const float od = __shfl_xor(cand, o);
                        const uint32_t oi = __shfl_xor(cand_id, o);
                        const uint32_t op = __shfl_xor(cand_pos, o);
                        if (od < cand || (od == cand && op < cand_pos)) { cand = od; cand_id = oi; cand_pos = op; }
                    }
                    if (cand_id != kInvalidNode && cand < best) {
                        best = cand;
                        best_id = cand_id;
                        improved = true;
                    }
                }
                ep = best_id;
                ep_dist = best;
            }
        }
        hd.entry = ep;
        if (lane == 0) {
            a.qhdr[qi] = hd;
            if (a.entry_dist) a.entry_dist[qi] = ep_dist;
        }
        __syncthreads();
    }
}

// ---- launch order of a batch ------------------------------------------------------------
// Queries whose layer-0 entry is close tend to expand the most vertices (dense neighbourhoods:
// rank correlation -0.8 on the benchmark index), and a batch only a few times larger than the
// number of resident query slots ends when its last-started long query ends.  The work queue
// therefore hands queries out closest-entry-first: a counting sort on the top 14 bits of the
// (non-negative) float key, one workgroup, LDS histogram.  Any order gives the same results.
constexpr uint32_t kOrderBuckets = 16384;
__global__ __launch_bounds__(1024) void order_kernel(const float* __restrict__ key, uint32_t nq,
                                                     uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[kOrderBuckets];
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < kOrderBuckets; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nq; i += 1024) {
        const uint32_t kb = __float_as_uint(key[i]);
        const uint32_t b = (kb & 0x80000000u) ? 0u : (kb >> 17);   // sign bit set (or NaN payloads) first
        atomicAdd(&hist[b < kOrderBuckets ? b : kOrderBuckets - 1], 1u);
    }
    __syncthreads();
    // exclusive prefix sum over the buckets: 16 consecutive buckets per thread
    uint32_t local[16], run = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { local[j] = run; run += hist[tid * 16 + j]; }
    uint32_t inc = run;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if ((tid & 63) >= (uint32_t)o) inc += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = inc - run;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) hist[tid * 16 + j] = base + local[j];
    __syncthreads();
    for (uint32_t i = tid; i < nq; i += 1024) {
        const uint32_t kb = __float_as_uint(key[i]);
        const uint32_t b = (kb & 0x80000000u) ? 0u : (kb >> 17);
        const uint32_t pos = atomicAdd(&hist[b < kOrderBuckets ? b : kOrderBuckets - 1], 1u);
        order[pos] = i;
    }
}

}  // namespace cph
