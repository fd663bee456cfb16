// device_build.h — index construction kernels, MI355X-first.
//
// What the reference does on the host with OpenMP loops is laid out here for the GPU:
//
//   encode_edges_kernel   per-edge RaBitQ / CAQ codes and aux values of a vertex' 32 edges
//                         (encoder/rabitq_encoder.hpp:138-181, 287-323, 371-467), written straight into
//                         the device block layout the search kernel reads.  One wave per vertex: the
//                         rotation of each edge vector is cooperative (FHT in LDS), the coordinate
//                         descent -- inherently sequential over the dimensions -- runs one edge per LANE
//                         on LDS rows, so 32 descents advance in lockstep.  Bit-identical to the
//                         reference's codes and floats (tests/test_gpu_builder.py against golden vectors).
//   encode_own_kernel     the vertex' own code against the centroid (encode_impl, :326-352, :224-262);
//                         same machinery, file-format output.
//   select_kernel         neighbour selection (graph/neighbor_selection.hpp:21-88's rule) for a vertex
//                         from its forward list plus its reverse edges: one wave per vertex, candidates
//                         rank-sorted through v_readlane, the occlusion tests of one candidate against
//                         all selected neighbours evaluated 8 at a time.
//   reverse_*             in-degree count and CSR fill of the reverse edges (atomics).
//   calib_kernel          one wave per calibration sample: one greedy hop, then the FastScan estimate
//                         of every edge of the chosen vertex next to the exact values (what
//                         api/hnsw_index.hpp:718-1139 gathers on the host); the statistics stay on the host.
//   gather / norms        row gathers for the BFS reorder and the upper-layer subsets.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cph_core.h"
#include "device_encode.h"
#include "device_fastscan.h"

namespace cph {
namespace build {

// ---- small helpers ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ x, uint64_t n, uint32_t D,
                                                        uint32_t dim, float* __restrict__ out) {
    // squared norm as an fmaf chain over the first `dim` elements (the padding is zero)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = x + i * D;
    float s = 0.0f;
    for (uint32_t j = 0; j < dim; ++j) s = __fmaf_rn(v[j], v[j], s);
    out[i] = s;
}

// out[r][0..D) = src[ids[r]][0..D)   (one wave per row)
__global__ __launch_bounds__(64) void gather_rows_kernel(const float* __restrict__ src, const uint32_t* __restrict__ ids,
                                                         uint64_t rows, uint32_t D, float* __restrict__ out) {
    for (uint64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* s = src + (size_t)ids[r] * D;
        float* o = out + r * D;
        for (uint32_t d = threadIdx.x; d < D; d += 64) o[d] = s[d];
    }
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const float* __restrict__ src, const uint32_t* __restrict__ ids,
                                                         uint64_t rows, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) out[i] = src[ids[i]];
}

// Column sums of x[n][D] over the first `dim` columns, in double (centroid).  grid x 256 threads; each
// block handles a slice of rows and adds its partial sums atomically.
__global__ __launch_bounds__(256) void column_sums_kernel(const float* __restrict__ x, uint64_t n, uint32_t D,
                                                          uint32_t dim, double* __restrict__ out) {
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t j = threadIdx.x; j < dim; j += 256) {
        double s = 0.0;
        for (uint64_t i = lo; i < hi; ++i) s += (double)x[i * D + j];
        atomicAdd(&out[j], s);
    }
}

// err[i] = scale * |x_i - centroid| over the first `dim` columns
__global__ __launch_bounds__(256) void centered_norm_kernel(const float* __restrict__ x, const float* __restrict__ centroid,
                                                            uint64_t n, uint32_t D, uint32_t dim, float scale,
                                                            float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (uint32_t j = 0; j < dim; ++j) {
        const float t = x[i * D + j] - centroid[j];
        s += t * t;
    }
    out[i] = scale * __builtin_sqrtf(s);
}

// ids[i] = map[ids[i]] (kInvalidNode stays)
__global__ __launch_bounds__(256) void remap_ids_kernel(uint32_t* __restrict__ ids, uint64_t count,
                                                        const uint32_t* __restrict__ map) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && ids[i] != kInvalidNode) ids[i] = map[ids[i]];
}

// out[r][j] = map[src[perm[r]][j]]: neighbour lists moved to their rows' new positions, ids renumbered
__global__ __launch_bounds__(256) void permute_lists_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ map, uint64_t rows,
                                                            uint32_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 32) return;
    const uint32_t v = src[(size_t)perm[i / 32] * 32 + (i % 32)];
    out[i] = v == kInvalidNode ? kInvalidNode : map[v];
}

// ---- reverse edges ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reverse_count_kernel(const uint32_t* __restrict__ knn, uint64_t n_edges,
                                                            uint32_t* __restrict__ indeg) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t v = knn[e];
    if (v != kInvalidNode) atomicAdd(&indeg[v], 1u);
}

__global__ __launch_bounds__(256) void reverse_fill_kernel(const uint32_t* __restrict__ knn, uint64_t n_edges,
                                                           uint32_t K, const uint64_t* __restrict__ offs,
                                                           uint32_t* __restrict__ cursor, uint32_t* __restrict__ rev) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t v = knn[e];
    if (v == kInvalidNode) return;
    const uint32_t pos = atomicAdd(&cursor[v], 1u);
    rev[offs[v] + pos] = (uint32_t)(e / K);
}

// ---- rotation of one vector held in LDS (whole wave, 64-thread workgroup) ------------------------
// x <- H S3 H S2 H S1 x * D^-1.5   (encoder/rotation.hpp:34-51 + the deferred normalisation, :81-86)
__device__ __forceinline__ void rotate_scaled_lds(float* x, const float* __restrict__ signs, uint32_t D,
                                                  float norm_factor, int lane) {
    for (int l = 0; l < 3; ++l) {
        for (uint32_t d = lane; d < D; d += 64) x[d] = x[d] * signs[l * D + d];
        __syncthreads();
        wave_fht(x, D, lane);
    }
    for (uint32_t d = lane; d < D; d += 64) x[d] = x[d] * norm_factor;
    __syncthreads();
}

// ---- data-side quantiser of one lane's row -------------------------------------------------------
// x = the lane's rotated, scaled vector (LDS, stride-1 elements), u = its code bytes (LDS), rp = rotated
// parent (LDS, shared) or null.  Operation for operation the reference as GCC compiles it
// (oracle/cph_oracle.cpp: orc_encode_edges documents where the compiler fuses and where its vectoriser
// keeps product and sum apart).  ctab[q] = (2q - K) / K.
struct LaneCode {
    float ip_qo, ip_cp;
    uint32_t msb_pop, weighted_pop;
};

// One lane's row of the rotated edge vectors and of their code bytes.  G = false: LDS rows (x[i] contiguous, the code
// bytes contiguous); G = true: the workgroup's transposed scratch in HBM -- element i of edge e at T[i * 32 + e] and
// the packed codes of coordinates 4j..4j+3 at U32[j * 32 + e] -- so that the 32 lanes of a wave read one 128-byte line
// per coordinate and the number of resident waves is not set by LDS (D = 1024: 5 KB per edge, 8 edges per workgroup).
template <bool G>
struct LaneRow {
    float* x;
    uint8_t* u;
    __device__ __forceinline__ float X(uint32_t i) const { return x[(size_t)i * (G ? 32 : 1)]; }
    __device__ __forceinline__ void setX(uint32_t i, float v) const { x[(size_t)i * (G ? 32 : 1)] = v; }
    __device__ __forceinline__ uint32_t U4(uint32_t i0) const {
        return *reinterpret_cast<const uint32_t*>(u + (size_t)(i0 >> 2) * (G ? 128 : 4));
    }
    __device__ __forceinline__ void setU4(uint32_t i0, uint32_t v) const {
        *reinterpret_cast<uint32_t*>(u + (size_t)(i0 >> 2) * (G ? 128 : 4)) = v;
    }
    __device__ __forceinline__ uint32_t Ub(uint32_t i) const { return u[(size_t)(i >> 2) * (G ? 128 : 4) + (i & 3)]; }
    __device__ __forceinline__ void setUb(uint32_t i, uint32_t v) const { u[(size_t)(i >> 2) * (G ? 128 : 4) + (i & 3)] = (uint8_t)v; }
};

template <int BW, bool G>
__device__ __forceinline__ LaneCode quantize_lane(const LaneRow<G> r, const float* rp, const float* ctab,
                                                  uint32_t D, float inv_sqrt_d) {
    LaneCode o{0.0f, 0.0f, 0u, 0u};
    if constexpr (BW == 1) {
        float l1 = 0.0f, ipcp = 0.0f;
        uint32_t pc = 0;
        for (uint32_t i = 0; i < D; ++i) {
            const float v = r.X(i);
            const bool pos = v >= 0.0f;
            r.setUb(i, pos ? 1u : 0u);
            l1 += __builtin_fabsf(v);
            pc += pos ? 1u : 0u;
        }
        if (rp)
            for (uint32_t i = 0; i < D; ++i) ipcp += (r.Ub(i) ? 1.0f : -1.0f) * rp[i];
        o.ip_qo = l1 * inv_sqrt_d;
        o.ip_cp = ipcp * inv_sqrt_d;
        o.msb_pop = o.weighted_pop = pc;
        return o;
    } else {
        constexpr int Ki = (1 << BW) - 1;
        const float K = (float)Ki;
        float mn = r.X(0), mx = r.X(0);
        for (uint32_t i = 1; i < D; ++i) {
            const float v = r.X(i);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        float delta = (mx - mn) / K;
        const float ceps = 1e-10f / (float)D;      // coordinate_epsilon
        if (delta < ceps) delta = ceps;
        const float inv_delta = 1.0f / delta;
        float dot = 0.0f, nrm = 0.0f;
        for (uint32_t i0 = 0; i0 < D; i0 += 4) {     // four coordinates' reads together, sums in order
            float xv[4], c[4];
            uint32_t un = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = r.X(i0 + j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int q = (int)__fmaf_rn(xv[j] - mn, inv_delta, 0.5f);
                q = q < 0 ? 0 : (q > Ki ? Ki : q);
                un |= (uint32_t)q << (8 * j);
                c[j] = ctab[q];
            }
            r.setU4(i0, un);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dot = dot + c[j] * xv[j];
                nrm = nrm + c[j] * c[j];
            }
        }
        // The descent is sequential in i (each coordinate sees the running dot / norm of all earlier ones), but what it
        // READS is not: four coordinates' values, their codes (one dword) and the levels around each code are fetched
        // together before the four dependent updates -- two LDS round trips per four coordinates instead of two per
        // coordinate (the loop is latency-bound: 8 to 32 lanes of the wave are live).  D is a multiple of 16.
        float ct[BW >= 4 ? 1 : Ki + 1];                 // 2-bit: all four levels in registers
        if constexpr (BW < 4) {
#pragma unroll
            for (int t = 0; t <= Ki; ++t) ct[t] = ctab[t];
        }
        float prev = 0.0f;
        for (int iter = 0; iter < 10; ++iter) {
            bool changed = false;
            for (uint32_t i0 = 0; i0 < D; i0 += 4) {
                const uint32_t up = r.U4(i0);
                float xv[4], oc[4], cl[4], ch[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = r.X(i0 + j);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ou = (int)((up >> (8 * j)) & 255u);
                    oc[j] = ctab[ou];
                    if constexpr (BW >= 4) {
                        cl[j] = ctab[ou > 0 ? ou - 1 : 0];
                        ch[j] = ctab[ou < Ki ? ou + 1 : Ki];
                    }
                }
                uint32_t un = up;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = xv[j];
                    const int ou = (int)((up >> (8 * j)) & 255u);
                    const float dwo = __fmaf_rn(-oc[j], v, dot);
                    const float nwo = __fmaf_rn(-oc[j], oc[j], nrm);
                    int bu = ou;
                    float bd = dot, bn = nrm;
                    if constexpr (BW >= 4) {
                        if (ou - 1 >= 0) {
                            const float c = cl[j];
                            const float nd = __fmaf_rn(c, v, dwo), nn = __fmaf_rn(c, c, nwo);
                            if (nd * nd * bn > bd * bd * nn) { bu = ou - 1; bd = nd; bn = nn; }
                        }
                        if (ou + 1 <= Ki) {
                            const float c = ch[j];
                            const float nd = __fmaf_rn(c, v, dwo), nn = __fmaf_rn(c, c, nwo);
                            if (nd * nd * bn > bd * bd * nn) { bu = ou + 1; bd = nd; bn = nn; }
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t <= Ki; ++t) {
                            if (t == ou) continue;
                            const float c = ct[t];
                            const float nd = __fmaf_rn(c, v, dwo), nn = __fmaf_rn(c, c, nwo);
                            if (nd * nd * bn > bd * bd * nn) { bu = t; bd = nd; bn = nn; }
                        }
                    }
                    if (bu != ou) {
                        // (the new level is one of the three just tried: its running sums are bd / bn, computed by the same
                        // two fused operations the reference repeats for the winner)
                        dot = bd;
                        nrm = bn;
                        un = (un & ~(255u << (8 * j))) | ((uint32_t)bu << (8 * j));
                        changed = true;
                    }
                }
                if (un != up) r.setU4(i0, un);
            }
            if (!changed) break;
            const float cs = nrm > 0.0f ? dot * dot / nrm : 0.0f;
            if (iter > 0 && (cs - prev) < 1e-4f) break;     // kCaqEarlyExitTol
            prev = cs;
        }
        float ipqo = 0.0f, ipcp = 0.0f;
        uint32_t msb = 0, wp = 0;
        for (uint32_t i0 = 0; i0 < D; i0 += 4) {
            const uint32_t up = r.U4(i0);
            float xv[4], rv[4], c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[j] = r.X(i0 + j);
                rv[j] = rp ? rp[i0 + j] : 0.0f;
                c[j] = ctab[(up >> (8 * j)) & 255u];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = (up >> (8 * j)) & 255u;
                ipqo = __fmaf_rn(c[j], xv[j], ipqo);
                if (rp) ipcp = __fmaf_rn(c[j], rv[j], ipcp);
                wp += q;
                msb += (q >> (BW - 1)) & 1u;
            }
        }
        o.ip_qo = ipqo * inv_sqrt_d;
        o.ip_cp = ipcp * inv_sqrt_d;
        o.msb_pop = msb;
        o.weighted_pop = wp;
        return o;
    }
}

// Squared norm of the lane's difference row as the reference's loop compiles (products rounded and added
// in order for the vectorised part, fused for the dim % 4 scalar remainder).
template <bool G>
__device__ __forceinline__ float seq_norm_sq(const LaneRow<G> r, uint32_t dim) {
    float s = 0.0f;
    const uint32_t body = dim - dim % 4;
    for (uint32_t i0 = 0; i0 < body; i0 += 4) {      // four reads in flight, sums in order
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = r.X(i0 + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) s = s + v[j] * v[j];
    }
    for (uint32_t i = body; i < dim; ++i) { const float v = r.X(i); s = __fmaf_rn(v, v, s); }
    return s;
}

struct EncodeArgsB {
    const float* x;          // [n][D] vectors, zero padded
    const uint32_t* nbr;     // [n][32] neighbour ids (kInvalidNode = empty slot), or null (own codes)
    const float* centroid;   // own codes: [D] centroid (zero padded)
    uint64_t n;
    uint32_t dim, D, epb;    // epb = edges per pass (LDS budget): 32, 16 or 8; 32 with the rows in HBM scratch
    float* scratch_f;        // rows in HBM (encode_edges_kernel<BW, true>): [grid][D][32] floats ...
    uint8_t* scratch_u;      // ... and [grid][D / 4][32] packed code dwords
    const float* signs;      // [3][D]
    float norm_factor, inv_sqrt_d;
    DevLayout L;
    uint8_t* blocks;         // edges: [n][L.stride] device blocks (pre-zeroed)
    // own codes: bit-packed planes + {nop, ip_qo}, file layout of the vertex header
    uint8_t* own;            // [n][own_stride]
    uint32_t own_stride, own_meta;   // bytes per vertex, offset of {nop, ip_qo}
    // parity hook: raw outputs of the edge encoder (optional)
    uint8_t* dbg_values;     // [n][32][D]
    float* dbg_aux;          // [n][32][3]
    uint32_t* dbg_pops;      // [n][32][2]
};

// LDS: pv[D] | rp[D] | work[D] | ctab[16] | nops[32] | rot[epb][D + 1] | codes[epb][D + 4]
__host__ __device__ inline size_t encode_edges_lds(uint32_t D, uint32_t epb) {
    return (size_t)(3 * D + 16 + 32) * 4 + (size_t)epb * (D + 1) * 4 + (size_t)epb * (D + 4);
}
// (rows in HBM scratch: only the parent, its rotation, the work row, the level table and the norms stay in LDS)
__host__ __device__ inline size_t encode_edges_lds_global(uint32_t D) { return (size_t)(3 * D + 16 + 32) * 4; }
inline uint32_t encode_edges_epb(uint32_t D) {
    uint32_t epb = 32;
    while (epb > 8 && encode_edges_lds(D, epb) > 72 * 1024) epb /= 2;
    return epb;
}

template <int BW, bool G>
__global__ __launch_bounds__(64) void encode_edges_kernel(EncodeArgsB a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t D = a.D, dim = a.dim, epb = G ? 32u : a.epb;
    float* pv = reinterpret_cast<float*>(smem);
    float* rp = pv + D;
    float* work = rp + D;
    float* ctab = work + D;
    float* nops = ctab + 16;
    // rows of the edges of one pass: LDS (stride D + 1 floats / D + 4 bytes per edge), or this workgroup's transposed
    // scratch in HBM
    float* rot = G ? a.scratch_f + (size_t)blockIdx.x * D * 32 : nops + 32;
    uint8_t* codes = G ? a.scratch_u + (size_t)blockIdx.x * D * 32 : reinterpret_cast<uint8_t*>(rot + (size_t)epb * (D + 1));
    const int lane = threadIdx.x;
    auto F = [&](uint32_t e, uint32_t d) -> float& { return G ? rot[(size_t)d * 32 + e] : rot[(size_t)e * (D + 1) + d]; };
    const LaneRow<G> myrow{G ? rot + (lane & 31) : rot + (size_t)(lane & 31) * (D + 1),
                           G ? codes + (size_t)(lane & 31) * 4 : codes + (size_t)(lane & 31) * (D + 4)};
    const bool own = a.nbr == nullptr;
    if (lane < 16) {
        const float K = (float)((1 << BW) - 1);
        ctab[lane] = (2.0f * (float)lane - K) / K;
    }
    // a workgroup = one vertex' edges, or 32 consecutive vertices' own codes
    const uint64_t units = own ? (a.n + 31) / 32 : a.n;
    for (uint64_t unit = blockIdx.x; unit < units; unit += gridDim.x) {
        __syncthreads();
        // ---- parent: the vertex (edges) or the centroid (own codes), and its rotation ----------
        for (uint32_t d = lane; d < D; d += 64) {
            const float v = own ? a.centroid[d] : a.x[unit * D + d];
            pv[d] = v;
            rp[d] = v;
        }
        __syncthreads();
        if (!own) rotate_scaled_lds(rp, a.signs, D, a.norm_factor, lane);
        uint32_t my_id = kInvalidNode;
        uint32_t cnt;
        if (own) {
            cnt = (uint32_t)((unit * 32 + 32 <= a.n) ? 32 : a.n - unit * 32);
            if (lane < 32 && (uint32_t)lane < cnt) my_id = (uint32_t)(unit * 32 + lane);
        } else {
            if (lane < 32) my_id = a.nbr[unit * 32 + lane];
            cnt = (uint32_t)__popcll(__ballot(my_id != kInvalidNode));    // valid ids form a prefix
        }
        for (uint32_t base = 0; base < cnt; base += epb) {
            const uint32_t m = cnt - base < epb ? cnt - base : epb;
            // differences (unnormalised) into the rows
            for (uint32_t e = 0; e < m; ++e) {
                const uint32_t vid = (uint32_t)__shfl((int)my_id, (int)(base + e));
                const float* v = a.x + (size_t)vid * D;
                for (uint32_t d = lane; d < D; d += 64) F(e, d) = d < dim ? v[d] - pv[d] : 0.0f;
            }
            __syncthreads();
            float nop = 0.0f;
            if ((uint32_t)lane < m) {
                nop = __builtin_sqrtf(seq_norm_sq<G>(myrow, dim));
                nops[lane] = nop;
            }
            __syncthreads();
            // normalise, rotate, scale: cooperative, one edge after the other
            const float neps = 1e-8f / (float)D;      // norm_epsilon
            for (uint32_t e = 0; e < m; ++e) {
                const float ne = nops[e];
                if (ne < neps) continue;               // degenerate edge: all-zero code, zero aux (uniform branch)
                const float inv = 1.0f / ne;
                for (uint32_t d = lane; d < D; d += 64) work[d] = F(e, d) * inv;
                __syncthreads();
                rotate_scaled_lds(work, a.signs, D, a.norm_factor, lane);
                for (uint32_t d = lane; d < D; d += 64) F(e, d) = work[d];
                __syncthreads();
            }
            // ---- one edge per lane: the sequential quantiser on its LDS row ------------------------
            LaneCode lc{0.0f, 0.0f, 0u, 0u};
            const bool live = (uint32_t)lane < m;
            const bool degenerate = live && nop < neps;
            if (live) {
                if (degenerate) {
                    for (uint32_t i = 0; i < D; i += 4) myrow.setU4(i, 0u);
                } else {
                    lc = quantize_lane<BW, G>(myrow, own ? nullptr : rp, ctab, D, a.inv_sqrt_d);
                }
            }
            __syncthreads();
            // ---- outputs --------------------------------------------------------------------------
            if (live) {
                const uint32_t slot = base + lane;
                if (own) {
                    uint8_t* o = a.own + (size_t)(unit * 32 + slot) * a.own_stride;
                    const uint32_t words = (D + 63) / 64;
                    for (uint32_t b = 0; b < (uint32_t)BW; ++b)
                        for (uint32_t by = 0; by < words * 8; ++by) {
                            uint32_t v = 0;
                            for (uint32_t t = 0; t < 8 && 8 * by + t < D; ++t)
                                v |= (uint32_t)((myrow.Ub(8 * by + t) >> (BW - 1 - b)) & 1) << t;
                            o[(size_t)b * words * 8 + by] = (uint8_t)v;
                        }
                    *reinterpret_cast<float*>(o + a.own_meta) = nop;
                    *reinterpret_cast<float*>(o + a.own_meta + 4) = lc.ip_qo;
                } else {
                    uint8_t* blk = a.blocks + unit * a.L.stride;
                    const uint32_t PW = a.L.PW;
                    const uint32_t vbits = D >= 32 ? 32 : D;
                    for (uint32_t b = 0; b < (uint32_t)BW; ++b)
                        for (uint32_t w = 0; w < PW; ++w) {
                            uint32_t v = 0;
                            for (uint32_t t = 0; t < vbits; ++t)
                                v |= (uint32_t)((myrow.Ub(32 * w + t) >> (BW - 1 - b)) & 1) << t;
                            const uint32_t tt = b * PW + w;
                            size_t off;
                            if (a.L.wide) {
                                const uint32_t ck = tt / 4, el = tt % 4;
                                const uint32_t hh = (a.L.NH == 2) ? ck / a.L.CPL : 0;
                                const uint32_t kk = (a.L.NH == 2) ? ck % a.L.CPL : ck;
                                off = ((size_t)(kk * a.L.NH * 32 + hh * 32 + slot) * 16 + el * 4);
                            } else {
                                off = ((size_t)tt * 32 + slot) * 4;
                            }
                            *reinterpret_cast<uint32_t*>(blk + off) = v;
                        }
                    uint4 aux = make_uint4(__float_as_uint(nop), __float_as_uint(lc.ip_qo), __float_as_uint(lc.ip_cp),
                                           (lc.msb_pop & 0xFFFFu) | ((lc.weighted_pop & 0xFFFFu) << 16));
                    if (degenerate) aux = make_uint4(__float_as_uint(nop), 0u, 0u, 0u);
                    reinterpret_cast<uint4*>(blk + a.L.aux_off)[slot] = aux;
                    if (a.dbg_values) {
                        for (uint32_t i = 0; i < D; ++i) a.dbg_values[((size_t)unit * 32 + slot) * D + i] = (uint8_t)myrow.Ub(i);
                        float* da = a.dbg_aux + ((size_t)unit * 32 + slot) * 3;
                        da[0] = nop; da[1] = degenerate ? 0.0f : lc.ip_qo; da[2] = degenerate ? 0.0f : lc.ip_cp;
                        a.dbg_pops[((size_t)unit * 32 + slot) * 2] = lc.msb_pop;
                        a.dbg_pops[((size_t)unit * 32 + slot) * 2 + 1] = lc.weighted_pop;
                    }
                }
            }
            __syncthreads();
        }
        if (!own && lane < 32) {
            uint8_t* blk = a.blocks + unit * a.L.stride;
            reinterpret_cast<uint32_t*>(blk + a.L.ids_off)[lane] = my_id;     // empty slots stay kInvalidNode
            if (lane == 0) *reinterpret_cast<uint32_t*>(blk + a.L.count_off) = cnt;
        }
    }
}

// ---- neighbour selection --------------------------------------------------------------------------
// Rule (graph/neighbor_selection.hpp:21-88): candidates in ascending distance to the vertex; a candidate c
// is occluded by an already selected e when
//     d(c, e) < la * d(c, v) + (err_c + err_e) - (la - 1) * tau,     la = clamp(alpha * sqrt(C / R), 1, alpha_max)
// and kept otherwise; if fewer than R survive, the nearest occluded ones fill the list.
struct SelectArgs {
    const float* x;           // [n][D]
    const uint32_t* fwd;      // [rows][32] forward candidates per row (ids into x; kInvalidNode = none)
    const uint64_t* rev_off;  // [rows + 1] CSR of reverse candidates (row indices), or null
    const uint32_t* rev;      // reverse candidates: ROW indices (translated through row_ids)
    const uint32_t* row_ids;  // [rows] vertex id of each row, or null (row == vertex)
    const float* err;         // [n] per-vertex error margin term (err_tol * |v - centroid|), or null
    uint64_t rows;
    uint32_t D, R;
    float alpha, tau, alpha_max;
    uint32_t* out;            // [rows][32] selected neighbour ids (vertex ids), kInvalidNode padded
    uint32_t* out_cnt;        // [rows]
};

constexpr int kSelCap = 128;   // candidates considered per vertex: two per lane

// LDS: cvec[D] | cid[128] | cd[128] | sel[32] | selerr[32]
__host__ __device__ inline size_t select_lds(uint32_t D) { return (size_t)D * 4 + kSelCap * 8 + 32 * 8; }

__global__ __launch_bounds__(64) void select_kernel(SelectArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* cvec = reinterpret_cast<float*>(smem);
    uint32_t* cid = reinterpret_cast<uint32_t*>(cvec + a.D);
    float* cd = reinterpret_cast<float*>(cid + kSelCap);
    uint32_t* sel = reinterpret_cast<uint32_t*>(cd + kSelCap);
    float* selerr = reinterpret_cast<float*>(sel + 32);
    const int lane = threadIdx.x;
    const uint32_t D = a.D, R = a.R;
    const float FMAX = 3.402823466e+38f;
    for (uint64_t row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const uint32_t vtx = a.row_ids ? a.row_ids[row] : (uint32_t)row;
        __syncthreads();
        for (uint32_t d = lane; d < D; d += 64) cvec[d] = a.x[(size_t)vtx * D + d];
        // ---- gather: forward list first, then the nearest reverse edges; two entries per lane --------
        uint32_t id0 = kInvalidNode, id1 = kInvalidNode;
        if (lane < 32) id0 = a.fwd[row * 32 + lane];
        uint64_t rb = 0, re = 0;
        if (a.rev_off) { rb = a.rev_off[row]; re = a.rev_off[row + 1]; }
        __syncthreads();
        float d0 = FMAX, d1 = FMAX;
        auto exact_dists = [&]() {
            // exact squared distances of all 128 slots to the vertex (8 lanes per candidate, 8 per pass)
            for (int base = 0; base < kSelCap; base += 8) {
                const int slot = base + (lane >> 3);        // one half of the slots per pass (base is uniform)
                const uint32_t cand = base < 64 ? (uint32_t)__shfl((int)id0, slot) : (uint32_t)__shfl((int)id1, slot - 64);
                const bool have = cand != kInvalidNode && cand != vtx;
                const float dd = group_l2sq8(cvec, a.x + (size_t)(have ? cand : vtx) * D, D, lane & 7);
                if ((lane & 7) == 0) { cd[slot] = have ? dd : FMAX; cid[slot] = have ? cand : kInvalidNode; }
            }
            __syncthreads();
            id0 = cid[lane]; d0 = cd[lane];
            id1 = cid[lane + 64]; d1 = cd[lane + 64];
        };
        auto rank_sort = [&]() {
            // sort the 128 slots by (distance, id); duplicates of an id keep their first copy only
            uint32_t r0 = 0, r1 = 0;
            bool dup0 = false, dup1 = false;
#pragma unroll 4
            for (int j = 0; j < 64; ++j) {
                const float ej0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d0), j));
                const uint32_t ij0 = (uint32_t)__builtin_amdgcn_readlane((int)id0, j);
                const float ej1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d1), j));
                const uint32_t ij1 = (uint32_t)__builtin_amdgcn_readlane((int)id1, j);
                // entry (j, half 0) against mine
                const bool b00 = ej0 < d0 || (ej0 == d0 && (ij0 < id0 || (ij0 == id0 && j < lane)));
                const bool b01 = ej0 < d1 || (ej0 == d1 && (ij0 < id1 || ij0 == id1));
                const bool b10 = ej1 < d0 || (ej1 == d0 && ij1 < id0);
                const bool b11 = ej1 < d1 || (ej1 == d1 && (ij1 < id1 || (ij1 == id1 && j < lane)));
                r0 += (b00 ? 1u : 0u) + (b10 ? 1u : 0u);
                r1 += (b01 ? 1u : 0u) + (b11 ? 1u : 0u);
                dup0 |= (ij0 == id0 && b00) || (ij1 == id0 && b10);
                dup1 |= (ij0 == id1 && b01) || (ij1 == id1 && b11);
            }
            __syncthreads();
            cid[r0] = dup0 ? kInvalidNode : id0; cd[r0] = dup0 ? FMAX : d0;
            cid[r1] = dup1 ? kInvalidNode : id1; cd[r1] = dup1 ? FMAX : d1;
            __syncthreads();
            id0 = cid[lane]; d0 = cd[lane];
            id1 = cid[lane + 64]; d1 = cd[lane + 64];
        };
        if (re - rb <= 96) {
            // forward list in slots 0..31, reverse edges in slots 32..127
            const uint64_t take = re - rb;
            if (lane >= 32 && (uint64_t)(lane - 32) < take) {
                const uint32_t r = a.rev[rb + (lane - 32)];
                id0 = a.row_ids ? a.row_ids[r] : r;
            }
            if ((uint64_t)(lane + 32) < take) {
                const uint32_t r = a.rev[rb + lane + 32];
                id1 = a.row_ids ? a.row_ids[r] : r;
            }
            exact_dists();
            rank_sort();
        } else {
            // a hub: stream all reverse edges through the upper 64 slots, keeping the nearest 64 candidates
            // seen so far in the lower ones -- the result does not depend on the order of the reverse list
            while (rb < re) {
                const uint64_t take = re - rb < 64 ? re - rb : 64;
                id1 = kInvalidNode;
                if ((uint64_t)lane < take) {
                    const uint32_t r = a.rev[rb + lane];
                    id1 = a.row_ids ? a.row_ids[r] : r;
                }
                rb += take;
                exact_dists();
                rank_sort();
                rank_sort();      // duplicates became empty entries in the middle: pack before cutting
            }
            id1 = kInvalidNode;
            d1 = FMAX;
        }
        // entries with FMAX distance sort last; duplicates were turned into such entries, so one more
        // pass packs the real ones to the front
        rank_sort();
        const uint32_t C = (uint32_t)__popcll(__ballot(id0 != kInvalidNode)) + (uint32_t)__popcll(__ballot(id1 != kInvalidNode));
        uint32_t nsel = 0;
        if (C <= R) {
            // nothing to prune (select returns the sorted candidates)
            if ((uint32_t)lane < C && lane < 32) a.out[row * 32 + lane] = id0;
            if ((uint32_t)lane >= C && lane < 32) a.out[row * 32 + lane] = kInvalidNode;
            if (lane == 0) a.out_cnt[row] = C;
            continue;
        }
        float la = a.alpha * __builtin_sqrtf((float)C / (float)R);
        const float amax = a.alpha_max > 0.0f ? a.alpha_max : 2.0f * a.alpha;
        la = la < 1.0f ? 1.0f : (la > amax ? amax : la);
        unsigned long long taken_lo = 0, taken_hi = 0;       // candidate slots already selected
        for (uint32_t ci = 0; ci < C && nsel < R; ++ci) {
            const uint32_t c = cid[ci];
            const float dcq = cd[ci];
            const float errc = a.err ? a.err[c] : 0.0f;
            __syncthreads();
            for (uint32_t d = lane; d < D; d += 64) cvec[d] = a.x[(size_t)c * D + d];
            __syncthreads();
            bool occluded = false;
            for (uint32_t base = 0; base < nsel; base += 8) {
                const uint32_t j = base + (lane >> 3);
                const bool have = j < nsel;
                const uint32_t e = have ? sel[j] : c;
                const float dce = group_l2sq8(cvec, a.x + (size_t)e * D, D, lane & 7);
                const float thr = la * dcq + (errc + (have ? selerr[j] : 0.0f)) - (la - 1.0f) * a.tau;
                occluded |= have && dce < thr;
            }
            if (!__any(occluded)) {
                if (lane == 0) { sel[nsel] = c; selerr[nsel] = errc; }
                if (ci < 64) taken_lo |= 1ull << ci; else taken_hi |= 1ull << (ci - 64);
                ++nsel;
            }
        }
        __syncthreads();
        // fill with the nearest candidates not yet taken
        for (uint32_t ci = 0; ci < C && nsel < R; ++ci) {
            const bool taken = ci < 64 ? (taken_lo >> ci) & 1ull : (taken_hi >> (ci - 64)) & 1ull;
            if (taken) continue;
            if (lane == 0) sel[nsel] = cid[ci];
            ++nsel;
        }
        __syncthreads();
        if (lane < 32) a.out[row * 32 + lane] = (uint32_t)lane < nsel ? sel[lane] : kInvalidNode;
        if (lane == 0) a.out_cnt[row] = nsel;
    }
}

// ---- calibration samples ----------------------------------------------------------------------------
// One wave per sample query: hop from the given start vertex to its nearest neighbour if that is nearer
// (one greedy step, as the reference's sampler does), then evaluate the FastScan estimator of every edge
// of that vertex next to the exact quantities.  Record per (sample, slot):
//   {nop, ipc = ip_est_raw - ip_cp, |ip_qo| floored at 1e-10, true <q - p, o - p> / nop, exact |q - o|^2, ip_qo}
struct CalibArgs {
    const uint8_t* blocks;
    const float* raw;          // [n][D]
    DevLayout L;
    uint64_t n;
    const float* queries;      // [ns][D] padded raw queries
    const uint4* qmasks;       // [ns][PW]
    const QueryHeader* qhdr;   // [ns]  (A, B, C of the encoded query)
    const uint32_t* start;     // [ns] start vertex
    uint32_t ns;
    float* rec;                // [ns][32][6]
    uint32_t* rec_cnt;         // [ns] valid slots
    float* dqp_out;            // [ns] exact squared distance to the chosen vertex (the sample's 1-hop NN distance)
};

template <int BW>
__global__ __launch_bounds__(64) void calib_kernel(CalibArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t D = a.L.D, PW = a.L.PW;
    uint4* qm = reinterpret_cast<uint4*>(smem);
    float* qv = reinterpret_cast<float*>(smem + (size_t)PW * 16);
    float* pv = qv + D;
    const int lane = threadIdx.x;
    for (uint32_t s = blockIdx.x; s < a.ns; s += gridDim.x) {
        __syncthreads();
        for (uint32_t w = lane; w < PW; w += 64) qm[w] = a.qmasks[(size_t)s * PW + w];
        for (uint32_t d = lane; d < D; d += 64) qv[d] = a.queries[(size_t)s * D + d];
        __syncthreads();
        uint32_t parent = a.start[s];
        float best = group_l2sq8(qv, a.raw + (size_t)parent * D, D, lane & 7);
        best = __shfl(best, 0);
        {
            const uint8_t* blk = a.blocks + (size_t)parent * a.L.stride;
            const uint32_t nid = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[lane & 31];
            float cand = 3.402823466e+38f;
            uint32_t cand_id = kInvalidNode, cand_pos = 64;
            for (int base = 0; base < 32; base += 8) {
                const int pos = base + (lane >> 3);
                const uint32_t nb = (uint32_t)__shfl((int)nid, pos);
                const bool have = nb != kInvalidNode;
                const float dd = group_l2sq8(qv, a.raw + (size_t)(have ? nb : parent) * D, D, lane & 7);
                if (have && dd < cand) { cand = dd; cand_id = nb; cand_pos = pos; }
            }
            for (int o = 8; o < 64; o <<= 1) {
                const float od = __shfl_xor(cand, o);
                const uint32_t oi = __shfl_xor(cand_id, o);
                const uint32_t op = __shfl_xor(cand_pos, o);
                if (od < cand || (od == cand && op < cand_pos)) { cand = od; cand_id = oi; cand_pos = op; }
            }
            cand = __shfl(cand, 0); cand_id = (uint32_t)__shfl((int)cand_id, 0);
            if (cand_id != kInvalidNode && cand < best) { best = cand; parent = cand_id; }
        }
        const uint8_t* blk = a.blocks + (size_t)parent * a.L.stride;
        for (uint32_t d = lane; d < D; d += 64) pv[d] = a.raw[(size_t)parent * D + d];
        __syncthreads();
        LaneEst v;
        load_block<BW, 0>(blk, a.L, qm, lane, v);
        const uint32_t nid = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[lane & 31];
        const QueryHeader hd = a.qhdr[s];
        float ipa;
        if constexpr (BW == 1) {
            ipa = hd.A * (float)v.nbit + hd.B * (float)v.pop + hd.C;
        } else {
            const float invK = 1.0f / (float)((1u << BW) - 1);
            ipa = hd.A * invK * (float)v.nbit + hd.B * invK * (float)v.wpop + hd.C;
        }
        const float ipc = ipa - v.ip_cp;
        const float ipq = __builtin_fabsf(v.ip_qo) > kEpsMedium ? __builtin_fabsf(v.ip_qo) : kEpsMedium;
        const float nop = v.nop > kEpsSmall ? v.nop : kEpsSmall;
        // exact quantities, 8 neighbours per pass
        float tip = 0.0f, dqo = 0.0f;
        for (int base = 0; base < 32; base += 8) {
            const int pos = base + (lane >> 3);
            const uint32_t nb = (uint32_t)__shfl((int)nid, pos);
            const bool have = nb != kInvalidNode;
            const float* ov = a.raw + (size_t)(have ? nb : parent) * D;
            float t = 0.0f, l2 = 0.0f;
            for (uint32_t d = lane & 7; d < D; d += 8) {
                const float o = ov[d], q = qv[d], p = pv[d];
                t = __fmaf_rn(q - p, o - p, t);
                l2 = __fmaf_rn(q - o, q - o, l2);
            }
            t = group_reduce8(t);
            l2 = group_reduce8(l2);
            // lane `pos` of the lower half owns neighbour `pos`
            const float tt = __shfl(t, (lane & 31) >= base && (lane & 31) < base + 8 ? ((lane & 31) - base) * 8 : 0);
            const float ll = __shfl(l2, (lane & 31) >= base && (lane & 31) < base + 8 ? ((lane & 31) - base) * 8 : 0);
            if ((lane & 31) >= base && (lane & 31) < base + 8) { tip = tt; dqo = ll; }
        }
        const bool valid = lane < 32 && nid != kInvalidNode;
        if (lane < 32) {
            float* r = a.rec + ((size_t)s * 32 + lane) * 6;
            r[0] = nop; r[1] = ipc; r[2] = ipq; r[3] = valid ? tip / nop : 0.0f; r[4] = dqo; r[5] = v.ip_qo;
        }
        const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid));
        if (lane == 0) { a.rec_cnt[s] = nvalid; a.dqp_out[s] = best; }
    }
}

}  // namespace build
}  // namespace cph
