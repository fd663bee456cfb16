// device_fastscan.h — wave-level FastScan RaBitQ estimator for gfx950 (wave64).
//
// Replaces, on the GPU, the reference's AVX2 kernels in distance/fastscan_kernel.hpp:
//   compute_inner_products            :17-87     one bit-plane, 32 neighbours
//   compute_msb_only_inner_products   :349-368   2*S0 + S1
//   compute_nbit_inner_products       :197-217   sum_b 2^(BW-1-b) * S_b, and S0
//   convert_to_distances_with_bounds  :89-194    1-bit epilogue
//   convert_msb_to_lower_bounds       :371-425   stage-1 lower bound
//   convert_nbit_to_distances_with_bounds :220-346  N-bit epilogue
//
// Formulation.  The reference's LUT holds, per 4-dim segment, every subset sum of the
// query's 4-bit scalars q_u (encoder/rabitq_encoder.hpp:119-130), and VPSHUFB gathers
// lut[seg][code nibble].  Summed over segments this is exactly  S_b = sum_d q_u[d] *
// bit_b[d].  With the query scalars bit-sliced into four D-bit masks Q_j (bit j of q_u),
//      S_b = sum_j 2^j * popcount(code_plane_b AND Q_j)
// — the identical integer, computed with v_and + v_bcnt on 32 dims per instruction, no
// per-lane table gather.  The query masks (16 B per 32 dims) sit in LDS and are read as
// wave-uniform broadcasts.  One wave handles one vertex' 32-neighbour block: lane
// (h = lane>>5, i = lane&31) owns half of neighbour i's code dwords, so each load
// instruction of the wave reads 1 KiB contiguous.
#pragma once
#include <hip/hip_runtime.h>

#include "cph_core.h"

namespace cph {

struct QP {  // RaBitQQuery scalars (core/codes.hpp:79-93)
    float A, B, C, affine_a, affine_b, floor, slack;
};

struct LaneEst {      // per-lane (neighbour i = lane & 31) results, valid on all 64 lanes
    uint32_t nbit;    // 1-bit: plane sum; N-bit: weighted N-bit sum
    uint32_t msb;     // plane-0 sum S0
    uint32_t msb2;    // 2*S0 + S1 (BW >= 2), S0 (BW == 1)
    float nop, ip_qo, ip_cp;
    uint32_t pop, wpop;
};

__device__ __forceinline__ float vmaxf(float a, float b) { return a > b ? a : b; }  // _mm256_max_ps
__device__ __forceinline__ float vminf(float a, float b) { return a < b ? a : b; }  // _mm256_min_ps

// popcount(x) + acc in ONE instruction: v_bcnt_u32_b32 has an accumulate operand.  Written as inline assembly
// because the compiler, given `acc += __popc(x)`, emits the popcounts with a zero addend and then re-associates
// the sums into a tree of v_add3 (shorter chains, a quarter more instructions -- and this kernel is bound by
// instruction issue, not by latency: four independent accumulators are interleaved anyway).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ void acc4(uint32_t c, uint4 q, uint32_t& a0, uint32_t& a1,
                                     uint32_t& a2, uint32_t& a3) {
    a0 = bcnt_acc(c & q.x, a0);
    a1 = bcnt_acc(c & q.y, a1);
    a2 = bcnt_acc(c & q.z, a2);
    a3 = bcnt_acc(c & q.w, a3);
}

// Integer sums of one block.  qm = LDS, uint4 per 32-dim word: {Q0,Q1,Q2,Q3}.
// SD = compile-time D (0 = runtime D from the layout).
// Both halves of the wave see both values: lo = x of lane (l & 31), hi = x of lane (l | 32).
// One v_permlane32_swap (gfx950) instead of a ds_bpermute through the LDS crossbar.
__device__ __forceinline__ void half_pair(uint32_t x, uint32_t& lo, uint32_t& hi) {
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    lo = r[0];
    hi = r[1];
}

template <int BW, int SD>
__device__ __forceinline__ void block_sums(const uint8_t* __restrict__ blk, const DevLayout& L,
                                           const uint4* qm, int lane, uint32_t& nbit,
                                           uint32_t& msb, uint32_t& msb2) {
    const int h = lane >> 5;
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : L.PW;
    const bool wide = SD ? (SD >= 128) : (L.wide != 0);
    if (wide) {
        const uint4* cp = reinterpret_cast<const uint4*>(blk) + lane;
        if constexpr (BW >= 2) {
            constexpr int PPH = BW / 2;  // planes per lane half (NH == 2 always here)
            const uint32_t G = PW >> 2;  // 16-B chunks per plane
            uint32_t S[PPH];
#pragma unroll
            for (int pl = 0; pl < PPH; ++pl) {
                uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 2
                for (uint32_t g = 0; g < G; ++g) {
                    uint4 c = cp[(pl * G + g) * 64];
                    acc4(c.x, qm[4 * g + 0], a0, a1, a2, a3);
                    acc4(c.y, qm[4 * g + 1], a0, a1, a2, a3);
                    acc4(c.z, qm[4 * g + 2], a0, a1, a2, a3);
                    acc4(c.w, qm[4 * g + 3], a0, a1, a2, a3);
                }
                S[pl] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
            }
            // plane b = h*PPH + pl carries weight 2^(BW-1-b)
            uint32_t part = 0;
#pragma unroll
            for (int pl = 0; pl < PPH; ++pl) part += S[pl] << (BW - 1 - (h * PPH + pl));
            uint32_t plo, phi, s0, s0hi;
            half_pair(part, plo, phi);
            nbit = plo + phi;
            half_pair(S[0], s0, s0hi);          // plane 0 lives in the h = 0 half
            if constexpr (BW == 2) {
                msb = s0;
                msb2 = 2 * s0 + s0hi;           // plane 1 = S[0] of the h = 1 half
            } else {
                uint32_t s1, s1hi;
                half_pair(S[1], s1, s1hi);      // planes 0,1 both in the h = 0 half
                msb = s0;
                msb2 = 2 * s0 + s1;
            }
        } else {
            const uint32_t NH = SD ? ((SD / 32) / 4 >= 2 ? 2 : 1) : L.NH;
            const uint32_t CPL = SD ? ((SD / 32) / 4 / NH) : L.CPL;
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            if ((uint32_t)h < NH) {
                const uint4* cq = reinterpret_cast<const uint4*>(blk) + (lane & (NH * 32 - 1));
#pragma unroll 2
                for (uint32_t k = 0; k < CPL; ++k) {
                    uint4 c = cq[k * NH * 32];
                    uint32_t w0 = (h * CPL + k) * 4;
                    acc4(c.x, qm[w0 + 0], a0, a1, a2, a3);
                    acc4(c.y, qm[w0 + 1], a0, a1, a2, a3);
                    acc4(c.z, qm[w0 + 2], a0, a1, a2, a3);
                    acc4(c.w, qm[w0 + 3], a0, a1, a2, a3);
                }
            }
            uint32_t s = a0 + 2 * a1 + 4 * a2 + 8 * a3, slo, shi;
            half_pair(s, slo, shi);
            s = (NH == 2) ? slo + shi : slo;
            nbit = msb = msb2 = s;
        }
    } else {
        // small D (16/32/64): dwords ordered [plane*PW + w][neighbour]; both lane halves
        // compute their neighbour redundantly.
        const uint32_t* cw = reinterpret_cast<const uint32_t*>(blk) + (lane & 31);
        uint32_t S[BW];
#pragma unroll
        for (int b = 0; b < BW; ++b) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (uint32_t w = 0; w < PW; ++w) acc4(cw[(b * PW + w) * 32], qm[w], a0, a1, a2, a3);
            S[b] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
        }
        uint32_t t = 0;
#pragma unroll
        for (int b = 0; b < BW; ++b) t += S[b] << (BW - 1 - b);
        nbit = t;
        msb = S[0];
        if constexpr (BW >= 2) msb2 = 2 * S[0] + S[1]; else msb2 = S[0];
    }
}

// Issue/reduce split: with a compile-time D the code and aux loads of a block are issued
// up front (so that the caller can put other independent loads and the estimated-set atomic
// between issue and use); with a runtime D the loads happen inside reduce().
template <int BW, int SD>
struct BlockLoads {
    static constexpr bool kStatic = (SD >= 128);
    static constexpr int kT = kStatic ? BW * (SD / 32) : 4;          // dwords per neighbour
    static constexpr int kNH = (kT / 4 >= 2) ? 2 : 1;
    static constexpr int kCPL = kStatic ? kT / 4 / kNH : 1;          // 16-B chunks per lane
    uint4 c[kCPL];
    uint4 aux;

    // aux offset known at compile time (see cph_core.h StaticLayout)
    __device__ __forceinline__ void issue_static(const uint8_t* __restrict__ blk, int lane) {
        static_assert(kStatic, "static D only");
        const uint4* cp = reinterpret_cast<const uint4*>(blk) + (lane & (kNH * 32 - 1));
#pragma unroll
        for (int k = 0; k < kCPL; ++k) c[k] = cp[k * kNH * 32];
        aux = reinterpret_cast<const uint4*>(blk + StaticLayout<BW, SD>::kAuxOff)[lane & 31];
    }

    __device__ __forceinline__ void issue(const uint8_t* __restrict__ blk, const DevLayout& L, int lane) {
        if constexpr (kStatic) {
            const uint4* cp = reinterpret_cast<const uint4*>(blk) + (lane & (kNH * 32 - 1));
#pragma unroll
            for (int k = 0; k < kCPL; ++k) c[k] = cp[k * kNH * 32];
        }
        aux = reinterpret_cast<const uint4*>(blk + L.aux_off)[lane & 31];
    }

    // Makes the compiler retire (wait for) the loads of issue() at this point of the program.
    __device__ __forceinline__ void retire() {
        if constexpr (kStatic) {
#pragma unroll
            for (int k = 0; k < kCPL; ++k)
                asm volatile("" : "+v"(c[k].x), "+v"(c[k].y), "+v"(c[k].z), "+v"(c[k].w));
        }
        asm volatile("" : "+v"(aux.x), "+v"(aux.y), "+v"(aux.z), "+v"(aux.w));
    }

    __device__ __forceinline__ void reduce(const uint8_t* __restrict__ blk, const DevLayout& L,
                                           const uint4* qm, int lane, LaneEst& o) {
        if constexpr (kStatic) {
            const int h = lane >> 5;
            constexpr int PW = SD / 32;
            if constexpr (BW >= 2) {
                constexpr int PPH = BW / 2;
                constexpr int G = PW / 4;
                uint32_t S[PPH];
#pragma unroll
                for (int pl = 0; pl < PPH; ++pl) {
                    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const uint4 cc = c[pl * G + g];
                        acc4(cc.x, qm[4 * g + 0], a0, a1, a2, a3);
                        acc4(cc.y, qm[4 * g + 1], a0, a1, a2, a3);
                        acc4(cc.z, qm[4 * g + 2], a0, a1, a2, a3);
                        acc4(cc.w, qm[4 * g + 3], a0, a1, a2, a3);
                    }
                    S[pl] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
                }
                uint32_t part = 0;
#pragma unroll
                for (int pl = 0; pl < PPH; ++pl) part += S[pl] << (BW - 1 - (h * PPH + pl));
                uint32_t plo, phi, s0, s0hi;
                half_pair(part, plo, phi);
                o.nbit = plo + phi;
                half_pair(S[0], s0, s0hi);      // plane 0 lives in the h = 0 half
                if constexpr (BW == 2) {
                    o.msb = s0;
                    o.msb2 = 2 * s0 + s0hi;     // plane 1 = S[0] of the h = 1 half
                } else {
                    uint32_t s1, s1hi;
                    half_pair(S[1], s1, s1hi);  // planes 0,1 both in the h = 0 half
                    o.msb = s0;
                    o.msb2 = 2 * s0 + s1;
                }
            } else {
                uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                if (h < kNH) {
#pragma unroll
                    for (int k = 0; k < kCPL; ++k) {
                        const uint4 cc = c[k];
                        const uint32_t w0 = (h * kCPL + k) * 4;
                        acc4(cc.x, qm[w0 + 0], a0, a1, a2, a3);
                        acc4(cc.y, qm[w0 + 1], a0, a1, a2, a3);
                        acc4(cc.z, qm[w0 + 2], a0, a1, a2, a3);
                        acc4(cc.w, qm[w0 + 3], a0, a1, a2, a3);
                    }
                }
                uint32_t sv = a0 + 2 * a1 + 4 * a2 + 8 * a3, slo, shi;
                half_pair(sv, slo, shi);
                sv = (kNH == 2) ? slo + shi : slo;
                o.nbit = o.msb = o.msb2 = sv;
            }
        } else {
            block_sums<BW, SD>(blk, L, qm, lane, o.nbit, o.msb, o.msb2);
        }
        o.nop = __uint_as_float(aux.x);
        o.ip_qo = __uint_as_float(aux.y);
        o.ip_cp = __uint_as_float(aux.z);
        o.pop = aux.w & 0xFFFFu;
        o.wpop = aux.w >> 16;
    }
};

template <int BW, int SD>
__device__ __forceinline__ void load_block(const uint8_t* __restrict__ blk, const DevLayout& L,
                                           const uint4* qm, int lane, LaneEst& o) {
    BlockLoads<BW, SD> b;
    b.issue(blk, L, lane);
    b.reduce(blk, L, qm, lane, o);
}

// ---- short neighbour lists: the reference's scalar tails ---------------------------------------
// The epilogues convert_to_distances_with_bounds / convert_nbit_to_distances_with_bounds take the AVX2 vector path for
// lanes below 8 * (count / 8) and a scalar loop for the remainder (fastscan_kernel.hpp:174-193, :324-345).  GCC
// contracts that loop into a different rounding sequence for the inner-product estimates (found against the compiled
// reference, pinned by tests/golden F/../c<count> vectors; see oracle/cph_oracle.cpp convert_1bit / convert_nbit):
//     vector  fma(A, s, fma(B, pc, C))            tail, N-bit and 1-bit sums   fma(B, pc, A * s) + C
//                                                 tail, plane-0 (msb) bound    fma(A, s, B * pc) + C
// and to the vector path's sequence for everything behind them.  count is the number of valid ids of the block (the
// repacker sets slots >= count to kInvalidNode); `any` is wave-uniform, so full blocks -- every block the reference's
// builder and ours write -- pay one scalar branch.
struct TailLanes {
    bool any;    // count % 8 != 0 (wave-uniform)
    bool mine;   // this lane's neighbour index >= 8 * (count / 8)
};
// valid: this lane's neighbour slot holds an id.  Both lane halves carry neighbour (lane & 31): the low half's ballot.
__device__ __forceinline__ TailLanes tail_lanes(bool valid, int lane) {
    const uint32_t count = (uint32_t)__popc((uint32_t)(__ballot(valid) & 0xFFFFFFFFull));
    TailLanes t;
    t.any = (count & 7u) != 0u;
    t.mine = (uint32_t)(lane & 31) >= (count & ~7u);
    return t;
}
__device__ __forceinline__ TailLanes no_tail() { return TailLanes{false, false}; }
// A * s + B * pc + C as the vector path / the scalar tail of the N-bit (and 1-bit) estimate evaluates it
__device__ __forceinline__ float ip_approx_est(float A, float s, float B, float pc, float C, const TailLanes& t) {
    float r = __fmaf_rn(A, s, __fmaf_rn(B, pc, C));
    if (__builtin_expect(t.any, 0)) {
        const float u = __fmaf_rn(B, pc, A * s) + C;
        r = t.mine ? u : r;
    }
    return r;
}
// ... and of the plane-0 numerator of the stage-2 lower bound (N-bit only; the 1-bit bound shares the estimate's)
__device__ __forceinline__ float ip_approx_msb(float A, float s, float B, float pc, float C, const TailLanes& t) {
    float r = __fmaf_rn(A, s, __fmaf_rn(B, pc, C));
    if (__builtin_expect(t.any, 0)) {
        const float u = __fmaf_rn(A, s, B * pc) + C;
        r = t.mine ? u : r;
    }
    return r;
}

// Shared tail of the AVX2 vector paths fastscan_kernel.hpp:148-169 / :287-317.
__device__ __forceinline__ void est_and_lower(const QP& q, float ip_est_approx, float ip_lb_approx,
                                              float nop, float ip_qo, float ip_cp, float dqp,
                                              float sqrt_dqp, float& est, float& lower) {
    float ipq = vmaxf(ip_qo, q.floor);
    bool good = ipq > kEpsMedium;
    float e = good ? (ip_est_approx - ip_cp) / ipq : 0.0f;
    e = __fmaf_rn(q.affine_a, e, q.affine_b);
    float dist = __fmaf_rn(nop, nop, dqp);
    dist = __fmaf_rn(-(2.0f * nop), e, dist);
    est = vmaxf(dist, 0.0f);
    float m = good ? (ip_lb_approx - ip_cp) / ipq : 0.0f;
    m = __fmaf_rn(q.affine_a, m, q.affine_b);
    float cosu = (m + q.slack) / vmaxf(sqrt_dqp, kEpsMedium);
    cosu = vminf(vmaxf(cosu, -1.0f), 1.0f);
    float lo = __fmaf_rn(nop, nop, dqp);
    lo = __fmaf_rn(-((2.0f * nop) * sqrt_dqp), cosu, lo);
    lo = vmaxf(lo, 0.0f);
    lower = good ? lo : 0.0f;
}

// Stage-1 lower bound, op-for-op as the compiled reference evaluates
// convert_msb_to_lower_bounds (fastscan_kernel.hpp:403-424; see oracle/cph_oracle.cpp).
template <int BW>
__device__ __forceinline__ float stage1_lower(const QP& q, const LaneEst& v, float dqp,
                                              float sqrt_dqp) {
    if (dqp < kEpsSmall) return 0.0f;
    const float invk = (BW >= 2) ? (1.0f / 3.0f) : 1.0f;
    float A = q.A * invk, B = q.B * invk;
    float ipq = (v.ip_qo < q.floor) ? q.floor : v.ip_qo;
    if (ipq <= kEpsMedium) return 0.0f;
    float ipa = __fmaf_rn(B, (float)v.pop, A * (float)v.msb2) + q.C;
    float e = (ipa - v.ip_cp) / ipq;
    e = __fmaf_rn(e, q.affine_a, q.affine_b);
    float c = (e + q.slack) / sqrt_dqp;
    c = (c < -1.0f) ? -1.0f : ((1.0f < c) ? 1.0f : c);
    float lo = __fmaf_rn(-c, (v.nop + v.nop) * sqrt_dqp, __fmaf_rn(v.nop, v.nop, dqp));
    return (lo < 0.0f) ? 0.0f : lo;
}

// Stage-2 (full) estimate + lower bound.
template <int BW>
__device__ __forceinline__ void stage2_est(const QP& q, const LaneEst& v, float dqp,
                                           float sqrt_dqp, float& est, float& lower,
                                           const TailLanes& tl) {
    if (dqp < kEpsSmall) {  // fastscan_kernel.hpp:112-119 / :249-256 (scalar, fused by GCC)
        est = __fmaf_rn(v.nop, v.nop, dqp);
        lower = 0.0f;
        return;
    }
    if constexpr (BW == 1) {
        float ipa = ip_approx_est(q.A, (float)v.nbit, q.B, (float)v.pop, q.C, tl);
        est_and_lower(q, ipa, ipa, v.nop, v.ip_qo, v.ip_cp, dqp, sqrt_dqp, est, lower);
    } else {
        constexpr float K = (float)((1u << BW) - 1);
        constexpr float invK = 1.0f / K;
        float An = q.A * invK, Bn = q.B * invK;
        float ipn = ip_approx_est(An, (float)v.nbit, Bn, (float)v.wpop, q.C, tl);
        float ipm = ip_approx_msb(q.A, (float)v.msb, q.B, (float)v.pop, q.C, tl);
        est_and_lower(q, ipn, ipm, v.nop, v.ip_qo, v.ip_cp, dqp, sqrt_dqp, est, lower);
    }
}

// The two lower bounds of the N-bit path in ONE instruction stream: the stage-1 bound (stage1_lower above,
// fastscan_kernel.hpp:403-424) on lanes 0..31 and the stage-2 bound (the `lower` half of est_and_lower,
// :287-317) on lanes 32..63, for the same neighbour i = lane & 31.  Past their numerators the two are the same
// sequence of IEEE operations -- (num - ip_cp) / ipq, the affine map (a product commutes), (.. + slack) / sqrt(dqp)
// (stage 2 divides by max(sqrt(dqp), 1e-10), which IS sqrt(dqp) once dqp >= 1e-12), the clamp to [-1, 1] (equal for
// every non-NaN value, and none of the operands can be NaN), nop^2 + dqp - c * 2 nop sqrt(dqp) (the sign moves
// between the factors of an exact product), max(.., 0), and 0 when ipq <= 1e-10 -- so each half of the wave does one of
// them and a v_permlane32_swap hands both to both halves: two IEEE divisions and a dozen other instructions less per
// block than evaluating them one after the other on 64 lanes.  (Zero results may differ in sign from the two
// originals -- `lo < 0 ? 0 : lo` against max(lo, 0) -- which no comparison downstream can see; the block hook
// and the streaming kernels, whose outputs are compared bit for bit, keep the separate functions.)
// Requires dqp >= kEpsSmall (the caller tests that wave-uniform fact).
template <int BW>
__device__ __forceinline__ void lower_bounds_split(const QP& q, const LaneEst& v, float dqp, float sqrt_dqp,
                                                   int lane, float& lo_stage1, float& lo_stage2, const TailLanes& tl) {
    static_assert(BW >= 2, "N-bit path");
    const float invk = 1.0f / 3.0f;
    const float n1 = __fmaf_rn(q.B * invk, (float)v.pop, (q.A * invk) * (float)v.msb2) + q.C;   // (a scalar loop: no tail of its own)
    const float n2 = ip_approx_msb(q.A, (float)v.msb, q.B, (float)v.pop, q.C, tl);
    const float num = lane >= 32 ? n2 : n1;
    const float ipq = (v.ip_qo < q.floor) ? q.floor : v.ip_qo;
    float e = (num - v.ip_cp) / ipq;
    e = __fmaf_rn(e, q.affine_a, q.affine_b);
    float c = (e + q.slack) / sqrt_dqp;
    c = (c < -1.0f) ? -1.0f : ((1.0f < c) ? 1.0f : c);
    float lo = __fmaf_rn(-c, (v.nop + v.nop) * sqrt_dqp, __fmaf_rn(v.nop, v.nop, dqp));
    lo = (lo < 0.0f) ? 0.0f : lo;
    lo = (ipq > kEpsMedium) ? lo : 0.0f;
    uint32_t a, b;
    half_pair(__float_as_uint(lo), a, b);
    lo_stage1 = __uint_as_float(a);
    lo_stage2 = __uint_as_float(b);
}

// The estimate half of stage2_est / est_and_lower (BW >= 2), for callers that took the bounds from lower_bounds_split.
// `tiny` = dqp < kEpsSmall, which the caller knows as a wave-uniform fact (dqp is the popped vertex' distance): tested on
// the per-lane value it compiled to an exec-mask save / branch / restore around the whole estimate.  The division is made
// unconditionally and its result selected -- written as `good ? x / ipq : 0` the compiler branches around it (another
// exec-mask round and a reload of the query constants on the scalar unit, which this kernel runs out of); the quotient
// of a lane with ipq <= 1e-10 is discarded, the others are the same IEEE division.
template <int BW>
__device__ __forceinline__ float stage2_est_only(const QP& q, const LaneEst& v, float dqp, const TailLanes& tl, bool tiny) {
    static_assert(BW >= 2, "N-bit path");
    if (tiny) return __fmaf_rn(v.nop, v.nop, dqp);
    constexpr float K = (float)((1u << BW) - 1);
    constexpr float invK = 1.0f / K;
    const float ipn = ip_approx_est(q.A * invK, (float)v.nbit, q.B * invK, (float)v.wpop, q.C, tl);
    const float ipq = vmaxf(v.ip_qo, q.floor);
    float e = (ipn - v.ip_cp) / ipq;
    asm volatile("" : "+v"(e));
    e = (ipq > kEpsMedium) ? e : 0.0f;
    e = __fmaf_rn(q.affine_a, e, q.affine_b);
    float dist = __fmaf_rn(v.nop, v.nop, dqp);
    dist = __fmaf_rn(-(2.0f * v.nop), e, dist);
    return vmaxf(dist, 0.0f);
}

// ---- exact arithmetic: core/memory.hpp:65-95 -----------------------------------------
// 8 FMA chains (chain j = elements j, j+8, ...) on the 8 lanes of a lane group, then
// ((c0+c4)+(c1+c5)) + ((c2+c6)+(c3+c7)) via xor-4, xor-1, xor-2 exchanges (fp add is
// commutative, so every lane of the group ends with the same bits).
__device__ __forceinline__ float group_reduce8(float c) {
    float a = c + __shfl_xor(c, 4);
    float b = a + __shfl_xor(a, 1);
    return b + __shfl_xor(b, 2);
}
// The same sum with DPP operands instead of LDS-crossbar shuffles (three v_add_f32_dpp).  Only
// lanes 0..3 of each 8-lane group end up with the result (row_shl:4 feeds lanes 4..7 from the
// next group); fp add is commutative, so the pairing ((c0+c4)+(c1+c5))+((c2+c6)+(c3+c7)) and
// its rounding are unchanged.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_reduce8_lo(float c) {
    const float a = c + dpp_mov<0x104>(c);   // row_shl:4         lane j <- lane j+4
    const float b = a + dpp_mov<0xB1>(a);    // quad_perm [1,0,3,2]  lane j <- lane j^1
    return b + dpp_mov<0x4E>(b);             // quad_perm [2,3,0,1]  lane j <- lane j^2
}

// N chain elements of lane j: loads first (all in flight together), then the FMA chain.
template <int N>
__device__ __forceinline__ void chain_load(const float* __restrict__ v, int j, float (&r)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = v[j + 8 * i];
}
template <int N>
__device__ __forceinline__ float chain_dot(const float* q_lds, int j, const float (&r)[N], float c) {
#pragma unroll
    for (int i = 0; i < N; ++i) c = __fmaf_rn(q_lds[j + 8 * i], r[i], c);
    return c;
}
// the same chain with both operands in LDS (v staged there by an LDS-DMA load)
template <int N>
__device__ __forceinline__ float chain_dot_lds(const float* q_lds, const float* v_lds, int j, float c) {
    // eight elements of both operands are read before the FMAs that use them (two LDS round trips for
    // N = 16 instead of one per pair of FMAs); the chain order itself is unchanged
    static_assert(N % 8 == 0, "chain length");
#pragma unroll
    for (int base = 0; base < N; base += 8) {
        float q[8], v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            q[i] = q_lds[j + 8 * (base + i)];
            v[i] = v_lds[j + 8 * (base + i)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __fmaf_rn(q[i], v[i], c);
    }
    return c;
}
template <int N>
__device__ __forceinline__ float chain_l2(const float* q_lds, int j, const float (&r)[N], float c) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float d = q_lds[j + 8 * i] - r[i];
        c = __fmaf_rn(d, d, c);
    }
    return c;
}

// dot(q, v): q in LDS, v a global row; j = lane & 7.  128-element chunks keep 16 loads per
// lane in flight per round trip.
__device__ __forceinline__ float group_dot8(const float* q_lds, const float* __restrict__ v,
                                             uint32_t D, int j) {
    float c = 0.0f;
    if (D >= 128) {
        for (uint32_t base = 0; base < D; base += 128) {
            float r[16];
            chain_load<16>(v + base, j, r);
            c = chain_dot<16>(q_lds + base, j, r, c);
        }
    } else {
        for (uint32_t i = j; i < D; i += 8) c = __fmaf_rn(q_lds[i], v[i], c);
    }
    return group_reduce8(c);
}

// group_dot8 with the DPP reduction: the result is valid in lanes 0..3 of each 8-lane group only.
// dot(q, v): q in LDS, v a global row; j = lane & 7.  128-element chunks keep 16 loads per
// lane in flight per round trip.
__device__ __forceinline__ float group_dot8_lo(const float* q_lds, const float* __restrict__ v,
                                             uint32_t D, int j) {
    float c = 0.0f;
    if (D >= 128) {
        for (uint32_t base = 0; base < D; base += 128) {
            float r[16];
            chain_load<16>(v + base, j, r);
            c = chain_dot<16>(q_lds + base, j, r, c);
        }
    } else {
        for (uint32_t i = j; i < D; i += 8) c = __fmaf_rn(q_lds[i], v[i], c);
    }
    return group_reduce8_lo(c);
}

// sum (q - v)^2 with the same chain structure (l2_distance_simd, core/memory.hpp:65-79).
__device__ __forceinline__ float group_l2sq8(const float* q_lds, const float* __restrict__ v,
                                              uint32_t D, int j) {
    float c = 0.0f;
    if (D >= 128) {
        for (uint32_t base = 0; base < D; base += 128) {
            float r[16];
            chain_load<16>(v + base, j, r);
            c = chain_l2<16>(q_lds + base, j, r, c);
        }
    } else {
        for (uint32_t i = j; i < D; i += 8) {
            const float d = q_lds[i] - v[i];
            c = __fmaf_rn(d, d, c);
        }
    }
    return group_reduce8(c);
}

// search/rabitq_search.hpp:90-93
__device__ __forceinline__ float exact_from_dot(float qnorm, float norm, float dot) {
    float v = (qnorm + norm) - 2.0f * dot;
    return (v < 0.0f) ? 0.0f : v;  // std::max(v, 0.0f)
}

}  // namespace cph
