// device_stream.h — streaming FastScan (no graph): the kernel the HBM roofline is quoted on,
// plus the single-block hook used by the parity tests.
//
// One wave = one 32-neighbour block per iteration, grid-stride over the block array; both
// N-bit stages are evaluated for every block (as search/rabitq_search.hpp:170-200 does when
// no batch is skipped), reading the block exactly once.
#pragma once
#include <hip/hip_runtime.h>

#include "cph_core.h"
#include "device_fastscan.h"

namespace cph {

struct StreamArgs {
    const uint8_t* blocks;
    uint64_t n_blocks;
    DevLayout L;
    const uint4* qmask;   // [PW]
    QP qp;
    float dqp;
    float* sink;          // one float per wave (checksum, defeats DCE)
    float* out_est;       // optional [count][32] for blocks [first, first+count)
    float* out_lower;
    uint64_t first, count;
};

template <int BW, int SD>
__global__ __launch_bounds__(256) void fastscan_stream_kernel(StreamArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* qm = reinterpret_cast<uint4*>(smem);
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : a.L.PW;
    for (uint32_t w = threadIdx.x; w < PW; w += blockDim.x) qm[w] = a.qmask[w];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const float sq = __builtin_sqrtf(a.dqp);
    float acc = 0.0f;
    const uint64_t lo = a.out_est ? a.first : 0;
    const uint64_t hi = a.out_est ? a.first + a.count : a.n_blocks;
    // software pipeline: the next block's loads (codes, aux, ids) are in flight while the
    // current block is reduced
    BlockLoads<BW, SD> cur, nxt;
    uint32_t cur_id = 0, nxt_id = 0;
    uint64_t b = lo + wave;
    if (b < hi) {
        const uint8_t* blk = a.blocks + b * a.L.stride;
        cur.issue(blk, a.L, lane);
        cur_id = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[lane & 31];
    }
    for (; b < hi; b += nwaves) {
        const uint64_t bn = b + nwaves;
        if (bn < hi) {
            const uint8_t* nblk = a.blocks + bn * a.L.stride;
            nxt.issue(nblk, a.L, lane);
            nxt_id = reinterpret_cast<const uint32_t*>(nblk + a.L.ids_off)[lane & 31];
        }
        const uint8_t* blk = a.blocks + b * a.L.stride;
        LaneEst v;
        cur.reduce(blk, a.L, qm, lane, v);
        float est, lower;
        const TailLanes tl = tail_lanes(cur_id != kInvalidNode, lane);     // short lists: the reference's scalar tails
        if constexpr (BW == 1) {
            stage2_est<1>(a.qp, v, a.dqp, sq, est, lower, tl);
            acc += est + lower;
        } else {
            float lo1 = stage1_lower<BW>(a.qp, v, a.dqp, sq);
            stage2_est<BW>(a.qp, v, a.dqp, sq, est, lower, tl);
            acc += est + lower + lo1;
        }
        if (cur_id == kInvalidNode) acc = 0.0f;  // the neighbour ids are part of the unit of work
        if (a.out_est && lane < 32) {
            a.out_est[(b - a.first) * 32 + lane] = est;
            a.out_lower[(b - a.first) * 32 + lane] = lower;
        }
        cur = nxt;
        cur_id = nxt_id;
    }
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && !a.out_est) a.sink[wave] = acc;
}

// Narrow codes at D = 128 (1 and 2 bits: 512 B / 1 KiB of codes per block).  With one block per wave iteration
// the kernel above is bound by the VECTOR ALU, not by HBM: the estimator epilogue runs on 64 lanes for 32
// neighbours and costs as much as for 4-bit codes while the block is less than half as long (measured 0.50 /
// 0.60 of peak).  Here every wave iteration takes TWO consecutive blocks, one per lane half: lane (h, i) loads
// all of neighbour i's code chunks of block 2p + h (each load instruction still reads 2 x 512 contiguous bytes),
// needs no cross-half exchange, and the epilogue's instructions serve 64 neighbours.
template <int BW>
__global__ __launch_bounds__(256) void fastscan_stream_pair_kernel(StreamArgs a) {
    static_assert(BW == 1 || BW == 2, "pair kernel: narrow codes only");
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* qm = reinterpret_cast<uint4*>(smem);
    for (uint32_t w = threadIdx.x; w < 4; w += blockDim.x) qm[w] = a.qmask[w];     // D = 128: four mask words
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, i = lane & 31;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const float sq = __builtin_sqrtf(a.dqp);
    float acc = 0.0f;
    const uint64_t lo = a.out_est ? a.first : 0;
    const uint64_t hi = a.out_est ? a.first + a.count : a.n_blocks;
    struct Loads { uint4 c[BW]; uint4 aux; uint32_t id; };
    auto issue = [&](uint64_t b, Loads& L) {
        const uint8_t* blk = a.blocks + b * a.L.stride;
        const uint4* cp = reinterpret_cast<const uint4*>(blk) + i;
#pragma unroll
        for (int k = 0; k < BW; ++k) L.c[k] = cp[k * 32];          // plane k of neighbour i ([k][half][neighbour] order)
        L.aux = reinterpret_cast<const uint4*>(blk + a.L.aux_off)[i];
        L.id = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[i];
    };
    Loads cur{}, nxt{};
    uint64_t b = lo + 2 * wave + h;                 // this lane half's block
    if (b < hi) issue(b, cur);
    for (; b - h < hi; b += 2 * nwaves) {           // the loop is wave-uniform: it runs while the pair's first block exists
        const uint64_t bn = b + 2 * nwaves;
        if (bn < hi) issue(bn, nxt);
        const bool live = b < hi;
        LaneEst v;
        uint32_t S[BW];
#pragma unroll
        for (int k = 0; k < BW; ++k) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            acc4(cur.c[k].x, qm[0], a0, a1, a2, a3);
            acc4(cur.c[k].y, qm[1], a0, a1, a2, a3);
            acc4(cur.c[k].z, qm[2], a0, a1, a2, a3);
            acc4(cur.c[k].w, qm[3], a0, a1, a2, a3);
            S[k] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
        }
        if constexpr (BW == 1) { v.nbit = v.msb = v.msb2 = S[0]; }
        else { v.nbit = 2 * S[0] + S[1]; v.msb = S[0]; v.msb2 = 2 * S[0] + S[1]; }
        v.nop = __uint_as_float(cur.aux.x);
        v.ip_qo = __uint_as_float(cur.aux.y);
        v.ip_cp = __uint_as_float(cur.aux.z);
        v.pop = cur.aux.w & 0xFFFFu;
        v.wpop = cur.aux.w >> 16;
        float est, lower;
        // short lists (one block per lane half here, so each half counts its own ids)
        TailLanes tl;
        {
            const unsigned long long vm = __ballot(live && cur.id != kInvalidNode);
            const uint32_t count = (uint32_t)__popc((uint32_t)(h ? vm >> 32 : vm & 0xFFFFFFFFull));
            tl.any = __any((count & 7u) != 0u);
            tl.mine = (uint32_t)i >= (count & ~7u);
        }
        if constexpr (BW == 1) {
            stage2_est<1>(a.qp, v, a.dqp, sq, est, lower, tl);
            if (live) acc += est + lower;
        } else {
            const float lo1 = stage1_lower<BW>(a.qp, v, a.dqp, sq);
            stage2_est<BW>(a.qp, v, a.dqp, sq, est, lower, tl);
            if (live) acc += est + lower + lo1;
        }
        if (live && cur.id == kInvalidNode) acc = 0.0f;    // the neighbour ids are part of the unit of work
        if (a.out_est && live) {
            a.out_est[(b - a.first) * 32 + i] = est;
            a.out_lower[(b - a.first) * 32 + i] = lower;
        }
        cur = nxt;
    }
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && !a.out_est) a.sink[wave] = acc;
}

// Parity hook: one block of a loaded index, all intermediate values.
struct BlockHookArgs {
    const uint8_t* blk;
    DevLayout L;
    const uint4* qmask;
    QP qp;
    float dqp, worst;
    int nn_full;
    uint32_t* sums;   // [32]
    uint32_t* msb;    // [32]
    float* est;       // [32]
    float* lower;     // [32]
    float* lower1;    // [32]
};

template <int BW, int SD>
__global__ __launch_bounds__(64) void block_hook_kernel(BlockHookArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* qm = reinterpret_cast<uint4*>(smem);
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : a.L.PW;
    const int lane = threadIdx.x;
    for (uint32_t w = lane; w < PW; w += 64) qm[w] = a.qmask[w];
    __syncthreads();
    LaneEst v;
    load_block<BW, SD>(a.blk, a.L, qm, lane, v);
    const bool valid = reinterpret_cast<const uint32_t*>(a.blk + a.L.ids_off)[lane & 31] != kInvalidNode;
    const float sq = __builtin_sqrtf(a.dqp);
    float est, lower, lo1;
    const TailLanes tl = tail_lanes(valid, lane);
    if constexpr (BW == 1) {
        stage2_est<1>(a.qp, v, a.dqp, sq, est, lower, tl);
        lo1 = lower;
    } else {
        lo1 = stage1_lower<BW>(a.qp, v, a.dqp, sq);
        bool surv = (!a.nn_full) || (valid && lo1 < a.worst);
        if (__any(surv)) {
            stage2_est<BW>(a.qp, v, a.dqp, sq, est, lower, tl);
        } else {
            est = 3.402823466e+38f;
            lower = lo1;
        }
    }
    // lanes 32..63 carry the same neighbour as lanes 0..31: let the upper half write so the
    // redundant-half path is what the test sees for half of the outputs
    const int i = lane & 31;
    const bool writer = (i < 16) ? (lane < 32) : (lane >= 32);
    if (writer) {
        a.sums[i] = v.nbit;
        a.msb[i] = v.msb;
        a.est[i] = est;
        a.lower[i] = lower;
        a.lower1[i] = lo1;
    }
}

// Exact-L2 hook: one query (padded, device) against ids[0..n).
__global__ __launch_bounds__(64) void exact_l2_hook_kernel(const float* __restrict__ q,
                                                           const float* __restrict__ raw,
                                                           const float* __restrict__ norm_sq,
                                                           const uint32_t* __restrict__ ids,
                                                           uint64_t n, uint32_t D, float* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* qv = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x;
    for (uint32_t d = lane; d < D; d += 64) qv[d] = q[d];
    __syncthreads();
    float c = 0.0f;
    for (uint32_t i = lane & 7; i < D; i += 8) c = __fmaf_rn(qv[i], qv[i], c);
    const float qnorm = group_reduce8(c);
    const int g = lane >> 3;
    for (uint64_t base = (uint64_t)blockIdx.x * 8; base < n; base += (uint64_t)gridDim.x * 8) {
        const bool have = base + g < n;
        const uint32_t id = have ? ids[base + g] : ids[0];
        float dot = group_dot8(qv, raw + (size_t)id * D, D, lane & 7);
        float ex = exact_from_dot(qnorm, norm_sq[id], dot);
        if (have && (lane & 7) == 0) out[base + g] = ex;
    }
}

}  // namespace cph
