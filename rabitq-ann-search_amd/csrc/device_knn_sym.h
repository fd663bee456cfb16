// device_knn_sym.h — the exact 32-NN SELF-join with every 128 x 128 tile of the distance matrix computed once.
//
// knn_mfma_kernel (device_knn.h) answers "rows of Q against all of B"; used for B against itself it computes every
// pair twice -- tile (I, J) for the rows of block I and tile (J, I) for the rows of block J hold the same dot products.
// The per-row candidate lists are the obstacle: they live in the LDS of the workgroup that owns the row block, and a tile
// computed by the owner of block I cannot append to lists owned by somebody else without a threshold to filter with.
// So the thresholds come first:
//
//   1. seed      knn_mfma_kernel, all n rows against a SAMPLE of the rows (every 16th: 1/16 of the full work); tau[r] = the
//                squared distance of row r's 8th nearest sample row -- in expectation the distance of its ~120th nearest
//                row overall
//   2. join      knn_sym_kernel (below): workgroup p takes row block p against column tiles p .. last, then row block
//                last - p against its own tail -- (number of blocks + 1) tiles for every workgroup -- and each tile
//                serves BOTH sides: element (r, c) is a candidate for row r if d <= tau[r] and for row c if d <= tau[c],
//                appended (global atomics; ~120 per row over the whole join) to fixed-capacity per-row buffers in HBM.
//                Same matrix-core main loop as knn_mfma_kernel (half-chunk software pipeline, LDS operand images); the
//                accumulators start at zero, d = (|r|^2 + |c|^2) - 2 r.c is formed in the tile epilogue, so a pair has
//                ONE distance whichever side reads it
//   3. select    knn_sym_select_kernel: one wave per row ranks its candidates by (d, id) and writes the best 32.  A row with
//                fewer than 32 candidates (its threshold was too tight: ~2 % of the rows) or more than the buffer holds is
//                put on a list instead ...
//   4. fallback  ... and answered exactly by knn_mfma_kernel on the gathered rows (q_ids: a row's own id to leave out).
//
// Exactness: if at least 32 candidates passed d <= tau[r], the 32 smallest of them are the row's 32 nearest rows (any row
// nearer than the 32nd passes too); otherwise the fallback computes the row against everything.  Work: 1/16 + 1/2 + ~2 %.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_knn.h"

namespace cph {

constexpr uint32_t kKnnSymSample = 16;   // the seed pass sees every 16th row
constexpr uint32_t kKnnSymRank = 8;      // tau = distance of the 8th nearest sample row
constexpr uint32_t kKnnSymCap = 256;     // candidates kept per row (expected ~120, sd ~45)
constexpr uint32_t kKnnSymQueue = 2048;  // appends a workgroup queues in LDS between two flushes
constexpr uint32_t kKnnSymFlushAt = 1024;

struct KnnSymArgs {
    const float* x;        // [n][D] rows (zero padded to D, a multiple of 32)
    const float* norm;     // [n] squared norms
    const float* tau;      // [n] candidate thresholds (squared distances)
    uint32_t n, D, nblk;   // nblk = row blocks of 128
    uint32_t wg_begin;     // first workgroup index of this launch (a launch is a slice of the ceil(nblk / 2) workgroups)
    uint32_t* cnt;         // [n] candidates appended so far (may run past kKnnSymCap: the row then takes the fallback)
    uint32_t* cid;         // [n][kKnnSymCap]
    float* cd;             // [n][kKnnSymCap]
};

// (A variant with two accumulator sets, a finished tile's distance arithmetic running behind the next tile's MFMAs one
// sub-tile per chunk, was built and measured in round 3: the second set and the arithmetic's operands do not fit next to the
// operand registers -- 512 registers used, 58 spilled, the previous set shuttled through v_accvgpr moves -- and the join took
// 2.08 s against 1.37 s at 1M x 128.  profiles/r3b_spilled_beam_experiments.md, section 7.)
template <int V> struct knn_ic { static constexpr int value = V; };

__global__ __launch_bounds__(256, 1) void knn_sym_kernel(KnnSymArgs a) {
    __shared__ __align__(16) float Qs[2][kKnnTile * kKnnLd];     // double-buffered operand images
    __shared__ __align__(16) float Bs[2][kKnnTile * kKnnLd];
    __shared__ float bn_s[3][256];                               // by tile index mod 3: |c|^2 of the tile's 128 columns, then their thresholds
    // Appends are queued in LDS and handed to the per-row buffers in batches: the slot of an append is an atomic add WITH a
    // result, a ~2-us round trip for a wave that has its SIMD to itself, and two tiles in three have one
    __shared__ uint32_t q_row[kKnnSymQueue], q_id[kKnnSymQueue];
    __shared__ float q_d[kKnnSymQueue];
    __shared__ uint32_t q_n;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int h = lane >> 5, c = lane & 31;
    const uint32_t p = a.wg_begin + blockIdx.x;
    const float NEG_INF = -__builtin_inff();
    const uint32_t D = a.D;
    const uint32_t nchunk = D / kKnnKC;
    const uint32_t srow = tid >> 3, skq = (tid & 7) * 4;
    const int wrow = wave * 32;
    if (tid == 0) q_n = 0;
    // one entry into its row's buffer (the slow, direct way: also taken when the queue is full)
    auto append = [&](uint32_t row, uint32_t id, float d) {
        const uint32_t pos = atomicAdd(&a.cnt[row], 1u);
        if (pos < kKnnSymCap) { a.cid[(size_t)row * kKnnSymCap + pos] = id; a.cd[(size_t)row * kKnnSymCap + pos] = d; }
    };
    // whole workgroup, between two barriers of its own: every queued entry to its row, the queue emptied
    auto flush = [&]() {
        const uint32_t m = q_n < kKnnSymQueue ? q_n : kKnnSymQueue;
        for (uint32_t e = tid; e < m; e += 256) append(q_row[e], q_id[e], q_d[e]);
        __syncthreads();
        if (tid == 0) q_n = 0;
        __syncthreads();
    };
    auto enqueue = [&](uint32_t row, uint32_t id, float d) {
        const uint32_t pos = atomicAdd(&q_n, 1u);           // LDS atomic: tens of cycles
        if (pos < kKnnSymQueue) { q_row[pos] = row; q_id[pos] = id; q_d[pos] = d; }
        else append(row, id, d);
    };

    for (int phase = 0; phase < 2; ++phase) {
        const uint32_t blk = phase == 0 ? p : a.nblk - 1 - p;
        if (phase == 1 && blk <= p) break;                       // (the middle block of an odd count is done in phase 0)
        const uint32_t row0 = blk * kKnnTile;
        const uint32_t tile0 = blk, ntile = a.nblk;

        const float* qsrc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t qr = row0 + srow + 32 * i;
            qsrc[i] = a.x + (size_t)(qr < a.n ? qr : a.n - 1) * D + skq;
        }
        // this lane's 16 rows (C/D layout of the 32 x 32 MFMA): thresholds and norms; rows past the end never pass
        float sg[16], nr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t r = row0 + (uint32_t)(wrow + (i & 3) + 8 * (i >> 2) + 4 * h);
            sg[i] = r < a.n ? a.tau[r] : NEG_INF;
            nr[i] = a.norm[r < a.n ? r : a.n - 1];
        }

        knn_f32x4 pq[4], pb[4];
        float pbn = 0.0f;
        uint32_t f_tile = tile0, f_ch = 0, f_slot = 0;     // the chunk the next fetch() brings in; f_slot = (f_tile - tile0) mod 3
        uint32_t f_slot_staged = 0;
        auto fetch = [&]() {
            const uint32_t k0 = f_ch * kKnnKC;
            f_slot_staged = f_slot;
            {   // threads 0..127: |c|^2 of column tid (+inf past the end: d = +inf never passes); 128..255: its threshold
                const uint32_t col = f_tile * kKnnTile + (tid & (kKnnTile - 1));
                const uint32_t cc = col < a.n ? col : a.n - 1;
                const float v = tid < kKnnTile ? a.norm[cc] : a.tau[cc];
                // (the diagonal tile holds (r, c) and (c, r): its row side covers both, its column thresholds are -inf)
                pbn = tid < kKnnTile ? (col < a.n ? v : __builtin_inff()) : ((col < a.n && f_tile != blk) ? v : NEG_INF);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t br = f_tile * kKnnTile + srow + 32 * i;
                pq[i] = *reinterpret_cast<const knn_f32x4*>(qsrc[i] + k0);
                pb[i] = *reinterpret_cast<const knn_f32x4*>(a.x + (size_t)(br < a.n ? br : a.n - 1) * D + skq + k0);
            }
            const bool wrap = f_ch + 1 == nchunk;
            f_ch = wrap ? 0 : f_ch + 1;
            const bool next = wrap && f_tile + 1 < ntile;
            f_tile = next ? f_tile + 1 : f_tile;
            f_slot = next ? (f_slot == 2 ? 0 : f_slot + 1) : f_slot;
        };
        auto stage = [&](int buf) {
            bn_s[f_slot_staged][tid] = pbn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<knn_f32x4*>(&Qs[buf][(srow + 32 * i) * kKnnLd + skq]) = pq[i];
                *reinterpret_cast<knn_f32x4*>(&Bs[buf][(srow + 32 * i) * kKnnLd + skq]) = pb[i];
            }
        };
        struct Ops { knn_f32x4 av[2], bv[2][4]; };
        auto read_ops = [&](int buf, int half, Ops& o) {
            const float* Qb = Qs[buf];
            const float* Bb = Bs[buf];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * half + jj;
                o.av[jj] = *reinterpret_cast<const knn_f32x4*>(&Qb[(wrow + c) * kKnnLd + 8 * j + 4 * h]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o.bv[jj][t] = *reinterpret_cast<const knn_f32x4*>(&Bb[(t * 32 + c) * kKnnLd + 8 * j + 4 * h]);
            }
        };
        knn_f32x16 acc[1][4];
        auto mfma_half = [&](auto SC, const Ops& o) {
            constexpr int S = decltype(SC)::value;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[S][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].x, o.bv[jj][t].x, acc[S][t], 0, 0, 0);
                    acc[S][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].y, o.bv[jj][t].y, acc[S][t], 0, 0, 0);
                    acc[S][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].z, o.bv[jj][t].z, acc[S][t], 0, 0, 0);
                    acc[S][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].w, o.bv[jj][t].w, acc[S][t], 0, 0, 0);
                }
            }
        };
        // "does any of this lane's 16 distances of sub-tile t pass either threshold": max over (threshold - d), >= 0 if so
        auto lead_of = [&](auto PC, auto TC, uint32_t slot) {
            constexpr int P = decltype(PC)::value;
            constexpr int t = decltype(TC)::value;
            const float nc = bn_s[slot][t * 32 + c];
            const float tc = bn_s[slot][kKnnTile + t * 32 + c];
            float lead = NEG_INF;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float dd = __builtin_fmaf(-2.0f, acc[P][t][i], nr[i] + nc);
                lead = __builtin_fmaxf(lead, __builtin_fmaxf(sg[i], tc) - dd);
            }
            return lead;
        };
        // the appends of tile `tile` (accumulator set P, norms / thresholds in bn_s[slot]) for the sub-tiles whose lead says so
        auto appends = [&](auto PC, uint32_t tile, uint32_t slot, const float (&lead)[4]) {
            constexpr int P = decltype(PC)::value;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!__any(lead[t] >= 0.0f)) continue;
                const float nc = bn_s[slot][t * 32 + c];
                const float tc = bn_s[slot][kKnnTile + t * 32 + c];
                const uint32_t col = tile * kKnnTile + t * 32 + c;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t row = row0 + (uint32_t)(wrow + (i & 3) + 8 * (i >> 2) + 4 * h);
                    const float dd = __builtin_fmaf(-2.0f, acc[P][t][i], nr[i] + nc);
                    const float d = dd > 0.0f ? dd : 0.0f;
                    if (dd <= sg[i] && col != row) enqueue(row, col, d);      // (sg = -inf for rows past the end, d = +inf for columns)
                    if (dd <= tc && row < a.n) enqueue(col, row, d);          // (tc = -inf on the diagonal and for columns past the end)
                }
            }
        };
        auto epilogue_now = [&](auto PC, uint32_t tile, uint32_t slot) {
            float lead[4];
            lead[0] = lead_of(PC, knn_ic<0>{}, slot);
            lead[1] = lead_of(PC, knn_ic<1>{}, slot);
            lead[2] = lead_of(PC, knn_ic<2>{}, slot);
            lead[3] = lead_of(PC, knn_ic<3>{}, slot);
            appends(PC, tile, slot, lead);
        };

        // prologue: image(0) staged and published, the next chunk in flight, first-half operands in registers
        Ops op0, op1;
        __syncthreads();                    // (second phase: every wave is done with the images of the first)
        fetch();
        stage(0);
        fetch();
        __syncthreads();
        read_ops(0, 0, op0);
        int buf = 0;
        // one K chunk of the current tile
        auto chunk_body = [&](auto SC, uint32_t ch) {
            constexpr int S = decltype(SC)::value;
            if (ch == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[S][t][i] = 0.0f;
            }
            // ---- first half (see knn_mfma_kernel for the schedule) ----
            __builtin_amdgcn_sched_barrier(0);
            read_ops(buf, 1, op1);
            stage(buf ^ 1);
            fetch();
            mfma_half(SC, op0);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // VALU (addresses of the fetch)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            buf ^= 1;
            // (every wave is past the previous tile's appends here and none appends before the next ones: q_n is the same
            // for all of them, the branch is uniform across the workgroup)
            if (ch == 0 && q_n >= kKnnSymFlushAt) flush();
            // ---- second half ----
            read_ops(buf, 0, op0);
            mfma_half(SC, op1);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 22, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        // slots of the norm / threshold images: tile (tile0 + j) lives in bn_s[j mod 3]
        uint32_t cslot = 0;
        for (uint32_t tile = tile0; tile < ntile; ++tile) {
            for (uint32_t ch = 0; ch < nchunk; ++ch) chunk_body(knn_ic<0>{}, ch);
            // ---- tile finished: both sides of its 4 x (32 x 32) distances against their thresholds ----
            epilogue_now(knn_ic<0>{}, tile, cslot);
            cslot = cslot == 2 ? 0 : cslot + 1;
        }
        __syncthreads();                    // every wave's last appends of this row block are in the queue
        flush();
    }
}

// tau[r] = squared distance of row r's kKnnSymRank-th nearest SAMPLE row (the sample holds rows 0, S, 2S, ...: a row that
// is in the sample finds itself first and takes the next one)
__global__ __launch_bounds__(256) void knn_sym_tau_kernel(const float* __restrict__ seed_dist /* [n][32] ascending */, uint32_t n,
                                                          float* __restrict__ tau) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t j = kKnnSymRank - 1 + (r % kKnnSymSample == 0 ? 1u : 0u);
    tau[r] = seed_dist[(size_t)r * kKnnK + j];
}

// One wave per row: rank the candidates by (d, id), write the best 32; rows that cannot be answered from their buffer go
// on the fallback list.
__global__ __launch_bounds__(64) void knn_sym_select_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cid,
                                                            const float* __restrict__ cd, uint32_t n, uint32_t* __restrict__ out_ids,
                                                            float* __restrict__ out_dist, uint32_t* __restrict__ redo, uint32_t* __restrict__ redo_count) {
    __shared__ float sd[kKnnSymCap];
    __shared__ uint32_t si[kKnnSymCap];
    const int lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint32_t c = cnt[r];
        if (c < (uint32_t)kKnnK || c > kKnnSymCap) {
            if (lane == 0) redo[atomicAdd(redo_count, 1u)] = r;
            continue;
        }
        __syncthreads();
        for (uint32_t j = lane; j < c; j += 64) { sd[j] = cd[(size_t)r * kKnnSymCap + j]; si[j] = cid[(size_t)r * kKnnSymCap + j]; }
        __syncthreads();
        for (uint32_t j = lane; j < c; j += 64) {
            const float d = sd[j];
            const uint32_t id = si[j];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < c; ++q) {
                const float dq = sd[q];
                rank += (dq < d || (dq == d && si[q] < id)) ? 1u : 0u;
            }
            if (rank < (uint32_t)kKnnK) { out_ids[(size_t)r * kKnnK + rank] = id; out_dist[(size_t)r * kKnnK + rank] = d; }
        }
    }
}

// out[rows[i]][0..32) = src[i][0..32)
__global__ __launch_bounds__(256) void knn_sym_scatter_kernel(const uint32_t* __restrict__ rows, uint32_t nr, const uint32_t* __restrict__ src_ids,
                                                              const float* __restrict__ src_dist, uint32_t* __restrict__ out_ids,
                                                              float* __restrict__ out_dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr * (uint32_t)kKnnK) return;
    const uint32_t r = rows[i / kKnnK], j = i % kKnnK;
    out_ids[(size_t)r * kKnnK + j] = src_ids[i];
    out_dist[(size_t)r * kKnnK + j] = src_dist[i];
}

// sample ids 0, S, 2S, ...
__global__ __launch_bounds__(256) void knn_sym_iota_kernel(uint32_t* out, uint32_t count, uint32_t step) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = i * step;
}

}  // namespace cph
