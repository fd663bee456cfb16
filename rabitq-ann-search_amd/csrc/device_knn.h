// device_knn.h — exact k-nearest-neighbour lists by brute force on the matrix cores (index
// construction and the recall ground truth).
//
// Replaces the reference's CPU NNDescent (graph/graph_refinement.hpp:71-263, 455-515), whose product
// is a working list of the R = 32 nearest neighbours per node, and its distance routine
// l2_distance_simd (core/memory.hpp:65-79).  On MI355X the exact answer is cheaper than the
// approximation: the n_q x n_b x D contraction runs on v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate: bit-for-bit a k-ordered fmaf chain, so nothing is lost against a VALU kernel) and a
// threshold-filtered top-32 per query row lives in LDS, so nothing but the 32 results per row is
// ever written to HBM.
//
// Geometry.  One workgroup (4 waves, one per SIMD) = 128 query rows against every base row, in
// 128-column tiles; the K dimension is staged through LDS in chunks of 32 floats (64 MFMAs per wave and chunk),
// software-pipelined at half-chunk granularity: operand reads, staging of the next chunk and the global loads of the
// one after run behind the MFMAs (see the main loop).
// Wave w owns rows 32w..32w+31 of the tile and computes all 128 columns for them (1 x 4 MFMA tiles of
// 32 x 32), so the per-row candidate lists are touched by one wave only and the selection needs no
// workgroup barrier.
//
// Ranking key.  The accumulators start at -|b|^2 / 2, so a finished accumulator holds
//     s = q.b - |b|^2 / 2,        |q - b|^2 = |q|^2 - 2 s,
// and "nearer" is "larger s" with a per-row constant removed: the filter is a subtract and a max per element against
// the row's threshold (kept in registers), one compare per 32 x 32 sub-tile.  Passing elements are appended to the row's
// 64-entry LDS list; only when an append would not fit is the list sorted and cut back to its best 32 by
// rank computation (every lane holds one entry and counts the entries that beat it, broadcast through
// v_readlane), which also yields the new threshold: the 32nd best key seen so far.  Between two sorts the
// threshold is stale by at most 32 appends, which costs a few extra appends and never a wrong result.
//
// LDS operand image: row-major [128][32 + 4] floats.  Lane (h = lane >> 5, c = lane & 31) reads
// its row's k = 8j + 4h .. 8j + 4h + 3 with one ds_read_b128 and feeds the four values to four
// consecutive MFMAs; A and B use the same k for the same (lane half, step), which is all the
// instruction needs -- the order in which k is summed is irrelevant to an exact-kNN list.  The 36-float
// stride makes the 16 rows of each ds_read_b128 lane group land on 16 distinct 4-bank spans.
//
// Where the time goes (s_memtime stamps per phase, 262,144 x 128, after the half-chunk pipeline below): the chunk loop
// runs at 77 % of the matrix pipe's rate (second halves fully covered, first halves +25 %, the barrier ~350 cycles per
// chunk); the tile epilogue -- accumulator reads, threshold test, the appends of the 40 % of the 32 x 32 sub-tiles that
// have one, accumulator re-initialisation -- is ~6,000 cycles per tile next to the tile's 16,400 cycles of MFMAs at
// D = 128 (a tenth of that share at D = 1024).  Earlier attempts: a whole second operand set (three-stage pipeline)
// does not fit in 256 architectural registers -- the compiler parks it in AGPRs and shuttles it with hundreds of
// v_accvgpr moves per chunk, 70 TF/s; two workgroups per CU spill around the selection code.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cph {

constexpr int kKnnK = 32;        // neighbours kept per row (R)
constexpr int kKnnTile = 128;    // rows per workgroup, columns per tile
constexpr int kKnnKC = 32;       // floats of K per LDS stage
constexpr int kKnnLd = kKnnKC + 4;
constexpr int kKnnCap = 64;      // per-row candidate list: one entry per lane when it is sorted

struct KnnArgs {
    const float* q;        // [nq][D] query rows (zero padded to D)
    const float* b;        // [nb][D] base rows
    const float* bnorm;    // [nb] squared norms of the base rows
    const float* qnorm;    // [nq]
    uint32_t nq, nb, D;
    uint32_t row_begin, row_end;   // query rows handled by this launch
    uint32_t exclude_self;         // q == b: row i never lists column i
    uint32_t* out_ids;     // [nq][32] ascending by distance, 0xFFFFFFFF padded
    float* out_dist;       // [nq][32] squared L2, FLT_MAX padded
    const uint32_t* q_ids; // optional with exclude_self: query row i is base row q_ids[i] (a gathered subset) instead of row i
};

typedef float knn_f32x16 __attribute__((ext_vector_type(16)));
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers

// Sorts row r's list (c <= 64 entries, one per lane) by (key descending, id ascending), keeps the best
// 32 and publishes the new threshold.  Whole wave; the row belongs to this wave.  Every lane's entry is
// broadcast through the scalar registers (v_readlane) and compared on all lanes at once: a lane's rank is
// the number of entries that beat its own -- no LDS round trips, no data-dependent shuffles.
__device__ __attribute__((noinline)) void knn_compact(float* cs, uint32_t* ci, uint32_t* cnt, float* sig, int r, int lane) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[r]);
    const float NEG = -__builtin_inff();
    const float s = (uint32_t)lane < c ? cs[r * kKnnCap + lane] : NEG;
    const uint32_t id = (uint32_t)lane < c ? ci[r * kKnnCap + lane] : 0xFFFFFFFFu;
    uint32_t rank = 0;
#pragma unroll
    for (int j = 0; j < kKnnCap; ++j) {
        const float sj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(s), j));
        const uint32_t ij = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
        rank += (sj > s || (sj == s && ij < id)) ? 1u : 0u;
    }
    if ((uint32_t)lane < c && rank < (uint32_t)kKnnK) {
        cs[r * kKnnCap + rank] = s;
        ci[r * kKnnCap + rank] = id;
        if (rank == (uint32_t)kKnnK - 1) sig[r] = s;
    }
    if (lane == 0) cnt[r] = c < (uint32_t)kKnnK ? c : (uint32_t)kKnnK;
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, 1) void knn_mfma_kernel(KnnArgs a) {
    __shared__ __align__(16) float Qs[2][kKnnTile * kKnnLd];     // double-buffered operand images
    __shared__ __align__(16) float Bs[2][kKnnTile * kKnnLd];
    __shared__ float cs[kKnnTile * kKnnCap];
    __shared__ uint32_t ci[kKnnTile * kKnnCap];
    __shared__ uint32_t cnt[kKnnTile];
    __shared__ float sig[kKnnTile];
    __shared__ float bn_s[2][256];                               // |b|^2 of the tile's columns, by tile parity (first 128 of each)

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int h = lane >> 5, c = lane & 31;
    const uint32_t row0 = a.row_begin + blockIdx.x * kKnnTile;
    if (row0 >= a.row_end) return;
    const float NEG_INF = -__builtin_inff();
    __shared__ uint32_t self_s[kKnnTile];                        // the base row a query row must not list (exclude_self)
    if (tid < kKnnTile) {
        cnt[tid] = 0; sig[tid] = NEG_INF;
        const uint32_t qr = row0 + (uint32_t)tid;
        self_s[tid] = a.q_ids ? a.q_ids[qr < a.nq ? qr : a.nq - 1] : qr;
    }

    const uint32_t D = a.D;                       // multiple of kKnnKC (the host pads)
    const uint32_t nchunk = D / kKnnKC;
    const uint32_t ntile = (a.nb + kKnnTile - 1) / kKnnTile;

    // Staging: 4 x 16 B per operand per thread and chunk; element e = tid + 256 i -> row e >> 3, float4
    // e & 7 of the chunk.  Rows past the end are clamped to the last row instead of predicated (no
    // branches between the MFMAs): a clamped query row computes keys nobody reads, a clamped base row
    // belongs to a column whose accumulator starts at -inf and therefore never passes the filter.
    const uint32_t srow = tid >> 3, skq = (tid & 7) * 4;
    const float* qsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t qr = row0 + srow + 32 * i;
        qsrc[i] = a.q + (size_t)(qr < a.nq ? qr : a.nq - 1) * D + skq;
    }
    // Software pipeline at HALF-chunk granularity.  A chunk's 64 MFMAs split into two halves of 32 (k-steps j = 0,1 and
    // j = 2,3), each with its own operand registers (2 A + 8 B reads of 16 B: 40 registers per half, the 80 the kernel
    // had for a whole chunk).  While a half's MFMAs run, the other half's operands are read from LDS:
    //   first half of chunk g   (operands op0):  read op1 = (g, second half);  stage chunk g+1 into the other image;
    //                                            fetch chunk g+2 from HBM
    //   --- barrier: image(g+1) published, every wave done reading image(g) ---
    //   second half of chunk g  (operands op1):  read op0 = (g+1, first half)
    // so no LDS round trip, no staging and no address arithmetic stands in front of the matrix pipe any more (one wave
    // per SIMD: nothing else would cover them -- they were ~2,000 of every 6,100 cycles per chunk); what remains exposed
    // is the barrier's skew.  Same registers, same LDS, same arithmetic.
    knn_f32x4 pq[4], pb[4];
    float pbn = 0.0f;                   // |b|^2 of column (tid & 127) of the tile being fetched
    uint32_t f_tile = 0, f_ch = 0;      // the chunk the next fetch() brings in (past the end: the last tile again, unused)
    uint32_t f_tile_staged = 0;
    auto fetch = [&]() {
        const uint32_t k0 = f_ch * kKnnKC;
        f_tile_staged = f_tile;
        {   // every chunk, every thread, no branch (a half is one basic block, or the interleave below falls apart);
            // columns past the end get +inf: their accumulators start at -inf and never pass the filter
            const uint32_t col = f_tile * kKnnTile + (tid & (kKnnTile - 1));
            const float bn = a.bnorm[col < a.nb ? col : a.nb - 1];
            pbn = col < a.nb ? bn : __builtin_inff();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t br = f_tile * kKnnTile + srow + 32 * i;
            pq[i] = *reinterpret_cast<const knn_f32x4*>(qsrc[i] + k0);
            pb[i] = *reinterpret_cast<const knn_f32x4*>(a.b + (size_t)(br < a.nb ? br : a.nb - 1) * D + skq + k0);
        }
        const bool wrap = f_ch + 1 == nchunk;
        f_ch = wrap ? 0 : f_ch + 1;
        f_tile = (wrap && f_tile + 1 < ntile) ? f_tile + 1 : f_tile;
    };
    auto stage = [&](int buf) {         // what fetch() brought in -> image `buf` (and its tile's column norms)
        bn_s[f_tile_staged & 1][tid] = pbn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<knn_f32x4*>(&Qs[buf][(srow + 32 * i) * kKnnLd + skq]) = pq[i];
            *reinterpret_cast<knn_f32x4*>(&Bs[buf][(srow + 32 * i) * kKnnLd + skq]) = pb[i];
        }
    };
    const int wrow = wave * 32;         // first tile row of this wave
    // operand registers of half a chunk: lane (h, c) reads its row's k = 8j + 4h .. + 3 for j = 2 half, 2 half + 1
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
    knn_f32x16 acc[4];
    auto mfma_half = [&](const Ops& o) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].x, o.bv[jj][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].y, o.bv[jj][t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].z, o.bv[jj][t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.av[jj].w, o.bv[jj][t].w, acc[t], 0, 0, 0);
            }
        }
    };
    float sg[16];                       // thresholds of this lane's 16 rows
#pragma unroll
    for (int i = 0; i < 16; ++i) sg[i] = NEG_INF;

    // prologue: image(0) staged and published, chunk 1 in flight, first-half operands of chunk 0 in registers
    Ops op0, op1;
    fetch();
    stage(0);
    fetch();
    __syncthreads();                    // image(0), cnt and sig are in place
    read_ops(0, 0, op0);
    int buf = 0;
    for (uint32_t tile = 0; tile < ntile; ++tile)
    for (uint32_t ch = 0; ch < nchunk; ++ch) {
        if (ch == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // (the column norms come through LDS with the tile's first operand image: a global load here would be a
                // memory round trip in front of every tile)
                const float init = -0.5f * bn_s[tile & 1][t * 32 + c];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = init;
            }
        }
        // ---- first half: MFMAs on op0; behind them op1 of this chunk, the staging of the next chunk, the fetch of the
        // one after -----------------------------------------------------------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        read_ops(buf, 1, op1);
        stage(buf ^ 1);
        fetch();
        mfma_half(op0);
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
        // ---- second half: MFMAs on op1; behind them the first-half operands of the next chunk -------------------
        read_ops(buf, 0, op0);
        mfma_half(op1);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 22, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 != nchunk) continue;

        // ---- tile finished: filter its 4 x (32 x 32) keys against the row thresholds ----------
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // "does any of my 16 keys beat its row's threshold" as one max-reduction and one compare: sixteen compares
            // into scalar registers OR-ed together made a dependent chain through the scalar unit (a key of a column past
            // the end is -inf: against a threshold that is still -inf the difference is NaN, which v_max drops)
            float lead = NEG_INF;
#pragma unroll
            for (int i = 0; i < 16; ++i) lead = __builtin_fmaxf(lead, acc[t][i] - sg[i]);
            if (!__any(lead > 0.0f)) continue;
            const uint32_t col = tile * kKnnTile + t * 32 + c;
            bool compacted = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rlo = wrow + (i & 3) + 8 * (i >> 2);           // C/D layout of the 32x32 MFMA:
                const int r = rlo + 4 * h;                               // lane half h holds row rlo + 4h
                const float s = acc[t][i];
                bool pass = s > sg[i] && !(a.exclude_self && col == self_s[r]);
                if (__ballot(pass) == 0) continue;
                // the 32 lanes of a half share the row; the two halves take their turns (wave-uniform)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int rr = rlo + 4 * hh;
                    bool mine = pass && h == hh;
                    uint32_t mh = (uint32_t)(__ballot(mine) >> (32 * hh));
                    if (mh == 0) continue;
                    uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[rr]);
                    if (base + __popc(mh) > (uint32_t)kKnnCap) {
                        // the list is full: sort, keep the best 32, raise the threshold and look again
                        knn_compact(cs, ci, cnt, sig, rr, lane);
                        compacted = true;
                        mine = mine && s > sig[rr];
                        mh = (uint32_t)(__ballot(mine) >> (32 * hh));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[rr]);
                    }
                    if (mine) {
                        const uint32_t pos = base + __popc(mh & ((1u << c) - 1u));
                        cs[rr * kKnnCap + pos] = s;
                        ci[rr * kKnnCap + pos] = col;
                    }
                    if (lane == 0) cnt[rr] = base + __popc(mh);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (compacted) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sg[i] = sig[wrow + (i & 3) + 8 * (i >> 2) + 4 * h];
            }
        }
    }
    // ---- results: every row sorted ascending by distance ----------------------------------------
    for (int r = 0; r < 32; ++r) {
        const uint32_t qr = row0 + wrow + r;
        if (qr >= a.nq || qr >= a.row_end) break;
        knn_compact(cs, ci, cnt, sig, wrow + r, lane);
        if (lane < kKnnK) {
            const bool have = (uint32_t)lane < cnt[wrow + r];
            float d = 3.402823466e+38f;
            if (have) {
                d = a.qnorm[qr] - 2.0f * cs[(wrow + r) * kKnnCap + lane];
                d = d > 0.0f ? d : 0.0f;
            }
            a.out_ids[(size_t)qr * kKnnK + lane] = have ? ci[(wrow + r) * kKnnCap + lane] : 0xFFFFFFFFu;
            a.out_dist[(size_t)qr * kKnnK + lane] = d;
        }
    }
}

}  // namespace cph
