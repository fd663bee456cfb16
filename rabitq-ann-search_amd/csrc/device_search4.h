// device_search4.h — layer-0 beam search with REGISTER-RESIDENT heaps driven by the scalar unit.
//
// Same algorithm and bit-exact semantics as device_search.h (reference citations there).  The
// one-query-per-wave kernel spends ~60 % of its VALU instructions in single-lane heap code
// (libstdc++-exact sift loops over LDS).  Here heap entry e lives in lane (e & 63) of VGPR set
// (e >> 6): the first 256 beam entries {est, lower, id} occupy 12 VGPRs, the result heap
// {dist, id} (k <= 256) 8 VGPRs.  All heap indices are wave-uniform, so an access is one
// v_readlane (read) or compare + v_cndmask (write) with a scalar lane select, and the sift loops — compares, index
// arithmetic, branches — run on the scalar unit.  Keys are non-negative floats, whose order is
// the order of their bit patterns, so key compares are integer compares (s_cmp).
// Beam entries beyond 256 spill to the slot's global array exactly as before.
//
// Status: bit-exact (GPU parity suite passes with CPH_SEARCH_KERNEL=4), VALU work roughly halved —
// but a CU has ONE scalar unit for all its waves, so the serial work merely moves from "1 VALU
// instruction per 4 cycles per SIMD" to "1 SALU instruction per cycle per CU", the same budget:
// measured 6.5 ms vs 5.5 ms for the LDS-heap kernel on the 1M/10k-query case.  Kept as a tested
// alternative, not the default.
#pragma once
#include <hip/hip_runtime.h>

#include "cph_core.h"
#include "device_fastscan.h"
#include "device_search.h"

namespace cph {

constexpr uint32_t kRegBeam = 256;   // beam entries held in registers (4 sets x 64 lanes)
constexpr uint32_t kRegNn = 256;     // largest k this kernel serves

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t ln) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)ln);
}
// write `val` into lane `ln` of a VGPR: this toolchain exposes no v_writelane builtin, so it is a
// compare against the (uniform) lane number plus a v_cndmask
__device__ __forceinline__ uint32_t wl32(uint32_t val, uint32_t ln, uint32_t old) {
    return (threadIdx.x == ln) ? val : old;
}

struct RBeam {
    uint32_t k0, k1, k2, k3;   // est bits
    uint32_t l0, l1, l2, l3;   // lower bits
    uint32_t d0, d1, d2, d3;   // node id
    uint4* g;                  // spill, indexed by heap index (>= kRegBeam)
};

__device__ __forceinline__ uint32_t rb_key(const RBeam& h, uint32_t idx) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) return rl32(h.k0, ln);
    if (set == 1) return rl32(h.k1, ln);
    if (set == 2) return rl32(h.k2, ln);
    if (set == 3) return rl32(h.k3, ln);
    return bcast_u32(h.g[idx].x);
}
__device__ __forceinline__ void rb_get(const RBeam& h, uint32_t idx, uint32_t& k, uint32_t& l, uint32_t& d) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) { k = rl32(h.k0, ln); l = rl32(h.l0, ln); d = rl32(h.d0, ln); }
    else if (set == 1) { k = rl32(h.k1, ln); l = rl32(h.l1, ln); d = rl32(h.d1, ln); }
    else if (set == 2) { k = rl32(h.k2, ln); l = rl32(h.l2, ln); d = rl32(h.d2, ln); }
    else if (set == 3) { k = rl32(h.k3, ln); l = rl32(h.l3, ln); d = rl32(h.d3, ln); }
    else {
        const uint4 v = h.g[idx];
        k = bcast_u32(v.x); l = bcast_u32(v.y); d = bcast_u32(v.z);
    }
}
__device__ __forceinline__ void rb_put(RBeam& h, uint32_t idx, uint32_t k, uint32_t l, uint32_t d) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) { h.k0 = wl32(k, ln, h.k0); h.l0 = wl32(l, ln, h.l0); h.d0 = wl32(d, ln, h.d0); }
    else if (set == 1) { h.k1 = wl32(k, ln, h.k1); h.l1 = wl32(l, ln, h.l1); h.d1 = wl32(d, ln, h.d1); }
    else if (set == 2) { h.k2 = wl32(k, ln, h.k2); h.l2 = wl32(l, ln, h.l2); h.d2 = wl32(d, ln, h.d2); }
    else if (set == 3) { h.k3 = wl32(k, ln, h.k3); h.l3 = wl32(l, ln, h.l3); h.d3 = wl32(d, ln, h.d3); }
    else if (threadIdx.x == 0) h.g[idx] = make_uint4(k, l, d, 0u);
}
// std::priority_queue<BeamEntry, vector, greater>: comparator "a.est > b.est"
__device__ __forceinline__ void rb_sift_up(RBeam& h, uint32_t hole, uint32_t top, uint32_t vk, uint32_t vl,
                                           uint32_t vd) {
    while (hole > top) {
        const uint32_t p = (hole - 1) >> 1;
        uint32_t pk, pl, pd;
        rb_get(h, p, pk, pl, pd);
        if (!(pk > vk)) break;
        rb_put(h, hole, pk, pl, pd);
        hole = p;
    }
    rb_put(h, hole, vk, vl, vd);
}
__device__ __forceinline__ void rb_adjust(RBeam& h, uint32_t hole, uint32_t len, uint32_t vk, uint32_t vl,
                                          uint32_t vd) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rb_key(h, child) > rb_key(h, child - 1)) --child;
        uint32_t ck, cl, cd;
        rb_get(h, child, ck, cl, cd);
        rb_put(h, hole, ck, cl, cd);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        uint32_t ck, cl, cd;
        rb_get(h, child - 1, ck, cl, cd);
        rb_put(h, hole, ck, cl, cd);
        hole = child - 1;
    }
    rb_sift_up(h, hole, top, vk, vl, vd);
}

struct RNn {                   // BoundedMaxHeap<SearchResult>: max-heap on dist
    uint32_t x0, x1, x2, x3;   // dist bits
    uint32_t i0, i1, i2, i3;   // node id
};
__device__ __forceinline__ uint32_t rn_key(const RNn& h, uint32_t idx) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) return rl32(h.x0, ln);
    if (set == 1) return rl32(h.x1, ln);
    if (set == 2) return rl32(h.x2, ln);
    return rl32(h.x3, ln);
}
__device__ __forceinline__ void rn_get(const RNn& h, uint32_t idx, uint32_t& x, uint32_t& i) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) { x = rl32(h.x0, ln); i = rl32(h.i0, ln); }
    else if (set == 1) { x = rl32(h.x1, ln); i = rl32(h.i1, ln); }
    else if (set == 2) { x = rl32(h.x2, ln); i = rl32(h.i2, ln); }
    else { x = rl32(h.x3, ln); i = rl32(h.i3, ln); }
}
__device__ __forceinline__ void rn_put(RNn& h, uint32_t idx, uint32_t x, uint32_t i) {
    const uint32_t ln = idx & 63u, set = idx >> 6;
    if (set == 0) { h.x0 = wl32(x, ln, h.x0); h.i0 = wl32(i, ln, h.i0); }
    else if (set == 1) { h.x1 = wl32(x, ln, h.x1); h.i1 = wl32(i, ln, h.i1); }
    else if (set == 2) { h.x2 = wl32(x, ln, h.x2); h.i2 = wl32(i, ln, h.i2); }
    else { h.x3 = wl32(x, ln, h.x3); h.i3 = wl32(i, ln, h.i3); }
}
__device__ __forceinline__ void rn_sift_up(RNn& h, uint32_t hole, uint32_t top, uint32_t vx, uint32_t vi) {
    while (hole > top) {
        const uint32_t p = (hole - 1) >> 1;
        uint32_t px, pi;
        rn_get(h, p, px, pi);
        if (!(px < vx)) break;
        rn_put(h, hole, px, pi);
        hole = p;
    }
    rn_put(h, hole, vx, vi);
}
__device__ __forceinline__ void rn_adjust(RNn& h, uint32_t hole, uint32_t len, uint32_t vx, uint32_t vi) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rn_key(h, child) < rn_key(h, child - 1)) --child;
        uint32_t cx, ci;
        rn_get(h, child, cx, ci);
        rn_put(h, hole, cx, ci);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        uint32_t cx, ci;
        rn_get(h, child - 1, cx, ci);
        rn_put(h, hole, cx, ci);
        hole = child - 1;
    }
    rn_sift_up(h, hole, top, vx, vi);
}
// BoundedMaxHeap::push (rabitq_search.hpp:26-35)
__device__ __forceinline__ void rn_push(RNn& h, uint32_t& size, uint32_t k, uint32_t id, uint32_t dist_bits) {
    if (size < k) {
        rn_sift_up(h, size, 0, dist_bits, id);
        ++size;
    } else if (dist_bits < rn_key(h, 0)) {
        if (size > 1) {
            uint32_t vx, vi;
            rn_get(h, size - 1, vx, vi);
            rn_adjust(h, 0, size - 1, vx, vi);
        }
        rn_sift_up(h, size - 1, 0, dist_bits, id);
    }
}
__device__ __forceinline__ void rn_sort(RNn& h, uint32_t size) {  // std::sort_heap
    while (size > 1) {
        uint32_t vx, vi, tx, ti;
        rn_get(h, size - 1, vx, vi);
        rn_get(h, 0, tx, ti);
        rn_put(h, size - 1, tx, ti);
        rn_adjust(h, 0, size - 1, vx, vi);
        --size;
    }
}

// LDS: qm[PW*16] | qv[D*4] | exact[128] | list[64] | slack[128] | ratio[16]
__host__ __device__ inline size_t search4_lds_bytes(uint32_t D, uint32_t PW) {
    return (size_t)PW * 16 + (size_t)D * 4 + 128 + 64 + 128 + 16;
}

#ifndef CPH_SEARCH4_WAVES_PER_SIMD
#define CPH_SEARCH4_WAVES_PER_SIMD 4
#endif

template <int BW, int SD>
__global__ __launch_bounds__(64, CPH_SEARCH4_WAVES_PER_SIMD) void search_kernel4(SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int li = lane & 31;
    const uint32_t D = SD ? SD : a.L.D;
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : a.L.PW;
    const uint32_t k = a.k;
    uint4* qm = reinterpret_cast<uint4*>(smem);
    float* qv = reinterpret_cast<float*>(smem + (size_t)PW * 16);
    float* s_exact = reinterpret_cast<float*>(smem + (size_t)PW * 16 + (size_t)D * 4);
    uint8_t* s_list = reinterpret_cast<uint8_t*>(s_exact + 32);
    float* s_slack = reinterpret_cast<float*>(s_list + 64);
    double* s_ratio = reinterpret_cast<double*>(s_slack + 32);

    const uint32_t slot = blockIdx.x;
    uint32_t* bm = a.bitmaps + (size_t)slot * a.bm_words;
    uint32_t* logi = a.log_ids + (size_t)slot * a.cap;
    const uint32_t FMAXB = 0x7F7FFFFFu;   // bits of FLT_MAX
    const float FMAX = 3.402823466e+38f;
    if (lane < kMaxSlack) s_slack[lane] = a.sc.slack[lane];

    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(a.counter, 1u);
        t = bcast_u32(t);
        if (t >= a.nq) break;
        const uint32_t qi = a.todo ? a.todo[t] : t;

        for (uint32_t w = lane; w < PW; w += 64) qm[w] = a.qmasks[(size_t)qi * PW + w];
        for (uint32_t d = lane; d < D; d += 64) qv[d] = a.queries[(size_t)qi * D + d];
        const QueryHeader hd = a.qhdr[qi];
        if (lane == 0) { s_ratio[0] = 0.0; s_ratio[1] = 0.0; }
        __syncthreads();

        QP qp;
        qp.A = hd.A; qp.B = hd.B; qp.C = hd.C;
        qp.affine_a = a.sc.affine_a; qp.affine_b = a.sc.affine_b; qp.floor = a.sc.ip_qo_floor;
        qp.slack = s_slack[0];
        const float gamma = a.sc.gamma;

        float qnorm;
        {
            float c = 0.0f;
            for (uint32_t i = lane & 7; i < D; i += 8) c = __fmaf_rn(qv[i], qv[i], c);
            qnorm = group_reduce8(c);
        }

        // wave-uniform search state
        RBeam heap{};
        heap.g = a.beam + (size_t)slot * a.cap;
        RNn nn{};
        uint32_t beam_size = 0, nn_size = 0;
        float gamma_q = gamma;
        uint32_t ratio_count = 0;
        uint32_t log_count = 0;
        int slack_batch = 0;
        bool overflow = false;
        uint32_t st_exp = 0, st_exact = 0, st_new = 0, st_push = 0, st_skip = 0;
        uint32_t pf_sink = 0;

        // entry (:95-97)
        {
            const uint32_t ep = hd.entry;
            float dot = group_dot8(qv, a.raw + (size_t)ep * D, D, lane & 7);
            float ex = exact_from_dot(qnorm, a.norm_sq[ep], dot);
            st_exact++;
            rb_put(heap, 0, bcast_u32(__float_as_uint(ex)), 0u, bcast_u32(ep));
            beam_size = 1;
            if (lane == 0) {
                logi[0] = ep;
                atomicOr(&bm[ep >> 5], 1u << (ep & 31));
            }
            log_count = 1;
        }

        for (;;) {
            // ---- pop + termination tests (:106-122), scalar -----------------------------------
            if (beam_size == 0) break;
            uint32_t tk, tl, cur_id;
            rb_get(heap, 0, tk, tl, cur_id);
            if (beam_size > 1) {
                uint32_t vk, vl, vd;
                rb_get(heap, beam_size - 1, vk, vl, vd);
                rb_adjust(heap, 0, beam_size - 1, vk, vl, vd);
            }
            --beam_size;
            const uint32_t next_id = beam_size ? rl32(heap.d0, 0) : cur_id;
            {
                const float cur_est = __uint_as_float(tk), cur_lower = __uint_as_float(tl);
                const float worst = nn_size ? __uint_as_float(rn_key(nn, 0)) : FMAX;
                if (nn_size >= k && cur_est >= gamma_q * worst) break;
                if (nn_size >= k && cur_lower > worst) continue;
            }

            // ---- loads, probe, exact distance of the popped node ---------------------------------
            const uint8_t* blk = a.blocks + (size_t)cur_id * a.L.stride;
            const uint32_t nid = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[li];
            BlockLoads<BW, SD> bl;
            bl.issue(blk, a.L, lane);
            const float* vrow = a.raw + (size_t)cur_id * D;
            const float cur_norm = a.norm_sq[cur_id];
            float vr[16];
            if constexpr (SD == 128) chain_load<16>(vrow, lane & 7, vr);
            __builtin_amdgcn_sched_barrier(0);
            const bool valid = nid != kInvalidNode;
            const bool active = lane < 32 && valid;
            uint32_t old_bits = 0;
            const uint32_t my_bit = 1u << (nid & 31);
            if (active) old_bits = atomicOr(&bm[nid >> 5], my_bit);

            float exact_dist;
            {
                float dot;
                if constexpr (SD == 128) dot = group_reduce8(chain_dot<16>(qv, lane & 7, vr, 0.0f));
                else dot = group_dot8(qv, vrow, D, lane & 7);
                exact_dist = exact_from_dot(qnorm, cur_norm, dot);
            }
            LaneEst v;
            bl.reduce(blk, a.L, qm, lane, v);
            uint32_t pf = 0;
            {
                const uint32_t off = (uint32_t)lane * 64u;
                const uint8_t* nblk = a.blocks + (size_t)next_id * a.L.stride;
                if (off < a.L.stride) pf = *reinterpret_cast<const volatile uint32_t*>(nblk + off);
                if (off < D * 4u)
                    pf ^= *reinterpret_cast<const volatile uint32_t*>(
                        reinterpret_cast<const uint8_t*>(a.raw + (size_t)next_id * D) + off);
            }
            st_exact++;
            st_exp++;
            rn_push(nn, nn_size, k, cur_id, bcast_u32(__float_as_uint(exact_dist)));   // (:133)
            const uint32_t nn_sz = nn_size;
            const float worst0 = nn_size ? __uint_as_float(rn_key(nn, 0)) : FMAX;
            if (!__any(active)) { pf_sink ^= pf; continue; }   // n_neighbors == 0 (:137)

            if (a.sc.num_slack > 0) {
                int lvl = slack_batch < a.sc.num_slack - 1 ? slack_batch : a.sc.num_slack - 1;
                qp.slack = s_slack[lvl];
                ++slack_batch;
            }
            const float dqp = exact_dist;
            const float sq = __builtin_sqrtf(dqp);

            float est, lower;
            if constexpr (BW == 1) {
                stage2_est<1>(qp, v, dqp, sq, est, lower);
            } else {
                float lo1 = stage1_lower<BW>(qp, v, dqp, sq);
                bool surv = (nn_sz < k) || (valid && lo1 < worst0);
                if (__any(surv)) {
                    stage2_est<BW>(qp, v, dqp, sq, est, lower);
                } else {
                    est = FMAX;
                    lower = lo1;
                    st_skip++;
                }
            }

            bool is_new = active && (old_bits & my_bit) == 0;
            if ((a.flags & 1u) && __any(is_new)) {
                for (int j = 0; j < 31; ++j) {
                    uint32_t oj = __shfl(nid, j);
                    bool nj = __shfl((int)is_new, j) != 0;
                    if (nj && lane > j && lane < 32 && oj == nid) is_new = false;
                }
            }
            const uint32_t new_mask = (uint32_t)(__ballot(is_new) & 0xFFFFFFFFull);
            const bool warmup = nn_sz < k;
            const bool cand = is_new && (warmup || (lower < worst0 && est < worst0));
            const uint32_t cand_mask = (uint32_t)(__ballot(cand) & 0xFFFFFFFFull);
            const uint32_t n_new = __popc(new_mask);
            const uint32_t my_rank = __popc(new_mask & ((1u << li) - 1u));
            if (log_count + n_new > a.cap) { overflow = true; break; }
            if (is_new) logi[log_count + my_rank] = nid;
            st_new += n_new;

            // ---- speculative exact L2 of the candidates, 8 per pass --------------------------------
            if (cand_mask) {
                if (cand) s_list[__popc(cand_mask & ((1u << li) - 1u))] = (uint8_t)lane;
                __syncthreads();
                const uint32_t n_cand = __popc(cand_mask);
                const int g = lane >> 3;
                for (uint32_t base = 0; base < n_cand; base += 8) {
                    const bool have = base + g < n_cand;
                    const uint32_t idx = have ? s_list[base + g] : 0;
                    const uint32_t cid_l = (uint32_t)__shfl((int)nid, (int)idx);
                    const uint32_t cid = have ? cid_l : cur_id;
                    float dot = group_dot8(qv, a.raw + (size_t)cid * D, D, lane & 7);
                    float ex = exact_from_dot(qnorm, a.norm_sq[cid], dot);
                    if (have && (lane & 7) == 0) s_exact[idx] = ex;
                }
                st_exact += n_cand;
                __syncthreads();
            }

            // ---- replay of the neighbour loop (:218-273): wave-uniform, scalar control flow -------
            {
                uint32_t m = new_mask;
                while (m) {
                    const int i = __ffs((int)m) - 1;
                    m &= m - 1;
                    const uint32_t id_i = rl32(nid, (uint32_t)i);
                    const uint32_t e_b = rl32(__float_as_uint(est), (uint32_t)i);
                    const uint32_t lo_b = rl32(__float_as_uint(lower), (uint32_t)i);
                    const uint32_t worst_b = nn_size ? rn_key(nn, 0) : FMAXB;
                    const float worst = __uint_as_float(worst_b);
                    const float dabs = (nn_size >= k) ? gamma_q * worst : FMAX;
                    uint32_t key_b = 0, low_b = lo_b;
                    bool push = false;
                    if (warmup) {
                        const float ex = bcast_f32(s_exact[i]);
                        rn_push(nn, nn_size, k, id_i, __float_as_uint(ex));
                        if (ex < dabs) { push = true; key_b = __float_as_uint(ex); low_b = key_b; }
                    } else if (!(__uint_as_float(lo_b) >= worst)) {
                        const float e = __uint_as_float(e_b);
                        if (e < worst) {
                            const float ex = bcast_f32(s_exact[i]);
                            rn_push(nn, nn_size, k, id_i, __float_as_uint(ex));
                            if (ex < dabs) { push = true; key_b = __float_as_uint(ex); }
                            if (ex > kEpsSmall) {
                                double r = (double)(e / ex);
                                const double rs = s_ratio[0] + r;
                                const double rq = fma(r, r, s_ratio[1]);
                                if (lane == 0) { s_ratio[0] = rs; s_ratio[1] = rq; }
                                ++ratio_count;
                                if (ratio_count >= a.sc.gamma_warmup) {
                                    double cnt = (double)ratio_count;
                                    double mean = rs / cnt;
                                    double var = fma(-mean, mean, rq / cnt);
                                    double sd = sqrt(var < 0.0 ? 0.0 : var);
                                    float gq = gamma * (float)fma((double)a.sc.gamma_beta, sd, 1.0);
                                    gamma_q = (gq < gamma) ? gamma : ((a.sc.gamma_max < gq) ? a.sc.gamma_max : gq);
                                }
                                __syncthreads();   // s_ratio is re-read by the next rerank
                            }
                        } else if (e < dabs) {
                            push = true;
                            key_b = e_b;
                        }
                    }
                    if (push) {
                        rb_sift_up(heap, beam_size, 0, key_b, low_b, id_i);
                        ++beam_size;
                        ++st_push;
                    }
                }
            }
            pf_sink ^= pf;
            log_count += n_new;
        }

        // ---- results (:276; src/bindings.cpp:202-210) -------------------------------------------
        if (!overflow) {
            rn_sort(nn, nn_size);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
                if (j < k) {
                    const uint32_t xb = s == 0 ? nn.x0 : s == 1 ? nn.x1 : s == 2 ? nn.x2 : nn.x3;
                    const uint32_t ib = s == 0 ? nn.i0 : s == 1 ? nn.i1 : s == 2 ? nn.i2 : nn.i3;
                    if (j < nn_size) {
                        a.out_ids[(size_t)qi * k + j] = (int64_t)ib;
                        a.out_dist[(size_t)qi * k + j] = __uint_as_float(xb);
                    } else {
                        a.out_ids[(size_t)qi * k + j] = -1;
                        a.out_dist[(size_t)qi * k + j] = FMAX;
                    }
                }
            }
        }
        if (lane == 0) {
            a.out_count[qi] = nn_size;
            a.status[qi] = overflow ? kStatusOverflow : kStatusOk;
            atomicAdd(&a.stats[0], (unsigned long long)st_exp);
            atomicAdd(&a.stats[1], (unsigned long long)st_exact);
            atomicAdd(&a.stats[2], (unsigned long long)st_new);
            atomicAdd(&a.stats[3], (unsigned long long)st_push);
            atomicAdd(&a.stats[4], (unsigned long long)st_skip);
            if (pf_sink == 0x9E3779B9u) atomicAdd(&a.stats[7], 1ull);
            if (overflow) atomicAdd(&a.stats[5], 1ull);
        }
        __syncthreads();
        if (overflow) {
            for (uint64_t w = lane; w < a.bm_words; w += 64) bm[w] = 0u;
        } else {
            for (uint32_t j = lane; j < log_count; j += 64) bm[logi[j] >> 5] = 0u;
        }
        __syncthreads();
    }
}

}  // namespace cph
