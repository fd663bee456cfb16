// device_search2.h — layer-0 beam search, TWO queries per wavefront (32 lanes each).
//
// Same algorithm, data structures and bit-exact semantics as device_search.h (see there for
// the reference citations); what changes is the lane mapping.  Profiling the one-query-per-wave
// kernel on MI355X showed it VALU-issue-bound (SQ_ACTIVE_INST_VALU ~95 % of SIMD time at 5
// waves/SIMD, ~630 VALU instructions per expansion, about half of them the single-lane serial
// heap/decision code).  Here lane (q = lane>>5, i = lane&31) serves neighbour i of query slot q:
// the serial sections run on lanes 0 and 32 in the same instruction stream, so their VALU cost
// is shared by two queries, and the epilogue/probe/ballot code likewise covers two blocks.
// Every lane reduces the whole code of its neighbour (no cross-half combine).
//
// Status: bit-exact (the whole GPU parity suite passes with CPH_SEARCH_KERNEL=2) and 33 % fewer
// VALU instructions per expansion, but its 168 VGPRs allow only 3 waves/SIMD and it measures
// 6.2 ms against 5.4 ms for the one-query kernel on the 1M/10k-query case, so it is not the
// default.
#pragma once
#include <hip/hip_runtime.h>

#include "cph_core.h"
#include "device_fastscan.h"
#include "device_search.h"

namespace cph {

constexpr uint32_t kBeamLds2 = 127;  // 7 full levels per query slot, 16 B per entry

struct Beam2 {
    uint4* l;
    uint4* g;
    __device__ __forceinline__ uint4 raw(uint32_t i) const { return i < kBeamLds2 ? l[i] : g[i]; }
    __device__ __forceinline__ void put(uint32_t i, uint4 v) const {
        if (i < kBeamLds2) l[i] = v; else g[i] = v;
    }
};
__device__ __forceinline__ void beam2_sift_up(const Beam2& h, uint32_t hole, uint32_t top, uint4 v) {
    const float vk = __uint_as_float(v.x);
    while (hole > top) {
        const uint32_t p = (hole - 1) >> 1;
        const uint4 pv = h.raw(p);
        if (!(__uint_as_float(pv.x) > vk)) break;
        h.put(hole, pv);
        hole = p;
    }
    h.put(hole, v);
}
__device__ __forceinline__ void beam2_adjust(const Beam2& h, uint32_t hole, uint32_t len, uint4 v) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        uint4 r = h.raw(child);
        const uint4 lft = h.raw(child - 1);
        if (__uint_as_float(r.x) > __uint_as_float(lft.x)) { --child; r = lft; }
        h.put(hole, r);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.put(hole, h.raw(child - 1));
        hole = child - 1;
    }
    beam2_sift_up(h, hole, top, v);
}

// Whole code of neighbour `li` reduced by one lane.
template <int BW, int SD>
struct LaneCodes {
    static constexpr bool kStatic = (SD >= 128);
    static constexpr int kT = kStatic ? BW * (SD / 32) : 4;
    static constexpr int kNH = (kT / 4 >= 2) ? 2 : 1;
    static constexpr int kChunks = kStatic ? kT / 4 : 1;   // 16-B chunks per neighbour
    uint4 c[kChunks];
    uint4 aux;

    __device__ __forceinline__ void issue(const uint8_t* __restrict__ blk, const DevLayout& L, int li) {
        if constexpr (kStatic) {
            constexpr int CPL = kChunks / kNH;
            const uint4* cp = reinterpret_cast<const uint4*>(blk) + li;
#pragma unroll
            for (int ck = 0; ck < kChunks; ++ck) {
                const int h = ck / CPL, k = ck % CPL;
                c[ck] = cp[k * kNH * 32 + h * 32];
            }
        }
        aux = reinterpret_cast<const uint4*>(blk + L.aux_off)[li];
    }

    __device__ __forceinline__ void reduce(const uint8_t* __restrict__ blk, const DevLayout& L,
                                           const uint4* qm, int li, LaneEst& o) {
        uint32_t S[BW];
#pragma unroll
        for (int b = 0; b < BW; ++b) S[b] = 0;
        if constexpr (kStatic) {
            constexpr int PW = SD / 32;
#pragma unroll
            for (int ck = 0; ck < kChunks; ++ck) {
                const int b = (ck * 4) / PW, w0 = (ck * 4) % PW;
                uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                acc4(c[ck].x, qm[w0 + 0], a0, a1, a2, a3);
                acc4(c[ck].y, qm[w0 + 1], a0, a1, a2, a3);
                acc4(c[ck].z, qm[w0 + 2], a0, a1, a2, a3);
                acc4(c[ck].w, qm[w0 + 3], a0, a1, a2, a3);
                S[b] += a0 + 2 * a1 + 4 * a2 + 8 * a3;
            }
        } else {
            const uint32_t PW = L.PW;
            if (L.wide) {
                const uint32_t NH = L.NH, CPL = L.CPL;
                const uint4* cp = reinterpret_cast<const uint4*>(blk) + li;
#pragma unroll
                for (int b = 0; b < BW; ++b) {
                    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    for (uint32_t w0 = 0; w0 < PW; w0 += 4) {
                        const uint32_t ck = (b * PW + w0) >> 2;
                        const uint32_t h = NH == 2 ? ck / CPL : 0, k = NH == 2 ? ck % CPL : ck;
                        const uint4 cc = cp[k * NH * 32 + h * 32];
                        acc4(cc.x, qm[w0 + 0], a0, a1, a2, a3);
                        acc4(cc.y, qm[w0 + 1], a0, a1, a2, a3);
                        acc4(cc.z, qm[w0 + 2], a0, a1, a2, a3);
                        acc4(cc.w, qm[w0 + 3], a0, a1, a2, a3);
                    }
                    S[b] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
                }
            } else {
                const uint32_t* cw = reinterpret_cast<const uint32_t*>(blk) + li;
#pragma unroll
                for (int b = 0; b < BW; ++b) {
                    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    for (uint32_t w = 0; w < PW; ++w) acc4(cw[(b * PW + w) * 32], qm[w], a0, a1, a2, a3);
                    S[b] = a0 + 2 * a1 + 4 * a2 + 8 * a3;
                }
            }
        }
        uint32_t t = 0;
#pragma unroll
        for (int b = 0; b < BW; ++b) t += S[b] << (BW - 1 - b);
        o.nbit = t;
        o.msb = S[0];
        if constexpr (BW >= 2) o.msb2 = 2 * S[0] + S[1]; else o.msb2 = S[0];
        o.nop = __uint_as_float(aux.x);
        o.ip_qo = __uint_as_float(aux.y);
        o.ip_cp = __uint_as_float(aux.z);
        o.pop = aux.w & 0xFFFFu;
        o.wpop = aux.w >> 16;
    }
};

// per query slot: qm[PW*16] | qv[D*4] | nn[k*8] | est lower exact ids (4*128) | list[64] | beam
__host__ __device__ inline size_t search2_slot_bytes(uint32_t D, uint32_t PW, uint32_t k) {
    size_t b = (size_t)PW * 16 + (size_t)D * 4 + (size_t)k * 8 + 4 * 128 + 64;
    b = (b + 15) & ~(size_t)15;
    return b + 16 * (kBeamLds2 + 1);
}
__host__ __device__ inline size_t search2_lds_bytes(uint32_t D, uint32_t PW, uint32_t k) {
    return 2 * search2_slot_bytes(D, PW, k) + 128;
}

// 3 waves/SIMD (168 VGPRs, no spills) measured best of {3,4,5} for this variant
#ifndef CPH_SEARCH2_WAVES_PER_SIMD
#define CPH_SEARCH2_WAVES_PER_SIMD 3
#endif

template <int BW, int SD>
__global__ __launch_bounds__(64, CPH_SEARCH2_WAVES_PER_SIMD) void search_kernel2(SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int half = lane >> 5;
    const int li = lane & 31;
    const int hbase = half << 5;
    const uint32_t D = SD ? SD : a.L.D;
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : a.L.PW;
    const uint32_t k = a.k;
    const size_t slot_bytes = search2_slot_bytes(D, PW, k);
    float* s_slack = reinterpret_cast<float*>(smem);
    unsigned char* my = smem + 128 + (size_t)half * slot_bytes;
    uint4* qm = reinterpret_cast<uint4*>(my);
    float* qv = reinterpret_cast<float*>(my + (size_t)PW * 16);
    Result* nn = reinterpret_cast<Result*>(my + (size_t)PW * 16 + (size_t)D * 4);
    float* s_est = reinterpret_cast<float*>(my + (size_t)PW * 16 + (size_t)D * 4 + (size_t)k * 8);
    float* s_lower = s_est + 32;
    float* s_exact = s_lower + 32;
    uint32_t* s_ids = reinterpret_cast<uint32_t*>(s_exact + 32);
    uint8_t* s_list = reinterpret_cast<uint8_t*>(s_ids + 32);
    uint4* s_beam = reinterpret_cast<uint4*>(my + slot_bytes - 16 * (kBeamLds2 + 1));

    // global scratch slot of this half
    const uint32_t slot = blockIdx.x * 2 + half;
    uint32_t* bm = a.bitmaps + (size_t)slot * a.bm_words;
    uint32_t* logi = a.log_ids + (size_t)slot * a.cap;
    Beam2 heap;
    heap.l = s_beam;
    heap.g = a.beam + (size_t)slot * a.cap;
    const float FMAX = 3.402823466e+38f;
    if (lane < kMaxSlack) s_slack[lane] = a.sc.slack[lane];
    __syncthreads();

    auto hb32 = [&](uint32_t v) -> uint32_t { return (uint32_t)__shfl((int)v, hbase); };
    auto hbf = [&](float v) -> float { return __shfl(v, hbase); };

    // ---- per-half state (uniform within a half unless noted) -------------------------------
    bool have_q = false, exhausted = false;
    uint32_t qi = 0;
    QP qp{};
    float qnorm = 0.0f;
    uint32_t log_count = 0;
    int slack_batch = 0;
    bool overflow = false;
    // lane li==0 of each half only
    uint32_t beam_size = 0, nn_size = 0;
    float gamma_q = 0.0f;
    double ratio_sum = 0.0, ratio_sq_sum = 0.0;
    unsigned long long ratio_count = 0;
    unsigned long long st_exp = 0, st_exact = 0, st_new = 0, st_push = 0, st_skip = 0;
    uint32_t pf_sink = 0;
    const float gamma = a.sc.gamma;

    for (;;) {
        // ================= (1) halves without a query dequeue one ============================
        uint32_t t = 0xFFFFFFFFu;
        if (!have_q && !exhausted && li == 0) t = atomicAdd(a.counter, 1u);
        t = hb32(t);
        const bool newq = !have_q && !exhausted && t < a.nq;
        if (!have_q && !exhausted && t >= a.nq) exhausted = true;
        if (__all(exhausted && !have_q)) break;
        if (__any(newq)) {
            QueryHeader hd{};
            if (newq) {
                qi = a.todo ? a.todo[t] : t;
                for (uint32_t w = li; w < PW; w += 32) qm[w] = a.qmasks[(size_t)qi * PW + w];
                for (uint32_t d = li; d < D; d += 32) qv[d] = a.queries[(size_t)qi * D + d];
                hd = a.qhdr[qi];
            }
            __syncthreads();
            if (newq) {
                qp.A = hd.A; qp.B = hd.B; qp.C = hd.C;
                qp.affine_a = a.sc.affine_a; qp.affine_b = a.sc.affine_b; qp.floor = a.sc.ip_qo_floor;
                qp.slack = s_slack[0];
                float c = 0.0f;
                for (uint32_t i = lane & 7; i < D; i += 8) c = __fmaf_rn(qv[i], qv[i], c);
                qnorm = group_reduce8(c);
                const uint32_t ep = hd.entry;
                float dot = group_dot8(qv, a.raw + (size_t)ep * D, D, lane & 7);
                float ex = exact_from_dot(qnorm, a.norm_sq[ep], dot);
                beam_size = 0; nn_size = 0; gamma_q = gamma;
                ratio_sum = 0.0; ratio_sq_sum = 0.0; ratio_count = 0;
                st_exp = 0; st_exact = 1; st_new = 0; st_push = 0; st_skip = 0;
                slack_batch = 0; overflow = false;
                if (li == 0) {
                    logi[0] = ep;
                    heap.put(0, make_uint4(__float_as_uint(ex), __float_as_uint(0.0f), ep, 0u));
                    beam_size = 1;
                    atomicOr(&bm[ep >> 5], 1u << (ep & 31));
                }
                log_count = 1;
                have_q = true;
            }
            __syncthreads();
        }

        // ================= (2) pop + termination tests (lanes 0 and 32) ======================
        uint32_t state = 0;  // 0 = done, 2 = expand, 3 = idle (no query)
        uint32_t cur_id = 0, next_id = 0;
        if (!have_q) state = 3;
        if (have_q && li == 0) {
            state = 0;
            while (beam_size > 0) {
                const uint4 top = heap.raw(0);
                if (beam_size > 1) beam2_adjust(heap, 0, beam_size - 1, heap.raw(beam_size - 1));
                --beam_size;
                const float cur_est = __uint_as_float(top.x);
                const float cur_lower = __uint_as_float(top.y);
                const float worst = nn_size ? nn[0].dist : FMAX;
                if (nn_size >= k && cur_est >= gamma_q * worst) { state = 0; break; }
                if (nn_size >= k && cur_lower > worst) continue;   // lower-bound pruned pop
                cur_id = top.z;
                next_id = beam_size ? heap.l[0].z : cur_id;
                state = 2;
                break;
            }
        }
        state = hb32(state);
        cur_id = hb32(cur_id);
        next_id = hb32(next_id);
        const bool expand = state == 2;
        const bool done = have_q && state == 0;

        // ================= (3) expansion ==================================================
        uint32_t nid = kInvalidNode;
        bool valid = false, is_new = false, cand = false, warmup = false;
        uint32_t new_mask = 0, cand_mask = 0, n_new = 0;
        float est = FMAX, lower = 0.0f;
        uint32_t pf = 0;
        if (expand) {
            const uint8_t* blk = a.blocks + (size_t)cur_id * a.L.stride;
            nid = reinterpret_cast<const uint32_t*>(blk + a.L.ids_off)[li];
            LaneCodes<BW, SD> bl;
            bl.issue(blk, a.L, li);
            const float* vrow = a.raw + (size_t)cur_id * D;
            const float cur_norm = a.norm_sq[cur_id];
            float vr[16];
            if constexpr (SD == 128) chain_load<16>(vrow, lane & 7, vr);
            __builtin_amdgcn_sched_barrier(0);
            valid = nid != kInvalidNode;
            uint32_t old_bits = 0;
            const uint32_t my_bit = 1u << (nid & 31);
            if (valid) old_bits = atomicOr(&bm[nid >> 5], my_bit);

            float exact_dist;
            {
                float dot;
                if constexpr (SD == 128) dot = group_reduce8(chain_dot<16>(qv, lane & 7, vr, 0.0f));
                else dot = group_dot8(qv, vrow, D, lane & 7);
                exact_dist = exact_from_dot(qnorm, cur_norm, dot);
            }
            LaneEst v;
            bl.reduce(blk, a.L, qm, li, v);
            {
                const uint32_t off = (uint32_t)li * 128u;   // 32 lanes x 2 dwords cover 4 KB
                const uint8_t* nblk = a.blocks + (size_t)next_id * a.L.stride;
                if (off < a.L.stride) pf = *reinterpret_cast<const volatile uint32_t*>(nblk + off);
                if (off + 64 < a.L.stride) pf ^= *reinterpret_cast<const volatile uint32_t*>(nblk + off + 64);
                if ((uint32_t)li * 64u < D * 4u)
                    pf ^= *reinterpret_cast<const volatile uint32_t*>(
                        reinterpret_cast<const uint8_t*>(a.raw + (size_t)next_id * D) + li * 64u);
            }
            st_exact++;
            st_exp++;
            if (li == 0) nn_push(nn, nn_size, k, Result{cur_id, exact_dist});
            const uint32_t nn_sz = hb32(nn_size);
            float worst0;
            {
                float w = 0.0f;
                if (li == 0) w = nn_size ? nn[0].dist : FMAX;
                worst0 = hbf(w);
            }
            const uint64_t hmask = 0xFFFFFFFFull << hbase;
            const bool any_valid = (__ballot(valid) & hmask) != 0;
            if (any_valid) {
                if (a.sc.num_slack > 0) {
                    int lvl = slack_batch < a.sc.num_slack - 1 ? slack_batch : a.sc.num_slack - 1;
                    qp.slack = s_slack[lvl];
                    ++slack_batch;
                }
                const float dqp = exact_dist;
                const float sq = __builtin_sqrtf(dqp);
                if constexpr (BW == 1) {
                    stage2_est<1>(qp, v, dqp, sq, est, lower);
                } else {
                    float lo1 = stage1_lower<BW>(qp, v, dqp, sq);
                    bool surv = (nn_sz < k) || (valid && lo1 < worst0);
                    if ((__ballot(surv) & hmask) != 0) {
                        stage2_est<BW>(qp, v, dqp, sq, est, lower);
                    } else {
                        est = FMAX;
                        lower = lo1;
                        st_skip++;
                    }
                }
                is_new = valid && (old_bits & my_bit) == 0;
                if ((a.flags & 1u) && (__ballot(is_new) & hmask) != 0) {
                    for (int j = 0; j < 31; ++j) {
                        uint32_t oj = __shfl(nid, hbase + j);
                        bool nj = __shfl((int)is_new, hbase + j) != 0;
                        if (nj && li > j && oj == nid) is_new = false;
                    }
                }
                new_mask = (uint32_t)(__ballot(is_new) >> hbase);
                warmup = nn_sz < k;
                cand = is_new && (warmup || (lower < worst0 && est < worst0));
                cand_mask = (uint32_t)(__ballot(cand) >> hbase);
                n_new = __popc(new_mask);
                if (log_count + n_new > a.cap) {
                    overflow = true;
                    is_new = false; cand = false; new_mask = 0; cand_mask = 0; n_new = 0;
                } else {
                    const uint32_t my_rank = __popc(new_mask & ((1u << li) - 1u));
                    if (is_new) logi[log_count + my_rank] = nid;
                }
                s_est[li] = est;
                s_lower[li] = lower;
                s_ids[li] = nid;
                if (cand) s_list[__popc(cand_mask & ((1u << li) - 1u))] = (uint8_t)li;
                st_new += n_new;
            }
        }
        __syncthreads();

        // ---- speculative exact L2 of the candidates: 4 lane groups per half, 4 per pass ---------
        {
            const uint32_t n_cand = expand ? __popc(cand_mask) : 0;
            uint32_t n_max = n_cand;
            {
                const uint32_t other = (uint32_t)__shfl((int)n_cand, hbase ^ 32);
                n_max = n_cand > other ? n_cand : other;
            }
            const int g = li >> 3;
            for (uint32_t base = 0; base < n_max; base += 4) {
                const bool have = base + g < n_cand;
                const uint32_t idx = have ? s_list[base + g] : 0;
                const uint32_t cid = have ? s_ids[idx] : 0;
                float dot = group_dot8(qv, a.raw + (size_t)cid * D, D, lane & 7);
                float ex = exact_from_dot(qnorm, a.norm_sq[cid], dot);
                if (have && (lane & 7) == 0) s_exact[idx] = ex;
            }
            st_exact += n_cand;
        }
        __syncthreads();

        // ---- serial replay of the neighbour loop, lanes 0 and 32 -----------------------------
        if (expand && li == 0 && !overflow) {
            uint32_t m = new_mask;
            while (m) {
                const int i = __ffs((int)m) - 1;
                m &= m - 1;
                const uint32_t id_i = s_ids[i];
                const float worst = nn_size ? nn[0].dist : FMAX;
                const float dabs = (nn_size >= k) ? gamma_q * worst : FMAX;
                float key = 0.0f, lo = 0.0f;
                bool push = false;
                if (warmup) {
                    const float ex = s_exact[i];
                    nn_push(nn, nn_size, k, Result{id_i, ex});
                    if (ex < dabs) { push = true; key = ex; lo = ex; }
                } else {
                    const float e = s_est[i];
                    lo = s_lower[i];
                    if (lo >= worst) continue;
                    if (e < worst) {
                        const float ex = s_exact[i];
                        nn_push(nn, nn_size, k, Result{id_i, ex});
                        if (ex < dabs) { push = true; key = ex; }
                        if (ex > kEpsSmall) {
                            double r = (double)(e / ex);
                            ratio_sum += r;
                            ratio_sq_sum = fma(r, r, ratio_sq_sum);
                            ++ratio_count;
                            if (ratio_count >= a.sc.gamma_warmup) {
                                double cnt = (double)ratio_count;
                                double mean = ratio_sum / cnt;
                                double var = fma(-mean, mean, ratio_sq_sum / cnt);
                                double sd = sqrt(var < 0.0 ? 0.0 : var);
                                float gq = gamma * (float)fma((double)a.sc.gamma_beta, sd, 1.0);
                                gamma_q = (gq < gamma) ? gamma : ((a.sc.gamma_max < gq) ? a.sc.gamma_max : gq);
                            }
                        }
                    } else if (e < dabs) {
                        push = true;
                        key = e;
                    }
                }
                if (push) {
                    beam2_sift_up(heap, beam_size, 0,
                                  make_uint4(__float_as_uint(key), __float_as_uint(lo), id_i, 0u));
                    ++beam_size;
                    ++st_push;
                }
            }
        }
        if (expand) {
            pf_sink ^= pf;
            log_count += n_new;
        }
        __syncthreads();

        // ================= (4) finished queries: results, un-mark, release the slot ============
        const bool fin = done || (have_q && overflow);
        if (__any(fin)) {
            if (fin && li == 0 && !overflow) nn_sort(nn, nn_size);
            __syncthreads();
            if (fin) {
                const uint32_t nn_final = hb32(nn_size);
                if (!overflow) {
                    for (uint32_t j = li; j < k; j += 32) {
                        if (j < nn_final) {
                            a.out_ids[(size_t)qi * k + j] = (int64_t)nn[j].id;
                            a.out_dist[(size_t)qi * k + j] = nn[j].dist;
                        } else {
                            a.out_ids[(size_t)qi * k + j] = -1;
                            a.out_dist[(size_t)qi * k + j] = FMAX;
                        }
                    }
                }
                if (li == 0) {
                    a.out_count[qi] = nn_final;
                    a.status[qi] = overflow ? kStatusOverflow : kStatusOk;
                    atomicAdd(&a.stats[0], st_exp);
                    atomicAdd(&a.stats[1], st_exact);
                    atomicAdd(&a.stats[2], st_new);
                    atomicAdd(&a.stats[3], st_push);
                    atomicAdd(&a.stats[4], st_skip);
                    if (overflow) atomicAdd(&a.stats[5], 1ull);
                    if (pf_sink == 0x9E3779B9u) atomicAdd(&a.stats[7], 1ull);
                }
            }
            __syncthreads();
            if (fin) {
                if (overflow) {
                    for (uint64_t w = li; w < a.bm_words; w += 32) bm[w] = 0u;
                } else {
                    for (uint32_t j = li; j < log_count; j += 32) bm[logi[j] >> 5] = 0u;
                }
                have_q = false;
                overflow = false;
            }
            __syncthreads();
        }
    }
}

}  // namespace cph
