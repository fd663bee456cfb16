// TEST INFRASTRUCTURE — not product code.
//
// Thin extern "C" hooks around the *real* reference headers under
// /root/reference/include (included where they lie; nothing is copied).  Built
// only in the authoring container by oracle/Makefile into
// oracle/_ref/libcph_refhooks.so and used to (a) generate the golden vectors in
// tests/golden/ and (b) pin oracle/cph_oracle.cpp (the CPU restatement).
//
// Every hook instantiates a reference template for a runtime (D, BitWidth) and
// forwards raw buffers.  Reference entry points exercised:
//   encoder/rabitq_encoder.hpp:73-79,98-136,197-209  encode_query_raw/build_lut
//   encoder/rotation.hpp:34-51                        apply_copy
//   distance/fastscan_kernel.hpp:17-87,197-217,349-368 integer FastScan sums
//   distance/fastscan_kernel.hpp:89-194,220-346,371-425 fp32 epilogues
//   core/memory.hpp:65-95                              l2_distance_simd / dot_product_simd
#include <cphnsw/core/codes.hpp>
#include <cphnsw/core/memory.hpp>
#include <cphnsw/distance/fastscan_kernel.hpp>
#include <cphnsw/distance/fastscan_layout.hpp>
#include <cphnsw/encoder/rabitq_encoder.hpp>
#include <cphnsw/graph/rabitq_graph.hpp>

#include <cstdint>
#include <cstring>

using namespace cphnsw;

#define DISPATCH_D(Dv, ...)                                    \
    switch (Dv) {                                              \
        case 16:   { constexpr size_t D = 16;   __VA_ARGS__; } break; \
        case 32:   { constexpr size_t D = 32;   __VA_ARGS__; } break; \
        case 64:   { constexpr size_t D = 64;   __VA_ARGS__; } break; \
        case 128:  { constexpr size_t D = 128;  __VA_ARGS__; } break; \
        case 256:  { constexpr size_t D = 256;  __VA_ARGS__; } break; \
        case 512:  { constexpr size_t D = 512;  __VA_ARGS__; } break; \
        case 1024: { constexpr size_t D = 1024; __VA_ARGS__; } break; \
        case 2048: { constexpr size_t D = 2048; __VA_ARGS__; } break; \
        default: return -1;                                    \
    }

#define DISPATCH_BW(BWv, ...)                                   \
    switch (BWv) {                                              \
        case 2: { constexpr size_t BW = 2; __VA_ARGS__; } break;       \
        case 4: { constexpr size_t BW = 4; __VA_ARGS__; } break;       \
        default: return -2;                                     \
    }

namespace {

template <size_t D>
void fill_query(RaBitQQuery<D>& q, const uint8_t* lut, const float* p) {
    // p = {coeff_fastscan, coeff_popcount, coeff_constant, affine_a, affine_b,
    //      ip_qo_floor, dot_slack}
    if (lut) std::memcpy(q.lut, lut, sizeof(q.lut));
    q.coeff_fastscan = p[0];
    q.coeff_popcount = p[1];
    q.coeff_constant = p[2];
    q.affine_a = p[3];
    q.affine_b = p[4];
    q.ip_qo_floor = p[5];
    q.dot_slack = p[6];
}

}  // namespace

extern "C" {

// sizeof(VertexSearchData<D,32,BW>) and neighbour-block field offsets.
// out[0]=sizeof vertex, [1]=offset of neighbour block inside the vertex,
// [2..8] = offsets inside the neighbour block of: code_blocks, nop, ip_qo,
// ip_cp, popcounts, weighted_popcounts (or -1), neighbor_ids; [9]=count offset.
int ref_layout(int Dv, int bits, long* out) {
    if (bits == 1) {
        DISPATCH_D(Dv, {
            using V = VertexSearchData<D, 32, 1>;
            using N = typename V::NeighborBlockType;
            out[0] = sizeof(V);
            out[1] = offsetof(V, neighbors);
            out[2] = offsetof(N, code_blocks);
            out[3] = offsetof(N, nop);
            out[4] = offsetof(N, ip_qo);
            out[5] = offsetof(N, ip_cp);
            out[6] = offsetof(N, popcounts);
            out[7] = -1;
            out[8] = offsetof(N, neighbor_ids);
            out[9] = offsetof(N, count);
        })
        return 0;
    }
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        using V = VertexSearchData<D, 32, BW>;
        using N = typename V::NeighborBlockType;
        out[0] = sizeof(V);
        out[1] = offsetof(V, neighbors);
        out[2] = offsetof(N, code_blocks);
        out[3] = offsetof(N, nop);
        out[4] = offsetof(N, ip_qo);
        out[5] = offsetof(N, ip_cp);
        out[6] = offsetof(N, popcounts);
        out[7] = offsetof(N, weighted_popcounts);
        out[8] = offsetof(N, neighbor_ids);
        out[9] = offsetof(N, count);
    }))
    return 0;
}

// rotation + norm_factor scaling + LUT, as Index::search does for a raw query
// (api/hnsw_index.hpp:174-182).  q has `dim` floats; lut gets D/4*16 bytes;
// coeffs gets 3 floats; rotated (optional) gets D floats (post norm_factor).
int ref_encode_query(int dim, int Dv, const float* q, uint8_t* lut, float* coeffs,
                     float* rotated) {
    DISPATCH_D(Dv, {
        RaBitQEncoder<D> enc(static_cast<size_t>(dim));
        alignas(64) float padded[D];
        std::memcpy(padded, q, dim * sizeof(float));
        for (size_t i = dim; i < D; ++i) padded[i] = 0.0f;
        RaBitQQuery<D> enc_q = enc.encode_query_raw(padded);
        std::memcpy(lut, enc_q.lut, sizeof(enc_q.lut));
        coeffs[0] = enc_q.coeff_fastscan;
        coeffs[1] = enc_q.coeff_popcount;
        coeffs[2] = enc_q.coeff_constant;
        if (rotated) enc.rotate_raw_vector(padded, rotated);
    })
    return 0;
}

// Data-side encoder of one vertex' edges (encoder/rabitq_encoder.hpp:138-181 for 1 bit, :287-323 +
// caq_quantize :371-467 for N bits): parent and cnt neighbours (dim floats each) ->
// values u8[cnt][D] (code value per dimension, 0..2^bits-1), aux f32[cnt][3] = {nop, ip_qo, ip_cp},
// pops u32[cnt][2] = {msb popcount, weighted popcount}.  Checker for the GPU edge encoder.
int ref_encode_edges(int dim, int Dv, int bits, const float* parent, const float* nbrs, int cnt,
                     uint8_t* values, float* aux, uint32_t* pops) {
    if (bits == 1) {
        DISPATCH_D(Dv, {
            RaBitQEncoder<D> enc(static_cast<size_t>(dim));
            alignas(64) float pp[D];
            alignas(64) float rp[D];
            std::memcpy(pp, parent, dim * sizeof(float));
            for (size_t i = dim; i < D; ++i) pp[i] = 0.0f;
            enc.rotate_raw_vector(pp, rp);
            for (int e = 0; e < cnt; ++e) {
                BinaryCodeStorage<D> code;
                VertexAuxData a = enc.compute_neighbor_aux(parent, nbrs + (size_t)e * dim, rp, code);
                uint32_t pc = 0;
                for (size_t d = 0; d < D; ++d) {
                    const uint8_t bit = (code.signs[d / 64] >> (d % 64)) & 1u;
                    values[(size_t)e * D + d] = bit;
                    pc += bit;
                }
                aux[3 * e] = a.nop; aux[3 * e + 1] = a.ip_qo; aux[3 * e + 2] = a.ip_cp;
                pops[2 * e] = pc; pops[2 * e + 1] = pc;
            }
        })
        return 0;
    }
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        NbitRaBitQEncoder<D, BW> enc(static_cast<size_t>(dim));
        alignas(64) float pp[D];
        alignas(64) float rp[D];
        std::memcpy(pp, parent, dim * sizeof(float));
        for (size_t i = dim; i < D; ++i) pp[i] = 0.0f;
        enc.rotate_raw_vector(pp, rp);
        for (int e = 0; e < cnt; ++e) {
            auto r = enc.compute_neighbor_aux_nbit(parent, nbrs + (size_t)e * dim, rp);
            for (size_t d = 0; d < D; ++d) {
                uint8_t v = 0;
                for (size_t b = 0; b < BW; ++b)
                    v = (uint8_t)((v << 1) | ((r.code.planes[b][d / 64] >> (d % 64)) & 1u));
                values[(size_t)e * D + d] = v;
            }
            aux[3 * e] = r.aux.nop; aux[3 * e + 1] = r.aux.ip_qo; aux[3 * e + 2] = r.aux.ip_cp;
            pops[2 * e] = r.code.msb_popcount(); pops[2 * e + 1] = r.code.weighted_popcount();
        }
    }))
    return 0;
}

// One plane: block = u8[D/8][32]; out = u32[32].
int ref_fastscan_plane(int Dv, const uint8_t* lut, const uint8_t* block, uint32_t* out) {
    DISPATCH_D(Dv, {
        alignas(64) FastScanCodeBlock<D, 32> b;
        std::memcpy(b.packed, block, sizeof(b.packed));
        alignas(64) uint8_t l[num_sub_segments<D>][16];
        std::memcpy(l, lut, sizeof(l));
        fastscan::compute_inner_products<D>(l, b, out);
    })
    return 0;
}

// planes = u8[BW][D/8][32]
int ref_fastscan_msb(int Dv, int bits, const uint8_t* lut, const uint8_t* planes,
                     uint32_t* out_msb) {
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        alignas(64) NbitFastScanCodeBlock<D, BW, 32> b;
        std::memcpy(&b, planes, sizeof(b));
        alignas(64) uint8_t l[num_sub_segments<D>][16];
        std::memcpy(l, lut, sizeof(l));
        fastscan::compute_msb_only_inner_products<D, BW>(l, b, out_msb);
    }))
    return 0;
}

int ref_fastscan_nbit(int Dv, int bits, const uint8_t* lut, const uint8_t* planes,
                      uint32_t* out_nbit, uint32_t* out_msb) {
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        alignas(64) NbitFastScanCodeBlock<D, BW, 32> b;
        std::memcpy(&b, planes, sizeof(b));
        alignas(64) uint8_t l[num_sub_segments<D>][16];
        std::memcpy(l, lut, sizeof(l));
        fastscan::compute_nbit_inner_products<D, BW>(l, b, out_nbit, out_msb);
    }))
    return 0;
}

// qp = 7 floats (see fill_query).
int ref_convert_1bit(int Dv, const float* qp, const uint32_t* sums, const float* nop,
                     const float* ip_qo, const float* ip_cp, const uint16_t* pop,
                     int count, float dqp, float* est, float* lower) {
    DISPATCH_D(Dv, {
        RaBitQQuery<D> q;
        fill_query<D>(q, nullptr, qp);
        fastscan::convert_to_distances_with_bounds<D>(q, sums, nop, ip_qo, ip_cp, pop,
                                                      (size_t)count, est, lower, dqp);
    })
    return 0;
}

int ref_convert_msb(int Dv, int bits, const float* qp, const uint32_t* msb_sums,
                    const float* nop, const float* ip_qo, const float* ip_cp,
                    const uint16_t* pop, int count, float dqp, float* lower) {
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        RaBitQQuery<D> q;
        fill_query<D>(q, nullptr, qp);
        fastscan::convert_msb_to_lower_bounds<D, BW>(q, msb_sums, nop, ip_qo, ip_cp, pop,
                                                     (size_t)count, lower, dqp);
    }))
    return 0;
}

int ref_convert_nbit(int Dv, int bits, const float* qp, const uint32_t* nbit_sums,
                     const uint32_t* msb_sums, const float* nop, const float* ip_qo,
                     const float* ip_cp, const uint16_t* pop, const uint16_t* wpop,
                     int count, float dqp, float* est, float* lower) {
    DISPATCH_BW(bits, DISPATCH_D(Dv, {
        RaBitQQuery<D> q;
        fill_query<D>(q, nullptr, qp);
        fastscan::convert_nbit_to_distances_with_bounds<D, BW>(
            q, nbit_sums, msb_sums, nop, ip_qo, ip_cp, pop, wpop, (size_t)count, est,
            lower, dqp);
    }))
    return 0;
}

int ref_dot(int Dv, const float* a, const float* b, float* out) {
    DISPATCH_D(Dv, { *out = dot_product_simd<D>(a, b); })
    return 0;
}

int ref_l2(int Dv, const float* a, const float* b, float* out) {
    DISPATCH_D(Dv, { *out = l2_distance_simd<D>(a, b); })
    return 0;
}

// Streaming FastScan throughput of the reference kernels (both N-bit stages per
// block, as search/rabitq_search.hpp:170-200 runs them; 1-bit: :159-168).
// blocks = n_blocks contiguous reference-layout neighbour blocks
// (FastScanNeighborBlock / NbitFastScanNeighborBlock).  Returns a checksum so
// the work cannot be elided.  OpenMP-parallel over blocks.
int ref_fastscan_stream(int Dv, int bits, const uint8_t* lut, const float* qp,
                        const uint8_t* blocks, long n_blocks, float dqp, int reps,
                        double* checksum);

}  // extern "C"

namespace {

template <size_t D>
double stream_1bit(const uint8_t* lut, const float* qp, const uint8_t* blocks,
                   long n_blocks, float dqp, int reps) {
    using N = FastScanNeighborBlock<D, 32, 32>;
    RaBitQQuery<D> q;
    fill_query<D>(q, lut, qp);
    const N* nb = reinterpret_cast<const N*>(blocks);
    double total = 0.0;
    for (int r = 0; r < reps; ++r) {
#pragma omp parallel for reduction(+ : total) schedule(static)
        for (long i = 0; i < n_blocks; ++i) {
            alignas(64) uint32_t sums[32];
            alignas(64) float est[32], lower[32];
            fastscan::compute_inner_products<D>(q.lut, nb[i].code_blocks[0], sums);
            fastscan::convert_to_distances_with_bounds<D>(q, sums, nb[i].nop, nb[i].ip_qo,
                                                          nb[i].ip_cp, nb[i].popcounts, 32,
                                                          est, lower, dqp);
            total += est[i & 31] + lower[(i >> 5) & 31];
        }
    }
    return total;
}

template <size_t D, size_t BW>
double stream_nbit(const uint8_t* lut, const float* qp, const uint8_t* blocks,
                   long n_blocks, float dqp, int reps) {
    using N = NbitFastScanNeighborBlock<D, 32, BW, 32>;
    RaBitQQuery<D> q;
    fill_query<D>(q, lut, qp);
    const N* nb = reinterpret_cast<const N*>(blocks);
    double total = 0.0;
    for (int r = 0; r < reps; ++r) {
#pragma omp parallel for reduction(+ : total) schedule(static)
        for (long i = 0; i < n_blocks; ++i) {
            alignas(64) uint32_t sums[32], msb[32];
            alignas(64) float est[32], lower[32];
            fastscan::compute_msb_only_inner_products<D, BW>(q.lut, nb[i].code_blocks[0], msb);
            fastscan::convert_msb_to_lower_bounds<D, BW>(q, msb, nb[i].nop, nb[i].ip_qo,
                                                         nb[i].ip_cp, nb[i].popcounts, 32,
                                                         lower, dqp);
            fastscan::compute_nbit_inner_products<D, BW>(q.lut, nb[i].code_blocks[0], sums, msb);
            fastscan::convert_nbit_to_distances_with_bounds<D, BW>(
                q, sums, msb, nb[i].nop, nb[i].ip_qo, nb[i].ip_cp, nb[i].popcounts,
                nb[i].weighted_popcounts, 32, est, lower, dqp);
            total += est[i & 31] + lower[(i >> 5) & 31];
        }
    }
    return total;
}

}  // namespace

extern "C" int ref_fastscan_stream(int Dv, int bits, const uint8_t* lut, const float* qp,
                                   const uint8_t* blocks, long n_blocks, float dqp,
                                   int reps, double* checksum) {
    if (bits == 1) {
        DISPATCH_D(Dv, *checksum = stream_1bit<D>(lut, qp, blocks, n_blocks, dqp, reps))
        return 0;
    }
    DISPATCH_BW(bits, DISPATCH_D(Dv, *checksum = (stream_nbit<D, BW>(lut, qp, blocks, n_blocks, dqp, reps))))
    return 0;
}
