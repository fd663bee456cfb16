// cph_core.h — shared host/device definitions of the MI355X-native CP-HNSW hot path.
//
// Data layout in HBM.  The reference stores, per vertex, a
// VertexSearchData<D,32,BW> (graph/rabitq_graph.hpp:19-29): the vertex' own code (never
// read at query time) followed by the neighbour block (bit-plane FastScan code blocks +
// nop/ip_qo/ip_cp/popcounts/ids, distance/fastscan_layout.hpp:51-155).  The device copy
// keeps only what the query path reads and re-lays it out for a 64-lane wavefront:
//
//   device block (one per vertex, `stride` bytes, 64-B aligned)
//     codes   32 neighbours x BW planes x PW dwords, bit t of dword w of plane b of
//             neighbour i = code bit of dimension 32w+t.  "wide" (D >= 128): 16-B chunks
//             ordered [k][half][neighbour] so that one wave instruction reads 1 KiB
//             contiguous; lane (half h, neighbour i) owns chunks h*CPL .. h*CPL+CPL-1 of
//             its neighbour's BW*PW dwords (plane-major).  "small" (D < 128): dwords
//             ordered [plane*PW+w][neighbour].
//     aux     32 x {nop, ip_qo, ip_cp, popcount | weighted_popcount << 16}   (16 B each)
//     ids     32 x u32
//     count   u32
#pragma once
#include <cstddef>
#include <cstdint>

namespace cph {

constexpr uint32_t kInvalidNode = 0xFFFFFFFFu;
constexpr int kR = 32;                 // graph degree, src/bindings.cpp:42
constexpr float kEpsTiny = 1e-20f;     // core/constants.hpp:12
constexpr float kEpsSmall = 1e-12f;    // :13
constexpr float kEpsMedium = 1e-10f;   // :14
constexpr int kMaxSlack = 32;          // :29

struct DevLayout {
    uint32_t D;        // padded dimension (power of two, 16..2048)
    uint32_t BW;       // bits per dimension (1, 2, 4)
    uint32_t PW;       // dwords per plane per neighbour = max(1, D/32)
    uint32_t wide;     // D >= 128
    uint32_t NH;       // lane halves that carry codes (wide: 2 if BW*PW >= 8 else 1)
    uint32_t CPL;      // 16-B chunks per lane (wide)
    uint32_t codes_bytes;
    uint32_t aux_off;  // byte offsets inside a block
    uint32_t ids_off;
    uint32_t count_off;
    uint32_t stride;   // block bytes (multiple of 64)
};

#ifndef CPH_BLOCK_ALIGN
#define CPH_BLOCK_ALIGN 64u   // (128 -- a block never straddles one more 128-byte line than it needs -- measured: see DESIGN.md section 6)
#endif
// The byte offsets of make_dev_layout as compile-time constants, for the kernels with a compile-time D: a block's
// address arithmetic then needs neither scalar registers for the layout nor (what the compiler did when it ran out of
// them) a reload of it from the kernarg segment inside the expansion loop.
template <int BW, int SD>
struct StaticLayout {
    static constexpr uint32_t kCodesBytes = 32u * (uint32_t)BW * (SD >= 32 ? (uint32_t)SD / 32u : 1u) * 4u;
    static constexpr uint32_t kAuxOff = kCodesBytes;
    static constexpr uint32_t kIdsOff = kAuxOff + 32u * 16u;
    static constexpr uint32_t kCountOff = kIdsOff + 32u * 4u;
    static constexpr uint32_t kStride = (kCountOff + 4u + CPH_BLOCK_ALIGN - 1u) / CPH_BLOCK_ALIGN * CPH_BLOCK_ALIGN;
};

inline DevLayout make_dev_layout(uint32_t D, uint32_t BW) {
    DevLayout L{};
    L.D = D;
    L.BW = BW;
    L.PW = D >= 32 ? D / 32 : 1;
    L.wide = D >= 128 ? 1 : 0;
    uint32_t T = BW * L.PW;  // dwords per neighbour
    if (L.wide) {
        L.NH = (T / 4 >= 2) ? 2 : 1;
        L.CPL = T / 4 / L.NH;
    } else {
        L.NH = 1;
        L.CPL = 0;
    }
    L.codes_bytes = 32 * T * 4;
    L.aux_off = L.codes_bytes;           // multiple of 128
    L.ids_off = L.aux_off + 32 * 16;
    L.count_off = L.ids_off + 32 * 4;
    L.stride = (L.count_off + 4 + CPH_BLOCK_ALIGN - 1) / CPH_BLOCK_ALIGN * CPH_BLOCK_ALIGN;
    return L;
}

// Reference (file) layout of VertexSearchData<D,32,BW>; SURVEY.md §5.4.
struct RefLayout {
    size_t vertex_bytes, nb_off;
    size_t codes, nop, ip_qo, ip_cp, pop, wpop, ids, count;  // inside the neighbour block
    size_t plane_bytes, nb_bytes;
};

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline RefLayout make_ref_layout(size_t D, size_t bw) {
    RefLayout L{};
    size_t words = (D + 63) / 64;
    size_t code = round_up(round_up(bw * words * 8, 64) + 8, 64);
    L.plane_bytes = (((D + 3) / 4 + 1) / 2) * 32;
    size_t o = 0;
    L.codes = o; o += bw * round_up(L.plane_bytes, 64);
    L.nop = o;   o += 128;
    L.ip_qo = o; o += 128;
    L.ip_cp = o; o += 128;
    L.pop = o;   o += 64;
    if (bw > 1) { L.wpop = o; o += 64; } else { L.wpop = (size_t)-1; }
    L.ids = o;   o += 128;
    L.count = o; o += 4;
    L.nb_bytes = round_up(o, 64);
    L.nb_off = code;
    L.vertex_bytes = code + L.nb_bytes;
    return L;
}

// Per-query record handed to the search kernel (host- or device-encoded).
// qparams mirror RaBitQQuery's scalars (core/codes.hpp:79-93).
struct QueryHeader {
    float A, B, C;        // coeff_fastscan, coeff_popcount, coeff_constant
    uint32_t entry;       // layer-0 entry point after the upper-layer descent
};

// Search-wide constants (CalibrationSnapshot, api/hnsw_index.hpp:33-58).
struct SearchConsts {
    float affine_a, affine_b, ip_qo_floor;
    float gamma, gamma_max, gamma_beta;
    uint64_t gamma_warmup;
    int32_t num_slack;
    float slack[kMaxSlack];
};

enum QueryStatus : uint32_t { kStatusOk = 0, kStatusOverflow = 1 };

}  // namespace cph
