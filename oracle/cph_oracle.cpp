// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product path
// (rabitq-ann-search_amd/) never links, imports or calls it.
//
// CPU restatement of the reference's layer-0 hot path (CP-HNSW,
// indrajeetadityaroy9/rabitq-ann-search), written from the algorithm, scalar C++17.
// Parity pinning: every function below is checked bit-for-bit against the compiled
// reference (oracle/_ref, built from /root/reference by oracle/Makefile) in the
// authoring container, and against the committed golden vectors in tests/golden/
// (generated from the reference by tests/golden/make_golden.py) everywhere else.
//
// Build: -ffp-contract=off — every fused multiply-add below is written explicitly
// (std::fmaf / std::fma) exactly where the compiled reference has one, either because
// the reference source uses an FMA intrinsic or because GCC's default
// -ffp-contract=fast fused that scalar expression (verified in the reference's
// x86-64 assembly; noted per function).
//
// Reference citations are relative to /root/reference/include/cphnsw/.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include <omp.h>

namespace {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr float kEpsTiny = 1e-20f;    // core/constants.hpp:12
constexpr float kEpsSmall = 1e-12f;   // :13
constexpr float kEpsMedium = 1e-10f;  // :14

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------
// Layout of VertexSearchData<D,32,BW>  (graph/rabitq_graph.hpp:19-29,
// distance/fastscan_layout.hpp:51-92,114-155, core/codes.hpp:12-43,96-135)
// ---------------------------------------------------------------------------------
struct Layout {
    size_t vertex_bytes, nb_off;                     // neighbour block offset in vertex
    size_t codes, nop, ip_qo, ip_cp, pop, wpop, ids, count;  // offsets inside nb block
    size_t plane_bytes;                              // D/8 * 32
};

Layout make_layout(size_t D, size_t bw) {
    Layout L{};
    size_t words = (D + 63) / 64;
    size_t storage = round_up(bw * words * 8, 64);
    size_t code = round_up(storage + 8, 64);  // + nop, ip_qo; struct alignas(64)
    size_t sub_pairs = ((D + 3) / 4 + 1) / 2;
    L.plane_bytes = sub_pairs * 32;
    size_t o = 0;
    L.codes = o; o += bw * round_up(L.plane_bytes, 64);
    L.nop = o;   o += 128;
    L.ip_qo = o; o += 128;
    L.ip_cp = o; o += 128;
    L.pop = o;   o += 64;
    if (bw > 1) { L.wpop = o; o += 64; } else { L.wpop = (size_t)-1; }
    L.ids = o;   o += 128;
    L.count = o; o += 4;
    size_t nb = round_up(o, 64);
    L.nb_off = code;
    L.vertex_bytes = code + nb;
    return L;
}

// ---------------------------------------------------------------------------------
// Rotation: 3 x (random ±1 diagonal -> unnormalised Walsh-Hadamard)
// encoder/rotation.hpp:15-67, encoder/transform/fht.hpp:23-57
// ---------------------------------------------------------------------------------
struct Rotation {
    size_t D;
    std::vector<float> signs;  // [3][D]
    explicit Rotation(size_t D_, uint64_t seed = 42) : D(D_), signs(3 * D_) {
        // rotation.hpp:23-31: one mt19937_64 stream, uniform_int_distribution<int>(0,1)
        // drawn layer-major.  libstdc++ maps a [0,1] range on a 64-bit URBG to the top
        // bit of each draw (checked against the reference for every supported D).
        std::mt19937_64 rng(seed);
        for (size_t i = 0; i < 3 * D; ++i) signs[i] = (rng() >> 63) ? 1.0f : -1.0f;
    }
    static void fht(float* v, size_t len) {
        // fht.hpp:26-45: stages h=1,2,4 keep (a+b) in the lower slot and (upper-lower)
        // in the upper slot; fht.hpp:47-56: stages h>=8 store (lower-upper) on top.
        for (size_t h = 1; h < len; h *= 2) {
            for (size_t i = 0; i < len; i += 2 * h) {
                for (size_t j = i; j < i + h; ++j) {
                    float x = v[j], y = v[j + h];
                    v[j] = x + y;
                    v[j + h] = (h < 8) ? (y - x) : (x - y);
                }
            }
        }
    }
    void apply(float* x) const {  // rotation.hpp:34-43
        for (int l = 0; l < 3; ++l) {
            const float* s = &signs[l * D];
            for (size_t i = 0; i < D; ++i) x[i] = x[i] * s[i];
            fht(x, D);
        }
    }
};

// ---------------------------------------------------------------------------------
// Query encoding: encoder/rabitq_encoder.hpp:73-79,197-209 (rotate, scale by D^-1.5)
// and :98-136 (build_lut).  Scalar reference code; GCC fused (buf-vl)*inv_delta+0.5f
// and Df*vl+delta*sum_qu (seen in the reference's assembly) — mirrored with fmaf.
// ---------------------------------------------------------------------------------
struct QueryCode {
    std::vector<uint8_t> lut;  // [D/4][16]
    float A, B, C;             // coeff_fastscan, coeff_popcount, coeff_constant
};

void encode_query(const Rotation& rot, const float* padded /*D*/, QueryCode& out,
                  float* rotated_out) {
    const size_t D = rot.D;
    std::vector<float> buf(padded, padded + D);
    rot.apply(buf.data());
    float d = static_cast<float>(D);
    float norm_factor = 1.0f / (d * std::sqrt(d));  // rabitq_encoder.hpp:37-38
    float inv_sqrt_d = 1.0f / std::sqrt(d);          // :39
    for (size_t i = 0; i < D; ++i) buf[i] = buf[i] * norm_factor;
    if (rotated_out) std::memcpy(rotated_out, buf.data(), D * sizeof(float));

    float vl = buf[0], vmax = buf[0];
    for (size_t i = 1; i < D; ++i) {
        if (buf[i] < vl) vl = buf[i];
        if (buf[i] > vmax) vmax = buf[i];
    }
    float delta = (vmax - vl) / 15.0f;
    if (delta < kEpsTiny) delta = kEpsTiny;
    float inv_delta = 1.0f / delta;
    std::vector<uint8_t> u(D);
    float sum_qu = 0.0f;
    for (size_t i = 0; i < D; ++i) {
        int q = static_cast<int>(std::fmaf(buf[i] - vl, inv_delta, 0.5f));
        if (q < 0) q = 0;
        if (q > 15) q = 15;
        u[i] = static_cast<uint8_t>(q);
        sum_qu += static_cast<float>(q);
    }
    out.lut.assign(D / 4 * 16, 0);
    for (size_t j = 0; j < D / 4; ++j)
        for (unsigned p = 0; p < 16; ++p) {
            uint8_t s = 0;
            for (unsigned b = 0; b < 4; ++b)
                if (p & (1u << b)) s = static_cast<uint8_t>(s + u[4 * j + b]);
            out.lut[j * 16 + p] = s;
        }
    out.A = (2.0f * delta) * inv_sqrt_d;
    out.B = (2.0f * vl) * inv_sqrt_d;
    out.C = -std::fmaf(d, vl, delta * sum_qu) * inv_sqrt_d;
}

// ---------------------------------------------------------------------------------
// FastScan integer sums.  distance/fastscan_kernel.hpp:17-87 (one plane),
// :349-368 (2·S0+S1), :197-217 (Σ 2^(BW-1-b)·S_b and S0).
// plane = u8[D/8][32]; byte [sp][i] = seg(2sp) nibble | seg(2sp+1) nibble << 4.
// ---------------------------------------------------------------------------------
void plane_sums(size_t D, const uint8_t* lut, const uint8_t* plane, uint32_t* out) {
    for (int i = 0; i < 32; ++i) out[i] = 0;
    for (size_t sp = 0; sp < D / 8; ++sp)
        for (int i = 0; i < 32; ++i) {
            uint8_t c = plane[sp * 32 + i];
            out[i] += lut[(2 * sp) * 16 + (c & 15)];
            out[i] += lut[(2 * sp + 1) * 16 + (c >> 4)];
        }
}

void msb_sums(size_t D, size_t bw, const uint8_t* lut, const uint8_t* planes, uint32_t* out) {
    plane_sums(D, lut, planes, out);
    if (bw >= 2) {
        uint32_t p1[32];
        plane_sums(D, lut, planes + D * 4, p1);
        for (int i = 0; i < 32; ++i) out[i] = 2 * out[i] + p1[i];
    }
}

void nbit_sums(size_t D, size_t bw, const uint8_t* lut, const uint8_t* planes,
               uint32_t* out_nbit, uint32_t* out_msb) {
    for (int i = 0; i < 32; ++i) out_nbit[i] = 0;
    uint32_t ps[32];
    for (size_t b = 0; b < bw; ++b) {
        plane_sums(D, lut, planes + b * D * 4, ps);
        if (b == 0) std::memcpy(out_msb, ps, sizeof(ps));
        uint32_t w = 1u << (bw - 1 - b);
        for (int i = 0; i < 32; ++i) out_nbit[i] += w * ps[i];
    }
}

// ---------------------------------------------------------------------------------
// fp32 epilogues
// ---------------------------------------------------------------------------------
struct QParams {  // RaBitQQuery scalars, core/codes.hpp:79-93
    float A, B, C, affine_a, affine_b, floor, slack;
};

inline float vmax(float a, float b) { return a > b ? a : b; }  // _mm256_max_ps(a,b)
inline float vmin(float a, float b) { return a < b ? a : b; }  // _mm256_min_ps(a,b)

// Shared tail of the AVX2 vector paths fastscan_kernel.hpp:148-169 / :287-317.
inline void est_and_lower(const QParams& q, float ip_approx_est, float ip_approx_lb,
                          float nop, float ip_qo, float ip_cp, float dqp, float sqrt_dqp,
                          float* est, float* lower) {
    float ipq = vmax(ip_qo, q.floor);
    bool good = ipq > kEpsMedium;
    float e = good ? (ip_approx_est - ip_cp) / ipq : 0.0f;
    e = std::fmaf(q.affine_a, e, q.affine_b);
    float dist = std::fmaf(nop, nop, dqp);
    dist = std::fmaf(-(2.0f * nop), e, dist);
    *est = vmax(dist, 0.0f);
    float m = good ? (ip_approx_lb - ip_cp) / ipq : 0.0f;
    m = std::fmaf(q.affine_a, m, q.affine_b);
    float cosu = (m + q.slack) / vmax(sqrt_dqp, kEpsMedium);
    cosu = vmin(vmax(cosu, -1.0f), 1.0f);
    float lo = std::fmaf(nop, nop, dqp);
    lo = std::fmaf(-((2.0f * nop) * sqrt_dqp), cosu, lo);
    lo = vmax(lo, 0.0f);
    *lower = good ? lo : 0.0f;
}

// fastscan_kernel.hpp:89-194.  Lanes below 8 * (count / 8) take the AVX2 vector path (:138-173, explicit FMA
// intrinsics); the remainder count % 8 takes the scalar tail (:176-193), which GCC contracts
// (-ffp-contract=fast) into a DIFFERENT rounding sequence for the inner-product estimate:
//   vector  ip_approx = fma(A, fs, fma(B, pc, C))
//   tail    ip_approx = fma(B, pc, A * fs) + C
// (found by comparing every fused / unfused variant of the tail against the compiled reference,
// oracle/_ref; the rest of the tail -- affine map, distance, bound -- contracts to the vector path's
// sequence).  The reference's graphs have count == 32 on every vertex we have seen, but the v2 format
// allows short lists (fastscan_layout.hpp:51-92) and the search passes batch_count = count
// (rabitq_search.hpp:152-154), so the tail is part of the path.
void convert_1bit(const QParams& q, const uint32_t* fs, const float* nop, const float* ip_qo,
                  const float* ip_cp, const uint16_t* pop, int count, float dqp, float* est,
                  float* lower) {
    if (dqp < kEpsSmall) {  // :112-119 (scalar; GCC fuses nop*nop+dqp)
        for (int i = 0; i < count; ++i) { est[i] = std::fmaf(nop[i], nop[i], dqp); lower[i] = 0.0f; }
        return;
    }
    float s = std::sqrt(dqp);
    const int vec_end = count & ~7;
    for (int i = 0; i < count; ++i) {
        float ipa = i < vec_end ? std::fmaf(q.A, (float)fs[i], std::fmaf(q.B, (float)pop[i], q.C))
                                : std::fmaf(q.B, (float)pop[i], q.A * (float)fs[i]) + q.C;
        est_and_lower(q, ipa, ipa, nop[i], ip_qo[i], ip_cp[i], dqp, s, &est[i], &lower[i]);
    }
}

// fastscan_kernel.hpp:371-425.  Scalar loop; the compiled reference evaluates
//   ip_approx = fma(B, pc, A*msb) + C   (A, B pre-divided by 3 for BW>=2)
//   ip_est    = fma(ip_corrected/ipq, a, b)
//   lower     = fnma(cos, (2*nop)*sqrt_dqp, fma(nop,nop,dqp))   clamped at 0
// (GCC -ffp-contract=fast; read from the assembly of the inlined loop inside
// rabitq_search::search).  popcounts = plane-0 popcounts (rabitq_search.hpp:172-176).
void convert_msb(size_t bw, const QParams& q, const uint32_t* msb, const float* nop,
                 const float* ip_qo, const float* ip_cp, const uint16_t* pop, int count,
                 float dqp, float* lower) {
    float invk = (bw >= 2) ? (1.0f / 3.0f) : 1.0f;
    float A = q.A * invk, B = q.B * invk;
    if (dqp < kEpsSmall) { for (int i = 0; i < count; ++i) lower[i] = 0.0f; return; }
    float s = std::sqrt(dqp);
    for (int i = 0; i < count; ++i) {
        float ipq = (ip_qo[i] < q.floor) ? q.floor : ip_qo[i];  // std::max(a,b)
        if (ipq <= kEpsMedium) { lower[i] = 0.0f; continue; }
        float ipa = std::fmaf(B, (float)pop[i], A * (float)msb[i]) + q.C;
        float e = (ipa - ip_cp[i]) / ipq;
        e = std::fmaf(e, q.affine_a, q.affine_b);
        float c = (e + q.slack) / s;
        c = (c < -1.0f) ? -1.0f : ((1.0f < c) ? 1.0f : c);  // std::clamp
        float lo = std::fmaf(-c, (nop[i] + nop[i]) * s, std::fmaf(nop[i], nop[i], dqp));
        lower[i] = (lo < 0.0f) ? 0.0f : lo;
    }
}

// fastscan_kernel.hpp:220-346.  Vector path (:277-321) below 8 * (count / 8); scalar tail (:324-345) for the
// remainder, as GCC compiles it (see convert_1bit):
//   vector  ip_nbit = fma(A/K, nbit, fma(B/K, wpc, C))     ip_msb = fma(A, msb, fma(B, mpc, C))
//   tail    ip_nbit = fma(B/K, wpc, (A/K) * nbit) + C      ip_msb = fma(A, msb, B * mpc) + C
void convert_nbit(size_t bw, const QParams& q, const uint32_t* nb, const uint32_t* msb,
                  const float* nop, const float* ip_qo, const float* ip_cp,
                  const uint16_t* pop, const uint16_t* wpop, int count, float dqp, float* est,
                  float* lower) {
    float K = (float)((1u << bw) - 1);
    float invK = 1.0f / K;
    float An = q.A * invK, Bn = q.B * invK;
    if (dqp < kEpsSmall) {  // :249-256 (scalar; fused as above)
        for (int i = 0; i < count; ++i) { est[i] = std::fmaf(nop[i], nop[i], dqp); lower[i] = 0.0f; }
        return;
    }
    float s = std::sqrt(dqp);
    const int vec_end = count & ~7;
    for (int i = 0; i < count; ++i) {
        float ipn, ipm;
        if (i < vec_end) {
            ipn = std::fmaf(An, (float)nb[i], std::fmaf(Bn, (float)wpop[i], q.C));
            ipm = std::fmaf(q.A, (float)msb[i], std::fmaf(q.B, (float)pop[i], q.C));
        } else {
            ipn = std::fmaf(Bn, (float)wpop[i], An * (float)nb[i]) + q.C;
            ipm = std::fmaf(q.A, (float)msb[i], q.B * (float)pop[i]) + q.C;
        }
        est_and_lower(q, ipn, ipm, nop[i], ip_qo[i], ip_cp[i], dqp, s, &est[i], &lower[i]);
    }
}

// ---------------------------------------------------------------------------------
// Exact arithmetic: core/memory.hpp:65-95.  8 independent FMA chains (chain j takes
// elements j, j+8, ...), then ((c0+c4)+(c1+c5)) + ((c2+c6)+(c3+c7)).
// ---------------------------------------------------------------------------------
inline float reduce8(const float* c) {
    float s0 = c[0] + c[4], s1 = c[1] + c[5], s2 = c[2] + c[6], s3 = c[3] + c[7];
    return (s0 + s1) + (s2 + s3);
}
float dot8(size_t D, const float* a, const float* b) {
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < D; i += 8)
        for (int j = 0; j < 8; ++j) c[j] = std::fmaf(a[i + j], b[i + j], c[j]);
    return reduce8(c);
}
float l2sq8(size_t D, const float* a, const float* b) {
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < D; i += 8)
        for (int j = 0; j < 8; ++j) { float d = a[i + j] - b[i + j]; c[j] = std::fmaf(d, d, c[j]); }
    return reduce8(c);
}

// ---------------------------------------------------------------------------------
// Binary heaps with libstdc++'s push_heap / pop_heap / sort_heap element movement
// (tie order is observable on integer-valued data; search/rabitq_search.hpp:17-49,
// :79-80 use std::push_heap/pop_heap/sort_heap and std::priority_queue).
// `before(a,b)` is the comparator "a is ordered below b" (std::less semantics for a
// max-heap on the key).
// ---------------------------------------------------------------------------------
template <class T, class Before>
void sift_up(T* h, size_t hole, size_t top, T v, Before before) {
    while (hole > top) {
        size_t parent = (hole - 1) / 2;
        if (!before(h[parent], v)) break;
        h[hole] = h[parent];
        hole = parent;
    }
    h[hole] = v;
}
template <class T, class Before>
void heap_push(T* h, size_t len_after, Before before) {  // last element is the new one
    T v = h[len_after - 1];
    sift_up(h, len_after - 1, 0, v, before);
}
template <class T, class Before>
void adjust(T* h, size_t hole, size_t len, T v, Before before) {
    const size_t top = hole;
    size_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (before(h[child], h[child - 1])) --child;
        h[hole] = h[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    sift_up(h, hole, top, v, before);
}
template <class T, class Before>
void heap_pop(T* h, size_t len, Before before) {  // moves the top to h[len-1]
    if (len <= 1) return;
    T v = h[len - 1];
    h[len - 1] = h[0];
    adjust(h, 0, len - 1, v, before);
}
template <class T, class Before>
void heap_sort(T* h, size_t len, Before before) {
    while (len > 1) { heap_pop(h, len, before); --len; }
}

struct Result { uint32_t id; float dist; };
struct BeamEntry { float est, lower; uint32_t id; };
inline bool result_before(const Result& a, const Result& b) { return a.dist < b.dist; }
// priority_queue<BeamEntry, vector, greater>: comp(a,b) = a > b on est only (:57)
inline bool beam_before(const BeamEntry& a, const BeamEntry& b) { return a.est > b.est; }

// ---------------------------------------------------------------------------------
// Index (v2 file): api/hnsw_index.hpp:217-303 (writer) / :305-443 (reader)
// ---------------------------------------------------------------------------------
struct UpperEdge { uint32_t node; std::vector<uint32_t> nbrs; };

struct Index {
    size_t D = 0, bw = 0, dim = 0, n = 0;
    int max_level = 0;
    uint32_t entry = kInvalid;
    float upper_tau = 0, upper_alpha = 0;
    double mL = 0;
    uint64_t seed = 0;
    uint8_t calib[248];
    uint8_t profile[72];
    std::vector<float> centroid;
    std::vector<int32_t> levels;
    std::vector<float> norm_sq;
    std::vector<float> raw;       // [n][D]
    std::vector<uint8_t> search;  // n * vertex_bytes
    std::vector<std::vector<UpperEdge>> upper;
    Layout L;
    Rotation* rot = nullptr;
    // calibration fields used by search (CalibrationSnapshot, hnsw_index.hpp:33-58)
    float affine_a, affine_b, ip_qo_floor, gamma_max, gamma_beta, search_gamma;
    uint64_t gamma_warmup;
    float slack_levels[32];
    int num_slack_levels;
    ~Index() { delete rot; }

    const uint8_t* nb(uint32_t id) const { return &search[(size_t)id * L.vertex_bytes + L.nb_off]; }
    const float* vec(uint32_t id) const { return &raw[(size_t)id * D]; }
};

template <class T> T rd(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }

Index* load_index(const char* path, std::string& err) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { err = std::string("Cannot open file for reading: ") + path; return nullptr; }
    auto fail = [&](const std::string& m) { err = m; std::fclose(f); return (Index*)nullptr; };
    auto rdn = [&](void* p, size_t b) { return std::fread(p, 1, b, f) == b; };
    uint8_t hdr[68];
    if (!rdn(hdr, 68)) return fail("Read error or truncated file: " + std::string(path));
    if (rd<uint64_t>(hdr) != 0x57534E48504300ULL) return fail("Invalid magic bytes (not a CP-HNSW index file).");
    if (rd<uint32_t>(hdr + 8) != 2) return fail("Unsupported index file version: " + std::to_string(rd<uint32_t>(hdr + 8)));
    Index* ix = new Index();
    ix->D = rd<uint32_t>(hdr + 12);
    uint32_t R = rd<uint32_t>(hdr + 16);
    ix->bw = rd<uint32_t>(hdr + 20);
    ix->dim = rd<uint32_t>(hdr + 24);
    ix->n = rd<uint64_t>(hdr + 28);
    ix->max_level = rd<int32_t>(hdr + 36);
    ix->entry = rd<uint32_t>(hdr + 40);
    ix->upper_tau = rd<float>(hdr + 44);
    ix->upper_alpha = rd<float>(hdr + 48);
    ix->mL = rd<double>(hdr + 52);
    ix->seed = rd<uint64_t>(hdr + 60);
    if (R != 32 || (ix->bw != 1 && ix->bw != 2 && ix->bw != 4)) { delete ix; return fail("Index file template parameters mismatch"); }
    ix->L = make_layout(ix->D, ix->bw);
    bool ok = rdn(ix->calib, 248) && rdn(ix->profile, 72);
    size_t n = ix->n, D = ix->D;
    ix->centroid.resize(ix->dim); ok = ok && rdn(ix->centroid.data(), ix->dim * 4);
    ix->levels.resize(n);   ok = ok && rdn(ix->levels.data(), n * 4);
    ix->norm_sq.resize(n);  ok = ok && rdn(ix->norm_sq.data(), n * 4);
    ix->raw.resize(n * D);  ok = ok && rdn(ix->raw.data(), n * D * 4);
    ix->search.resize(n * ix->L.vertex_bytes); ok = ok && rdn(ix->search.data(), ix->search.size());
    uint32_t nl = 0; ok = ok && rdn(&nl, 4);
    if (ok) {
        ix->upper.resize(nl);
        for (uint32_t l = 0; l < nl && ok; ++l) {
            uint32_t sz = 0; ok = rdn(&sz, 4);
            if (!ok) break;
            ix->upper[l].resize(sz);
            for (uint32_t e = 0; e < sz && ok; ++e) {
                uint32_t cnt = 0;
                ok = rdn(&ix->upper[l][e].node, 4) && rdn(&cnt, 4);
                if (!ok) break;
                ix->upper[l][e].nbrs.resize(cnt);
                if (cnt) ok = rdn(ix->upper[l][e].nbrs.data(), cnt * 4);
            }
        }
    }
    if (!ok) { delete ix; return fail("Read error or truncated file: " + std::string(path)); }
    std::fclose(f);
    const uint8_t* c = ix->calib;
    ix->affine_a = rd<float>(c + 0); ix->affine_b = rd<float>(c + 4); ix->ip_qo_floor = rd<float>(c + 8);
    ix->gamma_max = rd<float>(c + 84); ix->gamma_beta = rd<float>(c + 88);
    ix->gamma_warmup = rd<uint64_t>(c + 96);
    std::memcpy(ix->slack_levels, c + 108, 128);
    ix->num_slack_levels = rd<int32_t>(c + 236);
    ix->search_gamma = rd<float>(c + 240);
    ix->rot = new Rotation(D, ix->seed);
    return ix;
}

// api/hnsw_index.hpp:617-638 + find_edge :468-474
uint32_t greedy_layer(const Index& ix, const float* q, uint32_t ep, int level) {
    float best = l2sq8(ix.D, q, ix.vec(ep));
    uint32_t best_id = ep;
    bool improved = true;
    const auto& layer = ix.upper[level - 1];
    while (improved) {
        improved = false;
        auto it = std::lower_bound(layer.begin(), layer.end(), best_id,
                                   [](const UpperEdge& e, uint32_t v) { return e.node < v; });
        if (it == layer.end() || it->node != best_id) break;
        for (uint32_t nb : it->nbrs) {
            float d = l2sq8(ix.D, q, ix.vec(nb));
            if (d < best) { best = d; best_id = nb; improved = true; }
        }
    }
    return best_id;
}

struct Counters {
    uint64_t expansions, exact_l2, nbr_seen, nbr_new, beam_push, beam_max, stage2_skipped,
        lb_pruned_pops, gamma_breaks;
};

// diagnostic (scripts/probe_locality.py): the ids probed by one query, in order, expansion by expansion
thread_local std::vector<uint32_t>* g_probe_trace = nullptr;   // [vertex expanded | 0x80000000, then its count neighbour ids]*

// search/rabitq_search.hpp:60-277 with the Index::search prologue
// (api/hnsw_index.hpp:168-211).
int search_one(const Index& ix, const float* query /*dim*/, size_t k, std::vector<Result>& out,
               Counters* ctr) {
    const size_t D = ix.D, bw = ix.bw;
    const Layout& L = ix.L;
    std::vector<float> q(D, 0.0f);
    std::memcpy(q.data(), query, ix.dim * sizeof(float));
    QueryCode qc;
    encode_query(*ix.rot, q.data(), qc, nullptr);
    QParams qp{qc.A, qc.B, qc.C, ix.affine_a, ix.affine_b, ix.ip_qo_floor, ix.slack_levels[0]};
    if (k < 1) k = 1;
    float gamma = ix.search_gamma;

    uint32_t ep = ix.entry;
    if (ix.max_level > 0)
        for (int level = ix.max_level; level >= 1; --level) ep = greedy_layer(ix, q.data(), ep, level);
    if (ep == kInvalid || ep >= ix.n) return -1;

    std::vector<uint8_t> estimated(ix.n, 0), visited(ix.n, 0);
    std::vector<BeamEntry> beam;
    std::vector<Result> nn;
    nn.reserve(k + 1);
    auto nn_worst = [&]() { return nn.empty() ? std::numeric_limits<float>::max() : nn[0].dist; };
    auto nn_push = [&](Result r) {
        if (nn.size() < k) { nn.push_back(r); heap_push(nn.data(), nn.size(), result_before); }
        else if (r.dist < nn[0].dist) {
            heap_pop(nn.data(), nn.size(), result_before);
            nn.back() = r;
            heap_push(nn.data(), nn.size(), result_before);
        }
    };
    auto beam_push = [&](BeamEntry e) {
        beam.push_back(e); heap_push(beam.data(), beam.size(), beam_before);
        if (ctr) { ctr->beam_push++; if (beam.size() > ctr->beam_max) ctr->beam_max = beam.size(); }
    };

    float gamma_q = gamma;
    double ratio_sum = 0.0, ratio_sq_sum = 0.0;
    uint64_t ratio_count = 0;
    float qnorm = dot8(D, q.data(), q.data());
    auto exact_l2 = [&](uint32_t id) {
        if (ctr) ctr->exact_l2++;
        // (qnorm + norm) - 2*dot: 2*dot is exact, so the fused form GCC emits is identical
        float v = (qnorm + ix.norm_sq[id]) - 2.0f * dot8(D, q.data(), ix.vec(id));
        return v > 0.0f ? v : 0.0f;  // std::max(v, 0.0f)
    };

    beam_push({exact_l2(ep), 0.0f, ep});
    estimated[ep] = 1;
    uint32_t fs[32], msb[32];
    float est[32], lower[32];
    int slack_batch = 0;

    while (!beam.empty()) {
        BeamEntry cur{};
        bool found = false;
        while (!beam.empty()) {
            cur = beam[0];
            heap_pop(beam.data(), beam.size(), beam_before);
            beam.pop_back();
            if (visited[cur.id]) continue;
            found = true;
            break;
        }
        if (!found) break;
        if (nn.size() >= k && cur.est >= gamma_q * nn_worst()) { if (ctr) ctr->gamma_breaks++; break; }
        if (nn.size() >= k && cur.lower > nn_worst()) { if (ctr) ctr->lb_pruned_pops++; continue; }
        visited[cur.id] = 1;
        float exact_dist = exact_l2(cur.id);
        nn_push({cur.id, exact_dist});
        if (ctr) ctr->expansions++;

        const uint8_t* nb = ix.nb(cur.id);
        uint32_t count = rd<uint32_t>(nb + L.count);
        if (count == 0) continue;
        float dqp = exact_dist;
        if (ix.num_slack_levels > 0) {
            int li = std::min(slack_batch, ix.num_slack_levels - 1);
            qp.slack = ix.slack_levels[li];
            ++slack_batch;
        }
        const float* nop = (const float*)(nb + L.nop);
        const float* ipqo = (const float*)(nb + L.ip_qo);
        const float* ipcp = (const float*)(nb + L.ip_cp);
        const uint16_t* pop = (const uint16_t*)(nb + L.pop);
        const uint32_t* ids = (const uint32_t*)(nb + L.ids);
        int bc = (int)std::min<uint32_t>(32, count);
        if (bw == 1) {
            plane_sums(D, qc.lut.data(), nb + L.codes, fs);
            convert_1bit(qp, fs, nop, ipqo, ipcp, pop, bc, dqp, est, lower);
        } else {
            const uint16_t* wpop = (const uint16_t*)(nb + L.wpop);
            msb_sums(D, bw, qc.lut.data(), nb + L.codes, msb);
            convert_msb(bw, qp, msb, nop, ipqo, ipcp, pop, bc, dqp, lower);
            float thr = nn_worst();
            bool any = nn.size() < k;
            if (!any) for (int j = 0; j < bc; ++j) if (lower[j] < thr) { any = true; break; }
            if (any) {
                nbit_sums(D, bw, qc.lut.data(), nb + L.codes, fs, msb);
                convert_nbit(bw, qp, fs, msb, nop, ipqo, ipcp, pop, wpop, bc, dqp, est, lower);
            } else {
                if (ctr) ctr->stage2_skipped++;
                for (int j = 0; j < bc; ++j) est[j] = std::numeric_limits<float>::max();
            }
        }
        bool warmup = nn.size() < k;
        if (g_probe_trace) g_probe_trace->push_back(cur.id | 0x80000000u);
        for (uint32_t i = 0; i < count; ++i) {
            uint32_t nid = ids[i];
            if (g_probe_trace)   // bit 30: the neighbour's bounds would let it act at its turn if it were new (filter of :246)
                g_probe_trace->push_back(nid | ((warmup || !(nn.size() >= k && lower[i] >= nn_worst())) ? 0x40000000u : 0u));
            if (ctr) ctr->nbr_seen++;
            if (estimated[nid]) continue;
            estimated[nid] = 1;
            if (ctr) ctr->nbr_new++;
            float dabs = (nn.size() >= k) ? gamma_q * nn_worst() : std::numeric_limits<float>::max();
            if (warmup) {
                float ex = exact_l2(nid);
                nn_push({nid, ex});
                if (ex < dabs) beam_push({ex, ex, nid});
                continue;
            }
            float e = est[i], lo = lower[i];
            if (nn.size() >= k && lo >= nn_worst()) continue;
            if (e < nn_worst()) {
                float ex = exact_l2(nid);
                nn_push({nid, ex});
                if (ex < dabs) beam_push({ex, lo, nid});
                if (ex > kEpsSmall) {
                    // :255-267; compiled as: sum += r; sq = fma(r,r,sq);
                    // var = fnma(mean,mean,sq/n); g = gamma*float(fma(beta,std,1.0))
                    double r = (double)(e / ex);
                    ratio_sum += r;
                    ratio_sq_sum = std::fma(r, r, ratio_sq_sum);
                    ++ratio_count;
                    if (ratio_count >= ix.gamma_warmup) {
                        double nn_ = (double)ratio_count;
                        double mean = ratio_sum / nn_;
                        double var = std::fma(-mean, mean, ratio_sq_sum / nn_);
                        double sd = std::sqrt(var < 0.0 ? 0.0 : var);
                        float g = gamma * (float)std::fma((double)ix.gamma_beta, sd, 1.0);
                        gamma_q = (g < gamma) ? gamma : ((ix.gamma_max < g) ? ix.gamma_max : g);
                    }
                }
            } else if (e < dabs) {
                beam_push({e, lo, nid});
            }
        }
    }
    heap_sort(nn.data(), nn.size(), result_before);
    out = nn;
    return 0;
}

}  // namespace

extern "C" {

int orc_layout(int D, int bits, long* out) {
    Layout L = make_layout(D, bits);
    out[0] = L.vertex_bytes; out[1] = L.nb_off; out[2] = L.codes; out[3] = L.nop;
    out[4] = L.ip_qo; out[5] = L.ip_cp; out[6] = L.pop; out[7] = (bits > 1) ? (long)L.wpop : -1;
    out[8] = L.ids; out[9] = L.count;
    return 0;
}

int orc_rotation_signs(int D, float* out) {
    Rotation r(D);
    std::memcpy(out, r.signs.data(), 3 * D * sizeof(float));
    return 0;
}

int orc_encode_query(int dim, int D, const float* q, uint8_t* lut, float* coeffs, float* rotated) {
    Rotation rot(D);
    std::vector<float> p(D, 0.0f);
    std::memcpy(p.data(), q, dim * sizeof(float));
    QueryCode qc;
    encode_query(rot, p.data(), qc, rotated);
    std::memcpy(lut, qc.lut.data(), qc.lut.size());
    coeffs[0] = qc.A; coeffs[1] = qc.B; coeffs[2] = qc.C;
    return 0;
}

int orc_fastscan_plane(int D, const uint8_t* lut, const uint8_t* block, uint32_t* out) {
    plane_sums(D, lut, block, out); return 0;
}
int orc_fastscan_msb(int D, int bits, const uint8_t* lut, const uint8_t* planes, uint32_t* out) {
    msb_sums(D, bits, lut, planes, out); return 0;
}
int orc_fastscan_nbit(int D, int bits, const uint8_t* lut, const uint8_t* planes, uint32_t* out_nbit,
                      uint32_t* out_msb) {
    nbit_sums(D, bits, lut, planes, out_nbit, out_msb); return 0;
}
static QParams mkq(const float* p) { return QParams{p[0], p[1], p[2], p[3], p[4], p[5], p[6]}; }
int orc_convert_1bit(int, const float* qp, const uint32_t* sums, const float* nop, const float* ip_qo,
                     const float* ip_cp, const uint16_t* pop, int count, float dqp, float* est,
                     float* lower) {
    convert_1bit(mkq(qp), sums, nop, ip_qo, ip_cp, pop, count, dqp, est, lower); return 0;
}
int orc_convert_msb(int, int bits, const float* qp, const uint32_t* msb, const float* nop,
                    const float* ip_qo, const float* ip_cp, const uint16_t* pop, int count, float dqp,
                    float* lower) {
    convert_msb(bits, mkq(qp), msb, nop, ip_qo, ip_cp, pop, count, dqp, lower); return 0;
}
int orc_convert_nbit(int, int bits, const float* qp, const uint32_t* nbit, const uint32_t* msb,
                     const float* nop, const float* ip_qo, const float* ip_cp, const uint16_t* pop,
                     const uint16_t* wpop, int count, float dqp, float* est, float* lower) {
    convert_nbit(bits, mkq(qp), nbit, msb, nop, ip_qo, ip_cp, pop, wpop, count, dqp, est, lower);
    return 0;
}
int orc_dot(int D, const float* a, const float* b, float* out) { *out = dot8(D, a, b); return 0; }
int orc_l2(int D, const float* a, const float* b, float* out) { *out = l2sq8(D, a, b); return 0; }

// ---------------------------------------------------------------------------------
// Data-side encoder of one vertex' edges: encoder/rabitq_encoder.hpp:138-181 (1 bit),
// :287-323 + caq_quantize :371-467 (N bits).  Scalar reference code compiled with GCC's
// default -ffp-contract=fast: every a*b+c below is the fused form the compiler emits
// (pinned bit-for-bit against ref_encode_edges by tests/test_oracle_golden.py).
// values u8[cnt][D], aux f32[cnt][3] = {nop, ip_qo, ip_cp}, pops u32[cnt][2] = {msb, weighted}.
// ---------------------------------------------------------------------------------
// The beam as the reference keeps it -- std::priority_queue<BeamEntry, vector, greater> is std::push_heap / std::pop_heap
// on a vector (search/rabitq_search.hpp:53-58, :79-80) -- driven by an explicit operation list (1 = push the next
// (key, id), 0 = pop), with libstdc++'s own algorithms: the reference for the GPU heap routines' self-test
// (cph_debug_heap_ops).  Returns the heap array after the last operation.
extern "C" int orc_std_heap_ops(const uint8_t* ops, uint64_t n_ops, const float* keys, const uint32_t* ids,
                                float* out_keys, uint32_t* out_ids, uint32_t* out_size) {
    std::vector<BeamEntry> h;
    size_t next = 0;
    auto comp = [](const BeamEntry& a, const BeamEntry& b) { return a.est > b.est; };
    for (uint64_t j = 0; j < n_ops; ++j) {
        if (ops[j] == 2 && !h.empty()) {        // 2 = pop, then push (one expansion's heap traffic)
            std::pop_heap(h.begin(), h.end(), comp);
            h.pop_back();
        }
        if (ops[j]) {
            h.push_back(BeamEntry{keys[next], 0.0f, ids[next]});
            ++next;
            std::push_heap(h.begin(), h.end(), comp);
        } else if (!h.empty()) {
            std::pop_heap(h.begin(), h.end(), comp);
            h.pop_back();
        }
    }
    for (size_t i = 0; i < h.size(); ++i) { out_keys[i] = h[i].est; out_ids[i] = h[i].id; }
    *out_size = (uint32_t)h.size();
    return 0;
}

int orc_encode_edges(int dim, int D, int bits, const float* parent, const float* nbrs, int cnt,
                     uint8_t* values, float* aux, uint32_t* pops) {
    Rotation rot(D);
    const float d = static_cast<float>(D);
    const float norm_factor = 1.0f / (d * std::sqrt(d));
    const float inv_sqrt_d = 1.0f / std::sqrt(d);
    std::vector<float> rp(D, 0.0f), diff(D), x(D);
    std::memcpy(rp.data(), parent, dim * sizeof(float));
    rot.apply(rp.data());
    for (int i = 0; i < D; ++i) rp[i] *= norm_factor;
    const int Ki = (1 << bits) - 1;
    const float K = static_cast<float>(Ki);
    std::vector<int> u(D);
    for (int e = 0; e < cnt; ++e) {
        const float* nb = nbrs + (size_t)e * dim;
        uint8_t* val = values + (size_t)e * D;
        std::memset(val, 0, D);
        aux[3 * e] = aux[3 * e + 1] = aux[3 * e + 2] = 0.0f;
        pops[2 * e] = pops[2 * e + 1] = 0;
        float nsq = 0.0f;
        for (int i = 0; i < dim; ++i) { diff[i] = nb[i] - parent[i]; 
            // GCC vectorises this in-order reduction (8-wide, then a 4-wide epilogue: products rounded, added
            // in order); only the scalar remainder of dim % 4 elements gets the fused form
            if (i < dim - dim % 4) nsq += diff[i] * diff[i]; else nsq = std::fmaf(diff[i], diff[i], nsq);
        }
        for (int i = dim; i < D; ++i) diff[i] = 0.0f;
        const float nop = std::sqrt(nsq);
        aux[3 * e] = nop;
        if (nop < 1e-8f / d) continue;
        const float inv = 1.0f / nop;
        for (int i = 0; i < D; ++i) x[i] = diff[i] * inv;
        rot.apply(x.data());
        for (int i = 0; i < D; ++i) x[i] *= norm_factor;
        if (bits == 1) {
            float l1 = 0.0f, ipcp = 0.0f;
            uint32_t pc = 0;
            for (int i = 0; i < D; ++i) { val[i] = x[i] >= 0.0f ? 1 : 0; l1 += std::fabs(x[i]); pc += val[i]; }
            for (int i = 0; i < D; ++i) ipcp += (val[i] ? 1.0f : -1.0f) * rp[i];
            aux[3 * e + 1] = l1 * inv_sqrt_d;
            aux[3 * e + 2] = ipcp * inv_sqrt_d;
            pops[2 * e] = pops[2 * e + 1] = pc;
            continue;
        }
        float mn = x[0], mx = x[0];
        for (int i = 1; i < D; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
        float delta = (mx - mn) / K;
        if (delta < 1e-10f / d) delta = 1e-10f / d;
        const float inv_delta = 1.0f / delta;
        float dot = 0.0f, nrm = 0.0f;
        for (int i = 0; i < D; ++i) {
            int q = static_cast<int>(std::fmaf(x[i] - mn, inv_delta, 0.5f));
            q = q < 0 ? 0 : (q > Ki ? Ki : q);
            u[i] = q;
            const float c = (2.0f * q - K) / K;
            dot += c * x[i];      // vectorised in-order reductions (the pass always runs whole vectors of
            nrm += c * c;         // the padded dimension): products rounded before the add
        }
        float prev = 0.0f;
        for (int iter = 0; iter < 10; ++iter) {
            bool changed = false;
            for (int i = 0; i < D; ++i) {
                const int ou = u[i];
                const float oc = (2.0f * ou - K) / K;
                const float dwo = std::fmaf(-oc, x[i], dot);
                const float nwo = std::fmaf(-oc, oc, nrm);
                int bu = ou;
                float bd = dot, bn = nrm;
                auto consider = [&](int t) {
                    const float c = (2.0f * t - K) / K;
                    const float nd = std::fmaf(c, x[i], dwo);
                    const float nn = std::fmaf(c, c, nwo);
                    if (nd * nd * bn > bd * bd * nn) { bu = t; bd = nd; bn = nn; }
                };
                if (bits >= 4) {
                    if (ou - 1 >= 0) consider(ou - 1);
                    if (ou + 1 <= Ki) consider(ou + 1);
                } else {
                    for (int t = 0; t <= Ki; ++t) if (t != ou) consider(t);
                }
                if (bu != ou) {
                    const float nc = (2.0f * bu - K) / K;
                    dot = std::fmaf(nc, x[i], dwo);
                    nrm = std::fmaf(nc, nc, nwo);
                    u[i] = bu;
                    changed = true;
                }
            }
            if (!changed) break;
            const float cs = nrm > 0.0f ? dot * dot / nrm : 0.0f;
            if (iter > 0 && (cs - prev) < 1e-4f) break;
            prev = cs;
        }
        float ipqo = 0.0f, ipcp = 0.0f;
        uint32_t msb = 0, wp = 0;
        for (int i = 0; i < D; ++i) {
            val[i] = static_cast<uint8_t>(u[i]);
            const float c = (2.0f * u[i] - K) / K;
            ipqo = std::fmaf(c, x[i], ipqo);
            ipcp = std::fmaf(c, rp[i], ipcp);
            wp += (uint32_t)u[i];
            msb += (uint32_t)((u[i] >> (bits - 1)) & 1);
        }
        aux[3 * e + 1] = ipqo * inv_sqrt_d;
        aux[3 * e + 2] = ipcp * inv_sqrt_d;
        pops[2 * e] = msb;
        pops[2 * e + 1] = wp;
    }
    return 0;
}

static thread_local std::string g_err;
const char* orc_last_error() { return g_err.c_str(); }

void* orc_load(const char* path) {
    std::string err;
    Index* ix = load_index(path, err);
    if (!ix) g_err = err;
    return ix;
}
void orc_free(void* h) { delete static_cast<Index*>(h); }

// out: n, dim, D, bits, max_level, entry, vertex_bytes, n_upper_layers
int orc_info(void* h, long* out) {
    Index* ix = static_cast<Index*>(h);
    out[0] = ix->n; out[1] = ix->dim; out[2] = ix->D; out[3] = ix->bw; out[4] = ix->max_level;
    out[5] = ix->entry; out[6] = ix->L.vertex_bytes; out[7] = ix->upper.size();
    return 0;
}

// Upper-layer descent only: returns the layer-0 entry for a query.
int orc_entry_point(void* h, const float* query, uint32_t* ep_out) {
    Index* ix = static_cast<Index*>(h);
    std::vector<float> q(ix->D, 0.0f);
    std::memcpy(q.data(), query, ix->dim * sizeof(float));
    uint32_t ep = ix->entry;
    if (ix->max_level > 0)
        for (int l = ix->max_level; l >= 1; --l) ep = greedy_layer(*ix, q.data(), ep, l);
    *ep_out = ep;
    return 0;
}

// ids/dists are [n][k], padded with -1 / FLT_MAX (src/bindings.cpp:202-210); counts[i] =
// number of real results of query i; counters (optional) = 9 u64 per query.
int orc_search_batch(void* h, const float* queries, long n, long k, int64_t* ids, float* dists,
                     int32_t* counts, uint64_t* counters, int nthreads) {
    Index* ix = static_cast<Index*>(h);
    int rc = 0;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (long i = 0; i < n; ++i) {
        std::vector<Result> res;
        Counters c{};
        int r = search_one(*ix, queries + i * ix->dim, (size_t)k, res, counters ? &c : nullptr);
        if (r != 0) {
#pragma omp atomic write
            rc = r;
        }
        long kk = k < 1 ? 1 : k;
        long j = 0;
        for (; j < kk && j < (long)res.size() && j < k; ++j) {
            ids[i * k + j] = res[j].id; dists[i * k + j] = res[j].dist;
        }
        if (counts) counts[i] = (int32_t)res.size();
        for (; j < k; ++j) { ids[i * k + j] = -1; dists[i * k + j] = std::numeric_limits<float>::max(); }
        if (counters) std::memcpy(counters + i * 9, &c, sizeof(c));
    }
    return rc;
}

// FastScan estimates of one vertex's neighbour block for an encoded query (both N-bit
// stages): the unit the GPU fastscan kernel is checked against on a loaded index.
int orc_fastscan_vertex(void* h, const uint8_t* lut, const float* qp7, uint32_t vertex, float dqp,
                        uint32_t* sums, uint32_t* msb_out, float* est, float* lower,
                        float* lower_stage1) {
    Index* ix = static_cast<Index*>(h);
    const Layout& L = ix->L;
    const uint8_t* nb = ix->nb(vertex);
    QParams qp = mkq(qp7);
    const float* nop = (const float*)(nb + L.nop);
    const float* ipqo = (const float*)(nb + L.ip_qo);
    const float* ipcp = (const float*)(nb + L.ip_cp);
    const uint16_t* pop = (const uint16_t*)(nb + L.pop);
    // the epilogues run over the list's own length, as the search calls them (rabitq_search.hpp:152-154): a list whose
    // length is not a multiple of 8 ends in their scalar tails; outputs behind the list are zero
    const int bc = (int)std::min<uint32_t>(32, rd<uint32_t>(nb + L.count));
    std::memset(est, 0, 128); std::memset(lower, 0, 128); std::memset(lower_stage1, 0, 128);
    if (ix->bw == 1) {
        plane_sums(ix->D, lut, nb + L.codes, sums);
        std::memcpy(msb_out, sums, 128);
        convert_1bit(qp, sums, nop, ipqo, ipcp, pop, bc, dqp, est, lower);
        std::memcpy(lower_stage1, lower, 128);
    } else {
        const uint16_t* wpop = (const uint16_t*)(nb + L.wpop);
        uint32_t m2[32];
        msb_sums(ix->D, ix->bw, lut, nb + L.codes, m2);
        convert_msb(ix->bw, qp, m2, nop, ipqo, ipcp, pop, bc, dqp, lower_stage1);
        nbit_sums(ix->D, ix->bw, lut, nb + L.codes, sums, msb_out);
        convert_nbit(ix->bw, qp, sums, msb_out, nop, ipqo, ipcp, pop, wpop, bc, dqp, est, lower);
    }
    return 0;
}

// diagnostic: the probe sequence of one query (see g_probe_trace); returns the number of words, writes at most cap
long orc_probe_trace(void* h, const float* query, long k, uint32_t* out, long cap) {
    Index* ix = static_cast<Index*>(h);
    std::vector<uint32_t> tr;
    std::vector<Result> res;
    g_probe_trace = &tr;
    search_one(*ix, query, (size_t)k, res, nullptr);
    g_probe_trace = nullptr;
    const long n = (long)tr.size();
    std::memcpy(out, tr.data(), (size_t)std::min(n, cap) * 4);
    return n;
}

int orc_neighbor_count(void* h, uint32_t vertex) {
    Index* ix = static_cast<Index*>(h);
    return (int)std::min<uint32_t>(32, rd<uint32_t>(ix->nb(vertex) + ix->L.count));
}

int orc_exact_l2(void* h, const float* query, const uint32_t* ids, long n, float* out) {
    Index* ix = static_cast<Index*>(h);
    std::vector<float> q(ix->D, 0.0f);
    std::memcpy(q.data(), query, ix->dim * sizeof(float));
    float qn = dot8(ix->D, q.data(), q.data());
    for (long i = 0; i < n; ++i) {
        float v = (qn + ix->norm_sq[ids[i]]) - 2.0f * dot8(ix->D, q.data(), ix->vec(ids[i]));
        out[i] = v > 0.0f ? v : 0.0f;
    }
    return 0;
}

// graph/neighbor_selection.hpp:21-88 (select_neighbors_alpha_cng) for one vertex: candidates = ids into x[n][D]
// (kInvalid entries and the vertex itself are dropped, an id counts once); distances and the occlusion tests use
// the 8-chain squared L2 of core/memory.hpp:65-79.  The reference sorts the deduplicated candidates with
// std::sort on the distance alone, which leaves the order of equal distances unspecified; here ties are ordered
// by id (the GPU kernel's order: a deterministic refinement).  err may be null (all margins zero).
int orc_select_neighbors(int D, const float* x, uint32_t vtx, const uint32_t* cand, int n_cand, int R, float alpha,
                         float tau, float alpha_max, const float* err, uint32_t* out, uint32_t* out_cnt) {
    struct C { uint32_t id; float d; };
    std::vector<C> c;
    for (int i = 0; i < n_cand; ++i) {
        if (cand[i] == kInvalid || cand[i] == vtx) continue;
        bool seen = false;
        for (const C& e : c) seen |= e.id == cand[i];
        if (!seen) c.push_back({cand[i], l2sq8((size_t)D, x + (size_t)vtx * D, x + (size_t)cand[i] * D)});
    }
    std::sort(c.begin(), c.end(), [](const C& a, const C& b) { return a.d < b.d || (a.d == b.d && a.id < b.id); });
    std::vector<uint32_t> sel;
    if ((int)c.size() <= R) {
        for (const C& e : c) sel.push_back(e.id);
    } else {
        if (alpha_max <= 0.0f) alpha_max = 2.0f * alpha;
        float la = alpha * std::sqrt((float)c.size() / (float)R);
        la = la < 1.0f ? 1.0f : (la > alpha_max ? alpha_max : la);
        std::vector<char> taken(c.size(), 0);
        for (size_t i = 0; i < c.size() && (int)sel.size() < R; ++i) {
            const float errc = err ? err[c[i].id] : 0.0f;
            bool add = true;
            for (uint32_t e : sel) {
                const float dce = l2sq8((size_t)D, x + (size_t)c[i].id * D, x + (size_t)e * D);
                const float thr = la * c[i].d + (errc + (err ? err[e] : 0.0f)) - (la - 1.0f) * tau;
                if (dce < thr) { add = false; break; }
            }
            if (add) { sel.push_back(c[i].id); taken[i] = 1; }
        }
        for (size_t i = 0; i < c.size() && (int)sel.size() < R; ++i)
            if (!taken[i]) sel.push_back(c[i].id);
    }
    *out_cnt = (uint32_t)sel.size();
    for (int i = 0; i < 32; ++i) out[i] = i < (int)sel.size() ? sel[i] : kInvalid;
    return 0;
}

// One calibration sample as the builder evaluates it (api/hnsw_index.hpp:770-1040 gathers the same quantities on
// the host): a greedy hop from `start` to its nearest neighbour if that is nearer to the query, then per edge of
// the vertex arrived at: {nop, ip_est_raw - ip_cp, max(|ip_qo|, 1e-10), <q - p, o - p> / nop, |q - o|^2, ip_qo},
// with the raw estimator  ip_est_raw = A/K * S_nbit + B/K * weighted_popcount + C  (K = 2^BW - 1; 1-bit: plain
// sums and popcounts) of distance/fastscan_kernel.hpp:281.  rec = [32][6]; *dqp = exact |q - p|^2.
int orc_calib_record(void* h, const float* query, uint32_t start, float* rec, uint32_t* cnt_out, float* dqp_out) {
    Index* ix = static_cast<Index*>(h);
    const size_t D = ix->D;
    const Layout& L = ix->L;
    std::vector<float> q(D, 0.0f);
    std::memcpy(q.data(), query, ix->dim * sizeof(float));
    QueryCode qc;
    encode_query(*ix->rot, q.data(), qc, nullptr);
    uint32_t parent = start;
    float best = l2sq8(D, q.data(), ix->vec(parent));
    {
        const uint8_t* nb = ix->nb(parent);
        const uint32_t* ids = (const uint32_t*)(nb + L.ids);
        const uint32_t cnt = rd<uint32_t>(nb + L.count);
        float cand = std::numeric_limits<float>::max();
        uint32_t cand_id = kInvalid;
        for (uint32_t i = 0; i < cnt && i < 32; ++i) {
            const float dd = l2sq8(D, q.data(), ix->vec(ids[i]));
            if (dd < cand) { cand = dd; cand_id = ids[i]; }
        }
        if (cand_id != kInvalid && cand < best) { best = cand; parent = cand_id; }
    }
    const uint8_t* nb = ix->nb(parent);
    const float* nop = (const float*)(nb + L.nop);
    const float* ipqo = (const float*)(nb + L.ip_qo);
    const float* ipcp = (const float*)(nb + L.ip_cp);
    const uint16_t* pop = (const uint16_t*)(nb + L.pop);
    const uint32_t* ids = (const uint32_t*)(nb + L.ids);
    const uint32_t cnt = std::min<uint32_t>(32, rd<uint32_t>(nb + L.count));
    uint32_t sums[32], msb[32];
    if (ix->bw == 1) plane_sums(D, qc.lut.data(), nb + L.codes, sums);
    else nbit_sums(D, ix->bw, qc.lut.data(), nb + L.codes, sums, msb);
    const float* p = ix->vec(parent);
    for (uint32_t i = 0; i < 32; ++i) {
        float ipa;
        if (ix->bw == 1) {
            ipa = qc.A * (float)sums[i] + qc.B * (float)pop[i] + qc.C;
        } else {
            const uint16_t* wpop = (const uint16_t*)(nb + L.wpop);
            const float invK = 1.0f / (float)((1u << ix->bw) - 1);
            ipa = qc.A * invK * (float)sums[i] + qc.B * invK * (float)wpop[i] + qc.C;
        }
        const float ipq = std::fabs(ipqo[i]) > 1e-10f ? std::fabs(ipqo[i]) : 1e-10f;
        const float nopf = nop[i] > 1e-12f ? nop[i] : 1e-12f;
        float tip = 0.0f, dqo = 0.0f;
        if (i < cnt) {
            const float* o = ix->vec(ids[i]);
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t d = 0; d < D; d += 8)
                for (int j = 0; j < 8; ++j) {
                    t[j] = std::fmaf(q[d + j] - p[d + j], o[d + j] - p[d + j], t[j]);
                    l[j] = std::fmaf(q[d + j] - o[d + j], q[d + j] - o[d + j], l[j]);
                }
            tip = reduce8(t);
            dqo = reduce8(l);
        }
        float* r = rec + i * 6;
        r[0] = nopf; r[1] = ipa - ipcp[i]; r[2] = ipq; r[3] = i < cnt ? tip / nopf : 0.0f; r[4] = dqo; r[5] = ipqo[i];
    }
    *cnt_out = cnt;
    *dqp_out = best;
    return 0;
}

// Streaming FastScan over contiguous reference-layout neighbour blocks (scalar port;
// the cpu_baseline "port" leg of bench.py when oracle/_ref is absent).
int orc_fastscan_stream(int D, int bits, const uint8_t* lut, const float* qp7, const uint8_t* blocks,
                        long n_blocks, float dqp, int reps, double* checksum, int nthreads) {
    Layout L = make_layout(D, bits);
    size_t nb_bytes = L.vertex_bytes - L.nb_off;
    QParams qp = mkq(qp7);
    double total = 0.0;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    for (int r = 0; r < reps; ++r) {
#pragma omp parallel for reduction(+ : total) schedule(static) num_threads(nthreads)
        for (long i = 0; i < n_blocks; ++i) {
            const uint8_t* nb = blocks + (size_t)i * nb_bytes;
            uint32_t sums[32], msb[32];
            float est[32], lower[32];
            const float* nop = (const float*)(nb + L.nop);
            const float* ipqo = (const float*)(nb + L.ip_qo);
            const float* ipcp = (const float*)(nb + L.ip_cp);
            const uint16_t* pop = (const uint16_t*)(nb + L.pop);
            if (bits == 1) {
                plane_sums(D, lut, nb + L.codes, sums);
                convert_1bit(qp, sums, nop, ipqo, ipcp, pop, 32, dqp, est, lower);
            } else {
                const uint16_t* wpop = (const uint16_t*)(nb + L.wpop);
                msb_sums(D, bits, lut, nb + L.codes, msb);
                convert_msb(bits, qp, msb, nop, ipqo, ipcp, pop, 32, dqp, lower);
                nbit_sums(D, bits, lut, nb + L.codes, sums, msb);
                convert_nbit(bits, qp, sums, msb, nop, ipqo, ipcp, pop, wpop, 32, dqp, est, lower);
            }
            total += est[i & 31] + lower[(i >> 5) & 31];
        }
    }
    *checksum = total;
    return 0;
}

}  // extern "C"
