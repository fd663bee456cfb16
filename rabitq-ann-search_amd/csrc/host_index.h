// host_index.h — host side of the hot path: v2 index file reader/writer, the repacker that
// turns reference-layout neighbour blocks into the device layout, and the host mirror of the
// query encoder (used to make the synthetic query of the streaming benchmark; queries of a
// search are encoded on the device, device_encode.h).
//
// Reference counterparts (relative to /root/reference/include/cphnsw/):
//   api/hnsw_index.hpp:217-303 save, :305-443 load            -> HostIndex::save / load
//   graph/rabitq_graph.hpp:19-29, distance/fastscan_layout.hpp -> repack_* (layout only)
//   encoder/rotation.hpp:15-67, encoder/transform/fht.hpp:23-57 -> Rotation
//   encoder/rabitq_encoder.hpp:73-79,98-136,197-209            -> encode_query
// This file is compiled with -ffp-contract=off; each fused multiply-add is explicit and
// sits where the compiled reference has one.
#pragma once
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "cph_core.h"

namespace cph {

struct UpperEdge {
    uint32_t node;
    std::vector<uint32_t> nbrs;
};

struct Rotation {
    size_t D = 0;
    std::vector<float> signs;  // [3][D]
    void init(size_t D_, uint64_t seed) {
        D = D_;
        signs.resize(3 * D);
        // one mt19937_64 stream, uniform_int_distribution<int>(0,1) drawn layer-major
        // (rotation.hpp:23-31); libstdc++ resolves that range to the top bit of each draw.
        std::mt19937_64 rng(seed);
        for (size_t i = 0; i < 3 * D; ++i) signs[i] = (rng() >> 63) ? 1.0f : -1.0f;
    }
    static void fht(float* v, size_t len) {
        // butterflies of fht.hpp:26-56: the in-register stages (h = 1,2,4) keep
        // (upper - lower) in the upper slot, the strided stages (h >= 8) (lower - upper)
        for (size_t h = 1; h < len; h *= 2)
            for (size_t i = 0; i < len; i += 2 * h)
                for (size_t j = i; j < i + h; ++j) {
                    float x = v[j], y = v[j + h];
                    v[j] = x + y;
                    v[j + h] = (h < 8) ? (y - x) : (x - y);
                }
    }
    void apply(float* x) const {
        for (int l = 0; l < 3; ++l) {
            const float* s = &signs[l * D];
            for (size_t i = 0; i < D; ++i) x[i] = x[i] * s[i];
            fht(x, D);
        }
    }
};

struct EncodedQuery {
    float A, B, C;
    std::vector<uint8_t> qu;  // 4-bit scalars per dimension [D]
};

// buf: padded raw query (D floats), overwritten with the rotated, scaled vector.
inline void encode_query(const Rotation& rot, float* buf, EncodedQuery& out) {
    const size_t D = rot.D;
    rot.apply(buf);
    const float d = static_cast<float>(D);
    const float norm_factor = 1.0f / (d * std::sqrt(d));
    const float inv_sqrt_d = 1.0f / std::sqrt(d);
    for (size_t i = 0; i < D; ++i) buf[i] = buf[i] * norm_factor;
    float vl = buf[0], vmax = buf[0];
    for (size_t i = 1; i < D; ++i) {
        if (buf[i] < vl) vl = buf[i];
        if (buf[i] > vmax) vmax = buf[i];
    }
    float delta = (vmax - vl) / 15.0f;
    if (delta < kEpsTiny) delta = kEpsTiny;
    const float inv_delta = 1.0f / delta;
    out.qu.resize(D);
    float sum_qu = 0.0f;
    for (size_t i = 0; i < D; ++i) {
        int u = static_cast<int>(std::fmaf(buf[i] - vl, inv_delta, 0.5f));
        u = u < 0 ? 0 : (u > 15 ? 15 : u);
        out.qu[i] = static_cast<uint8_t>(u);
        sum_qu += static_cast<float>(u);
    }
    out.A = (2.0f * delta) * inv_sqrt_d;
    out.B = (2.0f * vl) * inv_sqrt_d;
    out.C = -std::fmaf(d, vl, delta * sum_qu) * inv_sqrt_d;
}

// Bit-slice the 4-bit scalars into the device query masks: word w -> {Q0,Q1,Q2,Q3}, bit t of
// Q_j = bit j of qu[32w + t].
inline void qu_to_masks(const uint8_t* qu, size_t D, uint32_t* masks /*[PW][4]*/) {
    const size_t PW = D >= 32 ? D / 32 : 1;
    std::memset(masks, 0, PW * 16);
    for (size_t d = 0; d < D; ++d)
        for (int j = 0; j < 4; ++j)
            if ((qu[d] >> j) & 1) masks[(d / 32) * 4 + j] |= 1u << (d % 32);
}

// Reference LUT format (u8[D/4][16]) <-> 4-bit scalars.
inline void qu_to_lut(const uint8_t* qu, size_t D, uint8_t* lut) {
    for (size_t s = 0; s < D / 4; ++s)
        for (unsigned p = 0; p < 16; ++p) {
            uint8_t v = 0;
            for (unsigned b = 0; b < 4; ++b)
                if (p & (1u << b)) v = static_cast<uint8_t>(v + qu[4 * s + b]);
            lut[s * 16 + p] = v;
        }
}
inline void lut_to_qu(const uint8_t* lut, size_t D, uint8_t* qu) {
    for (size_t d = 0; d < D; ++d) qu[d] = lut[(d / 4) * 16 + (1u << (d % 4))];
}

// A file that appears under its final name only when it is complete: written as <path>.tmp.<pid>, flushed, synced to
// the device (fsync: a rename can reach the disk before the data it names, and a crash in between would leave a
// truncated file under the final name) and closed with the results checked, then renamed over the target and the
// directory entry synced.  A handle that serves an index out of a mapping of `path` keeps its (old) inode; a failed or
// interrupted save leaves the previous file untouched.
struct AtomicFile {
    std::string path, tmp;
    FILE* f = nullptr;
    explicit AtomicFile(const std::string& p) : path(p), tmp(p + ".tmp." + std::to_string((long)getpid())) {
        f = std::fopen(tmp.c_str(), "wb");
        if (!f) throw std::runtime_error("Cannot open file for writing: " + path);
    }
    AtomicFile(const AtomicFile&) = delete;
    AtomicFile& operator=(const AtomicFile&) = delete;
    void write(const void* p, size_t b) {
        if (b && std::fwrite(p, 1, b, f) != b) throw std::runtime_error("Write error: " + path);
    }
    void commit() {
        const bool ok = std::fflush(f) == 0 && ::fsync(fileno(f)) == 0;
        const bool closed = std::fclose(f) == 0;
        f = nullptr;
        if (!ok || !closed) throw std::runtime_error("Write error: " + path);
        if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("Cannot rename the finished file into place: " + path);
        tmp.clear();
        // the new directory entry (best effort: some file systems refuse to open a directory for this)
        const size_t slash = path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : path.substr(0, slash));
        const int dfd = ::open(dir.c_str(), O_RDONLY | O_DIRECTORY);
        if (dfd >= 0) { (void)::fsync(dfd); ::close(dfd); }
    }
    ~AtomicFile() {
        if (f) std::fclose(f);
        if (!tmp.empty()) std::remove(tmp.c_str());
    }
};

struct HostIndex {
    size_t D = 0, bw = 0, dim = 0, n = 0;
    int32_t max_level = 0;
    uint32_t entry = kInvalidNode;
    float upper_tau = 0.0f, upper_alpha = 1.2f;
    double mL = 0.0;
    uint64_t seed = 42;
    uint8_t calib[248];
    uint8_t profile[72];
    std::vector<float> centroid;
    std::vector<int32_t> levels;
    std::vector<float> norm_sq;
    std::vector<float> raw;           // [n][D]  (empty when the vectors are served from a mapped native file)
    const float* raw_view = nullptr;  // [n][D] inside a mapped native file (csrc/native_file.h), else null
    std::vector<uint8_t> search_data; // n * vertex_bytes, reference layout (kept for save; a native load rebuilds it on demand)
    std::vector<std::vector<UpperEdge>> upper;
    RefLayout RL;
    Rotation rot;
    bool has_dup_neighbors = false;

    template <class T>
    static T rd(const uint8_t* p) {
        T v;
        std::memcpy(&v, p, sizeof(T));
        return v;
    }

    SearchConsts consts() const {
        SearchConsts s{};
        s.affine_a = rd<float>(calib + 0);
        s.affine_b = rd<float>(calib + 4);
        s.ip_qo_floor = rd<float>(calib + 8);
        s.gamma_max = rd<float>(calib + 84);
        s.gamma_beta = rd<float>(calib + 88);
        s.gamma_warmup = rd<uint64_t>(calib + 96);
        std::memcpy(s.slack, calib + 108, 128);
        s.num_slack = rd<int32_t>(calib + 236);
        if (s.num_slack > kMaxSlack) s.num_slack = kMaxSlack;
        s.gamma = rd<float>(calib + 240);
        return s;
    }

    // api/hnsw_index.hpp:305-443.  expect_* come from the handle (the reference's template
    // parameters / constructor argument) and produce the same error texts.
    void load(const std::string& path, size_t expect_D, size_t expect_bw, size_t expect_dim) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("Cannot open file for reading: " + path);
        struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
        auto rdn = [&](void* p, size_t b) {
            if (b && std::fread(p, 1, b, f) != b)
                throw std::runtime_error("Read error or truncated file: " + path);
        };
        uint8_t hdr[68];
        rdn(hdr, 12);
        if (rd<uint64_t>(hdr) != 0x57534E48504300ULL)
            throw std::runtime_error("Invalid magic bytes (not a CP-HNSW index file).");
        if (rd<uint32_t>(hdr + 8) != 2)
            throw std::runtime_error("Unsupported index file version: " +
                                     std::to_string(rd<uint32_t>(hdr + 8)));
        rdn(hdr + 12, 56);
        const uint32_t fD = rd<uint32_t>(hdr + 12), fR = rd<uint32_t>(hdr + 16),
                       fBW = rd<uint32_t>(hdr + 20), fdim = rd<uint32_t>(hdr + 24);
        if (fD != expect_D || fR != 32 || fBW != expect_bw)
            throw std::runtime_error(
                "Index file template parameters mismatch: file D=" + std::to_string(fD) +
                " R=" + std::to_string(fR) + " BW=" + std::to_string(fBW) + ", expected D=" +
                std::to_string(expect_D) + " R=32 BW=" + std::to_string(expect_bw));
        if (fdim != expect_dim)
            throw std::runtime_error("Index file dim=" + std::to_string(fdim) +
                                     " mismatches Index dim=" + std::to_string(expect_dim));
        if (rd<uint64_t>(hdr + 60) != 42) throw std::runtime_error("Index file rotation seed mismatch.");

        HostIndex t;
        t.D = fD; t.bw = fBW; t.dim = fdim;
        t.n = rd<uint64_t>(hdr + 28);
        t.max_level = rd<int32_t>(hdr + 36);
        t.entry = rd<uint32_t>(hdr + 40);
        t.upper_tau = rd<float>(hdr + 44);
        t.upper_alpha = rd<float>(hdr + 48);
        t.mL = rd<double>(hdr + 52);
        t.seed = rd<uint64_t>(hdr + 60);
        t.RL = make_ref_layout(t.D, t.bw);
        rdn(t.calib, 248);
        rdn(t.profile, 72);
        const size_t n = t.n;
        {   // the counts come from the file: check them against its size before anything is allocated from them
            const long here = std::ftell(f);
            std::fseek(f, 0, SEEK_END);
            const long total = std::ftell(f);
            std::fseek(f, here, SEEK_SET);
            const unsigned __int128 need = (unsigned __int128)n * (8 + t.D * 4 + t.RL.vertex_bytes) + t.dim * 4 + 4;
            if (here < 0 || total < here || need > (unsigned __int128)(total - here))
                throw std::runtime_error("Read error or truncated file: " + path);
        }
        t.centroid.resize(t.dim);  rdn(t.centroid.data(), t.dim * 4);
        t.levels.resize(n);        rdn(t.levels.data(), n * 4);
        t.norm_sq.resize(n);       rdn(t.norm_sq.data(), n * 4);
        t.raw.resize(n * t.D);     rdn(t.raw.data(), n * t.D * 4);
        t.search_data.resize(n * t.RL.vertex_bytes);
        rdn(t.search_data.data(), t.search_data.size());
        uint32_t nl = 0;
        rdn(&nl, 4);
        if (nl > 64) throw std::runtime_error("Corrupt index: too many upper layers");
        t.upper.resize(nl);
        for (uint32_t l = 0; l < nl; ++l) {
            uint32_t sz = 0;
            rdn(&sz, 4);
            if (sz > n) throw std::runtime_error("Corrupt index: upper layer larger than the index");
            t.upper[l].resize(sz);
            for (uint32_t e = 0; e < sz; ++e) {
                uint32_t cnt = 0;
                rdn(&t.upper[l][e].node, 4);
                rdn(&cnt, 4);
                if (cnt > n) throw std::runtime_error("Corrupt index: upper-layer degree out of range");
                t.upper[l][e].nbrs.resize(cnt);
                rdn(t.upper[l][e].nbrs.data(), (size_t)cnt * 4);
            }
        }
        t.validate();
        t.rot.init(t.D, t.seed);
        *this = std::move(t);
    }

    // The reference trusts the file; the GPU path must not chase an out-of-range id.
    void validate() {
        has_dup_neighbors = false;
        if (n != 0 && entry != kInvalidNode && entry >= n) throw std::runtime_error("Corrupt index: entry point out of range");
        for (size_t v = 0; v < n; ++v) {
            const uint8_t* nb = &search_data[v * RL.vertex_bytes + RL.nb_off];
            uint32_t cnt = rd<uint32_t>(nb + RL.count);
            if (cnt > 32) throw std::runtime_error("Corrupt index: neighbour count > 32");
            const uint32_t* ids = reinterpret_cast<const uint32_t*>(nb + RL.ids);
            for (uint32_t i = 0; i < cnt; ++i) {
                if (ids[i] >= n) throw std::runtime_error("Corrupt index: neighbour id out of range");
                for (uint32_t j = 0; j < i; ++j)
                    if (ids[j] == ids[i]) has_dup_neighbors = true;
            }
        }
        for (auto& layer : upper)
            for (auto& e : layer) {
                if (e.node >= n) throw std::runtime_error("Corrupt index: upper-layer node out of range");
                for (uint32_t x : e.nbrs)
                    if (x >= n) throw std::runtime_error("Corrupt index: upper-layer neighbour out of range");
            }
    }

    // api/hnsw_index.hpp:217-303
    void save(const std::string& path) const {
        AtomicFile out(path);
        auto wr = [&](const void* p, size_t b) { out.write(p, b); };
        const uint64_t magic = 0x57534E48504300ULL;
        const uint32_t version = 2, hD = (uint32_t)D, hR = 32, hBW = (uint32_t)bw, hdim = (uint32_t)dim;
        const uint64_t hn = n;
        wr(&magic, 8); wr(&version, 4); wr(&hD, 4); wr(&hR, 4); wr(&hBW, 4); wr(&hdim, 4);
        wr(&hn, 8); wr(&max_level, 4); wr(&entry, 4); wr(&upper_tau, 4); wr(&upper_alpha, 4);
        wr(&mL, 8); wr(&seed, 8);
        wr(calib, 248); wr(profile, 72);
        wr(centroid.data(), dim * 4);
        wr(levels.data(), n * 4);
        wr(norm_sq.data(), n * 4);
        wr(vec(0), n * D * 4);
        wr(search_data.data(), search_data.size());
        const uint32_t nl = (uint32_t)upper.size();
        wr(&nl, 4);
        for (const auto& layer : upper) {
            const uint32_t sz = (uint32_t)layer.size();
            wr(&sz, 4);
            for (const auto& e : layer) {
                const uint32_t cnt = (uint32_t)e.nbrs.size();
                wr(&e.node, 4); wr(&cnt, 4);
                wr(e.nbrs.data(), (size_t)cnt * 4);
            }
        }
        out.commit();
    }

    const uint8_t* nb(size_t v) const { return &search_data[v * RL.vertex_bytes + RL.nb_off]; }
    const float* vec(size_t v) const { return (raw_view ? raw_view : raw.data()) + v * D; }
};

// ---- reference neighbour block <-> device block -----------------------------------------
inline size_t dev_dword_offset(const DevLayout& L, uint32_t plane, uint32_t w, uint32_t i) {
    const uint32_t t = plane * L.PW + w;
    if (L.wide) {
        const uint32_t ck = t / 4, e = t % 4;
        const uint32_t h = (L.NH == 2) ? ck / L.CPL : 0;
        const uint32_t k = (L.NH == 2) ? ck % L.CPL : ck;
        return ((size_t)(k * L.NH * 32 + h * 32 + i) * 16 + e * 4);
    }
    return ((size_t)t * 32 + i) * 4;
}

inline void repack_ref_to_dev(const uint8_t* ref_nb, const RefLayout& RL, const DevLayout& L,
                              uint8_t* dev) {
    std::memset(dev, 0, L.stride);
    const size_t bytes_per_plane_nb = (L.D + 7) / 8;  // code bytes per neighbour per plane
    const size_t plane_stride = round_up(RL.plane_bytes, 64);
    for (uint32_t b = 0; b < L.BW; ++b) {
        const uint8_t* plane = ref_nb + RL.codes + b * plane_stride;
        for (uint32_t w = 0; w < L.PW; ++w)
            for (uint32_t i = 0; i < 32; ++i) {
                uint32_t v = 0;
                for (uint32_t s = 0; s < 4; ++s) {
                    size_t sp = (size_t)4 * w + s;  // byte [sp][i] = dims 8sp..8sp+7
                    if (sp < bytes_per_plane_nb) v |= (uint32_t)plane[sp * 32 + i] << (8 * s);
                }
                std::memcpy(dev + dev_dword_offset(L, b, w, i), &v, 4);
            }
    }
    const float* nop = reinterpret_cast<const float*>(ref_nb + RL.nop);
    const float* ipqo = reinterpret_cast<const float*>(ref_nb + RL.ip_qo);
    const float* ipcp = reinterpret_cast<const float*>(ref_nb + RL.ip_cp);
    const uint16_t* pop = reinterpret_cast<const uint16_t*>(ref_nb + RL.pop);
    const uint16_t* wpop = L.BW > 1 ? reinterpret_cast<const uint16_t*>(ref_nb + RL.wpop) : nullptr;
    for (uint32_t i = 0; i < 32; ++i) {
        uint32_t a[4];
        std::memcpy(&a[0], &nop[i], 4);
        std::memcpy(&a[1], &ipqo[i], 4);
        std::memcpy(&a[2], &ipcp[i], 4);
        a[3] = (uint32_t)pop[i] | ((uint32_t)(wpop ? wpop[i] : 0) << 16);
        std::memcpy(dev + L.aux_off + i * 16, a, 16);
    }
    // slots >= count may hold stale ids in the file (graph_refinement.hpp:46-47); the device
    // copy marks them invalid so that the kernels never need `count` on their critical path
    uint32_t cnt;
    std::memcpy(&cnt, ref_nb + RL.count, 4);
    if (cnt > 32) cnt = 32;
    uint32_t ids[32];
    std::memcpy(ids, ref_nb + RL.ids, 128);
    for (uint32_t i = cnt; i < 32; ++i) ids[i] = kInvalidNode;
    std::memcpy(dev + L.ids_off, ids, 128);
    std::memcpy(dev + L.count_off, &cnt, 4);
}

inline void repack_dev_to_ref(const uint8_t* dev, const DevLayout& L, const RefLayout& RL,
                              uint8_t* ref_nb) {
    std::memset(ref_nb, 0, RL.nb_bytes);
    const size_t bytes_per_plane_nb = (L.D + 7) / 8;
    const size_t plane_stride = round_up(RL.plane_bytes, 64);
    for (uint32_t b = 0; b < L.BW; ++b) {
        uint8_t* plane = ref_nb + RL.codes + b * plane_stride;
        for (uint32_t w = 0; w < L.PW; ++w)
            for (uint32_t i = 0; i < 32; ++i) {
                uint32_t v;
                std::memcpy(&v, dev + dev_dword_offset(L, b, w, i), 4);
                for (uint32_t s = 0; s < 4; ++s) {
                    size_t sp = (size_t)4 * w + s;
                    if (sp < bytes_per_plane_nb) plane[sp * 32 + i] = (uint8_t)(v >> (8 * s));
                }
            }
    }
    for (uint32_t i = 0; i < 32; ++i) {
        uint32_t a[4];
        std::memcpy(a, dev + L.aux_off + i * 16, 16);
        std::memcpy(ref_nb + RL.nop + 4 * i, &a[0], 4);
        std::memcpy(ref_nb + RL.ip_qo + 4 * i, &a[1], 4);
        std::memcpy(ref_nb + RL.ip_cp + 4 * i, &a[2], 4);
        uint16_t p = (uint16_t)(a[3] & 0xFFFF), wp = (uint16_t)(a[3] >> 16);
        std::memcpy(ref_nb + RL.pop + 2 * i, &p, 2);
        if (L.BW > 1) std::memcpy(ref_nb + RL.wpop + 2 * i, &wp, 2);
    }
    std::memcpy(ref_nb + RL.ids, dev + L.ids_off, 128);
    std::memcpy(ref_nb + RL.count, dev + L.count_off, 4);
}

}  // namespace cph
