// native_file.h — the GPU-native index file (second half).
//
// The reference's v2 file (api/hnsw_index.hpp:217-443) stores every vertex in the AVX2 FastScan layout, so
// loading it means reading 3.4 GB (1M x 128, 4-bit) and re-laying-out every neighbour block on the host
// (0.99 s).  The native file keeps what the GPU reads in the form the GPU reads it:
//
//   header | calibration, profile, centroid, levels, norms, upper layers      (the v2 file's small fields)
//   own    [n][own_stride]   the vertices' own codes + {nop, ip_qo} (only needed to write a v2 file again)
//   raw    [n][D] f32        4096-aligned
//   blocks [n][stride]       4096-aligned, device block layout (cph_core.h)
//
// load = mmap + two host-to-device copies straight out of the mapping; the vectors stay mapped for
// cph_get_vectors / save.  A v2 file can always be regenerated from a native one (cph_save after
// cph_load_native): the reference-layout blocks are re-derived from the device blocks.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <atomic>

#include "host_index.h"
#include "host_parallel.h"

namespace cph {

constexpr uint64_t kNativeMagic = 0x3535334948504300ULL;   // "\0CPHI355"
constexpr uint32_t kNativeVersion = 1;

struct NativeHeader {
    uint64_t magic;
    uint32_t version, D, bw, dim;
    uint64_t n;
    uint32_t stride, own_stride;
    int32_t max_level;
    uint32_t entry;
    float upper_tau, upper_alpha;
    double mL;
    uint64_t seed;
    uint32_t has_dup, n_layers;
    uint64_t small_bytes;     // bytes of the small section that follows the header
    uint64_t own_off, raw_off, blocks_off, file_bytes;
};

struct NativeMapping {
    void* base = nullptr;
    size_t bytes = 0;
    NativeMapping() = default;
    NativeMapping(const NativeMapping&) = delete;
    NativeMapping& operator=(const NativeMapping&) = delete;
    NativeMapping(NativeMapping&& o) noexcept : base(o.base), bytes(o.bytes) { o.base = nullptr; o.bytes = 0; }
    NativeMapping& operator=(NativeMapping&& o) noexcept {
        if (this != &o) { reset(); base = o.base; bytes = o.bytes; o.base = nullptr; o.bytes = 0; }
        return *this;
    }
    void reset() {
        if (base) munmap(base, bytes);
        base = nullptr;
        bytes = 0;
    }
    ~NativeMapping() { reset(); }
};

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// `own` = [n][own_stride] vertex headers, `blocks` = [n][stride] device blocks (both host pointers).
inline void write_native(const std::string& path, const HostIndex& hi, uint32_t stride, const uint8_t* own,
                         uint32_t own_stride, const uint8_t* blocks) {
    std::vector<uint8_t> small;
    auto put = [&](const void* p, size_t b) { const uint8_t* q = static_cast<const uint8_t*>(p); small.insert(small.end(), q, q + b); };
    put(hi.calib, 248);
    put(hi.profile, 72);
    put(hi.centroid.data(), hi.dim * 4);
    put(hi.levels.data(), hi.n * 4);
    put(hi.norm_sq.data(), hi.n * 4);
    for (const auto& layer : hi.upper) {
        const uint32_t sz = (uint32_t)layer.size();
        put(&sz, 4);
        for (const auto& e : layer) {
            const uint32_t cnt = (uint32_t)e.nbrs.size();
            put(&e.node, 4);
            put(&cnt, 4);
            put(e.nbrs.data(), (size_t)cnt * 4);
        }
    }
    NativeHeader h{};
    h.magic = kNativeMagic; h.version = kNativeVersion; h.D = (uint32_t)hi.D; h.bw = (uint32_t)hi.bw; h.dim = (uint32_t)hi.dim;
    h.n = hi.n; h.stride = stride; h.own_stride = own_stride; h.max_level = hi.max_level; h.entry = hi.entry;
    h.upper_tau = hi.upper_tau; h.upper_alpha = hi.upper_alpha; h.mL = hi.mL; h.seed = hi.seed;
    h.has_dup = hi.has_dup_neighbors ? 1u : 0u; h.n_layers = (uint32_t)hi.upper.size();
    h.small_bytes = small.size();
    h.own_off = align_up(sizeof(NativeHeader) + small.size(), 4096);
    h.raw_off = align_up(h.own_off + hi.n * own_stride, 4096);
    h.blocks_off = align_up(h.raw_off + hi.n * hi.D * 4, 4096);
    h.file_bytes = h.blocks_off + hi.n * (uint64_t)stride;
    AtomicFile out(path);       // saving over the file this handle is mapped from keeps the old inode mapped
    uint64_t pos = 0;
    auto wr = [&](const void* p, size_t b) { out.write(p, b); pos += b; };
    auto pad_to = [&](uint64_t off) {
        std::vector<uint8_t> z((size_t)(off - pos), 0);
        wr(z.data(), z.size());
    };
    wr(&h, sizeof(h));
    wr(small.data(), small.size());
    pad_to(h.own_off);
    wr(own, hi.n * own_stride);
    pad_to(h.raw_off);
    wr(hi.vec(0), hi.n * hi.D * 4);
    pad_to(h.blocks_off);
    wr(blocks, hi.n * (uint64_t)stride);
    out.commit();
}

// Maps the file and fills everything of `hi` except raw / search_data (raw_view points into the mapping).
inline NativeHeader read_native(const std::string& path, size_t expect_D, size_t expect_bw, size_t expect_dim, HostIndex& hi,
                                NativeMapping& map) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Cannot open file for reading: " + path);
    struct stat st{};
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(NativeHeader)) { ::close(fd); throw std::runtime_error("Read error or truncated file: " + path); }
    void* base = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) throw std::runtime_error("Cannot map file: " + path);
    NativeMapping m;
    m.base = base;
    m.bytes = (size_t)st.st_size;
    NativeHeader h;
    std::memcpy(&h, base, sizeof(h));
    if (h.magic != kNativeMagic) throw std::runtime_error("Invalid magic bytes (not a CP-HNSW MI355X native index file).");
    if (h.version != kNativeVersion) throw std::runtime_error("Unsupported native index file version: " + std::to_string(h.version));
    if (h.D != expect_D || h.bw != expect_bw)
        throw std::runtime_error("Index file template parameters mismatch: file D=" + std::to_string(h.D) + " R=32 BW=" +
                                 std::to_string(h.bw) + ", expected D=" + std::to_string(expect_D) + " R=32 BW=" + std::to_string(expect_bw));
    if (h.dim != expect_dim)
        throw std::runtime_error("Index file dim=" + std::to_string(h.dim) + " mismatches Index dim=" + std::to_string(expect_dim));
    // Nothing below trusts the header: every section must lie inside the mapping, in order and without overlap,
    // with the strides this library computes itself (a wrong offset would otherwise be a SIGBUS in a memcpy, which
    // no exception handler catches).
    {
        typedef unsigned __int128 u128;
        const DevLayout DL = make_dev_layout(h.D, h.bw);
        const RefLayout RLc = make_ref_layout(h.D, h.bw);
        const bool ok = h.n >= 1 && h.n < 0xFFFFFFFFull && h.stride == DL.stride && h.own_stride == RLc.nb_off &&
                        h.n_layers <= 64 &&
                        (u128)sizeof(NativeHeader) + h.small_bytes <= h.own_off &&
                        (u128)h.own_off + (u128)h.n * h.own_stride <= h.raw_off &&
                        (u128)h.raw_off + (u128)h.n * h.D * 4 <= h.blocks_off &&
                        (u128)h.blocks_off + (u128)h.n * h.stride == h.file_bytes && h.file_bytes <= m.bytes &&
                        h.raw_off % 4 == 0 && h.blocks_off % 4 == 0 &&
                        (u128)h.small_bytes >= (u128)320 + (u128)h.dim * 4 + (u128)h.n * 8;
        if (!ok) throw std::runtime_error("Read error or truncated file: " + path);
        if (h.entry != kInvalidNode && h.entry >= h.n) throw std::runtime_error("Corrupt index: entry point out of range");
    }
    HostIndex t;
    t.D = h.D; t.bw = h.bw; t.dim = h.dim; t.n = h.n; t.max_level = h.max_level; t.entry = h.entry;
    t.upper_tau = h.upper_tau; t.upper_alpha = h.upper_alpha; t.mL = h.mL; t.seed = h.seed;
    t.has_dup_neighbors = h.has_dup != 0;
    t.RL = make_ref_layout(t.D, t.bw);
    const uint8_t* p = static_cast<const uint8_t*>(base) + sizeof(NativeHeader);
    const uint8_t* end = p + h.small_bytes;
    auto get = [&](void* dst, size_t b) {
        if (p + b > end) throw std::runtime_error("Read error or truncated file: " + path);
        std::memcpy(dst, p, b);
        p += b;
    };
    get(t.calib, 248);
    get(t.profile, 72);
    t.centroid.resize(t.dim);  get(t.centroid.data(), t.dim * 4);
    t.levels.resize(t.n);      get(t.levels.data(), t.n * 4);
    t.norm_sq.resize(t.n);     get(t.norm_sq.data(), t.n * 4);
    t.upper.resize(h.n_layers);
    for (auto& layer : t.upper) {
        uint32_t sz = 0;
        get(&sz, 4);
        if (sz > t.n) throw std::runtime_error("Corrupt index: upper layer larger than the index");
        layer.resize(sz);
        for (auto& e : layer) {
            uint32_t cnt = 0;
            get(&e.node, 4);
            get(&cnt, 4);
            if ((size_t)cnt * 4 > (size_t)(end - p)) throw std::runtime_error("Read error or truncated file: " + path);
            e.nbrs.resize(cnt);
            get(e.nbrs.data(), (size_t)cnt * 4);
            if (e.node >= t.n) throw std::runtime_error("Corrupt index: upper-layer node out of range");
            for (uint32_t x : e.nbrs)
                if (x >= t.n) throw std::runtime_error("Corrupt index: upper-layer neighbour out of range");
        }
    }
    // The device blocks are handed to the GPU as they are: a neighbour id the search kernel would chase must be a
    // vertex (the v2 loader's validate() makes the same promise), unused slots must carry the invalid marker the
    // kernels rely on instead of `count`, and the repeated-id flag is recomputed rather than believed.
    {
        const DevLayout DL = make_dev_layout(h.D, h.bw);
        const uint8_t* blocks = static_cast<const uint8_t*>(base) + h.blocks_off;
        std::atomic<int> bad{0}, dup{0};
        parallel_for(t.n, 4096, [&](size_t lo, size_t hi_) {
            int b = 0, d = 0;
            for (size_t v = lo; v < hi_ && !b; ++v) {
                const uint8_t* blk = blocks + v * (size_t)DL.stride;
                uint32_t cnt, ids[32];
                std::memcpy(&cnt, blk + DL.count_off, 4);
                std::memcpy(ids, blk + DL.ids_off, 128);
                if (cnt > 32) { b = 1; break; }
                for (uint32_t i = 0; i < 32; ++i) {
                    if (i < cnt) {
                        if (ids[i] >= t.n) b = 2;
                        for (uint32_t j = 0; j < i; ++j) d |= ids[j] == ids[i];
                    } else if (ids[i] != kInvalidNode) b = 3;
                }
            }
            if (b) bad.store(b);
            if (d) dup.store(1);
        });
        if (bad.load() == 1) throw std::runtime_error("Corrupt index: neighbour count > 32");
        if (bad.load()) throw std::runtime_error("Corrupt index: neighbour id out of range");
        t.has_dup_neighbors = dup.load() != 0;
    }
    t.raw_view = reinterpret_cast<const float*>(static_cast<const uint8_t*>(base) + h.raw_off);
    t.rot.init(t.D, t.seed);
    hi = std::move(t);
    map = std::move(m);
    return h;
}

}  // namespace cph
