// cphnsw_mi355x.hip — C-ABI of the MI355X-native CP-HNSW hot path (include/cphnsw_mi355x.h).
//
// Host orchestration only; the arithmetic lives in device_fastscan.h / device_search.h /
// device_stream.h (GPU) and host_index.h (per-query feeders still on the host).
// The product path has no CPU fallback: without a HIP device every compute entry point
// fails with CPH_RUNTIME_ERROR.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cphnsw_mi355x.h"
#include "cph_core.h"
#include "device_buf.h"
#include "device_encode.h"
#include "device_fastscan.h"
#include "device_search.h"
#include "device_stream.h"
#include "device_heap_test.h"
#include "host_index.h"
#include "host_parallel.h"
#include "search_coalescer.h"
#include "native_file.h"
#include "builder_pipeline.h"

using namespace cph;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

struct InvalidArg : std::invalid_argument {
    using std::invalid_argument::invalid_argument;
};

template <class F>
int guarded(F&& f) {
    try {
        f();
        return CPH_OK;
    } catch (const std::invalid_argument& e) {
        return fail(CPH_INVALID_ARGUMENT, e.what());
    } catch (const std::bad_alloc&) {
        return fail(CPH_OUT_OF_MEMORY, "out of memory");
    } catch (const std::exception& e) {
        return fail(CPH_RUNTIME_ERROR, e.what());
    }
}

size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p *= 2;
    return p;
}

// kernel dispatch over (bits, static D): D == 128 and D == 1024 (the BASELINE shapes) get instantiations with a
// compile-time D -- every code load of a block in flight at once, the vertex vector through LDS-DMA
#define CPH_LAUNCH_BITS(KERNEL, bits, SDV, grid, block, lds, st, args)                            \
    do {                                                                                          \
        if ((bits) == 1) hipLaunchKernelGGL((KERNEL<1, SDV>), grid, block, lds, st, args);        \
        else if ((bits) == 2) hipLaunchKernelGGL((KERNEL<2, SDV>), grid, block, lds, st, args);   \
        else hipLaunchKernelGGL((KERNEL<4, SDV>), grid, block, lds, st, args);                    \
    } while (0)
#define CPH_LAUNCH(KERNEL, bits, D, grid, block, lds, st, args)                                   \
    do {                                                                                          \
        if ((D) == 128) CPH_LAUNCH_BITS(KERNEL, bits, 128, grid, block, lds, st, args);           \
        else if ((D) == 1024) CPH_LAUNCH_BITS(KERNEL, bits, 1024, grid, block, lds, st, args);    \
        else CPH_LAUNCH_BITS(KERNEL, bits, 0, grid, block, lds, st, args);                        \
        HIP_CHECK(hipGetLastError());                                                             \
    } while (0)

// Sets bit 1 of *flags when some vertex' neighbour list has a length that is not a multiple of 8: such a list ends in the
// reference's scalar tail (device_fastscan.h: TailLanes), which the probe-first search instantiation leaves out.
__global__ void scan_counts_kernel(const uint8_t* blocks, uint64_t n, uint32_t stride, uint32_t count_off, uint32_t* flags) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t cnt = *reinterpret_cast<const uint32_t*>(blocks + v * stride + count_off);
    if (cnt & 7u) atomicOr(flags, 2u);
}

}  // namespace

// One in-flight batch: query staging, outputs, statistics and per-slot scratch.  A handle owns two,
// used alternately, so that a batch enqueued on one stream can start while the previous one (on
// another stream) is still draining its longest queries.
struct BatchSet {
    DevBuf<float> d_queries_raw, d_queries, d_entry_dist, d_dist;
    DevBuf<uint4> d_qmasks;
    DevBuf<QueryHeader> d_qhdr;
    DevBuf<int64_t> d_ids;
    DevBuf<uint32_t> d_count, d_status, d_order, d_redo;
    // u64 words: [0..15] counters | [16] lo = work-queue counter, hi = length of the re-run list |
    // [17] lo = work-queue counter of the re-run launch
    DevBuf<unsigned long long> d_stats;
    unsigned long long* pin_stats = nullptr;   // pinned host copy, written at the end of every batch
    // small batches (cph_search, cph_search_batch with a handful of queries): queries are read and results written by
    // the kernels straight from / to this pinned, device-mapped host buffer -- no copy commands at all
    uint8_t* pin_io = nullptr;
    uint8_t* pin_io_dev = nullptr;
    size_t pin_io_bytes = 0;
    // per-slot scratch (estimated-set bitmap, beam spill area, id log), `cap` entries per slot
    DevBuf<uint32_t> d_bitmaps, d_logids;
    DevBuf<uint32_t> d_beam, d_beam_tail;
    uint32_t slots = 0;
    uint64_t cap = 0;
    // full-capacity (n + 1) scratch of the overflow re-run launch
    DevBuf<uint32_t> r_bitmaps, r_logids;
    DevBuf<uint32_t> r_beam, r_beam_tail;
    uint32_t r_slots = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    bool used = false;        // a batch has been enqueued on this set (ev_done is meaningful)
    uint32_t nq = 0;          // size of that batch
    uint32_t run_slots = 0;   // slots its search launch used
    uint64_t run_cap = 0;     // per-slot capacity that launch ran with (n + 1 on the full-capacity slots of the small-batch path)
    bool stats_in_hbm = false; // the last batch's statistics were not copied to pin_stats yet
};
constexpr int kStatWords = 18;
constexpr int kMaxBatchSets = 4;
constexpr uint64_t kSmallBatch = 32;        // batches up to this size take the copy-free path of cph_search / cph_search_batch

struct cph_index {
    uint64_t dim = 0;
    uint32_t bits = 0;
    uint32_t D = 0;
    int device = 0;
    bool finalized = false;
    bool needs_build = false;          // build() done, finalize() pending
    std::vector<float> pending;        // vectors handed to build()
    uint64_t pending_n = 0;
    HostIndex host;
    DevLayout L{};
    SearchConsts sc{};
    uint32_t flags = 0;
    int num_cus = 256;
    bool waves_from_env = false;
    uint32_t waves_per_cu = 4 * CPH_SEARCH_WAVES_PER_SIMD;  // resident waves per CU (launch bounds of the search kernel)
    // device-resident index
    DevBuf<uint8_t> d_blocks;
    DevBuf<float> d_raw, d_norm;
    // per-query feeders on the device: rotation signs + upper layers (CSR)
    DevBuf<float> d_signs;
    DevBuf<uint32_t> d_upper;          // all layers' nodes | offsets | nbrs, concatenated
    DevBuf<uint32_t> d_row_of;
    UpperLayerDev layers[kMaxUpperLayers];
    int32_t dev_max_level = 0;
    float norm_factor = 0.0f, inv_sqrt_d = 0.0f;
    // an index loaded from a native file keeps the mapping: vectors and own-code headers are served from it
    NativeMapping native_map;
    const uint8_t* own_view = nullptr;
    std::vector<uint8_t> own_store;    // own-code headers of an index built here (own_view points into it)
    // [0, kMaxBatchSets): the sets batches rotate over; behind them one private set per leader slot of cph_search
    BatchSet sets[kMaxBatchSets + kLeaderSlots];
    int n_sets = 2;                    // sets in rotation (cph_set_batch_sets): batches that may be in flight together
    int last_set = kMaxBatchSets - 1;  // the set handed out last (they take turns)
    int last_search = -1;              // the set the most recent search went to
    hipStream_t own_stream = nullptr;  // host-API calls (cph_search_batch, cph_search, hooks)
    bool order_queries = true;         // CPH_QUERY_ORDER=0 disables the closest-entry-first launch order
    // knobs
    uint32_t want_slots = 0;
    uint64_t want_cap = 0;
    uint64_t auto_cap = 0;             // grown when a batch had to re-run queries
    bool pf_off = false;               // probe first switched off: a batch sent > 2 % of its queries to the re-run launch for a stage-2 decision
    bool pf_dense = false;             // ... or suspended: the last batches found more than 6 new neighbours per expansion
    std::mutex mu;
    // concurrent cph_search callers (the reference: shared lock, T threads search in parallel, api/hnsw_index.hpp:172):
    // whoever finds no launch in flight leads one for everybody queued so far (search_coalescer.h: the policy, host only)
    SearchCoalescer coal;
    // a leader slot's resources: a stream, a pinned device-mapped I/O buffer and a batch set (sets[kMaxBatchSets + i])
    struct LeaderSlot {
        hipStream_t stream = nullptr;
        uint8_t* pin = nullptr;        // [flags kLeaderGroup x u32 | ids n*k*8 | dist n*k*4 | counts n*4 | queries n*dim*4]
        uint8_t* pin_dev = nullptr;
        size_t pin_bytes = 0;
        uint32_t seq = 0;              // launch counter: the value the kernels write into the flags of THIS launch
        uint64_t cur_n = 0, cur_k = 0; // shape of the launch in flight (for the callers' own copies)
    } leaders[kLeaderSlots];

    void use_device() const { HIP_CHECK(hipSetDevice(device)); }
};

namespace {

// The batch launch of this index runs the probe-first instantiation of the search kernel (D = 128, D = 1024): it sees only the new
// neighbours' codes, so a query whose stage-2 decision needs the others, and every index with short neighbour lists (flags
// bit 1: scalar tails), goes to the instantiation without it; so does a workload on which most neighbours are new (pf_dense).
bool probe_first(const cph_index* h) {
    return (h->L.D == 128 || h->L.D == 1024) && !(h->flags & 2u) && !h->pf_off && !h->pf_dense;
}

void require_finalized(cph_index* h) {
    // the reference does not check (it would hit the invalid-entry RuntimeError or garbage,
    // api/hnsw_index.hpp:203-205); we raise that error up front
    if (!h->finalized) throw std::runtime_error("Search failed: invalid entry point after finalize.");
}

void release_scratch(BatchSet& s) {
    s.d_bitmaps.release(); s.d_logids.release(); s.d_beam.release(); s.d_beam_tail.release();
    s.r_bitmaps.release(); s.r_logids.release(); s.r_beam.release(); s.r_beam_tail.release();
    s.slots = 0; s.cap = 0; s.r_slots = 0;
}

// Waits (on the host) until nothing enqueued on this handle is running any more.
void quiesce(cph_index* h) {
    for (auto& s : h->sets)
        if (s.used && s.ev_done) HIP_CHECK(hipEventSynchronize(s.ev_done));
}

// The big arrays of a loaded index: blocks repacked into the device layout, vectors, norms.  (An index
// built here already has them in HBM: build::build_graph.)
void upload_arrays(cph_index* h) {
    h->use_device();
    quiesce(h);
    const HostIndex& hi = h->host;
    h->L = make_dev_layout((uint32_t)hi.D, (uint32_t)hi.bw);
    const size_t n = hi.n;
    const size_t stride = h->L.stride;
    h->d_blocks.alloc(n * stride + 64);
    h->d_raw.alloc(n * hi.D);
    h->d_norm.alloc(n);
    // repack in chunks through a staging buffer
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, (256u << 20) / stride));
    std::vector<uint8_t> stage(chunk * stride);
    for (size_t base = 0; base < n; base += chunk) {
        const size_t cnt = std::min(chunk, n - base);
        parallel_for(cnt, 256, [&](size_t lo, size_t hi_) {
            for (size_t v = lo; v < hi_; ++v)
                repack_ref_to_dev(hi.nb(base + v), hi.RL, h->L, &stage[v * stride]);
        });
        HIP_CHECK(hipMemcpy(h->d_blocks.p + base * stride, stage.data(), cnt * stride,
                            hipMemcpyHostToDevice));
    }
    HIP_CHECK(hipMemcpy(h->d_raw.p, hi.raw.data(), n * hi.D * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->d_norm.p, hi.norm_sq.data(), n * 4, hipMemcpyHostToDevice));
}

// Everything else the query path needs on the device: search constants, rotation signs, upper layers.
void upload_feeders(cph_index* h) {
    h->use_device();
    quiesce(h);
    const HostIndex& hi = h->host;
    const size_t n = hi.n;
    h->L = make_dev_layout((uint32_t)hi.D, (uint32_t)hi.bw);
    h->sc = hi.consts();
    h->flags = hi.has_dup_neighbors ? 1u : 0u;
    if (n != 0) {   // bit 1: short lists with a scalar tail (counted on the device: the blocks of a built index never visit the host)
        DevBuf<uint32_t> d_flag(1);
        HIP_CHECK(hipMemset(d_flag.p, 0, 4));
        hipLaunchKernelGGL(scan_counts_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, h->d_blocks.p, (uint64_t)n,
                           h->L.stride, h->L.count_off, d_flag.p);
        HIP_CHECK(hipGetLastError());
        uint32_t f = 0;
        HIP_CHECK(hipMemcpy(&f, d_flag.p, 4, hipMemcpyDeviceToHost));
        h->flags |= f & 2u;
    }
    // rotation signs and the scale factors of the query encoder (rabitq_encoder.hpp:37-39)
    h->d_signs.alloc(3 * hi.D);
    HIP_CHECK(hipMemcpy(h->d_signs.p, hi.rot.signs.data(), 3 * hi.D * 4, hipMemcpyHostToDevice));
    const float d = static_cast<float>(hi.D);
    h->norm_factor = 1.0f / (d * std::sqrt(d));
    h->inv_sqrt_d = 1.0f / std::sqrt(d);
    // upper layers as CSR (edges are stored sorted by node id, api/hnsw_index.hpp:152)
    const int nl = std::min<int>({(int)hi.upper.size(), (int)hi.max_level, kMaxUpperLayers});
    h->dev_max_level = hi.max_level > 0 ? nl : 0;
    std::vector<uint32_t> flat;
    std::vector<size_t> off_nodes(nl), off_offs(nl), off_nbrs(nl);
    for (int l = 0; l < nl; ++l) {
        const auto& layer = hi.upper[l];
        off_nodes[l] = flat.size();
        for (const auto& e : layer) flat.push_back(e.node);
        off_offs[l] = flat.size();
        uint32_t acc = 0;
        for (const auto& e : layer) { flat.push_back(acc); acc += (uint32_t)e.nbrs.size(); }
        flat.push_back(acc);
        off_nbrs[l] = flat.size();
        for (const auto& e : layer) flat.insert(flat.end(), e.nbrs.begin(), e.nbrs.end());
    }
    h->d_upper.alloc(flat.size() + 1);
    if (!flat.empty())
        HIP_CHECK(hipMemcpy(h->d_upper.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < kMaxUpperLayers; ++l) h->layers[l] = UpperLayerDev{nullptr, nullptr, nullptr, nullptr, 0};
    // dense vertex -> row maps for the layers where a binary search would be a long chain of
    // dependent loads (4 B x n each)
    int mapped = 0;
    for (int l = 0; l < nl; ++l) mapped += hi.upper[l].size() > 16 ? 1 : 0;
    h->d_row_of.alloc((size_t)mapped * n + 1);
    std::vector<uint32_t> row_of;
    for (int l = 0, m = 0; l < nl; ++l) {
        const uint32_t* dmap = nullptr;
        if (hi.upper[l].size() > 16) {
            // vertex -> first edge | degree << 26 (the encoder then needs neither `nodes` nor `offsets`)
            row_of.assign(n, kInvalidNode);
            uint64_t first = 0;
            bool fits = true;
            for (size_t r = 0; r < hi.upper[l].size(); ++r) {
                const uint64_t deg = hi.upper[l][r].nbrs.size();
                if (first >= (1u << 26) || deg > 62) { fits = false; break; }
                row_of[hi.upper[l][r].node] = (uint32_t)first | ((uint32_t)deg << 26);
                first += deg;
            }
            if (!fits) { h->layers[l] = UpperLayerDev{h->d_upper.p + off_nodes[l], h->d_upper.p + off_offs[l], h->d_upper.p + off_nbrs[l], nullptr, (uint32_t)hi.upper[l].size()}; ++m; continue; }
            HIP_CHECK(hipMemcpy(h->d_row_of.p + (size_t)m * n, row_of.data(), n * 4, hipMemcpyHostToDevice));
            dmap = h->d_row_of.p + (size_t)m * n;
            ++m;
        }
        h->layers[l] = UpperLayerDev{h->d_upper.p + off_nodes[l], h->d_upper.p + off_offs[l],
                                     h->d_upper.p + off_nbrs[l], dmap, (uint32_t)hi.upper[l].size()};
    }
    for (auto& s : h->sets) release_scratch(s);
    h->auto_cap = 0;
    h->pf_off = false;
    h->pf_dense = false;
    h->last_search = -1;
}

// The reference-layout image of an index that came from a native file: own-code headers from the mapping,
// neighbour blocks re-derived from the device blocks.
void materialize_search_data(cph_index* h) {
    HostIndex& hi = h->host;
    if (!hi.search_data.empty() || hi.n == 0) return;
    h->use_device();
    quiesce(h);
    const size_t n = hi.n, stride = h->L.stride, own_stride = hi.RL.nb_off;
    hi.search_data.assign(n * hi.RL.vertex_bytes, 0);
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, (512u << 20) / stride));
    std::vector<uint8_t> stage(chunk * stride);
    for (size_t base = 0; base < n; base += chunk) {
        const size_t c = std::min(chunk, n - base);
        HIP_CHECK(hipMemcpy(stage.data(), h->d_blocks.p + base * stride, c * stride, hipMemcpyDeviceToHost));
        parallel_for(c, 256, [&](size_t lo, size_t hi_) {
            for (size_t v = lo; v < hi_; ++v) {
                uint8_t* dst = &hi.search_data[(base + v) * hi.RL.vertex_bytes];
                if (h->own_view) std::memcpy(dst, h->own_view + (base + v) * own_stride, own_stride);
                repack_dev_to_ref(&stage[v * stride], h->L, hi.RL, dst + hi.RL.nb_off);
            }
        });
    }
}

// Picks the set for the next batch (the two alternate) and makes `st` wait for the batch that
// used it before.  If that batch had to re-run queries, later batches get a larger capacity.
void init_set(BatchSet& s) {
    if (s.ev0) return;
    HIP_CHECK(hipEventCreate(&s.ev0));
    HIP_CHECK(hipEventCreate(&s.ev1));
    HIP_CHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.pin_stats), kStatWords * 8, hipHostMallocDefault));
    std::memset(s.pin_stats, 0, kStatWords * 8);
}

BatchSet& next_set(cph_index* h, hipStream_t st) {
    h->last_set = (h->last_set + 1) % h->n_sets;
    BatchSet& s = h->sets[h->last_set];
    init_set(s);
    // (either set's finished batch counts: the other set's is the more recent one)
    for (BatchSet& o : h->sets) {
        if (!(o.used && o.ev_done && !o.stats_in_hbm && hipEventQuery(o.ev_done) == hipSuccess)) continue;
        // [5] = queries re-run, [8] = those of them that were re-run for a stage-2 decision (probe first), not for capacity
        if (o.pin_stats[5] > o.pin_stats[8] && o.cap < h->host.n + 1)
            h->auto_cap = std::max<uint64_t>(h->auto_cap, std::min<uint64_t>(h->host.n + 1, o.cap * 4));
        if (o.pin_stats[8] * 50 > o.nq) h->pf_off = true;
        // Probe first pays when few of a block's neighbours are new (C2: 3.3 of 32 -- 40 % of the code lines are never
        // fetched); on a workload where most expansions find new neighbours in every group of eight it only adds a
        // dependent round trip (Gaussian 1M at 4 bits: 9.7 new per expansion, 13 % slower with it).  Decided from the
        // finished batches' own counters, with hysteresis; results are the same either way.
        // Narrow codes have less to skip (512 B / 1 KB of codes against 2 KB) and lose the estimator's overlap with the
        // probe: their break-even is lower (2-bit: +12 % at 1.5 new per expansion, -3.5 % at 5.3; 1-bit: +8 % at 1.0).
        if (o.pin_stats[0] >= 10000) {
            const double new_per_exp = (double)o.pin_stats[2] / (double)o.pin_stats[0];
            const double off = h->bits == 4 ? 6.0 : 2.8, on = h->bits == 4 ? 4.5 : 2.2;
            if (new_per_exp > off) h->pf_dense = true;
            else if (new_per_exp < on) h->pf_dense = false;
        }
    }
    if (s.used) HIP_CHECK(hipStreamWaitEvent(st, s.ev_done, 0));
    return s;
}

// Encode the queries on the device (rotation, 4-bit scalars -> masks, coefficients) and run the
// upper-layer descent; d_raw_q = [nq][dim] raw queries already in HBM.
void stage_queries(cph_index* h, BatchSet& s, const float* d_raw_q, uint64_t nq, hipStream_t st) {
    const HostIndex& hi = h->host;
    const uint32_t D = (uint32_t)hi.D, PW = h->L.PW;
    if (hi.entry == kInvalidNode || hi.entry >= hi.n)
        throw std::runtime_error("Search failed: invalid entry point after finalize.");
    if (s.d_queries.n < nq * D || s.d_qmasks.n < nq * PW || s.d_qhdr.n < nq) {
        if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));   // growing: the old buffers must be idle
        s.d_queries.alloc(nq * D);
        s.d_qmasks.alloc(nq * PW);
        s.d_qhdr.alloc(nq);
        s.d_entry_dist.alloc(nq);
        s.d_order.alloc(nq);
    }
    EncodeArgs a{};
    a.queries_raw = d_raw_q;
    a.nq = (uint32_t)nq;
    a.dim = (uint32_t)hi.dim;
    a.D = D;
    a.PW = PW;
    a.signs = h->d_signs.p;
    a.norm_factor = h->norm_factor;
    a.inv_sqrt_d = h->inv_sqrt_d;
    a.raw = h->d_raw.p;
    a.n = hi.n;
    a.entry = hi.entry;
    a.max_level = h->dev_max_level;
    for (int l = 0; l < kMaxUpperLayers; ++l) a.layers[l] = h->layers[l];
    a.queries_padded = s.d_queries.p;
    a.qmasks = s.d_qmasks.p;
    a.qhdr = s.d_qhdr.p;
    a.entry_dist = s.d_entry_dist.p;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)h->num_cus * 32);
    hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(64), encode_lds_bytes(D), st, a);
    HIP_CHECK(hipGetLastError());
}

// host queries -> the set's staging buffer
const float* upload_queries(cph_index* h, BatchSet& s, const float* queries, uint64_t nq, hipStream_t st) {
    if (s.d_queries_raw.n < nq * h->dim) {
        if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));
        s.d_queries_raw.alloc(nq * h->dim);
    }
    HIP_CHECK(hipMemcpyAsync(s.d_queries_raw.p, queries, nq * h->dim * 4, hipMemcpyHostToDevice, st));
    return s.d_queries_raw.p;
}

// (Re)allocates the per-slot scratch of a set; the set must be idle.
void ensure_scratch(cph_index* h, BatchSet& s, uint32_t slots, uint64_t cap, hipStream_t st) {
    const uint64_t n = h->host.n;
    const uint64_t bm_words = (n + 31) / 32;
    if (!(s.slots >= slots && s.cap == cap)) {
        if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));
        s.d_bitmaps.alloc((size_t)slots * bm_words);
        HIP_CHECK(hipMemsetAsync(s.d_bitmaps.p, 0, (size_t)slots * bm_words * 4, st));
        s.d_logids.alloc((size_t)slots * cap);
        s.d_beam.alloc((size_t)slots * kBeamPagesDwords);
        s.d_beam_tail.alloc((size_t)slots * beam_tail_dwords(cap));
        s.slots = slots;
        s.cap = cap;
    }
    // the re-run launch (capacity overflows; stage-2 decisions of the probe-first instantiation) needs room for every
    // vertex: a few full-capacity slots
    if ((cap < n + 1 || probe_first(h)) && s.r_slots == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const uint64_t per = (n + 1) * 4 + ((uint64_t)kBeamPagesDwords + beam_tail_dwords(n + 1)) * 4 + bm_words * 4;
        // (a leader slot's private set answers at most kLeaderGroup callers per launch)
        const uint64_t want = (&s - h->sets) >= kMaxBatchSets ? kLeaderGroup : kSmallBatch;
        const uint32_t rs = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)(free_b * 0.25) / per));
        s.r_bitmaps.alloc((size_t)rs * bm_words);
        HIP_CHECK(hipMemsetAsync(s.r_bitmaps.p, 0, (size_t)rs * bm_words * 4, st));
        s.r_logids.alloc((size_t)rs * (n + 1));
        s.r_beam.alloc((size_t)rs * kBeamPagesDwords);
        s.r_beam_tail.alloc((size_t)rs * beam_tail_dwords(n + 1));
        s.r_slots = rs;
    }
}

// mode 0: the batch on the set's slots (capacity s.cap); 1: the overflow re-run on the full-capacity slots (its list of
// queries lives on the device); 2: a batch small enough for the full-capacity slots, run there directly (no re-run needed)
struct DoneFlags {          // per-query completion flags in pinned host memory (coalesced cph_search), or none
    uint32_t* flags = nullptr;
    uint32_t seq = 0;
};

void launch_search(cph_index* h, BatchSet& s, uint32_t nq, uint32_t k, int64_t* d_ids, float* d_dist, uint32_t* d_count,
                   const uint32_t* d_todo, int mode, hipStream_t st, DoneFlags done = DoneFlags()) {
    SearchArgs a{};
    a.done_flags = done.flags;
    a.done_seq = done.seq;
    a.blocks = h->d_blocks.p;
    a.raw = h->d_raw.p;
    a.norm_sq = h->d_norm.p;
    a.n = h->host.n;
    a.L = h->L;
    a.flags = h->flags;
    a.queries = s.d_queries.p;
    a.qmasks = s.d_qmasks.p;
    a.qhdr = s.d_qhdr.p;
    a.k = k;
    a.sc = h->sc;
    a.bm_words = (h->host.n + 31) / 32;
    a.out_ids = d_ids;
    a.out_dist = d_dist;
    a.out_count = d_count;
    a.status = s.d_status.p;
    a.stats = s.d_stats.p;
    uint32_t* words = reinterpret_cast<uint32_t*>(s.d_stats.p + 16);   // [0] queue, [1] re-run list length, [2] re-run queue
    uint32_t grid;
    if (mode == 0) {
        a.todo = d_todo;
        a.nq = nq;
        a.counter = words;
        a.cap = s.cap;
        a.bitmaps = s.d_bitmaps.p;
        a.beam_pages = s.d_beam.p;
        a.beam_tail = s.d_beam_tail.p;
        a.log_ids = s.d_logids.p;
        a.redo = (s.cap < h->host.n + 1 || probe_first(h)) ? s.d_redo.p : nullptr;
        a.redo_count = words + 1;
        grid = s.run_slots;
    } else {
        a.todo = mode == 1 ? s.d_redo.p : nullptr;
        a.nq = mode == 1 ? 0 : nq;
        a.nq_dev = mode == 1 ? words + 1 : nullptr;
        a.counter = mode == 1 ? words + 2 : words;
        a.cap = h->host.n + 1;
        a.bitmaps = s.r_bitmaps.p;
        a.beam_pages = s.r_beam.p;
        a.beam_tail = s.r_beam_tail.p;
        a.log_ids = s.r_logids.p;
        grid = mode == 1 ? s.r_slots : nq;
    }
    const size_t lds = search_lds_bytes(h->L.D, h->L.PW, k);
    if (lds > 160 * 1024) throw InvalidArg("k too large for the on-chip result heap");
    if ((mode != 0 || !probe_first(h)) && h->L.D == 1024) {
        if (h->bits == 1) hipLaunchKernelGGL((search_kernel<1, 1024, false>), dim3(grid), dim3(64), lds, st, a);
        else if (h->bits == 2) hipLaunchKernelGGL((search_kernel<2, 1024, false>), dim3(grid), dim3(64), lds, st, a);
        else hipLaunchKernelGGL((search_kernel<4, 1024, false>), dim3(grid), dim3(64), lds, st, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if ((mode != 0 || !probe_first(h)) && h->L.D == 128) {
        // a handful of queries: latency, not traffic -- the order of loads without the third dependent round trip.  Also
        // the instantiation of the re-run launch (it takes the stage-2 decisions the probe-first one hands over), of an
        // index with short neighbour lists (flags bit 1: it evaluates their scalar tails) and of workloads on which
        // probe first does not pay.
        if (h->bits == 1) hipLaunchKernelGGL((search_kernel<1, 128, false>), dim3(grid), dim3(64), lds, st, a);
        else if (h->bits == 2) hipLaunchKernelGGL((search_kernel<2, 128, false>), dim3(grid), dim3(64), lds, st, a);
        else hipLaunchKernelGGL((search_kernel<4, 128, false>), dim3(grid), dim3(64), lds, st, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    CPH_LAUNCH(search_kernel, h->bits, h->L.D, dim3(grid), dim3(64), lds, st, a);
}

// Core: queries already staged in the set; results into device buffers.  Everything is enqueued on
// `st` and nothing waits for the device: a query that outgrows its scratch is answered by the
// full-capacity re-run launch that always follows the main one (it finds an empty list otherwise).
void enqueue_search(cph_index* h, BatchSet& s, uint32_t nq, uint32_t k, int64_t* d_ids, float* d_dist,
                    hipStream_t st, uint32_t* d_count_out = nullptr, DoneFlags done = DoneFlags()) {
    const uint64_t n = h->host.n;
    if (s.d_count.n < nq) {
        if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));
        s.d_count.alloc(nq);
        s.d_status.alloc(nq);
        s.d_redo.alloc(nq);
    }
    s.d_stats.alloc(kStatWords);
    HIP_CHECK(hipMemsetAsync(s.d_stats.p, 0, kStatWords * 8, st));
    // resident query slots: one wave each
    uint32_t wpc = h->waves_per_cu;
    if (!h->waves_from_env) {
        // registers (launch bounds of the instantiation) and LDS (160 KB per CU) both cap the resident waves
        wpc = 4 * (uint32_t)search_waves_per_simd(search_static_d(h->L.D) ? (int)h->L.D : 0, (int)h->bits);
        const size_t lds_wave = search_lds_bytes(h->L.D, h->L.PW, k);
        wpc = (uint32_t)std::max<size_t>(1, std::min<size_t>(wpc, (160u * 1024u) / lds_wave));
    }
    uint32_t max_slots = h->want_slots ? h->want_slots : (uint32_t)h->num_cus * wpc;
    // More than two sets in rotation: a batch gets exactly half of the resident slots, so that two batches run side by
    // side at full occupancy while the others queue behind them -- the drain of one is filled by the start of the
    // next-but-one (C2, 10k-query batches on four streams: 5.34 -> 5.61 M QPS; 2,560 / 3,584 / 4,096 slots per batch or
    // three sets all lose against two sets with every slot, profiles/r3_streams_sweep.md)
    if (!h->want_slots && h->n_sets > 2) max_slots = std::max<uint32_t>(64, max_slots / 2);
    // balanced rounds: every slot runs the same number of queries (10k queries on 4096 slots
    // would leave 56% of the slots idle during the third round)
    const uint32_t rounds = (nq + max_slots - 1) / max_slots;
    uint32_t slots = std::min<uint32_t>(nq, (nq + rounds - 1) / rounds);
    // ... unless the batch is launched longest-first (below): then every slot is worth having, the
    // short queries at the end of the order fill the gaps
    const bool ordered = h->dev_max_level > 0 && h->order_queries;
    if (ordered) {
        slots = std::min<uint32_t>(nq, max_slots);
        // A batch that fills more than half of the slots gains from a short queue: the order starts with the longest
        // queries, the last eighth -- the shortest -- fills the gaps they leave (C2, 6,144 slots: 4,000 queries
        // 1.37 -> 1.31 ms, 6,000 queries 1.65 -> 1.61 ms; below 3,000 queries all resident is 0-7 % faster;
        // scripts/slot_fraction_sweep.py, profiles/r2_slot_fraction.md)
        if (!h->want_slots && nq > max_slots / 2) slots = std::min<uint32_t>(max_slots, nq - nq / 8);
    }
    const uint64_t bm_bytes = ((n + 31) / 32) * 4;
    uint64_t cap = h->want_cap ? h->want_cap : std::max<uint64_t>(h->auto_cap, std::min<uint64_t>(n + 1, 1u << 16));
    cap = std::max<uint64_t>(64, std::min<uint64_t>(cap, n + 1));
    if (!(s.slots >= slots && s.cap == cap)) {
        // budget: at most 30% of what is free (plus what this set already holds) -- there are two sets
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        auto slot_bytes = [&](uint64_t c) { return c * 4 + ((uint64_t)kBeamPagesDwords + beam_tail_dwords(c)) * 4 + bm_bytes; };   // id log + beam spill + bitmap
        const uint64_t held = (uint64_t)s.slots * slot_bytes(s.cap);
        const uint64_t budget = (uint64_t)((free_b + held) * 0.3);
        while (slots > 64 && (uint64_t)slots * slot_bytes(cap) > budget) slots /= 2;
        while (cap > 4096 && (uint64_t)slots * slot_bytes(cap) > budget) cap /= 2;
    }
    ensure_scratch(h, s, slots, cap, st);
    slots = std::min(slots, s.slots);
    s.run_slots = slots;
    s.run_cap = s.cap;
    s.nq = nq;
    // closest-entry-first launch order (device_encode.h) when the batch outnumbers the slots
    const uint32_t* d_order = nullptr;
    if (nq > slots && ordered) {
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, st, s.d_entry_dist.p, nq, s.d_order.p);
        HIP_CHECK(hipGetLastError());
        d_order = s.d_order.p;
    }
    uint32_t* d_count = d_count_out ? d_count_out : s.d_count.p;
    HIP_CHECK(hipEventRecord(s.ev0, st));
    const bool rerun = s.cap < n + 1 || probe_first(h);
    if (rerun && nq <= s.r_slots && !h->want_cap && !h->want_slots) {   // (explicit search params keep the general path)
        // a handful of queries: straight onto the full-capacity slots -- one launch, nothing can overflow
        s.run_slots = nq;
        s.run_cap = n + 1;
        launch_search(h, s, nq, k, d_ids, d_dist, d_count, nullptr, 2, st, done);
    } else {
        launch_search(h, s, nq, k, d_ids, d_dist, d_count, d_order, 0, st, done);
        if (rerun) launch_search(h, s, nq, k, d_ids, d_dist, d_count, nullptr, 1, st, done);
    }
    HIP_CHECK(hipEventRecord(s.ev1, st));
    // the statistics block lands in pinned host memory; it is only read when somebody asks.  (The private sets of the
    // cph_search leader slots leave it in HBM until then: their callers wait for the stream, and the copy command would
    // sit on that path -- nothing of theirs can overflow, so nobody reads the block unasked.)
    s.stats_in_hbm = (&s - h->sets) >= kMaxBatchSets;
    if (!s.stats_in_hbm) HIP_CHECK(hipMemcpyAsync(s.pin_stats, s.d_stats.p, kStatWords * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(s.ev_done, st));
    s.used = true;
    h->last_search = (int)(&s - h->sets);
}

hipStream_t own_stream(cph_index* h) {
    if (!h->own_stream) HIP_CHECK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    return h->own_stream;
}

// Views into a set's pinned, device-mapped I/O buffer for a batch of `n` queries: [ids n*k*8 | dist n*k*4 | counts n*4 |
// queries n*dim*4], host and device addresses of the same bytes.
struct SmallIo {
    int64_t* h_ids; float* h_dist; uint32_t* h_count; float* h_query;
    int64_t* d_ids; float* d_dist; uint32_t* d_count; const float* d_query;
};
SmallIo small_io(cph_index* h, BatchSet& s, uint64_t n, uint64_t k) {
    const size_t o_dist = n * k * 8, o_cnt = o_dist + n * k * 4, o_q = (o_cnt + n * 4 + 15) & ~(size_t)15;
    const size_t need = o_q + n * h->dim * 4;
    if (s.pin_io_bytes < need) {
        if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));
        if (s.pin_io) HIP_CHECK(hipHostFree(s.pin_io));
        s.pin_io = nullptr;
        s.pin_io_bytes = 0;
        const size_t bytes = std::max<size_t>(need * 2, 64 * 1024);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.pin_io), bytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.pin_io_dev), s.pin_io, 0));
        s.pin_io_bytes = bytes;
    }
    uint8_t* hb = s.pin_io;
    uint8_t* db = s.pin_io_dev;
    return SmallIo{reinterpret_cast<int64_t*>(hb), reinterpret_cast<float*>(hb + o_dist), reinterpret_cast<uint32_t*>(hb + o_cnt),
                   reinterpret_cast<float*>(hb + o_q),
                   reinterpret_cast<int64_t*>(db), reinterpret_cast<float*>(db + o_dist), reinterpret_cast<uint32_t*>(db + o_cnt),
                   reinterpret_cast<const float*>(db + o_q)};
}

}  // namespace

// diagnostic build (-DCPH_SEARCH_TRACE): where a coalesced cph_search launch spends its host time
#ifdef CPH_SEARCH_TRACE
static std::atomic<uint64_t> g_tr[6];   // groups, callers, ns waiting for the handle mutex (per group), ns enqueuing (per group), ns each caller waited for its own query
static inline uint64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CPH_TR(i, v) g_tr[i] += (v)
#else
#define CPH_TR(i, v) do {} while (0)
static inline uint64_t now_ns() { return 0; }
#endif

// ---------------------------------------------------------------------------------------
extern "C" {

const char* cph_last_error(void) { return g_err.c_str(); }
int cph_version(void) { return 100; }

int cph_create(uint64_t dim, uint64_t bits, int device, cph_index** out) {
    return guarded([&] {
        if (!out) throw InvalidArg("out must not be null");
        *out = nullptr;
        // src/bindings.cpp:100-113, :77-98; api/hnsw_index.hpp:90
        if (bits != 1 && bits != 2 && bits != 4)
            throw InvalidArg("Unsupported bits=" + std::to_string(bits) + ". Supported: 1, 2, 4.");
        const size_t pd = next_pow2(dim);
        if (dim == 0) throw InvalidArg("dim must be > 0");
        if (pd < 16 || pd > 2048) {
            if (pd < 16) {
                // the reference pads only up to the next power of two; dims below 9 land on
                // padded sizes it does not instantiate
                throw InvalidArg("Unsupported dimension " + std::to_string(dim) + " (padded to " +
                                 std::to_string(pd) +
                                 "). Supported padded dims: 16, 32, 64, 128, 256, 512, 1024, 2048.");
            }
            throw InvalidArg("Unsupported dimension " + std::to_string(dim) + " (padded to " +
                             std::to_string(pd) +
                             "). Supported padded dims: 16, 32, 64, 128, 256, 512, 1024, 2048.");
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw std::runtime_error("No HIP device available: the MI355X path has no CPU fallback.");
        if (device < 0 || device >= ndev) throw InvalidArg("invalid device ordinal");
        auto* h = new cph_index();
        h->dim = dim;
        h->bits = (uint32_t)bits;
        h->D = (uint32_t)pd;
        h->device = device;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cus = prop.multiProcessorCount;
        if (const char* e = getenv("CPH_QUERY_ORDER")) h->order_queries = atoi(e) != 0;
        if (const char* e = getenv("CPH_LEADER_SLOTS")) h->coal.n_slots = std::max(1, std::min(kLeaderSlots, atoi(e)));
        if (const char* e = getenv("CPH_GATHER_US")) h->coal.gather_us = std::max(0, atoi(e));
        if (const char* e = getenv("CPH_WAVES_PER_CU")) { h->waves_per_cu = (uint32_t)std::max(1, atoi(e)); h->waves_from_env = true; }
        *out = h;
    });
}

int cph_destroy(cph_index* h) {
    return guarded([&] {
        if (!h) return;
        (void)hipSetDevice(h->device);
        for (auto& s : h->sets) {
            if (s.used && s.ev_done) (void)hipEventSynchronize(s.ev_done);
            if (s.ev0) (void)hipEventDestroy(s.ev0);
            if (s.ev1) (void)hipEventDestroy(s.ev1);
            if (s.ev_done) (void)hipEventDestroy(s.ev_done);
            if (s.pin_stats) (void)hipHostFree(s.pin_stats);
            if (s.pin_io) (void)hipHostFree(s.pin_io);
        }
#ifdef CPH_SEARCH_TRACE
        if (g_tr[0]) fprintf(stderr, "[search trace] groups=%llu callers=%llu per group: mutex wait %.1f us, enqueue %.1f us; per caller: own-query wait %.1f us\n",
                             (unsigned long long)g_tr[0], (unsigned long long)g_tr[1], g_tr[2] / 1e3 / g_tr[0], g_tr[3] / 1e3 / g_tr[0], g_tr[4] / 1e3 / g_tr[1]);
        for (auto& x : g_tr) x = 0;
#endif
        if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
        for (auto& ls : h->leaders) {
            if (ls.stream) { (void)hipStreamSynchronize(ls.stream); (void)hipStreamDestroy(ls.stream); }
            if (ls.pin) (void)hipHostFree(ls.pin);
        }
        delete h;
    });
}

// Called once the new host-side index has been read and validated: from here until the end of a load the handle is
// not searchable, so a failed device allocation or copy leaves it unfinalized (an error on the next search) instead of
// finalized over null or stale device pointers.
static void begin_device_swap(cph_index* h) {
    h->use_device();
    quiesce(h);
    h->finalized = false;
    for (auto& s : h->sets) release_scratch(s);
    h->last_search = -1;
}

int cph_load(cph_index* h, const char* path) {
    return guarded([&] {
        if (!h || !path) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        HostIndex t;
        t.load(path, h->D, h->bits, h->dim);        // a file that fails to parse leaves the handle as it was
        begin_device_swap(h);
        h->host = std::move(t);
        h->needs_build = false;                       // api/hnsw_index.hpp:442
        std::vector<float>().swap(h->pending);
        h->pending_n = 0;
        h->native_map.reset();
        h->own_view = nullptr;
        std::vector<uint8_t>().swap(h->own_store);
        upload_arrays(h);
        upload_feeders(h);
        h->finalized = true;
    });
}

int cph_save(cph_index* h, const char* path) {
    return guarded([&] {
        if (!h || !path) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->finalized) throw std::runtime_error("Index must be finalized before saving.");
        materialize_search_data(h);
        h->host.save(path);
    });
}

int cph_save_native(cph_index* h, const char* path) {
    return guarded([&] {
        if (!h || !path) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->finalized) throw std::runtime_error("Index must be finalized before saving.");
        h->use_device();
        quiesce(h);
        const HostIndex& hi = h->host;
        const size_t n = hi.n, stride = h->L.stride, own_stride = hi.RL.nb_off;
        std::vector<uint8_t> blocks(n * stride), own;
        HIP_CHECK(hipMemcpy(blocks.data(), h->d_blocks.p, n * stride, hipMemcpyDeviceToHost));
        const uint8_t* own_p = h->own_view;
        if (!own_p) {
            own.resize(n * own_stride);
            for (size_t v = 0; v < n; ++v) std::memcpy(&own[v * own_stride], &hi.search_data[v * hi.RL.vertex_bytes], own_stride);
            own_p = own.data();
        }
        write_native(path, hi, (uint32_t)stride, own_p, (uint32_t)own_stride, blocks.data());
    });
}

int cph_load_native(cph_index* h, const char* path) {
    return guarded([&] {
        if (!h || !path) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        HostIndex t;
        NativeMapping map;
        const NativeHeader nh = read_native(path, h->D, h->bits, h->dim, t, map);   // validates everything it maps
        begin_device_swap(h);
        h->host = std::move(t);
        h->needs_build = false;
        std::vector<float>().swap(h->pending);
        h->pending_n = 0;
        h->native_map = std::move(map);
        const uint8_t* base = static_cast<const uint8_t*>(h->native_map.base);
        std::vector<uint8_t>().swap(h->own_store);
        h->own_view = base + nh.own_off;
        h->L = make_dev_layout((uint32_t)h->host.D, (uint32_t)h->host.bw);
        const size_t n = h->host.n;
        h->d_blocks.alloc(n * nh.stride + 64);
        h->d_raw.alloc(n * h->host.D);
        h->d_norm.alloc(n);
        HIP_CHECK(hipMemcpy(h->d_blocks.p, base + nh.blocks_off, n * (size_t)nh.stride, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(h->d_raw.p, h->host.raw_view, n * h->host.D * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(h->d_norm.p, h->host.norm_sq.data(), n * 4, hipMemcpyHostToDevice));
        upload_feeders(h);
        h->finalized = true;
    });
}

int cph_size(cph_index* h, uint64_t* n) {
    return guarded([&] { *n = h->needs_build ? h->pending_n : h->host.n; });
}
int cph_dim(cph_index* h, uint64_t* dim) {
    return guarded([&] { *dim = h->dim; });
}
int cph_is_finalized(cph_index* h, int* flag) {
    return guarded([&] { *flag = h->finalized ? 1 : 0; });
}

int cph_build(cph_index* h, const float* vectors, uint64_t n) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        std::lock_guard<std::mutex> lk(h->mu);
        // api/hnsw_index.hpp:93-120: build() replaces any previous state
        if (n == 0) throw InvalidArg("build requires at least one vector.");
        if (!vectors) throw InvalidArg("null vectors");
        h->use_device();
        quiesce(h);
        h->host = HostIndex();
        h->native_map.reset();
        h->own_view = nullptr;
        std::vector<uint8_t>().swap(h->own_store);
        h->finalized = false;
        h->d_blocks.release(); h->d_raw.release(); h->d_norm.release();
        for (auto& s : h->sets) release_scratch(s);
        h->pending.assign(vectors, vectors + n * h->dim);
        h->pending_n = n;
        h->needs_build = true;
    });
}

int cph_finalize(cph_index* h) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        std::lock_guard<std::mutex> lk(h->mu);
        // api/hnsw_index.hpp:122-166
        const uint64_t n = h->needs_build ? h->pending_n : h->host.n;
        if (n == 0) throw std::runtime_error("Cannot finalize an empty index.");
        if (!h->needs_build) throw std::runtime_error("Finalize called without a pending build.");
        if (n < 50) throw std::runtime_error("Calibration requires at least 50 nodes.");
        if (n >= 0xFFFFFFFFull) throw InvalidArg("too many vectors");
        h->use_device();
        const bool verbose = getenv("CPH_BUILD_VERBOSE") != nullptr;
        quiesce(h);
        h->d_blocks.release(); h->d_raw.release(); h->d_norm.release();
        for (auto& s : h->sets) release_scratch(s);
        build::BuiltDevice dev;
        build::build_graph(h->host, dev, h->pending.data(), n, h->dim, h->D, h->bits, h->num_cus, verbose);
        std::vector<float>().swap(h->pending);
        h->pending_n = 0;
        h->needs_build = false;
        // the pipeline's device arrays are the searchable index: adopt them, no second upload
        h->d_blocks = std::move(dev.blocks);
        h->d_raw = std::move(dev.raw);
        h->d_norm = std::move(dev.norm);
        h->native_map.reset();
        h->own_store = std::move(dev.own_host);
        h->own_view = h->own_store.data();
        upload_feeders(h);
        build::DeviceIndexView view{h->d_blocks.p, h->d_raw.p, h->d_signs.p, h->L, h->norm_factor, h->inv_sqrt_d};
        build::calibrate(h->host, view, h->num_cus, verbose);
        h->sc = h->host.consts();
        h->finalized = true;
    });
}

int cph_knn_bruteforce(int device, const float* vectors, uint64_t n, uint64_t dim, const float* queries,
                       uint64_t nq, uint32_t* ids, float* dist) {
    return guarded([&] {
        if (!vectors || !ids || !dist || n == 0 || dim == 0) throw InvalidArg("bad arguments");
        if (n >= 0xFFFFFFFFull || nq >= 0xFFFFFFFFull) throw InvalidArg("too many rows");
        HIP_CHECK(hipSetDevice(device));
        const size_t D = std::max<size_t>(16, next_pow2(dim));
        auto pad = [&](const float* src, uint64_t rows, std::vector<float>& x, std::vector<float>& nrm) {
            x.assign(rows * D, 0.0f);
            nrm.resize(rows);
            parallel_for(rows, 1024, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    std::memcpy(&x[i * D], src + i * dim, dim * 4);
                    float s = 0.0f;
                    for (uint64_t j = 0; j < dim; ++j) s = std::fmaf(x[i * D + j], x[i * D + j], s);
                    nrm[i] = s;
                }
            });
        };
        std::vector<float> x, nrm, q, qn;
        pad(vectors, n, x, nrm);
        if (queries) pad(queries, nq, q, qn);
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
        build::gpu_knn(queries ? q.data() : nullptr, queries ? qn.data() : nullptr, nq, x.data(), nrm.data(), n, D,
                       cus, ids, dist);
    });
}

int cph_debug_heap_ops(int device, const uint8_t* ops, uint64_t n_ops, const float* keys, const uint32_t* ids, uint64_t n_push,
                       float* out_keys, uint32_t* out_ids, uint32_t* out_size) {
    return guarded([&] {
        if (!ops || !out_keys || !out_ids || !out_size || n_ops == 0 || n_ops > 0x7FFFFFFFull) throw InvalidArg("bad arguments");
        if (n_push && (!keys || !ids)) throw InvalidArg("bad arguments");
        uint64_t pushes = 0;
        for (uint64_t i = 0; i < n_ops; ++i) pushes += ops[i] ? 1 : 0;
        if (pushes != n_push) throw InvalidArg("n_push must equal the number of push operations");
        HIP_CHECK(hipSetDevice(device));
        DevBuf<uint8_t> d_ops(n_ops);
        DevBuf<float> d_keys(std::max<uint64_t>(1, n_push)), d_ok(std::max<uint64_t>(1, n_push));
        DevBuf<uint32_t> d_ids(std::max<uint64_t>(1, n_push)), d_oi(std::max<uint64_t>(1, n_push)), d_sz(1);
        DevBuf<uint32_t> d_spill(kBeamPagesDwords), d_spill_tail(beam_tail_dwords(std::max<uint64_t>(1, n_push) + 64));
        HIP_CHECK(hipMemcpy(d_ops.p, ops, n_ops, hipMemcpyHostToDevice));
        if (n_push) {
            HIP_CHECK(hipMemcpy(d_keys.p, keys, n_push * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(d_ids.p, ids, n_push * 4, hipMemcpyHostToDevice));
        }
        HeapTestArgs a{d_ops.p, d_keys.p, d_ids.p, (uint32_t)n_ops, d_spill.p, d_spill_tail.p, d_ok.p, d_oi.p, d_sz.p};
        hipLaunchKernelGGL(heap_selftest_kernel, dim3(1), dim3(64), 0, nullptr, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_size, d_sz.p, 4, hipMemcpyDeviceToHost));
        if (*out_size > n_push) throw std::runtime_error("heap self-test: size out of range");
        if (*out_size) {
            HIP_CHECK(hipMemcpy(out_keys, d_ok.p, (size_t)*out_size * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(out_ids, d_oi.p, (size_t)*out_size * 4, hipMemcpyDeviceToHost));
        }
    });
}

int cph_encode_edges(int device, uint64_t dim, uint64_t bits, const float* parent, const float* nbrs, uint64_t cnt,
                     uint8_t* values, float* aux, uint32_t* pops) {
    return guarded([&] {
        if (!parent || !nbrs || !values || !aux || !pops || cnt == 0 || cnt > 32 || dim == 0) throw InvalidArg("bad arguments");
        if (bits != 1 && bits != 2 && bits != 4) throw InvalidArg("bits must be 1, 2 or 4");
        const size_t D = std::max<size_t>(16, next_pow2(dim));
        if (D > 2048) throw InvalidArg("unsupported dimension");
        HIP_CHECK(hipSetDevice(device));
        const size_t n = cnt + 1;                       // vertex 0 = the parent, 1..cnt = its neighbours
        std::vector<float> x(n * D, 0.0f);
        std::memcpy(x.data(), parent, dim * 4);
        for (uint64_t e = 0; e < cnt; ++e) std::memcpy(&x[(e + 1) * D], nbrs + e * dim, dim * 4);
        std::vector<uint32_t> nbr(n * 32, kInvalidNode);
        for (uint64_t e = 0; e < cnt; ++e) nbr[e] = (uint32_t)(e + 1);
        Rotation rot;
        rot.init(D, 42);
        const DevLayout L = make_dev_layout((uint32_t)D, (uint32_t)bits);
        DevBuf<float> d_x(n * D), d_signs(3 * D), d_aux(n * 32 * 3);
        DevBuf<uint32_t> d_nbr(n * 32), d_pops(n * 32 * 2);
        DevBuf<uint8_t> d_blocks(n * L.stride), d_vals(n * 32 * D);
        HIP_CHECK(hipMemcpy(d_x.p, x.data(), n * D * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_nbr.p, nbr.data(), n * 32 * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_signs.p, rot.signs.data(), 3 * D * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(d_blocks.p, 0, n * L.stride));
        HIP_CHECK(hipMemset(d_vals.p, 0, n * 32 * D));
        HIP_CHECK(hipMemset(d_aux.p, 0, n * 32 * 3 * 4));
        HIP_CHECK(hipMemset(d_pops.p, 0, n * 32 * 2 * 4));
        build::EncodeArgsB a{};
        const float df = (float)D;
        a.x = d_x.p; a.nbr = d_nbr.p; a.n = n; a.dim = (uint32_t)dim; a.D = (uint32_t)D;
        a.signs = d_signs.p; a.norm_factor = 1.0f / (df * std::sqrt(df)); a.inv_sqrt_d = 1.0f / std::sqrt(df);
        a.L = L; a.blocks = d_blocks.p;
        a.dbg_values = d_vals.p; a.dbg_aux = d_aux.p; a.dbg_pops = d_pops.p;
        build::run_encode(bits, a, 1, 1);
        HIP_CHECK(hipMemcpy(values, d_vals.p, cnt * D, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(aux, d_aux.p, cnt * 3 * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(pops, d_pops.p, cnt * 2 * 4, hipMemcpyDeviceToHost));
    });
}

int cph_select_hook(int device, const float* x, uint64_t n, uint64_t D, uint32_t vertex, const uint32_t* fwd,
                    const uint32_t* rev, uint64_t n_rev, uint32_t R, float alpha, float tau, float alpha_max,
                    const float* err, uint32_t* out_ids, uint32_t* out_cnt) {
    return guarded([&] {
        if (!x || !fwd || !out_ids || !out_cnt || n == 0 || vertex >= n) throw InvalidArg("bad arguments");
        if (D < 16 || D > 2048 || (D & (D - 1))) throw InvalidArg("D must be a power of two in 16..2048");
        if (R == 0 || R > 32 || n_rev > 96) throw InvalidArg("R must be 1..32 and n_rev <= 96 (the hub path is not a unit case)");
        for (int i = 0; i < 32; ++i)
            if (fwd[i] != kInvalidNode && fwd[i] >= n) throw InvalidArg("candidate out of range");
        for (uint64_t i = 0; i < n_rev; ++i)
            if (rev[i] >= n) throw InvalidArg("candidate out of range");
        HIP_CHECK(hipSetDevice(device));
        // one row: the vertex is row 0 of a one-row layer (row_ids = {vertex}), reverse candidates are row indices of
        // a table that maps them back to the given vertex ids
        DevBuf<float> d_x(n * D), d_err(n);
        DevBuf<uint32_t> d_fwd(32), d_rev(std::max<uint64_t>(1, n_rev)), d_rows(std::max<uint64_t>(1, n_rev) + 1), d_out(32), d_cnt(1);
        DevBuf<uint64_t> d_off(2);
        HIP_CHECK(hipMemcpy(d_x.p, x, n * D * 4, hipMemcpyHostToDevice));
        if (err) HIP_CHECK(hipMemcpy(d_err.p, err, n * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_fwd.p, fwd, 128, hipMemcpyHostToDevice));
        // row 0 = the vertex itself; reverse candidate i is "row i + 1", whose vertex id is rev[i]
        std::vector<uint32_t> rows(n_rev + 1), revrows(std::max<uint64_t>(1, n_rev));
        rows[0] = vertex;
        for (uint64_t i = 0; i < n_rev; ++i) { rows[i + 1] = rev[i]; revrows[i] = (uint32_t)(i + 1); }
        HIP_CHECK(hipMemcpy(d_rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_rev.p, revrows.data(), revrows.size() * 4, hipMemcpyHostToDevice));
        const uint64_t off[2] = {0, n_rev};
        HIP_CHECK(hipMemcpy(d_off.p, off, 16, hipMemcpyHostToDevice));
        build::SelectArgs a{};
        a.x = d_x.p; a.fwd = d_fwd.p; a.rev_off = d_off.p; a.rev = d_rev.p; a.row_ids = d_rows.p; a.err = err ? d_err.p : nullptr;
        a.rows = 1; a.D = (uint32_t)D; a.R = R; a.alpha = alpha; a.tau = tau; a.alpha_max = alpha_max;
        a.out = d_out.p; a.out_cnt = d_cnt.p;
        hipLaunchKernelGGL(build::select_kernel, dim3(1), dim3(64), build::select_lds(a.D), nullptr, a);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(out_ids, d_out.p, 128, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(out_cnt, d_cnt.p, 4, hipMemcpyDeviceToHost));
    });
}

int cph_calib_hook(cph_index* h, const float* queries, const uint32_t* start, uint64_t ns, float* rec, uint32_t* rec_cnt,
                   float* dqp) {
    return guarded([&] {
        if (!h || !queries || !start || !rec || !rec_cnt || !dqp || ns == 0 || ns > 0xFFFFFFull) throw InvalidArg("bad arguments");
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        for (uint64_t i = 0; i < ns; ++i)
            if (start[i] >= h->host.n) throw InvalidArg("start vertex out of range");
        h->use_device();
        hipStream_t st = own_stream(h);
        BatchSet& s = next_set(h, st);
        stage_queries(h, s, upload_queries(h, s, queries, ns, st), ns, st);     // the search's own encoder
        HIP_CHECK(hipEventRecord(s.ev_done, st));
        s.used = true;
        DevBuf<uint32_t> d_start(ns), d_rc(ns);
        DevBuf<float> d_rec(ns * 32 * 6), d_dqp(ns);
        HIP_CHECK(hipMemcpyAsync(d_start.p, start, ns * 4, hipMemcpyHostToDevice, st));
        build::CalibArgs c{};
        c.blocks = h->d_blocks.p; c.raw = h->d_raw.p; c.L = h->L; c.n = h->host.n;
        c.queries = s.d_queries.p; c.qmasks = s.d_qmasks.p; c.qhdr = s.d_qhdr.p;
        c.start = d_start.p; c.ns = (uint32_t)ns; c.rec = d_rec.p; c.rec_cnt = d_rc.p; c.dqp_out = d_dqp.p;
        const size_t lds = (size_t)h->L.PW * 16 + (size_t)h->L.D * 8;
        const dim3 grid((uint32_t)std::min<uint64_t>(ns, (uint64_t)h->num_cus * 16));
        if (h->bits == 1) hipLaunchKernelGGL(build::calib_kernel<1>, grid, dim3(64), lds, st, c);
        else if (h->bits == 2) hipLaunchKernelGGL(build::calib_kernel<2>, grid, dim3(64), lds, st, c);
        else hipLaunchKernelGGL(build::calib_kernel<4>, grid, dim3(64), lds, st, c);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(s.ev_done, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipMemcpy(rec, d_rec.p, ns * 32 * 6 * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(rec_cnt, d_rc.p, ns * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(dqp, d_dqp.p, ns * 4, hipMemcpyDeviceToHost));
    });
}

int cph_get_vectors(cph_index* h, uint64_t first, uint64_t count, float* out) {
    return guarded([&] {
        if (!h || !out) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        if (first + count > h->host.n) throw InvalidArg("vector range out of bounds");
        for (uint64_t i = 0; i < count; ++i)
            std::memcpy(out + i * h->dim, h->host.vec(first + i), h->dim * sizeof(float));
    });
}

int cph_set_search_params(cph_index* h, uint32_t slots, uint64_t beam_capacity) {
    return guarded([&] {
        std::lock_guard<std::mutex> lk(h->mu);
        h->want_slots = slots;
        h->want_cap = beam_capacity;
    });
}

int cph_set_batch_sets(cph_index* h, uint32_t n_sets) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        if (n_sets < 1 || n_sets > (uint32_t)kMaxBatchSets) throw InvalidArg("n_sets must be 1.." + std::to_string(kMaxBatchSets));
        std::lock_guard<std::mutex> lk(h->mu);
        h->use_device();
        quiesce(h);
        for (int i = (int)n_sets; i < kMaxBatchSets; ++i) release_scratch(h->sets[i]);
        h->n_sets = (int)n_sets;
        h->last_set = (int)n_sets - 1;
    });
}

int cph_last_search_stats(cph_index* h, uint64_t out[12]) {
    return guarded([&] {
        if (!h || !out) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        for (int i = 0; i < 12; ++i) out[i] = 0;
        if (h->last_search < 0) return;
        BatchSet& s = h->sets[h->last_search];
        h->use_device();
        HIP_CHECK(hipEventSynchronize(s.ev_done));
        if (s.stats_in_hbm) {
            HIP_CHECK(hipMemcpy(s.pin_stats, s.d_stats.p, kStatWords * 8, hipMemcpyDeviceToHost));
            s.stats_in_hbm = false;
        }
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, s.ev0, s.ev1));
        for (int i = 0; i < 6; ++i) out[i] = s.pin_stats[i];
        out[6] = (uint64_t)(ms * 1000.0);
        out[7] = s.pin_stats[7];
        out[8] = s.run_slots;
        out[9] = s.run_cap;
#if !defined(CPH_PHASE_TIMERS) && !defined(CPH_TRAFFIC_STATS)
        out[10] = s.pin_stats[8];
        out[11] = s.pin_stats[9];
#endif
#if CPH_PHASE_TIMERS + 0 == 2
        fprintf(stderr, "[fine cycles] head+issue=%llu pop=%llu block_wait=%llu probe_issue+exact+nnpush=%llu estimator=%llu probe_wait=%llu mark+cand=%llu pushes+tail=%llu\n",
                s.pin_stats[8], s.pin_stats[9], s.pin_stats[10], s.pin_stats[11], s.pin_stats[12], s.pin_stats[13], s.pin_stats[14], s.pin_stats[15]);
#elif defined(CPH_PHASE_TIMERS)
        fprintf(stderr, "[phase cycles] pop=%llu load+exact+nnpush=%llu sums+epi=%llu atomic+log+stage=%llu spec_exact=%llu replay=%llu tail=%llu other=%llu\n",
                s.pin_stats[8], s.pin_stats[9], s.pin_stats[10], s.pin_stats[11], s.pin_stats[12], s.pin_stats[13], s.pin_stats[14], s.pin_stats[15]);
#endif
#ifdef CPH_TRAFFIC_STATS
        fprintf(stderr, "[traffic] hybrid_pops=%llu windows=%llu hybrid_pushes=%llu hbm_appends=%llu probe_lines=%llu sum_beam_at_pop=%llu max_beam=%llu pops_beyond_8191=%llu\n",
                s.pin_stats[8], s.pin_stats[9], s.pin_stats[10], s.pin_stats[11], s.pin_stats[12], s.pin_stats[13], s.pin_stats[14], s.pin_stats[15]);
#endif
    });
}

int cph_last_query_expansions(cph_index* h, uint32_t* out, uint64_t n) {
    return guarded([&] {
        if (!h || !out) throw InvalidArg("null argument");
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->last_search < 0 || n != h->sets[h->last_search].nq)
            throw InvalidArg("n must equal the size of the last batch");
        BatchSet& s = h->sets[h->last_search];
        h->use_device();
        HIP_CHECK(hipEventSynchronize(s.ev_done));
        HIP_CHECK(hipMemcpy(out, s.d_status.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) out[i] >>= 8;
    });
}

int cph_synchronize(cph_index* h) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        std::lock_guard<std::mutex> lk(h->mu);
        h->use_device();
        quiesce(h);
    });
}

int cph_order_queries(cph_index* h, const float* keys, uint64_t n, uint32_t* order) {
    return guarded([&] {
        if (!h || !keys || !order) throw InvalidArg("null argument");
        if (n == 0 || n > 0xFFFFFFFFull) throw InvalidArg("bad n");
        std::lock_guard<std::mutex> lk(h->mu);
        h->use_device();
        DevBuf<float> dk;
        DevBuf<uint32_t> dord;
        dk.alloc(n);
        dord.alloc(n);
        HIP_CHECK(hipMemcpy(dk.p, keys, n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, nullptr, dk.p, (uint32_t)n, dord.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(order, dord.p, n * 4, hipMemcpyDeviceToHost));
    });
}

int cph_search_batch(cph_index* h, const float* queries, uint64_t n, uint64_t k, int64_t* ids,
                     float* dist) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        if (n == 0 || k == 0) {
            // (n, 0) outputs: nothing to write; still validates the entry point like search()
            return;
        }
        if (n > 0xFFFFFFFFull || k > 0xFFFFFFFFull) throw InvalidArg("batch too large");
        if (!queries || !ids || !dist) throw InvalidArg("null argument");
        h->use_device();
        hipStream_t st = own_stream(h);
        BatchSet& s = next_set(h, st);
        if (n <= kSmallBatch && n * k <= (1u << 20)) {
            // a handful of queries: no copy commands, the kernels read the queries and write the results over PCIe
            SmallIo io = small_io(h, s, n, k);
            std::memcpy(io.h_query, queries, n * h->dim * sizeof(float));
            stage_queries(h, s, io.d_query, n, st);
            enqueue_search(h, s, (uint32_t)n, (uint32_t)k, io.d_ids, io.d_dist, st, io.d_count);
            HIP_CHECK(hipStreamSynchronize(st));
            std::memcpy(ids, io.h_ids, n * k * 8);
            std::memcpy(dist, io.h_dist, n * k * 4);
            return;
        }
        stage_queries(h, s, upload_queries(h, s, queries, n, st), n, st);
        if (s.d_ids.n < n * k) {
            if (s.used) HIP_CHECK(hipEventSynchronize(s.ev_done));
            s.d_ids.alloc(n * k);
            s.d_dist.alloc(n * k);
        }
        enqueue_search(h, s, (uint32_t)n, (uint32_t)k, s.d_ids.p, s.d_dist.p, st);
        HIP_CHECK(hipMemcpyAsync(ids, s.d_ids.p, n * k * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(dist, s.d_dist.p, n * k * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(s.ev_done, st));
        HIP_CHECK(hipStreamSynchronize(st));
    });
}

int cph_search_batch_device(cph_index* h, const float* d_queries, uint64_t n, uint64_t k,
                            int64_t* d_ids, float* d_dist, void* stream) {
    return guarded([&] {
        if (!h) throw InvalidArg("null handle");
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        if (n == 0 || k == 0) return;
        if (n > 0xFFFFFFFFull || k > 0xFFFFFFFFull) throw InvalidArg("batch too large");
        if (!d_queries || !d_ids || !d_dist) throw InvalidArg("null argument");
        h->use_device();
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        BatchSet& s = next_set(h, st);
        stage_queries(h, s, d_queries, n, st);
        enqueue_search(h, s, (uint32_t)n, (uint32_t)k, d_ids, d_dist, st);
    });
}

// One launch for up to kLeaderGroup single-query callers with the same k: queries gathered into the leader slot's
// pinned, device-mapped buffer, the copy-free small-batch path on the slot's own stream.  The handle mutex is held while
// the launch is ENQUEUED, not while it runs: the next leader's launch (other slot, other stream, its own batch set)
// overlaps this one.  Nobody waits for the launch as a whole: the kernels raise a flag per query in the pinned buffer
// once its results are visible to the host, and every caller waits for its own (wait_search_one).
constexpr size_t kFlagBytes = kLeaderGroup * 4;
struct GroupLayout {
    size_t o_ids, o_dist, o_cnt, o_q, need;
    GroupLayout(uint64_t n, uint64_t kk, uint64_t dim) {
        o_ids = kFlagBytes;
        o_dist = o_ids + n * kk * 8;
        o_cnt = o_dist + n * kk * 4;
        o_q = (o_cnt + n * 4 + 15) & ~(size_t)15;
        need = o_q + n * dim * 4;
    }
};

static void launch_search_group(cph_index* h, cph_index::LeaderSlot& ls, const std::vector<SearchReq*>& group) {
    const uint64_t n = group.size(), kk = group[0]->k;
    const uint64_t t0 = now_ns();
    std::lock_guard<std::mutex> lk(h->mu);
    const uint64_t t1 = now_ns();
    require_finalized(h);
    h->use_device();
    if (!ls.stream) HIP_CHECK(hipStreamCreateWithFlags(&ls.stream, hipStreamNonBlocking));
    const GroupLayout g(n, kk, h->dim);
    if (ls.pin_bytes < g.need) {          // (the slot is ours alone: nothing in flight reads the old buffer)
        if (ls.pin) HIP_CHECK(hipHostFree(ls.pin));
        ls.pin = nullptr; ls.pin_bytes = 0;
        const size_t bytes = std::max<size_t>(g.need * 2, 64 * 1024);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ls.pin), bytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ls.pin_dev), ls.pin, 0));
        std::memset(ls.pin, 0, kFlagBytes);
        ls.pin_bytes = bytes;
    }
    float* h_query = reinterpret_cast<float*>(ls.pin + g.o_q);
    for (uint64_t i = 0; i < n; ++i) std::memcpy(h_query + i * h->dim, group[i]->query, h->dim * sizeof(float));
    if (++ls.seq == 0) ls.seq = 1;                                     // (a flag never holds a future launch's number)
    ls.cur_n = n; ls.cur_k = kk;
    BatchSet& s = h->sets[kMaxBatchSets + (&ls - h->leaders)];         // the slot's own set: its launches are ordered by its stream
    init_set(s);
    stage_queries(h, s, reinterpret_cast<const float*>(ls.pin_dev + g.o_q), n, ls.stream);      // the encoder reads the queries over PCIe
    DoneFlags done;
    done.flags = reinterpret_cast<uint32_t*>(ls.pin_dev);
    done.seq = ls.seq;
    enqueue_search(h, s, (uint32_t)n, (uint32_t)kk, reinterpret_cast<int64_t*>(ls.pin_dev + g.o_ids), reinterpret_cast<float*>(ls.pin_dev + g.o_dist),
                   ls.stream, reinterpret_cast<uint32_t*>(ls.pin_dev + g.o_cnt), done);           // ... the search writes the results back
    CPH_TR(0, 1); CPH_TR(1, n); CPH_TR(2, t1 - t0); CPH_TR(3, now_ns() - t1);
}

// Caller `index` of the launch in flight on this slot: wait for ITS query's flag, copy ITS rows out.  A launch lasts as
// long as its longest query; a caller does not have to.  Polls the flag (pinned host memory, written by the kernel behind
// a system-scope fence), looks at the stream every thousand polls so that a launch that died cannot hang its callers,
// and gives the core away once it has spun for a while.
static void wait_search_one(cph_index* h, cph_index::LeaderSlot& ls, uint32_t index, SearchReq& r) {
    const uint64_t t0 = now_ns();
    const uint32_t seq = ls.seq;
    const uint64_t n = ls.cur_n, kk = ls.cur_k;
    const uint32_t* flag = reinterpret_cast<const uint32_t*>(ls.pin) + index;
    const auto spin_start = std::chrono::steady_clock::now();
    bool dev_set = false;
    for (uint64_t it = 1;; ++it) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((it & 1023) == 0) {
            if (!dev_set) { HIP_CHECK(hipSetDevice(h->device)); dev_set = true; }
            const hipError_t e = hipStreamQuery(ls.stream);
            if (e == hipSuccess) {                 // the stream is idle: the flag is there now, or it never will be
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                throw std::runtime_error("search launch finished without an answer for this query");
            }
            if (e != hipErrorNotReady) HIP_CHECK(e);
            if (std::chrono::steady_clock::now() - spin_start > std::chrono::microseconds(300)) std::this_thread::yield();
        } else {
            __builtin_ia32_pause();
        }
    }
    const GroupLayout g(n, kk, h->dim);
    // the reference returns every result it found (<= max(k,1)); the caller's buffers hold max(k,1) entries
    const uint32_t cnt = reinterpret_cast<const uint32_t*>(ls.pin + g.o_cnt)[index];
    std::memcpy(r.ids, reinterpret_cast<const int64_t*>(ls.pin + g.o_ids) + (size_t)index * kk, (size_t)cnt * 8);
    std::memcpy(r.dist, reinterpret_cast<const float*>(ls.pin + g.o_dist) + (size_t)index * kk, (size_t)cnt * 4);
    *r.m = cnt;
    CPH_TR(4, now_ns() - t0);
}

int cph_search(cph_index* h, const float* query, uint64_t k, int64_t* ids, float* dist,
               uint64_t* m) {
    SearchReq r{};
    const int rc = guarded([&] {
        if (!h) throw InvalidArg("null handle");
        if (!query || !ids || !dist || !m) throw InvalidArg("null argument");
        const uint64_t kk = std::max<uint64_t>(k, 1);  // api/hnsw_index.hpp:187
        if (kk > 0xFFFFFFFFull) throw InvalidArg("k too large");
        r.query = query; r.k = kk; r.ids = ids; r.dist = dist; r.m = m;
        // concurrent callers are gathered into shared launches (search_coalescer.h); the status codes it records are cph_status
        h->coal.submit(r,
                       [&](int slot, const std::vector<SearchReq*>& group) { launch_search_group(h, h->leaders[slot], group); },
                       [&](int slot, uint32_t index, SearchReq& me) { wait_search_one(h, h->leaders[slot], index, me); });
    });
    if (rc != CPH_OK) return rc;
    return r.rc == CPH_OK ? CPH_OK : fail(r.rc, r.err);
}

// ---- hooks ---------------------------------------------------------------------------
int cph_encode_query(cph_index* h, const float* query, uint8_t* lut, float* coeffs) {
    return guarded([&] {
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);  // the device encoder lives with the loaded index
        h->use_device();
        hipStream_t st = own_stream(h);
        BatchSet& s = next_set(h, st);
        stage_queries(h, s, upload_queries(h, s, query, 1, st), 1, st);
        HIP_CHECK(hipEventRecord(s.ev_done, st));
        s.used = true;
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t D = h->D, PW = h->L.PW;
        std::vector<uint32_t> masks(PW * 4);
        QueryHeader hd;
        HIP_CHECK(hipMemcpy(masks.data(), s.d_qmasks.p, PW * 16, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(&hd, s.d_qhdr.p, sizeof(hd), hipMemcpyDeviceToHost));
        std::vector<uint8_t> qu(D);
        for (uint32_t d = 0; d < D; ++d) {
            uint8_t u = 0;
            for (int j = 0; j < 4; ++j) u |= (uint8_t)(((masks[(d / 32) * 4 + j] >> (d % 32)) & 1u) << j);
            qu[d] = u;
        }
        qu_to_lut(qu.data(), D, lut);
        coeffs[0] = hd.A; coeffs[1] = hd.B; coeffs[2] = hd.C;
    });
}

int cph_entry_point(cph_index* h, const float* query, uint32_t* entry) {
    return guarded([&] {
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        h->use_device();
        hipStream_t st = own_stream(h);
        BatchSet& s = next_set(h, st);
        stage_queries(h, s, upload_queries(h, s, query, 1, st), 1, st);
        HIP_CHECK(hipEventRecord(s.ev_done, st));
        s.used = true;
        HIP_CHECK(hipStreamSynchronize(st));
        QueryHeader hd;
        HIP_CHECK(hipMemcpy(&hd, s.d_qhdr.p, sizeof(hd), hipMemcpyDeviceToHost));
        *entry = hd.entry;
    });
}

int cph_fastscan_block(cph_index* h, const uint8_t* lut, const float* qparams, uint32_t vertex,
                       float dist_qp_sq, float worst, int nn_full, uint32_t* sums, uint32_t* msb,
                       float* est, float* lower, float* lower_stage1) {
    return guarded([&] {
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        if (vertex >= h->host.n) throw InvalidArg("vertex out of range");
        h->use_device();
        const uint32_t D = h->D, PW = h->L.PW;
        std::vector<uint8_t> qu(D);
        lut_to_qu(lut, D, qu.data());
        std::vector<uint32_t> masks(PW * 4);
        qu_to_masks(qu.data(), D, masks.data());
        DevBuf<uint4> d_mask;
        DevBuf<uint32_t> d_u;
        DevBuf<float> d_f;
        d_mask.alloc(PW);
        d_u.alloc(64);
        d_f.alloc(96);
        HIP_CHECK(hipMemcpy(d_mask.p, masks.data(), PW * 16, hipMemcpyHostToDevice));
        BlockHookArgs a{};
        a.blk = h->d_blocks.p + (size_t)vertex * h->L.stride;
        a.L = h->L;
        a.qmask = d_mask.p;
        a.qp = QP{qparams[0], qparams[1], qparams[2], qparams[3], qparams[4], qparams[5], qparams[6]};
        a.dqp = dist_qp_sq;
        a.worst = worst;
        a.nn_full = nn_full;
        a.sums = d_u.p; a.msb = d_u.p + 32;
        a.est = d_f.p; a.lower = d_f.p + 32; a.lower1 = d_f.p + 64;
        CPH_LAUNCH(block_hook_kernel, h->bits, D, dim3(1), dim3(64), (size_t)PW * 16, nullptr, a);
        HIP_CHECK(hipDeviceSynchronize());
        uint32_t hu[64];
        float hf[96];
        HIP_CHECK(hipMemcpy(hu, d_u.p, sizeof(hu), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hf, d_f.p, sizeof(hf), hipMemcpyDeviceToHost));
        std::memcpy(sums, hu, 128); std::memcpy(msb, hu + 32, 128);
        std::memcpy(est, hf, 128); std::memcpy(lower, hf + 32, 128);
        std::memcpy(lower_stage1, hf + 64, 128);
    });
}

int cph_exact_l2(cph_index* h, const float* query, const uint32_t* ids, uint64_t n, float* out) {
    return guarded([&] {
        std::lock_guard<std::mutex> lk(h->mu);
        require_finalized(h);
        if (n == 0) return;
        for (uint64_t i = 0; i < n; ++i)
            if (ids[i] >= h->host.n) throw InvalidArg("id out of range");
        h->use_device();
        const uint32_t D = h->D;
        std::vector<float> buf(D, 0.0f);
        std::memcpy(buf.data(), query, h->dim * sizeof(float));
        DevBuf<float> d_q, d_out;
        DevBuf<uint32_t> d_i;
        d_q.alloc(D); d_out.alloc(n); d_i.alloc(n);
        HIP_CHECK(hipMemcpy(d_q.p, buf.data(), D * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_i.p, ids, n * 4, hipMemcpyHostToDevice));
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 7) / 8, 1024);
        hipLaunchKernelGGL(exact_l2_hook_kernel, dim3(grid), dim3(64), (size_t)D * 4, nullptr, d_q.p,
                           h->d_raw.p, h->d_norm.p, d_i.p, n, D, d_out.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(out, d_out.p, n * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"

// ---- host-only hooks -------------------------------------------------------------------------
extern "C" {

int cph_host_rewrite_index(const char* path_in, const char* path_out) {
    return guarded([&] {
        if (!path_in || !path_out) throw InvalidArg("null argument");
        // peek the header for the parameters a handle would carry
        FILE* f = std::fopen(path_in, "rb");
        if (!f) throw std::runtime_error(std::string("Cannot open file for reading: ") + path_in);
        uint8_t hdr[28];
        const size_t got = std::fread(hdr, 1, sizeof(hdr), f);
        std::fclose(f);
        if (got != sizeof(hdr)) throw std::runtime_error(std::string("Read error or truncated file: ") + path_in);
        uint32_t D, bw, dim;
        std::memcpy(&D, hdr + 12, 4); std::memcpy(&bw, hdr + 20, 4); std::memcpy(&dim, hdr + 24, 4);
        HostIndex hi;
        hi.load(path_in, D, bw, dim);
        hi.save(path_out);
    });
}

int cph_host_repack_block(uint32_t D, uint32_t bits, const uint8_t* ref_block, uint8_t* dev_block,
                          uint64_t* dev_bytes, uint8_t* ref_roundtrip) {
    return guarded([&] {
        if (bits != 1 && bits != 2 && bits != 4) throw InvalidArg("bits must be 1, 2 or 4");
        if (D < 16 || D > 2048 || (D & (D - 1))) throw InvalidArg("D must be a power of two in 16..2048");
        const DevLayout L = make_dev_layout(D, bits);
        const RefLayout RL = make_ref_layout(D, bits);
        repack_ref_to_dev(ref_block, RL, L, dev_block);
        if (dev_bytes) *dev_bytes = L.stride;
        if (ref_roundtrip) repack_dev_to_ref(dev_block, L, RL, ref_roundtrip);
    });
}

int cph_host_encode_query(uint64_t dim, const float* query, uint8_t* lut, float* coeffs, uint32_t* masks) {
    return guarded([&] {
        const size_t D = std::max<size_t>(16, next_pow2(dim));
        if (dim == 0 || D > 2048) throw InvalidArg("unsupported dimension");
        Rotation rot;
        rot.init(D, 42);
        std::vector<float> buf(D, 0.0f);
        std::memcpy(buf.data(), query, dim * sizeof(float));
        EncodedQuery eq;
        encode_query(rot, buf.data(), eq);
        if (lut) qu_to_lut(eq.qu.data(), D, lut);
        if (coeffs) { coeffs[0] = eq.A; coeffs[1] = eq.B; coeffs[2] = eq.C; }
        if (masks) qu_to_masks(eq.qu.data(), D, masks);
    });
}

}  // extern "C"

// ---- streaming FastScan benchmark object ---------------------------------------------------
struct cph_stream {
    int device = 0;
    DevLayout L{};
    RefLayout RL{};
    uint64_t n_blocks = 0;
    DevBuf<uint8_t> d_blocks;
    DevBuf<uint4> d_mask;
    DevBuf<float> d_sink;
    std::vector<uint8_t> lut;
    QP qp{};
    float dqp = 0.0f;
    int num_cus = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}

// Fills device-layout blocks with seeded random codes and consistent aux data.
__global__ __launch_bounds__(64) void stream_fill_kernel(uint8_t* blocks, uint64_t n_blocks,
                                                         DevLayout L, uint64_t seed) {
    const int lane = threadIdx.x;
    const int i = lane & 31, h = lane >> 5;
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        uint8_t* blk = blocks + b * L.stride;
        uint32_t pc[4] = {0, 0, 0, 0};  // popcount per plane of neighbour i (this lane's share)
        const uint32_t T = L.BW * L.PW;
        const uint32_t valid_bits = L.D >= 32 ? 32 : L.D;
        const uint32_t vmask = valid_bits == 32 ? 0xFFFFFFFFu : ((1u << valid_bits) - 1u);
        for (uint32_t t = h; t < T; t += 2) {  // the two lanes of a neighbour split its dwords
            const uint32_t plane = t / L.PW;
            uint32_t v = mix32(seed ^ (b * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)(t * 32 + i) << 20)) & vmask;
            size_t off;
            if (L.wide) {
                const uint32_t ck = t / 4, e = t % 4;
                const uint32_t hh = (L.NH == 2) ? ck / L.CPL : 0;
                const uint32_t k = (L.NH == 2) ? ck % L.CPL : ck;
                off = ((size_t)(k * L.NH * 32 + hh * 32 + i) * 16 + e * 4);
            } else {
                off = ((size_t)t * 32 + i) * 4;
            }
            *reinterpret_cast<uint32_t*>(blk + off) = v;
            const uint32_t c = __popc(v);
            if (plane == 0) pc[0] += c; else if (plane == 1) pc[1] += c;
            else if (plane == 2) pc[2] += c; else pc[3] += c;
        }
        for (int p = 0; p < 4; ++p) pc[p] += __shfl_xor(pc[p], 32);
        if (lane < 32) {
            uint32_t wp = 0;
            for (uint32_t p = 0; p < L.BW; ++p) wp += pc[p] << (L.BW - 1 - p);
            const uint32_t r0 = mix32(seed ^ (b * 1315423911ULL) ^ (0x1000 + i));
            const uint32_t r1 = mix32(seed ^ (b * 2654435761ULL) ^ (0x2000 + i));
            const uint32_t r2 = mix32(seed ^ (b * 40503ULL) ^ (0x3000 + i));
            float nop = 1.0f + 19.0f * (r0 >> 8) * (1.0f / 16777216.0f);
            float ipqo = 0.5f + 0.4f * (r1 >> 8) * (1.0f / 16777216.0f);
            float ipcp = -0.5f + (r2 >> 8) * (1.0f / 16777216.0f);
            uint4 aux = make_uint4(__float_as_uint(nop), __float_as_uint(ipqo), __float_as_uint(ipcp),
                                   (pc[0] & 0xFFFFu) | ((wp & 0xFFFFu) << 16));
            reinterpret_cast<uint4*>(blk + L.aux_off)[i] = aux;
            reinterpret_cast<uint32_t*>(blk + L.ids_off)[i] = mix32(seed ^ b ^ ((uint64_t)i << 40)) >> 1;
            if (lane == 0) *reinterpret_cast<uint32_t*>(blk + L.count_off) = 32u;
        }
    }
}

void stream_launch(cph_stream* s, float* out_est, float* out_lower, uint64_t first, uint64_t count,
                   hipStream_t st) {
    StreamArgs a{};
    a.blocks = s->d_blocks.p;
    a.n_blocks = s->n_blocks;
    a.L = s->L;
    a.qmask = s->d_mask.p;
    a.qp = s->qp;
    a.dqp = s->dqp;
    a.sink = s->d_sink.p;
    a.out_est = out_est;
    a.out_lower = out_lower;
    a.first = first;
    a.count = count;
    static const int mult = getenv("CPH_STREAM_GRID_MULT") ? atoi(getenv("CPH_STREAM_GRID_MULT")) : 32;
    const uint32_t grid = (uint32_t)s->num_cus * mult;
    static const bool pair = !(getenv("CPH_STREAM_PAIR") && atoi(getenv("CPH_STREAM_PAIR")) == 0);
    if (pair && s->L.D == 128 && s->L.BW <= 2) {
        // narrow codes: two blocks per wave iteration, one per lane half (device_stream.h)
        if (s->L.BW == 1) hipLaunchKernelGGL((fastscan_stream_pair_kernel<1>), dim3(grid), dim3(256), 64, st, a);
        else hipLaunchKernelGGL((fastscan_stream_pair_kernel<2>), dim3(grid), dim3(256), 64, st, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    // (the instantiation with a compile-time D = 1024 holds a whole 16-KB block in registers twice over and loses
    // its occupancy: 0.63 against 0.75 of peak for the runtime-D loop -- the stream kernel uses the latter there)
    const uint32_t dsel = s->L.D == 128 ? 128u : 0u;
    CPH_LAUNCH(fastscan_stream_kernel, s->L.BW, dsel, dim3(grid), dim3(256), (size_t)s->L.PW * 16, st, a);
}

}  // namespace

extern "C" {

int cph_fastscan_stream_create(int device, uint32_t D, uint32_t bits, uint64_t n_blocks,
                               uint64_t seed, cph_stream** out, uint64_t* block_bytes) {
    return guarded([&] {
        *out = nullptr;
        if (bits != 1 && bits != 2 && bits != 4) throw InvalidArg("bits must be 1, 2 or 4");
        if (D < 16 || D > 2048 || (D & (D - 1))) throw InvalidArg("D must be a power of two in 16..2048");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw std::runtime_error("No HIP device available: the MI355X path has no CPU fallback.");
        HIP_CHECK(hipSetDevice(device));
        auto s = std::unique_ptr<cph_stream>(new cph_stream());
        s->device = device;
        s->L = make_dev_layout(D, bits);
        s->RL = make_ref_layout(D, bits);
        s->n_blocks = n_blocks;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount;
        s->d_blocks.alloc(n_blocks * s->L.stride + 64);
        s->d_mask.alloc(s->L.PW);
        s->d_sink.alloc((size_t)s->num_cus * 64 * 4);
        HIP_CHECK(hipMemset(s->d_sink.p, 0, s->d_sink.n * sizeof(float)));
        hipLaunchKernelGGL(stream_fill_kernel, dim3(s->num_cus * 16), dim3(64), 0, nullptr,
                           s->d_blocks.p, n_blocks, s->L, seed);
        HIP_CHECK(hipGetLastError());
        // one seeded query: random rotated vector -> 4-bit scalars -> masks / LUT / coefficients
        std::mt19937_64 rng(seed * 7919 + 13);
        std::normal_distribution<float> nd(0.0f, 1.0f);
        std::vector<float> q(D);
        for (auto& x : q) x = nd(rng);
        Rotation rot;
        rot.init(D, 42);
        EncodedQuery eq;
        encode_query(rot, q.data(), eq);
        s->lut.resize(D / 4 * 16);
        qu_to_lut(eq.qu.data(), D, s->lut.data());
        std::vector<uint32_t> masks(s->L.PW * 4);
        qu_to_masks(eq.qu.data(), D, masks.data());
        HIP_CHECK(hipMemcpy(s->d_mask.p, masks.data(), masks.size() * 4, hipMemcpyHostToDevice));
        s->qp = QP{eq.A, eq.B, eq.C, 1.0f, 0.0f, 0.0f, 0.05f};
        s->dqp = 140.0f;
        HIP_CHECK(hipEventCreate(&s->ev0));
        HIP_CHECK(hipEventCreate(&s->ev1));
        HIP_CHECK(hipDeviceSynchronize());
        if (block_bytes) *block_bytes = s->L.stride;
        *out = s.release();
    });
}

int cph_fastscan_stream_run(cph_stream* s, int reps, double* avg_ms, double* checksum) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(s->device));
        if (reps < 1) reps = 1;
        hipStream_t st = nullptr;
        HIP_CHECK(hipEventRecord(s->ev0, st));
        for (int r = 0; r < reps; ++r) stream_launch(s, nullptr, nullptr, 0, 0, st);
        HIP_CHECK(hipEventRecord(s->ev1, st));
        HIP_CHECK(hipEventSynchronize(s->ev1));
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        if (avg_ms) *avg_ms = (double)ms / reps;
        if (checksum) {
            std::vector<float> sink(s->d_sink.n);
            HIP_CHECK(hipMemcpy(sink.data(), s->d_sink.p, sink.size() * 4, hipMemcpyDeviceToHost));
            double t = 0.0;
            for (float x : sink) t += x;
            *checksum = t;
        }
    });
}

int cph_fastscan_stream_export(cph_stream* s, uint64_t first, uint64_t count, uint8_t* ref_blocks,
                               uint8_t* lut, float* qparams, float* dist_qp_sq) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(s->device));
        if (first + count > s->n_blocks) throw InvalidArg("block range out of bounds");
        if (count && ref_blocks) {
            std::vector<uint8_t> dev(count * s->L.stride);
            HIP_CHECK(hipMemcpy(dev.data(), s->d_blocks.p + first * s->L.stride, dev.size(),
                                hipMemcpyDeviceToHost));
            parallel_for(count, 1024, [&](size_t lo, size_t hi) {
                for (size_t b = lo; b < hi; ++b)
                    repack_dev_to_ref(&dev[b * s->L.stride], s->L, s->RL, ref_blocks + b * s->RL.nb_bytes);
            });
        }
        if (lut) std::memcpy(lut, s->lut.data(), s->lut.size());
        if (qparams) {
            qparams[0] = s->qp.A; qparams[1] = s->qp.B; qparams[2] = s->qp.C;
            qparams[3] = s->qp.affine_a; qparams[4] = s->qp.affine_b; qparams[5] = s->qp.floor;
            qparams[6] = s->qp.slack;
        }
        if (dist_qp_sq) *dist_qp_sq = s->dqp;
    });
}

int cph_fastscan_stream_eval(cph_stream* s, uint64_t first, uint64_t count, float* est, float* lower) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(s->device));
        if (first + count > s->n_blocks) throw InvalidArg("block range out of bounds");
        if (!count) return;
        DevBuf<float> d_e, d_l;
        d_e.alloc(count * 32);
        d_l.alloc(count * 32);
        stream_launch(s, d_e.p, d_l.p, first, count, nullptr);
        HIP_CHECK(hipMemcpy(est, d_e.p, count * 128, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(lower, d_l.p, count * 128, hipMemcpyDeviceToHost));
    });
}

int cph_fastscan_stream_destroy(cph_stream* s) {
    return guarded([&] {
        if (!s) return;
        (void)hipSetDevice(s->device);
        if (s->ev0) (void)hipEventDestroy(s->ev0);
        if (s->ev1) (void)hipEventDestroy(s->ev1);
        delete s;
    });
}

}  // extern "C"
