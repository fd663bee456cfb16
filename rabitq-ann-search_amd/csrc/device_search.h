// device_search.h — persistent layer-0 beam search, one wavefront per query.
//
// GPU counterpart of rabitq_search::search (search/rabitq_search.hpp:60-277) and its
// helpers BoundedMaxHeap (:17-49) / BeamEntry (:53-58).  Results are bit-identical to the
// reference: the data-parallel parts (FastScan block, estimated-set probes, speculative
// exact L2 of every lane whose estimate beats the threshold at loop entry) run on all 64
// lanes; the beam's pop_heap is executed by the whole wave (beam_pop_wave); the
// order-dependent decisions of the 32-neighbour loop (:218-273) are taken by all lanes at
// once when no neighbour is reranked, and replayed by lane 0 in neighbour order otherwise --
// always with the reference's heap algorithms (std::push_heap / pop_heap / sort_heap element
// movement, so ties break identically).
//
// Observations that shape the kernel:
//  * every node enters the beam at most once (each push is guarded by the estimated-set
//    test), so the reference's second "visited" table is never observable — only the
//    estimated set is kept (a per-slot bitmap, cleared by un-marking the logged ids);
//  * in a non-warm-up expansion the result-heap threshold only decreases, so
//    {est < worst at loop entry} is a superset of the lanes the serial loop will rerank;
//  * an expansion whose neighbours are all estimated already has no observable effect beyond
//    the result-heap push of the popped vertex: its FastScan arithmetic is skipped;
//  * what bounds it depends on the workload: on the SIFT-like benchmark a launch lasts as long as its
//    longest query and the steady state shares the CU's one scalar unit between 24 waves; on the recall-gate workloads
//    (beams of thousands of entries, no id locality) it is HBM-bound on the bytes it actually moves.  Hence: no
//    software prefetch, the probe is a load (atomics only for new ids), shuffles are DPP / v_permlane32_swap, and
//    everything a lane can do without an exec-mask round trip on the scalar unit is done that way.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

#include "cph_core.h"
#include "device_fastscan.h"

namespace cph {

struct Result {
    uint32_t id;
    float dist;
};

struct SearchArgs {
    // index
    const uint8_t* blocks;
    const float* raw;       // [n][D]
    const float* norm_sq;   // [n]
    uint64_t n;
    DevLayout L;
    uint32_t flags;         // bit 0: some vertex repeats a neighbour id; bit 1: some list's length is not a multiple of 8
    // encoded queries
    const float* queries;   // [nq][D] zero-padded
    const uint4* qmasks;    // [nq][PW]
    const QueryHeader* qhdr;
    const uint32_t* todo;   // optional: query indices to run (launch order / re-run list), else 0..nq-1
    uint32_t nq;
    const uint32_t* nq_dev; // optional: the batch size lives in device memory (the overflow re-run's list length)
    uint32_t k;
    SearchConsts sc;
    // work queue + per-slot scratch
    uint32_t* counter;
    uint64_t cap;           // per-slot log/heap capacity
    uint64_t bm_words;      // per-slot bitmap words
    uint32_t* bitmaps;
    uint32_t* beam_pages;   // beam heap levels 8..12: kBeamPagesDwords per slot (beam_page_off)
    uint32_t* beam_tail;    // beam heap levels 13+: beam_tail_dwords(cap) per slot (beam_tail_off)
    uint32_t* log_ids;      // every newly estimated id, in discovery order (for un-marking)
    // outputs
    int64_t* out_ids;       // [nq][k]
    float* out_dist;        // [nq][k]
    uint32_t* out_count;    // [nq]
    uint32_t* status;       // [nq]
    unsigned long long* stats;  // [16]
    // queries whose beam or id log outgrew `cap` are appended here and answered by the full-capacity
    // re-run launch that follows on the same stream (no host round trip)
    uint32_t* redo;         // [nq] or null
    uint32_t* redo_count;
    // optional (coalesced cph_search callers): done_flags[qi] = done_seq once query qi's results are visible to the HOST --
    // the outputs then live in pinned, device-mapped memory and every caller waits for its own query only
    uint32_t* done_flags;
    uint32_t done_seq;
};

// Kernel arguments that are touched once per query (work queue, encoded-query arrays, outputs, statistics) are read from
// the kernarg segment where they are used instead of living in scalar registers across the expansion loop: the
// pointer is laundered through an empty asm so that the loads cannot be hoisted.  (With every argument resident the
// allocator spilled ~50 SGPRs to VGPR lanes -- 133 v_readlane reloads in the <4,128> instantiation; now 19 and 27.
// The reloads were mostly off the expansion loop's usual path: the kernel time did not change measurably.)
template <class T>
__device__ __forceinline__ T cold_arg(size_t offset) {
    typedef __attribute__((address_space(4))) const unsigned char kbyte;
    kbyte* p = (kbyte*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(__attribute__((address_space(4))) const T*)(p + offset);
}
#define CPH_COLD(field) cold_arg<decltype(SearchArgs::field)>(offsetof(SearchArgs, field))

// ---- libstdc++-compatible binary heaps ------------------------------------------------
// Result heap: max-heap on dist (std::less via SearchResult::operator<, core/types.hpp:16).
__device__ __forceinline__ void nn_sift_up(Result* h, uint32_t hole, uint32_t top, Result v) {
    while (hole > top) {
        uint32_t p = (hole - 1) >> 1;
        Result pv = h[p];
        if (!(pv.dist < v.dist)) break;
        h[hole] = pv;
        hole = p;
    }
    h[hole] = v;
}
__device__ __forceinline__ void nn_adjust(Result* h, uint32_t hole, uint32_t len, Result v) {
    const uint32_t top = hole;
    uint32_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        Result r = h[child], l = h[child - 1];
        if (r.dist < l.dist) { --child; r = l; }
        h[hole] = r;
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    nn_sift_up(h, hole, top, v);
}
// BoundedMaxHeap::push, rabitq_search.hpp:26-35
__device__ __forceinline__ void nn_push(Result* h, uint32_t& size, uint32_t k, Result r) {
    if (size < k) {
        nn_sift_up(h, size, 0, r);
        ++size;
    } else if (r.dist < h[0].dist) {
        if (size > 1) {
            Result v = h[size - 1];
            nn_adjust(h, 0, size - 1, v);   // pop_heap
        }
        nn_sift_up(h, size - 1, 0, r);      // back() = r; push_heap
    }
}
__device__ __forceinline__ void nn_sort(Result* h, uint32_t size) {  // std::sort_heap
    while (size > 1) {
        Result v = h[size - 1];
        h[size - 1] = h[0];
        nn_adjust(h, 0, size - 1, v);
        --size;
    }
}

// Beam: std::priority_queue<BeamEntry, vector, greater> — a heap whose comparator is
// "a.est > b.est" (rabitq_search.hpp:57,79-80).  The LOGICAL heap -- which entry sits at which heap index after
// every operation -- is libstdc++'s, move for move; where a heap index lives physically is ours to choose:
//
//   heap indices 0..254      (levels 0..7)   LDS, 16 B per entry: the levels every pop walks through
//   heap indices 255..8190   (levels 8..12)  HBM, PAGES: the 62 descendants of one level-7 node over the next five
//                                            levels are one contiguous block of 64 x 12 B {est, lower, id} (six
//                                            128-byte lines; entry r = the node's 1-based index inside that subtree,
//                                            entries 0 and 1 unused) -- a B-heap page.  A pop's window (the 62
//                                            candidates of its next five levels) is one coalesced 744-byte read that
//                                            brings the WHOLE entries, so the moves along the path need no second
//                                            round trip; a push finds all its ancestors of levels 8..11 in the first
//                                            three lines of one page
//   heap indices >= 8191     (levels 13+)    HBM, 12 B per entry in heap order behind the 128 pages
//
// (Round 2 kept the HBM part as 16-byte entries in heap order: a window's 62 keys were 248 useful bytes spread over
// nine to twelve lines in five places, and the moves re-read the path's entries in a second, dependent round trip.)
constexpr uint32_t kBeamLds = 255;                    // 8 full levels, 16 B per entry
constexpr uint32_t kBeamPaged = 8191;                 // heap indices below this and >= kBeamLds live in pages
constexpr uint32_t kPageDwords = 192;                 // 64 entries x 3 dwords
constexpr uint32_t kBeamPagesDwords = 128 * kPageDwords;   // one slot's pages: 96 KB
// The pages and the tails are two arrays: [slot][pages] is what spilled beams actually touch (a beam of the recall
// workload holds ~5,000 entries, 13,800 at most), dense -- 96 KB per slot, 600 MB for 6,144 slots -- instead of 96 KB
// at the head of every slot's (cap x 12 B)-sized region, i.e. one 2-MB translation per slot; the tails (heap indices
// >= 8191, sized by `cap`) are rarely touched.
__host__ __device__ inline size_t beam_tail_dwords(uint64_t cap) {   // per slot, a multiple of 32
    const size_t d = 3 * (size_t)(cap > kBeamPaged ? cap - kBeamPaged : 0) + 4;
    return (d + 31) & ~(size_t)31;
}

struct BeamEntry {
    float est, lower;
    uint32_t id;
};

// The lane id as a value the optimiser cannot see through.  Everything a routine derives from the lane id alone is
// loop-invariant, so the compiler computes it once per kernel and keeps it in a VGPR across the whole expansion loop -- for
// the spilled-beam routines that is a dozen registers the usual expansion never uses, i.e. spills, and a spill reload inside
// the loop is a vector-memory operation in the middle of the round trips the loop overlaps (it waits for all of them:
// scripts/loop_reloads.py counts them, the count has to stay at zero on the usual path).  Routines off the usual path
// start from a laundered copy: two or three cheap instructions recomputed where they are used.
__device__ __forceinline__ int opaque_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}

// The LDS part is addressed through address-space-3 pointers so that every access is a ds_*
// instruction (a generic pointer would make them flat_* accesses, which also wait on vmcnt).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// dword offset of heap index i (kBeamLds <= i < kBeamPaged) inside the slot's pages
__device__ __forceinline__ uint32_t beam_page_off(uint32_t i) {
    const uint32_t hp = i + 1;
    const uint32_t l = ((31u - (uint32_t)__builtin_clz(hp)) - 7u) & 7u;       // levels below the LDS part: 1..5
    const uint32_t r = (hp & ((1u << l) - 1u)) | (1u << l);                   // index inside the level-7 node's subtree
    return ((hp >> l) - 128u) * kPageDwords + 3u * r;
}
// dword offset of heap index i >= kBeamPaged inside the slot's tail
__device__ __forceinline__ uint32_t beam_tail_off(uint32_t i) { return 3u * (i - kBeamPaged); }

struct Beam {
    typedef uint4 E;
    lds_u32x4* l;   // LDS, [kBeamLds + 1]   {est bits, lower bits, id, -}
    uint32_t* gp;   // global, the slot's pages (beam_page_off)
    uint32_t* gt;   // global, the slot's tail (beam_tail_off)
    // the heap's comparator (std::greater on est: a min-heap) and the key of an entry
    static __device__ __forceinline__ bool before(float a, float b) { return a > b; }
    static __device__ __forceinline__ float key_of(uint4 e) { return __uint_as_float(e.x); }
    __device__ __forceinline__ uint4 lds(uint32_t i) const {
        const u32x4 t = l[i];
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    __device__ __forceinline__ void lds_put(uint32_t i, uint4 v) const {
        u32x4 t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        l[i] = t;
    }
    __device__ __forceinline__ float lds_key(uint32_t i) const {
        return reinterpret_cast<lds_f32*>(l)[4 * i];
    }
    // one 12-byte entry at a dword offset of the pages / of the tail
    static __device__ __forceinline__ uint4 load3(const uint32_t* p) {
        const u32x3 t = *reinterpret_cast<const u32x3 __attribute__((aligned(4)))*>(p);
        return make_uint4(t.x, t.y, t.z, 0u);
    }
    static __device__ __forceinline__ void store3(uint32_t* p, uint4 v) {
        u32x3 t;
        t.x = v.x; t.y = v.y; t.z = v.z;
        *reinterpret_cast<u32x3 __attribute__((aligned(4)))*>(p) = t;
    }
    __device__ __forceinline__ uint4 pload(uint32_t off) const { return load3(gp + off); }
    __device__ __forceinline__ void pstore(uint32_t off, uint4 v) const { store3(gp + off, v); }
    __device__ __forceinline__ uint4 tload(uint32_t off) const { return load3(gt + off); }
    __device__ __forceinline__ void tstore(uint32_t off, uint4 v) const { store3(gt + off, v); }
    __device__ __forceinline__ uint32_t* spill_ptr(uint32_t i) const {       // i >= kBeamLds
        return i < kBeamPaged ? gp + beam_page_off(i) : gt + beam_tail_off(i);
    }
    __device__ __forceinline__ uint4 raw(uint32_t i) const {
        uint4 v;
        if (i < kBeamLds) v = lds(i); else v = load3(spill_ptr(i));
        return v;
    }
    __device__ __forceinline__ void put(uint32_t i, uint4 v) const {
        if (i < kBeamLds) lds_put(i, v); else store3(spill_ptr(i), v);
    }
    __device__ __forceinline__ float raw_key(uint32_t i) const {
        float k;
        if (i < kBeamLds) k = lds_key(i); else k = __uint_as_float(*spill_ptr(i));
        return k;
    }
    __device__ __forceinline__ BeamEntry get(uint32_t i) const {
        const uint4 v = raw(i);
        return BeamEntry{__uint_as_float(v.x), __uint_as_float(v.y), v.z};
    }
};

// The result heap (BoundedMaxHeap, rabitq_search.hpp:17-49) as the wave-parallel heap routines see it: k entries
// {id, dist} of 8 bytes in LDS, comparator std::less on dist (a max-heap).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
struct NnLds {
    typedef uint2 E;
    lds_u32x2* l;
    static __device__ __forceinline__ bool before(float a, float b) { return a < b; }
    static __device__ __forceinline__ float key_of(uint2 e) { return __uint_as_float(e.y); }
    __device__ __forceinline__ uint2 lds(uint32_t i) const {
        const u32x2 t = l[i];
        return make_uint2(t.x, t.y);
    }
    __device__ __forceinline__ void lds_put(uint32_t i, uint2 v) const {
        u32x2 t;
        t.x = v.x; t.y = v.y;
        l[i] = t;
    }
    __device__ __forceinline__ float lds_key(uint32_t i) const {
        return reinterpret_cast<lds_f32*>(l)[2 * i + 1];
    }
};
constexpr uint32_t kWaveHeapMax = 256;   // largest heap the wave-parallel pop handles (two ballot masks)

__device__ __forceinline__ uint4 beam_pack(const BeamEntry& e) {
    return make_uint4(__float_as_uint(e.est), __float_as_uint(e.lower), e.id, 0u);
}

// std::pop_heap of a beam that lives entirely in LDS (size <= kBeamLds), executed by the whole
// wave instead of one lane.  The element movement is libstdc++'s __adjust_heap + __push_heap
// (same comparisons, same operand order, so ties and NaNs resolve identically); only the
// schedule differs: every internal node's "which child moves up" comparison is evaluated at
// once (lane j <-> nodes j and j+64, one ballot each), the root-to-leaf path is then walked on
// wave-uniform bit masks with no memory access, and the moves along the path are one parallel
// LDS read plus one parallel write.  Two LDS round trips instead of one dependent round trip
// per heap level.
template <class H>
__device__ __forceinline__ void heap_pop_wave(const H& h, uint32_t size, int lane_in) {
#ifdef CPH_LAUNDER_HOT      // (A/B switch: the routines of the usual path recompute their lane arithmetic as well -- fewer VGPRs, more instructions)
    const int lane = opaque_lane(lane_in);
#else
    const int lane = lane_in;
#endif
    const uint32_t len = size - 1;                // heap length once the last element is taken out
    const typename H::E v = h.lds(len);            // the value __adjust_heap re-inserts
    const uint32_t nint = (len - 1) >> 1;         // nodes j < nint have both children below len
    // true: the left child moves up.  (Read by every lane -- entries up to index 128 / 256 exist in LDS whatever the heap
    // holds -- and masked afterwards: a predicated read costs an exec-mask save / restore, and the scalar unit is what
    // this kernel runs out of.)
    const bool b0 = H::before(h.lds_key(2 * lane + 2), h.lds_key(2 * lane + 1)) && (uint32_t)lane < nint;
    bool b1 = false;
    const unsigned long long m0 = __ballot(b0);
    unsigned long long m1 = 0;
    if (nint > 64) {
        if ((uint32_t)lane + 64 < nint) b1 = H::before(h.lds_key(2 * lane + 130), h.lds_key(2 * lane + 129));
        m1 = __ballot(b1);
    }
    // Walk on hp = hole + 1: taking the left child doubles it, the right child doubles it and adds 1,
    // so after d steps hp is the binary string "1 r0 r1 .. r(d-1)" of the choices and every node on
    // the path is a prefix of it -- p_t = (hp >> (d - t)) - 1.  The loop is scalar-only; each lane
    // then derives its own pair of path positions with two shifts.
    uint32_t hp = 1, d = 0;
    if (nint >= 1) {
        // Scalar walk, four instructions per level: with the mask shifted up by one, node hp - 1's bit sits at position
        // hp; s_bitcmp0 puts "the right child moves up" into SCC and s_addc turns hp into 2 hp + SCC.  Nodes 63..126
        // (hp = 64..127: at most the last step of a heap of up to 255 entries) take their bit from a second mask at
        // position hp - 64, which is hp's low six bits -- all the instruction looks at.  The depth is recovered from
        // hp's leading one afterwards.
        const unsigned long long ms0 = m0 << 1;
        const uint32_t lim = nint < 63u ? nint : 63u;
        asm volatile(
            "1:\n\t"
            "s_bitcmp0_b64 %[m], %[hp]\n\t"
            "s_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_cmp_le_u32 %[hp], %[n]\n\t"
            "s_cbranch_scc1 1b"
            : [hp] "+s"(hp)
            : [m] "s"(ms0), [n] "s"(lim)
            : "scc");
        if (hp <= nint) {                          // 64 <= hp <= 126
            const unsigned long long ms1 = (m1 << 1) | (m0 >> 63);
            asm volatile(
                "s_bitcmp0_b64 %[m], %[hp]\n\t"
                "s_addc_u32 %[hp], %[hp], %[hp]"
                : [hp] "+s"(hp)
                : [m] "s"(ms1)
                : "scc");
        }
        d = 31u - (uint32_t)__builtin_clz(hp);
    }
    if (2 * hp == len) {                           // a last node with a left child only: len even, hole == (len - 2) / 2
        hp = 2 * hp;
        ++d;
    }
    const uint32_t hole = hp - 1;                                   // p_d, where the walk ends
    const uint32_t t = (uint32_t)lane < d ? (uint32_t)lane : 0u;
    const uint32_t my_dst = (hp >> (d - t)) - 1;                    // p_t: lane t moves entry p_{t+1} into it
    const uint32_t my_src = (hp >> (d - t - ((uint32_t)lane < d ? 1u : 0u))) - 1;   // p_{t+1}
    // __push_heap: parent (now e_t) > v -> parent moves down.  Lanes >= d read p_1 (t = 0) and are masked out below.
    const typename H::E e = h.lds(my_src);
    const bool c = H::before(H::key_of(e), H::key_of(v));
    const unsigned long long stay = ~__ballot(c) & ((1ull << d) - 1ull);
    const uint32_t fin = stay ? 64u - (uint32_t)__builtin_clzll(stay) : 0u;   // v ends at p_fin
    if ((uint32_t)lane < fin) h.lds_put(my_dst, e);
    // p_fin = (hp >> (d - fin)) - 1 for every fin (= the hole when fin == d): one writer, one wave-uniform address, behind
    // the moves in program order (lane 0's own move, if any, went to p_0 != p_fin)
    if (lane == 0) h.lds_put((hp >> (d - fin)) - 1, v);
    (void)hole;
}

// std::push_heap of `v` at index `hole` (= the heap's size before the push) for a beam that stays inside the LDS
// levels, executed by the whole wave.  libstdc++'s __push_heap moves the hole up while the parent compares greater
// than the value; the ancestors of a leaf are known from its index alone -- p_t = ((hole + 1) >> t) - 1 -- so lane
// t - 1 reads ancestor p_t, all comparisons are made at once, the number of leading "parent moves down" answers m is
// a ballot and a count, and the moves are one parallel write: ancestors 1..m each drop one step along the path, the
// value lands on p_m.  One LDS read round trip and one write instead of one dependent round trip per heap level.
template <class H>
__device__ __forceinline__ void heap_push_wave(const H& h, uint32_t hole, typename H::E v, int lane_in) {
#ifdef CPH_LAUNDER_HOT
    const int lane = opaque_lane(lane_in);
#else
    const int lane = lane_in;
#endif
    const uint32_t hp = hole + 1;
    const uint32_t depth = 31u - (uint32_t)__builtin_clz(hp);          // ancestors of the leaf (0 for the root)
    const uint32_t t = (uint32_t)lane + 1;                              // lane t - 1 looks at ancestor p_t
    const bool on = t <= depth;
    const uint32_t pt = on ? (hp >> t) - 1 : 0u;
    const typename H::E e = h.lds(pt);                                 // (lanes past the root read it again and are masked)
    const bool down = on && H::before(H::key_of(e), H::key_of(v));
    const unsigned long long stop = ~__ballot(down);                   // first ancestor that stays (bits >= depth are set)
    const uint32_t m = (uint32_t)__builtin_ctzll(stop);                 // ancestors p_1..p_m move down
    if (t <= m) h.lds_put((hp >> (t - 1)) - 1, e);                      // p_t's entry to p_(t-1), p_0 = the leaf
    if ((uint32_t)lane == m) h.lds_put((hp >> m) - 1, v);              // (m <= depth <= 8 < 64: that lane exists)
}

// ---- the same two operations for a beam that has outgrown its LDS levels (entries kBeamLds.. live in HBM) --------
// A lane-0 sift is one dependent HBM round trip per heap level below the LDS part -- 5 to 10 of them per pop once a beam
// holds thousands of entries (the recall >= 0.95 workload keeps ~5,000 on average).  The moves stay libstdc++'s; the
// schedule becomes: (1) the 7 LDS levels are walked on ballot masks as in heap_pop_wave; (2) below them the wave reads
// the PAGE of the level-7 node it arrived at -- the 62 descendants of the next five levels, one whole entry per lane,
// one coalesced read -- decides every "which child moves up" of the window with one ballot and walks five levels on
// the mask; (3) the path is then a bit string, every lane knows its pair of positions, and the entries of levels 8..12
// that move are already in registers (handed to the path's lanes by ds_bpermute): one HBM round trip per pop, followed
// only by the stores.  Beams beyond 8,191 entries (6 % of that workload's pops) continue with windows of keys over the
// heap-ordered tail and fetch those levels' path entries in a second round trip.
// History: lane-0 sifts 334 ms per 10,000 queries of that workload; windows of keys over 16-byte entries in heap
// order + a dependent read of the path (round 2) 216 ms; pages (round 3): see DESIGN.md section 6.
__device__ __forceinline__ void beam_pop_hybrid(const Beam& h, uint32_t size, int lane_in) {
    const int lane = opaque_lane(lane_in);
    const uint32_t len = size - 1;                 // >= kBeamLds: the last element lives in HBM
    const uint4 v = Beam::load3(h.spill_ptr(len));       // the value __adjust_heap re-inserts (same address in every lane)
    // LDS levels 0..6: nodes 0..126, both children always inside the LDS part
    const bool b0 = Beam::before(h.lds_key(2 * lane + 2), h.lds_key(2 * lane + 1));
    const unsigned long long m0 = __ballot(b0);
    bool b1 = false;
    if (lane < 63) b1 = Beam::before(h.lds_key(2 * lane + 130), h.lds_key(2 * lane + 129));
    const unsigned long long m1 = __ballot(b1);
    uint32_t hp = 1, d = 7;                        // hp = hole + 1, the path as a bit string (see heap_pop_wave)
    {
        // seven steps, two scalar instructions each (heap_pop_wave's walk, unrolled: every node of levels 0..6 has both
        // children); the last one reads nodes 63..126 from the second mask at position hp - 64 = hp's low six bits
        const unsigned long long ms0 = m0 << 1, ms1 = (m1 << 1) | (m0 >> 63);
        asm volatile(
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[a], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]\n\t"
            "s_bitcmp0_b64 %[b], %[hp]\n\ts_addc_u32 %[hp], %[hp], %[hp]"
            : [hp] "+s"(hp)
            : [a] "s"(ms0), [b] "s"(ms1)
            : "scc");
    }
    // relative node of this lane inside a five-level window: 1-based, the window's root = 1, lanes 0..61 hold 2..63
    const uint32_t r = (uint32_t)lane + 2;
    const uint32_t dr = 31u - (uint32_t)__builtin_clz(r);
    // one window: E2 = pairs of children that both exist, M = "the left child moves up"; walks up to five levels.
    // The pair under relative node P sits in lanes 2P - 2 and 2P - 1; with both masks shifted up by two its bit is at
    // position 2P: per level s_lshl (2P), s_bitcmp1 (does the pair exist), branch, s_bitcmp0 + s_addc (P <- 2P + "right").
    auto walk = [&](float key, uint32_t idx) {
        const float key_r = __shfl_down(key, 1);   // even lanes hold left children, their right siblings sit one lane up
        const bool both = (lane & 1) == 0 && lane < 62 && idx + 1 < len;
        const unsigned long long E2 = __ballot(both) << 2;
        const unsigned long long M = __ballot(both && Beam::before(key_r, key)) << 2;
        uint32_t P = 1, tmp;
        asm volatile(
            "1:\n\t"
            "s_lshl_b32 %[t], %[P], 1\n\t"
            "s_bitcmp1_b64 %[e], %[t]\n\t"
            "s_cbranch_scc0 2f\n\t"
            "s_bitcmp0_b64 %[m], %[t]\n\t"
            "s_addc_u32 %[P], %[P], %[P]\n\t"
            "s_cmp_lt_u32 %[P], 32\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:"
            : [P] "+s"(P), [t] "=&s"(tmp)
            : [e] "s"(E2), [m] "s"(M)
            : "scc");
        const uint32_t steps = 31u - (uint32_t)__builtin_clz(P);
        hp = (hp << steps) | (P & ((1u << steps) - 1u));
        d += steps;
        return steps;
    };
    // levels 8..12: the page of the level-7 node (read whenever that node has a child at all, so that every path
    // entry of these levels -- including a last left-only child -- is in some lane's registers)
    const uint32_t pbase = (hp - 128u) * kPageDwords;     // wave-uniform: everything below addresses the page relative to it
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    uint32_t steps = 0;
    if (2 * hp - 1 < len) {
        const uint32_t idx = ((hp << dr) | (r & ((1u << dr) - 1u))) - 1u;
        if (lane < 62 && idx < len) w = h.pload(pbase + 3u * r);
        steps = walk(__uint_as_float(w.x), idx);
    }
    // levels 13+: windows of keys over the heap-ordered tail
    while (steps == 5 && 2 * hp < len) {
        const uint32_t idx = ((hp << dr) | (r & ((1u << dr) - 1u))) - 1u;
        float key = 0.0f;
        if (lane < 62 && idx < len) key = __uint_as_float(h.gt[beam_tail_off(idx)]);
        steps = walk(key, idx);
    }
    if (2 * hp == len) {                           // a last node with a left child only: len even, hole == (len - 2) / 2
        hp = 2 * hp;
        ++d;
    }
    // the moves (heap_pop_wave's second half; d <= 31 lanes take part).  Lane t moves the entry at p_(t+1) to p_t;
    // p_t sits on heap level t: LDS up to level 7, the page (relative node = the level's bits of the path string
    // under a leading one) up to level 12, the tail beyond.
    const uint32_t t = (uint32_t)lane < d ? (uint32_t)lane : 0u;
    const uint32_t dst_hp = hp >> (d - t);                                     // p_t + 1
    const uint32_t src_hp = hp >> (d - t - ((uint32_t)lane < d ? 1u : 0u));    // p_(t+1) + 1, level t + 1
    auto page_rel = [](uint32_t node_hp, uint32_t lvl) {                        // lvl = heap level - 7, 1..5
        return (node_hp & ((1u << lvl) - 1u)) | (1u << lvl);
    };
    auto put_level = [&](uint32_t level, uint32_t node_hp, uint4 val) {
        if (level <= 7) h.lds_put(node_hp - 1, val);
        else if (level <= 12) h.pstore(pbase + 3u * page_rel(node_hp, level - 7u), val);
        else h.tstore(beam_tail_off(node_hp - 1), val);
    };
    // sources on levels 8..12 come out of the window lane that read them: relative node -> lane r - 2
    const bool from_page = (uint32_t)lane < d && t >= 7 && t <= 11;
    const int page_lane = (int)page_rel(src_hp, from_page ? t - 6u : 1u) - 2;
    const uint4 pw = make_uint4((uint32_t)__shfl((int)w.x, page_lane), (uint32_t)__shfl((int)w.y, page_lane),
                                (uint32_t)__shfl((int)w.z, page_lane), 0u);
    uint4 e = v;
    bool c = false;
    if ((uint32_t)lane < d) {
        if (t < 7) e = h.lds(src_hp - 1);
        else if (t <= 11) e = pw;
        else e = h.tload(beam_tail_off(src_hp - 1));
        c = Beam::before(Beam::key_of(e), Beam::key_of(v));
    }
    const unsigned long long stay = ~__ballot(c) & ((1ull << d) - 1ull);
    const uint32_t fin = stay ? 64u - (uint32_t)__builtin_clzll(stay) : 0u;
    if ((uint32_t)lane < fin) put_level(t, dst_hp, e);
    if (fin == d) { if (lane == 0) put_level(d, hp, v); }
    else if ((uint32_t)lane == fin) put_level(t, dst_hp, v);
}

// std::push_heap of `v` at index `hole` >= kBeamLds: heap_push_wave on the hybrid accessors -- the ancestors of the
// leaf (LDS or HBM, by index) are read by one lane each, one round trip whatever the depth.  A leaf inside the pages
// (hole < 8191: the usual case) has all its HBM ancestors in its own page, at the leaf's relative index shifted right.
__device__ __forceinline__ void beam_push_hybrid(const Beam& h, uint32_t hole, uint4 v, int lane_in) {
    const int lane = opaque_lane(lane_in);
    const uint32_t hp = hole + 1;
    const uint32_t depth = 31u - (uint32_t)__builtin_clz(hp);
    const uint32_t t = (uint32_t)lane + 1;
    const bool on = t <= depth;
    uint4 e = v;
    bool down = false;
    if (hp <= kBeamPaged) {
        const uint32_t lvl = depth - 7u;                                       // 1..5 (wave-uniform, like everything up to rl)
        const uint32_t pbase = ((hp >> lvl) - 128u) * kPageDwords;
        const uint32_t rl = (hp & ((1u << lvl) - 1u)) | (1u << lvl);           // the leaf's relative index in its page
        // ancestor p_t sits on level depth - t: in the page while t < lvl (relative index rl >> t), in LDS above
        if (on) {
            if (t < lvl) e = h.pload(pbase + 3u * (rl >> t)); else e = h.lds((hp >> t) - 1);
            down = Beam::before(Beam::key_of(e), Beam::key_of(v));
        }
        const unsigned long long stop = ~__ballot(down);
        const uint32_t m = (uint32_t)__builtin_ctzll(stop);
        auto put_anc = [&](uint32_t s, uint4 val) {                            // ancestor p_s (s = 0: the leaf)
            if (s < lvl) h.pstore(pbase + 3u * (rl >> s), val); else h.lds_put((hp >> s) - 1, val);
        };
        if (t <= m) put_anc(t - 1, e);
        if ((uint32_t)lane == m) put_anc(m, v);
        return;
    }
    const uint32_t pt = on ? (hp >> t) - 1 : 0u;
    if (on) {
        e = h.raw(pt);
        down = Beam::before(Beam::key_of(e), Beam::key_of(v));
    }
    const unsigned long long stop = ~__ballot(down);
    const uint32_t m = (uint32_t)__builtin_ctzll(stop);
    if (t <= m) h.put((hp >> (t - 1)) - 1, e);
    if ((uint32_t)lane == m) h.put((hp >> m) - 1, v);
}

// BoundedMaxHeap::push (rabitq_search.hpp:26-35) by the whole wave; every argument is wave-uniform, `top` is the
// heap's current root key.  Same element movement as nn_push (which stays as the path for heaps too large for
// the two ballot masks of the wave-parallel pop).
__device__ __forceinline__ void nn_push_wave(const NnLds& w, Result* h, uint32_t& size, uint32_t k, uint32_t id, float dist,
                                             float top, int lane_in) {
#ifndef CPH_NO_LAUNDER_NN
    const int lane = opaque_lane(lane_in);     // (one expansion in four or fewer gets here: see opaque_lane)
#else
    const int lane = lane_in;
#endif
    const uint2 r = make_uint2(id, __float_as_uint(dist));
    if (size < k) {
        if (size < kWaveHeapMax) {
            heap_push_wave(w, size, r, lane);
        } else {
            if (lane == 0) nn_sift_up(h, size, 0, Result{id, dist});
            __builtin_amdgcn_wave_barrier();
        }
        ++size;
    } else if (dist < top) {
        if (size <= kWaveHeapMax) {
            if (size > 1) heap_pop_wave(w, size, lane);        // std::pop_heap: the old last value sinks in from the root
            heap_push_wave(w, size - 1, r, lane);              // back() = r; std::push_heap
        } else {
            if (lane == 0) {
                nn_adjust(h, 0, size - 1, h[size - 1]);
                nn_sift_up(h, size - 1, 0, Result{id, dist});
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ uint32_t bcast_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float bcast_f32(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}

// LDS carve-up (bytes): qm[PW*16] | qv[D*4] | vec[512 or D*4] (popped vertex's vector, LDS-DMA target) |
// exact[128] | list[64] | slack[128] | ratio[16] | totals[80] | nn[k*8] |
// beam top levels (kBeamLds+1) x 16
constexpr uint32_t kLdsTail = 128 + 64 + 128 + 16 + 80;       // exact | list | slack | ratio | totals
// vec[] holds the popped vertex' whole vector in the instantiations with a compile-time D (LDS-DMA target)
__host__ __device__ inline uint32_t search_vec_bytes(uint32_t D, bool static_d) { return static_d ? D * 4 : 512; }
__host__ __device__ inline bool search_static_d(uint32_t D) { return D == 128 || D == 1024; }
__host__ __device__ inline size_t search_lds_bytes(uint32_t D, uint32_t PW, uint32_t k) {
    return (size_t)PW * 16 + (size_t)D * 4 + search_vec_bytes(D, search_static_d(D)) + kLdsTail +
           (((size_t)k * 8 + 15) & ~(size_t)15) + 16 * (kBeamLds + 1);
}

// LDS-DMA loads (global -> LDS, no VGPR destination), written as inline assembly on purpose: hipcc
// waits vmcnt(0) at every later load once it has issued one itself, which would serialise the
// neighbour-id wait behind the whole block.  The compiler therefore does not know about these
// loads; loads retire in order, so its own counted waits stay correct (merely conservative),
// and the one place that reads a DMA target waits explicitly.  LDS destination = M0 + lane * size.
// M0 is written inside the asm without appearing in the clobber list: it is a reserved register for
// LLVM's AMDGPU backend (clang warns that "m0" in a clobber list is not honoured); the backend never
// keeps a value live in it across other instructions -- every instruction that reads M0 gets its own
// copy glued directly in front of it -- so overwriting it here cannot be observed.
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_void*)p);
}
__device__ __forceinline__ void lds_dma16(const void* g, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory");
}

// Waves per SIMD: 6 (<= 80 VGPRs) for the static D = 128 instantiations, 5 (<= 96 VGPRs) for the
// generic ones -- measured on MI355X
#ifndef CPH_SEARCH_WAVES_PER_SIMD
#define CPH_SEARCH_WAVES_PER_SIMD 5
#endif
#ifndef CPH_SEARCH_WAVES_PER_SIMD_128
#define CPH_SEARCH_WAVES_PER_SIMD_128 6
#endif
// D = 1024: the whole block's codes (up to 16 x 16 B per lane) are in flight at once: 2 waves per SIMD (3 spill:
// 7.4 against 5.2 ms per 1,000 queries at C3); LDS (query + vertex vector, 4 KB each) allows 11 per CU anyway
#ifndef CPH_SEARCH_WAVES_PER_SIMD_1024
#define CPH_SEARCH_WAVES_PER_SIMD_1024 2
#endif
#ifndef CPH_SEARCH_WAVES_PER_SIMD_128_NARROW
#define CPH_SEARCH_WAVES_PER_SIMD_128_NARROW CPH_SEARCH_WAVES_PER_SIMD_128
#endif
__host__ __device__ constexpr int search_waves_per_simd(int sd, int bw) {
    return sd == 128 ? (bw <= 2 ? CPH_SEARCH_WAVES_PER_SIMD_128_NARROW : CPH_SEARCH_WAVES_PER_SIMD_128)
                     : (sd == 1024 ? CPH_SEARCH_WAVES_PER_SIMD_1024 : CPH_SEARCH_WAVES_PER_SIMD);
}
// PF: the probe-first order of the loads (see the expansion loop).  The small-batch launch uses the instantiation without it:
// a handful of queries is a matter of latency, and the third dependent round trip costs a single query 10 % (565 -> 624 us).
template <int BW, int SD, bool PF = true>
__global__ __launch_bounds__(64, search_waves_per_simd(SD, BW)) void search_kernel(SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int li = lane & 31;
    const uint32_t D = SD ? SD : a.L.D;
    const uint32_t PW = SD ? (SD >= 32 ? SD / 32 : 1) : a.L.PW;
    const uint32_t k = a.k;
    uint4* qm = reinterpret_cast<uint4*>(smem);
    float* qv = reinterpret_cast<float*>(smem + (size_t)PW * 16);
    unsigned char* fixed = smem + (size_t)PW * 16 + (size_t)D * 4;
    float* s_vec = reinterpret_cast<float*>(fixed);
    constexpr uint32_t vsz = SD >= 128 ? (uint32_t)SD * 4u : 512u;
    float* s_exact = reinterpret_cast<float*>(fixed + vsz);
    uint8_t* s_list = fixed + vsz + 128;
    float* s_slack = reinterpret_cast<float*>(fixed + vsz + 192);
    double* s_ratio = reinterpret_cast<double*>(fixed + vsz + 320);   // [2]
    // The work counters of this wave's queries are summed here and reach the statistics line in HBM ONCE, when the wave
    // leaves: six atomics per query on one line from six thousand waves are a queue in the L2's atomic unit that every
    // query's first load has to wait behind (vmcnt counts them).
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(fixed + vsz + 336);   // [10], indexed like stats[]
    if (lane < 10) s_tot[lane] = 0ull;
    Result* nn = reinterpret_cast<Result*>(fixed + vsz + kLdsTail);
    NnLds nnw;
    nnw.l = (lds_u32x2*)(fixed + vsz + kLdsTail);
    // the beam's LDS levels, 16-B aligned, addressed as LDS (address space 3)
    const uint32_t beam_off = PW * 16 + D * 4 + vsz + kLdsTail + ((k * 8 + 15) & ~15u);
    lds_u32x4* s_beam = (lds_u32x4*)(smem + beam_off);

    // LDS byte offset of s_vec for the DMA's M0: the dynamic LDS starts right behind the kernel's static
    // LDS (none here), so this is a compile-time constant -- a generic-to-LDS pointer cast would be
    // re-derived (with its null check) by ten scalar instructions in every expansion
    const uint32_t vec_off = __builtin_amdgcn_groupstaticsize() + PW * 16 + D * 4;
    const uint32_t slot = blockIdx.x;
    uint32_t* bm = CPH_COLD(bitmaps) + (size_t)slot * CPH_COLD(bm_words);
    uint32_t* logi = CPH_COLD(log_ids) + (size_t)slot * a.cap;
    Beam heap;
    heap.l = s_beam;
    heap.gp = CPH_COLD(beam_pages) + (size_t)slot * kBeamPagesDwords;
    heap.gt = CPH_COLD(beam_tail) + (size_t)slot * beam_tail_dwords(a.cap);
    const float FMAX = 3.402823466e+38f;
    if (lane < kMaxSlack) s_slack[lane] = a.sc.slack[lane];

    // No next-top prefetch (the reference has one, :124-128).  It was kept for batches that fit the resident slots
    // ("latency mode") until the end of round 2, when it measured as a loss at every batch size with the current kernel:
    // a single query 718 -> 622 us per call without it, 32 queries 960 -> 829 us, 2,000 queries 1,396 -> 1,129 us
    // (scripts/slot_fraction_sweep.py, scripts/single_query_latency.py) -- the ~40 % of predictions that a later push
    // invalidates cost bandwidth and issue slots, and the pop already runs under the block's own latency.  (Earlier:
    // switching it on when a larger batch starts to drain, or per query for the head or the tail of the launch order,
    // were 1-4 % slower on the 10k batch or far worse, profiles/r2_lat_sweep.jsonl.)
    // Hiding more latency per wave buys nothing either -- tried and measured on one box: issuing the next expansion's loads
    // before this expansion's beam pushes (the next top is known without doing them) 2.36 -> 2.46 ms per 10k queries; doing
    // the same at a second issue site made the register allocator copy the loaded registers at the loop head, i.e. wait
    // for them: 2.50 ms.  Round 4: raising the priority of a wave whose query has passed 200 / 300 / 400 expansions
    // (s_setprio; a launch lasts as long as its longest query) changed nothing either way.
    const uint32_t* nq_dev = CPH_COLD(nq_dev);
    const uint32_t nq = nq_dev ? *nq_dev : CPH_COLD(nq);
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(CPH_COLD(counter), 1u);
        t = bcast_u32(t);
        if (t >= nq) break;
        const uint32_t* todo = CPH_COLD(todo);
        const uint32_t qi = todo ? todo[t] : t;

        {
            const uint4* qmasks = CPH_COLD(qmasks);
            const float* queries = CPH_COLD(queries);
            for (uint32_t w = lane; w < PW; w += 64) qm[w] = qmasks[(size_t)qi * PW + w];
            for (uint32_t d = lane; d < D; d += 64) qv[d] = queries[(size_t)qi * D + d];
        }
        const QueryHeader hd = CPH_COLD(qhdr)[qi];
        __syncthreads();

        QP qp;
        qp.A = hd.A; qp.B = hd.B; qp.C = hd.C;
        {
            // The estimator's calibration constants as opaque scalar values: left as plain kernarg reads the compiler
            // re-reads them from the kernarg segment inside every expansion when it is short of scalar registers -- an
            // s_load + s_waitcnt lgkmcnt(0) three times on the dependent chain (1 % of the kernel's time).
            float af_a = a.sc.affine_a, af_b = a.sc.affine_b, fl = a.sc.ip_qo_floor;
            asm volatile("" : "+s"(af_a), "+s"(af_b), "+s"(fl));
            qp.affine_a = af_a; qp.affine_b = af_b; qp.floor = fl;
        }
        qp.slack = s_slack[0];
        const float gamma = a.sc.gamma;

        // query_norm_sq = dot(q, q)  (:88)
        float qnorm;
        {
            float c = 0.0f;
            for (uint32_t i = lane & 7; i < D; i += 8) c = __fmaf_rn(qv[i], qv[i], c);
            qnorm = bcast_f32(group_reduce8_lo(c));
        }

        // wave-uniform heap sizes (re-broadcast after every lane-0 section)
        uint32_t beam_size = 0, nn_size = 0;
        float gamma_q = gamma;
        // gamma-adaptation running sums live in LDS (touched only when a neighbour is reranked)
        if (lane == 0) { s_ratio[0] = 0.0; s_ratio[1] = 0.0; nn[0].dist = FMAX; }   // (an empty result heap reads as threshold FLT_MAX)
        uint32_t ratio_count = 0;
        // uniform state
        uint32_t log_count = 0;
        int slack_batch = 0;
        bool overflow = false, wipe = false, stage2_redo = false;
        uint32_t st_exp = 0, st_exact = 0, st_new = 0, st_push = 0, st_skip = 0, st_allseen = 0;
        // probe first only: expansions where the reference's stage-2 decision is unobservable and was left open (the reference
        // may have skipped stage 2 there or not -- st_skip counts the skips that were decided)
        uint32_t st_undecided = 0;
#ifdef CPH_TRAFFIC_STATS
        // diagnostic build only: what the spilled beam and the estimated-set probe touch (stats[8..15])
        unsigned long long trf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef CPH_PHASE_TIMERS
        unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long tlast = clock64();
#define CPH_TICK_(i) do { unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; } while (0)
        // -DCPH_PHASE_TIMERS=2: finer buckets along one expansion's dependent chain (CPH_TICKF), for the latency-bound workloads:
        // 0 head + load issue | 1 pop (incl. its windows) | 2 rest of the block wait | 3 probe issue, exact distance, result push |
        // 4 estimator | 5 rest of the probe wait | 6 marking, candidates, speculative rerank | 7 beam pushes + loop tail
#if CPH_PHASE_TIMERS + 0 == 2
#define CPH_TICK(i) do {} while (0)
#define CPH_TICKF(i) CPH_TICK_(i)
#else
#define CPH_TICK(i) CPH_TICK_(i)
#define CPH_TICKF(i) do {} while (0)
#endif
#else
#define CPH_TICK(i) do {} while (0)
#define CPH_TICKF(i) do {} while (0)
#endif

        // entry: ep_est = exact_l2(ep); beam.push({ep_est, 0, ep}); mark estimated (:95-97)
        {
            const uint32_t ep = hd.entry;
            const float enorm = a.norm_sq[ep];
            float dot = group_dot8_lo(qv, a.raw + (size_t)ep * D, D, lane & 7);
            float ex = exact_from_dot(qnorm, enorm, dot);
            st_exact++;
            if (lane == 0) {
                logi[0] = ep;
                heap.put(0, beam_pack(BeamEntry{ex, 0.0f, ep}));
                atomicOr(&bm[ep >> 5], 1u << (ep & 31));
            }
            beam_size = 1;
            log_count = 1;
        }
        __syncthreads();

        for (;;) {
            // ---- termination tests on the beam's top (:106-122); the pop itself comes later -----
            CPH_TICK(7);
            if (beam_size == 0) break;
            uint32_t cur_id;
            float worst_pop;   // result-heap threshold as read here (wave-uniform)
            // std::pop_heap of the beam (whole wave while it lives in LDS, lane 0 once it has spilled)
            auto pop_beam = [&](uint32_t size) {
#ifdef CPH_TRAFFIC_STATS
                trf[5] += size;
                if (size > trf[6]) trf[6] = size;
                if (size > kBeamLds) { trf[0]++; uint32_t lv = 31u - (uint32_t)__builtin_clz(size); trf[1] += (lv - 7 + 4) / 5; }
                if (size > 8191) trf[7]++;
#endif
                if (size > 1) {
                    if (size <= kBeamLds) heap_pop_wave(heap, size, lane);
                    else beam_pop_hybrid(heap, size, lane);
                }
            };
            {
                const uint4 topv = heap.lds(0);
                const float worst = nn[0].dist;       // FLT_MAX while the result heap is empty (set at query start)
                cur_id = bcast_u32(topv.z);
                const float cur_est = __uint_as_float(topv.x);
                const float cur_lower = __uint_as_float(topv.y);
                uint32_t verdict = 2;  // 0 = done, 1 = skip (lower-bound pruned), 2 = expand
                if (nn_size >= k && cur_est >= gamma_q * worst) verdict = 0;
                else if (nn_size >= k && cur_lower > worst) verdict = 1;
                verdict = bcast_u32(verdict);   // every lane read the same words: make it provably uniform
                if (verdict == 0) break;
                if (verdict == 1) {             // pruned: the entry leaves the beam, nothing is loaded
                    pop_beam(beam_size);
                    --beam_size;
                    continue;
                }
                worst_pop = bcast_f32(worst);
            }
            CPH_TICK(0);

            // ---- this expansion's loads, then the estimated-set probe (:227), a dependent round
            // trip that overlaps the exact distance and the result-heap push below ------------
            // (compile-time layout for the probe-first instantiations -- cph_core.h StaticLayout: they re-read the layout from
            // the kernarg segment inside the loop otherwise; the instantiations without it keep it in scalar registers, and
            // lose their clean loop -- two scratch reloads in the pop -- when the constants change the allocation)
            constexpr bool kStaticLayout = SD >= 128 && PF;
            const uint32_t blk_stride = kStaticLayout ? StaticLayout<BW, SD>::kStride : a.L.stride;
            const uint32_t blk_ids_off = kStaticLayout ? StaticLayout<BW, SD>::kIdsOff : a.L.ids_off;
            const uint8_t* blk = a.blocks + (size_t)cur_id * blk_stride;
            const float* vrow = a.raw + (size_t)cur_id * D;
            // Nothing that matters is in flight here (the previous expansion's prefetch is thousands
            // of cycles old, its marking atomics return nothing).  Saying so with a wait the compiler
            // can see keeps it from protecting registers it believes some path around the loop left
            // a load pending on -- such a wait, between the loads below, would be a real one.
#ifndef CPH_NO_LOOPHEAD_WAIT
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#endif
            // The popped vertex's own vector goes straight to LDS (LDS-DMA, 16 B per lane, one
            // instruction for the 512 B; the strided per-chain reads then come from LDS).  It is
            // issued FIRST: loads retire in order, so retiring the other loads below also covers it.
            if constexpr (SD == 128) {
                if (lane < 32) lds_dma16(vrow + 4 * lane, vec_off);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (SD > 128) {
#pragma unroll
                for (int c = 0; c < SD / 256; ++c) lds_dma16(vrow + 4 * lane + 256 * c, vec_off + 1024 * c);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float norm_ld = a.norm_sq[cur_id];
            const uint32_t nid_ld = reinterpret_cast<const uint32_t*>(blk + blk_ids_off)[li];
            // ---- everything else this expansion reads is issued before the probe ----------
            BlockLoads<BW, SD> bl;
            // PROBE FIRST (the static-D instantiations, D = 128 and D = 1024; 4-bit codes at D = 128 in round 3, every width and
            // both shapes since round 4 -- the host picks the instantiation per workload).  The reference evaluates all 32 neighbours of a block, but only the NEW ones
            // -- 3.3 of 32 on the SIFT-like benchmark, none in a quarter of the expansions -- have any observable effect.  The
            // ids (128 B) decide that, so they, the norm and the vector go out here; the codes and aux values -- 2,560 of the
            // block's 2,752 bytes -- are fetched after the probe, by the lanes of the new neighbours only (both lane halves
            // of a neighbour; eight neighbours share a 128-byte line, so ~40 % of the lines drop out) and not at all when
            // nothing is new.  One more dependent round trip in three expansions out of four against ~40 % less traffic:
            // full queue 16.4 -> 15.3 ms per 100,000 queries, the 10,000-query launch 2.19 -> 2.14 ms, results identical.
            // Whether it pays depends on how many neighbours are new: on the SIFT-like data (1-3 of 32) it gains 7-13 % at
            // every bit width, on the Gaussian gate workloads (5-10 new, nothing to skip) it only adds a round trip (-3 % to
            // -13 %): the library chooses from the previous batches' own counters (cphnsw_mi355x.hip: probe_first).  Without
            // it the narrow codes run their estimator under the probe's round trip (kSpeculate).
#ifdef CPH_NO_PROBE_FIRST
            constexpr bool kProbeFirst = false;
#else
            constexpr bool kProbeFirst = PF && SD >= 128;
#endif
            if constexpr (kProbeFirst) {
#pragma unroll
                for (int kk = 0; kk < BlockLoads<BW, SD>::kCPL; ++kk) bl.c[kk] = make_uint4(0u, 0u, 0u, 0u);
                bl.aux = make_uint4(0u, 0u, 0u, 0u);
            } else {
                bl.issue(blk, a.L, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            CPH_TICKF(0);
            // ---- the pop, while those loads are in flight.  It needs only the id of the top, which the loads
            // above already have; its three dependent LDS round trips and its scalar path walk (a fifth of an
            // expansion's time when it ran in front of the loads) now hide behind the block's memory latency.
            // LDS and scalar work only (lgkmcnt), so no wait on the loads (vmcnt) is forced here.
            // Spilled beams: the pushes at the end of this expansion will compare against the ancestors of the leaf the
            // pop is about to free (heap index beam_size - 1).  Those of them that live in the pages are touched now, one
            // per lane, in the same round trip as the block: the pushes' own reads then hit in L2 instead of being another
            // dependent trip to HBM.  The value is consumed, unused, where the block's loads are retired anyway (the touch
            // is the youngest load there, so the compiler's wait for it is the wait the block needs in any case); beams
            // that fit the LDS levels -- every C2 expansion -- skip the whole thing on a wave-uniform branch.
            uint32_t touch = 0;
            if (beam_size > kBeamLds) {
                // leaf = heap index beam_size - 1, i.e. hp = beam_size on level `depth`; its ancestor on level 7 + j
                // (j = 1..5, below the leaf's own level) is relative node ((hp >> (depth - 7 - j)) & (2^j - 1)) | 2^j
                // of the page of hp >> (depth - 7): lane j - 1 touches it
                const uint32_t depth = 31u - (uint32_t)__builtin_clz(beam_size);
                const uint32_t j = (uint32_t)lane + 1u;
                const bool on = j <= 5u && 7u + j < depth;
                const uint32_t sh = on ? depth - 7u - j : 0u;
                const uint32_t rj = ((beam_size >> sh) & ((1u << (j & 7u)) - 1u)) | (1u << (j & 7u));
                const uint32_t pb = ((beam_size >> (depth - 7u)) - 128u) * kPageDwords;
                touch = heap.gp[on ? pb + 3u * rj : 0u];
            }
            pop_beam(beam_size);
            --beam_size;
            __builtin_amdgcn_sched_barrier(0);
            CPH_TICKF(1);
            // All of this expansion's loads come back together (issued back to back, retired in
            // order); they are retired HERE, before the probe goes out.  The probe is only issued when
            // a lane has something to look up, and a load that is only maybe in flight makes every
            // later wait of the compiler a vmcnt(0) -- the norm or the codes would then wait for the
            // probe's round trip.  This way that round trip (and the prefetch behind it) overlaps the
            // whole estimator arithmetic; the vector DMA is older than these loads, so it has landed.
            uint32_t nid = nid_ld;
            float cur_norm = norm_ld;
            float dot_generic = 0.0f;
            if constexpr (SD < 128) dot_generic = bcast_f32(group_dot8_lo(qv, vrow, D, lane & 7));   // its loads, too, go first
            if constexpr (!kProbeFirst) bl.retire();
            asm volatile("" : "+v"(cur_norm), "+v"(nid) : "v"(touch));
            CPH_TICKF(2);
            const bool valid = nid != kInvalidNode;  // slot < count (set by the repacker)
            const bool active = lane < 32 && valid;
            // The probe is a plain (device-coherent: it must not be served from this CU's L1, which
            // the marking atomics below bypass) load of the slot's bitmap word; only the few
            // neighbours that turn out to be new pay for a read-modify-write, and that one needs
            // no return value.  (A test-and-set for all 32 made the L2 atomic units the bottleneck.)
            uint32_t old_bits = 0;
            const uint32_t my_bit = 1u << (nid & 31);
            // (every lane loads -- the upper half and empty slots read word 0 -- instead of a predicated load: one select
            // against an exec-mask save / branch / restore on the scalar unit)
            old_bits = __hip_atomic_load(&bm[active ? nid >> 5 : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
#ifdef CPH_TRAFFIC_STATS
            {   // distinct 128-byte lines of the bitmap this probe touches
                const uint32_t line = nid >> 10;
                bool first = active;
                for (int j = 0; j < 31; ++j) {
                    const uint32_t oj = __shfl(line, j);
                    const bool aj = __shfl((int)active, j) != 0;
                    if (aj && lane > j && oj == line) first = false;
                }
                trf[4] += __popcll(__ballot(first));
            }
#endif

            // ---- exact distance of the popped node; nn.push (:130-133) ----------------
            float exact_dist;
            {
                float dot;
                if constexpr (SD >= 128) dot = bcast_f32(group_reduce8_lo(chain_dot_lds<SD / 8>(qv, s_vec, lane & 7, 0.0f)));
                else dot = dot_generic;
                exact_dist = exact_from_dot(qnorm, cur_norm, dot);
            }
            st_exact++;
            st_exp++;
            // BoundedMaxHeap::push changes the heap only while it is filling or when the new distance
            // beats the threshold: both are wave-uniform facts, so the three expansions out of four
            // that leave it alone skip the lane-0 section and the re-read of the threshold
            float worst0 = worst_pop;
            const bool nn_changes = bcast_u32((nn_size < k || exact_dist < worst_pop) ? 1u : 0u) != 0u;   // provably uniform
            if (nn_changes) {
                nn_push_wave(nnw, nn, nn_size, k, cur_id, exact_dist, worst_pop, lane);
                __builtin_amdgcn_wave_barrier();
                worst0 = bcast_f32(nnw.lds_key(0));   // (the heap holds at least the entry just pushed)
            }
            const uint32_t nn_sz = nn_size;
            CPH_TICK(1);
            CPH_TICKF(3);
            if (!__any(active)) {  // n_neighbors == 0 (:137)
                // (retire the probe's destination register on this path too: a load left pending
                // across the back edge costs every expansion a wait where that register is reused)
                asm volatile("" ::"v"(old_bits));
                continue;
            }

            // slack level schedule (:141-145)
            if (a.sc.num_slack > 0) {
                int lvl = slack_batch < a.sc.num_slack - 1 ? slack_batch : a.sc.num_slack - 1;
                qp.slack = s_slack[lvl];
                ++slack_batch;
            }
            // ---- FastScan estimates (:159-206) ----------------------------------------
            float est = 0.0f, lower = 0.0f;
            bool skipped2 = false, undecided2 = false;
            // a list shorter than 32 whose length is not a multiple of 8 ends in the reference's scalar tail
            // (device_fastscan.h: TailLanes); wave-uniform, never taken on graphs the builders write.  The probe-first
            // instantiation sits on its register limit and is only launched on indexes without such lists (the loader
            // knows: SearchArgs::flags bit 1 sends the others to the instantiation without it).
            const TailLanes tl = kProbeFirst ? no_tail() : tail_lanes(valid, lane);
            // `fetched`: the lanes whose codes and aux values bl holds -- all of them, or (probe first) the new ones.
            // Returns false when the reference's stage-2 decision cannot be taken from the fetched lanes (probe first only).
            auto estimate = [&](bool fetched) -> bool {
                LaneEst v;
                const float dqp = exact_dist;
                const float sq = __builtin_sqrtf(dqp);
                bl.reduce(blk, a.L, qm, lane, v);
                if constexpr (BW == 1) {
                    stage2_est<1>(qp, v, dqp, sq, est, lower, tl);
                } else {
                    // both lower bounds at once, one per lane half (see lower_bounds_split); dqp is the popped vertex's
                    // distance, the same in every lane
                    float lo1 = 0.0f, lo2 = 0.0f;
                    const bool tiny = bcast_f32(dqp) < kEpsSmall;
                    if (!tiny) lower_bounds_split<BW>(qp, v, dqp, sq, lane, lo1, lo2, tl);
                    // any_survivor (:178-187) ranges over ALL the list's neighbours, estimated before or not
                    if (__builtin_expect(nn_sz < k || __any(fetched && valid && lo1 < worst0), 1)) {
                        est = stage2_est_only<BW>(qp, v, dqp, tl, tiny);
                        lower = lo2;
                    } else if constexpr (!kProbeFirst) {            // the reference's skipped batch (:201-205)
                        est = FMAX;
                        lower = lo1;
                        skipped2 = true;
                    } else {
                        // Probe first: only the NEW neighbours' codes are here, and none of them survives stage 1.
                        // Whether the reference runs stage 2 then depends on the bounds of neighbours it will skip
                        // anyway (:227) -- which matters only if a new neighbour would ACT on its stage-2 values where
                        // the skipped batch (est = FLT_MAX, lower = stage-1 bound >= worst) makes it do nothing: the two
                        // bounds have different numerators (2 S0 + S1 over 3 against S0) and are not ordered.  No
                        // rerank can happen before such a neighbour's turn without being such an action itself, so the
                        // thresholds of the loop's entry decide: if no new neighbour passes them, both branches leave
                        // the heaps alone and the block's other codes are never needed.  Else the query is handed to
                        // the re-run launch, whose instantiation fetches whole blocks and takes the reference's
                        // decision on the real bounds.  (The stage-1 bounds are loose -- the reference itself never
                        // skips a batch on its own graphs, SURVEY F4 -- so neither happens outside the fixtures that
                        // force it: 0 of 2.65 M expansions on the C2 benchmark.)
                        // (The estimate is computed from a laundered copy of the sum: left recognisable, the compiler
                        // merges this call with the one of the usual branch and hoists both above the ballot, which
                        // measured 15 % slower on the C2 benchmark -- the estimate's division chain then sits in front
                        // of the branch every expansion waits on.)
                        LaneEst w = v;
                        asm volatile("" : "+v"(w.nbit));
                        est = stage2_est_only<BW>(qp, w, dqp, tl, tiny);
                        lower = lo2;
                        const bool acts = fetched && !(lo2 >= worst0) && (est < worst0 || est < gamma_q * worst0);
                        if (__any(acts)) return false;
                        undecided2 = true;
                    }
                }
                return true;
            };
            // Narrow codes (1- and 2-bit) estimate poorly, so their searches run long, almost always find something new
            // (9 % of the gate workload's expansions are all-seen, 24 % of C2's) and are bound by dependent round trips,
            // not by bandwidth: there the estimator runs BEFORE the probe's result is looked at, under its round trip.
            // The 4-bit kernel keeps the order that skips the arithmetic of all-seen expansions.
            constexpr bool kSpeculate = BW <= 2 && !kProbeFirst;
            if constexpr (kSpeculate) estimate(true);
            CPH_TICKF(4);
#if CPH_PHASE_TIMERS + 0 == 2
            asm volatile("" : "+v"(old_bits));
            CPH_TICKF(5);
#endif

            // ---- estimated set: result of the probe issued above --------------------------------
            bool is_new = active && (old_bits & my_bit) == 0;
            // a vertex whose neighbour list repeats an id: only the first copy is new
            if ((a.flags & 1u) && __any(is_new)) {
                // (graphs written by the reference never repeat an id; the loader sets the
                // flag when one does, and only then is this screen paid for)
#pragma unroll 1   // cold path: rolled, or its 31 lane masks are hoisted and spilled for everyone
                for (int j = 0; j < 31; ++j) {
                    uint32_t oj = __shfl(nid, j);
                    bool nj = __shfl((int)is_new, j) != 0;
                    if (nj && lane > j && lane < 32 && oj == nid) is_new = false;
                }
            }
            const uint32_t new_mask = (uint32_t)(__ballot(is_new) & 0xFFFFFFFFull);
            // mark the new ids (two may share a word) and log them for the un-marking at query end (discovery order =
            // neighbour order) under ONE predicate; a log that would overflow ends the attempt before anything is marked
            const uint32_t n_new = __popc(new_mask);
            const uint32_t my_rank = __builtin_amdgcn_mbcnt_lo(new_mask, 0u);   // set bits below this lane (lanes < 32)
            // The beam must have room for this expansion's pushes (at most one per new id); a query that outgrows its
            // slot is answered by the full-capacity re-run.  The id log only exists to clear the bitmap cheaply: once it
            // is full the query simply stops logging and wipes its whole bitmap at the end (n / 8 bytes -- less than the
            // log it would have needed: a query that discovers more than `cap` ids has touched a good part of the graph).
            if (beam_size + n_new > a.cap) { overflow = true; break; }
            if (!wipe && log_count + n_new > a.cap) wipe = true;
            if (is_new) {
                atomicOr(&bm[nid >> 5], my_bit);
                if (!wipe) logi[log_count + my_rank] = nid;
            }
            // Every neighbour already estimated (a quarter to a third of the expansions, most of
            // them late in the search): the reference evaluates the block and then skips all 32
            // neighbours (:227), so nothing it computes is observable -- the FastScan arithmetic,
            // more than half of an expansion's vector instructions, is not issued at all.
            if (new_mask == 0) {
                ++st_allseen;
                continue;
            }
            if constexpr (kProbeFirst) {
                const bool fetch = ((new_mask >> (lane & 31)) & 1u) != 0u;       // both lane halves of a new neighbour
                if (fetch) bl.issue_static(blk, lane);
                if (__builtin_expect(!estimate(fetch), 0)) { overflow = true; stage2_redo = true; break; }
            } else if constexpr (!kSpeculate) {
                estimate(true);
            }
            if (skipped2) st_skip++;
            if (undecided2) st_undecided++;

            CPH_TICK(2);
            const bool warmup = nn_sz < k;  // (:210)
            bool cand = is_new && (warmup || (!(lower >= worst0) && est < worst0));
            const uint32_t cand_mask = (uint32_t)(__ballot(cand) & 0xFFFFFFFFull);

            st_new += n_new;
            CPH_TICK(3);

            // ---- speculative exact L2 of the candidates, 8 per pass (rare: ~0.15 per expansion) --
            if (cand_mask) {
                if (cand) s_list[__builtin_amdgcn_mbcnt_lo(cand_mask, 0u)] = (uint8_t)lane;
                __syncthreads();
                const uint32_t n_cand = __popc(cand_mask);
                const int g = lane >> 3;
                for (uint32_t base = 0; base < n_cand; base += 8) {
                    const bool have = base + g < n_cand;
                    const uint32_t idx = have ? s_list[base + g] : 0;
                    const uint32_t cid_l = (uint32_t)__shfl((int)nid, (int)idx);
                    const uint32_t cid = have ? cid_l : cur_id;
                    const float cnorm = a.norm_sq[cid];   // issued with the vector loads, not after them
                    float dot = group_dot8_lo(qv, a.raw + (size_t)cid * D, D, lane & 7);
                    float ex = exact_from_dot(qnorm, cnorm, dot);
                    asm volatile("" ::"v"(ex));   // keeps the norm load out of the branch below
                    if (have && (lane & 7) == 0) s_exact[idx] = ex;
                }
                st_exact += n_cand;
                __syncthreads();
            }
            CPH_TICK(4);
            CPH_TICKF(6);

            // ---- serial replay of the neighbour loop (:218-273).  The loop runs on lane 0 (heap
            // state lives there) but `i` comes from the wave-uniform ballot mask, so neighbour i's
            // est / lower / id are read straight out of lane i's registers with v_readlane ------
            if (!warmup && cand_mask == 0) {
                // No neighbour of this vertex is reranked, so the result heap, its threshold and
                // gamma_q stay what they were at loop entry: every decision of the serial loop is
                // `lower < worst0 <= est < gamma_q * worst0` and can be taken by all lanes at once;
                // only the beam pushes themselves keep neighbour order.
                const float dabs0 = gamma_q * worst0;
                const bool p = is_new && !(lower >= worst0) && est < dabs0;
                uint32_t pm = (uint32_t)(__ballot(p) & 0xFFFFFFFFull);
                // std::push_heap of each in neighbour order.  Usual case: every new entry is no
                // better than the parent of the leaf it lands on, so nothing moves and the pushes
                // are independent appends -- each lane checks its own parent and writes its own leaf.
                const uint32_t np = __popc(pm);
                bool appended = false;
                // (a single push onto a spilled beam goes straight to the push routine: the parent check of the append path
                // would be a dependent round trip of its own in front of the one the routine makes anyway)
#ifdef CPH_NO_SKIP_APPEND
                if (np != 0 && np <= beam_size + 1) {
#else
                if (np != 0 && np <= beam_size + 1 && !(np == 1 && beam_size >= kBeamLds)) {
#endif
                    const uint32_t pos = beam_size + __builtin_amdgcn_mbcnt_lo(pm, 0u);
                    const bool in_lds = beam_size + np <= kBeamLds;   // wave-uniform: the usual case keeps its ds_* accesses
                    bool stay = true;
                    if (in_lds) {
                        // every lane reads a parent (lanes without a push: the root) and is masked afterwards
                        const bool chk = p && pos > 0;
                        stay = !(heap.lds_key(chk ? (pos - 1) >> 1 : 0u) > est) || !chk;
                    } else if (p && pos > 0) {
                        stay = !(heap.raw_key((pos - 1) >> 1) > est);
                    }
                    if (__all(stay)) {
#ifdef CPH_TRAFFIC_STATS
                        if (!in_lds) trf[3]++;
#endif
                        const uint4 ent = make_uint4(__float_as_uint(est), __float_as_uint(lower), nid, 0u);
                        if (p) { if (in_lds) heap.lds_put(pos, ent); else heap.put(pos, ent); }
                        beam_size += np;
                        st_push += np;
                        appended = true;
                    }
                }
                if (!appended) {
                    while (pm) {
                        const int i = __ffs((int)pm) - 1;
                        pm &= pm - 1;
                        const uint32_t id_i = (uint32_t)__builtin_amdgcn_readlane((int)nid, i);
                        const uint32_t e_i = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(est), i);
                        const uint32_t lo_i = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lower), i);
#ifdef CPH_TRAFFIC_STATS
                        if (beam_size >= kBeamLds) trf[2]++;
#endif
                        if (beam_size < kBeamLds) heap_push_wave(heap, beam_size, make_uint4(e_i, lo_i, id_i, 0u), lane);
                        else beam_push_hybrid(heap, beam_size, make_uint4(e_i, lo_i, id_i, 0u), lane);
                        ++beam_size;
                        ++st_push;
                    }
                }
            } else {
                // past the warm-up the threshold only falls: a lower bound that fails it at loop
                // entry fails it at its turn too (:241), so those neighbours are dropped here
                uint32_t m = warmup ? new_mask
                                    : new_mask & (uint32_t)(__ballot(!(lower >= worst0)) & 0xFFFFFFFFull);
                // Every quantity of the reference's loop body is wave-uniform here (neighbour i's values come out
                // of lane i with v_readlane, the heaps live in LDS), so the whole wave walks the loop together and
                // the heap operations are the wave-parallel ones: a rerank costs two LDS round trips per heap
                // operation instead of one dependent round trip per heap level on lane 0.
                while (m) {
                    const int i = __ffs((int)m) - 1;
                    m &= m - 1;
                    const uint32_t id_i = (uint32_t)__builtin_amdgcn_readlane((int)nid, i);
                    const float e = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(est), i));
                    const float lo_i = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lower), i));
                    const float worst = bcast_f32(nnw.lds_key(0));   // FLT_MAX while the heap is empty
                    const float dabs = (nn_size >= k) ? gamma_q * worst : FMAX;
                    // 1 = rerank (result-heap push), 4 = ... and feed the gamma adaptation, 2 = DABS enqueue on the estimate
                    uint32_t act = 0;
                    if (warmup) act = 1;
                    else if (!(lo_i >= worst)) act = (e < worst) ? 5u : ((e < dabs) ? 2u : 0u);
                    act = bcast_u32(act);
                    float key = e, lo = lo_i;
                    bool push = (act & 2u) != 0;
                    if (act & 1u) {
                        const float ex = bcast_f32(s_exact[i]);
                        nn_push_wave(nnw, nn, nn_size, k, id_i, ex, worst, lane);
                        if (ex < dabs) { push = true; key = ex; if (warmup) lo = ex; }
                        if ((act & 4u) && ex > kEpsSmall) {
                            // gamma adaptation (:255-267), fused as the reference compiles it; fp64 on one lane
                            ++ratio_count;
                            if (lane == 0) {
                                double r = (double)(e / ex);
                                const double rs = s_ratio[0] + r;
                                const double rq = fma(r, r, s_ratio[1]);
                                s_ratio[0] = rs;
                                s_ratio[1] = rq;
                                if (ratio_count >= a.sc.gamma_warmup) {
                                    double cnt = (double)ratio_count;
                                    double mean = rs / cnt;
                                    double var = fma(-mean, mean, rq / cnt);
                                    double sd = sqrt(var < 0.0 ? 0.0 : var);
                                    float gq = gamma * (float)fma((double)a.sc.gamma_beta, sd, 1.0);
                                    gamma_q = (gq < gamma) ? gamma
                                                           : ((a.sc.gamma_max < gq) ? a.sc.gamma_max : gq);
                                }
                            }
                            gamma_q = bcast_f32(gamma_q);
                        }
                    }
                    if (bcast_u32(push ? 1u : 0u)) {
                        const uint4 ent = beam_pack(BeamEntry{key, lo, id_i});
#ifdef CPH_TRAFFIC_STATS
                        if (beam_size >= kBeamLds) trf[2]++;
#endif
                        if (beam_size < kBeamLds) heap_push_wave(heap, beam_size, ent, lane);
                        else beam_push_hybrid(heap, beam_size, ent, lane);
                        ++beam_size;
                        ++st_push;
                    }
                }
                CPH_TICK(4);      // (timer builds: the serial replay is booked with the speculative rerank)
            }
            CPH_TICK(5);
            log_count += n_new;
            // one wave per workgroup: LDS accesses of a wave execute in program order, so only the
            // compiler needs a fence here -- an s_barrier would also drain the prefetch (vmcnt)
            __builtin_amdgcn_wave_barrier();
            CPH_TICK(6);
            CPH_TICKF(7);
        }

        // ---- results (:276; src/bindings.cpp:202-210) ----------------------------------
        if (lane == 0 && !overflow) nn_sort(nn, nn_size);
        const uint32_t nn_final = bcast_u32(nn_size);
        __syncthreads();
        if (!overflow) {
            int64_t* out_ids = CPH_COLD(out_ids);
            float* out_dist = CPH_COLD(out_dist);
            for (uint32_t j = lane; j < k; j += 64) {
                if (j < nn_final) {
                    out_ids[(size_t)qi * k + j] = (int64_t)nn[j].id;
                    out_dist[(size_t)qi * k + j] = nn[j].dist;
                } else {
                    out_ids[(size_t)qi * k + j] = -1;
                    out_dist[(size_t)qi * k + j] = FMAX;
                }
            }
        }
        if (lane == 0) {
            CPH_COLD(out_count)[qi] = nn_final;
            // bits 0..7: QueryStatus; bits 8..31: vertices expanded (per-query work, for load analysis)
            CPH_COLD(status)[qi] = (overflow ? kStatusOverflow : kStatusOk) | (st_exp << 8);
            s_tot[0] += st_exp;
            s_tot[1] += st_exact;
            s_tot[2] += st_new;
            s_tot[3] += st_push;
            s_tot[4] += st_skip;
            s_tot[7] += st_allseen;
            if (stage2_redo) s_tot[8] += 1;                       // queries handed to the re-run launch for a stage-2 decision
            s_tot[9] += st_undecided;
#if defined(CPH_PHASE_TIMERS) || defined(CPH_TRAFFIC_STATS)
            unsigned long long* stats = CPH_COLD(stats);
#endif
#ifdef CPH_PHASE_TIMERS
            for (int i = 0; i < 8; ++i) atomicAdd(&stats[8 + i], tph[i]);
#endif
#ifdef CPH_TRAFFIC_STATS
            for (int i = 0; i < 8; ++i) { if (i == 6) atomicMax(&stats[8 + i], trf[i]); else atomicAdd(&stats[8 + i], trf[i]); }
#endif
            if (overflow) {
                s_tot[5] += 1;
                uint32_t* redo = CPH_COLD(redo);
                if (redo) redo[atomicAdd(CPH_COLD(redo_count), 1u)] = qi;
            }
        }
        if (uint32_t* done_flags = CPH_COLD(done_flags); done_flags != nullptr && !overflow) {
            __threadfence_system();      // every lane's result stores, and lane 0's count, before the flag
            if (lane == 0) __hip_atomic_store(&done_flags[qi], CPH_COLD(done_seq), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- clear the estimated set: un-mark the logged ids (or wipe after overflow) --
        __syncthreads();
        if (overflow || wipe) {
            const uint64_t bm_words = CPH_COLD(bm_words);
            for (uint64_t w = lane; w < bm_words; w += 64) bm[w] = 0u;
        } else {
            for (uint32_t j = lane; j < log_count; j += 64) bm[logi[j] >> 5] = 0u;
        }
        __syncthreads();
    }
    // the wave leaves: its totals go to the statistics line
#if defined(CPH_PHASE_TIMERS) || defined(CPH_TRAFFIC_STATS)
    if (lane < 8) {            // (stats[8..15] carry the diagnostic counters in these builds)
#else
    if (lane < 10) {
#endif
        const unsigned long long v = s_tot[lane];
        // (an atomic add of a constant zero is turned into an atomic LOAD of the line by the compiler -- a synchronous
        // round trip; the values here are not constants, and zeros are skipped anyway)
        if (v != 0ull && lane != 6) atomicAdd(&CPH_COLD(stats)[lane], v);
    }
}

}  // namespace cph
