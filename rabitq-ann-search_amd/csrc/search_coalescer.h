// search_coalescer.h — concurrent single-query callers on one handle, gathered into shared launches (host only, no HIP).
//
// The reference answers concurrent search() calls in parallel under a shared lock (src/bindings.cpp:146-175,
// api/hnsw_index.hpp:172).  A GPU answers them best TOGETHER: a caller that finds a free leader slot takes everybody who
// queued up so far with the same k (at most kLeaderGroup) into ONE launch; callers arriving while every slot is in flight
// wait on a condition variable and are gathered by the next leader.  No spinning kernel, no extra thread; a lone caller
// pays one uncontended mutex.  Callers that block on their answers come back together, so a leader whose slot's previous
// launch answered several holds its launch for up to `gather_us` while the queue fills to that size (a lone caller never
// waits).
//
// A launch has two phases.  `launch(slot, group)` enqueues it (may throw: every member then fails with that error);
// after it every member of the group -- the leader too -- calls `wait(slot, index, request)` for ITS OWN answer: a
// launch lasts as long as its longest query, but a caller need not (the search kernel raises a flag per query in
// pinned host memory).  The slot is free again when the last member has its answer.  The policy is separate from what
// a launch IS so that it can be exercised without a GPU: tests/host_san runs it under ThreadSanitizer with a stand-in.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace cph {

constexpr int kLeaderSlots = 4;             // launches of coalesced callers that may be in flight together, at most
constexpr uint64_t kLeaderGroup = 16;       // callers per such launch, at most

// One caller waiting for its answer.
struct SearchReq {
    const float* query = nullptr;
    uint64_t k = 0;        // already clamped to >= 1
    int64_t* ids = nullptr;
    float* dist = nullptr;
    uint64_t* m = nullptr;
    int rc = 0;            // cph_status of the launch that answered (or failed) this caller
    std::string err;
    // set by the leader under the coalescer's mutex
    bool launched = false; // the launch carrying this request is enqueued: wait() may be called
    bool failed = false;   // ... or could not be: rc / err are set
    int slot = -1;
    uint32_t index = 0;    // position inside the launch
};

struct SearchCoalescer {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<SearchReq*> waiting;
    struct Slot {
        bool busy = false;
        size_t pending = 0;        // members of the launch in flight that have not got their answer yet
        size_t last_group = 0;     // callers its previous launch answered
    } slots[kLeaderSlots];
    int gathering = 0;             // leaders holding their launch back for callers that are about to come back
    // Measured on the 1M x 128 benchmark index with 16 / 32 caller threads (profiles/r4_concurrent_search.md): few, large
    // launches -- one launch answers 1 or 16 callers in nearly the same time, while more than three small launches in
    // flight slow each other down.
    int n_slots = 3;
    int gather_us = 80;
    size_t group_bytes_max = 1u << 20;   // result entries (k x callers) one launch may carry

    // Blocks until `r` has its answer.  launch(slot, group): enqueue one launch for `group` (all with the same k); may
    // throw -- status codes 1 = invalid argument, 2 = runtime error, 3 = out of memory (cph_status) then go to every
    // member.  wait(slot, index, req): block until request `index` of the launch in `slot` is answered and copy it out;
    // may throw (that member alone fails).
    template <class Launch, class Wait>
    void submit(SearchReq& r, Launch&& launch, Wait&& wait) {
        std::unique_lock<std::mutex> lk(mu);
        waiting.push_back(&r);
        if (gathering) cv.notify_all();
        while (!r.launched && !r.failed) {
            int slot = -1;
            for (int i = 0; i < n_slots; ++i) if (!slots[i].busy) { slot = i; break; }
            // (our request may already ride in another leader's launch: then there is nothing to lead)
            const bool queued = std::find(waiting.begin(), waiting.end(), &r) != waiting.end();
            if (slot < 0 || !queued) {
                cv.wait(lk);
                continue;
            }
            slots[slot].busy = true;
            if (slots[slot].last_group > 1 && waiting.size() < slots[slot].last_group && gather_us > 0) {
                const size_t want = std::min<size_t>(slots[slot].last_group, kLeaderGroup);
                ++gathering;
                // (system_clock: libstdc++ turns a steady-clock wait into pthread_cond_clockwait, which GCC 11's
                // ThreadSanitizer does not intercept -- it then reports a double lock that is not there; a clock step
                // during an 80-us window at worst ends the window early or late once)
                cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(gather_us),
                              [&] { return waiting.size() >= want; });
                --gathering;
                // (the mutex was released meanwhile: another leader may have taken this caller along)
                if (std::find(waiting.begin(), waiting.end(), &r) == waiting.end()) {
                    slots[slot].busy = false;
                    cv.notify_all();
                    continue;
                }
            }
            std::vector<SearchReq*> group;
            const uint64_t kk = r.k;
            const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(kLeaderGroup, group_bytes_max / kk));
            group.reserve(cap);
            for (auto it = waiting.begin(); it != waiting.end() && group.size() < cap;) {
                if ((*it)->k == kk) { group.push_back(*it); it = waiting.erase(it); } else ++it;
            }
            lk.unlock();
            int rc = 0;
            std::string err;
            try {
                launch(slot, group);
            } catch (const std::invalid_argument& e) { rc = 1; err = e.what();
            } catch (const std::bad_alloc&) { rc = 3; err = "out of memory";
            } catch (const std::exception& e) { rc = 2; err = e.what(); }
            lk.lock();
            slots[slot].last_group = group.size();
            if (rc != 0) {
                for (SearchReq* g : group) { g->rc = rc; g->err = err; g->failed = true; }
                slots[slot].busy = false;
            } else {
                slots[slot].pending = group.size();
                for (uint32_t i = 0; i < group.size(); ++i) { group[i]->slot = slot; group[i]->index = i; group[i]->launched = true; }
            }
            cv.notify_all();
        }
        if (r.failed) return;
        const int slot = r.slot;
        lk.unlock();
        try {
            wait(slot, r.index, r);
        } catch (const std::invalid_argument& e) { r.rc = 1; r.err = e.what();
        } catch (const std::bad_alloc&) { r.rc = 3; r.err = "out of memory";
        } catch (const std::exception& e) { r.rc = 2; r.err = e.what(); }
        lk.lock();
        if (--slots[slot].pending == 0) {      // the launch's pinned buffer is nobody's any more
            slots[slot].busy = false;
            cv.notify_all();
        }
    }
};

}  // namespace cph
