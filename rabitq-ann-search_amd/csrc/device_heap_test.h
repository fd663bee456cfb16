// device_heap_test.h — self-test hook for the beam's heap routines (device_search.h): a sequence of pushes and pops on
// one wave, with the first kBeamLds entries in LDS and the rest in HBM exactly as the search kernel keeps its beam,
// dispatching to heap_push_wave / beam_push_hybrid / heap_pop_wave / beam_pop_hybrid the way the kernel does.  The test
// (tests/test_gpu_parity.py) replays the same sequence with libstdc++'s std::push_heap / std::pop_heap and compares
// the heap arrays element for element -- sizes, tie patterns and LDS / HBM boundary cases the search fixtures only
// reach by chance.  Test infrastructure; not on the product path.
#pragma once
#include <hip/hip_runtime.h>

#include "device_search.h"

namespace cph {

struct HeapTestArgs {
    const uint8_t* ops;      // [n_ops] 1 = push the next (key, id), 0 = pop, 2 = pop, then push the next (key, id): what one
                             // expansion of the search does to its beam
    const float* keys;       // [pushes]
    const uint32_t* ids;     // [pushes]
    uint32_t n_ops;
    uint32_t* pages;         // [kBeamPagesDwords] HBM part of the heap, levels 8..12
    uint32_t* tail;          // [beam_tail_dwords(cap)] levels 13+
    float* out_keys;         // [final size]
    uint32_t* out_ids;
    uint32_t* out_size;
};

__global__ __launch_bounds__(64) void heap_selftest_kernel(HeapTestArgs a) {
    __shared__ __align__(16) unsigned char lds[(kBeamLds + 1) * 16];
    const int lane = threadIdx.x;
    Beam heap;
    heap.l = (lds_u32x4*)lds;
    heap.gp = a.pages;
    heap.gt = a.tail;
    uint32_t size = 0, next = 0;
    for (uint32_t j = 0; j < a.n_ops; ++j) {
        if (a.ops[j] == 2 && size > 0) {
            if (size > 1) {
                if (size <= kBeamLds) heap_pop_wave(heap, size, lane);
                else beam_pop_hybrid(heap, size, lane);
            }
            --size;
            __builtin_amdgcn_wave_barrier();
        }
        if (a.ops[j]) {
            const uint4 v = make_uint4(__float_as_uint(a.keys[next]), 0u, a.ids[next], 0u);
            ++next;
            if (size < kBeamLds) heap_push_wave(heap, size, v, lane);
            else beam_push_hybrid(heap, size, v, lane);
            ++size;
        } else if (size > 0) {
            if (size > 1) {
                if (size <= kBeamLds) heap_pop_wave(heap, size, lane);
                else beam_pop_hybrid(heap, size, lane);
            }
            --size;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (uint32_t i = lane; i < size; i += 64) {
        const uint4 e = heap.raw(i);
        a.out_keys[i] = __uint_as_float(e.x);
        a.out_ids[i] = e.z;
    }
    if (lane == 0) *a.out_size = size;
}

}  // namespace cph
