// The memory-access shape of one expansion of the recall-gate workload (100k x 128, 2-bit) and nothing
// else: how many expansions per second the memory system sustains for it, as a ceiling for search_kernel<2,128> there.
// Per iteration a wave makes the two dependent round trips an expansion cannot avoid:
//   trip 1   the vertex block (1,728 B at a random vertex, 16 B per lane), its vector row (512 B), one 744-byte beam page
//            of the wave's own 96-KB page area (the pop's window) and the 12-byte last heap entry
//   trip 2   the estimated-set probe: 32 lanes read one random word each of the wave's own n/8-byte bitmap; a few lanes
//            then mark (atomic OR, no return) and three 12-byte entries are stored into the page area (the pushes)
// The next vertex depends on the loaded data, as in a search.  Nothing is computed.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gate_shape.hip -o /tmp/gate_shape && /tmp/gate_shape [n] [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int kStride = 1728, kPagesBytes = 128 * 768;
__global__ __launch_bounds__(64) void shape_kernel(const unsigned char* __restrict__ blocks, const unsigned char* __restrict__ rows,
                                                   unsigned* __restrict__ bitmaps, unsigned char* __restrict__ pages, unsigned n,
                                                   unsigned bm_words, int iters, int with_probe, unsigned* out) {
    const int lane = threadIdx.x;
    unsigned x = blockIdx.x * 2654435761u + 12345u;
    unsigned acc = 0;
    unsigned* bm = bitmaps + (size_t)blockIdx.x * bm_words;
    unsigned char* pg = pages + (size_t)blockIdx.x * kPagesBytes;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned v = (x >> 8) % n;
        const uint4* b = reinterpret_cast<const uint4*>(blocks + (size_t)v * kStride);
        const uint4 c0 = b[lane];                                       // 1 KB of codes
        uint4 c1 = make_uint4(0, 0, 0, 0), r = c1;
        if (lane < 44) c1 = b[64 + lane];                               // aux + ids + count (704 B)
        if (lane < 32) r = reinterpret_cast<const uint4*>(rows + (size_t)v * 512)[lane];
        const unsigned page = (x >> 3) & 127u;
        unsigned w0 = 0;
        if (lane < 62) w0 = reinterpret_cast<const unsigned*>(pg + page * 768)[3 * lane];   // the window (12-byte entries)
        acc ^= c0.x ^ c1.y ^ r.w ^ w0;
        if (with_probe) {
            // dependent trip: ids come out of the block.  with_probe bits: 1 probe (device-coherent load), 2 marks as atomic OR,
            // 4 marks as plain stores, 8 probe as a plain load instead, 16 the pushes' stores
            unsigned h = (c1.x ^ x ^ (lane * 2246822519u)) * 2654435761u;
            const unsigned word = (h >> 7) % bm_words;
            unsigned bits = 0;
            if (lane < 32) {
                if (with_probe & 8) bits = bm[word];
                else bits = __hip_atomic_load(&bm[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc ^= bits;
            if ((with_probe & 2) && lane < 4) atomicOr(&bm[word], 1u << (h & 31));            // ~3.7 new ids per expansion
            if ((with_probe & 4) && lane < 4) bm[word] = bits | (1u << (h & 31));
            if ((with_probe & 16) && lane < 3) reinterpret_cast<unsigned*>(pg + page * 768)[3 * (lane + 5)] = acc;   // the pushes' stores
        }
        x ^= acc & 1u;
    }
    if (acc == 0x12345678u) out[0] = 1;
}
int main(int argc, char** argv) {
    const unsigned n = argc > 1 ? atoi(argv[1]) : 100000;
    const int wpc_only = argc > 2 ? atoi(argv[2]) : 0;
    const unsigned bm_words = (n + 31) / 32;
    const int max_grid = 256 * 32;
    unsigned char *b, *r, *pg; unsigned *bm, *o;
    hipMalloc(&b, (size_t)n * kStride); hipMalloc(&r, (size_t)n * 512); hipMalloc(&o, 4);
    hipMalloc(&bm, (size_t)max_grid * bm_words * 4); hipMalloc(&pg, (size_t)max_grid * kPagesBytes);
    hipMemset(b, 0, (size_t)n * kStride); hipMemset(r, 0, (size_t)n * 512);
    hipMemset(bm, 0, (size_t)max_grid * bm_words * 4); hipMemset(pg, 0, (size_t)max_grid * kPagesBytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int probe : {0, 1, 1 | 2, 1 | 4, 1 | 8, 1 | 2 | 16, 1 | 4 | 16})
        for (int wpc : {8, 24}) {
            if (wpc_only && wpc != wpc_only) continue;
            const int grid = 256 * wpc;
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(shape_kernel, dim3(grid), dim3(64), 0, 0, b, r, bm, pg, n, bm_words, iters, probe, o);
            hipEventRecord(e0);
            const int reps = 5;
            for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(shape_kernel, dim3(grid), dim3(64), 0, 0, b, r, bm, pg, n, bm_words, iters, probe, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double exps = (double)grid * iters / (ms / reps) / 1e6;      // G expansions / s
            printf("n=%u %2d waves/CU, %s: %.3f ms -> %.3f G expansions/s, %.2f us per expansion and wave, %.0f GB/s algorithmic (2,180 B)\n", n, wpc,
                   probe == 0 ? "block+row+page only" : probe == 1 ? "... then probe (coherent load)" : probe == 3 ? "... probe, atomic marks" :
                   probe == 5 ? "... probe, plain-store marks" : probe == 9 ? "... probe as a plain load" : probe == 19 ? "... probe, atomic marks, push stores" :
                   "... probe, plain-store marks, push stores", ms / reps, exps, grid / exps / 1e3, exps * 2180);
        }
    return 0;
}
