// Achievable HBM bandwidth for the search kernel's access shape: every wave reads whole, randomly chosen
// 2752-byte vertex blocks (plus a 512-byte vector row from a second array), 16 B per lane, nothing else.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/hbm_gather.hip -o /tmp/hbm_gather && /tmp/hbm_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int kStride = 2752;
__global__ __launch_bounds__(64) void gather_kernel(const unsigned char* __restrict__ blocks,
                                                    const unsigned char* __restrict__ rows, unsigned n,
                                                    int iters, unsigned* out) {
    const int lane = threadIdx.x;
    unsigned x = blockIdx.x * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned v = (x >> 8) % n;
        const uint4* b = reinterpret_cast<const uint4*>(blocks + (size_t)v * kStride);
        const uint4 c0 = b[lane], c1 = b[64 + lane];                 // 2 KB of codes
        uint4 c2 = make_uint4(0, 0, 0, 0);
        if (lane < 44) c2 = b[128 + lane];                            // aux + ids + count (704 B)
        uint4 r = make_uint4(0, 0, 0, 0);
        if (lane < 32) r = reinterpret_cast<const uint4*>(rows + (size_t)v * 512)[lane];
        acc.x ^= c0.x ^ c1.y ^ c2.z ^ r.w;
        x ^= acc.x & 1u;                                              // next vertex depends on the data (a chain, as in a search)
    }
    if (acc.x == 0x12345678u) out[0] = 1;
}
int main(int argc, char** argv) {
    const unsigned n = 1000000;
    unsigned char *b, *r; unsigned* o;
    hipMalloc(&b, (size_t)n * kStride); hipMalloc(&r, (size_t)n * 512); hipMalloc(&o, 4);
    hipMemset(b, 0, (size_t)n * kStride); hipMemset(r, 0, (size_t)n * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    for (int wpc : {8, 16, 24, 32, 40}) {
        const int grid = 256 * wpc;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, b, r, n, iters, o);
        hipEventRecord(e0);
        const int reps = 5;
        for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, b, r, n, iters, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * iters * (2752 + 512);
        printf("%2d waves/CU, dependent random vertex reads: %.3f ms -> %.0f GB/s, %.2f G vertices/s\n", wpc, ms / reps,
               bytes / (ms / reps) / 1e6, (double)grid * iters / (ms / reps) / 1e6);
    }
    return 0;
}
