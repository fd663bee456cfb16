// Micro-benchmark: issue rate of v_dot8_u32_u4 against the v_and_b32 + v_bcnt_u32_b32 pair on gfx950.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/dot8_rate.hip -o /tmp/dot8_rate && /tmp/dot8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_dot(unsigned* out, unsigned a0, unsigned b0, int iters) {
    unsigned a = a0 + threadIdx.x, b = b0, c0 = 0, c1 = 1, c2 = 2, c3 = 3;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_udot8(a, b, c0, false);
        c1 = __builtin_amdgcn_udot8(a, b, c1, false);
        c2 = __builtin_amdgcn_udot8(a, b, c2, false);
        c3 = __builtin_amdgcn_udot8(a, b, c3, false);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}
__global__ void k_and(unsigned* out, unsigned a0, unsigned b0, int iters) {
    unsigned a = a0 + threadIdx.x, b = b0, c0 = 0, c1 = 1, c2 = 2, c3 = 3;
    for (int i = 0; i < iters; ++i) {
        c0 += __popc(a & b);
        c1 += __popc(a & (b >> 1));
        asm volatile("" : "+v"(a), "+v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_dot, dim3(grid), dim3(block), 0, 0, d, 3u, 5u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dot8: %.3f ms for %d x 4 dot8 per lane -> %.1f G wave-instr/s\n", ms, iters, 4.0 * iters * grid * block / 64 / ms / 1e6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_and, dim3(grid), dim3(block), 0, 0, d, 3u, 5u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("and+bcnt: %.3f ms for %d x (2 and + 2 bcnt + shift) per lane -> %.1f G wave-instr/s\n", ms, iters, 5.0 * iters * grid * block / 64 / ms / 1e6);
    }
    return 0;
}
