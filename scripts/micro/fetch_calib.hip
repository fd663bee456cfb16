// What FETCH_SIZE reports for scattered accesses on gfx950 (MI355X_MICROARCH.md section HBM calibrates only wide
// coalesced reads: FETCH_SIZE = half of the bytes).  Every kernel below touches a known number of distinct, randomly
// placed 128-byte lines of a 4 GiB table (far beyond the 256 MiB Infinity Cache), each line once:
//   scatter4    every lane one dword of its own line           (64 lines per wave instruction)
//   scatter16   every lane 16 bytes of its own line
//   line128     32 lanes x 4 B = one whole line per wave instruction
//   line64      16 lanes x 4 B = the first half of a line per wave instruction
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <out> -o p -- /tmp/fetch_calib      (then TCC_EA0_RDREQ etc.)
// The program prints the lines each kernel touched; divide the counter by it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr size_t kLines = (size_t)4 << (30 - 7);   // 4 GiB of 128-byte lines
__device__ __forceinline__ size_t pick(unsigned long long x) {
    // an odd multiplier modulo a power of two is a bijection: every line at most once per kernel
    return (size_t)((x & (kLines - 1)) * 0x9E3779B97F4A7C15ull) & (kLines - 1);
}
__global__ __launch_bounds__(256) void scatter4(const unsigned* t, int iters, unsigned* out) {
    unsigned acc = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) acc ^= t[pick(tid * iters + i) * 32 + (threadIdx.x & 31)];
    if (acc == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void scatter16(const uint4* t, int iters, unsigned* out) {
    unsigned acc = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) acc ^= t[pick(tid * iters + i) * 8 + (threadIdx.x & 7)].x;
    if (acc == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void line128(const unsigned* t, int iters, unsigned* out) {
    unsigned acc = 0;
    const unsigned long long wid = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < iters; ++i)
        if ((threadIdx.x & 63) < 32) acc ^= t[pick(wid * iters + i) * 32 + (threadIdx.x & 31)];
    if (acc == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void line64(const unsigned* t, int iters, unsigned* out) {
    unsigned acc = 0;
    const unsigned long long wid = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < iters; ++i)
        if ((threadIdx.x & 63) < 16) acc ^= t[pick(wid * iters + i) * 32 + (threadIdx.x & 15)];
    if (acc == 0x12345678u) out[0] = 1;
}
int main() {
    unsigned* t; unsigned* o;
    if (hipMalloc(&t, kLines * 128) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0, kLines * 128);
    hipDeviceSynchronize();
    const int grid = 2048, block = 256;
    {   const int iters = 16;    // 2048*256*16 = 8.4 M lines (1 GiB of lines)
        hipLaunchKernelGGL(scatter4, dim3(grid), dim3(block), 0, 0, t, iters, o);
        printf("scatter4  lines=%zu\n", (size_t)grid * block * iters);
        hipLaunchKernelGGL(scatter16, dim3(grid), dim3(block), 0, 0, (const uint4*)t, iters, o);
        printf("scatter16 lines=%zu\n", (size_t)grid * block * iters); }
    {   const int iters = 512;   // 2048*4 waves*512 = 4.2 M lines
        hipLaunchKernelGGL(line128, dim3(grid), dim3(block), 0, 0, t, iters, o);
        printf("line128   lines=%zu\n", (size_t)grid * (block / 64) * iters);
        hipLaunchKernelGGL(line64, dim3(grid), dim3(block), 0, 0, t, iters, o);
        printf("line64    lines=%zu\n", (size_t)grid * (block / 64) * iters); }
    hipDeviceSynchronize();
    return 0;
}
