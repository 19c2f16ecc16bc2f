// tools/ubench_clock.hip — what is the shader clock while every SIMD issues v_dot8_u32_u4 back to
// back for about a second (the tile kernel's regime)? Each workgroup brackets its loop with
// s_memtime (shader cycles) and s_memrealtime (constant 100 MHz): cycles / time = the clock under
// load; lane-ops / cycles = the issue rate per clock, independent of the clock assumed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// 8 x 8 accumulators and 4 x 64 dot8 per loop trip: the tile kernel's inner-loop shape without its LDS reads
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* clk, unsigned a0, unsigned b0, int iters) {
    unsigned acc[64];
    unsigned a[8], b[8];
    for (int i = 0; i < 64; ++i) acc[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i * 0x01010101u + threadIdx.x; b[i] = b0 + i * 0x00010001u; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r * 8 + c] = __builtin_amdgcn_udot8(a[r], b[c], acc[r * 8 + c], false);
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned s = 0;
    for (int i = 0; i < 64; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    for (int wg_per_cu : {1, 2, 3, 4}) {
        const int grid = 256 * wg_per_cu;
        unsigned* out; unsigned long long* clk;
        hipMalloc(&out, (size_t)grid * 256 * 4);
        hipMalloc(&clk, (size_t)grid * 16);
        const int iters = 2000000 / wg_per_cu;   // ~1 s per launch
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, 3u, 5u, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, 3u, 5u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid * 2);
        hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double mhz = 0;
        for (int b = 0; b < grid; ++b) {
            mhz += (double)h[2 * b] / ((double)h[2 * b + 1] / 100.0);          // cycles per microsecond
        }
        mhz /= grid;
        const double ops = (double)grid * 256 * 256.0 * iters;
        const double lanes_per_clk_cu = ops / (ms * 1e-3) / 256.0 / (mhz * 1e6);  // at the measured clock
        printf("v_dot8_u32_u4 %d waves/SIMD, %.0f ms: shader clock %.0f MHz under load, %.1f lanes/clk/CU, %.1f T MAC/s\n", wg_per_cu, ms, mhz,
               lanes_per_clk_cu, ops * 8 / (ms * 1e-3) / 1e12);
        hipFree(out); hipFree(clk);
    }
    return 0;
}
