// tools/ubench_mfma_i8.hip — NOT part of the product path. BASELINE.json's north star rules MFMA out
// for this engine ("integer counting, not a dense contraction"); the dense dataflow the counting turned
// into IS a Gram product of count panels, so this measures what that rule costs on MI355X: the issue
// rate of v_mfma_i32_32x32x32_i8 (exact int32 accumulation of u8/i8 counts, counts <= 127 per plane)
// next to v_dot8_u32_u4, the instruction k_dense_tile_dma is built on.
// A count panel widened from 4-bit to 8-bit planes doubles the panel bytes (12.7 -> 25.4 GB at config 5,
// still L2/HBM-cheap next to the multiply time), so the MAC rate is what decides.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 4 independent 32x32 accumulator tiles per wave (64 AGPR/VGPR accumulators), back-to-back MFMAs
__global__ __launch_bounds__(256) void k_mfma(int* out, int a0, int b0, int iters) {
    v16i acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = (int)threadIdx.x + i;
    v4i a = {a0, a0 + 1, a0 + 2, a0 + 3}, b = {b0, b0 + 5, b0 + 7, b0 + 11};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
    }
    int s = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s ^= acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_dot8(unsigned* out, unsigned a0, unsigned b0, int iters) {
    unsigned acc[64], a[8], b[8];
    for (int i = 0; i < 64; ++i) acc[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i * 0x01010101u + threadIdx.x; b[i] = b0 + i * 0x00010001u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[r * 8 + c] = __builtin_amdgcn_udot8(a[r], b[c], acc[r * 8 + c], false);
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    }
    unsigned s = 0;
    for (int i = 0; i < 64; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int* out;
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    for (int wg_per_cu : {1, 2, 4}) {
        const int grid = 256 * wg_per_cu;
        const int it_m = 200000 / wg_per_cu, it_d = 400000 / wg_per_cu;
        const double ms_m = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, 3, 5, it_m); });
        const double ms_d = time_ms([&] { hipLaunchKernelGGL(k_dot8, dim3(grid), dim3(256), 0, 0, (unsigned*)out, 3u, 5u, it_d); });
        // one 32x32x32 MFMA = 32768 MACs per wave; one dot8 = 8 MACs per lane = 512 per wave
        const double mac_m = (double)grid * 4 /*waves*/ * it_m * 16.0 * 32768.0;
        const double mac_d = (double)grid * 4 * it_d * 64.0 * 512.0;
        printf("%d waves/SIMD: v_mfma_i32_32x32x32_i8 %.1f T MAC/s (%.0f ms) | v_dot8_u32_u4 %.1f T MAC/s (%.0f ms) | ratio %.2f\n", wg_per_cu,
               mac_m / (ms_m * 1e-3) / 1e12, ms_m, mac_d / (ms_d * 1e-3) / 1e12, ms_d, (mac_m / ms_m) / (mac_d / ms_d));
    }
    hipFree(out);
    return 0;
}
