// tools/ubench_valu.hip — measures the integer-VALU issue rates that bound k_dense_tile:
// v_dot4_u32_u8, v_dot8_u32_u4, v_mad_u32_u24, v_pk_mad_u16 and v_fma_f32 for reference,
// at 1/2/4 waves per SIMD with 16 independent accumulator chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a0, unsigned b0, int iters) {
    unsigned acc[16];
    unsigned a[4], b[4];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
    for (int i = 0; i < 4; ++i) { a[i] = a0 + i * 0x01010101u + threadIdx.x; b[i] = b0 + i * 0x00010001u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned& x = acc[r * 4 + c];
                if (OP == 0) x = __builtin_amdgcn_udot4(a[r], b[c], x, false);
                if (OP == 1) x = __builtin_amdgcn_udot8(a[r], b[c], x, false);
                if (OP == 2) x = (a[r] & 0xffffffu) * (b[c] & 0xffffffu) + x;
                if (OP == 3) { float f = __builtin_fmaf(__uint_as_float(a[r]), __uint_as_float(b[c]), __uint_as_float(x)); x = __float_as_uint(f); }
                if (OP == 4) x = __builtin_amdgcn_udot2(*(reinterpret_cast<__attribute__((ext_vector_type(2))) unsigned short*>(&a[r])),
                                                        *(reinterpret_cast<__attribute__((ext_vector_type(2))) unsigned short*>(&b[c])), x, false);
            }
        // keep the operands moving so nothing is hoisted
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    }
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int macs_per_op) {
    unsigned* out;
    hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {1, 2, 4, 8}) {
        int grid = 256 * wg_per_cu;
        int iters = 20000;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, 5u, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, 5u, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)grid * 256 * 16.0 * iters;
        double lanes_per_clk_cu = ops / (ms * 1e-3) / 256.0 / 2.4e9;
        printf("%-16s %d waves/SIMD: %8.3f ms  %7.2f T lane-ops/s  %7.2f T MAC/s  (%.1f lanes/clk/CU at 2.4 GHz)\n", name, wg_per_cu,
               ms, ops / (ms * 1e-3) / 1e12, ops * macs_per_op / (ms * 1e-3) / 1e12, lanes_per_clk_cu);
    }
    hipFree(out);
}

int main() {
    run<0>("v_dot4_u32_u8", 4);
    run<1>("v_dot8_u32_u4", 8);
    run<2>("v_mad_u32_u24", 1);
    run<3>("v_fma_f32", 1);
    run<4>("v_dot2_u32_u16", 2);
    return 0;
}
