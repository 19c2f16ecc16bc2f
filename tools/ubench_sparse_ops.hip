// tools/ubench_sparse_ops.hip — issue rates of the instructions the SPARSE kernels are made of (k_sx_emit / scatter /
// seg_write / consume): v_and_b32, v_add_u32, v_lshl_add_u32, v_bfe_u32 / v_bfe_i32, v_cmp + v_cndmask, v_bitop3_b32,
// v_mbcnt_lo/hi, v_readlane, v_max_i32_dpp, s_and_b64 (SALU beside VALU), ds_add_u32, ds_read_b32 / b64, ds_write_b32,
// at 1 / 2 / 4 / 6 / 8 waves a SIMD, 16 independent chains a lane. Reports cycles per wave64 instruction on one SIMD
// (4 = a quarter-rate 16-lane issue, 2 = the full 32-lane rate) so that SQ_INSTS_VALU can be priced by measurement.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sparse_ops.hip -o tools/ubench_sparse_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Op { AND, ADD, LSHL_ADD, BFE_U, BFE_I, CMP_CND, BITOP3, MBCNT, READLANE, MAX_DPP, MAD24, XOR_SALU, DS_ADD, DS_READ32, DS_READ64, DS_WRITE32,
          DS_ADD_RANDOM, N_OPS };
static const char* const NAMES[N_OPS] = {"v_and_b32", "v_add_u32", "v_lshl_add_u32", "v_bfe_u32", "v_bfe_i32", "v_cmp_lt_u32+v_cndmask", "v_bitop3_b32",
                                          "v_mbcnt_lo+hi", "v_readlane_b32", "v_max_i32_dpp row_shr:1", "v_mad_u32_u24", "v_xor_b32 + s_and_b64",
                                          "ds_add_u32 (lane-linear)", "ds_read_b32 (lane-linear)", "ds_read_b64 (lane-linear)", "ds_write_b32 (lane-linear)",
                                          "ds_add_u32 (random bank)"};
// instructions per chain step (for the per-instruction price)
static const int INSTS[N_OPS] = {1, 1, 1, 1, 1, 2, 1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a0, unsigned b0, int iters) {
    __shared__ unsigned lds[256 * 4 + 64];
    unsigned acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 2654435761u + i * 40503u + a0;
    unsigned b = b0 + (threadIdx.x & 7), c = 5;
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) lds[i] = i;
    __syncthreads();
    const unsigned la = threadIdx.x * 4;                              // lane-linear LDS byte address: conflict-free
    const unsigned lr = ((threadIdx.x * 2654435761u) >> 22) & 0xffcu;  // pseudo-random dword inside 4 KB
    unsigned long long sm = 0x5555555555555555ull;
    for (int it = 0; it < iters; ++it) {
#define STEP(i)                                                                                                                         \
    if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));                                                        \
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));                                                        \
    if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(b));                                           \
    if (OP == BFE_U) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(acc[i]));                                                            \
    if (OP == BFE_I) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(acc[i]));                                                             \
    if (OP == CMP_CND) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(acc[i]) : "v"(b), "v"(c) : "vcc"); \
    if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(acc[i]) : "v"(b), "v"(c));                          \
    if (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n v_mbcnt_hi_u32_b32 %0, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c));       \
    if (OP == READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(acc[i])); sm ^= s; }                       \
    if (OP == MAX_DPP) asm volatile("s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));          \
    if (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(b), "v"(c));                                      \
    if (OP == XOR_SALU) asm volatile("v_xor_b32 %0, %0, %2\n s_and_b64 %1, %1, exec" : "+v"(acc[i]), "+s"(sm) : "v"(b));                 \
    if (OP == DS_ADD) asm volatile("ds_add_u32 %0, %1" ::"v"(la), "v"(acc[i]) : "memory");                                              \
    if (OP == DS_ADD_RANDOM) asm volatile("ds_add_u32 %0, %1" ::"v"(lr), "v"(acc[i]) : "memory");                                       \
    if (OP == DS_READ32) asm volatile("ds_read_b32 %0, %1" : "=v"(acc[i]) : "v"(la) : "memory");                                        \
    if (OP == DS_READ64) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(la * 2u & 0xff8u) : "memory"); acc[i] = (unsigned)v; } \
    if (OP == DS_WRITE32) asm volatile("ds_write_b32 %0, %1" ::"v"(la), "v"(acc[i]) : "memory");
        REP16(STEP)
#undef STEP
        if (OP >= DS_ADD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned s = (unsigned)sm ^ (unsigned)(sm >> 32);
    for (int i = 0; i < 16; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
}

template <int OP>
void run(unsigned* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {1, 2, 4, 6, 8}) {
        const int grid = 256 * wg_per_cu, iters = OP >= DS_ADD ? 4000 : 10000;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, 5u, 50);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, 5u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        // wave-instructions issued on one SIMD: wg_per_cu waves a SIMD x 16 chains x iters x INSTS
        const double winst = (double)wg_per_cu * 16.0 * iters * INSTS[OP];
        const double cyc = best * 1e-3 * 2.4e9 / winst;  // at the 2.4 GHz peak clock (tools/ubench_clock.hip measures the real one)
        printf("%-28s %d waves/SIMD: %8.3f ms  %6.2f cycles per wave64 instruction and SIMD  (%5.1f lanes/clk/CU)\n", NAMES[OP], wg_per_cu, best, cyc,
               4.0 * 64.0 / cyc);
    }
}

int main() {
    unsigned* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<AND>(out); run<ADD>(out); run<LSHL_ADD>(out); run<BFE_U>(out); run<BFE_I>(out); run<CMP_CND>(out); run<BITOP3>(out); run<MBCNT>(out);
    run<READLANE>(out); run<MAX_DPP>(out); run<MAD24>(out); run<XOR_SALU>(out); run<DS_ADD>(out); run<DS_READ32>(out); run<DS_READ64>(out);
    run<DS_WRITE32>(out); run<DS_ADD_RANDOM>(out);
    hipFree(out);
    return 0;
}
