// fsk_platform.h — the one place that names the runtime the kernels are compiled against.
//
// Product build: hipcc --offload-arch=gfx950, <hip/hip_runtime.h>.
// Test build (tests/emu, -DFSK_EMU): the same kernel source against tests/emu/hip_emu.h so that
// `pytest -m "not gpu"` can exercise kernel logic without a GPU. The emulated library is never
// loaded by the fastsk_amd package.
#pragma once

#ifdef FSK_EMU
#include "hip_emu.h"
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#else
#include "fsk_gfx950.h"
#define FSK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
// dynamic LDS, 16-byte aligned base (cdna_hip_programming.md Guideline 17)
#define FSK_DYN_SHARED(type, name)                                               \
    extern __shared__ __attribute__((aligned(16))) unsigned char fsk_dyn_smem[]; \
    type* name = reinterpret_cast<type*>(fsk_dyn_smem)
#endif

#include <cstdint>

typedef unsigned long long u64;
typedef unsigned __int128 u128;  // sort records of the sparse dataflow when k-mer bits + sequence bits exceed 64
