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
#include <hip/hip_runtime.h>
#define FSK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
// dynamic LDS, 16-byte aligned base (cdna_hip_programming.md Guideline 17)
#define FSK_DYN_SHARED(type, name)                                               \
    extern __shared__ __attribute__((aligned(16))) unsigned char fsk_dyn_smem[]; \
    type* name = reinterpret_cast<type*>(fsk_dyn_smem)
#endif

#include <cstdint>

// how many bits of the 64-lane mask (lo, hi) are set in the lanes below the calling one: v_mbcnt_lo + v_mbcnt_hi
#ifdef FSK_EMU
static inline unsigned fsk_mbcnt(unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    return (unsigned)__builtin_popcountll(m & ((1ull << emu::lane_id()) - 1ull));
}
#else
__device__ __forceinline__ unsigned fsk_mbcnt(unsigned lo, unsigned hi) {
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}
#endif

// a 64-bit load from a pointer KNOWN to point into LDS, kept in the LDS address space (ds_read_b64) whatever
// pointer type the surrounding code uses
// (FSK_LDS_VOLATILE_U32: a 32-bit LDS word that is re-read at every use — a counter other lanes of the wave
// add to between two reads. Declared `volatile uint32_t*` it is a GENERIC volatile pointer: every read becomes
// flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0))
#ifdef FSK_EMU
#define FSK_LDS_LOAD_U64(ptr) (*reinterpret_cast<const unsigned long long*>(ptr))
#define FSK_LDS_LOAD_U32(ptr) (*reinterpret_cast<const unsigned int*>(ptr))
#define FSK_LDS_VOLATILE_U32(ptr) (*reinterpret_cast<volatile unsigned int*>(ptr))
#define FSK_LDS_LOAD_U8(ptr) (*reinterpret_cast<const unsigned char*>(ptr))
#else
#define FSK_LDS_LOAD_U8(ptr) (*(const __attribute__((address_space(3))) unsigned char*)(ptr))
#define FSK_LDS_LOAD_U64(ptr) (*(const __attribute__((address_space(3))) unsigned long long*)(ptr))
#define FSK_LDS_LOAD_U32(ptr) (*(const __attribute__((address_space(3))) unsigned int*)(ptr))
#define FSK_LDS_VOLATILE_U32(ptr) (*(volatile __attribute__((address_space(3))) unsigned int*)(ptr))
#endif

typedef unsigned long long u64;
typedef unsigned __int128 u128;  // sort records of the sparse dataflow when k-mer bits + sequence bits exceed 64
