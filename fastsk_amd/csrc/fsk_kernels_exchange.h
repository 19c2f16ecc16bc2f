// fsk_kernels_exchange.h — the kernels of the multi-GPU exchange (fsk_multi.hip only): the narrowing /
// widening copies around an int32 all-reduce of a band of the uint64 triangle, and the engine's own
// all-reduce over peer mappings (FSK_COLL_P2P). What they replace: the K += Ks reduce over the reference's
// worker threads (fastsk_kernel.cpp:286-315), here over the GPUs of one node.
#pragma once
#include "fsk_common.h"

namespace fsk {

constexpr int XC_ITEMS = 4;  // 16-byte pieces per thread and trip

// out[i] = (int32) K[i]: the caller has checked that every cell fits 31 bits. Two cells per 16-byte load.
__global__ __launch_bounds__(256) void k_narrow_u64_i32(const u64* K, int32_t* out, u64 n) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = (int32_t)(uint32_t)K[i];
}
// K[i] = (u64) in[i] (non-negative sums)
__global__ __launch_bounds__(256) void k_widen_i32_u64(const int32_t* in, u64* K, u64 n) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) K[i] = (u64)(uint32_t)in[i];
}

// FSK_COLL_P2P: the buffers of all R engines are mapped into this device (peer access over xGMI, or the
// same device). Rank r sums elements [lo, hi) — its slice, the reduce-scatter — over the R buffers in rank
// order and writes the sum back into every buffer — the all-gather. No other kernel touches [lo, hi) of
// any buffer between the two event fences of the collective, so reading all and writing all in one
// thread is race free. T = int32_t / u64 / double (fp64: the sum is taken in rank order).
struct PeerBufs { void* p[16]; };
template <typename T>
__global__ __launch_bounds__(256) void k_p2p_allreduce(PeerBufs bufs, int R, u64 lo, u64 hi) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = lo + (u64)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) {
        T s = static_cast<const T*>(bufs.p[0])[i];
        for (int q = 1; q < R; ++q) s += static_cast<const T*>(bufs.p[q])[i];
        for (int q = 0; q < R; ++q) static_cast<T*>(bufs.p[q])[i] = s;
    }
}
// the same on 16-byte pieces (V = 4 x int32, 2 x u64, 2 x fp64): what the links like; lo, hi count pieces
template <typename T, int PER>
struct alignas(16) Piece { T v[PER]; };
template <typename T, int PER>
__global__ __launch_bounds__(256) void k_p2p_allreduce_wide(PeerBufs bufs, int R, u64 lo, u64 hi) {
    typedef Piece<T, PER> P;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = lo + (u64)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) {
        P s = static_cast<const P*>(bufs.p[0])[i];
        for (int q = 1; q < R; ++q) {
            const P x = static_cast<const P*>(bufs.p[q])[i];
#pragma unroll
            for (int c = 0; c < PER; ++c) s.v[c] += x.v[c];
        }
        for (int q = 0; q < R; ++q) static_cast<P*>(bufs.p[q])[i] = s;
    }
}


#ifdef FSK_TEST_HOOKS
// Test builds only (tuning fault_kind = 2): keeps one wave busy for about `ms` milliseconds of the constant-rate wall clock
// — an engine whose exchange is late — and then ends by itself: bounded by the clock AND by a trip count, so that
// no value of the clock can leave the GPU hanging.
__global__ __launch_bounds__(64) void k_spin_ms(u64 ms) {
    const u64 t0 = fsk_hw::wall_ticks(), want = ms * fsk_hw::WALL_TICKS_PER_MS;
    for (u64 trip = 0; trip < ms * 4000 + 1000; ++trip) {  // (a sleep of 127 x 64 clocks is ~3.4 us: 4000 trips > 1 ms)
        if (fsk_hw::wall_ticks() - t0 >= want) break;
        fsk_hw::nap();
    }
}
#endif

}  // namespace fsk
