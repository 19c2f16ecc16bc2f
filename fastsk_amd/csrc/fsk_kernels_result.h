// fsk_kernels_result.h — what happens to the integer triangle after the accumulation: raw diagonal, cosine
// normalisation and the getters (fastsk_kernel.cpp:96-103, fastsk.cpp:190-217). Included by fsk_engine.hip only.
#pragma once
#include "fsk_common.h"

namespace fsk {

// =============================================================================================
// NORMALISATION / GETTERS  (fastsk_kernel.cpp:96-103, fastsk.cpp:190-217)
// IEEE fp64 multiply, correctly rounded sqrt and divide, no contraction.
// =============================================================================================
template <typename SrcT>
__global__ __launch_bounds__(256) void k_diag(const SrcT* K, double* diag, uint32_t N) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < N) diag[i] = (double)K[tri_index(i, i)];
}

__device__ __forceinline__ double normalised_cell(double x, double di, double dj, bool is_diag) {
    // off-diagonal: K_ij / sqrt(K_ii * K_jj) with the RAW diagonals; diagonal: K_ii / sqrt(K_ii*K_ii)
    const double prod = is_diag ? __dmul_rn(x, x) : __dmul_rn(di, dj);
    return __ddiv_rn(x, __dsqrt_rn(prod));
}

template <typename SrcT>
__global__ __launch_bounds__(256) void k_block(const SrcT* K, const double* diag, u64 i0, u64 rows, u64 j0, u64 cols,
                                               double* out) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    if (c >= rows * cols) return;
    const u64 i = i0 + c / cols, j = j0 + c % cols;
    const u64 a = i > j ? i : j, b = i > j ? j : i;
    const double x = (double)K[tri_index(a, b)];
    out[c] = normalised_cell(x, diag[a], diag[b], a == b);
}

template <typename SrcT>
__global__ __launch_bounds__(256) void k_block_raw(const SrcT* K, u64 i0, u64 rows, u64 j0, u64 cols, SrcT* out) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    if (c >= rows * cols) return;
    const u64 i = i0 + c / cols, j = j0 + c % cols;
    const u64 a = i > j ? i : j, b = i > j ? j : i;
    out[c] = K[tri_index(a, b)];
}

// arbitrary cells (rows[q], cols[q]) of the symmetric matrix: scattered parity checks at sizes where
// no block of the triangle can be compared whole
__global__ __launch_bounds__(256) void k_cells_raw(const u64* K, const int64_t* rows, const int64_t* cols, u64 n, u64* out) {
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u64 i = (u64)rows[q], j = (u64)cols[q];
    out[q] = K[i > j ? tri_index(i, j) : tri_index(j, i)];
}

// whole triangle, cells [c0, c0+count) of the reference layout (fastsk_kernel.cpp:96-103 over every cell). A thread
// takes TR_ITEMS cells 256 apart (coalesced 8-byte loads and stores): the row of its first cell comes from one
// fp64 square root (corrected to the exact integer), the later ones from stepping the column by 256 and carrying
// into the row — no division or root per cell. 16 bytes of HBM per cell, the diagonal gathers stay in L2.
constexpr int TR_ITEMS = 16;
template <typename SrcT>
__global__ __launch_bounds__(256) void k_triangle(const SrcT* K, const double* diag, u64 c0, u64 count, double* out) {
    const u64 base = (u64)blockIdx.x * (256 * TR_ITEMS) + threadIdx.x;
    if (base >= count) return;
    const u64 c = c0 + base;
    u64 i = (u64)((sqrt(8.0 * (double)c + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > c) --i;
    while ((i + 1) * (i + 2) / 2 <= c) ++i;
    u64 j = c - i * (i + 1) / 2;
#pragma unroll 4
    for (int q = 0; q < TR_ITEMS; ++q) {
        const u64 t = base + (u64)q * 256;
        if (t >= count) break;
        const double x = (double)K[c0 + t];
        out[t] = normalised_cell(x, diag[i], diag[j], i == j);
        j += 256;
        while (j > i) { j -= i + 1; ++i; }
    }
}


// Order-free digest of count cells [c0, c0 + count): out[0] += sum of the cells, out[1] ^= xor of
// cell * (index | 1), both mod 2^64 (fsk_counts_digest). Streams the cells once, 16 bytes per load.
constexpr int DG_ITEMS = 16;  // cells per thread and trip
__global__ __launch_bounds__(256) void k_digest(const u64* K, u64 c0, u64 count, u64* out) {
    u64 s = 0, x = 0;
    const u64 stride = (u64)gridDim.x * 256 * DG_ITEMS;
    for (u64 base = (u64)blockIdx.x * 256 * DG_ITEMS; base < count; base += stride) {
#pragma unroll
        for (int q = 0; q < DG_ITEMS; ++q) {
            const u64 c = base + (u64)q * 256 + threadIdx.x;
            if (c < count) {
                const u64 v = K[c0 + c];
                s += v;
                x ^= v * ((c0 + c) | 1ull);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d);
        x ^= __shfl_xor(x, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (s) atomicAdd(&out[0], s);
        if (x) atomicXor(&out[1], x);
    }
}

}  // namespace fsk
