// fsk_common.h — shared by every kernel header of the engine and by its host code: the packed-sequence view,
// triangle indexing, block scans and the count-panel geometry. The kernels themselves live in
// fsk_kernels_dense.h (count panels + tile accumulate), fsk_kernels_sparse.h (sort -> segments -> update
// streams), fsk_kernels_variance.h (Welford + exact sequential sum) and fsk_kernels_result.h (normalise,
// getters, digests, the narrowing copies of the multi-GPU exchange); each is included by exactly one
// translation unit of libfastsk_amd.so.
//
// What the reference does per mismatch combination (fastsk_kernel.cpp:216-241, shared.cpp):
//   gather kept positions -> stable LSD counting sort (cntsrtna, shared.cpp:156-191)
//   -> permute -> run-length co-occurrence count into the triangle (countAndUpdateTri,
//   shared.cpp:268-333).
// Here the same mathematics, K[i][j] += sum_v cnt_i(v) * cnt_j(v), is computed by one of two
// dataflows, both integer VALU/LDS/atomic work (no MFMA):
//
//   DENSE  (small key space, e.g. DNA with k=4: 256 keys, every key present in most sequences)
//     k_dense_count  per-sequence counting sort in LDS: 64 sequences per workgroup, symbols
//                    unpacked once from bit-packed HBM, one LDS atomic per g-mer; the segment
//                    counts leave as 4-bit "count panels" (count = lo + 16*hi, two nibble planes)
//                    laid out [panel][combo][key/8][64 seqs] so that the tile kernel streams
//                    them with 16-byte coalesced loads.
//     k_dense_tile_dma  output-stationary 128x128 tile of K per workgroup: panels DMA'd into
//                    LDS, 8x8 register block per lane, v_dot8_u32_u4 multiply-adds summed in
//                    registers over ALL combos of the launch, then ONE 64-bit atomicAdd per cell.
//
//   SPARSE (large key space, e.g. protein: 24^4 keys, runs of 3-5) — the reference's dataflow
//     k_sx_extract      packed record (k-mer << sb | sequence id) per g-mer, straight from packed HBM
//     k_sx_hist/scatter LDS-staged 8-bit LSD radix sort, one independent sort per combo of the batch
//                       (wave64 ballot match ranking, stable)
//     k_sx_seg_*        run heads by neighbour compare, block prefix sums -> distinct (k-mer,seq)
//                       entries with multiplicities and ranks inside their run
//     k_sx_emit + k_sx_consume  every (run, pair) update becomes a 32-bit word in the stream of the
//                       workgroup that owns the rows; the owner sums its stream in LDS and adds the
//                       non-zero cells into the 64-bit triangle (DIRECT: 64-bit atomicAdd per pair)
//
// Everything is written for 64-wide wavefronts; lane = threadIdx.x & 63.
#pragma once
#include "fsk_platform.h"

namespace fsk {

struct SeqView {
    const uint32_t* words;   // bit-packed symbols, every sequence starts on a 32-bit word
    const uint32_t* wstart;  // [n_seq] first word of sequence i
    const uint32_t* len;     // [n_seq] length in symbols
    uint32_t n_seq;
    int bits;                // 2, 4 or 8 bits per symbol (a symbol never straddles a word)
};

__device__ __forceinline__ uint32_t fetch_sym(const uint32_t* words, uint32_t wbase, uint32_t pos, int bits) {
    uint32_t bitpos = pos * (uint32_t)bits;
    return (words[wbase + (bitpos >> 5)] >> (bitpos & 31u)) & ((1u << bits) - 1u);
}

// a / b for a divisor shared by many dividends, rb = 1 / b (correctly rounded, computed once): q = RN(a rb) is within an
// ulp of a / b, the remainder a - q b is exact in one FMA, and RN(q + rem rb) is the correctly rounded quotient
// (Markstein's theorem; it needs b's significand not to be all ones — b is an iteration number here — and no
// under/overflow on the way: the dividends are differences of counts and their means). Three full-rate FMA-class
// instructions instead of the dozen of a division with its quarter-rate reciprocal; bit for bit the IEEE quotient.
__device__ __forceinline__ double div_by_shared(double a, double b, double rb) {
    const double q = __dmul_rn(a, rb);
    const double rem = __fma_rn(-q, b, a);
    return __fma_rn(rem, rb, q);
}

__device__ __forceinline__ u64 tri_index(u64 i, u64 j) {  // j <= i  (tri_access, shared.cpp:97-117)
    return i * (i + 1) / 2 + j;
}

// ---------------------------------------------------------------------------------------------
// block-wide exclusive scan of one value per thread, 256 threads (4 waves). tmp: >= 4 entries.
// Every thread of the block must call it.
template <typename T>
__device__ __forceinline__ T block_excl_scan_256(T v, T* tmp, T* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();  // tmp may still be read from a previous call
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        T t = tmp[w];
        if (w < wave) base += t;
        tot += t;
    }
    if (total) *total = tot;
    return base + x - v;
}

// the same for a block of NW waves (NW * 64 threads). tmp: >= NW entries.
template <typename T, int NW>
__device__ __forceinline__ T block_excl_scan(T v, T* tmp, T* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        T t = tmp[w];
        if (w < wave) base += t;
        tot += t;
    }
    if (total) *total = tot;
    return base + x - v;
}

constexpr int PANEL = 64;        // sequences per count panel (= one wave of lanes)
constexpr int TILE = 128;        // K tile edge (2 panels)
constexpr int STAGE_KQ = 32;     // key quads (4 keys = one dword of u8 counts) per LDS stage

// Where sequence r (0..63) of a panel sits inside each 64-dword panel row. Interleaving by 16
// makes the four dwords a lane fetches with one ds_read_b128 belong to sequences t, t+16, t+32,
// t+48, so that for a fixed register the 16 lanes of a row group own 16 CONSECUTIVE columns of
// K and the flush atomics of a wave fall into 128-byte contiguous segments.
__device__ __forceinline__ uint32_t panel_slot(uint32_t r) { return ((r & 15u) << 2) | (r >> 4); }
// tile-local row/column (0..127) of register e (0..7) of lane group t (0..15): inverse of the above
__device__ __forceinline__ uint32_t tile_index(uint32_t t, uint32_t e) { return (e >> 2) * 64u + t + 16u * (e & 3u); }

// a * b + c with 24-bit operands: v_mad_u32_u24 issues at full rate, a 32-bit multiply-add does
// not (the compiler turns __umul24 of small known ranges back into one, hence the asm).
// b is wave-uniform.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return fsk_hw::mad24(a, b, c); }

}  // namespace fsk
