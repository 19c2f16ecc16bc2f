// fsk_engine_dense_small.hip — the DENSE dataflow's tile accumulate at SMALL N (configs 2 and 3, the sizes the reference's own
// CI runs: test/run_check.py:45), countAndUpdateTri (shared.cpp:268-333) for a few hundred tiles of K.
//
// With few tiles the combo range of a launch is split over several workgroups a tile (fsk_engine_dense.hip), and every one of
// them used to flush its 128 x 128 partial sums with one 64-bit atomicAdd a cell: config 2 (528 tiles x 12 splits) 1.0 * 10^8
// atomics = 0.64 ms of the chip's atomic throughput inside a 2.0 ms launch; config 3 (1653 x 10) 2.7 * 10^8 = 1.7 ms inside
// 4.3 ms. Here a workgroup's sums leave as plain, fully coalesced 32-bit stores into a staging block of its own
// (k_dense_tile_small / k_dense_tile_small_compact: the body of the headline kernel, fsk_tile_kernel_dma.inc, with
// FSK_DMA_STAGE32), and k_dense_widen adds the blocks of a tile into the 64-bit triangle once: 4 bytes written and read a cell
// and split at streaming rate instead of an atomic each.
//
// A translation unit of its own: k_dense_tile_dma (126 VGPRs, no scratch, 96 % of the v_dot8 issue peak on the headline) is
// compiled where it always was, from the same body, and nothing here can reach its register allocation
// (tests/test_abi.py::test_headline_tile_kernel_resources asserts both from the compiler's resource report).
#include "fsk_engine_internal.h"
#include "fsk_common.h"

using namespace fsk_detail;

namespace fsk {

#define FSK_DMA_STAGE32 1
#define FSK_DMA_KERNEL k_dense_tile_small
#define FSK_DMA_COMPACT 0
#include "fsk_tile_kernel_dma.inc"
#undef FSK_DMA_KERNEL
#undef FSK_DMA_COMPACT
#define FSK_DMA_KERNEL k_dense_tile_small_compact
#define FSK_DMA_COMPACT 1
#include "fsk_tile_kernel_dma.inc"
#undef FSK_DMA_KERNEL
#undef FSK_DMA_COMPACT
#undef FSK_DMA_STAGE32

// K += the n_splits staging blocks of every tile. One workgroup of 256 threads a tile, the thread that computed a cell in
// the tile kernel adds it (the blocks are [register][thread]: a wave reads 256 contiguous bytes a register and split; its
// cells of K are the 64-byte row segments of the tile kernels' own flush). Plain read-modify-write: a cell has one writer,
// launches are ordered on the stream.
__global__ __launch_bounds__(256) void k_dense_widen(const uint32_t* stage32, const uint32_t* tile_tab, uint32_t n_tiles, int n_splits, uint32_t N,
                                                     u64* K) {
    const uint32_t tid = threadIdx.x, ty = tid >> 4, tx = tid & 15u;
    const uint32_t tile = tile_tab[blockIdx.x];
    const uint32_t ti = tile >> 16, tj = tile & 0xffffu;
    const size_t block = (size_t)TILE * TILE, split_stride = (size_t)n_tiles * block;
    const uint32_t* base = stage32 + (size_t)blockIdx.x * block + tid;
#pragma unroll 2
    for (int a = 0; a < 8; ++a) {
        const u64 i = (u64)ti * TILE + tile_index(ty, (uint32_t)a);
        u64 sum[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) sum[b] = 0;
        for (int s = 0; s < n_splits; ++s) {
            uint32_t v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = base[(size_t)s * split_stride + (size_t)((a * 8 + b) * 256)];
#pragma unroll
            for (int b = 0; b < 8; ++b) sum[b] += v[b];
        }
        if (i >= N) continue;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const u64 j = (u64)tj * TILE + tile_index(tx, (uint32_t)b);
            if (j <= i && sum[b] != 0) K[tri_index(i, j)] += sum[b];
        }
    }
}

}  // namespace fsk

namespace fsk_detail {

// bytes of staging a launch of n_tiles x n_splits workgroups needs
size_t dense_small_stage_bytes(u64 n_tiles, int n_splits) { return (size_t)n_tiles * (size_t)n_splits * fsk::TILE * fsk::TILE * sizeof(uint32_t); }

// The split tile launch of accumulate_dense through the staging blocks (the caller has zeroed what of K is lazily zero).
int dense_tile_small(fsk_engine* e, bool compact, u64 n_tiles, int n_splits, int nb, uint32_t Vq8, uint32_t nst, u64* K, int slots_per_split) {
    FSK_HIP(e->d_stage32.reserve(dense_small_stage_bytes(n_tiles, n_splits) / sizeof(uint32_t)));
    if (compact)
        FSK_LAUNCH(fsk::k_dense_tile_small_compact, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, (const uint32_t*)e->d_C4.p,
                   (const uint32_t*)e->d_C4H.p, (const uint32_t*)e->d_rowmask.p, (const uint32_t*)e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N,
                   e->d_stage32.p, slots_per_split, 0, (const uint16_t*)e->d_vc.p);
    else
        FSK_LAUNCH(fsk::k_dense_tile_small, dim3((uint32_t)n_tiles, n_splits), dim3(256), 0, e->stream, (const uint32_t*)e->d_C4.p,
                   (const uint32_t*)e->d_C4H.p, (const uint32_t*)e->d_rowmask.p, (const uint32_t*)e->d_tiletab.p, nb, Vq8, nst, (uint32_t)e->N,
                   e->d_stage32.p, slots_per_split, 0);
    FSK_LAUNCH(fsk::k_dense_widen, dim3((uint32_t)n_tiles), dim3(256), 0, e->stream, (const uint32_t*)e->d_stage32.p, (const uint32_t*)e->d_tiletab.p,
               (uint32_t)n_tiles, n_splits, (uint32_t)e->N, K);
    e->st.launches += 1;
    return FSK_OK;
}

}  // namespace fsk_detail
