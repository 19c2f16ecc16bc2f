// fsk_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the gapped-k-mer engine.
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
//     k_dense_tile   output-stationary 128x128 tile of K per workgroup: panels staged through
//                    LDS, 8x8 register block per lane, v_dot8_u32_u4 multiply-adds summed in
//                    registers over ALL combos of the launch, then ONE 64-bit atomicAdd per cell.
//
//   SPARSE (large key space, e.g. protein: 24^4 keys, runs of 3-5) — the reference's dataflow
//     k_sparse_extract  packed (combo,k-mer) key + sequence id per g-mer, straight from packed HBM
//     k_rs_*            LDS-staged 8-bit LSD radix sort (wave64 ballot match ranking, stable)
//     k_seg_*           run heads by neighbour compare, wave prefix sums -> distinct (k-mer,seq)
//                       entries with multiplicities and run starts
//     k_bucket_* + k_slice_pairs  (run, pair) updates summed in LDS by the workgroup that owns the
//                       rows, non-zero cells flushed with 64-bit atomicAdd (k_sparse_pairs: direct
//                       per-pair atomics, used when a row band of K does not fit in LDS)
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

// =============================================================================================
// DENSE PATH
// =============================================================================================
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
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
#ifdef FSK_EMU
    return (a & 0xffffffu) * (b & 0xffffffu) + c;
#else
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
#endif
}

// One wave's share of the windows [j0, hi) of a staging chunk: key from K kept positions held
// in registers (K is a template parameter so the K LDS byte reads of a window are independent
// and issue back to back; K = 0 is the generic loop for k > 8). MARK: only record which keys
// occur (bitmap pre-pass of the key compaction); else one LDS atomic per window, the key first
// mapped through the combo's compaction table when LUT.
// kcache (several key sweeps over one staging pass): the first sweep stores every window's
// (compacted) key in LDS, kmode 1; the later sweeps read it back instead of recomputing, kmode 2.
template <int K, bool MARK, bool LUT>
__device__ __forceinline__ void count_windows(const uint8_t* symT, uint32_t* hist, const uint16_t* lut, const uint32_t (&pr)[16],
                                              int k, uint32_t sigma, uint32_t j0, uint32_t hi, uint32_t cb, uint32_t nwin,
                                              uint32_t r, uint32_t half, uint32_t key_lo, uint32_t key_n, uint16_t* kcache,
                                              int kmode) {
    if (K > 0 && !MARK && !LUT && kmode == 0) {
        // The common case (every BASELINE config): WPT windows per trip. Window j+4 of a lane lies
        // 4 rows = 256 bytes further in every symbol column, an immediate offset of the same
        // address registers, so the loop bookkeeping is paid once per WPT windows. Rows past a
        // sequence's end are zero padding inside symT (the trip condition keeps them in range);
        // their updates are predicated off.
        constexpr int WPT = 4;
        const uint8_t* p[K > 0 ? K : 1];
#pragma unroll
        for (int c = 0; c < K; ++c) p[c] = symT + (j0 - cb + pr[c]) * PANEL + r;
        uint32_t j = j0;
        for (; j + 4u * (WPT - 1) < hi; j += 4u * WPT) {
            uint32_t kk[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) kk[u] = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
#pragma unroll
                for (int u = 0; u < WPT; ++u) kk[u] = mad24(kk[u], sigma, p[c][u * 4 * PANEL]);
                p[c] += 4 * WPT * PANEL;
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const uint32_t key = kk[u] - key_lo;  // wraps for keys below the sweep: rejected by the compare
                if (j + 4u * u < nwin && key < key_n) atomicAdd(&hist[key * 32u + (r >> 1)], 1u << half);
            }
        }
        for (; j < hi; j += 4u) {  // the last few windows of the chunk
            uint32_t k0 = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                k0 = mad24(k0, sigma, p[c][0]);
                p[c] += 4 * PANEL;
            }
            k0 -= key_lo;
            if (j < nwin && k0 < key_n) atomicAdd(&hist[k0 * 32u + (r >> 1)], 1u << half);
        }
        return;
    }
    for (uint32_t j = j0; j < hi; j += 4) {
        if (j < nwin) {
            uint32_t key = 0;  // keys stay below 2^24 on this path (V <= 16384): 24-bit multiplies issue at full rate
            if (!MARK && kmode == 2) {  // workgroup-uniform
                key = kcache[j * PANEL + r];
                key -= key_lo;
                if (key < key_n) atomicAdd(&hist[key * 32u + (r >> 1)], 1u << half);
                continue;
            }
            if (K > 0) {
#pragma unroll
                for (int c = 0; c < K; ++c) key = mad24(key, sigma, symT[(j - cb + pr[c]) * PANEL + r]);
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < k) key = mad24(key, sigma, symT[(j - cb + pr[c]) * PANEL + r]);
            }
            if (MARK) {
                atomicOr(&hist[key >> 5], 1u << (key & 31u));  // hist doubles as the key bitmap
            } else {
                if (LUT) key = lut[key];
                if (kmode == 1) kcache[j * PANEL + r] = (uint16_t)key;
                key -= key_lo;  // wraps for keys below the sweep: rejected by the compare
                if (key < key_n) atomicAdd(&hist[key * 32u + (r >> 1)], 1u << half);
            }
        }
    }
}

template <bool MARK, bool LUT>
__device__ __forceinline__ void count_windows_k(const uint8_t* symT, uint32_t* hist, const uint16_t* lut, const uint32_t (&pr)[16],
                                                int k, uint32_t sigma, uint32_t j0, uint32_t hi, uint32_t cb, uint32_t nwin,
                                                uint32_t r, uint32_t half, uint32_t key_lo, uint32_t key_n, uint16_t* kcache,
                                                int kmode) {
    switch (k) {  // workgroup-uniform
        case 1: count_windows<1, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 2: count_windows<2, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 3: count_windows<3, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 4: count_windows<4, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 5: count_windows<5, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 6: count_windows<6, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 7: count_windows<7, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        case 8: count_windows<8, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
        default: count_windows<0, MARK, LUT>(symT, hist, lut, pr, k, sigma, j0, hi, cb, nwin, r, half, key_lo, key_n, kcache, kmode); break;
    }
}

// Eight consecutive keys' counts of lane r's sequence -> one dword of lo nibbles and one of hi
// nibbles; `seen` collects every count (a bit above bit 7 = some count exceeded 255).
template <bool TAIL>
__device__ __forceinline__ void pack_count_row(const uint16_t* hist16, uint32_t key0, uint32_t key_n, uint32_t r, uint32_t& plo,
                                               uint32_t& phi, uint32_t& seen) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t key = key0 + q;
        const uint32_t c = (!TAIL || key < key_n) ? (uint32_t)hist16[key * 64u + r] : 0u;
        seen |= c;
        plo |= (c & 15u) << (4 * q);
        phi |= ((c >> 4) & 15u) << (4 * q);
    }
}

// Per-sequence counting sort of the k-mers selected by each combo ("segment counts").
// grid = (n_panels, n_chunks), block = 256 (wave w takes windows j = w mod 4; lane = sequence).
// dynamic LDS: symT[CH+g-1][64] u8 | hist[4*Vcq][32] u32 (two u16 counters per dword) | lut[V] u16.
// LDS banking: lane r touches dword (key*32 + r/2): bank depends on r only -> conflict-free for
// any key mix; the two lanes sharing a dword add to different halves (same-address atomics).
//
// Key compaction (alphabets with rare symbols, e.g. DNA with a few 'n'): a first launch with
// MARK = true only records, per combo, which of the sigma^k keys occur anywhere (keybits);
// k_dense_keylut turns that into a rank table, and the counting launch (LUT = true) maps every
// key through it, so panels and the tile kernel only carry the keys that exist (vc[slot] of them).
template <bool MARK, bool LUT>
__global__ __launch_bounds__(256) void k_dense_count(SeqView S, int g, int k, uint32_t sigma, uint32_t Vq,
                                                     uint32_t Vcq, uint32_t max_win, uint32_t CH, const uint8_t* combo_pos,
                                                     int n_slots, int slots_per_chunk, uint32_t* C4, uint32_t* C4H,
                                                     uint32_t* rowmask, uint32_t nst, uint32_t* overflow_flag, uint32_t V,
                                                     const uint16_t* lut_g, const uint16_t* vc, uint32_t* keybits,
                                                     uint32_t kc_rows) {
    // Counts leave as two 4-bit planes, count = lo + 16 * hi (8 keys per dword): C4 holds lo and
    // is all the tile kernel multiplies for almost every key; C4H holds hi, zero unless a k-mer
    // occurs more than 15 times in one sequence (poly-A, runs of 'n'); rowmask[panel][slot][..]
    // has one bit per dword row saying whether any of the panel's 64 sequences has hi != 0 there.
    // Counts above 255 raise overflow_flag (the host then takes the sparse dataflow).
    __shared__ uint32_t srowmask[64];
    // Vcq = key quads per histogram chunk (the LDS histogram covers 4*Vcq keys at a time; key
    // spaces beyond that are counted in several sweeps over the same staged symbols).
    // CH = windows per staging chunk: symT holds CH + g - 1 symbols per sequence. CH >= max_win
    // (every BASELINE config) means the sequences are unpacked once and reused by all the combos
    // of this workgroup; longer sequences are re-staged chunk by chunk inside the combo loop.
    FSK_DYN_SHARED(unsigned char, smem);
    uint8_t* symT = smem;
    const uint32_t sym_rows = CH + (uint32_t)g - 1u;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)sym_rows * PANEL);
    uint16_t* lut = reinterpret_cast<uint16_t*>(smem + (size_t)sym_rows * PANEL + (size_t)Vcq * 512);
    // window-key cache [kc_rows][64] u16 behind the table (kc_rows = max_win when the host enabled it)
    uint16_t* kcache = lut + (LUT ? V : 0u);
    const int tid = threadIdx.x, r = tid & 63, w = tid >> 6;
    const uint32_t panel = blockIdx.x;
    const uint32_t seq = panel * PANEL + r;
    const uint32_t len = seq < S.n_seq ? S.len[seq] : 0u;
    const uint32_t wbase = seq < S.n_seq ? S.wstart[seq] : 0u;
    const uint32_t nwin = len >= (uint32_t)g ? len - g + 1 : 0u;
    const bool single = CH >= max_win;
    const int slot0 = blockIdx.y * slots_per_chunk;
    const int slot1 = slot0 + slots_per_chunk < n_slots ? slot0 + slots_per_chunk : n_slots;
    const uint32_t half = (uint32_t)(r & 1) * 16u;
    const uint32_t Vq8 = (Vq + 1u) >> 1;
    const uint32_t Vw = (V + 31u) >> 5;  // words of the key bitmap
    uint32_t seen = 0;  // OR of every count read out: a bit above bit 7 means some count exceeded 255
    for (int slot = slot0; slot < slot1; ++slot) {
        if (!MARK) {
            __syncthreads();  // previous combo's mask written out
            if (tid < 64) srowmask[tid] = 0u;
        }
        // this combo's kept positions, once per combo, into registers (k <= 16 on this path)
        uint32_t pr[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) pr[c] = c < k ? (uint32_t)combo_pos[(size_t)slot * k + c] : 0u;
        // rows (key quads) this combo really has: all of them, or the compacted count
        const uint32_t Vq_s = LUT ? ((uint32_t)vc[slot] + 3u) >> 2 : Vq;
        if (LUT) {
            __syncthreads();  // previous combo's table no longer read
            for (uint32_t i = tid; i < V; i += 256) lut[i] = lut_g[(size_t)slot * V + i];
        }
        uint32_t* out4 = C4 + ((size_t)panel * n_slots + slot) * ((size_t)Vq8 * PANEL);
        uint32_t* out4h = C4H + ((size_t)panel * n_slots + slot) * ((size_t)Vq8 * PANEL);
        const uint32_t sweep_end = MARK ? 1u : Vq_s;
        for (uint32_t kc0 = 0; kc0 < sweep_end; kc0 += Vcq) {  // key-space sweep
            const uint32_t key_lo = 4u * kc0, key_n = 4u * (kc0 + Vcq < Vq_s ? Vcq : Vq_s - kc0);
            const uint32_t hist_dwords = MARK ? Vw : 4u * Vcq * 32u;
            __syncthreads();  // previous read-out finished
            for (uint32_t i = tid; i < hist_dwords; i += 256) hist[i] = 0u;
            for (uint32_t cb = 0; cb < max_win; cb += CH) {
                if (!single || (slot == slot0 && kc0 == 0)) {
                    __syncthreads();  // everyone is done with the previous chunk's symbols
                    for (uint32_t p = w; p < sym_rows; p += 4)
                        symT[p * PANEL + r] = cb + p < len ? (uint8_t)fetch_sym(S.words, wbase, cb + p, S.bits) : (uint8_t)0;
                }
                __syncthreads();  // symbols staged, histogram zeroed, table loaded
                const uint32_t hi = cb + CH < max_win ? cb + CH : max_win;
                const int kmode = (MARK || kc_rows == 0u) ? 0 : (kc0 == 0u ? 1 : 2);
                count_windows_k<MARK, LUT>(symT, hist, lut, pr, k, sigma, cb + (uint32_t)w, hi, cb, nwin, (uint32_t)r, half, key_lo, key_n,
                                           kcache, kmode);
            }
            __syncthreads();
            if (MARK) {  // merge this panel's key bitmap into the combo's
                for (uint32_t i = tid; i < Vw; i += 256)
                    if (hist[i]) atomicOr(&keybits[(size_t)slot * Vw + i], hist[i]);
                continue;
            }
            // read-out: 8 keys per dword row, lo and hi nibbles (Vcq is even, so a sweep starts on
            // an 8-key boundary); one 256-B row per wave and plane
            // (lane r's 16-bit counter of `key` is halfword key*64 + r of the histogram; only the
            // last row of an odd number of key quads has keys to mask out)
            const uint32_t n8 = ((key_n >> 2) + 1u) >> 1;
            const uint16_t* hist16 = reinterpret_cast<const uint16_t*>(hist);
            for (uint32_t k8 = w; k8 < n8; k8 += 4) {
                uint32_t plo = 0, phi = 0;
                if (8u * k8 + 8u <= key_n) pack_count_row<false>(hist16, 8u * k8, key_n, (uint32_t)r, plo, phi, seen);
                else pack_count_row<true>(hist16, 8u * k8, key_n, (uint32_t)r, plo, phi, seen);
                const uint32_t row = (kc0 >> 1) + k8;
                out4[(size_t)row * PANEL + panel_slot(r)] = plo;
                out4h[(size_t)row * PANEL + panel_slot(r)] = phi;
                if (__ballot(phi != 0u) != 0ull && r == 0) atomicOr(&srowmask[row >> 5], 1u << (row & 31u));
            }
        }
        if (!MARK) {
            __syncthreads();
            if ((uint32_t)tid < nst) rowmask[((size_t)panel * n_slots + slot) * nst + tid] = srowmask[tid];
        }
    }
    if (!MARK && seen > 255u) atomicOr(overflow_flag, 1u);
}

// Key compaction table of one combo: rank of every key that occurs, 0xFFFF otherwise; vc = how
// many occur. grid = n_slots, block = 256. V <= 8192 (256 bitmap words).
__global__ __launch_bounds__(256) void k_dense_keylut(const uint32_t* keybits, uint32_t V, uint16_t* lut_g, uint16_t* vc) {
    __shared__ uint32_t tmp[4];
    const uint32_t slot = blockIdx.x, tid = threadIdx.x;
    const uint32_t Vw = (V + 31u) >> 5;
    const uint32_t word = tid < Vw ? keybits[(size_t)slot * Vw + tid] : 0u;
    uint32_t tot;
    uint32_t rank = block_excl_scan_256<uint32_t>((uint32_t)__popc(word), tmp, &tot);
    if (tid < Vw) {
        for (uint32_t b = 0; b < 32u; ++b) {
            const uint32_t key = tid * 32u + b;
            if (key < V) lut_g[(size_t)slot * V + key] = (word >> b) & 1u ? (uint16_t)rank++ : (uint16_t)0xffff;
        }
    }
    if (tid == 0) vc[slot] = (uint16_t)tot;
}

// U = sum over (combo, key) of d(d+1)/2, d = number of sequences in which the key occurs: the
// number of `+=` the reference's countAndUpdateTri issues (shared.cpp:316-327), i.e. the
// algorithmic update count the roofline is priced on. Read straight off the count
// panels (a key occurs in a sequence iff its lo or hi nibble is non-zero); profiling aid only.
// grid = (Vq8, n_slots), block = 64 (lane = sequence within panel).
__global__ __launch_bounds__(64) void k_dense_distinct(const uint32_t* C4, const uint32_t* C4H, uint32_t n_panels, int n_slots,
                                                       uint32_t Vq8, u64* U, const uint16_t* vc) {
    const uint32_t k8 = blockIdx.x, slot = blockIdx.y, r = threadIdx.x;
    if (vc && k8 >= ((uint32_t)vc[slot] + 7u) >> 3) return;  // rows beyond the compacted keys are not written
    uint32_t d[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (uint32_t p = 0; p < n_panels; ++p) {
        const size_t o = ((size_t)p * n_slots + slot) * ((size_t)Vq8 * PANEL) + (size_t)k8 * PANEL + r;
        const uint32_t v = C4[o] | C4H[o];
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] += ((v >> (4 * q)) & 15u) ? 1u : 0u;
    }
    u64 u = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t x = d[q];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s);
        u += (u64)x * (x + 1) / 2;
    }
    if (r == 0 && u) atomicAdd(U, u);
}

#define FSK_TILE_KERNEL k_dense_tile
#define FSK_TILE_COMPACT 0
#include "fsk_tile_kernel.inc"
#undef FSK_TILE_KERNEL
#undef FSK_TILE_COMPACT
#define FSK_TILE_KERNEL k_dense_tile_compact
#define FSK_TILE_COMPACT 1
#include "fsk_tile_kernel.inc"
#undef FSK_TILE_KERNEL
#undef FSK_TILE_COMPACT
#include "fsk_tile_kernel_dma.inc"

// =============================================================================================
// SPARSE PATH
// =============================================================================================
// key = slot * V + sum_c sym[j+pos_c] * sigma^(k-1-c): one record per (slot, g-mer).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_sparse_extract(SeqView S, const uint32_t* feat_seq, const uint32_t* fstart,
                                                        uint32_t nfeat, int k, uint32_t sigma, u64 V,
                                                        const uint8_t* combo_pos, KeyT* keys, uint32_t* vals) {
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;
    const uint32_t slot = blockIdx.y;
    if (f >= nfeat) return;
    const uint32_t seq = feat_seq[f];
    const uint32_t j = f - fstart[seq];
    const uint32_t wbase = S.wstart[seq];
    const uint8_t* pos = combo_pos + (size_t)slot * k;
    u64 key = 0;
    for (int c = 0; c < k; ++c) key = key * sigma + fetch_sym(S.words, wbase, j + pos[c], S.bits);
    const size_t o = (size_t)slot * nfeat + f;
    keys[o] = (KeyT)((u64)slot * V + key);
    vals[o] = seq;
}

constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = 256 * RS_ITEMS;

// pass 1: per-workgroup digit histogram, stored digit-major so each digit row scans linearly
template <typename KeyT>
__global__ __launch_bounds__(256) void k_rs_hist(const KeyT* keys, u64 n, int shift, uint32_t* blockhist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0u;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RS_TILE;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const u64 i = base + (u64)it * 256 + tid;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockhist[(size_t)tid * nblocks + blockIdx.x] = h[tid];
}

// pass 2: one workgroup per digit: exclusive scan of its row in place, row total to totals[digit]
__global__ __launch_bounds__(256) void k_rs_scan_rows(uint32_t* blockhist, uint32_t nblocks, uint32_t* totals) {
    __shared__ uint32_t tmp[4];
    __shared__ uint32_t carry_s;
    uint32_t* row = blockhist + (size_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        uint32_t v[4], s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = base + (uint32_t)tid * 4u + q;
            v[q] = i < nblocks ? row[i] : 0u;
            s += v[q];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan_256<uint32_t>(s, tmp, &tot) + carry;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = base + (uint32_t)tid * 4u + q;
            if (i < nblocks) row[i] = ex;
            ex += v[q];
        }
        carry += tot;
    }
    if (tid == 0) { totals[blockIdx.x] = carry; carry_s = carry; }
    (void)carry_s;
}

// pass 3: stable scatter. Ranking inside the workgroup uses wave64 ballot matching (8 ballots
// give the set of lanes holding the same digit); elements are first permuted into digit order
// in LDS, then written out so that equal digits go to consecutive addresses.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_rs_scatter(const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out,
                                                    uint32_t* vals_out, u64 n, int shift, const uint32_t* blockhist,
                                                    const uint32_t* totals, uint32_t nblocks) {
    __shared__ uint32_t goff[256];         // global destination of this workgroup's first `digit`
    __shared__ uint32_t wave_cnt[4][256];  // per round: count, then offset, of digit in wave
    __shared__ uint32_t running[256];      // digit count in earlier rounds
    __shared__ uint32_t blk_start[256];    // exclusive scan of this workgroup's digit totals
    __shared__ uint32_t tmp[4];
    __shared__ KeyT s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = (u64)blockIdx.x * RS_TILE;
    {
        const uint32_t tot = totals[tid];
        const uint32_t dbase = block_excl_scan_256<uint32_t>(tot, tmp, nullptr);
        goff[tid] = dbase + blockhist[(size_t)tid * nblocks + blockIdx.x];
        running[tid] = 0u;
    }
    KeyT key[RS_ITEMS];
    uint32_t val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const u64 i = base + (u64)it * 256 + tid;
        const bool valid = i < n;
        key[it] = valid ? keys_in[i] : (KeyT)0;
        val[it] = valid ? vals_in[i] : 0u;
        const uint32_t digit = (uint32_t)(key[it] >> shift) & 255u;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t cnt = (uint32_t)__popcll(peers);
#pragma unroll
        for (int q = 0; q < 4; ++q) wave_cnt[q][tid] = 0u;
        __syncthreads();
        if (valid && rank_in_wave == 0) wave_cnt[wave][digit] = cnt;
        __syncthreads();
        {
            uint32_t off = running[tid];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c = wave_cnt[q][tid];
                wave_cnt[q][tid] = off;
                off += c;
            }
            running[tid] = off;
        }
        __syncthreads();
        rank[it] = valid ? wave_cnt[wave][digit] + rank_in_wave : 0xffffffffu;
        __syncthreads();  // wave_cnt is cleared at the top of the next round
    }
    {
        const uint32_t mine = running[tid];
        const uint32_t ex = block_excl_scan_256<uint32_t>(mine, tmp, nullptr);
        blk_start[tid] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        if (rank[it] != 0xffffffffu) {
            const uint32_t digit = (uint32_t)(key[it] >> shift) & 255u;
            const uint32_t p = blk_start[digit] + rank[it];
            s_keys[p] = key[it];
            s_vals[p] = val[it];
        }
    }
    __syncthreads();
    const u64 remain = n - base;
    const uint32_t nvalid = remain < (u64)RS_TILE ? (uint32_t)remain : (uint32_t)RS_TILE;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const uint32_t p = (uint32_t)it * 256u + (uint32_t)tid;
        if (p < nvalid) {
            const KeyT kx = s_keys[p];
            const uint32_t digit = (uint32_t)(kx >> shift) & 255u;
            const size_t dst = (size_t)goff[digit] + (p - blk_start[digit]);
            keys_out[dst] = kx;
            vals_out[dst] = s_vals[p];
        }
    }
}

// ---- segments: distinct (key, seq) entries and runs of equal keys ---------------------------
// flag word: low 32 bits count entry heads, high 32 bits count run heads.
constexpr int SEG_ITEMS = 8;
constexpr int SEG_TILE = 256 * SEG_ITEMS;

template <typename KeyT>
__device__ __forceinline__ u64 seg_flag(const KeyT* keys, const uint32_t* vals, u64 i) {
    if (i == 0) return (1ull << 32) | 1ull;
    const bool hk = keys[i] != keys[i - 1];
    const bool hkv = hk || vals[i] != vals[i - 1];
    return ((u64)hk << 32) | (u64)hkv;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_seg_reduce(const KeyT* keys, const uint32_t* vals, u64 n, u64* blocksum) {
    __shared__ u64 tmp[4];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * SEG_TILE + (u64)tid * SEG_ITEMS;
    u64 s = 0;
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; ++q)
        if (base + q < n) s += seg_flag(keys, vals, base + q);
    u64 tot;
    block_excl_scan_256<u64>(s, tmp, &tot);
    if (tid == 0) blocksum[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of blocksum in place; totals[0]=entries D, totals[1]=runs R;
// plants the sentinel estart[D] = n.
__global__ __launch_bounds__(256) void k_seg_scan_blocks(u64* blocksum, uint32_t nblocks, u64 n, uint32_t* totals,
                                                         uint32_t* estart) {
    __shared__ u64 tmp[4];
    const int tid = threadIdx.x;
    u64 carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const u64 v = i < nblocks ? blocksum[i] : 0ull;
        u64 tot;
        const u64 ex = block_excl_scan_256<u64>(v, tmp, &tot) + carry;
        if (i < nblocks) blocksum[i] = ex;
        carry += tot;
    }
    if (tid == 0) {
        const uint32_t D = (uint32_t)(carry & 0xffffffffull);
        totals[0] = D;
        totals[1] = (uint32_t)(carry >> 32);
        estart[D] = (uint32_t)n;
    }
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_seg_write(const KeyT* keys, const uint32_t* vals, u64 n, const u64* blocksum,
                                                   uint32_t* estart, uint32_t* eseq, uint32_t* erun, uint32_t* rstart) {
    __shared__ u64 tmp[4];
    const int tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * SEG_TILE + (u64)tid * SEG_ITEMS;
    u64 fl[SEG_ITEMS], s = 0;
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; ++q) {
        fl[q] = base + q < n ? seg_flag(keys, vals, base + q) : 0ull;
        s += fl[q];
    }
    u64 ex = block_excl_scan_256<u64>(s, tmp, nullptr) + blocksum[blockIdx.x];
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; ++q) {
        if (fl[q] & 1ull) {  // entry head
            const uint32_t e = (uint32_t)(ex & 0xffffffffull);
            const uint32_t hk = (uint32_t)(fl[q] >> 32);
            const uint32_t run = (uint32_t)(ex >> 32) + hk - 1u;  // runs started up to here, minus 1
            estart[e] = (uint32_t)(base + q);
            eseq[e] = vals[base + q];
            erun[e] = run;
            if (hk) rstart[run] = e;
        }
        ex += fl[q];
    }
}

// per (run, pair) atomics: entry e pairs with every earlier entry of its run and itself —
// exactly the += the reference issues (shared.cpp:316-327). U counts them.
__global__ __launch_bounds__(256) void k_sparse_pairs(const uint32_t* totals, const uint32_t* estart, const uint32_t* eseq,
                                                      const uint32_t* erun, const uint32_t* rstart, u64 row0, u64 row1,
                                                      u64* K, u64* U) {
    const uint32_t D = totals[0];
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    const bool active = e < D;
    u64 work = 0;
    if (active) {
        const u64 sa = eseq[e];
        const u64 ca = estart[e + 1] - estart[e];
        const uint32_t rs = rstart[erun[e]];
        for (uint32_t b = rs; b <= e; ++b) {
            const u64 sb = eseq[b];
            const u64 cb = estart[b + 1] - estart[b];
            const u64 i = sa > sb ? sa : sb, j = sa > sb ? sb : sa;
            if (i >= row0 && i < row1) {  // only the requested band of rows
                atomicAdd(&K[tri_index(i, j)], ca * cb);
                ++work;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) work += __shfl_xor(work, d);
    if ((threadIdx.x & 63) == 0 && work) atomicAdd(U, work);
}

// ---- owner-slice pair accumulation -----------------------------------------------------------
// Scattered 64-bit atomics run at the chip's per-request rate (~16 G/s) and the protein configs
// issue ~450 of them per cell of K over a full run. When a band of `rps` rows of K fits in LDS
// (rps * N u32 cells), the entries are bucketed by the slice that owns their row (an LDS-
// privatised counting sort, any order inside a bucket), every slice is owned by ONE workgroup,
// the (run, pair) updates of a whole batch of combos are summed with LDS atomics, and only the
// non-zero cells are flushed, row-contiguous, with one 64-bit atomicAdd each.
constexpr int BK_TILE = 2048;       // entries per bucketing workgroup
constexpr uint32_t BK_SPLIT = 4;    // an entry with many partners becomes up to this many work records
constexpr uint32_t BK_CHUNK = 8;    // partners per record (at least)
// partners per record for an entry with P partners, and the number of records that makes
__device__ __forceinline__ uint32_t bk_chunk(uint32_t P) {
    const uint32_t c = (P + BK_SPLIT - 1u) / BK_SPLIT;
    return c > BK_CHUNK ? c : BK_CHUNK;
}
__device__ __forceinline__ uint32_t bk_records(uint32_t P) { return (P + bk_chunk(P) - 1u) / bk_chunk(P); }
constexpr int BK_MAX_SLICES = 8192; // LDS histogram / cursor size

// pass A: per-workgroup slice histogram -> blockhist[slice][block]; also packs (seq, multiplicity)
// per entry so that the pair loop fetches a partner with one 8-byte load
__global__ __launch_bounds__(256) void k_bucket_hist(const uint32_t* totals, const uint32_t* estart, const uint32_t* eseq,
                                                     const uint32_t* erun, const uint32_t* rstart, uint32_t rps,
                                                     uint32_t n_slices, uint32_t nblocks, uint32_t* blockhist, uint2* epair) {
    __shared__ uint32_t h[BK_MAX_SLICES];
    const uint32_t D = totals[0];
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n_slices; i += 256) h[i] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * BK_TILE;
    for (int it = 0; it < BK_TILE / 256; ++it) {
        const uint32_t e = base + (uint32_t)it * 256u + (uint32_t)tid;
        if (e < D) {
            const uint32_t sq = eseq[e];
            const uint32_t P = e - rstart[erun[e]] + 1u;  // partners: earlier entries of the run + itself
            atomicAdd(&h[sq / rps], bk_records(P));
            epair[e] = make_uint2(sq, estart[e + 1] - estart[e]);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_slices; i += 256) blockhist[(size_t)i * nblocks + blockIdx.x] = h[i];
}

// pass C: exclusive scan of the slice totals (single workgroup), slice_off[n_slices] = D
__global__ __launch_bounds__(256) void k_bucket_scan_totals(const uint32_t* totals_in, uint32_t n_slices, uint32_t* slice_off) {
    __shared__ uint32_t tmp[4];
    const int tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_slices; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t v = i < n_slices ? totals_in[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256<uint32_t>(v, tmp, &tot) + carry;
        if (i < n_slices) slice_off[i] = ex;
        carry += tot;
    }
    if (tid == 0) slice_off[n_slices] = carry;
}

// pass D: scatter self-contained work records {seq, multiplicity, first partner, last partner}
// into the slice's bucket (LDS cursors; the order inside a bucket is free). An entry with many
// partners is split into up to BK_SPLIT records so that the lanes of the pair kernel carry
// similar loads. The dependent lookups (run -> run start) are paid here, where thousands of
// workgroups hide them.
__global__ __launch_bounds__(256) void k_bucket_scatter(const uint32_t* totals, const uint2* epair, const uint32_t* erun,
                                                        const uint32_t* rstart, uint32_t rps, uint32_t n_slices,
                                                        uint32_t nblocks, const uint32_t* blockhist, const uint32_t* slice_off,
                                                        uint4* list) {
    __shared__ uint32_t cur[BK_MAX_SLICES];
    const uint32_t D = totals[0];
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n_slices; i += 256) cur[i] = slice_off[i] + blockhist[(size_t)i * nblocks + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * BK_TILE;
    for (int it = 0; it < BK_TILE / 256; ++it) {
        const uint32_t e = base + (uint32_t)it * 256u + (uint32_t)tid;
        if (e < D) {
            const uint2 sc = epair[e];
            const uint32_t rs = rstart[erun[e]];
            const uint32_t P = e - rs + 1u, ch = bk_chunk(P), nr = bk_records(P);
            const uint32_t dst = atomicAdd(&cur[sc.x / rps], nr);
            for (uint32_t q = 0; q < nr; ++q) {  // {seq, multiplicity, first partner, last partner}
                const uint32_t lo = rs + q * ch, hi = lo + ch - 1u < e ? lo + ch - 1u : e;
                list[dst + q] = make_uint4(sc.x, sc.y, lo, hi);
            }
        }
    }
}

// One workgroup per slice of `rps` rows: sums every (run, pair) update whose row it owns in LDS,
// then flushes the non-zero cells. dynamic LDS: rps * N u32. Partners are fetched four at a time
// so that their loads overlap.
__global__ __launch_bounds__(256) void k_slice_pairs(const uint32_t* slice_off, const uint4* list, const uint2* epair,
                                                     uint32_t rps, uint32_t N, uint32_t slice0, u64 row0, u64 row1, u64* K,
                                                     u64* U) {
    FSK_DYN_SHARED(uint32_t, sk);
    const int tid = threadIdx.x;
    const uint32_t slice = slice0 + blockIdx.x;
    const uint32_t r_lo = slice * rps;
    const uint32_t cells = rps * N;
    for (uint32_t i = tid; i < cells; i += 256) sk[i] = 0u;
    __syncthreads();
    const uint32_t lo = slice_off[slice], hi = slice_off[slice + 1];
    u64 work = 0;
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t t = base + (uint32_t)tid;
        if (t < hi) {
            const uint4 rec = list[t];  // {seq_a, cnt_a, first partner, last partner}
            if (rec.x >= row0 && rec.x < row1) {
                uint32_t* row = sk + (size_t)(rec.x - r_lo) * N;
                uint32_t b = rec.z;
                for (; b + 3 <= rec.w; b += 4) {
                    const uint2 p0 = epair[b], p1 = epair[b + 1], p2 = epair[b + 2], p3 = epair[b + 3];
                    atomicAdd(&row[p0.x], rec.y * p0.y);  // seq_b <= seq_a
                    atomicAdd(&row[p1.x], rec.y * p1.y);
                    atomicAdd(&row[p2.x], rec.y * p2.y);
                    atomicAdd(&row[p3.x], rec.y * p3.y);
                }
                for (; b <= rec.w; ++b) {
                    const uint2 p = epair[b];
                    atomicAdd(&row[p.x], rec.y * p.y);
                }
                work += (u64)(rec.w - rec.z + 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < cells; i += 256) {
        const uint32_t v = sk[i];
        if (v) {
            const u64 r = (u64)r_lo + i / N, c = i % N;
            atomicAdd(&K[tri_index(r, c)], (u64)v);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) work += __shfl_xor(work, d);
    if ((tid & 63) == 0 && work) atomicAdd(U, work);
}

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

// whole triangle, cells [c0, c0+count) of the reference layout
template <typename SrcT>
__global__ __launch_bounds__(256) void k_triangle(const SrcT* K, const double* diag, u64 c0, u64 count, double* out) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const u64 c = c0 + t;
    u64 i = (u64)((sqrt(8.0 * (double)c + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > c) --i;
    while ((i + 1) * (i + 2) / 2 <= c) ++i;
    const u64 j = c - i * (i + 1) / 2;
    const double x = (double)K[c];
    out[t] = normalised_cell(x, diag[i], diag[j], i == j);
}

// =============================================================================================
// APPROX / VARIANCE MODE  (get_variance, fastsk_kernel.cpp:108-143)
// =============================================================================================
// K_hat' = K_hat + (Ks - K_hat)/iter; prod = delta * (Ks - K_hat') for the train x train prefix.
// The reference then sums prod sequentially in index order; that one reduction stays on the host.
// K_hat is read from one buffer and written to another: the host runs a few iterations ahead of
// its stop test and keeps the state of every iteration it has not yet accepted.
__global__ __launch_bounds__(256) void k_welford(const u64* Ks, const double* K_hat_in, double* K_hat_out, double* prod, u64 pairs,
                                                 u64 train_pairs, double iter) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    const double x = (double)Ks[i];
    const double old = K_hat_in[i];
    const double delta = __dsub_rn(x, old);
    const double kh = __dadd_rn(old, __ddiv_rn(delta, iter));
    K_hat_out[i] = kh;
    if (i < train_pairs) prod[i] = __dmul_rn(delta, __dsub_rn(x, kh));
}

// K += val where val != 0 (fastsk_kernel.cpp:286-315)
template <typename SrcT>
__global__ __launch_bounds__(256) void k_add_nonzero(double* K, const SrcT* src, u64 pairs) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    const double v = (double)src[i];
    if (v != 0.0) K[i] = __dadd_rn(K[i], v);
}

}  // namespace fsk
