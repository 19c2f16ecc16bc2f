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
#define FSK_DMA_KERNEL k_dense_tile_dma
#define FSK_DMA_COMPACT 0
#include "fsk_tile_kernel_dma.inc"
#undef FSK_DMA_KERNEL
#undef FSK_DMA_COMPACT
#define FSK_DMA_KERNEL k_dense_tile_dma_compact
#define FSK_DMA_COMPACT 1
#include "fsk_tile_kernel_dma.inc"
#undef FSK_DMA_KERNEL
#undef FSK_DMA_COMPACT

#include "fsk_sparse_kernels.inc"

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
// The reference then sums prod SEQUENTIALLY in index order (fastsk_kernel.cpp:116-131) and the stop
// test reads that sum to the last bit; k_seq_prep / k_seq_chain below reproduce it on the device.
// K_hat is read from one buffer and written to another: the host runs a few iterations ahead of
// its stop test and keeps the state of every iteration it has not yet accepted.
// bsum[i / SQ_BLOCK] += prod (any order: only a PREDICTION of the running sum's binade is made from it).
constexpr int SQ_BLOCK = 8192;
constexpr int WF_ITEMS = 4;  // cells per thread: 1024 per workgroup, SQ_BLOCK / 1024 workgroups add to one block sum
template <typename SrcT>
__global__ __launch_bounds__(256) void k_welford(const SrcT* Ks, const double* K_hat_in, double* K_hat_out, double* prod, u64 pairs,
                                                 u64 train_pairs, double iter, double* bsum) {
    __shared__ double part[4];
    const u64 base = (u64)blockIdx.x * (256 * WF_ITEMS);
    double pr = 0.0;
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
        const u64 i = base + (u64)q * 256 + threadIdx.x;
        if (i < pairs) {
            const double x = (double)Ks[i];
            const double old = K_hat_in[i];
            const double delta = __dsub_rn(x, old);
            const double kh = __dadd_rn(old, __ddiv_rn(delta, iter));
            K_hat_out[i] = kh;
            if (i < train_pairs) {
                const double v = __dmul_rn(delta, __dsub_rn(x, kh));
                prod[i] = v;
                pr += v;
            }
        }
    }
    // (any order: the block sum only PREDICTS the running sum's binade; the 1024 cells of a workgroup lie
    // inside one SQ_BLOCK)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pr += __shfl_xor(pr, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = pr;
    __syncthreads();
    if (threadIdx.x == 0 && base < train_pairs) {
        const double t = (part[0] + part[1]) + (part[2] + part[3]);
        if (t != 0.0) atomicAdd(&bsum[base / SQ_BLOCK], t);
    }
}

// The same for the `nslots` consecutive iterations of a batch whose counts lie in triangles `pairs`
// cells apart (u32: sparse dataflow, grouped batches; u64: dense dataflow, one storing tile launch per iteration): K_hat is read once, carried through the iterations in
// a register and written once — the state after the batch; the states in between exist only if the
// host asks for them by running a prefix of the batch again (it does when its stop test fires inside
// the batch). Per cell and iteration the same IEEE operations in the same order as k_welford.
// prod / bsum of slot q: prod + q * prod_stride, bsum + q * nblk; write_prod = 0: only K_hat_out.
constexpr int WF_SLOTS = 4;
template <typename SrcT>
__global__ __launch_bounds__(256) void k_welford_batch(const SrcT* Ks, int nslots, const double* K_hat_in, double* K_hat_out, double* prod,
                                                       u64 prod_stride, u64 pairs, u64 train_pairs, double first_iter, double* bsum,
                                                       uint32_t nblk, int write_prod) {
    __shared__ double part[WF_SLOTS][4];
    const u64 base = (u64)blockIdx.x * (256 * WF_ITEMS);
    double pr[WF_SLOTS];
#pragma unroll
    for (int s = 0; s < WF_SLOTS; ++s) pr[s] = 0.0;
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
        const u64 i = base + (u64)q * 256 + threadIdx.x;
        if (i < pairs) {
            SrcT xs[WF_SLOTS];
#pragma unroll
            for (int s = 0; s < WF_SLOTS; ++s) xs[s] = s < nslots ? Ks[(u64)s * pairs + i] : (SrcT)0;  // (all loads before the dependent chain)
            double kh = K_hat_in[i];
#pragma unroll
            for (int s = 0; s < WF_SLOTS; ++s) {
                if (s < nslots) {
                    const double x = (double)xs[s];
                    const double delta = __dsub_rn(x, kh);
                    kh = __dadd_rn(kh, __ddiv_rn(delta, first_iter + (double)s));
                    if (write_prod && i < train_pairs) {
                        const double v = __dmul_rn(delta, __dsub_rn(x, kh));
                        prod[(u64)s * prod_stride + i] = v;
                        pr[s] += v;
                    }
                }
            }
            K_hat_out[i] = kh;
        }
    }
    if (!write_prod || base >= train_pairs) return;  // (uniform per workgroup)
#pragma unroll
    for (int s = 0; s < WF_SLOTS; ++s) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) pr[s] += __shfl_xor(pr[s], d);
        if ((threadIdx.x & 63) == 0) part[s][threadIdx.x >> 6] = pr[s];
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)nslots) {
        const double t = (part[threadIdx.x][0] + part[threadIdx.x][1]) + (part[threadIdx.x][2] + part[threadIdx.x][3]);
        if (t != 0.0) atomicAdd(&bsum[(size_t)threadIdx.x * nblk + base / SQ_BLOCK], t);
    }
}

// ---- exact sequential summation, in parallel ---------------------------------------------------
// s_i = fl(s_{i-1} + p_i), p_i >= 0, round to nearest even. While the running sum stays inside one
// binade [2^e, 2^(e+1)) every s_i is a multiple of u = 2^(e-52), so fl(s + p) = s + R(p) with R(p) the
// multiple of u nearest to p — independent of s unless p lies exactly half-way (a tie: the parity of
// s decides). Hence for a block of values with no tie whose integer total S/u + sum R(p)/u stays
// below 2^53 the sequential result is exactly that integer total times u, whatever the order of the
// additions. k_seq_prep computes sum R(p)/u per block for the binade PREDICTED from approximate block
// sums, on all CUs; k_seq_chain (one wave) walks the blocks with the exact running sum, accepts a
// block when the prediction was right and its conditions hold, and otherwise redoes the block
// itself, in sub-blocks, down to plain sequential additions — so the result is the sequential sum
// for ANY input; only the speed depends on the values being non-negative.
struct SeqBlk {
    int e;            // binade the block's integer total was computed for
    uint32_t flags;   // 1: negative or NaN value, 2: value too large for the integer total, 4: tie, 8: no prediction,
                      // 16: every value of the block is zero (the block changes no running sum),
                      // 32: the block has group records (below)
    u64 Q;            // sum of R(p) / u over the block
};
// A block that will not go through as one integer total — it holds a tie or an outsized value, or the
// approximate sums say the running sum crosses into the next binade inside it — also gets one record per
// group of SQ_GROUP values, for the predicted binade e (records 0..7) and for e + 1 (records 8..15): the
// chain then redoes only the group where the crossing / the tie actually is, not the block.
constexpr int SQ_GROUP = 1024, SQ_GROUPS = 8192 / SQ_GROUP;
struct SeqGrp {
    u64 Q;            // sum of R(p) / u over the group
    uint32_t flags;   // 1, 2, 4, 16 as above
    uint32_t pad;
};
union DblBits { double d; u64 u; };
// unbiased exponent of a normal positive double within +-900, else INT32_MIN (zero, subnormal, huge, NaN)
__device__ __forceinline__ int seq_exponent(double s) {
    DblBits b; b.d = s;
    if (b.u >> 63) return INT32_MIN;
    const int e = (int)((b.u >> 52) & 0x7ffu) - 1023;
    return (e < -900 || e > 900) ? INT32_MIN : e;
}
__device__ __forceinline__ double seq_pow2(int e) {  // 2^e, -1022 <= e <= 1023
    DblBits b; b.u = (u64)(e + 1023) << 52;
    return b.d;
}
// R(p)/u of one value for scale = 1/u, added to q; limit (<= 2^52) bounds a single value so that q cannot
// wrap and so that x + 2^52 is x rounded to the nearest integer (ties to even), held in the low 52 bits
__device__ __forceinline__ void seq_classify(double p, double scale, double limit, u64& q, uint32_t& flags) {
    if (!(p >= 0.0)) { flags |= 1u; return; }  // (-0.0 passes and adds nothing)
    const double x = p * scale;                // exact: a power of two
    if (!(x < limit)) { flags |= 2u; return; }
    DblBits y;
    y.d = __dadd_rn(x, 4503599627370496.0);
    if (fabs(__dsub_rn(x, __dsub_rn(y.d, 4503599627370496.0))) == 0.5) flags |= 4u;  // (both differences are exact)
    q += y.u & 0xfffffffffffffull;
}

// bsum[b] = sum of block b in any order (stand-alone use of the summation; variance mode gets these
// from k_welford); grid = blocks of SQ_BLOCK values
__global__ __launch_bounds__(256) void k_block_sums(const double* p, u64 n, double* bsum) {
    __shared__ double part[4];
    const uint32_t tid = threadIdx.x;
    const u64 lo = (u64)blockIdx.x * SQ_BLOCK;
    double a = 0.0;
    for (int j = 0; j < SQ_BLOCK / 256; ++j) {
        const u64 i = lo + (u64)j * 256 + tid;
        if (i < n) a += p[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    if ((tid & 63u) == 0) part[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) bsum[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// grid = blocks of SQ_BLOCK values, 256 threads
// (blockIdx.y = one of several independent sums laid out `stride` values / `nblk` blocks apart)
__global__ __launch_bounds__(256) void k_seq_prep(const double* p, u64 n, const double* bsum, SeqBlk* blk, u64 stride, uint32_t nblk) {
    __shared__ double s_pre[4];
    __shared__ u64 s_q[4];
    __shared__ uint32_t s_f[4];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    p += (u64)blockIdx.y * stride;
    bsum += (size_t)blockIdx.y * nblk;
    blk += (size_t)blockIdx.y * nblk;
    double pre = 0.0;
    for (uint32_t i = tid; i < b; i += 256) pre += bsum[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
    if ((tid & 63u) == 0) s_pre[tid >> 6] = pre;
    __syncthreads();
    pre = (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]);
    const int e = seq_exponent(pre);
    const double scale = seq_pow2(52 - (e == INT32_MIN ? 0 : e));
    const u64 lo = (u64)b * SQ_BLOCK;
    u64 q = 0;
    uint32_t fl = e == INT32_MIN ? 8u : 0u, nonzero = 0u;
#pragma unroll 4
    for (int j = 0; j < SQ_BLOCK / 256; ++j) {
        const u64 i = lo + (u64)j * 256 + tid;
        if (i < n) {
            const double v = p[i];
            if (v != 0.0) nonzero = 1u;  // (NaN counts as non-zero)
            seq_classify(v, scale, 1125899906842624.0 /* 2^50 */, q, fl);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        q += __shfl_xor(q, d);
        fl |= __shfl_xor(fl, d);
        nonzero |= __shfl_xor(nonzero, d);
    }
    if ((tid & 63u) == 0) { s_q[tid >> 6] = q; s_f[tid >> 6] = fl | (nonzero << 8); }
    __syncthreads();
    if (tid == 0) {
        const uint32_t all = s_f[0] | s_f[1] | s_f[2] | s_f[3];
        const bool zero = (all >> 8) == 0u;
        // group records when the block cannot be one integer total (a tie, an outsized or negative value)
        // or is predicted to end in another binade than it starts in
        const bool need = e != INT32_MIN && !zero && ((all & 7u) != 0u || seq_exponent(pre + bsum[b]) != e);
        blk[b].e = e == INT32_MIN ? 0 : e;
        blk[b].flags = (all & 0xffu) | (zero ? 16u : 0u) | (need ? 32u : 0u);
        blk[b].Q = s_q[0] + s_q[1] + s_q[2] + s_q[3];
    }
}

// the group records of the blocks k_seq_prep marked (a few per sum): same grid; a kernel of its own so
// that the common path above stays lean
__global__ __launch_bounds__(256) void k_seq_prep_groups(const double* p, u64 n, const SeqBlk* blk, u64 stride, uint32_t nblk, SeqGrp* grp) {
    static_assert(SQ_GROUP * SQ_GROUPS == SQ_BLOCK && SQ_GROUP % 256 == 0, "groups of a block");
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    blk += (size_t)blockIdx.y * nblk;
    if (!(blk[b].flags & 32u)) return;  // (uniform)
    __shared__ u64 s_gq[4][2 * SQ_GROUPS];
    __shared__ uint32_t s_gf[4][2 * SQ_GROUPS];
    p += (u64)blockIdx.y * stride;
    grp += ((size_t)blockIdx.y * nblk + b) * (2 * SQ_GROUPS);
    const u64 lo = (u64)b * SQ_BLOCK;
    // thread tid holds values lo + 256 j + tid: j / (SQ_GROUP / 256) is the group
    const double scale = seq_pow2(52 - blk[b].e), scale1 = scale * 0.5;  // binades e and e + 1
    for (int g = 0; g < SQ_GROUPS; ++g) {
        u64 q0 = 0, q1 = 0;
        uint32_t f0 = 0, f1 = 0, nz = 0;
#pragma unroll
        for (int jj = 0; jj < SQ_GROUP / 256; ++jj) {
            const u64 i = lo + (u64)(g * (SQ_GROUP / 256) + jj) * 256 + tid;
            if (i < n) {
                const double v = p[i];
                if (v != 0.0) nz = 1u;
                seq_classify(v, scale, 1125899906842624.0 /* 2^50 */, q0, f0);
                seq_classify(v, scale1, 1125899906842624.0, q1, f1);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            q0 += __shfl_xor(q0, d);
            q1 += __shfl_xor(q1, d);
            f0 |= __shfl_xor(f0, d);
            f1 |= __shfl_xor(f1, d);
            nz |= __shfl_xor(nz, d);
        }
        if ((tid & 63u) == 0) {
            s_gq[tid >> 6][g] = q0; s_gq[tid >> 6][SQ_GROUPS + g] = q1;
            s_gf[tid >> 6][g] = f0 | (nz << 8); s_gf[tid >> 6][SQ_GROUPS + g] = f1 | (nz << 8);
        }
    }
    __syncthreads();
    if (tid < 2u * SQ_GROUPS) {
        const uint32_t all = s_gf[0][tid] | s_gf[1][tid] | s_gf[2][tid] | s_gf[3][tid];
        SeqGrp r;
        r.Q = s_gq[0][tid] + s_gq[1][tid] + s_gq[2][tid] + s_gq[3][tid];
        r.flags = (all & 7u) | ((all >> 8) ? 0u : 16u);
        r.pad = 0u;
        grp[tid] = r;
    }
}

// value of lane `src` (wave-uniform index) in every lane: a scalar read, not an LDS permute
__device__ __forceinline__ uint32_t wave_bcast_u32(uint32_t x, uint32_t src) {
#ifdef FSK_EMU
    return __shfl(x, (int)src);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)src);
#endif
}
__device__ __forceinline__ u64 wave_bcast_u64(u64 x, uint32_t src) {
    return ((u64)wave_bcast_u32((uint32_t)(x >> 32), src) << 32) | wave_bcast_u32((uint32_t)x, src);
}
__device__ __forceinline__ double wave_bcast_f64(double x, uint32_t src) {
    DblBits b; b.d = x;
    b.u = wave_bcast_u64(b.u, src);
    return b.d;
}

// sum of x over the 64 lanes, in every lane. On the critical path of k_seq_chain (one wave, nothing to
// overlap with), so through the DPP row shifts / broadcasts of the VALU rather than six dependent LDS
// permutes per 32-bit half: rows of 16 lanes first, then row 0 -> 1 and 2 -> 3, then lane 31 -> rows 2-3.
__device__ __forceinline__ u64 wave_sum_u64(u64 x) {
#ifdef FSK_EMU
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    return x;
#else
#define FSK_DPP_ADD64(ctrl, rows)                                                                                   \
    {                                                                                                               \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rows, 0xf, true);      \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rows, 0xf, true); \
        x += ((u64)hi << 32) | lo;                                                                                  \
    }
    FSK_DPP_ADD64(0x111, 0xf)  // row_shr:1
    FSK_DPP_ADD64(0x112, 0xf)  // row_shr:2
    FSK_DPP_ADD64(0x114, 0xf)  // row_shr:4
    FSK_DPP_ADD64(0x118, 0xf)  // row_shr:8  -> lane 15 of every row holds the row's sum
    FSK_DPP_ADD64(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FSK_DPP_ADD64(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef FSK_DPP_ADD64
    return wave_bcast_u64(x, 63u);
#endif
}

// PER values per lane (held in registers; 0.0 where the range ended) added to s as one integer total,
// if the conditions hold
template <int PER>
__device__ __forceinline__ bool seq_try_regs(const double (&v)[PER], double& s) {
    const int e = seq_exponent(s);
    const bool usable = e != INT32_MIN && s > 0.0;
    const double scale = seq_pow2(52 - (usable ? e : 0));
    u64 q = 0;
    uint32_t fl = 0, nonzero = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (v[k] != 0.0) nonzero = 1u;
        seq_classify(v[k], scale, 4503599627370496.0 /* 2^52 */, q, fl);
    }
    if (__ballot(nonzero != 0u) == 0ull) return true;  // zeros change no running sum, whatever it is (s starts at +0 and +0 + -0 = +0)
    if (!usable || __ballot(fl != 0u) != 0ull) return false;
    const u64 tot = (u64)(s * scale) + wave_sum_u64(q);
    if (tot >= ((u64)1 << 53)) return false;
    s = (double)tot * seq_pow2(e - 52);
    return true;
}
// The values [lo, hi) added to s by one wave (a workgroup of its own): SQ_STAGE values at a time go
// through registers into LDS — the loads of the next stage are in flight while this one is summed — and
// are taken from there in groups of 1024 (16 per lane, value 64 k + lane in register k), then for a
// group that crosses a binade, holds a tie or a negative value in its 16 sub-groups of 64, then by plain
// sequential additions. s and every decision are wave-uniform.
constexpr int SQ_STAGE = 4096;
__device__ __forceinline__ double seq_range(const double* p, u64 lo, u64 hi, double s, double* stage) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int PER = SQ_STAGE / 64;
    double nx[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const u64 i = lo + (u64)k * 64 + lane;
        nx[k] = i < hi ? p[i] : 0.0;
    }
    for (u64 c0 = lo; c0 < hi; c0 += SQ_STAGE) {
        __syncthreads();  // (one wave: orders the LDS reads of the stage before with these writes)
#pragma unroll
        for (int k = 0; k < PER; ++k) stage[k * 64 + (int)lane] = nx[k];
        if (c0 + SQ_STAGE < hi) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const u64 i = c0 + SQ_STAGE + (u64)k * 64 + lane;
                nx[k] = i < hi ? p[i] : 0.0;
            }
        }
        __syncthreads();
        const uint32_t here = hi - c0 < (u64)SQ_STAGE ? (uint32_t)(hi - c0) : (uint32_t)SQ_STAGE;
        for (uint32_t g0 = 0; g0 < here; g0 += 1024u) {
            double cur[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) cur[k] = stage[g0 + (uint32_t)k * 64u + lane];  // (0.0 beyond hi)
            if (seq_try_regs<16>(cur, s)) continue;
#pragma unroll 1
            for (uint32_t k = 0; k < 16u; ++k) {  // (rolled: a rare path, kept small)
                const uint32_t q0 = g0 + k * 64u;
                const double one[1] = {stage[q0 + lane]};
                if (!seq_try_regs<1>(one, s)) {
                    const uint32_t cn = q0 >= here ? 0u : (here - q0 < 64u ? here - q0 : 64u);
                    for (uint32_t j = 0; j < cn; ++j) s = __dadd_rn(s, wave_bcast_f64(one[0], j));
                }
            }
        }
    }
    return s;
}

// one wave: out[0] = the sequential sum of p[0..n); zeroes bsum for the slot's next use
// (blockIdx.x = one of several independent sums laid out `stride` values / `nblocks` blocks apart)
__global__ __launch_bounds__(64) void k_seq_chain(const double* p, u64 n, const SeqBlk* blk, uint32_t nblocks, double* bsum, double* out,
                                                  u64 stride, const SeqGrp* grp) {
    __shared__ double stage[SQ_STAGE];
    const uint32_t lane = threadIdx.x;
    p += (u64)blockIdx.x * stride;
    blk += (size_t)blockIdx.x * nblocks;
    grp += (size_t)blockIdx.x * nblocks * (2 * SQ_GROUPS);
    bsum += (size_t)blockIdx.x * nblocks;
    out += blockIdx.x;
    // the running sum is kept either as a double s, or — between accepted blocks of one binade — as the
    // integer S = s / 2^(e-52) in [2^52, 2^53), so that an accepted block is one 64-bit add and a compare
    double s = 0.0;
    bool as_int = false;
    int e = 0;
    u64 S = 0;
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 64) {
        // 64 block records at a time, one per lane, handed round with scalar reads (no dependent loads in the chain)
        SeqBlk mine;
        mine.e = 0; mine.flags = 8u; mine.Q = 0;
        if (c0 + lane < nblocks) mine = blk[c0 + lane];
        const uint32_t cn = nblocks - c0 < 64u ? nblocks - c0 : 64u;
        for (uint32_t j = 0; j < cn; ++j) {
            const uint32_t kf = wave_bcast_u32(mine.flags, j);
            if (kf & 16u) continue;  // all zeros
            if (kf == 0u) {
                const int ke = (int)wave_bcast_u32((uint32_t)mine.e, j);
                const u64 kq = wave_bcast_u64(mine.Q, j);
                if (!as_int && s > 0.0 && seq_exponent(s) == ke) {
                    e = ke;
                    S = (u64)(s * seq_pow2(52 - e));
                    as_int = true;
                }
                if (as_int && e == ke && S + kq < ((u64)1 << 53)) {
                    S += kq;
                    continue;
                }
            }
            const u64 lo = (u64)(c0 + j) * SQ_BLOCK, hi = lo + SQ_BLOCK < n ? lo + SQ_BLOCK : n;
            if (kf & 32u) {
                // group records: the groups before and after the crossing / the tie go through as integer
                // totals of their binade; only the group in between is redone
                const int ke = (int)wave_bcast_u32((uint32_t)mine.e, j);
                SeqGrp mg;
                mg.Q = 0; mg.flags = 1u; mg.pad = 0u;
                if (lane < 2u * SQ_GROUPS) mg = grp[(size_t)(c0 + j) * (2 * SQ_GROUPS) + lane];
                for (uint32_t g = 0; g < (uint32_t)SQ_GROUPS; ++g) {
                    const u64 glo = lo + (u64)g * SQ_GROUP;
                    if (glo >= hi) break;
                    if (wave_bcast_u32(mg.flags, g) & 16u) continue;  // all zeros
                    if (!as_int && s > 0.0 && seq_exponent(s) != INT32_MIN) {
                        e = seq_exponent(s);
                        S = (u64)(s * seq_pow2(52 - e));
                        as_int = true;
                    }
                    if (as_int && (e == ke || e == ke + 1)) {
                        const uint32_t r = (e == ke ? 0u : (uint32_t)SQ_GROUPS) + g;
                        const u64 gq = wave_bcast_u64(mg.Q, r);
                        if (wave_bcast_u32(mg.flags, r) == 0u && S + gq < ((u64)1 << 53)) {
                            S += gq;
                            continue;
                        }
                    }
                    if (as_int) {
                        s = (double)S * seq_pow2(e - 52);
                        as_int = false;
                    }
                    const u64 ghi = glo + SQ_GROUP < hi ? glo + SQ_GROUP : hi;
                    s = seq_range(p, glo, ghi, s, stage);
                }
                continue;
            }
            if (as_int) {
                s = (double)S * seq_pow2(e - 52);
                as_int = false;
            }
            s = seq_range(p, lo, hi, s, stage);
        }
    }
    if (as_int) s = (double)S * seq_pow2(e - 52);
    if (lane == 0) {
        out[0] = s;
        __threadfence_system();  // (`out` may be pinned host memory: the variance mode reads it after the stream's event)
    }
    for (uint32_t b = lane; b < nblocks; b += 64) bsum[b] = 0.0;
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
