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

// =============================================================================================
// SPARSE PATH — the reference's dataflow (gather -> sort -> run-length -> K +=), as streams
// =============================================================================================
// Per batch of B combos ("slots"):
//   k_sx_extract   one packed record per (slot, g-mer): rec = (k-mer << sb) | sequence id, u32 when
//                  that fits 32 bits (every BASELINE config), else u64, else 128 bits; slot s owns rec[s*nfeat ..):
//                  the slot is implicit in the position, so B independent sorts run in one launch.
//   k_sx_hist / k_sx_scan_slot / k_sx_scatter   stable LSD radix sort over the k-mer bits only
//                  (records are generated in sequence order and every pass is stable), 8-bit digits,
//                  4096-record tiles ranked with wave64 ballot matching, permuted in LDS and written
//                  out in digit runs.
//   k_sx_seg_count / k_sx_seg_scan / k_sx_seg_write   sorted records -> compact entries
//                  E[e] = {sequence, multiplicity} (distinct (k-mer, sequence) pairs, compacted through
//                  LDS so the write is coalesced) and Pk[e] = rank of the entry inside its run of
//                  equal k-mers, counted from 1: entry e pairs with entries e-Pk[e]+1 .. e — exactly
//                  the `+=` of countAndUpdateTri (shared.cpp:316-327).
//   k_sx_emit      every (entry, partner) pair becomes one 32-bit update word {cell inside its owner's
//                  band of rows of K, product of the two multiplicities}; the words of a tile are
//                  binned by owner in LDS and leave as contiguous runs (long entries write their
//                  partner range directly, one wave per entry).
//   k_sx_consume   one workgroup per owner band sums its update stream in LDS (u32 cells) and adds
//                  the non-zero cells into the 64-bit triangle, row-contiguous, no atomics needed
//                  (a cell has one owner; launches are ordered on the stream).
// When a band of K does not fit the LDS budget (very large N) or FSK_SPARSE_GLOBAL=1, k_sx_emit adds
// every product into K with a 64-bit atomicAdd instead (DIRECT).
constexpr int SX_TILE = 4096;           // records per sort tile
constexpr int SX_ITEMS = SX_TILE / 256;
constexpr int SG_TILE = 2048;           // records per segment tile = most entries one emit workgroup holds
constexpr int SG_ITEMS = SG_TILE / 256;
constexpr int SX_MAX_OWNERS = 512;      // owner bands of K (bins of the update streams)
constexpr uint32_t SX_SHORT = 16;       // entries with up to this many partners are binned by owner in LDS

// block-wide exclusive running maximum of one int per thread (256 threads); identity = -1 (all
// values are >= -1). Every thread of the block must call it.
__device__ __forceinline__ int block_excl_maxscan_256(int v, int* tmp, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d && y > x) x = y;
    }
    int prev = __shfl_up(x, 1);
    if (lane == 0) prev = -1;
    __syncthreads();  // tmp may still be read from a previous call
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    int base = -1, tot = -1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = tmp[w];
        if (w < wave && t > base) base = t;
        if (t > tot) tot = t;
    }
    if (total) *total = tot;
    return base > prev ? base : prev;
}

// rec = (k-mer << sb) | sequence id, one per (slot, g-mer), and the first sort pass's digit histogram
// of the tile (blockhist[slot][tile][digit], digit = the lowest k-mer bits under dmask); grid = (sort tiles per slot, slots)
template <typename RecT>
__global__ __launch_bounds__(256) void k_sx_extract(SeqView S, const uint32_t* feat_seq, const uint32_t* fstart, uint32_t nfeat, uint32_t tps,
                                                    int k, uint32_t sigma, int sb, const uint8_t* combo_pos, RecT* rec,
                                                    uint32_t* blockhist, uint32_t dmask) {
    __shared__ uint32_t h[256];
    __shared__ uint8_t s_pos[16];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, slot = blockIdx.y;
    h[tid] = 0u;
    if (tid < 16u) s_pos[tid] = (int)tid < k ? combo_pos[(size_t)slot * k + tid] : (uint8_t)0;
    __syncthreads();
    const uint8_t* pos = k <= 16 ? s_pos : combo_pos + (size_t)slot * k;
    const uint32_t base = tile * (uint32_t)SX_TILE;
#pragma unroll 4
    for (int it = 0; it < SX_ITEMS; ++it) {
        const uint32_t f = base + (uint32_t)it * 256u + tid;
        if (f < nfeat) {
            const uint32_t seq = feat_seq[f];
            const uint32_t j = f - fstart[seq];
            const uint32_t wbase = S.wstart[seq];
            u64 key = 0;
            for (int c = 0; c < k; ++c) key = key * sigma + fetch_sym(S.words, wbase, j + pos[c], S.bits);
            rec[(size_t)slot * nfeat + f] = ((RecT)key << sb) | (RecT)seq;
            atomicAdd(&h[(uint32_t)key & dmask], 1u);
        }
    }
    __syncthreads();
    blockhist[((size_t)slot * tps + tile) * 256u + tid] = h[tid];
}

// digit histogram of one tile of one slot -> blockhist[slot][tile][digit]; grid = (tiles per slot, slots)
template <typename RecT>
__global__ __launch_bounds__(256) void k_sx_hist(const RecT* rec, uint32_t nfeat, uint32_t tps, int shift, uint32_t dmask,
                                                 uint32_t* blockhist) {
    __shared__ uint32_t h[256];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, slot = blockIdx.y;
    h[tid] = 0u;
    __syncthreads();
    const RecT* r = rec + (size_t)slot * nfeat;
    const uint32_t base = tile * (uint32_t)SX_TILE;
#pragma unroll
    for (int it = 0; it < SX_ITEMS; ++it) {
        const uint32_t i = base + (uint32_t)it * 256u + tid;
        if (i < nfeat) atomicAdd(&h[(uint32_t)(r[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    blockhist[((size_t)slot * tps + tile) * 256u + tid] = h[tid];
}

// One workgroup of 1024 threads per slot, thread = (quarter of the slot's tiles, digit): the digit's tile
// counts become exclusive offsets inside the digit's block (in place), and dbase[slot][digit] = where
// that block starts inside the slot. Two passes over the quarter (sum, then write), so that four
// chains of dependent loads run side by side.
__global__ __launch_bounds__(1024) void k_sx_scan_slot(uint32_t* blockhist, uint32_t tps, uint32_t* dbase) {
    __shared__ uint32_t part[4][256];
    __shared__ uint32_t tmp[16];
    const uint32_t tid = threadIdx.x, digit = tid & 255u, quarter = tid >> 8, slot = blockIdx.x;
    uint32_t* h = blockhist + (size_t)slot * tps * 256u + digit;
    const uint32_t per = (tps + 3u) / 4u;
    const uint32_t t0 = quarter * per < tps ? quarter * per : tps, t1 = t0 + per < tps ? t0 + per : tps;
    uint32_t sum = 0;
#pragma unroll 4
    for (uint32_t t = t0; t < t1; ++t) sum += h[(size_t)t * 256u];
    part[quarter][digit] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t v = part[q][digit];
        if (q < quarter) run += v;
        total += v;
    }
#pragma unroll 4
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t v = h[(size_t)t * 256u];
        h[(size_t)t * 256u] = run;
        run += v;
    }
    const uint32_t ex = block_excl_scan<uint32_t, 16>(quarter == 0 ? total : 0u, tmp, nullptr);
    if (quarter == 0) dbase[(size_t)slot * 256u + digit] = ex;
}

// exclusive scan of n totals (single workgroup), off[n] = grand total
__global__ __launch_bounds__(256) void k_scan_totals(const uint32_t* totals_in, uint32_t n, uint32_t* off) {
    __shared__ uint32_t tmp[4];
    const int tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t v = i < n ? totals_in[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256<uint32_t>(v, tmp, &tot) + carry;
        if (i < n) off[i] = ex;
        carry += tot;
    }
    if (tid == 0) off[n] = carry;
}

// Stable scatter of one tile of one slot. Wave w owns the contiguous quarter w of the tile, so the
// four waves rank their records independently (wave64 ballot matching: 8 ballots give the lanes
// holding the same digit; a per-wave LDS counter carries the digit's count from round to round)
// and meet at ONE barrier; the records are then permuted into digit order in LDS and written out
// so that equal digits go to consecutive addresses. grid = (tiles per slot, slots)
template <typename RecT>
__global__ __launch_bounds__(256) void k_sx_scatter(const RecT* in, RecT* out, uint32_t nfeat, uint32_t tps, int shift, int nbits,
                                                    const uint32_t* blockhist, const uint32_t* dbase) {
    __shared__ uint32_t goff[256];         // destination, inside the slot, of this tile's first record of each digit
    __shared__ uint32_t wave_run[4][256];  // per wave: records of the digit so far; later: where the wave's share starts in LDS
    __shared__ uint32_t blk_start[256];    // exclusive scan of the tile's digit totals
    __shared__ uint32_t tmp[4];
    __shared__ RecT s_rec[SX_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.x, slot = blockIdx.y;
    const uint32_t dmask = (1u << nbits) - 1u;
    const RecT* r = in + (size_t)slot * nfeat;
    RecT* o = out + (size_t)slot * nfeat;
    goff[tid] = dbase[(size_t)slot * 256u + tid] + blockhist[((size_t)slot * tps + tile) * 256u + tid];
#pragma unroll
    for (int q = 0; q < 4; ++q) wave_run[q][tid] = 0u;
    __syncthreads();
    RecT key[SX_ITEMS];
    uint32_t rank[SX_ITEMS];
    const uint32_t base = tile * (uint32_t)SX_TILE + (uint32_t)wave * (SX_TILE / 4);
    volatile uint32_t* my_run = wave_run[wave];
#pragma unroll
    for (int it = 0; it < SX_ITEMS; ++it) {
        if (base + (uint32_t)it * 64u >= nfeat) {  // (wave-uniform) nothing of the slot left for this wave
            key[it] = (RecT)0;
            rank[it] = 0xffffffffu;
            continue;
        }
        const uint32_t i = base + (uint32_t)it * 64u + (uint32_t)lane;
        const bool valid = i < nfeat;
        key[it] = valid ? r[i] : (RecT)0;
        const uint32_t digit = (uint32_t)(key[it] >> shift) & dmask;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b < nbits) {  // (uniform) a pass sorts nbits <= 8 bits: the k-mer bits are split evenly over the passes
                const bool bit = (digit >> b) & 1u;
                const u64 m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t prior = valid ? my_run[digit] : 0u;
        (void)__ballot(true);  // every lane has read the counter before the group's first lane adds to it
        if (valid && rank_in_wave == 0) atomicAdd(&wave_run[wave][digit], (uint32_t)__popcll(peers));
        rank[it] = valid ? prior + rank_in_wave : 0xffffffffu;
    }
    __syncthreads();
    {
        uint32_t c[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = wave_run[q][tid]; tot += c[q]; }
        uint32_t ex = block_excl_scan_256<uint32_t>(tot, tmp, nullptr);
        blk_start[tid] = ex;
#pragma unroll
        for (int q = 0; q < 4; ++q) { wave_run[q][tid] = ex; ex += c[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SX_ITEMS; ++it) {
        if (rank[it] != 0xffffffffu) {
            const uint32_t digit = (uint32_t)(key[it] >> shift) & dmask;
            s_rec[wave_run[wave][digit] + rank[it]] = key[it];
        }
    }
    __syncthreads();
    const uint32_t first = tile * (uint32_t)SX_TILE;
    const uint32_t nvalid = nfeat - first < (uint32_t)SX_TILE ? nfeat - first : (uint32_t)SX_TILE;
#pragma unroll
    for (int it = 0; it < SX_ITEMS; ++it) {
        const uint32_t p = (uint32_t)it * 256u + (uint32_t)tid;
        if (p < nvalid) {
            const RecT kx = s_rec[p];
            const uint32_t digit = (uint32_t)(kx >> shift) & dmask;
            o[goff[digit] + (p - blk_start[digit])] = kx;
        }
    }
}

// ---- segments --------------------------------------------------------------------------------
// Inside a slot, record j starts an ENTRY when it differs from record j-1 (new k-mer or new sequence)
// and a RUN when its k-mer differs; j = 0 starts both. Thread t of a tile looks at SG_ITEMS
// consecutive records.
// skip_from (skip_test_block): sequences >= skip_from are test sequences. Inside a run the entries are
// in sequence order, train entries first; the first test entry of a run is a TEST HEAD. tile_lth gets the
// tile-local index of the tile's last test head (only when skip_from != 0xffffffff).
template <typename RecT>
__global__ __launch_bounds__(256) void k_sx_seg_count(const RecT* rec, uint32_t nfeat, uint32_t tpg, int sb, uint32_t* tile_ent,
                                                      int* tile_lrh, uint32_t skip_from, int* tile_lth) {
    __shared__ uint32_t tmp[4];
    __shared__ int s_lrh, s_lth;
    const uint32_t tid = threadIdx.x, t = blockIdx.x, slot = blockIdx.y;
    const RecT* r = rec + (size_t)slot * nfeat;
    const uint32_t j0 = t * (uint32_t)SG_TILE + tid * (uint32_t)SG_ITEMS;
    if (tid == 0) { s_lrh = -1; s_lth = -1; }
    const RecT seq_mask = (RecT)(((u64)1 << sb) - 1);
    RecT prev = (j0 > 0 && j0 <= nfeat) ? r[j0 - 1] : (RecT)0;
    uint32_t n = 0;
    int lrh = -1, lth = -1;  // among this thread's entries, the last one that starts a run / is a test head
#pragma unroll
    for (int q = 0; q < SG_ITEMS; ++q) {
        const uint32_t j = j0 + (uint32_t)q;
        if (j < nfeat) {
            const RecT cur = r[j];
            if (j == 0 || cur != prev) {
                const bool run_head = j == 0 || (cur >> sb) != (prev >> sb);
                if (run_head) lrh = (int)n;
                if ((uint32_t)(cur & seq_mask) >= skip_from && (run_head || (uint32_t)(prev & seq_mask) < skip_from)) lth = (int)n;
                ++n;
            }
            prev = cur;
        }
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256<uint32_t>(n, tmp, &tot);
    if (lrh >= 0) atomicMax(&s_lrh, (int)ex + lrh);
    if (lth >= 0) atomicMax(&s_lth, (int)ex + lth);
    __syncthreads();
    if (tid == 0) {
        tile_ent[(size_t)slot * tpg + t] = tot;
        tile_lrh[(size_t)slot * tpg + t] = s_lrh;  // tile-local index of the last entry that starts a run, or -1
        if (tile_lth) tile_lth[(size_t)slot * tpg + t] = s_lth;
    }
}

// running maximum, exclusive, over a block of NW waves; identity -1. tmp: >= NW ints.
template <int NW>
__device__ __forceinline__ int block_excl_maxscan(int v, int* tmp, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d && y > x) x = y;
    }
    int prev = __shfl_up(x, 1);
    if (lane == 0) prev = -1;
    __syncthreads();
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    int base = -1, tot = -1;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int t = tmp[w];
        if (w < wave && t > base) base = t;
        if (t > tot) tot = t;
    }
    if (total) *total = tot;
    return base > prev ? base : prev;
}

// single workgroup of 1024 threads: ebase[tile] = entries before the tile (ebase[ntiles] = D);
// tile_rs[tile] = global index of the last run start before the tile (the run the tile's first entries
// may continue); tile_ts[tile] = likewise the last test head before the tile (skip_test_block only)
__global__ __launch_bounds__(1024) void k_sx_seg_scan(const uint32_t* tile_ent, const int* tile_lrh, uint32_t ntiles, uint32_t* ebase,
                                                      int* tile_rs, const int* tile_lth, int* tile_ts) {
    __shared__ uint32_t tmp[16];
    __shared__ int tmpi[16];
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0;
    int carry_h = -1, carry_t = -1;
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < ntiles ? tile_ent[i] : 0u;
        const int lr = i < ntiles ? tile_lrh[i] : -1;
        const int lt = (tile_lth && i < ntiles) ? tile_lth[i] : -1;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<uint32_t, 16>(v, tmp, &tot) + carry;
        const int h = lr >= 0 ? (int)ex + lr : -1;
        int htot;
        const int hx = block_excl_maxscan<16>(h, tmpi, &htot);
        if (i < ntiles) {
            ebase[i] = ex;
            tile_rs[i] = hx > carry_h ? hx : carry_h;
        }
        if (tile_lth) {  // (uniform)
            const int th = lt >= 0 ? (int)ex + lt : -1;
            int ttot;
            const int tx = block_excl_maxscan<16>(th, tmpi, &ttot);
            if (i < ntiles) tile_ts[i] = tx > carry_t ? tx : carry_t;
            if (ttot > carry_t) carry_t = ttot;
        }
        carry += tot;
        if (htot > carry_h) carry_h = htot;
    }
    if (tid == 0) ebase[ntiles] = carry;
}

// Owner band of row i: band o holds the rows whose first cell index tri_index(i, 0) lies in
// [o << own_shift, (o + 1) << own_shift), i.e. owner_r0[o] <= i < owner_r0[o + 1].
__device__ __forceinline__ uint32_t sx_owner_of(uint32_t seq, int own_shift) {
    return (uint32_t)(tri_index((u64)seq, 0) >> own_shift);
}
// update words one (entry, partner) pair becomes: 1 while multiplicity * max_windows cannot overflow the
// product field, i.e. multiplicity <= cmax = maxprod / max_windows (everything but extreme
// low-complexity sequences); beyond that every pair of the entry takes `S` words
__device__ __forceinline__ uint32_t sx_words_per_pair(uint32_t count, uint32_t cmax, uint32_t max_win, uint32_t maxprod) {
    if (count <= cmax) return 1u;
    return (uint32_t)(((u64)count * max_win + maxprod - 1) / maxprod);
}

// Entries of one tile: E (sequence, multiplicity), Pk (rank in run), and — unless `ucount` is null —
// the number of update words the tile will emit per owner band, ucount[tile][owner].
// stats[0] += pairs (the reference's `+=` count U), stats[1] += update words.
// skip_test_block (skip_from != 0xffffffff): Tk[e] = how many of the P partners of entry e, counted from
// the run start, it really pairs with before itself: all P - 1 for a train row; for a test row only the
// run's train entries (test x test cells other than the diagonal are left alone) — P minus its rank among
// the run's test entries. An entry then has Tk + 1 partners: Tk from the run start, and itself.
template <typename RecT>
__global__ __launch_bounds__(256) void k_sx_seg_write(const RecT* rec, uint32_t nfeat, uint32_t tpg, int sb, const uint32_t* ebase,
                                                      const int* tile_rs, uint2* E, uint32_t* Pk, int own_shift, uint32_t n_owners,
                                                      uint32_t* ucount, uint32_t row0, uint32_t row1, uint32_t max_win,
                                                      uint32_t maxprod, uint32_t cmax, u64* tile_stat, uint32_t skip_from,
                                                      const int* tile_ts, uint32_t* Tk) {
    __shared__ uint32_t tmp[4];
    __shared__ int tmpi[4];
    __shared__ __attribute__((aligned(8))) uint32_t s_pos[SG_TILE + 2];
    __shared__ uint2 s_ent[SG_TILE];
    __shared__ uint32_t s_P[SG_TILE];
    __shared__ uint32_t s_cnt[SX_MAX_OWNERS];
    __shared__ uint32_t s_end;
    const uint32_t tid = threadIdx.x, t = blockIdx.x, slot = blockIdx.y;
    const uint32_t tile = slot * tpg + t;
    const RecT* r = rec + (size_t)slot * nfeat;
    const uint32_t first = t * (uint32_t)SG_TILE;
    const uint32_t j0 = first + tid * (uint32_t)SG_ITEMS;
    if (ucount)
        for (uint32_t i = tid; i < n_owners; i += 256) s_cnt[i] = 0u;
    RecT cur[SG_ITEMS];
    uint32_t eh = 0, rh = 0, th = 0, n = 0;
    int lrh = -1, lth = -1;
    const RecT seq_mask = (RecT)(((u64)1 << sb) - 1);
    {
        RecT prev = (j0 > 0 && j0 <= nfeat) ? r[j0 - 1] : (RecT)0;
#pragma unroll
        for (int q = 0; q < SG_ITEMS; ++q) {
            const uint32_t j = j0 + (uint32_t)q;
            cur[q] = j < nfeat ? r[j] : (RecT)0;
            if (j < nfeat) {
                if (j == 0 || cur[q] != prev) {
                    eh |= 1u << q;
                    const bool run_head = j == 0 || (cur[q] >> sb) != (prev >> sb);
                    if (run_head) { rh |= 1u << q; lrh = (int)n; }
                    if ((uint32_t)(cur[q] & seq_mask) >= skip_from && (run_head || (uint32_t)(prev & seq_mask) < skip_from)) {
                        th |= 1u << q;
                        lth = (int)n;
                    }
                    ++n;
                }
                prev = cur[q];
            }
        }
    }
    if (tid == 255) {
        // where the tile's last entry ends: it may run on into the following tiles (bounded by the
        // number of windows of one sequence); the first look-ahead load rides with the loads above
        const uint32_t nv = nfeat - first < (uint32_t)SG_TILE ? nfeat - first : (uint32_t)SG_TILE;
        uint32_t j = first + nv;
        if (nv == (uint32_t)SG_TILE && j < nfeat && r[j] == cur[SG_ITEMS - 1]) {
            ++j;
            while (j < nfeat && r[j] == r[j - 1]) ++j;
        }
        s_end = j - first;
    }
    uint32_t n_tile;
    const uint32_t ex = block_excl_scan_256<uint32_t>(n, tmp, &n_tile);
    // tile-local index of the run start that governs this thread's first entries (-1: before the tile)
    int head = block_excl_maxscan_256(lrh >= 0 ? (int)ex + lrh : -1, tmpi, nullptr);
    const bool skipping = skip_from != 0xffffffffu;  // (uniform)
    int thead = -1;
    if (skipping) thead = block_excl_maxscan_256(lth >= 0 ? (int)ex + lth : -1, tmpi, nullptr);
    const uint32_t eb = ebase[tile];
    const int before = tile_rs[tile];
    const int before_t = skipping ? tile_ts[tile] : -1;
    {
        uint32_t e = ex;
#pragma unroll
        for (int q = 0; q < SG_ITEMS; ++q) {
            if (eh & (1u << q)) {
                if (rh & (1u << q)) head = (int)e;
                if (th & (1u << q)) thead = (int)e;
                const uint32_t rs = head >= 0 ? eb + (uint32_t)head : (uint32_t)before;
                const uint32_t seq = (uint32_t)(cur[q] & seq_mask);
                const uint32_t P = eb + e - rs + 1u;
                s_pos[e] = tid * (uint32_t)SG_ITEMS + (uint32_t)q;
                s_P[e] = P;
                s_ent[e].x = seq;
                // partners before itself: everything from the run start (train row), or only the run's train
                // entries = P - (rank among the run's test entries)
                uint32_t T = P - 1u;
                if (seq >= skip_from) {
                    const uint32_t ts = thead >= 0 ? eb + (uint32_t)thead : (uint32_t)before_t;
                    T = P - (eb + e - ts + 1u);
                }
                s_ent[e].y = T;  // (parked here until the multiplicity is known)
                ++e;
            }
        }
    }
    __syncthreads();
    u64 pairs = 0, words = 0;
    for (uint32_t e = tid; e < n_tile; e += 256) {
        const uint32_t c = (e + 1 < n_tile ? s_pos[e + 1] : s_end) - s_pos[e];
        const uint32_t seq = s_ent[e].x, P = s_P[e], T = s_ent[e].y;
        E[(size_t)eb + e] = make_uint2(seq, c);
        Pk[(size_t)eb + e] = P;
        if (Tk) Tk[(size_t)eb + e] = T;
        if (seq >= row0 && seq < row1) {
            const u64 w = (u64)(T + 1u) * sx_words_per_pair(c, cmax, max_win, maxprod);
            pairs += T + 1u;
            words += w;
            if (ucount) atomicAdd(&s_cnt[sx_owner_of(seq, own_shift)], (uint32_t)w);
        }
    }
    if (ucount) {
        __syncthreads();
        for (uint32_t o = tid; o < n_owners; o += 256) ucount[(size_t)tile * n_owners + o] = s_cnt[o];
    }
    // (one record per tile, summed by k_sx_stat_sum: hundreds of thousands of atomics on one address
    // would serialise at the memory side)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pairs += __shfl_xor(pairs, d);
        words += __shfl_xor(words, d);
    }
    __syncthreads();  // s_pos is free now
    u64* red = reinterpret_cast<u64*>(s_pos);
    if ((tid & 63u) == 0) { red[2 * (tid >> 6)] = pairs; red[2 * (tid >> 6) + 1] = words; }
    __syncthreads();
    if (tid == 0) {
        tile_stat[2 * (size_t)tile] = red[0] + red[2] + red[4] + red[6];
        tile_stat[2 * (size_t)tile + 1] = red[1] + red[3] + red[5] + red[7];
    }
}

// stats[0] += sum of tile_stat[2t], stats[1] += sum of tile_stat[2t + 1]; grid = 32 workgroups
__global__ __launch_bounds__(256) void k_sx_stat_sum(const u64* tile_stat, uint32_t ntiles, u64* stats) {
    u64 a = 0, b = 0;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < ntiles; t += gridDim.x * 256u) {
        a += tile_stat[2 * (size_t)t];
        b += tile_stat[2 * (size_t)t + 1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
    }
    if ((threadIdx.x & 63u) == 0 && (a | b)) {
        atomicAdd(&stats[0], a);
        atomicAdd(&stats[1], b);
    }
}

// ---- offsets of the update streams: column scan of ucount[tile][owner] -----------------------
constexpr int UC_CHUNK = 64;  // tiles per chunk
// chunk_tot[chunk][owner] = words of the chunk's tiles for the owner; grid = chunks
__global__ __launch_bounds__(256) void k_sx_ucol_sum(const uint32_t* ucount, uint32_t ntiles, uint32_t n_owners, uint32_t* chunk_tot) {
    const uint32_t chunk = blockIdx.x, t0 = chunk * (uint32_t)UC_CHUNK;
    const uint32_t t1 = t0 + UC_CHUNK < ntiles ? t0 + UC_CHUNK : ntiles;
    for (uint32_t o = threadIdx.x; o < n_owners; o += 256) {
        uint32_t s = 0;
#pragma unroll 8
        for (uint32_t t = t0; t < t1; ++t) s += ucount[(size_t)t * n_owners + o];
        chunk_tot[(size_t)chunk * n_owners + o] = s;
    }
}
// per owner: exclusive scan over the chunks in place, owner total to utot; grid = owners
__global__ __launch_bounds__(256) void k_sx_ucol_scan(uint32_t* chunk_tot, uint32_t nchunks, uint32_t n_owners, uint32_t* utot) {
    __shared__ uint32_t tmp[4];
    const uint32_t o = blockIdx.x, tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nchunks; base += 256) {
        const uint32_t c = base + tid;
        const uint32_t v = c < nchunks ? chunk_tot[(size_t)c * n_owners + o] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256<uint32_t>(v, tmp, &tot) + carry;
        if (c < nchunks) chunk_tot[(size_t)c * n_owners + o] = ex;
        carry += tot;
    }
    if (tid == 0) utot[o] = carry;
}
// ucount[tile][owner] -> words of the owner emitted by earlier tiles; grid = chunks
__global__ __launch_bounds__(256) void k_sx_ucol_apply(uint32_t* ucount, uint32_t ntiles, uint32_t n_owners, const uint32_t* chunk_tot) {
    const uint32_t chunk = blockIdx.x, t0 = chunk * (uint32_t)UC_CHUNK;
    const uint32_t t1 = t0 + UC_CHUNK < ntiles ? t0 + UC_CHUNK : ntiles;
    for (uint32_t o = threadIdx.x; o < n_owners; o += 256) {
        uint32_t run = chunk_tot[(size_t)chunk * n_owners + o];
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t v = ucount[(size_t)t * n_owners + o];
            ucount[(size_t)t * n_owners + o] = run;
            run += v;
        }
    }
}

// ---- pair updates ----------------------------------------------------------------------------
// Entry e = (i, c) with rank P in its run pairs with entries e-P+1 .. e = (j <= i, c_j): K[i][j] += c * c_j.
// Update word = (cell - first cell of the owner band) << pb | product, cell = tri_index(i, j).
// One workgroup of 512 threads per entry tile (the entries one k_sx_seg_write tile produced).
// DIRECT: 64-bit atomicAdd per pair straight into K instead of update words.
constexpr int EM_THREADS = 512, EM_WAVES = EM_THREADS / 64;
constexpr int EM_PER = SG_TILE / EM_THREADS;              // entries per thread when a tile's short entries go in one pass
constexpr uint32_t EM_SLOTS = 12288;                      // update words of short entries binned per pass (u16 each in LDS)
static_assert((SG_TILE / 4) * SX_SHORT <= EM_SLOTS, "a quarter tile of short entries must always fit one pass");
template <bool DIRECT, bool SKIP>
__global__ __launch_bounds__(EM_THREADS) void k_sx_emit(const uint2* E, const uint32_t* Pk, const uint32_t* ebase,
                                                        const uint32_t* owner_r0, int own_shift, uint32_t n_owners,
                                                        const uint32_t* list_off, const uint32_t* tile_off, uint32_t* list,
                                                        uint32_t row0, uint32_t row1, uint32_t max_win, uint32_t maxprod,
                                                        uint32_t cmax, int pb, u64* K, uint32_t tpg, u64 slot_stride,
                                                        const uint32_t* Tk) {
    __shared__ uint2 s_ent[SG_TILE];
    __shared__ uint32_t s_P[SG_TILE];
    // partners before the entry itself, from its run's start: P - 1 unless skip_test_block (SKIP)
    __shared__ uint32_t s_T[SKIP ? SG_TILE : 1];
    __shared__ uint32_t s_r0[SX_MAX_OWNERS + 1];
    __shared__ uint32_t s_cur[SX_MAX_OWNERS];   // next free word of this tile's share of each owner's stream
    __shared__ uint32_t s_cnt[SX_MAX_OWNERS];
    __shared__ uint32_t s_seg[SX_MAX_OWNERS];   // first slot of the owner's segment; then (address in the stream) - slot
    // a pass's update words in owner order: slot -> the entry it belongs to; per entry: its first slot,
    // the cell of column 0 of its row inside the owner band, its owner. (After the short entries the
    // slot array holds the list of long entries.)
    __shared__ uint16_t slot_ent[EM_SLOTS];
    __shared__ uint16_t ent_at[SG_TILE], ent_o[SG_TILE];
    __shared__ uint32_t ent_cbase[SG_TILE];
    __shared__ uint32_t s_nlong, s_total;
    __shared__ uint32_t tmp[EM_WAVES];
    uint16_t* const s_long = slot_ent;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t e0 = ebase[tile], n = ebase[tile + 1] - e0;
    if (DIRECT) K += (u64)(tile / tpg) * slot_stride;  // one triangle per slot (variance mode) when slot_stride != 0
    if (!DIRECT) {
        for (uint32_t i = tid; i <= n_owners; i += EM_THREADS) s_r0[i] = owner_r0[i];
        for (uint32_t o = tid; o < n_owners; o += EM_THREADS) s_cur[o] = list_off[o] + tile_off[(size_t)tile * n_owners + o];
    }
    for (uint32_t e = tid; e < n; e += EM_THREADS) {
        s_ent[e] = E[(size_t)e0 + e];
        const uint32_t P = Pk[(size_t)e0 + e];
        s_P[e] = P;
        if (SKIP) s_T[e] = Tk[(size_t)e0 + e];
    }
    __syncthreads();
    // partner `ge` of an entry: the tile's own entries sit in LDS, a run that started before the tile
    // continues in global memory
#define SX_PARTNER(ge) ((ge) >= e0 ? s_ent[(ge) - e0] : E[(ge)])
    // partner number b (0 .. T) of tile-local entry el: the run's first T entries, then the entry itself
#define SX_NPART(el) (SKIP ? s_T[el] + 1u : s_P[el])
#define SX_PARTNER_OF(el, b) SX_PARTNER((!SKIP || (b) < s_T[el]) ? e0 + (el) - s_P[el] + 1u + (b) : e0 + (el))
    // an entry is SHORT (binned below) with up to SX_SHORT partners and one word per pair, else LONG (a wave each, further down)
#define SX_IS_SHORT(el) (SX_NPART(el) <= SX_SHORT && (DIRECT || sx_words_per_pair(s_ent[el].y, cmax, max_win, maxprod) == 1u))
#define SX_IN_BAND(el) (s_ent[el].x >= row0 && s_ent[el].x < row1)
    if (DIRECT) {
        for (uint32_t e = tid; e < n; e += EM_THREADS) {
            if (SX_IN_BAND(e) && SX_IS_SHORT(e)) {
                const uint2 a = s_ent[e];
                const uint32_t P = SX_NPART(e);
                u64* row = K + tri_index((u64)a.x, 0);
                for (uint32_t b = 0; b < P; ++b) {
                    const uint2 pq = SX_PARTNER_OF(e, b);
                    atomicAdd(&row[pq.x], (u64)a.y * pq.y);
                }
            }
        }
    } else {
        // ---- short entries. All of the tile's in one pass when their words fit the slot array (the usual
        // case: ~4 partners per entry), else a quarter of the tile per pass (which always fits).
        uint32_t words = 0;
        for (uint32_t e = tid; e < n; e += EM_THREADS)
            if (SX_IN_BAND(e) && SX_IS_SHORT(e)) words += SX_NPART(e);
        uint32_t all;
        (void)block_excl_scan<uint32_t, EM_WAVES>(words, tmp, &all);
        const uint32_t span = all <= EM_SLOTS ? (uint32_t)SG_TILE : (uint32_t)(SG_TILE / 4);  // entries per pass (uniform)
        for (uint32_t ea = 0; ea < n; ea += span) {
            for (uint32_t o = tid; o < n_owners; o += EM_THREADS) s_cnt[o] = 0u;
            __syncthreads();
            uint32_t my_pos[EM_PER];
#pragma unroll
            for (int q = 0; q < EM_PER; ++q) {
                const uint32_t e = ea + (uint32_t)q * EM_THREADS + tid;
                my_pos[q] = 0xffffffffu;
                if (e < n && e < ea + span && SX_IN_BAND(e) && SX_IS_SHORT(e)) {
                    const uint32_t o = sx_owner_of(s_ent[e].x, own_shift);
                    ent_o[e] = (uint16_t)o;
                    my_pos[q] = atomicAdd(&s_cnt[o], SX_NPART(e));
                }
            }
            __syncthreads();
            {   // exclusive scan of the owner counts (n_owners <= 512 = one per thread)
                const uint32_t v = tid < n_owners ? s_cnt[tid] : 0u;
                uint32_t tot;
                const uint32_t ex = block_excl_scan<uint32_t, EM_WAVES>(v, tmp, &tot);
                if (tid < n_owners) s_seg[tid] = ex;
                if (tid == 0) s_total = tot;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < EM_PER; ++q) {
                if (my_pos[q] != 0xffffffffu) {
                    const uint32_t e = ea + (uint32_t)q * EM_THREADS + tid;
                    const uint32_t o = ent_o[e];
                    const uint32_t at = s_seg[o] + my_pos[q], P = SX_NPART(e);
                    ent_at[e] = (uint16_t)at;
                    ent_cbase[e] = (uint32_t)(tri_index((u64)s_ent[e].x, 0) - tri_index((u64)s_r0[o], 0));
                    for (uint32_t b = 0; b < P; ++b) slot_ent[at + b] = (uint16_t)e;
                }
            }
            __syncthreads();
            if (tid < n_owners) s_seg[tid] = s_cur[tid] - s_seg[tid];  // slot -> address in the owner's stream
            __syncthreads();
            // one update word per thread and trip, whatever the entries' partner counts: slot i is partner
            // i - ent_at of its entry; neighbouring slots go to neighbouring addresses of one stream
            const uint32_t total = s_total;
            for (uint32_t i = tid; i < total; i += EM_THREADS) {
                const uint32_t el = slot_ent[i];
                const uint32_t b = i - ent_at[el];
                const uint32_t ge = (!SKIP || b < s_T[el]) ? e0 + el - s_P[el] + 1u + b : e0 + el;
                const uint2 pq = SX_PARTNER(ge);
                list[s_seg[ent_o[el]] + i] = ((ent_cbase[el] + pq.x) << pb) | (s_ent[el].y * pq.y);
            }
            __syncthreads();
            if (tid < n_owners) s_cur[tid] += s_cnt[tid];
            // (the barriers of the next pass, or the one below, order this against its readers)
        }
    }
    __syncthreads();
    // entries with many partners (or a product that needs several words): listed now that the slot array is free
    if (tid == 0) s_nlong = 0u;
    __syncthreads();
    for (uint32_t e = tid; e < n; e += EM_THREADS)
        if (SX_IN_BAND(e) && !SX_IS_SHORT(e)) s_long[atomicAdd(&s_nlong, 1u)] = (uint16_t)e;
    __syncthreads();
    // ---- long entries: the lanes of a wave take the partners, the words of an entry are contiguous
    const uint32_t nlong = s_nlong;
    for (uint32_t q = wave; q < nlong; q += EM_WAVES) {
        const uint32_t e = s_long[q];
        const uint2 a = s_ent[e];
        const uint32_t P = SX_NPART(e);
        if (DIRECT) {
            u64* row = K + tri_index((u64)a.x, 0);
            for (uint32_t b = lane; b < P; b += 64) {
                const uint2 pq = SX_PARTNER_OF(e, b);
                atomicAdd(&row[pq.x], (u64)a.y * pq.y);
            }
        } else {
            const uint32_t S = sx_words_per_pair(a.y, cmax, max_win, maxprod);
            const uint32_t o = sx_owner_of(a.x, own_shift);
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_cur[o], P * S);
            at = __shfl(at, 0);
            const uint32_t cbase = (uint32_t)(tri_index((u64)a.x, 0) - tri_index((u64)s_r0[o], 0));
            if (S == 1u && !SKIP) {
                // the common case, kept lean (long runs make this loop most of the kernel): partner b is
                // entry first + b — in LDS from `split` on, in global memory before — one word each
                uint32_t* dst = list + at;
                const uint32_t first = e0 + e - P + 1u;                 // global index of partner 0
                const uint32_t split = first < e0 ? e0 - first : 0u;    // partners before the tile
                for (uint32_t b = lane; b < P; b += 64u) {
                    const uint2 pq = b >= split ? s_ent[first - e0 + b] : E[first + b];
                    dst[b] = ((cbase + pq.x) << pb) | (a.y * pq.y);
                }
                continue;
            }
            const uint32_t trips = (P + 63u) / 64u;
            for (uint32_t tr = 0; tr < trips; ++tr) {
                const uint32_t b = tr * 64u + lane;
                if (b < P) {
                    const uint2 pq = SX_PARTNER_OF(e, b);
                    u64 prod = (u64)a.y * pq.y;
                    for (uint32_t w = 0; w < S; ++w) {
                        const uint32_t part = prod < maxprod ? (uint32_t)prod : maxprod;
                        list[at + b * S + w] = ((cbase + pq.x) << pb) | part;
                        prod -= part;
                    }
                }
            }
        }
    }
#undef SX_IN_BAND
#undef SX_IS_SHORT
#undef SX_PARTNER_OF
#undef SX_NPART
#undef SX_PARTNER
}

// Streams differ a lot in length (a band that holds a long or low-complexity sequence receives many
// times the average), so a stream is cut into parts of about `target` words, one workgroup each:
// part_base[o] = parts of the bands before o (single workgroup of 512 threads, n_owners <= 512).
__global__ __launch_bounds__(512) void k_sx_parts(const uint32_t* list_off, uint32_t n_owners, uint32_t target, uint32_t* part_base) {
    __shared__ uint32_t tmp[8];
    const uint32_t tid = threadIdx.x;
    const uint32_t len = tid < n_owners ? list_off[tid + 1] - list_off[tid] : 0u;
    const uint32_t parts = (len + target - 1u) / target;
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t, 8>(parts, tmp, &tot);
    if (tid < n_owners) part_base[tid] = ex;
    if (tid == 0) part_base[n_owners] = tot;
}

// One workgroup of 1024 threads per (part of an owner band's stream, round): sums its words into `cap`
// u32 cells in LDS and adds the non-zero ones into K — a plain read-modify-write when the band has a
// single part (a cell then has one writer and launches are ordered on the stream), 64-bit atomics
// otherwise. A band larger than `cap` cells takes several rounds over its stream.
// dynamic LDS: cap * 4 bytes. grid = (upper bound of the number of parts, rounds)
constexpr uint32_t CS_THREADS = 1024;
// By-slot form (variance mode, slot_stride != 0): grid = (owner bands, rounds, slots); the words a slot
// put into a band's stream are contiguous (tiles are slot-major), bounded by the tile offsets of the
// slot's first tile, and the sums are STORED as u32 into the slot's own triangle (uint32_t*)K + slot *
// slot_stride — every cell of it, so it needs no zero fill.
__global__ __launch_bounds__(1024) void k_sx_consume(const uint32_t* list, const uint32_t* list_off, const uint32_t* owner_r0,
                                                     const uint32_t* part_base, uint32_t n_owners, uint32_t target, uint32_t cap,
                                                     int pb, u64* K, const uint32_t* tile_off, uint32_t tpg, u64 slot_stride) {
    FSK_DYN_SHARED(uint32_t, cells);
    __shared__ uint32_t s_base[SX_MAX_OWNERS + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t r = blockIdx.y;
    uint32_t o, a, b, nparts = 1u;
    uint32_t* K32 = nullptr;  // by-slot form: the slot's triangle is a u32 array that this launch WRITES (no zero fill needed)
    if (slot_stride != 0) {
        o = blockIdx.x;
        const uint32_t slot = blockIdx.z;
        a = list_off[o] + tile_off[(size_t)slot * tpg * n_owners + o];
        b = slot + 1u < gridDim.z ? list_off[o] + tile_off[(size_t)(slot + 1u) * tpg * n_owners + o] : list_off[o + 1];
        K32 = reinterpret_cast<uint32_t*>(K) + (u64)slot * slot_stride;
    } else {
        for (uint32_t i = tid; i <= n_owners; i += CS_THREADS) s_base[i] = part_base[i];
        __syncthreads();
        const uint32_t slot = blockIdx.x;
        if (slot >= s_base[n_owners]) return;
        uint32_t hi = n_owners;  // s_base[o] <= slot < s_base[hi]; bands without words share their successor's base
        o = 0;
        while (hi - o > 1u) {
            const uint32_t mid = (o + hi) >> 1;
            if (s_base[mid] <= slot) o = mid; else hi = mid;
        }
        const uint32_t part = slot - s_base[o];
        nparts = s_base[o + 1] - s_base[o];
        a = list_off[o] + part * target;
        const uint32_t end = list_off[o + 1];
        b = end - a < target ? end : a + target;
    }
    if (a == b && !K32) return;
    const u64 c0 = tri_index((u64)owner_r0[o], 0), c1 = tri_index((u64)owner_r0[o + 1], 0);
    const uint32_t ncell = (uint32_t)(c1 - c0);
    const uint32_t lo = r * cap;
    if (lo >= ncell) return;
    const uint32_t span = ncell - lo < cap ? ncell - lo : cap;
    for (uint32_t i = tid; i < span; i += CS_THREADS) cells[i] = 0u;
    __syncthreads();
    const uint32_t mask = (1u << pb) - 1u;
    uint32_t i = a + tid;
    for (; i + 3u * CS_THREADS < b; i += 4u * CS_THREADS) {  // four independent loads in flight per thread
        const uint32_t w0 = list[i], w1 = list[i + CS_THREADS], w2 = list[i + 2u * CS_THREADS], w3 = list[i + 3u * CS_THREADS];
        const uint32_t x0 = (w0 >> pb) - lo, x1 = (w1 >> pb) - lo, x2 = (w2 >> pb) - lo, x3 = (w3 >> pb) - lo;
        if (x0 < span) atomicAdd(&cells[x0], w0 & mask);
        if (x1 < span) atomicAdd(&cells[x1], w1 & mask);
        if (x2 < span) atomicAdd(&cells[x2], w2 & mask);
        if (x3 < span) atomicAdd(&cells[x3], w3 & mask);
    }
    for (; i < b; i += CS_THREADS) {
        const uint32_t w = list[i];
        const uint32_t x = (w >> pb) - lo;
        if (x < span) atomicAdd(&cells[x], w & mask);
    }
    __syncthreads();
    if (K32) {
        for (uint32_t c = tid; c < span; c += CS_THREADS) K32[c0 + lo + c] = cells[c];
        return;
    }
    u64* dst = K + c0 + lo;
    if (nparts == 1u) {
        for (uint32_t c = tid; c < span; c += CS_THREADS) {
            const uint32_t v = cells[c];
            if (v) dst[c] += (u64)v;
        }
    } else {
        for (uint32_t c = tid; c < span; c += CS_THREADS) {
            const uint32_t v = cells[c];
            if (v) atomicAdd(&dst[c], (u64)v);
        }
    }
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
                      // 16: every value of the block is zero (the block changes no running sum)
    u64 Q;            // sum of R(p) / u over the block
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
// R(p)/u of one value for scale = 1/u, added to q; limit bounds a single value so that q cannot wrap
__device__ __forceinline__ void seq_classify(double p, double scale, double limit, u64& q, uint32_t& flags) {
    if (!(p >= 0.0)) { flags |= 1u; return; }  // (-0.0 passes and adds nothing)
    const double x = p * scale;                // exact: a power of two
    if (!(x < limit)) { flags |= 2u; return; }
    if (x - floor(x) == 0.5) flags |= 4u;
    q += (u64)rint(x);
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
        blk[b].e = e == INT32_MIN ? 0 : e;
        blk[b].flags = (all & 0xffu) | ((all >> 8) ? 0u : 16u);
        blk[b].Q = s_q[0] + s_q[1] + s_q[2] + s_q[3];
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
        seq_classify(v[k], scale, 9007199254740992.0 /* 2^53 */, q, fl);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        q += __shfl_xor(q, d);
        fl |= __shfl_xor(fl, d);
        nonzero |= __shfl_xor(nonzero, d);
    }
    if (!nonzero) return true;  // zeros change no running sum, whatever it is (s starts at +0 and +0 + -0 = +0)
    if (!usable) return false;
    const u64 tot = (u64)(s * scale) + q;
    if (fl != 0u || tot >= ((u64)1 << 53)) return false;
    s = (double)tot * seq_pow2(e - 52);
    return true;
}
// The values [lo, hi) added to s: groups of 1024 (16 per lane, value lo + 64 k + lane in register k,
// the next group's loads in flight meanwhile), then the 16 sub-groups of 64, then plain sequential
// additions — all from registers — for a group that crosses a binade, holds a tie or a negative value.
// s and every decision are wave-uniform.
__device__ __forceinline__ double seq_range(const double* p, u64 lo, u64 hi, double s) {
    const uint32_t lane = threadIdx.x & 63u;
    double cur[16], nxt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u64 i = lo + (u64)k * 64 + lane;
        cur[k] = i < hi ? p[i] : 0.0;
    }
    for (u64 a0 = lo; a0 < hi; a0 += 1024) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const u64 i = a0 + 1024 + (u64)k * 64 + lane;
            nxt[k] = i < hi ? p[i] : 0.0;
        }
        if (!seq_try_regs<16>(cur, s)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double one[1] = {cur[k]};
                if (!seq_try_regs<1>(one, s)) {
                    const u64 c0 = a0 + (u64)k * 64;
                    const uint32_t cn = c0 >= hi ? 0u : (hi - c0 < 64 ? (uint32_t)(hi - c0) : 64u);
                    for (uint32_t j = 0; j < cn; ++j) s = __dadd_rn(s, wave_bcast_f64(cur[k], j));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
    return s;
}

// one wave: out[0] = the sequential sum of p[0..n); zeroes bsum for the slot's next use
// (blockIdx.x = one of several independent sums laid out `stride` values / `nblocks` blocks apart)
__global__ __launch_bounds__(64) void k_seq_chain(const double* p, u64 n, const SeqBlk* blk, uint32_t nblocks, double* bsum, double* out,
                                                  u64 stride) {
    const uint32_t lane = threadIdx.x;
    p += (u64)blockIdx.x * stride;
    blk += (size_t)blockIdx.x * nblocks;
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
            if (as_int) {
                s = (double)S * seq_pow2(e - 52);
                as_int = false;
            }
            const u64 lo = (u64)(c0 + j) * SQ_BLOCK, hi = lo + SQ_BLOCK < n ? lo + SQ_BLOCK : n;
            s = seq_range(p, lo, hi, s);
        }
    }
    if (as_int) s = (double)S * seq_pow2(e - 52);
    if (lane == 0) out[0] = s;
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
