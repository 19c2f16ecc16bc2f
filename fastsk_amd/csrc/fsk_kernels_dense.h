// fsk_kernels_dense.h — DENSE dataflow: per-sequence LDS counting sort into 4-bit count panels (k_dense_count),
// key compaction (k_dense_keylut), the exact update count U (k_dense_distinct) and the output-stationary
// 128x128 tile kernels (fsk_tile_kernel_dma.inc). Included by fsk_engine_dense.hip only.
#pragma once
#include "fsk_common.h"

namespace fsk {

// =============================================================================================
// DENSE PATH
// =============================================================================================
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
    if (K > 0 && kmode == 0) {
        // The common case (every BASELINE config; with key compaction both of its passes — MARK sets the key's bit, LUT
        // counts the key's rank — unless a window-key cache is in use): WPT windows per trip. Window j+4 of a lane lies
        // 4 rows = 256 bytes further in every symbol column, an immediate offset of the same
        // address registers, so the loop bookkeeping is paid once per WPT windows. Rows past a
        // sequence's end are zero padding inside symT (the trip condition keeps them in range);
        // their updates are predicated off.
        constexpr int WPT = 4;
        // (byte offsets into symT, read with explicit LDS loads: as an array of `const uint8_t*` the columns are
        // generic pointers and every symbol read becomes a flat_load_ubyte)
        uint32_t p[K > 0 ? K : 1];
#pragma unroll
        for (int c = 0; c < K; ++c) p[c] = (j0 - cb + pr[c]) * PANEL + r;
        uint32_t j = j0;
        for (; j + 4u * (WPT - 1) < hi; j += 4u * WPT) {
            uint32_t kk[WPT];
#pragma unroll
            for (int u = 0; u < WPT; ++u) kk[u] = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
#pragma unroll
                for (int u = 0; u < WPT; ++u) kk[u] = mad24(kk[u], sigma, FSK_LDS_LOAD_U8(symT + p[c] + u * 4 * PANEL));
                p[c] += 4 * WPT * PANEL;
            }
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                if (MARK) {
                    if (j + 4u * u < nwin) atomicOr(&hist[kk[u] >> 5], 1u << (kk[u] & 31u));  // hist doubles as the key bitmap
                    continue;
                }
                // (the table sits in LDS; a padding row's key is 0, inside it)
                const uint32_t key = (LUT ? (uint32_t)FSK_LDS_LOAD_U16(lut + kk[u]) : kk[u]) - key_lo;  // wraps for keys below the sweep: rejected by the compare
                if (j + 4u * u < nwin && key < key_n) atomicAdd(&hist[key * 32u + (r >> 1)], 1u << half);
            }
        }
        for (; j < hi; j += 4u) {  // the last few windows of the chunk
            uint32_t k0 = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                k0 = mad24(k0, sigma, FSK_LDS_LOAD_U8(symT + p[c]));
                p[c] += 4 * PANEL;
            }
            if (MARK) {
                if (j < nwin) atomicOr(&hist[k0 >> 5], 1u << (k0 & 31u));
                continue;
            }
            if (LUT) k0 = (uint32_t)FSK_LDS_LOAD_U16(lut + k0);
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

// Key compaction without the marking pass over every window (k_dense_count<true, .>): when the alphabet's rare symbols
// (an 'n' in DNA) occur at a few places only, the keys a combo can meet are (i) every key made of common symbols alone —
// the same set for all combos, taken as present — and (ii) the keys of the windows that hold a rare symbol at a KEPT
// position. k_dense_rare_scan lists the places once per set of sequences; per accumulate k_dense_keybits_init writes (i)
// into every combo's bitmap and k_dense_mark_rare adds (ii): g windows per place and combo instead of all of them. A key
// of (i) that no window has costs an empty panel row, never a wrong count (the tables map it like any other).
// grid = ceil(N / 4), block = 256: one wave per sequence.
__global__ __launch_bounds__(256) void k_dense_rare_scan(SeqView S, uint32_t rare_mask, u64* list, uint32_t cap, uint32_t* count) {
    const uint32_t seq = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seq >= S.n_seq) return;
    const uint32_t len = S.len[seq], wbase = S.wstart[seq];
    for (uint32_t p = lane; p < len; p += 64u) {
        const uint32_t sym = fetch_sym(S.words, wbase, p, S.bits);
        if ((rare_mask >> sym) & 1u) {
            const uint32_t at = atomicAdd(count, 1u);
            if (at < cap) list[at] = ((u64)seq << 32) | p;
        }
    }
}

// grid = ceil(Vw / 256): word i of `common` = the keys 32 i .. 32 i + 31 whose digits are all common symbols (the same for
// every combo: once per set of sequences)
__global__ __launch_bounds__(256) void k_dense_common_keys(uint32_t* common, uint32_t V, int k, uint32_t sigma, uint32_t rare_mask) {
    const uint32_t Vw = (V + 31u) >> 5, i = blockIdx.x * 256u + threadIdx.x;
    if (i >= Vw) return;
    uint32_t word = 0;
    for (uint32_t b = 0; b < 32u; ++b) {
        uint32_t x = 32u * i + b;
        if (x >= V) break;
        bool ok = true;
        for (int c = 0; c < k; ++c) { ok = ok && !((rare_mask >> (x % sigma)) & 1u); x /= sigma; }
        if (ok) word |= 1u << b;
    }
    common[i] = word;
}
// grid = (ceil(Vw / 256), n_slots): every combo's bitmap starts as the common keys
__global__ __launch_bounds__(256) void k_dense_keybits_init(uint32_t* keybits, const uint32_t* common, uint32_t V) {
    const uint32_t Vw = (V + 31u) >> 5, i = blockIdx.x * 256u + threadIdx.x;
    if (i < Vw) keybits[(size_t)blockIdx.y * Vw + i] = common[i];
}

// grid = (blocks, n_slots), block = 256: thread = (place, window offset d): the window that starts d symbols before the place
// holds the rare symbol at its position d — when the combo keeps that position, the window's key gets its bit
__global__ __launch_bounds__(256) void k_dense_mark_rare(SeqView S, const u64* list, const uint32_t* count, uint32_t cap, int g, int k,
                                                         uint32_t sigma, const uint8_t* combo_pos, uint32_t V, uint32_t* keybits) {
    const uint32_t slot = blockIdx.y, Vw = (V + 31u) >> 5;
    const uint32_t n = *count < cap ? *count : cap;
    const u64 work = (u64)n * (u64)g;
    const uint8_t* pos = combo_pos + (size_t)slot * k;
    for (u64 t = (u64)blockIdx.x * 256u + threadIdx.x; t < work; t += (u64)gridDim.x * 256u) {
        const uint32_t ent = (uint32_t)(t / (u64)g), d = (uint32_t)(t % (u64)g);
        const u64 rec = list[ent];
        const uint32_t seq = (uint32_t)(rec >> 32), p = (uint32_t)rec;
        const uint32_t len = S.len[seq];
        if (p < d || p - d + (uint32_t)g > len) continue;  // no such window
        bool kept = false;
        for (int c = 0; c < k; ++c) kept = kept || (uint32_t)pos[c] == d;
        if (!kept) continue;
        const uint32_t j = p - d, wbase = S.wstart[seq];
        uint32_t key = 0;
        for (int c = 0; c < k; ++c) key = key * sigma + fetch_sym(S.words, wbase, j + (uint32_t)pos[c], S.bits);
        atomicOr(&keybits[(size_t)slot * Vw + (key >> 5)], 1u << (key & 31u));
    }
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

// The remainder terms of the tile kernels, counted (profiling aid: fsk_stats.dense_macs prices what the kernels really
// multiply). A count above 15 lives in the hi plane; a dword row of a (panel, combo) block whose hi plane is not all zero is
// FLAGGED in rowmask, and for every flagged row of a tile the tile kernel adds the exact remainder 16 (hi_i lo_j + lo_i hi_j)
// + 256 hi_i hi_j: three more dot8 per cell (the key-compacted kernels: only the terms whose side is flagged — A, B, A and B).
// out += the number of such extra row products over the launch's tiles x combos. grid = (ceil(tiles / 256), combos).
__global__ __launch_bounds__(256) void k_dense_remainder_rows(const uint32_t* rowmask, const uint32_t* tiletab, uint32_t n_tiles,
                                                              uint32_t n_slots, uint32_t nst, int side_aware, u64* out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, sl = blockIdx.y;
    u64 rows = 0;
    if (t < n_tiles) {
        const uint32_t tt = tiletab[t], ti = tt >> 16, tj = tt & 0xffffu;
        const size_t np = (size_t)n_slots * nst;
        for (uint32_t st = 0; st < nst; ++st) {
            const size_t a = ((size_t)(ti * 2u) * n_slots + sl) * nst + st, b = ((size_t)(tj * 2u) * n_slots + sl) * nst + st;
            const uint32_t mA = rowmask[a] | rowmask[a + np], mB = rowmask[b] | rowmask[b + np];
            rows += side_aware ? (u64)(__popc(mA) + __popc(mB) + __popc(mA & mB)) : (u64)3 * (u64)__popc(mA | mB);
        }
    }
    rows = fsk_hw::wave_sum_u64(rows);
    if ((threadIdx.x & 63u) == 0 && rows) atomicAdd(out, rows);
}

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

}  // namespace fsk
