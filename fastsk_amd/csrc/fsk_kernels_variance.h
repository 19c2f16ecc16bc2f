// fsk_kernels_variance.h — approx / variance mode: Welford updates and the exact left-to-right fp64 sum of
// get_variance (fastsk_kernel.cpp:108-143) on the device. Included by fsk_engine_variance.hip only.
#pragma once
#include "fsk_common.h"

namespace fsk {

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
    const double r_iter = __ddiv_rn(1.0, iter);
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
        const u64 i = base + (u64)q * 256 + threadIdx.x;
        if (i < pairs) {
            const double x = (double)Ks[i];
            const double old = K_hat_in[i];
            const double delta = __dsub_rn(x, old);
            const double kh = __dadd_rn(old, div_by_shared(delta, iter, r_iter));
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
constexpr int WF_SLOTS = 8;
template <typename SrcT>
__global__ __launch_bounds__(256) void k_welford_batch(const SrcT* Ks, int nslots, const double* K_hat_in, double* K_hat_out, double* prod,
                                                       u64 prod_stride, u64 pairs, u64 train_pairs, double first_iter, double* bsum,
                                                       uint32_t nblk, int write_prod) {
    __shared__ double part[WF_SLOTS][4];
    const u64 base = (u64)blockIdx.x * (256 * WF_ITEMS);
    double pr[WF_SLOTS];
#pragma unroll
    for (int s = 0; s < WF_SLOTS; ++s) pr[s] = 0.0;
    double r_iter[WF_SLOTS];  // 1 / iteration number: see div_by_shared
#pragma unroll
    for (int s = 0; s < WF_SLOTS; ++s) r_iter[s] = __ddiv_rn(1.0, first_iter + (double)s);
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
                    kh = __dadd_rn(kh, div_by_shared(delta, first_iter + (double)s, r_iter[s]));
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

// k_welford_batch with 16-byte accesses: a thread takes V = 16 / sizeof(SrcT) CONSECUTIVE cells (4 u32 counts, 2 u64
// counts) — one 16-byte load per slot, the state and the products in 16-byte pieces too — instead of single cells 256
// apart: a quarter of the memory instructions for the same bytes, four times the bytes in flight per instruction (the
// scalar form sat at 2.5 TB/s with 90 % of its wave cycles waiting). The arithmetic per cell and iteration is the same
// sequence of IEEE operations. Needs every array 16-byte aligned: the slot triangles `slot_stride` cells apart, K_hat
// and the products with strides that are multiples of 4 (the host pads the strides: cells in [pairs, padded) hold
// nothing anybody reads). riter[s] = 1 / (first_iter + s), computed once on the host side of the launch (same IEEE
// division) and passed by value.
struct WfRecip { double r[WF_SLOTS]; };
template <typename SrcT> struct WfVec;
template <> struct WfVec<uint32_t> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const uint32_t* p, double (&x)[4]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        x[0] = (double)v.x; x[1] = (double)v.y; x[2] = (double)v.z; x[3] = (double)v.w;
    }
};
template <> struct WfVec<uint16_t> {  // (u16 slot triangles, 8 bytes per load: half the bytes of the u32 form for the same cells)
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const uint16_t* p, double (&x)[4]) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        x[0] = (double)(v.x & 0xffffu); x[1] = (double)(v.x >> 16); x[2] = (double)(v.y & 0xffffu); x[3] = (double)(v.y >> 16);
    }
};
template <> struct WfVec<u64> {
    static constexpr int V = 2;
    static __device__ __forceinline__ void load(const u64* p, double (&x)[2]) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p);
        x[0] = (double)v.x; x[1] = (double)v.y;
    }
};
template <typename SrcT>
__global__ __launch_bounds__(256) void k_welford_batch_v(const SrcT* Ks, u64 slot_stride, int nslots, const double* K_hat_in, double* K_hat_out,
                                                         double* prod, u64 prod_stride, u64 pairs, u64 train_pairs, double first_iter,
                                                         WfRecip riter, double* bsum, uint32_t nblk, int write_prod) {
    constexpr int V = WfVec<SrcT>::V, REPS = WF_ITEMS / V;
    __shared__ double part[WF_SLOTS][4];
    const u64 base = (u64)blockIdx.x * (256 * WF_ITEMS);
    double pr[WF_SLOTS];
#pragma unroll
    for (int s = 0; s < WF_SLOTS; ++s) pr[s] = 0.0;
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
        const u64 i0 = base + ((u64)rep * 256 + threadIdx.x) * V;
        if (i0 < pairs) {
            double xs[WF_SLOTS][V];
#pragma unroll
            for (int s = 0; s < WF_SLOTS; ++s) {  // (all loads before the dependent chain)
                if (s < nslots) WfVec<SrcT>::load(Ks + (u64)s * slot_stride + i0, xs[s]);
                else {
#pragma unroll
                    for (int v = 0; v < V; ++v) xs[s][v] = 0.0;
                }
            }
            double kh[V];
#pragma unroll
            for (int v = 0; v < V; v += 2) {
                const double2 k2 = *reinterpret_cast<const double2*>(K_hat_in + i0 + v);
                kh[v] = k2.x; kh[v + 1] = k2.y;
            }
#pragma unroll
            for (int s = 0; s < WF_SLOTS; ++s) {
                if (s < nslots) {
                    double pv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const double x = xs[s][v];
                        const double delta = __dsub_rn(x, kh[v]);
                        kh[v] = __dadd_rn(kh[v], div_by_shared(delta, first_iter + (double)s, riter.r[s]));
                        pv[v] = __dmul_rn(delta, __dsub_rn(x, kh[v]));
                    }
                    if (write_prod && i0 < train_pairs) {
                        double* pd = prod + (u64)s * prod_stride + i0;
                        if (i0 + V <= train_pairs) {
#pragma unroll
                            for (int v = 0; v < V; v += 2) {
                                double2 o; o.x = pv[v]; o.y = pv[v + 1];
                                *reinterpret_cast<double2*>(pd + v) = o;
                                pr[s] += pv[v];
                                pr[s] += pv[v + 1];
                            }
                        } else {
#pragma unroll
                            for (int v = 0; v < V; ++v)
                                if (i0 + v < train_pairs) { pd[v] = pv[v]; pr[s] += pv[v]; }
                        }
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < V; v += 2) {
                double2 o; o.x = kh[v]; o.y = kh[v + 1];
                *reinterpret_cast<double2*>(K_hat_out + i0 + v) = o;
            }
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
    // (the block's integer total and flags do not depend on the order of its values: two consecutive values per thread and
    // trip, one 16-byte load when the sum's values start on a 16-byte boundary — every caller's blocks do, except a
    // stand-alone sum over a caller's odd pointer)
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
#pragma unroll 4
        for (int j = 0; j < SQ_BLOCK / 512; ++j) {
            const u64 i = lo + ((u64)j * 256 + tid) * 2;
            if (i + 1 < n) {
                const double2 v = *reinterpret_cast<const double2*>(p + i);
                if (v.x != 0.0 || v.y != 0.0) nonzero = 1u;  // (NaN counts as non-zero)
                seq_classify(v.x, scale, 1125899906842624.0 /* 2^50 */, q, fl);
                seq_classify(v.y, scale, 1125899906842624.0, q, fl);
            } else if (i < n) {
                const double v = p[i];
                if (v != 0.0) nonzero = 1u;
                seq_classify(v, scale, 1125899906842624.0, q, fl);
            }
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < SQ_BLOCK / 256; ++j) {
            const u64 i = lo + (u64)j * 256 + tid;
            if (i < n) {
                const double v = p[i];
                if (v != 0.0) nonzero = 1u;  // (NaN counts as non-zero)
                seq_classify(v, scale, 1125899906842624.0 /* 2^50 */, q, fl);
            }
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
__device__ __forceinline__ uint32_t wave_bcast_u32(uint32_t x, uint32_t src) { return fsk_hw::readlane(x, src); }
__device__ __forceinline__ u64 wave_bcast_u64(u64 x, uint32_t src) {
    return ((u64)wave_bcast_u32((uint32_t)(x >> 32), src) << 32) | wave_bcast_u32((uint32_t)x, src);
}
__device__ __forceinline__ double wave_bcast_f64(double x, uint32_t src) {
    DblBits b; b.d = x;
    b.u = wave_bcast_u64(b.u, src);
    return b.d;
}

// sum of x over the 64 lanes, in every lane. On the critical path of k_seq_chain (one wave, nothing to overlap
// with): DPP row shifts / broadcasts (fsk_hw::wave_sum_u64), not six dependent LDS permutes per 32-bit half.
__device__ __forceinline__ u64 wave_sum_u64(u64 x) { return fsk_hw::wave_sum_u64(x); }

// The running sum of a chain: either the double s, or — while it stays inside one binade — the integer
// S = s / 2^(e-52) in [2^52, 2^53): values that go through as integer totals are then one 64-bit add and a compare, with
// no conversion between the two forms until a value has to be added the plain way.
struct SeqAcc {
    double s;
    u64 S;
    int e;
    bool as_int;
};
__device__ __forceinline__ void seq_acc_to_int(SeqAcc& a) {
    if (!a.as_int && a.s > 0.0 && seq_exponent(a.s) != INT32_MIN) {
        a.e = seq_exponent(a.s);
        a.S = (u64)(a.s * seq_pow2(52 - a.e));
        a.as_int = true;
    }
}
__device__ __forceinline__ void seq_acc_to_double(SeqAcc& a) {
    if (a.as_int) {
        a.s = (double)a.S * seq_pow2(a.e - 52);
        a.as_int = false;
    }
}
// registers cur[OFF .. OFF + LEN) of every lane (values 64 k + lane of a group; 0.0 where the range ended) added to the
// sum as one integer total, if the conditions hold
template <int OFF, int LEN>
__device__ __forceinline__ bool seq_try_span(const double (&cur)[16], SeqAcc& a) {
    uint32_t nonzero = 0;
#pragma unroll
    for (int k = OFF; k < OFF + LEN; ++k)
        if (cur[k] != 0.0) nonzero = 1u;  // (NaN counts as non-zero)
    if (__ballot(nonzero != 0u) == 0ull) return true;  // zeros change no running sum, whatever it is (s starts at +0 and +0 + -0 = +0)
    seq_acc_to_int(a);
    if (!a.as_int) return false;
    const double scale = seq_pow2(52 - a.e);
    u64 q = 0;
    uint32_t fl = 0;
#pragma unroll
    for (int k = OFF; k < OFF + LEN; ++k) seq_classify(cur[k], scale, 4503599627370496.0 /* 2^52 */, q, fl);
    if (__ballot(fl != 0u) != 0ull) return false;
    const u64 tot = a.S + wave_sum_u64(q);
    if (tot >= ((u64)1 << 53)) return false;
    a.S = tot;
    return true;
}
// ... and the same span added whatever it holds: as a whole, else by halves, down to one register (64 consecutive
// values), which is then added value by value — the place where the running sum crosses a binade, or a tie, or a
// negative value. `here`: values of the group that exist.
template <int OFF, int LEN>
__device__ __forceinline__ void seq_span(const double (&cur)[16], uint32_t here, SeqAcc& a) {
    if (seq_try_span<OFF, LEN>(cur, a)) return;
    if constexpr (LEN == 1) {
        seq_acc_to_double(a);
        const uint32_t q0 = (uint32_t)OFF * 64u;
        const uint32_t cn = q0 >= here ? 0u : (here - q0 < 64u ? here - q0 : 64u);
        double s = a.s;
        for (uint32_t j = 0; j < cn; ++j) s = __dadd_rn(s, wave_bcast_f64(cur[OFF], j));
        a.s = s;
    } else {
        seq_span<OFF, LEN / 2>(cur, here, a);
        seq_span<OFF + LEN / 2, LEN / 2>(cur, here, a);
    }
}
// The values [lo, hi) added to the sum by one wave: groups of 1024 straight from global memory into registers (16 per
// lane, value 64 k + lane in register k), the loads of the next group in flight while this one is summed (no LDS
// staging: the chain is one wave per sum and every microsecond of it is latency the next batch of iterations waits
// for). Every decision is wave-uniform.
__device__ __forceinline__ void seq_range(const double* p, u64 lo, u64 hi, SeqAcc& a) {
    const uint32_t lane = threadIdx.x & 63u;
    double nx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u64 i = lo + (u64)k * 64 + lane;
        nx[k] = i < hi ? p[i] : 0.0;
    }
    for (u64 g0 = lo; g0 < hi; g0 += 1024) {
        double cur[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nx[k];
        if (g0 + 1024 < hi) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const u64 i = g0 + 1024 + (u64)k * 64 + lane;
                nx[k] = i < hi ? p[i] : 0.0;
            }
        }
        seq_span<0, 16>(cur, hi - g0 < 1024 ? (uint32_t)(hi - g0) : 1024u, a);
    }
}

// one wave: out[0] = the sequential sum of p[0..n); zeroes bsum for the slot's next use
// (blockIdx.x = one of several independent sums laid out `stride` values / `nblocks` blocks apart)
__global__ __launch_bounds__(64) void k_seq_chain(const double* p, u64 n, const SeqBlk* blk, uint32_t nblocks, double* bsum, double* out,
                                                  u64 stride, const SeqGrp* grp) {
    const uint32_t lane = threadIdx.x;
    p += (u64)blockIdx.x * stride;
    blk += (size_t)blockIdx.x * nblocks;
    grp += (size_t)blockIdx.x * nblocks * (2 * SQ_GROUPS);
    bsum += (size_t)blockIdx.x * nblocks;
    out += blockIdx.x;
    SeqAcc a;
    a.s = 0.0; a.S = 0; a.e = 0; a.as_int = false;
    SeqBlk next;  // the block records come 64 at a time, one per lane; the next 64 are asked for a round ahead
    next.e = 0; next.flags = 8u; next.Q = 0;
    if (lane < nblocks) next = blk[lane];
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 64) {
        const SeqBlk mine = next;
        next.e = 0; next.flags = 8u; next.Q = 0;
        if (c0 + 64u + lane < nblocks) next = blk[c0 + 64u + lane];
        const uint32_t cn = nblocks - c0 < 64u ? nblocks - c0 : 64u;
        // what a lane's block adds when it goes through as an integer total of the binade at hand (all-zero blocks: nothing)
        const bool zero_blk = (mine.flags & 16u) != 0u;
        const u64 add = zero_blk ? (u64)0 : mine.Q;
        uint32_t j = 0;
        while (j < cn) {
            if (a.as_int) {
                // Every block from j on that goes through — predicted for this binade, no flag, and the running sum
                // still below 2^53 after it — at once: a prefix sum over the lanes and one ballot for the first that
                // does not.
                const bool mineok = lane >= cn || zero_blk || (mine.flags == 0u && mine.e == a.e && mine.Q < ((u64)1 << 53));
                const u64 incl = fsk_hw::wave_incl_sum_u64((lane >= j && lane < cn && mineok) ? add : (u64)0);
                const bool ok = lane < j || (mineok && a.S + incl < ((u64)1 << 53));
                const u64 bad = __ballot(!ok);
                uint32_t stop = bad ? (uint32_t)(__ffsll((long long)bad) - 1) : 64u;
                if (stop > cn) stop = cn;
                if (stop > j) {
                    a.S += wave_bcast_u64(incl, stop - 1u);
                    j = stop;
                    continue;
                }
            }
            // block j on its own
            const uint32_t kf = wave_bcast_u32(mine.flags, j);
            if (kf & 16u) { ++j; continue; }  // all zeros
            const int ke = (int)wave_bcast_u32((uint32_t)mine.e, j);
            if (kf == 0u) {
                const u64 kq = wave_bcast_u64(mine.Q, j);
                if (!a.as_int && a.s > 0.0 && seq_exponent(a.s) == ke) seq_acc_to_int(a);
                if (a.as_int && a.e == ke && a.S + kq < ((u64)1 << 53)) {
                    a.S += kq;
                    ++j;
                    continue;
                }
            }
            const u64 lo = (u64)(c0 + j) * SQ_BLOCK, hi = lo + SQ_BLOCK < n ? lo + SQ_BLOCK : n;
            if (kf & 32u) {
                // group records: the groups before and after the crossing / the tie go through as integer
                // totals of their binade; only the group in between is redone
                SeqGrp mg;
                mg.Q = 0; mg.flags = 1u; mg.pad = 0u;
                if (lane < 2u * SQ_GROUPS) mg = grp[(size_t)(c0 + j) * (2 * SQ_GROUPS) + lane];
                for (uint32_t g = 0; g < (uint32_t)SQ_GROUPS; ++g) {
                    const u64 glo = lo + (u64)g * SQ_GROUP;
                    if (glo >= hi) break;
                    if (wave_bcast_u32(mg.flags, g) & 16u) continue;  // all zeros
                    seq_acc_to_int(a);
                    if (a.as_int && (a.e == ke || a.e == ke + 1)) {
                        const uint32_t r = (a.e == ke ? 0u : (uint32_t)SQ_GROUPS) + g;
                        const u64 gq = wave_bcast_u64(mg.Q, r);
                        if (wave_bcast_u32(mg.flags, r) == 0u && a.S + gq < ((u64)1 << 53)) {
                            a.S += gq;
                            continue;
                        }
                    }
                    const u64 ghi = glo + SQ_GROUP < hi ? glo + SQ_GROUP : hi;
                    seq_range(p, glo, ghi, a);
                }
                ++j;
                continue;
            }
            seq_range(p, lo, hi, a);
            ++j;
        }
    }
    seq_acc_to_double(a);
    if (lane == 0) {
        out[0] = a.s;
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
