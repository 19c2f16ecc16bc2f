// tests/emu/hip_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A tiny single-process emulator of the subset of HIP that fastsk_amd/csrc uses, so that the
// kernels' *logic* (indexing, barriers, wave64 ballots/shuffles, LDS staging, atomics) can be
// exercised by `pytest -m "not gpu"` in a container that has no GPU. It is compiled ONLY into
// tests/emu/libfastsk_emu.so by tests/emu/build_emu.py; the product library (hipcc, gfx950)
// never sees this file and the product loader never loads the emulated library.
//
// Model: a launch runs its workgroups one after another on the calling OS thread; the threads of
// one workgroup are ucontext fibers scheduled round-robin, switching only at __syncthreads() and
// at wave-collective operations (__ballot/__shfl*). `__shared__` becomes `static`, which is sound
// because only one workgroup is alive at a time. Wavefront width is 64, as on gfx950.
// Limitations: wave collectives must be reached by every live lane of the wave (no divergent
// ballots); there is no memory-model emulation (races that need two CUs cannot be found here).
#pragma once
#include <ucontext.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define FSK_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
struct double2 { double x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace emu {
constexpr int WAVE = 64;
constexpr size_t STACK = 128 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
};

struct Block {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = 0;
    int nthreads = 0;
    int live = 0;
    // block barrier
    int bar_count = 0;
    unsigned bar_gen = 0;
    // per-wave collective slots
    struct WaveSlot {
        unsigned long long vals[WAVE];
        bool pred[WAVE];
        bool present[WAVE];
        int arrived = 0;
        int live = 0;
        unsigned gen = 0;
        unsigned long long ballot = 0;
        unsigned long long snap[WAVE];
    };
    std::vector<WaveSlot> waves;
    dim3 bid, bdim, gdim;
    char* dyn_shared = nullptr;
    const std::function<void()>* body = nullptr;
};

inline Block*& cur_block() {
    static Block* b = nullptr;
    return b;
}

inline void yield() {
    Block* b = cur_block();
    Fiber& f = b->fibers[b->cur];
    swapcontext(&f.ctx, &b->sched);
}

inline void fiber_entry() {
    Block* b = cur_block();
    (*b->body)();
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    b->live--;
    b->waves[b->cur / WAVE].live--;
    // a thread that exits counts as arrived for any barrier others are waiting on
    if (b->live > 0 && b->bar_count >= b->live) { b->bar_count = 0; b->bar_gen++; }
    Block::WaveSlot& w = b->waves[b->cur / WAVE];
    if (w.live > 0 && w.arrived >= w.live) {
        // complete a pending wave collective on behalf of the remaining lanes
        unsigned long long m = 0;
        for (int l = 0; l < WAVE; ++l) if (w.present[l] && w.pred[l]) m |= 1ull << l;
        w.ballot = m;
        memcpy(w.snap, w.vals, sizeof(w.snap));
        w.arrived = 0;
        for (int l = 0; l < WAVE; ++l) w.present[l] = false;
        w.gen++;
    }
    swapcontext(&f.ctx, &b->sched);
}

inline void run_block(Block& b) {
    cur_block() = &b;
    b.live = b.nthreads;
    b.bar_count = 0;
    for (auto& w : b.waves) { w.arrived = 0; w.live = 0; for (int l = 0; l < WAVE; ++l) w.present[l] = false; }
    for (int t = 0; t < b.nthreads; ++t) {
        Fiber& f = b.fibers[t];
        f.done = false;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &b.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        b.waves[t / WAVE].live++;
    }
    while (b.live > 0) {
        for (int t = 0; t < b.nthreads; ++t) {
            if (b.fibers[t].done) continue;
            b.cur = t;
            swapcontext(&b.sched, &b.fibers[t].ctx);
        }
    }
}

inline std::mutex& launch_mutex() {
    static std::mutex m;
    return m;
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    // one launch at a time in the process (`__shared__` is `static`, the block below is shared): the worker
    // threads of a multi-device group take turns
    std::lock_guard<std::mutex> one_at_a_time(launch_mutex());
    static Block b;  // fibers (and their stacks) are reused across launches
    int nthreads = (int)(block.x * block.y * block.z);
    if ((int)b.fibers.size() < nthreads) {
        size_t old = b.fibers.size();
        b.fibers.resize(nthreads);
        for (size_t i = old; i < b.fibers.size(); ++i) b.fibers[i].stack = (char*)malloc(STACK);
    }
    b.nthreads = nthreads;
    b.waves.resize((nthreads + WAVE - 1) / WAVE);
    b.bdim = block;
    b.gdim = grid;
    b.body = &body;
    std::vector<char> dyn(shmem + 64);
    b.dyn_shared = dyn.data();
    for (int t = 0; t < nthreads; ++t)
        b.fibers[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                b.bid = dim3(x, y, z);
                run_block(b);
            }
    cur_block() = nullptr;
}

inline void syncthreads() {
    Block* b = cur_block();
    unsigned gen = b->bar_gen;
    if (++b->bar_count >= b->live) { b->bar_count = 0; b->bar_gen++; return; }
    while (b->bar_gen == gen) yield();
}

// wave collective: every live lane deposits (pred, val); returns after all have.
inline Block::WaveSlot& wave_exchange(bool pred, unsigned long long val) {
    Block* b = cur_block();
    int lane = b->cur % WAVE;
    Block::WaveSlot& w = b->waves[b->cur / WAVE];
    w.pred[lane] = pred;
    w.vals[lane] = val;
    w.present[lane] = true;
    unsigned gen = w.gen;
    if (++w.arrived >= w.live) {
        unsigned long long m = 0;
        for (int l = 0; l < WAVE; ++l) if (w.present[l] && w.pred[l]) m |= 1ull << l;
        w.ballot = m;
        memcpy(w.snap, w.vals, sizeof(w.snap));
        w.arrived = 0;
        for (int l = 0; l < WAVE; ++l) w.present[l] = false;
        w.gen++;
    } else {
        while (w.gen == gen) yield();
    }
    return w;
}
inline int lane_id() { return cur_block()->cur % WAVE; }
}  // namespace emu

#define threadIdx (emu::cur_block()->fibers[emu::cur_block()->cur].tid)
#define blockIdx (emu::cur_block()->bid)
#define blockDim (emu::cur_block()->bdim)
#define gridDim (emu::cur_block()->gdim)
#define FSK_DYN_SHARED(type, name) type* name = (type*)(((uintptr_t)emu::cur_block()->dyn_shared + 15) & ~(uintptr_t)15)

static inline void __syncthreads() { emu::syncthreads(); }
static inline unsigned long long __ballot(int pred) { return emu::wave_exchange(pred != 0, 0).ballot; }
template <typename T> static inline T __shfl(T v, int src) {
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    auto& w = emu::wave_exchange(false, raw);
    // snapshot taken at completion: safe even if other lanes start the next collective first
    unsigned long long r = w.snap[src & 63];
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T> static inline T __shfl_up(T v, unsigned d) {
    int l = emu::lane_id();
    T o = __shfl(v, l - (int)d < 0 ? l : l - (int)d);
    return o;
}
template <typename T> static inline T __shfl_down(T v, unsigned d) {
    int l = emu::lane_id();
    return __shfl(v, l + (int)d > 63 ? l : l + (int)d);
}
template <typename T> static inline T __shfl_xor(T v, int m) { return __shfl(v, emu::lane_id() ^ m); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned x) { return x; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }

static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
}
static inline unsigned __builtin_amdgcn_udot8(unsigned a, unsigned b, unsigned c, bool) {
    for (int i = 0; i < 8; ++i) c += ((a >> (4 * i)) & 15u) * ((b >> (4 * i)) & 15u);
    return c;
}
static inline double __dsqrt_rn(double x) { return __builtin_sqrt(x); }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }

// ---- host runtime shims ---------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct emuEvent { std::chrono::steady_clock::time_point t; };
typedef emuEvent* hipEvent_t;
constexpr hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetLastError() { return 0; }
// eight pretend devices (one address space: "device memory" is host memory), current device per thread
namespace emu { inline int& cur_device() { static thread_local int d = 0; return d; } }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= 8) return 1; emu::cur_device() = d; return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = emu::cur_device(); return 0; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 8; return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline hipError_t hipHostMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
constexpr hipError_t hipErrorNotReady = 600;
static inline hipError_t hipEventQuery(hipEvent_t) { return 0; }  // (launches run to completion before they return)
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = *tot = (size_t)8 << 30; return 0; }

#define FSK_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

// ---- what fastsk_amd/csrc/fsk_gfx950.h gives the product build: the same names, plain C++ ---------------
namespace fsk_hw {
constexpr unsigned long long WALL_TICKS_PER_MS = 100000ull;
static inline unsigned long long wall_ticks() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10ull;
}
static inline void nap() {}
static inline unsigned mad24(unsigned a, unsigned b, unsigned c) { return (a & 0xffffffu) * (b & 0xffffffu) + c; }
static inline unsigned mul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xffffffu) * (b & 0xffffffu)); }
static inline int wave_incl_max_i32(int x) {
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if ((int)emu::lane_id() >= d && y > x) x = y;
    }
    return x;
}
static inline int sbfe1(unsigned x, int b) { return (x >> b) & 1u ? -1 : 0; }
static inline unsigned and_xnor(unsigned p, unsigned m, unsigned t) { return p & ~(m ^ t); }
static inline unsigned bfe(unsigned x, unsigned off, unsigned width) { return (x >> off) & (width >= 32u ? 0xffffffffu : (1u << width) - 1u); }
static inline unsigned readlane(unsigned x, unsigned src) { return __shfl(x, (int)src); }
static inline unsigned wave_sum_u32(unsigned x) {
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    return x;
}
static inline unsigned long long wave_sum_u64(unsigned long long x) {
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    return x;
}
static inline unsigned long long wave_incl_sum_u64(unsigned long long x) {
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(x, d);
        if ((int)emu::lane_id() >= d) x += y;
    }
    return x;
}
static inline unsigned wave_incl_sum_u32(unsigned x) {
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = __shfl_up(x, d);
        if ((int)emu::lane_id() >= d) x += y;
    }
    return x;
}
static inline unsigned vgpr_copy(unsigned x) { return x; }
// rows past `bytes` read as zero (the buffer descriptor's bounds check)
static inline uint4 rows16(const unsigned* rows, unsigned bytes, unsigned byte_off) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (byte_off + 16u <= bytes) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(rows) + byte_off);
    return v;
}
static inline void panel_rows_2x16(const unsigned* rows, unsigned bytes, unsigned lane_off, uint4& lo, uint4& hi) {
    lo = rows16(rows, bytes, lane_off * 4u);
    hi = rows16(rows, bytes, lane_off * 4u + 16u * 64u * 4u);
}
typedef uintptr_t lds_addr_t;
static inline lds_addr_t lds_address(const void* p) { return reinterpret_cast<uintptr_t>(p); }
// (a wave's 64 lanes land 16 bytes each, consecutively, at lds_dst)
static inline void panel_rows_to_lds(const unsigned* rows, unsigned bytes, lds_addr_t lds_dst, unsigned lane_off_b) {
    *reinterpret_cast<uint4*>(lds_dst + (uintptr_t)emu::lane_id() * 16u) = rows16(rows, bytes, lane_off_b);
}
static inline void wait_panel_rows() {}
template <typename F> static inline hipError_t allow_dynamic_lds(F, size_t) { return 0; }
}  // namespace fsk_hw
static inline unsigned fsk_mbcnt(unsigned lo, unsigned hi) {
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    return (unsigned)__builtin_popcountll(m & ((1ull << emu::lane_id()) - 1ull));
}
#define FSK_LDS_LOAD_U8(ptr) (*reinterpret_cast<const unsigned char*>(ptr))
#define FSK_LDS_LOAD_U16(ptr) (*reinterpret_cast<const unsigned short*>(ptr))
#define FSK_LDS_LOAD_U32(ptr) (*reinterpret_cast<const unsigned int*>(ptr))
#define FSK_LDS_LOAD_U64(ptr) (*reinterpret_cast<const unsigned long long*>(ptr))
#define FSK_LDS_VOLATILE_U32(ptr) (*reinterpret_cast<volatile unsigned int*>(ptr))
