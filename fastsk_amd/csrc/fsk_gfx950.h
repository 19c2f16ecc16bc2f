// fsk_gfx950.h — the gfx950 instructions the kernels ask for by name (inline asm and amdgcn builtins). The
// product build (hipcc --offload-arch=gfx950) includes this file through fsk_platform.h; the CPU emulation of the
// test-suite (tests/emu/hip_emu.h) provides functions of the same names, so the kernel sources themselves hold no
// emulator branches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fsk_hw {

// the constant-rate wall clock of the device (s_memrealtime: 100 MHz on gfx950, whatever the shader clock does)
constexpr unsigned long long WALL_TICKS_PER_MS = 100000ull;
__device__ __forceinline__ unsigned long long wall_ticks() { return wall_clock64(); }
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(127); }

// a * b + c with 24-bit operands: v_mad_u32_u24 issues at full rate, a 32-bit multiply-add does not (the
// compiler turns __umul24 of small known ranges back into one, hence the asm). b is wave-uniform.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}

// low 32 bits of a * b for a, b < 2^24: v_mul_u32_u24 issues at full rate, v_mul_lo_u32 at a quarter of it
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// bits [off, off + width) of x (v_bfe_u32: one instruction for a shift and a mask that are not compile-time constants)
__device__ __forceinline__ uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
// bit b of x, sign-extended (0 or -1): one v_bfe_i32 whatever b is
__device__ __forceinline__ int sbfe1(uint32_t x, int b) { return __builtin_amdgcn_sbfe((int)x, (unsigned)b, 1u); }
// p & ~(m ^ t) as ONE v_bitop3_b32 (truth table 0x90 over p = 0xf0, m = 0xcc, t = 0xaa). Left to itself the compiler
// regroups a chain of these into xors joined by v_or3 — 1.5 instructions a step instead of one.
__device__ __forceinline__ uint32_t and_xnor(uint32_t p, uint32_t m, uint32_t t) { return __builtin_amdgcn_bitop3_b32(p, m, t, 0x90); }

// value of lane `src` (wave-uniform index) in every lane: a scalar read, not an LDS permute
__device__ __forceinline__ uint32_t readlane(uint32_t x, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)src);
}

// sum of x over the 64 lanes, in every lane, through the DPP row shifts / broadcasts of the VALU rather than six
// dependent LDS permutes per 32-bit half: rows of 16 lanes first, then row 0 -> 1 and 2 -> 3, then lane 31 -> rows 2-3
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x) {
#define FSK_DPP_ADD64(ctrl, rows)                                                                                   \
    {                                                                                                               \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rows, 0xf, true);      \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rows, 0xf, true); \
        x += ((unsigned long long)hi << 32) | lo;                                                                   \
    }
    FSK_DPP_ADD64(0x111, 0xf)  // row_shr:1
    FSK_DPP_ADD64(0x112, 0xf)  // row_shr:2
    FSK_DPP_ADD64(0x114, 0xf)  // row_shr:4
    FSK_DPP_ADD64(0x118, 0xf)  // row_shr:8  -> lane 15 of every row holds the row's sum
    FSK_DPP_ADD64(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FSK_DPP_ADD64(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef FSK_DPP_ADD64
    return ((unsigned long long)readlane((uint32_t)(x >> 32), 63u) << 32) | readlane((uint32_t)x, 63u);
}

// running maximum over the lanes (lane l gets the maximum of lanes 0..l): DPP row shifts inside the rows of 16 lanes, then
// row 0 -> 1 and 2 -> 3, then lane 31 -> rows 2-3. Six v_max_i32_dpp: the shift rides on the maximum itself, and a lane
// whose source does not exist (the row's first lanes, the rows a broadcast leaves out) keeps its value because a DPP
// instruction without bound_ctrl does not write it. (Through __builtin_amdgcn_update_dpp the compiler issues a v_mov of the
// identity, a v_mov_dpp and the v_max per step — eighteen instructions, and k_sx_seg_write, which is bound by the VALU
// instructions it issues, runs one or two of these scans per 64 records.) s_nop 1: the two wait states a DPP read of a
// VGPR needs after the VALU write before it.
__device__ __forceinline__ int wave_incl_max_i32(int x) {
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x));
    return x;
}

// running sum over the lanes (lane l gets the sum of lanes 0..l): the same DPP steps as wave_sum_u64
__device__ __forceinline__ unsigned long long wave_incl_sum_u64(unsigned long long x) {
#define FSK_DPP_ADD64(ctrl, rows)                                                                                   \
    {                                                                                                               \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rows, 0xf, true);      \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rows, 0xf, true); \
        x += ((unsigned long long)hi << 32) | lo;                                                                   \
    }
    FSK_DPP_ADD64(0x111, 0xf)  // row_shr:1
    FSK_DPP_ADD64(0x112, 0xf)  // row_shr:2
    FSK_DPP_ADD64(0x114, 0xf)  // row_shr:4
    FSK_DPP_ADD64(0x118, 0xf)  // row_shr:8
    FSK_DPP_ADD64(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FSK_DPP_ADD64(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef FSK_DPP_ADD64
    return x;
}

// the same for 32-bit values (sums below 2^32): six VALU instructions
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t x) {
#define FSK_DPP_ADD32(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, true);
    FSK_DPP_ADD32(0x111, 0xf)  // row_shr:1
    FSK_DPP_ADD32(0x112, 0xf)  // row_shr:2
    FSK_DPP_ADD32(0x114, 0xf)  // row_shr:4
    FSK_DPP_ADD32(0x118, 0xf)  // row_shr:8
    FSK_DPP_ADD32(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FSK_DPP_ADD32(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef FSK_DPP_ADD32
    return x;
}

// sum over the lanes, in every lane (the running sum's last lane, read as a scalar)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { return readlane(wave_incl_sum_u32(x), 63u); }

// a copy of a VGPR the compiler cannot see through (see the flush of the tile kernels: a 64-bit operand built from
// the accumulator itself makes hipcc keep every accumulator in the low half of a register pair)
__device__ __forceinline__ uint32_t vgpr_copy(uint32_t x) {
    uint32_t y;
    asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));
    return y;
}

// Two 16-byte buffer loads of one count panel's rows: `rows` (uniform) = first dword row of the stage, `bytes` = the
// rows of it that exist (the rest reads as zero, no bounds branch), lane_off = the lane's dword offset inside 16 rows;
// `hi` = the same lane offset 16 rows further.
__device__ __forceinline__ void panel_rows_2x16(const uint32_t* rows, uint32_t bytes, uint32_t lane_off, uint4& lo, uint4& hi) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(rows), 0, (int)bytes, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane_off * 4u), 0, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane_off * 4u), 16 * 64 * 4, 0);
    lo = make_uint4(a.x, a.y, a.z, a.w);
    hi = make_uint4(b.x, b.y, b.z, b.w);
}

// LDS byte address of a pointer known to point into LDS
typedef uint32_t lds_addr_t;
__device__ __forceinline__ lds_addr_t lds_address(const void* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// 16 dword rows x 256 B of one count panel STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds): every wave lands 4 rows
// (1 KB, contiguous in HBM and in LDS) at `lds_wave` + the stage's offset; `rows` / `bytes` as above, lane_off_b = the
// lane's byte offset inside the 16 rows. M0 (the LDS address) is a reserved register for the compiler: it is saved
// and restored around the load. Inline asm: through the builtin the compiler would order every later ds_read behind
// the load (it cannot tell the two stage buffers apart); the wait is explicit (wait_panel_rows).
__device__ __forceinline__ void panel_rows_to_lds(const uint32_t* rows, uint32_t bytes, lds_addr_t lds_dst, uint32_t lane_off_b) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long base = (unsigned long long)(size_t)rows;
    i32x4 rs;
    rs.x = (int)(uint32_t)base;
    rs.y = (int)((uint32_t)(base >> 32) & 0xffffu);
    rs.z = (int)bytes;
    rs.w = 0x00020000;
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "s"(lds_dst), "v"(lane_off_b), "s"(rs)
                 : "memory");
}
__device__ __forceinline__ void wait_panel_rows() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// host: let a kernel use `bytes` of dynamic LDS (beyond the 64 KB a launch gets by default)
template <typename F>
inline hipError_t allow_dynamic_lds(F kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace fsk_hw

// how many bits of the 64-lane mask (lo, hi) are set in the lanes below the calling one: v_mbcnt_lo + v_mbcnt_hi
__device__ __forceinline__ unsigned fsk_mbcnt(unsigned lo, unsigned hi) {
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// Loads through a pointer KNOWN to point into LDS, kept in the LDS address space (ds_read_*) whatever pointer type
// the surrounding code uses: a `const T*` parameter is a GENERIC pointer, and where the compiler cannot prove the
// address space — a select between an LDS and a global source, an array of column pointers, a volatile counter —
// the access becomes a flat_load (FSK_LDS_VOLATILE_U32: a 32-bit word re-read at every use, a counter other lanes
// add to between two reads; as `volatile uint32_t*` it was flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0)).
#define FSK_LDS_LOAD_U8(ptr) (*(const __attribute__((address_space(3))) unsigned char*)(ptr))
#define FSK_LDS_LOAD_U16(ptr) (*(const __attribute__((address_space(3))) unsigned short*)(ptr))
#define FSK_LDS_LOAD_U32(ptr) (*(const __attribute__((address_space(3))) unsigned int*)(ptr))
#define FSK_LDS_LOAD_U64(ptr) (*(const __attribute__((address_space(3))) unsigned long long*)(ptr))
#define FSK_LDS_VOLATILE_U32(ptr) (*(volatile __attribute__((address_space(3))) unsigned int*)(ptr))
