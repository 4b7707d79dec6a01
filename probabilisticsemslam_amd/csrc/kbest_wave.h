// kbest_wave.h -- wave-level device helpers shared by the kernels (gfx950, wave64).  Device code only.
#ifndef KBEST_WAVE_H
#define KBEST_WAVE_H

#include <hip/hip_runtime.h>

#include "kbest_engine.h"

namespace kb {

__device__ __forceinline__ double d_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

// ---------------------------------------------------------------- wave tools
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double min_keep(double a, double b) { return b < a ? b : a; }

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// fp64 min over the 64 lanes, returned wave-uniform (set-up code only; the hot
// loop uses the integer-key form below).  All lanes must be active.
__device__ __forceinline__ double wave_min_f64(double x)
{
    x = min_keep(x, dpp_f64<0xB1, 0xF>(x));   // quad_perm [1,0,3,2]
    x = min_keep(x, dpp_f64<0x4E, 0xF>(x));   // quad_perm [2,3,0,1]
    x = min_keep(x, dpp_f64<0x141, 0xF>(x));  // row_half_mirror
    x = min_keep(x, dpp_f64<0x140, 0xF>(x));  // row_mirror
    x = min_keep(x, dpp_f64<0x142, 0xA>(x));  // row_bcast:15 -> rows 1,3
    x = min_keep(x, dpp_f64<0x143, 0xC>(x));  // row_bcast:31 -> rows 2,3
    return readlane_f64(x, 63);
}

// Order-preserving integer key of a double: (khi as int32, klo as uint32)
// compared lexicographically == IEEE '<' on the doubles (no NaNs; -0.0 cannot
// occur in a reduced cost).  Negative values (rounding can make a
// tight arc's reduced cost -1e-17) have their magnitude bits flipped.
__device__ __forceinline__ void to_key(double x, int &khi, u32 &klo)
{
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int s = hi >> 31;
    khi = hi ^ (int)((u32)s >> 1);
    klo = (u32)(lo ^ s);
}
__device__ __forceinline__ double from_key(int khi, u32 klo)  // same involution
{
    const int s = khi >> 31;
    return __hiloint2double(khi ^ (int)((u32)s >> 1), (int)klo ^ s);
}
constexpr int KEY_INF_HI = 0x7ff00000;  // key of +inf is (0x7ff00000, 0)

// One VOP2+DPP instruction per butterfly stage; `s_nop 1` covers the two wait
// states a DPP read needs after a VALU write of the same VGPR (the assembler
// does not pad inline asm).  The result lands in lane 63 and is read into an
// SGPR.  EXEC must be all ones.
#define KB_DPP_MIN_CHAIN(OP)                                                             \
    "s_nop 1\n\t" OP " %1, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"         \
    "s_nop 1\n\t" OP " %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"              \
    "s_nop 1\n\t" OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" "v_readlane_b32 %0, %1, 63\n\t"

__device__ __forceinline__ int wave_min_i32(int x)
{
    int r, t;
    asm volatile(KB_DPP_MIN_CHAIN("v_min_i32_dpp") : "=s"(r), "=&v"(t) : "v"(x));
    return r;
}
__device__ __forceinline__ u32 wave_min_u32(u32 x)
{
    u32 r, t;
    asm volatile(KB_DPP_MIN_CHAIN("v_min_u32_dpp") : "=s"(r), "=&v"(t) : "v"(x));
    return r;
}

// force a wave-uniform 64-bit value into SGPRs (values loaded from LDS live in VGPRs)
__device__ __forceinline__ u64 uni64(u64 x)
{
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(x >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ int uni32(int x) { return __builtin_amdgcn_readfirstlane(x); }

// per-lane select driven directly by a 64-bit scalar lane mask
__device__ __forceinline__ int sel32(u64 mask, int ifset, int ifclear)
{
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(ifclear), "v"(ifset), "s"(mask));
    return r;
}

__device__ __forceinline__ void wave_fence()
{
    // LDS traffic between the lanes of one wave: program order is enough in hardware, the fence only keeps the
    // compiler from moving the accesses across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ u64 bit64(int i) { return 1ull << (i & 63); }

// One entry of a result table (row4col / col4row): int32, the interface's type, or int8 with KBEST_FLAG_TABLES_I8 (the
// same values; every index of a problem of up to 127 rows fits, -1 stays -1).  `i8` is uniform over the launch.
__device__ __forceinline__ void put_index(int *table, long long i, int value, bool i8)
{
    if (i8) reinterpret_cast<signed char *>(table)[i] = (signed char)value;
    else table[i] = value;
}

}  // namespace kb
#endif
