// Common device/host helpers for the TAN HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/tan_hip.h"

namespace tal {

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN kept quiet: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer emulation it replaces
// was ~7 VALU operations per element and made every bf16 epilogue VALU-bound)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {      // bf16(lo) | bf16(hi) << 16
    f32x2_t v;
    v[0] = lo; v[1] = hi;
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &b, 4);
    return u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.0f) & 0xffffu); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};
template <typename T> __device__ __forceinline__ float ld_f(const T* p) { return Cvt<T>::to_f(*p); }
template <typename T> __device__ __forceinline__ void st_f(T* p, float v) { *p = Cvt<T>::from_f(v); }

// 4 consecutive elements <-> float4 (16 B for f32, 8 B for bf16); pointers must be suitably aligned
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = f2bf2(v.x, v.y);
    u.y = f2bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}

// Wave-wide (64-lane) reductions, result in every lane.  `__shfl_xor` compiles to six dependent ds_bpermute_b32 (an LDS-crossbar
// round trip each); here the four steps inside a row of 16 lanes are DPP modifiers of the add itself (quad_perm, row_half_mirror,
// row_mirror) and the two cross-row steps are gfx950's v_permlane16_swap / v_permlane32_swap (with both operands = v, the swap
// leaves [v.row0,v.row0,v.row2,v.row2] / [v.row1,v.row1,v.row3,v.row3], resp. [lo,lo] / [hi,hi]: their sum is the xor-16 / xor-32 step).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(r16[0]) + __int_as_float(r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r32[0]) + __int_as_float(r32[1]);
}
// Sums of 16 per-lane values over the 32 lanes that share lane >> 5, 38 instructions instead of 16 x 6: every level adds the
// partner lane's value for two values at once and keeps one of them per lane (v_permlane16_swap does both in one go for the
// lanes 16 apart; then row_ror:8, row_half_mirror, quad_perm [1,0,3,2], and a last quad_perm [2,3,0,1] add).  Lane l ends up
// with the total of value index ((l >> 4) & 1) | ((l >> 3) & 1) << 1 | ((l >> 2) & 1) << 2 | (l & 1) << 3 (lanes l and l ^ 2 hold the same).
__device__ __forceinline__ float pn_colsum16(const float (&v)[16], int lane) {
    float w[8], u[4], y[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v[2 * i]), __float_as_int(v[2 * i + 1]), false, false);
        w[i] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t0 = w[2 * j] + dpp_move<0x128>(w[2 * j]), t1 = w[2 * j + 1] + dpp_move<0x128>(w[2 * j + 1]);     // row_ror:8
        u[j] = (lane & 8) ? t1 : t0;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float x0 = u[2 * k] + dpp_move<0x141>(u[2 * k]), x1 = u[2 * k + 1] + dpp_move<0x141>(u[2 * k + 1]);      // row_half_mirror
        y[k] = (lane & 4) ? x1 : x0;
    }
    const float z0 = y[0] + dpp_move<0xB1>(y[0]), z1 = y[1] + dpp_move<0xB1>(y[1]);                                   // quad_perm [1,0,3,2]
    const float zz = (lane & 1) ? z1 : z0;
    return zz + dpp_move<0x4E>(zz);                                                                                  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ int pn_colsum16_index(int lane) {
    return ((lane >> 4) & 1) | (((lane >> 3) & 1) << 1) | (((lane >> 2) & 1) << 2) | ((lane & 1) << 3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = fmaxf(__int_as_float(r16[0]), __int_as_float(r16[1]));
    auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return fmaxf(__int_as_float(r32[0]), __int_as_float(r32[1]));
}

// 8 consecutive bf16 <-> 8 floats (one 16-byte access per lane: the widest, and per byte the cheapest, global access)
struct f8 { float v[8]; };
__device__ __forceinline__ f8 ld8(const bf16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    f8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const f8& a) {
    uint4 u;
    u.x = f2bf2(a.v[0], a.v[1]);
    u.y = f2bf2(a.v[2], a.v[3]);
    u.z = f2bf2(a.v[4], a.v[5]);
    u.w = f2bf2(a.v[6], a.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ f8 ld8f(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f8 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// d/dx [x*sigmoid(1.702x)] = s + 1.702*x*s*(1-s)
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s + 1.702f * x * s * (1.0f - s);
}

// bf16-output variants: v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU operations); the results
// are rounded to bf16 anyway.  The f32 parity mode keeps the exact forms above.
__device__ __forceinline__ float quick_gelu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
__device__ __forceinline__ float quick_gelu_grad_fast(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
    return s + 1.702f * x * s * (1.0f - s);
}
template <typename TC> __device__ __forceinline__ float quick_gelu_t(float x) {
    if constexpr (sizeof(TC) == 2) return quick_gelu_fast(x); else return quick_gelu(x);
}
template <typename TC> __device__ __forceinline__ float quick_gelu_grad_t(float x) {
    if constexpr (sizeof(TC) == 2) return quick_gelu_grad_fast(x); else return quick_gelu_grad(x);
}

inline int hip_ok(hipError_t e) { return (int)e; }
#define TAN_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
#define TAN_REQUIRE(cond) do { if (!(cond)) return TAN_ERR_BAD_ARG; } while (0)

// optional in-stream kernel timer (tan_api.hip); kinds: see TAN_PROF_* in tan_hip.h
int prof_begin(hipStream_t st, int kind, double work);
void prof_end(hipStream_t st, int rec);

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that launches with more than 48 KiB of dynamic LDS: once per DEVICE (a
// function attribute belongs to the device's copy of the code object) and safe under concurrent first calls (`done` = one bit per
// device id, set only after the attribute call succeeded; a plain `static bool` was neither -- ADVICE r4).
inline hipError_t ensure_dyn_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace tal
