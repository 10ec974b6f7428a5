// Common device/host helpers for the TAN HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tan_hip.h"

namespace tal {

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

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
    u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// d/dx [x*sigmoid(1.702x)] = s + 1.702*x*s*(1-s)
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s + 1.702f * x * s * (1.0f - s);
}

inline int hip_ok(hipError_t e) { return (int)e; }
#define TAN_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
#define TAN_REQUIRE(cond) do { if (!(cond)) return TAN_ERR_BAD_ARG; } while (0)

// optional in-stream kernel timer (tan_api.hip); kinds: see TAN_PROF_* in tan_hip.h
int prof_begin(hipStream_t st, int kind, double work);
void prof_end(hipStream_t st, int rec);

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace tal
