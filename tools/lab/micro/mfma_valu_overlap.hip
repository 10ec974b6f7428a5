// Do VALU instructions (v_exp_f32 / v_fma_f32) of one wave overlap with the MFMAs of the other wave on the same SIMD (gfx950)?
// 512 threads = 8 waves = 2 per SIMD.  mode 0: every wave MFMAs only; 1: every wave VALU only; 2: waves 0-3 MFMA, waves 4-7 VALU;
// 3: every wave alternates bursts of 16 MFMAs and 64 VALU ops; 4: every wave interleaves 1 MFMA : 4 VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define MFMA(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define VALU4(x) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
template <int MODE>
__global__ __launch_bounds__(512) void k(long* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
    float x[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && wave < 4)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) MFMA(i & 3);
        }
        if (MODE == 1 || (MODE == 2 && wave >= 4)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) VALU4(x);
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) MFMA(i & 3);
#pragma unroll
            for (int i = 0; i < 16; ++i) VALU4(x);
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { MFMA(i & 3); VALU4(x); }
        }
    }
    long t1 = __builtin_readcyclecounter();
    float s = x[0] + x[1] + x[2] + x[3]; for (int i = 0; i < 4; ++i) s += acc[i][0];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[8] = (long)s; }
}
template <int MODE> void run(const char* what, long* o) {
    const int iters = 500;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, iters); hipDeviceSynchronize();
    long r[9]; hipMemcpy(r, o, 72, hipMemcpyDeviceToHost);
    printf("%-58s cycles per iteration: wave0 %.0f  wave4 %.0f\n", what, r[0] / (double)iters, r[4] / (double)iters);
}
int main() {
    long* o; hipMalloc(&o, 128);
    run<0>("all waves: 16 MFMA (2 waves/SIMD -> 32 MFMA x 32 clk)", o);
    run<1>("all waves: 64 VALU (32 exp + 32 fma)", o);
    run<2>("waves 0-3: 16 MFMA | waves 4-7: 64 VALU", o);
    run<3>("all waves: 16 MFMA burst, then 64 VALU burst", o);
    run<4>("all waves: 1 MFMA : 4 VALU interleaved", o);
    return 0;
}
