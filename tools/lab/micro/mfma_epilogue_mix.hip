// What does one body of the row-panel MLP cost a SIMD when nothing but the issue slots is in play (gfx950)?
// A body = per wave 128 MFMAs (32x32x16 bf16: 4 096 clk of the matrix pipe, 8 192 per SIMD with two waves) + the QuickGELU epilogue
// of a chunk: 16 half-units x (2 exp + 2 rcp + ~20 plain VALU operations).  512 threads = 8 waves = 2 per SIMD, no memory, no LDS.
//   0  MFMAs only                                   1  epilogue arithmetic only
//   2  uniform interleave (1 unit per 8 MFMAs)      3  slot X = 96 MFMAs | barrier | slot Y = 32 MFMAs + 16 units   (in phase)
//   4  waves 0-3 run X while waves 4-7 run Y, then the other way round (barrier between)
//   7  the round-2 schedule (64 | 64 + 16 units) in opposite slots   8  80 | 48 + 16 units   9  mode 2 with a barrier per 64 MFMAs
//   5  transcendental rate: 64 v_exp_f32 per iteration, every wave        6  plain rate: 64 v_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define MFMA(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(a), "v"(b))
// one epilogue half-unit on two elements: scale, exp2, 1 + e, rcp, x * s, pack (+ a few address / select operations)
#define UNIT(x) asm volatile( \
    "v_mul_f32 %0, 0x3fb8aa3b, %0\n\tv_mul_f32 %1, 0x3fb8aa3b, %1\n\tv_exp_f32 %2, %0\n\tv_exp_f32 %3, %1\n\t" \
    "v_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t" \
    "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\tv_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\t" \
    "v_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\tv_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\t" \
    "v_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\tv_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\t" \
    "v_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5\n\tv_fma_f32 %4, %0, %1, %4\n\tv_fma_f32 %5, %2, %3, %5" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]))
template <int MODE>
__global__ __launch_bounds__(512) void k(long* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
    float x[6] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f};
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 128; ++i) MFMA(i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) UNIT(x);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 128; ++i) { MFMA(i); if ((i & 7) == 7) UNIT(x); }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 96; ++i) MFMA(i);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 32; ++i) { MFMA(i); if (i & 1) UNIT(x); }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 4) {
            if ((wave < 4) == ((it & 1) == 0)) {
#pragma unroll
                for (int i = 0; i < 96; ++i) MFMA(i);
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) { MFMA(i); if (i & 1) UNIT(x); }
            }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 7) {       // the round-2 schedule: 64 MFMAs | 64 MFMAs + 16 units (one per 4 MFMAs), groups in opposite slots
            if ((wave < 4) == ((it & 1) == 0)) {
#pragma unroll
                for (int i = 0; i < 64; ++i) MFMA(i);
            } else {
#pragma unroll
                for (int i = 0; i < 64; ++i) { MFMA(i); if ((i & 3) == 3) UNIT(x); }
            }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 8) {       // 80 MFMAs | 48 MFMAs + 16 units (one per 3 MFMAs)
            if ((wave < 4) == ((it & 1) == 0)) {
#pragma unroll
                for (int i = 0; i < 80; ++i) MFMA(i);
            } else {
#pragma unroll
                for (int i = 0; i < 48; ++i) { MFMA(i); if (i % 3 == 2) UNIT(x); }
            }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 9) {       // uniform, with the two slot barriers of a body
#pragma unroll
            for (int i = 0; i < 128; ++i) { MFMA(i); if ((i & 7) == 7) UNIT(x); if ((i & 63) == 63) __builtin_amdgcn_s_barrier(); }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
        }
    }
    long t1 = __builtin_readcyclecounter();
    float s = x[0] + x[1] + x[2] + x[3] + x[4] + x[5]; for (int i = 0; i < 4; ++i) s += acc[i][0];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[8] = (long)s; }
}
template <int MODE> void run(const char* what, long* o, double per) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, iters); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, o, iters);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long r[9]; hipMemcpy(r, o, 72, hipMemcpyDeviceToHost);
    printf("%-64s cycles per %s: wave0 %.0f  wave4 %.0f   (%.2f us per iteration)\n", what, (MODE == 4 || MODE == 7 || MODE == 8) ? "HALF body" : "iteration", r[0] / (double)iters / per,
           r[4] / (double)iters / per, ms * 1e3 / iters);
}
int main() {
    long* o; hipMalloc(&o, 128);
    run<0>("128 MFMAs per wave (8 192 clk of the pipe per SIMD)", o, 1);
    run<1>("16 epilogue half-units per wave", o, 1);
    run<2>("128 MFMAs, one unit after every 8th", o, 1);
    run<3>("96 MFMAs | barrier | 32 MFMAs + 16 units | barrier", o, 1);
    run<4>("wave groups in opposite slots (two iterations = one body)", o, 1);
    run<7>("round-2 schedule: 64 | 64 + 16 units, groups in opposite slots", o, 1);
    run<8>("80 | 48 + 16 units, groups in opposite slots", o, 1);
    run<9>("128 MFMAs, one unit after every 8th, barrier after every 64th", o, 1);
    run<5>("64 v_exp_f32", o, 1);
    run<6>("64 v_fma_f32", o, 1);
    return 0;
}
