// MFMA issue rate with TWO waves per SIMD (512-thread workgroups), 8 accumulator tiles per wave, compiler-allocated registers
// (builtin) vs AGPR-pinned asm, with and without a workgroup barrier every 64 MFMAs.  ns per MFMA per SIMD, 1 and 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int THREADS, int MODE, bool BAR>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[2];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(float)((threadIdx.x + e + j) & 7);
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(float)((threadIdx.x * 3 + e + j) & 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 1], b[i & 1], acc[i], 0, 0, 0);
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i >> 1]), "v"(b[i & 1]));
            }
            if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
}
template <int THREADS, int MODE, bool BAR>
void run(const char* name, float* o) {
    for (int blocks : {1, 256}) {
        const int iters = 400;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<THREADS, MODE, BAR>), dim3(blocks), dim3(THREADS), 0, 0, o, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL((k<THREADS, MODE, BAR>), dim3(blocks), dim3(THREADS), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = iters * 64.0 * (THREADS / 256);
        printf("%-44s blocks %3d: %.2f ns per MFMA per SIMD\n", name, blocks, ms * 1e6 / per_simd);
    }
}
int main() {
    float* o; hipMalloc(&o, 16);
    run<256, 0, false>("256 thr, builtin", o);
    run<256, 1, false>("256 thr, asm AGPR", o);
    run<512, 0, false>("512 thr, builtin", o);
    run<512, 1, false>("512 thr, asm AGPR", o);
    run<512, 0, true>("512 thr, builtin, barrier / 64", o);
    run<512, 1, true>("512 thr, asm AGPR, barrier / 64", o);
    return 0;
}
