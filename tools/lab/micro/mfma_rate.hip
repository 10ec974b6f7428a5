// Issue rate of independent v_mfma_f32_32x32x16_bf16 (AGPR accumulators), with and without an s_nop 1 in front of each.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int NOP>
__global__ __launch_bounds__(256) void k(long* out, int iters) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (NOP == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else if (NOP == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        }
    }
    long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long)s; }
}
int main() {
    long* o; hipMalloc(&o, 16);
    for (int v = 0; v < 3; ++v) for (int blocks : {1, 256}) {
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&] { if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, o, iters);
                         else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, o, iters);
                         else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, o, iters); };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long r[2]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("variant %d blocks %3d: %.2f ns per MFMA (wall), cycle counter %.1f per MFMA\n", v, blocks, ms * 1e6 / (iters * 16.0), r[0] / (iters * 16.0));
    }
    return 0;
}
