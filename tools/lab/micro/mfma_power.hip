// Is the dense MFMA rate data- and load-dependent?  v_mfma_f32_32x32x16_bf16, 2 waves per SIMD, 8 accumulator tiles per wave:
// operands = small integers (0..7: few toggling bits) vs pseudo-random bf16 in (-1, 1), on 32 ... 256 CUs, short and long runs.
// Prints ns per MFMA per SIMD and the TFLOP/s of the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int DATA>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[2];
    for (int j = 0; j < 6; ++j) for (int e = 0; e < 8; ++e) {
        const unsigned h = hash((blockIdx.x * 512 + threadIdx.x) * 64 + j * 8 + e);
        float v = DATA == 0 ? (float)((threadIdx.x + e + j) & 7) : DATA == 1 ? ((int)(h & 0xffff) - 32768) / 32768.0f * 0.05f : 0.f;
        if (j < 4) a[j][e] = (__bf16)v; else b[j - 4][e] = (__bf16)v;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 1], b[i & 1], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
}
template <int DATA>
void run(const char* name, float* o) {
    for (int iters : {25, 400}) for (int blocks : {32, 64, 128, 192, 256}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<DATA>), dim3(blocks), dim3(512), 0, 0, o, iters); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL((k<DATA>), dim3(blocks), dim3(512), 0, 0, o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = iters * 64.0 * 2, flops = (double)blocks * 8 * iters * 64.0 * 32768.0;
        printf("%-26s iters %3d blocks %3d: %7.1f us  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s\n", name, iters, blocks, ms * 1e3, ms * 1e6 / per_simd, flops / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* o; (void)hipMalloc(&o, 16);
    run<2>("zeros", o);
    run<0>("small integers", o);
    run<1>("random bf16 (|x| < 0.05)", o);
    return 0;
}
