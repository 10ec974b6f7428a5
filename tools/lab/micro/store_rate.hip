// Per-CU global store / load throughput on gfx950: G workgroups of 512 threads (160 KiB of LDS each: one per CU), each streaming
// its own `MB` MiB region with 16-byte accesses, one contiguous KiB per wave instruction.  Prints bytes per clock per CU.
// modes: 0 store, 1 nontemporal store, 2 load (sum kept), 3 store 8 B per lane, 4 load + store (copy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(512) void k(uint4* buf, const uint4* src, long per_wg, long long* clk, int* sink) {
    extern __shared__ char lds[];
    uint4* p = buf + (long)blockIdx.x * per_wg;
    const uint4* q = src + (long)blockIdx.x * per_wg;
    const int tid = threadIdx.x;
    uint4 v = make_uint4(tid, 1, 2, 3);
    unsigned acc = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 3) {
        uint2* p2 = (uint2*)p;
        for (long i = tid; i < per_wg * 2; i += 512 * 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) p2[i + u * 512] = make_uint2(v.x, v.y);
        }
    } else {
        for (long i = tid; i < per_wg; i += 512 * 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 0) p[i + u * 512] = v;
                if (MODE == 1) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (u4*)&p[i + u * 512]); }
                if (MODE == 2) { uint4 x = q[i + u * 512]; acc += x.x ^ x.w; }
                if (MODE == 4) { uint4 x = q[i + u * 512]; p[i + u * 512] = x; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    if (acc == 0x12345) sink[0] = acc;
}
template <int MODE> void run(const char* name, int G, long mb, uint4* buf, uint4* src, long long* clk, int* sink) {
    const long per_wg = mb * 1024 * 1024 / 16;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(512), 150 * 1024, 0, buf, src, per_wg, clk, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, clk, sizeof(long long) * G, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < G; ++i) mean += h[i]; mean /= G;
    // readcyclecounter ticks at 100 MHz on gfx9 (s_memtime): report wall-derived rates too
    printf("%-22s G=%3d %3ld MiB/WG: %.1f us, %.2f TB/s aggregate, %.1f GB/s per CU, ticks/WG %.0f\n", name, G, mb, ms * 1e3,
           (double)G * mb * 1048576 / (ms * 1e-3) / 1e12, (double)mb * 1048576 / (ms * 1e-3) / 1e9, mean);
}
int main() {
    const long total = 256L * 8 * 1024 * 1024;
    uint4 *buf, *src; long long* clk; int* sink;
    hipMalloc(&buf, total); hipMalloc(&src, total); hipMalloc(&clk, 8 * 256); hipMalloc(&sink, 4);
    hipMemset(src, 1, total);
    for (int G : {1, 32, 128, 256}) {
        for (long mb : {1L, 4L}) {
            run<0>("store 16B", G, mb, buf, src, clk, sink);
            run<1>("store 16B nontemporal", G, mb, buf, src, clk, sink);
            run<3>("store 8B", G, mb, buf, src, clk, sink);
            run<2>("load 16B", G, mb, buf, src, clk, sink);
            run<4>("copy 16B", G, mb, buf, src, clk, sink);
        }
    }
    return 0;
}
