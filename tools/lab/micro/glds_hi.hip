// Does the LDS-DMA (global_load_lds) reach LDS addresses >= 64 KiB on gfx950?  (M0 carries the destination base.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const int* src, int* out, int base) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    int* l = (int*)lds;
    for (int i = threadIdx.x; i < 160 * 1024 / 4 - 256; i += 64) l[i] = -1;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gptr_t)(src + threadIdx.x * 4), (lptr_t)(lds + base), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report where lane 0's first dword (value 1000) landed
    int found = -2;
    for (int i = 0; i < 160 * 1024 / 4 - 256; ++i) if (l[i] == 1000) { found = i * 4; break; }
    if (threadIdx.x == 0) { out[0] = found; out[1] = l[base / 4 + 5]; }
}
int main() {
    int h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    for (int base : {0, 32768, 65536, 98304, 131072}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024 - 1024, 0, d, o, base);
        int r[2]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
        printf("base %6d: first dword found at byte %d, l[base+20] = %d (want 1005)\n", base, r[0], r[1]);
    }
    return 0;
}
