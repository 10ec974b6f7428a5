// Probe of two gfx950 primitives used by the GEMM: ds_read_b64_tr_b16 lane mapping and global_load_lds_dwordx4.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/probe_tr ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe_tr(const int* addr_elems, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const unsigned byte_addr = (unsigned)(uintptr_t)(lds) + addr_elems[lane] * 2;
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

__global__ void probe_glds(const uint4* src, uint4* out) {
    __shared__ __attribute__((aligned(16))) uint4 lds[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave copies 64 x 16 B: lane reads src[perm(lane)], LDS dest = wave base + lane*16
    const int srcidx = wave * 64 + (lane ^ 5);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + srcidx),
                                     (void __attribute__((address_space(3)))*)(lds + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}

int main() {
    int h_addr[64];
    short* d_out; int* d_addr;
    hipMalloc(&d_out, 64 * 4 * 2); hipMalloc(&d_addr, 64 * 4);
    // pattern 1: lane l -> elements 4l..4l+3 (lane-linear 8-byte pieces)
    for (int l = 0; l < 64; ++l) h_addr[l] = 4 * l;
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe_tr<<<1, 64>>>(d_addr, d_out);
    short h_out[256];
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("TR pattern lane-linear (addr = 4*lane elems):\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    // pattern 2: per-lane addresses into a [k][ld=160] image: p=lane&15, r=p>>2, q=p&3, g=lane>>4 : elem = r*160 + 16*(g&1) + 4*q + 1000*(g>>1)
    for (int l = 0; l < 64; ++l) { int p = l & 15, r = p >> 2, q = p & 3, g = l >> 4; h_addr[l] = r * 160 + 16 * (g & 1) + 4 * q + 1000 * (g >> 1); }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe_tr<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("TR pattern strided image:\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    // glds
    uint4 h_src[256], h_dst[256]; uint4 *d_src, *d_dst;
    for (int i = 0; i < 256; ++i) h_src[i] = make_uint4(i, i, i, i);
    hipMalloc(&d_src, sizeof(h_src)); hipMalloc(&d_dst, sizeof(h_dst));
    hipMemcpy(d_src, h_src, sizeof(h_src), hipMemcpyHostToDevice);
    probe_glds<<<1, 256>>>(d_src, d_dst);
    hipMemcpy(h_dst, d_dst, sizeof(h_dst), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) { int w = i / 64, l = i % 64; if ((int)h_dst[i].x != w * 64 + (l ^ 5)) bad++; }
    printf("glds: %s (lds[wave*64+lane] == src[wave*64 + (lane^5)]) first: %u %u %u\n", bad ? "MISMATCH" : "OK", h_dst[0].x, h_dst[1].x, h_dst[64].x);
    return 0;
}
