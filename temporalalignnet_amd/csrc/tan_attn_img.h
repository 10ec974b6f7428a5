// LDS "head images" of the short-sequence attention kernels (gfx950): [row][64] bf16 head slices with 128-byte rows whose 16-byte
// chunk index is XOR-swizzled by the row, read as MFMA operand fragments either row-wise (ds_read_b128) or transposed
// (ds_read_b64_tr_b16).  Shared by tan_attn.hip (one workgroup per (video, head)) and tan_attnblk.hip (the attention branch of a
// block in one launch: the in_proj GEMM writes the images, the out_proj GEMM reads O back from them).
#pragma once
#include "tan_mma.h"

namespace tal {

typedef const __attribute__((address_space(1))) void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;
typedef short attn_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int img_swz(int row) { const int t = (row >> 1) & 7; return t ^ ((t & 1) << 2); }

// lane: outer index = row, k = 8 * chunk .. + 7   (image rows are the outer index)
__device__ __forceinline__ bf16x8 img_frag_kc(const char* img, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((chunk ^ img_swz(row)) << 4));
}
// lane: outer index = col0 + (lane & 31), contraction slots e = 0..7 <-> image rows rbase + 4*(lane>>5) + (e&3) + 8*(e>>2)
// (the order in which a 32x32 accumulator tile holds its rows, see acc_frag)
__device__ __forceinline__ bf16x8 img_frag_tr(const char* img, int rbase, int col0, int lane) {
    const int g = lane >> 4, p = lane & 15, r = p >> 2, q = p & 3;
    const int row = rbase + 4 * (g >> 1) + r;
    const int chunk = (col0 + 16 * (g & 1) + 4 * q) >> 3;
    typedef __attribute__((address_space(3))) attn_s16x4* lds_v4;
    const char* p0 = img + row * 128 + ((chunk ^ img_swz(row)) << 4) + (q & 1) * 8;
    const char* p1 = img + (row + 8) * 128 + ((chunk ^ img_swz(row + 8)) << 4) + (q & 1) * 8;
    union { bf16x8 v; attn_s16x4 h[2]; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p1);
    return u.v;
}
// registers 8j..8j+7 of an accumulator tile as an operand fragment (contraction = the tile's row index)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int j) {
    union { bf16x8 v; bf16_t s[8]; } u;
#pragma unroll
    for (int e = 0; e < 8; ++e) u.s[e] = f2bf(a[8 * j + e]);
    return u.v;
}
// stage `nimg` [LP][64] head slices (image i from src + i * img_stride, row stride ld) into consecutive LDS images
template <int NKB>
__device__ __forceinline__ void stage_images(char* lds, int first_img, int nimg, const bf16_t* src, long img_stride, long ld,
                                             int L, int wave, int lane, int row0 = 0) {
    constexpr int PPI = NKB * 4;                    // 1-KiB pieces (8 rows) per image
    for (int p = wave; p < nimg * PPI; p += NKB) {
        const int img = p / PPI, row = (p % PPI) * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ img_swz(row);
        const bf16_t* g = src + img * img_stride + (long)min(row0 + row, L - 1) * ld + chunk * 8;
        __builtin_amdgcn_global_load_lds((attn_gptr_t)g, (attn_lptr_t)(lds + (first_img * PPI + p) * 1024), 16, 0, 0);
    }
}



// Accumulator tiles [d][row] of one wave (32 rows x 64 d) -> 32 wave-private rows of an LDS image -> 16-byte row-contiguous
// global stores (8 rows x 128 B per instruction) instead of 8-byte stores scattered over 32 rows.
__device__ __forceinline__ void tiles_to_rows(char* img, int r0, int c, int hh, const f32x16 (&t)[2], float scale) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int row = r0 + c, chunk = 4 * db + g4;
            st4((bf16_t*)(img + row * 128 + ((chunk ^ img_swz(row)) << 4) + hh * 8),
                make_float4(t[db][4 * g4] * scale, t[db][4 * g4 + 1] * scale, t[db][4 * g4 + 2] * scale, t[db][4 * g4 + 3] * scale));
        }
}
__device__ __forceinline__ void rows_to_global(const char* img, int r0, int lane, bf16_t* out, long ld, int grow0, int L) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = r0 + it * 8 + (lane >> 3), chunk = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(img + row * 128 + ((chunk ^ img_swz(row)) << 4));
        if (grow0 + row < L) *reinterpret_cast<uint4*>(out + (long)(grow0 + row) * ld + chunk * 8) = v;
    }
}

}  // namespace tal
