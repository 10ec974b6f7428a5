// Fused AdamW + EMA target update + low-precision shadow weights over one flat parameter buffer (gfx950).
// Pure HBM streaming: 16 B/param read (p,g,m,v) + 12 B written, +8..10 B for the EMA twin and bf16 shadows.
//
// Arithmetic follows torch.optim.AdamW's single-tensor path step by step (train/main.py:397 uses it with
// the two parameter groups of optim_policy, main.py:330-356), then TwinTemporalAligner._momentum_update
// (model/tan_model.py:339-344) on the freshly updated online weights.
#include "tan_common.h"

namespace tal {

constexpr int PN_WAVES_OPT = 8;      // waves per row-panel workgroup the packed formats are laid out for (tan_panel_waves())

// One element of torch.optim.AdamW's single-tensor update, separately rounded operations in torch's order (no FMA contraction: the
// plain kernel and the image-writing kernel below must produce bit-identical parameters whatever the compiler would fuse in each).
__device__ __forceinline__ void adamw_elem(float& pi, float gi, float& mi, float& vi, bool decays, float decay, float w1, float beta2,
                                           float w2, float eps, float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
    if (decays) pi = pi * decay;
    const float d1 = (gi - mi) * w1;
    mi = mi + d1;
    const float a = vi * beta2, b = (gi * gi) * w2;
    vi = a + b;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float upd = step_size * (mi / denom);
    pi = pi - upd;
}

__device__ __forceinline__ float ema_elem(float e, float p, float m) {      // tan_model.py:339-344, separately rounded
#pragma clang fp contract(off)
    const float a = e * m, b = p * (1.0f - m);
    return a + b;
}

// mode[i]: 0 = no weight decay, 1 = weight decay, 2 = parameter never receives a gradient (torch skips
// params whose .grad is None entirely -- no decay, no state), 3 = frozen for the optimizer but still EMA'd
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const unsigned char* __restrict__ mode, long n,
                                                    float decay, float w1, float beta2, float w2, float eps, float step_size,
                                                    float bc2_sqrt, float grad_scale, bf16_t* __restrict__ p_lowp, float* __restrict__ ema,
                                                    float ema_m, bf16_t* __restrict__ ema_lowp) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float pi = p[i];
        const unsigned char md = mode ? mode[i] : 1;
        if (md < 2) {
            const float gi = g[i] * grad_scale;
            float mi = m[i], vi = v[i];
            adamw_elem(pi, gi, mi, vi, md == 1, decay, w1, beta2, w2, eps, step_size, bc2_sqrt);
            m[i] = mi; v[i] = vi; p[i] = pi;
            if (p_lowp) p_lowp[i] = f2bf(pi);
        }
        if (ema) {
            const float e = ema_elem(ema[i], pi, ema_m);
            ema[i] = e;
            if (ema_lowp) ema_lowp[i] = f2bf(e);
        }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n, float m,
                                                  bf16_t* __restrict__ tgt_lowp) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e = ema_elem(tgt[i], src[i], m);
        tgt[i] = e;
        if (tgt_lowp) tgt_lowp[i] = f2bf(e);
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tan_ema_update(float* target, const float* online, long n, float m, void* target_bf16, void* stream) {
    TAN_REQUIRE(target && online && n > 0);
    const unsigned grid = (unsigned)min((long)8192, (long)cdiv(n, 256));
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, target, online, n, m, (bf16_t*)target_bf16);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* mode, long n, double lr,
                              double beta1, double beta2, double eps, double weight_decay, int step, float grad_scale,
                              void* p_bf16, float* ema, float ema_m, void* ema_bf16, void* stream) {
    TAN_REQUIRE(p && g && m && v && n > 0 && step >= 1);
    // scalar prefactors in double, as the Python side of torch.optim computes them
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    const unsigned grid = (unsigned)min(8192L, (long)cdiv(n, 256));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, mode, n, decay, w1, (float)beta2, w2, (float)eps,
                       step_size, bc2_sqrt, grad_scale, (bf16_t*)p_bf16, ema, ema_m, (bf16_t*)ema_bf16);
    TAN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// AdamW that also writes every weight IMAGE the bf16 kernels read (tan_adamw_step_images).  After the plain launch above, a step
// still had to rebuild the W^T copies (transpose_batch) and the fragment-major packed images of W and W^T (pack_tiles x 2) before
// the next forward could start: 140 us of kernels behind the 190 us optimizer launch, on the critical path of the step boundary.
// Here one wave owns a 32 x 32 block of a weight matrix: the updated bf16 values go to LDS once, row-major and transposed, and leave
// as whole 1-KiB fragments of each image -- the row-major shadow, the packed W (tan_pack_weights formats, incl. "qkv16"), the
// row-major W^T and the packed W^T; the EMA twin's shadow and packed W follow the same way.  The arithmetic is adamw_kernel's, in
// the same order (bit-identical parameters).
namespace tal {

struct ImgArgs {
    float* p; const float* g; float *m, *v; const unsigned char* mode;
    float decay, w1, beta2, w2, eps, step_size, bc2_sqrt, grad_scale;
    bf16_t* p_lowp; float* ema; float ema_m; bf16_t* ema_lowp;
    const tan_image_entry* table; const long* unit_prefix; int n_entries; long n_units;
    bf16_t *p_packed, *p_t, *p_tpacked, *ema_packed;
    long unit_begin;
};

// element offset of the 1-KiB fragment (32 rows x 16 cols at (row0, col0)) of a [rows][cols] matrix packed in tiles [TN][TK]
__device__ __forceinline__ long img_frag_off(int TN, int TK, int cols, int row0, int col0) {
    const int tiles_k = cols / TK, nb = row0 / TN, kt = col0 / TK;
    const int rt = row0 - nb * TN, ct = col0 - kt * TK, RW = TN / PN_WAVES_OPT;
    const int w = rt / RW, rb = (rt - w * RW) >> 5, RB = RW >> 5, KS = TK >> 4;
    return (long)(nb * tiles_k + kt) * TN * TK + (long)((w * RB + rb) * KS + (ct >> 4)) * 512;
}

__device__ __forceinline__ int img_f(int rho) { return (rho & 3) + 4 * (rho >> 3) + 16 * ((rho >> 2) & 1); }      // pack_tiles_kernel's row permutation

// One workgroup (4 waves) owns a 64 x 64 block of a matrix, wave w its 32 x 32 quarter (rows 32 (w >> 1).., columns 32 (w & 1)..).
// LDS: rm = the block row-major [64][64] bf16 (128 B per row), tr = its transpose [64 k][64 n].  Every global access is whole
// 128-byte lines: f32 tensors 8 rows x 128 B per wave-instruction, the row-major shadow / W^T 8 rows x 128 B, packed images 1 KiB.
constexpr int IMG_PITCH = 144;      // 128 B of data + 16: the transposing 2-byte writes of eight k rows land in eight different bank groups

// a wave's 32 x 32 quarter (rows q_r.., columns q_c.. of the block at (n0, k0)) -> the packed images
__device__ __forceinline__ void img_store_packed(const char* rm, const char* tr, const tan_image_entry& e, int n0, int k0, int q_r, int q_c,
                                                 int lane, bf16_t* packed, bf16_t* tpacked) {
    const int rho = lane & 31, hi = lane >> 5;
    if (packed && e.tn_w == 384) {            // "qkv16": fragments of 16 rows x 32 k, lane = (row & 15, k chunk)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n = n0 + q_r + 16 * g, which = n >> 9, hh = (n & 511) >> 6, fblk = (n & 63) >> 4;
            const int p = (hh & 1) * 12 + which * 4 + fblk;
            const long dst = e.off + (long)((hh >> 1) * (e.K >> 5) + ((k0 + q_c) >> 5)) * (384 * 32) + p * 512 + lane * 8;
            *reinterpret_cast<uint4*>(packed + dst) =
                *reinterpret_cast<const uint4*>(rm + (q_r + 16 * g + (lane & 15)) * IMG_PITCH + q_c * 2 + (lane >> 4) * 16);
        }
    } else if (packed && e.tn_w) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const long dst = e.off + img_frag_off(e.tn_w, e.tk_w, e.K, n0 + q_r, k0 + q_c + 16 * ks) + lane * 8;
            *reinterpret_cast<uint4*>(packed + dst) = *reinterpret_cast<const uint4*>(rm + (q_r + img_f(rho)) * IMG_PITCH + q_c * 2 + (2 * ks + hi) * 16);
        }
    }
    if (tpacked && e.tn_t) {                  // W^T is [K][N]: rows = k, contraction = n
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const long dst = e.off + img_frag_off(e.tn_t, e.tk_t, e.N, k0 + q_c, n0 + q_r + 16 * ks) + lane * 8;
            *reinterpret_cast<uint4*>(tpacked + dst) = *reinterpret_cast<const uint4*>(tr + (q_c + img_f(rho)) * IMG_PITCH + q_r * 2 + (2 * ks + hi) * 16);
        }
    }
}

// lane's 16 values x[4 j + c] = element (row q_r + 8 j + (lane >> 3), column q_c + 4 (lane & 7) + c) -> both LDS tiles
__device__ __forceinline__ void img_to_lds(char* rm, char* tr, const float (&x)[16], int q_r, int q_c, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = q_r + 8 * j + (lane >> 3), c = q_c + 4 * (lane & 7);
        uint2 w;
        w.x = f2bf2(x[4 * j], x[4 * j + 1]); w.y = f2bf2(x[4 * j + 2], x[4 * j + 3]);
        *reinterpret_cast<uint2*>(rm + r * IMG_PITCH + c * 2) = w;
        *reinterpret_cast<bf16_t*>(tr + (c + 0) * IMG_PITCH + r * 2) = (bf16_t)(w.x & 0xffffu);
        *reinterpret_cast<bf16_t*>(tr + (c + 1) * IMG_PITCH + r * 2) = (bf16_t)(w.x >> 16);
        *reinterpret_cast<bf16_t*>(tr + (c + 2) * IMG_PITCH + r * 2) = (bf16_t)(w.y & 0xffffu);
        *reinterpret_cast<bf16_t*>(tr + (c + 3) * IMG_PITCH + r * 2) = (bf16_t)(w.y >> 16);
    }
}

// the whole 64 x 64 tile in `src` (128 B per row) -> rows of a row-major bf16 matrix with leading dimension ld, whole lines
__device__ __forceinline__ void img_store_rows(const char* src, bf16_t* dst, long ld, int tid) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int id = t * 256 + tid, r = id >> 3, c = id & 7;
        *reinterpret_cast<uint4*>(dst + (long)r * ld + c * 8) = *reinterpret_cast<const uint4*>(src + r * IMG_PITCH + c * 16);
    }
}

__global__ __launch_bounds__(256) void adamw_images_kernel(const ImgArgs A) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * IMG_PITCH];
    char* rm = lds;
    char* tr = lds + 64 * IMG_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long u = A.unit_begin + blockIdx.x;
    int lo = 0, hi = A.n_entries;                         // entry whose unit range holds u (block-uniform binary search)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.unit_prefix[mid] <= u) lo = mid; else hi = mid; }
    const tan_image_entry e = A.table[lo];
    const long ul = u - A.unit_prefix[lo];
    const int kg = e.K >> 6;                              // 64-column blocks per row of blocks
    const int n0 = (int)(ul / kg) * 64, k0 = (int)(ul % kg) * 64;
    const int q_r = 32 * (wave >> 1), q_c = 32 * (wave & 1);
    // element (row q_r + 8 j + (lane >> 3), columns q_c + 4 (lane & 7) ..): a wave-instruction reads 8 rows x 128 B
    const long base = e.off + (long)(n0 + q_r + (lane >> 3)) * e.K + k0 + q_c + 4 * (lane & 7);
    const long jstride = 8L * e.K;
    const unsigned char md = A.mode[e.off];
    float pi[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float4 t = *reinterpret_cast<const float4*>(A.p + base + j * jstride); pi[4 * j] = t.x; pi[4 * j + 1] = t.y; pi[4 * j + 2] = t.z; pi[4 * j + 3] = t.w; }
    float ei[16];
    if (A.ema) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 t = *reinterpret_cast<const float4*>(A.ema + base + j * jstride); ei[4 * j] = t.x; ei[4 * j + 1] = t.y; ei[4 * j + 2] = t.z; ei[4 * j + 3] = t.w; }
    }
    if (md < 2) {
        float gi[16], mi[16], vi[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 tg = *reinterpret_cast<const float4*>(A.g + base + j * jstride), tm = *reinterpret_cast<const float4*>(A.m + base + j * jstride),
                         tv = *reinterpret_cast<const float4*>(A.v + base + j * jstride);
            gi[4 * j] = tg.x; gi[4 * j + 1] = tg.y; gi[4 * j + 2] = tg.z; gi[4 * j + 3] = tg.w;
            mi[4 * j] = tm.x; mi[4 * j + 1] = tm.y; mi[4 * j + 2] = tm.z; mi[4 * j + 3] = tm.w;
            vi[4 * j] = tv.x; vi[4 * j + 1] = tv.y; vi[4 * j + 2] = tv.z; vi[4 * j + 3] = tv.w;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            adamw_elem(pi[i], gi[i] * A.grad_scale, mi[i], vi[i], md == 1, A.decay, A.w1, A.beta2, A.w2, A.eps, A.step_size, A.bc2_sqrt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(A.p + base + j * jstride) = make_float4(pi[4 * j], pi[4 * j + 1], pi[4 * j + 2], pi[4 * j + 3]);
            *reinterpret_cast<float4*>(A.m + base + j * jstride) = make_float4(mi[4 * j], mi[4 * j + 1], mi[4 * j + 2], mi[4 * j + 3]);
            *reinterpret_cast<float4*>(A.v + base + j * jstride) = make_float4(vi[4 * j], vi[4 * j + 1], vi[4 * j + 2], vi[4 * j + 3]);
        }
    }
    // (a skipped / frozen matrix still gets its images rewritten from the current values: they are the current images)
    img_to_lds(rm, tr, pi, q_r, q_c, lane);
    __syncthreads();
    const long blk = e.off + (long)n0 * e.K + k0;
    if (A.p_lowp) img_store_rows(rm, A.p_lowp + blk, e.K, tid);
    if (A.p_t) img_store_rows(tr, A.p_t + e.off + (long)k0 * e.N + n0, e.N, tid);
    img_store_packed(rm, tr, e, n0, k0, q_r, q_c, lane, A.p_packed, A.p_tpacked);
    if (A.ema) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ei[i] = ema_elem(ei[i], pi[i], A.ema_m);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(A.ema + base + j * jstride) = make_float4(ei[4 * j], ei[4 * j + 1], ei[4 * j + 2], ei[4 * j + 3]);
        __syncthreads();
        img_to_lds(rm, tr, ei, q_r, q_c, lane);
        __syncthreads();
        if (A.ema_lowp) img_store_rows(rm, A.ema_lowp + blk, e.K, tid);
        img_store_packed(rm, tr, e, n0, k0, q_r, q_c, lane, A.ema_packed, nullptr);
    }
}

// adamw_kernel for everything the image kernel does not own: the elements listed in idx (ascending, runs of consecutive indices)
__global__ __launch_bounds__(256) void adamw_rest_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, const unsigned char* __restrict__ mode,
                                                         const int* __restrict__ idx, long n,
                                                         float decay, float w1, float beta2, float w2, float eps, float step_size,
                                                         float bc2_sqrt, float grad_scale, bf16_t* __restrict__ p_lowp, float* __restrict__ ema,
                                                         float ema_m, bf16_t* __restrict__ ema_lowp) {
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long)gridDim.x * 256) {
        const long i = idx[j];
        const unsigned char md = mode[i];
        float pi = p[i];
        if (md < 2) {
            const float gi = g[i] * grad_scale;
            float mi = m[i], vi = v[i];
            adamw_elem(pi, gi, mi, vi, md == 1, decay, w1, beta2, w2, eps, step_size, bc2_sqrt);
            m[i] = mi; v[i] = vi; p[i] = pi;
            if (p_lowp) p_lowp[i] = f2bf(pi);
        }
        if (ema) {
            const float e = ema_elem(ema[i], pi, ema_m);
            ema[i] = e;
            if (ema_lowp) ema_lowp[i] = f2bf(e);
        }
    }
}

}  // namespace tal

extern "C" int tan_adamw_step_images(const tan_adamw_images_desc* d, void* stream) {
    TAN_REQUIRE(d && d->p && d->g && d->m && d->v && d->mode && d->n > 0 && d->step >= 1);
    TAN_REQUIRE(d->table && d->unit_prefix && d->n_entries > 0 && d->n_units > 0 && (d->n_rest == 0 || d->rest_idx));
    TAN_REQUIRE(tan_panel_waves() == PN_WAVES_OPT);
    const double bc1 = 1.0 - pow(d->beta1, (double)d->step), bc2 = 1.0 - pow(d->beta2, (double)d->step);
    ImgArgs A{};
    A.p = d->p; A.g = d->g; A.m = d->m; A.v = d->v; A.mode = d->mode;
    A.decay = (float)(1.0 - d->lr * d->weight_decay); A.w1 = (float)(1.0 - d->beta1); A.beta2 = (float)d->beta2; A.w2 = (float)(1.0 - d->beta2);
    A.eps = (float)d->eps; A.step_size = (float)(d->lr / bc1); A.bc2_sqrt = (float)sqrt(bc2); A.grad_scale = d->grad_scale;
    A.p_lowp = (bf16_t*)d->p_bf16; A.ema = d->ema; A.ema_m = d->ema_m; A.ema_lowp = (bf16_t*)d->ema_bf16;
    A.table = d->table; A.unit_prefix = d->unit_prefix; A.n_entries = d->n_entries; A.n_units = d->n_units;
    A.p_packed = (bf16_t*)d->p_packed; A.p_t = (bf16_t*)d->p_t; A.p_tpacked = (bf16_t*)d->p_tpacked; A.ema_packed = (bf16_t*)d->ema_packed;
    hipStream_t st = (hipStream_t)stream;
    // one 64 x 64 block per workgroup; [unit_begin, unit_end) of the table's units (unit_end == 0: all of them)
    const long u1 = d->unit_end > 0 ? d->unit_end : d->n_units;
    TAN_REQUIRE(d->unit_begin >= 0 && d->unit_begin <= u1 && u1 <= d->n_units);
    A.unit_begin = d->unit_begin;
    if (u1 > d->unit_begin) {
        hipLaunchKernelGGL(adamw_images_kernel, dim3((unsigned)(u1 - d->unit_begin)), dim3(256), 0, st, A);
        TAN_LAUNCH_CHECK();
    }
    if (d->n_rest <= 0) return 0;
    const unsigned grid2 = (unsigned)min((long)8192, (long)cdiv(d->n_rest, 256));
    hipLaunchKernelGGL(adamw_rest_kernel, dim3(grid2), dim3(256), 0, st, d->p, d->g, d->m, d->v, d->mode, d->rest_idx, d->n_rest, A.decay, A.w1, A.beta2, A.w2,
                       A.eps, A.step_size, A.bc2_sqrt, A.grad_scale, A.p_lowp, A.ema, A.ema_m, A.ema_lowp);
    TAN_LAUNCH_CHECK();
    return 0;
}
