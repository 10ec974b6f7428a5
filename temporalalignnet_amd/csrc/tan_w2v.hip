// Sentence-embedder helpers for model/word2vec_model.py:Word2VecModel: frozen word-vector gather with
// K-padding for the MFMA GEMM, and the masked max-pool over the <=32 words of a sentence (forward + backward).  The two
// Linear layers are tan_gemm calls (fc1 with the ReLU epilogue).  All HBM-bound streaming kernels.
#include "tan_common.h"

namespace tal {

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const long* __restrict__ ids, const float* __restrict__ table, T* __restrict__ out,
                                                     long rows, int D, int Dpad, long V) {
    const int cpr = Dpad / 4;                       // float4 chunks per output row
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) * 4;
        long id = ids ? ids[r] : r;
        if (id < 0 || id >= V) id = 0;
        const float* src = table + id * D + c;
        float4 v = make_float4(0, 0, 0, 0);
        if (c + 4 <= D) v = make_float4(src[0], src[1], src[2], src[3]);        // rows of 300 floats are only 4-byte aligned
        else {
            if (c + 0 < D) v.x = src[0];
            if (c + 1 < D) v.y = src[1];
            if (c + 2 < D) v.z = src[2];
        }
        st4(out + r * Dpad + c, v);
    }
}

__global__ __launch_bounds__(256) void unpad_add_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int D,
                                                        int Dpad) {
    const long total = rows * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const int c = (int)(i - r * D);
        dst[i] += src[r * Dpad + c];
    }
}

// one thread per (sentence, 4 channels): walks the W words
template <typename T>
__global__ __launch_bounds__(256) void wordpool_fwd_kernel(const T* __restrict__ h, const unsigned char* __restrict__ mask,
                                                           T* __restrict__ pooled, int* __restrict__ argmax, long M, int W, int H) {
    const int cpr = H / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * cpr) return;
    const long m = i / cpr;
    const int c = (int)(i - m * cpr) * 4;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int4 bi = make_int4(0, 0, 0, 0);
    for (int w = 0; w < W; ++w) {
        float4 v = ld4(h + (m * W + w) * H + c);
        if (mask && !mask[m * W + w]) v = make_float4(-6e4f, -6e4f, -6e4f, -6e4f);
        if (v.x > best.x) { best.x = v.x; bi.x = w; }
        if (v.y > best.y) { best.y = v.y; bi.y = w; }
        if (v.z > best.z) { best.z = v.z; bi.z = w; }
        if (v.w > best.w) { best.w = v.w; bi.w = w; }
    }
    st4(pooled + m * H + c, best);
    *reinterpret_cast<int4*>(argmax + m * H + c) = bi;
}

// one thread per (sentence, word, 4 channels); the first word of each sentence also folds the bias gradient
template <typename T>
__global__ __launch_bounds__(256) void wordpool_bwd_kernel(const T* __restrict__ d_pooled, const T* __restrict__ pooled,
                                                           const int* __restrict__ argmax, T* __restrict__ dh, long M, int W, int H) {
    const int cpr = H / 4;
    const long total = M * W * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long mw = i / cpr;
        const int c = (int)(i - mw * cpr) * 4;
        const long m = mw / W;
        const int w = (int)(mw - m * W);
        const float4 g = ld4(d_pooled + m * H + c), p = ld4(pooled + m * H + c);
        const int4 a = *reinterpret_cast<const int4*>(argmax + m * H + c);
        st4(dh + mw * H + c, make_float4((a.x == w && p.x > 0.f) ? g.x : 0.f, (a.y == w && p.y > 0.f) ? g.y : 0.f,
                                         (a.z == w && p.z > 0.f) ? g.z : 0.f, (a.w == w && p.w > 0.f) ? g.w : 0.f));
    }
}

// db[j] += sum_m (pooled[m,j] > 0 ? d_pooled[m,j] : 0)
template <typename T>
__global__ __launch_bounds__(256) void wordpool_db_kernel(const T* __restrict__ d_pooled, const T* __restrict__ pooled,
                                                          float* __restrict__ db, long M, int H) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    const long m0 = (long)blockIdx.y * 64, m1 = min(M, m0 + 64);
    float s = 0.f;
    for (long m = m0; m < m1; ++m)
        if (ld_f(pooled + m * H + j) > 0.f) s += ld_f(d_pooled + m * H + j);
    unsafeAtomicAdd(db + j, s);
}

#define W2V_T(dtype, ...)                                             \
    if (dtype == TAN_F32) { typedef float T; __VA_ARGS__; }           \
    else if (dtype == TAN_BF16) { typedef bf16_t T; __VA_ARGS__; }    \
    else return TAN_ERR_BAD_ARG;

}  // namespace tal

using namespace tal;

extern "C" int tan_embed_gather(const long* ids, const float* table, void* out, long rows, int D, int Dpad, long V, int dtype,
                                void* stream) {
    TAN_REQUIRE(table && out && rows > 0 && D > 0 && Dpad >= D && Dpad % 4 == 0 && V > 0);
    const unsigned grid = (unsigned)min((long)4096, (long)cdiv(rows * (Dpad / 4), 256));
    W2V_T(dtype, hipLaunchKernelGGL((gather_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, table, (T*)out, rows, D,
                                    Dpad, V));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_unpad_add(const float* src, float* dst, long rows, int D, int Dpad, void* stream) {
    TAN_REQUIRE(src && dst && rows > 0 && D > 0 && Dpad >= D);
    const unsigned grid = (unsigned)min((long)2048, (long)cdiv(rows * D, 256));
    hipLaunchKernelGGL(unpad_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, rows, D, Dpad);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_wordpool_fwd(const void* h, const unsigned char* mask, void* pooled, int* argmax, long M, int W, int H, int dtype,
                                void* stream) {
    TAN_REQUIRE(h && pooled && argmax && M > 0 && W > 0 && H > 0 && H % 4 == 0);
    W2V_T(dtype, hipLaunchKernelGGL((wordpool_fwd_kernel<T>), dim3(cdiv(M * (H / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                                    (const T*)h, mask, (T*)pooled, argmax, M, W, H));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_wordpool_bwd(const void* d_pooled, const void* pooled, const int* argmax, void* dh, float* db, long M, int W, int H,
                                int dtype, void* stream) {
    TAN_REQUIRE(d_pooled && pooled && argmax && dh && M > 0 && W > 0 && H > 0 && H % 4 == 0);
    const unsigned grid = (unsigned)min((long)8192, (long)cdiv(M * W * (H / 4), 256));
    W2V_T(dtype, hipLaunchKernelGGL((wordpool_bwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)d_pooled,
                                    (const T*)pooled, argmax, (T*)dh, M, W, H));
    if (db) {
        W2V_T(dtype, hipLaunchKernelGGL((wordpool_db_kernel<T>), dim3(cdiv(H, 256), cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)d_pooled, (const T*)pooled, db, M, H));
    }
    TAN_LAUNCH_CHECK();
    return 0;
}
