// Fused AdamW + EMA target update + low-precision shadow weights over one flat parameter buffer (gfx950).
// Pure HBM streaming: 16 B/param read (p,g,m,v) + 12 B written, +8..10 B for the EMA twin and bf16 shadows.
//
// Arithmetic follows torch.optim.AdamW's single-tensor path step by step (train/main.py:397 uses it with
// the two parameter groups of optim_policy, main.py:330-356), then TwinTemporalAligner._momentum_update
// (model/tan_model.py:339-344) on the freshly updated online weights.
#include "tan_common.h"

namespace tal {

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
            if (md == 1) pi *= decay;
            float mi = m[i], vi = v[i];
            mi = mi + (gi - mi) * w1;
            vi = vi * beta2 + gi * gi * w2;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi = pi - step_size * (mi / denom);
            m[i] = mi; v[i] = vi; p[i] = pi;
            if (p_lowp) p_lowp[i] = f2bf(pi);
        }
        if (ema) {
            const float e = ema[i] * ema_m + pi * (1.0f - ema_m);
            ema[i] = e;
            if (ema_lowp) ema_lowp[i] = f2bf(e);
        }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n, float m,
                                                  bf16_t* __restrict__ tgt_lowp) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e = tgt[i] * m + src[i] * (1.0f - m);
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
    static const long max_grid = []() { const char* e = getenv("TAN_OPT_GRID"); return e ? atol(e) : 8192L; }();
    const unsigned grid = (unsigned)min(max_grid, (long)cdiv(n, 256));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, mode, n, decay, w1, (float)beta2, w2, (float)eps,
                       step_size, bc2_sqrt, grad_scale, (bf16_t*)p_bf16, ema, ema_m, (bf16_t*)ema_bf16);
    TAN_LAUNCH_CHECK();
    return 0;
}
