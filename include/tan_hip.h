/* tan_hip.h -- C ABI of libtan_hip.so: the MI355X (gfx950) kernels behind the TemporalAlignNet hot path.
 *
 * The reference (TengdaHan/TemporalAlignNet) is pure Python and has no FFI layer of its own: its seam is
 * the nn.Module surface of model/tan_model.py + train/loss.py:get_loss, and every arithmetic op below that
 * seam is a PyTorch ATen call.  Each entry point here replaces the ATen kernels behind one reference call
 * site (cited per function as reference file:line).  temporalalignnet_amd/_lib.py binds them with ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless stated otherwise.
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no allocation, no
 *    synchronisation, no global state) and is safe to capture into a hipGraph.
 *  - return 0 on success, a hipError_t value, or TAN_ERR_BAD_ARG (-1) for an invalid argument.
 *  - `dtype` selects the activation type: TAN_F32 (parity mode, exact-f32 MFMA) or TAN_BF16 (throughput
 *    mode, bf16 operands, f32 accumulation).  Statistics, logits, losses and parameter gradients are f32.
 *  - activations are batch-first, row = b*L + t, channels contiguous.
 */
#ifndef TAN_HIP_H
#define TAN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TAN_F32 0
#define TAN_BF16 1
#define TAN_ERR_BAD_ARG (-1)

#define TAN_ACT_NONE 0
#define TAN_ACT_QUICKGELU 1      /* C = x*sigmoid(1.702x), x = acc+bias; x itself optionally stored to aux */
#define TAN_ACT_QUICKGELU_GRAD 2 /* C = acc * d/dx quickgelu(x), x read from aux */

int tan_version(void);

/* ---- GEMM (nn.Linear fwd/bwd: tfm_model.py:21-27, tan_model.py:48-49,70; einsum tan_model.py:118,138) ----
 * C[M,N] (=|+=) alpha * opA(A)[M,K] * opB(B)[K,N]  (+ bias[N]) (activation) (+ residual[M,N])
 *   a_kc=1: A stored [M,K] (lda = row stride)   a_kc=0: A stored [K,M] (lda = stride between k)
 *   b_kc=1: B stored [N,K] ("x @ W^T")          b_kc=0: B stored [K,N]
 * Operands are `dtype`, C/residual/aux are `out_dtype`.  accumulate=1 (requires out_dtype F32, no
 * bias/act/residual) adds into C with f32 atomics and allows split_k > 1.  Batched over `batch` with
 * element strides sA/sB/sC (residual and aux share sC).                                                  */
typedef struct tan_gemm_desc {
    int dtype, out_dtype;
    int M, N, K;
    int a_kc, b_kc;
    const void* A; long lda;
    const void* B; long ldb;
    void* C; long ldc;
    const float* bias;
    const void* residual; long ldr;
    int act;
    void* aux; long ldaux;
    int accumulate;
    int split_k;
    float alpha;
    int batch; long sA, sB, sC;
} tan_gemm_desc;
int tan_gemm(const tan_gemm_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TAN_HIP_H */
