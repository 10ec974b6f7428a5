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
#define TAN_ACT_RELU 3           /* C = max(acc+bias, 0)  (Word2VecModel fc1, model/word2vec_model.py:86) */

int tan_version(void);
/* sizeof(tan_gemm_desc | tan_layer_params | tan_layer_bufs | tan_encoder_desc | tan_simfam_desc) for which = 0..4 (binding self-check) */
int tan_abi_sizeof(int which);

/* Optional in-stream kernel timer (bench.py's `roofline` line): while enabled every tan_gemm / tan_attn_* launch is
 * bracketed by hipEvents on its own stream.  tan_prof_collect synchronises and returns per kind the summed duration
 * [ms], the summed ALGORITHMIC work [flop] (2*M*N*K per GEMM; 4*B*H*L*L*64 attention fwd, 14*... bwd incl. recompute)
 * and the launch count; returns 1 if the record buffer overflowed.  tan_prof_enable: on = 1 starts a fresh recording,
 * on = 0 pauses it (records are kept), on = 2 resumes -- bench.py samples every 4th timed step this way.  GEMM kinds: base + 2*(A K-strided) + (B K-strided). */
#define TAN_PROF_GEMM_BF16 0
#define TAN_PROF_GEMM_F32 4
#define TAN_PROF_ATTN_FWD 8
#define TAN_PROF_ATTN_BWD 9
#define TAN_PROF_SIMNCE 10
#define TAN_PROF_PANEL 11   /* row-panel fused MLP kernels (tan_mlp_fwd / tan_mlp_bwd) */
#define TAN_PROF_ATTNBLK 12 /* the attention branch of a block in one launch (tan_attnblk_*) */
#define TAN_PROF_NKINDS 13
int tan_prof_enable(int on, int max_records);
/* time only every stride-th eligible launch from now on (1 = all): a sample of the launches instead of each of them */
int tan_prof_stride(int stride);
/* work [flop] and launch count of EVERY eligible launch since tan_prof_enable(1, ..) / the last call (timed or not); resets them */
int tan_prof_collect_all(double* work_by_kind, long* count_by_kind, int nkinds);
int tan_prof_collect(double* ms_by_kind, double* work_by_kind, long* count_by_kind, int nkinds);

/* Stream-ordering events (no timing) for tan_encoder_desc.layer_done: the reference's DDP (end2end/main_nce.py:142-158) hooks
 * autograd to start reducing a gradient bucket while backward continues; here one call runs a whole stack's backward, so the
 * library records an event per layer and the caller makes its communication stream wait for it. */
int tan_event_create(void** event);
int tan_event_destroy(void* event);
int tan_stream_wait_event(void* stream, void* event);

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
    float* colsum; /* optional [N] f32: += column sums of the stored output (fused bias gradient); batch must be 1 */
} tan_gemm_desc;
int tan_gemm(const tan_gemm_desc* d, void* stream);

/* ---- LayerNorm(C), eps, affine; one wave per row; C in {256,512,1024} -------------------------------------
 * fwd: y = (x-mean)*rstd*gamma+beta (+ add[row % add_period], the broadcast position term of
 *      tan_model.py:167,199).  reference: tfm_model.py:22,28,35,37; tan_model.py:50-54,155,174,206.
 *      mean/rstd [rows] f32 are saved for backward (may be NULL).
 * bwd: dx = (dres +) LN'(dy); dgamma/dbeta (f32, may be NULL) are ACCUMULATED (+=); dx_colsum (may be NULL) += column
 *      sums of the OUTPUT dx, i.e. the bias gradient of the Linear that wrote the stream this LN normalises (saves a
 *      separate pass).  ws: f32 scratch of tan_layernorm_bwd_ws_floats(C) elements.                           */
int tan_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      const void* add, int add_period, long rows, int C, float eps, int dtype, void* stream);
long tan_layernorm_bwd_ws_floats(int C);
int tan_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, float* dx_colsum, float* ws, long rows, int C,
                      int dtype, void* stream);

/* ---- L2 normalisation over channels, no epsilon (tan_model.py:116-117,136-137) ---------------------------
 * Output rows r = 0..rows-1 are gathered from x row (r/grp)*src_grp_rows + src_off + r%grp, which extracts the
 * video rows (off 0, grp T) or the text rows (off T, grp N) of a joint [B, T+N, C] stage output; bwd scatters
 * dx = (dy - y<y,dy>) * inv_norm back to the same rows (plain store).                                        */
int tan_l2norm_fwd(const void* x, void* y, float* inv_norm, long rows, int C, int grp, int src_grp_rows, int src_off,
                   int dtype, void* stream);
int tan_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, long rows, int C, int grp,
                   int dst_grp_rows, int dst_off, int dtype, void* stream);
/* The same for up to 8 deep-supervision stages in ONE launch (tan_model.py:116-117,136-137 run per stage of the [B,S,T,C]
 * features): stage s reads / scatters to its own buffer xs->p[s] / dxs->p[s]; y, dy [nstage][rows][C] and inv_norm
 * [nstage][rows] are stage-major contiguous.                                                                             */
typedef struct tan_ptr8 { const void* p[8]; } tan_ptr8;
int tan_l2norm_fwd_multi(const tan_ptr8* xs, void* y, float* inv_norm, int nstage, long rows, int C, int grp,
                         int src_grp_rows, int src_off, int dtype, void* stream);
int tan_l2norm_bwd_multi(const void* dy, const void* y, const float* inv_norm, const tan_ptr8* dxs, int nstage, long rows,
                         int C, int grp, int dst_grp_rows, int dst_off, int dtype, void* stream);

/* ---- small HBM-bound helpers ----------------------------------------------------------------------------- */
/* out[c] += sum_r x[r][c]  (nn.Linear bias gradients) */
int tan_colsum_acc(const void* x, float* out, long rows, int C, int dtype, void* stream);
/* dst[(g*dst_grp_rows+dst_off+r)][c] (=|+=) src[(g*src_grp_rows+src_off+r)][c], g<G, r<R  (torch.cat at
 * tan_model.py:201 and its backward split) */
int tan_rows_copy(const void* src, void* dst, int G, int R, int C, long src_grp_rows, long src_off, long dst_grp_rows,
                  long dst_off, int accumulate, int dtype, void* stream);
/* dst[s][m][:] = map[m] >= 0 ? src[s][map[m]][:] : 0, s < S  (src [S, Msrc, C], dst [S, Mdst, C], map int32 [Mdst]): the compacted
 * text-feature gradient of the logits-free NCE back in the padded [B*N] row order -- torch.zeros().index_copy_() in one launch */
int tan_rows_gather(const void* src, void* dst, const int* map, int S, long Msrc, long Mdst, int C, int dtype, void* stream);
/* out[r][c] = sum_g x[g*R + r][c]  (backward of the broadcast position add) */
int tan_group_sum(const void* x, void* out, int G, int R, int C, int dtype, void* stream);
int tan_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* y = x * sigmoid(1.702 x): QuickGELU.forward called on its own (model/tfm_model.py:11-13); in a block it is a GEMM epilogue */
int tan_quickgelu(const void* x, void* y, long n, int dtype, void* stream);
/* out[i] += sum_s parts[s*n + i]  (folds split-K partial tiles into an f32 gradient); n % 4 == 0 */
int tan_reduce_add(const float* parts, float* out, int nparts, long n, void* stream);
/* Batched bf16 transpose inside one flat buffer: matrix i is src[table[3i] ...] with shape [table[3i+1], table[3i+2]] (row-major;
 * both multiples of 8, offsets multiples of 8); dst gets its transpose at the same element offset.  `table` is DEVICE memory.
 * Produces the K-contiguous weight copies tan_layer_params.wt_* (dX = dY W reads W^T rows) once per optimizer step.        */
int tan_transpose_batch(const void* src, void* dst, const long* table, int n, long max_rows, long max_cols, int dtype,
                        void* stream);
/* binary_head = nn.Linear(512,1) (tan_model.py:70,147-148): out[r] = <x[r],w> + b (f32 out); bwd accumulates dw, db */
int tan_head_fwd(const void* x, const float* w, const float* b, float* out, long rows, int C, int dtype, void* stream);
int tan_head_bwd(const float* dout, const void* x, const float* w, void* dx, float* dw, float* db, long rows, int C,
                 int accumulate_dx, int dtype, void* stream);
/* F.interpolate(mode='linear', align_corners=False) of a [L_in,C] f32 table (tan_model.py:157-160,189-192); bwd adds */
int tan_interp_linear(const float* src, float* dst, int L_in, int L_out, int C, void* stream);
int tan_interp_linear_bwd(const float* ddst, float* dsrc, int L_in, int L_out, int C, void* stream);

/* ---- attention core: softmax_k(q k^T / sqrt(64) + key_padding) v per (video, head) -------------------------
 * replaces the scaled-dot-product inside nn.MultiheadAttention (tfm_model.py:21,30-32); head dim 64, C = 64*H.
 * qkv [B*L, 3C] (output of the packed in-proj GEMM, q|k|v), key_padding_mask [B,L] bytes (1 = ignore) or NULL,
 * o [B*L, C], lse [B,H,L] f32 (saved for backward).  bwd recomputes P from lse and writes dqkv [B*L, 3C].
 * L <= 448 (f32) / 320 (bf16): the score panel of a 64-query tile lives in the 160 KB LDS.                      */
int tan_attn_fwd(const void* qkv, const unsigned char* key_padding_mask, void* o, float* lse, int B, int L, int H,
                 int dtype, void* stream);
int tan_attn_bwd(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                 const void* d_o, void* dqkv, int B, int L, int H, int dtype, void* stream);
/* the same, and g_b_qkv [3C] f32 += column sums of dqkv (the in_proj bias gradient of nn.MultiheadAttention): summed in
 * the backward kernel from the rows it is about to store when L <= 128 (bf16), by tan_colsum_acc over dqkv otherwise    */
int tan_attn_bwd_bias(const void* qkv, const unsigned char* key_padding_mask, const void* o, const float* lse,
                      const void* d_o, void* dqkv, float* g_b_qkv, int B, int L, int H, int dtype, void* stream);

/* ---- loss side (train/loss.py:get_loss) -------------------------------------------------------------------
 * logits: raw cosines, stage-major [S, R=B*T, Mp=B*N] f32 (the reference's [B,S,T,B,N] is a permuted view).
 * tgt [B,T,N] f32 {0,1}: same-video targets (loss.py:73-76 or the self-labelled ones, loss.py:227);
 * col_invalid [Mp] bytes: padded texts (dropped columns, loss.py:233,241); row_leak [R] bytes or NULL: frames whose
 * same-video logits read -6e4 (reference in-place quirk for model='init' + learn_agreement, loss.py:96-101).
 *
 * tan_nce_fwd: symmetric multi-positive NCE terms of loss.py:240-253 (and 262-274):
 *   v_terms[s,r] = LSE_cols(all) - LSE_cols(pos), t_terms[s,c] = LSE_rows(all) - LSE_rows(pos); the row/column sums of
 *   exp(l/0.07 - 1/0.07) are saved (rowsum [S,R], colsum [S,Mp], possum_v [S,R], possum_t [S,Mp]) for tan_nce_bwd, which
 *   turns upstream g_v [S,R], g_t [S,Mp] into dlogits [S,R,Mp] (out_dtype).  ws: tan_nce_ws_floats() f32 scratch.   */
long tan_nce_ws_floats(int S, int B, int T, int N);
int tan_nce_fwd(const float* logits, const float* tgt, const unsigned char* col_invalid, const unsigned char* row_leak,
                float* rowsum, float* colsum, float* possum_v, float* possum_t, float* v_terms, float* t_terms, float* ws,
                int S, int B, int T, int N, int n_valid_cols, void* stream);
int tan_nce_bwd(const float* logits, const float* tgt, const unsigned char* col_invalid, const unsigned char* row_leak,
                const float* rowsum, const float* colsum, const float* possum_v, const float* possum_t, const float* g_v,
                const float* g_t, void* dlogits, int out_dtype, int S, int B, int T, int N, void* stream);
/* `blocks`: the last-stage same-video logit blocks; element (b,t,n) at blocks[b*block_stride + t*row_stride + n].  For
 * materialised stage-major logits: blocks = logits + (S-1)*R*Mp, block_stride = T*Mp + N, row_stride = Mp; for a compact
 * [B,T,N] tensor: T*N and N.
 * self-labelling scan, loss.py:88-143 (joint) / 146-179 (dual): two-way softmax (over texts, /0.07, over time) of the
 * last-stage same-video logits, sliding-window mean with window length dur[b,n] (0 = padded text), first-index argmax.
 * Outputs max_pos [B,N] int32, max_prob/max_logit [B,N] f32, self_tgt [B,N,T] bytes (the chosen window).            */
int tan_selflabel(const float* blocks, long block_stride, long row_stride, const unsigned char* video_pad,
                  const unsigned char* text_pad, const float* dur, int* max_pos, float* max_prob, float* max_logit,
                  unsigned char* self_tgt, int B, int T, int N, void* stream);
/* out[b,n] = max_t block(b,t,n) / 0.07   (loss.py:280,283) */
int tan_diag_max(const float* blocks, long block_stride, long row_stride, const unsigned char* row_leak, float* out, int B, int T,
                 int N, void* stream);
/* IoU of the two self-labelled windows, confidence, target policy kind (0 'i', 1 'u', 2 'keep', 3 'keep-joint') and the
 * per-frame first-text de-duplication with restore, loss.py:181-226.  q_joint/q_dual: device scalars (0.3-quantiles of
 * the max logits).  tgt_out [B,T,N] f32, iou [B,N] f32, conf [B,N] bytes.  N <= 64.                                 */
int tan_agreement(const unsigned char* joint_tgt, const unsigned char* dual_tgt, const unsigned char* youtube_tgt,
                  const float* max_logit_joint, const float* max_logit_dual, const float* q_joint, const float* q_dual,
                  int kind, float* tgt_out, float* iou, unsigned char* conf, int B, int T, int N, void* stream);
/* torch.quantile(x[~invalid], q) ('linear' interpolation, at::lerp rounding) without a host sync; n <= 8192 */
int tan_masked_quantile(const float* x, const unsigned char* invalid, int n, float q, float* out, void* stream);

/* ---- logits-free similarity + NCE (bf16 features) ---------------------------------------------------------------
 * Fused replacement of einsum (tan_model.py:118,138) + NCE terms (loss.py:240-253): vn [S,R,C], tn [S or 1,Mp,C] unit
 * features (t_stage_stride = Mp*C, or 0 when the text features are shared by all stages), other arguments as tan_nce_fwd.
 * One workgroup sweeps a 128-row panel of one stage over all text columns; logits only ever exist as MFMA accumulators.
 * tan_simnce_bwd_dl recomputes them and writes d loss/d logits [S,R,Mp] in bf16 for the two follow-up tan_gemm calls.
 * ws: tan_simnce_ws_floats() f32 scratch.  Requires C % 64 == 0, B*N <= 2048.
 * Column compaction (optional, colmap != NULL): padded text columns take part in nothing (loss.py:64-70 drops them before
 * the log-sum-exps), so the sweep may run on a COMPACTED text matrix: then `tn` holds Mc <= B*N rows per stage (a multiple of 64 keeps
 * the follow-up GEMMs on the direct-to-LDS kernel; filler rows flagged in col_invalid), col_invalid / colsum / possum_t / t_terms / g_t have Mc entries
 * per stage, dl is [S,R,Mc]; `tn_blocks` (stage stride tb_stage_stride) is the UNcompacted [B*N,C] matrix, read only for the
 * same-video blocks, and colmap[b*N+k] is that sentence's compacted column or -1.
 * `phases` (0 = everything) selects the steps, for sweeps over SEVERAL column blocks (global negatives across ranks, row f3):
 *   TAN_SIM_SWEEP  the tile sweep (fwd: row sums += valid columns, column sums of this block -> colsum; bwd: d logits)
 *   TAN_SIM_ACC_ROWS  with SWEEP: keep accumulating into rowsum instead of zeroing it first
 *   TAN_SIM_DIAG   same-video blocks: positives + padded-frame quirk (only the block that holds the rows' own sentences)
 *   TAN_SIM_TERMS  (fwd) v_terms / t_terms from the final sums
 *   TAN_SIM_DIAG_KEEP  (bwd, with DIAG) `ws` still holds the same-video blocks tan_simnce_fwd computed for these features: do
 *                  not recompute them                                                                                      */
#define TAN_SIM_SWEEP 1
#define TAN_SIM_DIAG 2
#define TAN_SIM_TERMS 4
#define TAN_SIM_ACC_ROWS 8
#define TAN_SIM_DIAG_KEEP 16
#define TAN_SIM_CORR_KEEP 32 /* (tan_simnce_bwd_dl_dvn_kept) `ws` already holds the correction array of these upstream gradients */
long tan_simnce_ws_floats(int S, int B, int T, int N);
int tan_simnce_max_cols(void);   /* most text columns (B*N, or Mc when compacted) one sweep accepts: callers fall back to tan_nce_* above it */
int tan_simnce_fwd(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                   const unsigned char* row_leak, float* rowsum, float* colsum, float* possum_v, float* possum_t,
                   float* v_terms, float* t_terms, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                   long tb_stage_stride, const int* colmap, int Mc, int phases, void* stream);
int tan_simnce_bwd_dl(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                      const unsigned char* row_leak, const float* rowsum, const float* colsum, const float* possum_v,
                      const float* possum_t, const float* g_v, const float* g_t, void* dl, float* ws, int S, int B, int T, int N,
                      int C, const void* tn_blocks, long tb_stage_stride, const int* colmap, int Mc, int phases, void* stream);

/* The same pair with the exponentials KEPT: tan_simnce_fwd_keep also stores e = exp((cos - 1)/tau) of every (frame, sentence) pair of
 * the sweep as bf16, S * ceil(R/128) * ceil(Mp/128) tiles of 128 x 128 in the sweep's accumulator order (Mp = Mc when compacted:
 * tan_simnce_keep_elems() elements); tan_simnce_bwd_dl_kept turns them into d loss/d logits with one element-wise
 * pass (2 x S*R*Mp*2 bytes of HBM traffic) instead of a second 2*S*R*Mp*C-FLOP sweep.  tan_simnce_keeps(C) != 0 says whether the
 * pair is available for C channels (the LDS-resident sweep: C = 512); all other arguments as above.                          */
int tan_simnce_keeps(int C);
long tan_simnce_keep_elems(int S, int R, int Mp);
int tan_simnce_fwd_keep(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                        const unsigned char* row_leak, float* rowsum, float* colsum, float* possum_v, float* possum_t,
                        float* v_terms, float* t_terms, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                        long tb_stage_stride, const int* colmap, int Mc, int phases, void* e_keep, void* stream);
int tan_simnce_bwd_dl_kept(const void* e_keep, const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                           const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum, const float* colsum,
                           const float* possum_v, const float* possum_t, const float* g_v, const float* g_t, void* dl, float* ws,
                           int S, int B, int T, int N, int C, const void* tn_blocks, long tb_stage_stride, const int* colmap, int Mc,
                           int phases, void* stream);
/* tan_simnce_bwd_dl_kept that also returns d_vn [S, R, C] (bf16) = dl . t_hat, the gradient of the unit video features
 * (the autograd of tan_model.py:116-119,136-139's einsum towards its first operand): the 128 x 128 d-logits tiles are the MFMA operand
 * while they are in the LDS, so the [S*R, Mp] x [Mp, C] GEMM behind the element-wise pass and its read of the d-logits go away.  dl is
 * still written (the text-feature gradient contracts it over the rows; dl = NULL: not written).  C = 512 (tan_simnce_keeps), sweep
 * columns % 8 == 0 and < 32768, N <= 32; overwrites the sweep's text image inside `ws` and (unless phases has TAN_SIM_CORR_KEEP)
 * builds the dense same-video correction array there.                                                                                                          */
int tan_simnce_bwd_dl_dvn_kept(const void* e_keep, const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                               const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum, const float* colsum,
                               const float* possum_v, const float* possum_t, const float* g_v, const float* g_t, void* dl, void* d_vn,
                               float* ws, int S, int B, int T, int N, int C, const void* tn_blocks, long tb_stage_stride,
                               const int* colmap, int Mc, int phases, void* stream);
/* ---- one feature FAMILY (dual or joint) of the logits-free NCE, from the stacks' stage outputs to their gradients --------------
 * Everything between an encoder stack's forward and its backward in the training step, for one family of tan_model.py:116-119
 * (dual: video stack stages x the one text embedding) or :136-139 (joint: video rows x text rows of the joint stack's stages), with
 * loss.py:240-253 and its autograd:
 *   tan_simfam_fwd  L2-normalise the stage rows (video: one launch, or inside the sweep's panel load with TAN_SIMFAM_NORM_IN_SWEEP;
 *                   text: normalise + column compaction + both fragment-major text images in ONE launch), the statistics sweep keeping
 *                   its exponentials (tan_simnce_fwd_keep's kernel), and ONE finishing launch: same-video cosine blocks, column sums
 *                   over the row panels, positives / leaked frames, v_terms / t_terms -- and, when g_v / g_t are already known (the
 *                   two-chain step: they depend on the batch's masks only), the same-video corrections of the backward.
 *   tan_simfam_bwd  d logits + d v_hat in one pass over the kept exponentials whose epilogue applies the L2-normalisation's backward
 *                   and stores straight into the stacks' stage-gradient rows (d v_hat never exists in HBM); the text-feature gradient
 *                   GEMM d t_hat = dl^T v_hat (f32, split-K); one launch that gathers it back to the padded sentence order, applies
 *                   the normalisation's backward and stores the text stage-gradient rows.
 * 6-7 launches per family and step instead of 18.  bf16, C = 512, N <= 32, Mc % 8 == 0, Mc <= tan_simnce_max_cols(), S <= 8.
 * Row addressing: video row r = b*T + t of stage s lives at x_video.p[s] + ((r / T) * v_grp_rows + v_off + r % T) * C (video stack:
 * v_grp_rows = T; joint stack: T + N); padded sentence m = b*N + k at x_text.p[st] + ((m / N) * t_grp_rows + t_off + m % N) * C;
 * d_video / d_text are addressed the same way.  St = 1: one text embedding for all stages (dual), St = S: per stage (joint).
 * Column compaction as tan_simnce_fwd: idx [Mc] (sweep column -> padded sentence), colmap [B*N] (or both NULL: Mc = B*N, identity).
 * Saved between the two calls (caller-owned): vn [S,R,C], inv_v [S,R], tn [St,Mc,C], inv_t [St,Mc], rowsum / possum_v [S,R],
 * colsum / possum_t [S,Mc], e_keep (tan_simnce_keep_elems), ws (tan_simfam_ws_bytes).                                              */
#define TAN_SIMFAM_NORM_IN_SWEEP 1   /* flags: the sweep normalises its frame panel itself and writes vn / inv_v (no separate launch) */
#define TAN_SIMFAM_CORR_DONE 2       /* (set by tan_simfam_fwd when g_v / g_t were given) ws holds the corrections: bwd skips that launch */
/* tan_simfam_fwd in two calls, for a loss whose TARGETS are not known when the stack's forward ends (stage-2 co-training,
 * train/loss.py:88-229: the agreement targets come from the EMA model): SWEEP_ONLY = the normalisations, the text launch and the statistics
 * sweep (none of them reads tgt / row_leak / g_v / g_t); FINISH_ONLY = the finishing launch (same-video blocks, positives, terms), after
 * a SWEEP_ONLY call on the same descriptor.  Neither flag: both, as before.  After FINISH_ONLY the last stage's same-video cosines
 * [B, T, N] f32 (padded-sentence order; what train/loss.py:280-283 takes its per-sentence maxima from) are at tan_simfam_diag(). */
#define TAN_SIMFAM_SWEEP_ONLY 4
#define TAN_SIMFAM_FINISH_ONLY 8
/* d_tn_acc is already ZERO when tan_simfam_bwd is called (the caller cleared it off the critical chain): the corrections launch does
 * not clear it (16 MB for the joint family at B = 128: 86 us in front of the one-pass kernel when it is the first thing after the
 * stage-2 step's meeting point) */
#define TAN_SIMFAM_ACC_ZEROED 16
typedef struct tan_simfam_desc {
    int S, St, B, T, N, C, Mc, flags;
    tan_ptr8 x_video; long v_grp_rows, v_off;
    tan_ptr8 x_text; long t_grp_rows, t_off;
    const long* idx; const int* colmap; const unsigned char* col_invalid;   /* [Mc] | [B*N] | [Mc] pad flags of the sweep's columns */
    const float* tgt; const unsigned char* row_leak;                        /* [B,T,N] f32 | [B*T] or NULL */
    void* vn; float* inv_v; void* tn; float* inv_t;
    float *rowsum, *colsum, *possum_v, *possum_t;
    void* e_keep; void* ws;
    float *v_terms, *t_terms;                                               /* out: [S,R], [S,Mc] */
    const float *g_v, *g_t;                                                 /* d loss / d terms: [S,R], [S,Mc] (fwd: optional) */
    void* dl; float* d_tn_acc;                                              /* bwd scratch: [S,R,Mc] bf16 + 256 elements of slack, [St,Mc,C] f32 */
    tan_ptr8 d_video; tan_ptr8 d_text;                                      /* out: stage-gradient rows (addressed like x_*) */
    int dtn_split_k;   /* K slices of the text-gradient GEMM; 0: default (St = 1: 8 slices of tan_gemm; St = S: 2 of tan_gemm_atb), < 0: -n slices of tan_gemm_atb */
} tan_simfam_desc;
long tan_simfam_ws_bytes(int S, int St, int B, int T, int N, int Mc);
int tan_simfam_fwd(tan_simfam_desc* d, void* stream);
/* byte offset inside `ws` of stage s's same-video cosine blocks [B, T, N] f32 written by the finishing launch */
long tan_simfam_diag_offset(int S, int St, int B, int T, int N, int Mc, int s);
int tan_simfam_bwd(tan_simfam_desc* d, void* stream);

/* NCE tail (loss.py:236-237,254-275).  tan_pos_masks: rows_pos[b*T+t] = 1 if frame t of video b has a positive among its
 * unpadded sentences, cols_pos[b*N+k] = 1 if sentence k is unpadded and has a positive frame (tgt [B,T,N] f32, text_pad [B,N]).
 * tan_nce_tail_fwd: out2[0] = (mean(v_d | rows_mask) + mean(t_d | cols_mask)) / 2, out2[1] the same for the joint terms and
 * out2[2] = (out2[0] + out2[1]) / 2 (THREE floats; tan_nce_tail_bwd takes d loss / d each of them, any may be NULL = 0),
 * mean(x | m) = sum_{s,k} x[s,k] m[k] / (S sum m)  (NaN for an empty mask, like .mean() of nothing); counts2 = mask sums, kept
 * for tan_nce_tail_bwd, which turns d loss/d out2 into the gradients of the four term tensors.  One launch each.
 * counts_in (optional, [2]): divide by these (GLOBAL) mask sums instead of the local ones -- global negatives, row f3.     */
int tan_pos_masks(const float* tgt, const unsigned char* text_pad, float* rows_pos, float* cols_pos, int B, int T, int N, void* stream);
/* Stage-2 glue of get_loss (train/loss.py:280-290,309-328,345) in one launch, B*N <= 8192: from md / mj [B*N] (tan_diag_max of the dual /
 * joint last-stage same-video logits): metric = -(zscore(md) + zscore(mj)) over the real sentences, th = quantile(metric, q_th),
 * th_mask (bytes) / th_f = (metric <= th) & real, rows_pos_th [B*T] = rows that own a positive among the kept real sentences of tgt
 * [B,T,N]; with use_align: lab [B*N] (1 / 0 / 2 = ignore; 0 where the sentence's centre abs_text_pos.mean(-1) is < 0.2 or > 0.8;
 * NaN on padded sentences), sel = (lab != 2) & real, y = lab * sel; scal8 = {n_sel, n_pos, pos_weight = n_sel / n_pos - 1,
 * confidence ratio = mean(conf | real) (conf bytes or NULL), n_real, th, median(md), median(mj)}.
 * tan_bce_sel_fwd: out2 = {sum(BCEWithLogits(x, y, pos_weight) * sel) / n_sel, sum(((x > 0) == y) * sel) / n_sel};
 * tan_bce_sel_bwd: dx = g[0] * d bce / dx * sel / n_sel.                                                                           */
int tan_stage2_masks(const float* md, const float* mj, const unsigned char* text_pad, const float* tgt, const float* abs_text_pos,
                     const unsigned char* conf, float q_th, int use_align, int B, int T, int N, float* metric, unsigned char* th_mask,
                     float* th_f, float* rows_pos_th, float* lab, float* sel, float* y, float* scal8, void* stream);
int tan_bce_sel_fwd(const float* x, const float* y, const float* sel, const float* scal8, int n, float* out2, void* stream);
int tan_bce_sel_bwd(const float* x, const float* y, const float* sel, const float* scal8, const float* g, int n, float* dx, void* stream);
/* Everything get_loss derives from the batch's masks alone (train/loss.py:58-70: pad masks, the [B,T,N] start/end target) in one launch,
 * plus the text-column compaction of the logits-free sweeps: text_pad as f32 0/1 (train/main.py:62-65) OR as bytes (exactly one non-NULL),
 * video_pad bytes [B,T], tgt_raw bytes [B,N,T] (get_mask_from_time) -> tpad_u8 / valid (bytes) / valid_f (f32) [B*N], vpad_u8 [B*T],
 * tgt f32 [B,T,N]; with idx != NULL also idx [Mc] (int64: real sentences in order, then padded columns in order = a stable sort of the
 * pad flags), colmap [B*N] (int32 rank among the real sentences, -1 = padded), ci_run [Mc] (pad flags of the compacted columns).      */
int tan_loss_prep(const float* text_pad_f32, const unsigned char* text_pad_u8, const unsigned char* video_pad_u8,
                  const unsigned char* tgt_raw, unsigned char* tpad_u8, unsigned char* vpad_u8, unsigned char* valid, float* valid_f,
                  float* tgt, long* idx, int* colmap, unsigned char* ci_run, int B, int T, int N, int Mc, void* stream);
int tan_nce_tail_fwd(const float* v_d, const float* t_d, const float* v_j, const float* t_j, const float* rows_mask,
                     const float* cols_mask, int Sd, int Sj, long R, long M, float* out2, float* counts2, const float* counts_in,
                     void* stream);
int tan_nce_tail_bwd(const float* g_dual, const float* g_joint, const float* g_mean, const float* rows_mask, const float* cols_mask,
                     const float* counts2, int Sd, int Sj, long R, long M, float* g_v_d, float* g_t_d, float* g_v_j, float* g_t_j,
                     void* stream);

/* ---- sentence embedder (model/word2vec_model.py:76-102) ----------------------------------------
 * tan_embed_gather: out[r, 0:D] = table[ids[r], :] cast to `dtype`, out[r, D:Dpad] = 0 (ids NULL = identity: a padded
 *   cast of a [rows, D] f32 matrix).  Pads the 300-d word vectors / fc1 weight rows to a multiple of 64 for the MFMA GEMM.
 * tan_unpad_add: dst[r, 0:D] += src[r, 0:D] for src [rows, Dpad] f32 (the padded dW1 back into the 300-wide gradient).
 * tan_wordpool_fwd: pooled[m,j] = max_w (mask[m,w] ? h[m,w,j] : -6e4), argmax[m,j] = first maximising w
 *   (masked_fill + torch.max, word2vec_model.py:94-96); tan_wordpool_bwd: dh[m,w,j] = (w == argmax && pooled > 0) ? d_pooled : 0,
 *   i.e. the max routing followed by the in-place ReLU's gradient; db1 (optional, f32 [H]) += column sums of dh.            */
int tan_embed_gather(const long* ids, const float* table, void* out, long rows, int D, int Dpad, long V, int dtype, void* stream);
int tan_unpad_add(const float* src, float* dst, long rows, int D, int Dpad, void* stream);
int tan_wordpool_fwd(const void* h, const unsigned char* mask, void* pooled, int* argmax, long M, int W, int H, int dtype,
                     void* stream);
int tan_wordpool_bwd(const void* d_pooled, const void* pooled, const int* argmax, void* dh, float* db, long M, int W, int H,
                     int dtype, void* stream);

/* ---- fused AdamW (+ EMA twin, + bf16 shadow weights) over one flat f32 parameter buffer ---------------------
 * torch.optim.AdamW single-tensor arithmetic (train/main.py:397, groups of main.py:330-356) followed by
 * TwinTemporalAligner._momentum_update (tan_model.py:339-344).  mode[i]: 0 no decay, 1 decay, 2 parameter never
 * receives a gradient (skipped like a .grad-is-None parameter); NULL = all decay.  step >= 1 is the 1-based count. */
int tan_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* mode, long n, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int step, float grad_scale, void* p_bf16,
                   float* ema, float ema_m, void* ema_bf16, void* stream);
/* tan_adamw_step plus every weight IMAGE the bf16 kernels read, in the same pass: for each [N][K] matrix of `table` (DEVICE memory;
 * off = element offset in the flat buffers, shared by all images) the updated values are also written as the row-major W^T (p_t), the
 * tan_pack_weights image of W in tiles [tn_w][tk_w] (p_packed; tn_w = 384 selects "qkv16", 0 = none) and of W^T in tiles [tn_t][tk_t]
 * (p_tpacked), and the EMA twin's packed W (ema_packed); any image pointer may be NULL.  unit_prefix (DEVICE, [n_entries + 1]) =
 * running sum of N/64 * K/64 over the table, n_units its last element; N % 64 == 0, K % 64 == 0.  mode (required) as in
 * tan_adamw_step; rest_idx (DEVICE, int32 [n_rest], ascending) lists every element of the flat buffer OUTSIDE the table's matrices:
 * those are updated by the plain kernel in a second launch.  Replaces tan_adamw_step + tan_transpose_batch + 2 x tan_pack_weights of a training step (train/main.py:112-122). */
typedef struct tan_image_entry { long off; int N, K, tn_w, tk_w, tn_t, tk_t; } tan_image_entry;
typedef struct tan_adamw_images_desc {
    float* p; const float* g; float *m, *v; const unsigned char* mode; long n;
    double lr, beta1, beta2, eps, weight_decay; int step; float grad_scale;
    void* p_bf16; float* ema; float ema_m; void* ema_bf16;
    const tan_image_entry* table; const long* unit_prefix; int n_entries; long n_units;
    void *p_packed, *p_t, *p_tpacked, *ema_packed;
    const int* rest_idx; long n_rest;
    long unit_begin, unit_end;   /* only the table's units [unit_begin, unit_end) (unit_end == 0: to n_units): a step may update the matrices whose
                                    gradients are final early -- e.g. one stack's while the other's backward still runs -- and the rest,
                                    with rest_idx, in a second call */
} tan_adamw_images_desc;
int tan_adamw_step_images(const tan_adamw_images_desc* d, void* stream);
/* target = m*target + (1-m)*online  (TwinTemporalAligner._momentum_update, tan_model.py:339-344) */
int tan_ema_update(float* target, const float* online, long n, float m, void* target_bf16, void* stream);

/* ---- one TemporalEncoder stack (tfm_model.py:41-55), forward and backward in one call each -----------------
 * Weights are `dtype` (f32, or the bf16 shadow copies); biases / LayerNorm affine / all gradients are f32 and
 * gradients are ACCUMULATED.  Layouts: w_qkv [3C,C] (in_proj_weight, q|k|v), w_out [C,C], w_fc [4C,C], w_proj [C,4C]. */
typedef struct tan_layer_params {
    const void *w_qkv, *w_out, *w_fc, *w_proj;
    const float *b_qkv, *b_out, *b_fc, *b_proj;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float *g_w_qkv, *g_w_out, *g_w_fc, *g_w_proj;
    float *g_b_qkv, *g_b_out, *g_b_fc, *g_b_proj;
    float *g_ln1_g, *g_ln1_b, *g_ln2_g, *g_ln2_b;
    const void *wt_qkv, *wt_out, *wt_fc, *wt_proj; /* optional (bf16): W^T copies, [in, out] row-major, for the dX GEMMs; NULL = read W K-strided */
    const void *wp_qkv, *wp_out, *wp_fc, *wp_proj; /* optional (bf16): tan_pack_weights images of w_* for the row-panel kernels; NULL = unfused path */
    const void *wtp_qkv, *wtp_out, *wtp_fc, *wtp_proj; /* optional (bf16): tan_pack_weights images of wt_* (backward row-panel kernels) */
} tan_layer_params;

/* per-layer saved activations, rows R = B*L */
typedef struct tan_layer_bufs {
    void *xn1, *qkv, *attn_o, *x_mid, *xn2, *h_pre, *h_act, *x_out; /* dtype: [R,C] [R,3C] [R,C] [R,C] [R,C] [R,4C] [R,4C] [R,C] */
    float *mean1, *rstd1, *mean2, *rstd2;                            /* [R] */
    float* lse;                                                      /* [B,H,L] */
} tan_layer_bufs;

typedef struct tan_encoder_desc {
    int dtype, B, L, C, H, layers;
    const unsigned char* key_padding_mask; /* [B,L] bytes, 1 = ignore, or NULL */
    const void* x0;                        /* [R,C] stack input */
    const tan_layer_params* params;        /* HOST array [layers] of device pointers */
    const tan_layer_bufs* bufs;            /* HOST array [layers] */
    const float *post_g, *post_b;          /* ln_video_post_enc / ln_joint_post_enc (tan_model.py:174,206) */
    float *g_post_g, *g_post_b;
    void* post_out;                        /* [R,C] last stage = LN_post(x_out[last]); NULL to skip */
    float *post_mean, *post_rstd;
    /* backward only */
    void *scr_dx, *scr_dx2, *scr_do, *scr_dxn; /* [R,C] */
    void* scr_dh;                               /* [R,4C] */
    void* scr_dqkv;                             /* [R,3C] */
    float* ln_ws;                               /* tan_layernorm_bwd_ws_floats(C) */
    float* dw_ws;                               /* split-K partial tiles for the dW GEMMs, or NULL (then f32 atomics) */
    long dw_ws_floats;                          /* >= 32 * 4*C*C to cover every layer shape */
    const void* const* d_stage;                 /* HOST array [layers]: grad w.r.t. stage s ([R,C] dtype) or NULL */
    void* d_x0;                                 /* [R,C] out: grad w.r.t. x0 */
    void* const* layer_done;                    /* HOST array [layers] of tan_event handles or NULL: layer_done[i] is recorded on
                                                   the stream once every gradient of layer i's parameters is final (layers finish
                                                   last to first) -- lets a data-parallel caller start reducing a layer's slice of
                                                   the flat gradient while the earlier layers' backward still runs */
    /* forward only */
    int no_save;                                /* != 0: no backward will follow (torch.no_grad(): the EMA target's forward,
                                                   tan_model.py:348-351, and every evaluation entry point) -- the tensors that exist
                                                   only for backward (h_pre, h_act, xn2, mean2 / rstd2, and on the fused attention
                                                   path qkv, attn_o, lse) are NOT written; bufs[] may then leave them NULL.  The stage
                                                   outputs (xn1 of layers >= 1, post_out) and x_mid / x_out are written as usual. */
    /* forward only: != 0: bufs[0].xn1 / mean1 / rstd1 already hold the first block's ln_1(x0) (tan_embed_fwd wrote them) */
    int xn1_ready;
    /* backward only, optional: the grouped weight-gradient launches of blocks dw_tail - 1 .. 0 -- the LAST blocks of a stack's backward;
     * they only feed the optimizer -- go to dw_stream (made to wait for the stack's stream first), so that the stack's remaining dX kernels
     * and whatever the caller queues behind tan_encoder_bwd (the embeddings' backward) do not wait for them.  Block 0 needs nothing else
     * (nothing overwrites its operands any more); dw_tail > 1 needs the second set of the four scratch buffers such a launch reads
     * (scr2_*): the tail blocks alternate between the sets, and a block waits for the launch two blocks above it before it reuses
     * a set.  The CALLER joins dw_stream before it reads those weight gradients or reuses the scratch buffers. */
    void* dw_stream; int dw_tail;
    void *scr2_dx, *scr2_dx2, *scr2_dh, *scr2_dqkv;
    /* optional, both directions: scratch [8, R, C] f32 -- with it, stacks of few row panels (R / 64 <= TAN_SPLIT_PANELS, default 48)
     * run the MLP branch through tan_mlp_fwd_split / tan_mlp_bwd_split */
    float* split_part;
} tan_encoder_desc;
int tan_encoder_fwd(const tan_encoder_desc* e, void* stream);
int tan_encoder_bwd(const tan_encoder_desc* e, void* stream);

/* Weight gradient of a Linear layer (what autograd produces for nn.Linear.weight inside model/tfm_model.py's blocks):
 * gw[N,K] (f32) += dy[M,N]^T x[M,K].  The long M contraction is cut into slices; with a workspace `ws` (>= slices*N*K floats,
 * slices <= 32) they are written as partial tiles and folded by tan_reduce_add, otherwise accumulated with f32 atomics. */
int tan_linear_wgrad(const void* dy, const void* x, float* gw, long M, int N, int K, float* ws, long ws_floats, int dtype,
                     void* stream);
/* C_p [M_p, N_p] (=|+=) A_p^T B_p for p < n <= 8 problems sharing the contraction length K (a multiple of 128): A_p [K, lda_p] and
 * B_p [K, N_p] row-major bf16 -- the contraction runs over ROWS.  256 x 256 tiles (N_p % 256 == 0; M_p ragged: the last tile of A_p's
 * columns reads up to 255 elements past each row, so the caller pads the END of A_p's buffer by 512 bytes).  out_dtype TAN_BF16: plain
 * store (split == 1, accumulate == 0); TAN_F32: split == 1 stores or (accumulate) adds in place, split > 1 K slices add with f32
 * atomics (accumulate must be set; C zeroed by the caller).  This is the kernel behind tan_linear_wgrad_group (the weight gradients
 * dW = dY^T X of a block, autograd of model/tfm_model.py:21-27) with its full shape range exposed. */
int tan_gemm_atb(int n, const void* const* A, const void* const* B, void* const* C, const int* lda, const int* M, const int* N, long K,
                 int out_dtype, int accumulate, int split, void* stream);
/* The weight gradients of up to four Linear layers over the same M rows (the in_proj, out_proj, c_fc and c_proj of one
 * ResidualAttentionBlock, tfm_model.py:17-27) in ONE launch: gw[i][N[i],K[i]] += dy[i][M,N[i]]^T x[i][M,K[i]].  This is the call
 * tan_encoder_bwd makes per block; not eligible groups (f32, ragged M) run tan_linear_wgrad one by one.               */
int tan_linear_wgrad_group(int n, const void* const* dy, const void* const* x, float* const* gw, const int* N, const int* K,
                           long M, float* ws, long ws_floats, int dtype, void* stream);

/* ---- row-panel fused kernels: one launch per half block ("fused layer") ------------------------
 * A workgroup owns a 64-row panel of the residual stream and keeps it in LDS while the weights stream past it from L2; the
 * weights are PRE-PACKED (once per optimizer step) into the LDS image of the tiles the kernels consume, in consumption order.
 * tan_pack_weights: entry i packs the row-major bf16 matrix src[src_off ..] of shape [N][K] (the nn.Linear layout, or a W^T
 *   copy) into dst[dst_off ..] as tiles [TN][TK] in (n-block, k-block) order; TN*TK*2 bytes must be 16384, TK in {16,32,64}
 *   (kernels here: TN=256,TK=32 for N > 512, TN=512,TK=16 for N == 512; TN=384,TK=32 with N=1536 selects the "qkv16" format of
 *   tan_attnblk_fwd: 24-KiB tiles of 16 x 32 fragments, see there).  `table` is DEVICE memory; max_tiles = the largest
 *   N/TN * K/TK of the table. */
typedef struct tan_pack_entry { long src_off, dst_off; int N, K, TN, TK; } tan_pack_entry;
int tan_panel_waves(void);   /* waves per row-panel workgroup the library was built for (fragment ownership inside a packed tile) */
int tan_pack_weights(const void* src, void* dst, const tan_pack_entry* table, int n, int max_tiles, void* stream);

/* tan_mlp_fwd: the MLP half of ResidualAttentionBlock_Step.forward (model/tfm_model.py:23-27,37) for rows % 64 == 0, bf16:
 *   xn2 = LN2(x_mid) ; h = QuickGELU(xn2 W_fc^T + b_fc) ; x_out = x_mid + h W_proj^T + b_proj ; xn_next = LN_next(x_out)
 * xn2 / mean2 / rstd2 / h_pre / h_act are saved for backward (each group may be NULL: not stored -- h_pre and h_act together,
 * mean2 and rstd2 together, xn2 alone; the no-grad forward passes none of them); xn_next (optional) is the
 * NEXT block's ln_1 output (= this block's deep-supervision feature, tfm_model.py:48-55) or the stack's post-LN.
 * pw_fc = packed c_fc.weight [2048][512] (TN=256,TK=32), pw_proj = packed c_proj.weight [512][2048] (TN=512,TK=16).      */
typedef struct tan_mlp_desc {
    long rows; int C, FF;                    /* 512, 2048 */
    const void* x_mid;                       /* [rows, C] bf16 */
    const float *ln_g, *ln_b;                /* ln_2 */
    const void *pw_fc, *pw_proj;
    const float *b_fc, *b_proj;
    void* xn2; float *mean2, *rstd2;         /* out */
    void *h_pre, *h_act;                     /* out [rows, FF] bf16 or NULL */
    void* x_out;                             /* out [rows, C] */
    const float *nln_g, *nln_b;              /* optional fused LayerNorm of x_out */
    void* xn_next; float *nmean, *nrstd;
    float eps;
    int variant;                             /* 0; (1: stream only, 2: no streaming -- timing experiments, results undefined) */
    /* Optional head (pw_out != NULL): x_mid is not read but WRITTEN first, x_mid = x_in + attn_o W_out^T + b_out -- the attention
     * out-projection + bias + residual of the block (tfm_model.py:30-36: what follows the attention core) for shapes the one-launch
     * attention branch (tan_attnblk_fwd) does not take (L > 80): attn_o [rows, C] bf16 = tan_attn_fwd's output, pw_out =
     * tan_pack_weights image of out_proj.weight [C][C] (TN = 512, TK = 16), b_out f32 [C], x_in [rows, C] bf16 the block's input. */
    const void* attn_o; const void* pw_out; const float* b_out; const void* x_in;
} tan_mlp_desc;
int tan_mlp_fwd(const tan_mlp_desc* d, void* stream);

/* Backward of the same branch in ONE launch (replaces, in tan_encoder_bwd: the c_proj dX GEMM with its quickgelu' epilogue and
 * bias column sums, the c_fc dX GEMM, and the ln_2 backward kernel; reference: autograd of model/tfm_model.py:23-27,37,44):
 *   dh  = (dx W_proj) o quickgelu'(h_pre)                  -> dh [rows, FF] bf16 (operand of the c_fc weight gradient)
 *   dx2 = dx + LayerNorm-backward(dh W_fc; x_mid, mean2, rstd2, ln_g)      -> dx2 [rows, C] bf16
 *   g_b_fc += colsum(dh), g_ln_g += colsum(dxn o xhat), g_ln_b += colsum(dxn), g_b_out += colsum(dx2)     (f32 atomics)
 * pwt_proj = tan_pack_weights image of W_proj^T [2048][512] (TN=256,TK=32), pwt_fc = of W_fc^T [512][2048] (TN=512,TK=16).
 * rows % 64 == 0, C = 512, FF = 2048, bf16 only.                                                                             */
typedef struct tan_mlp_bwd_desc {
    long rows; int C, FF;
    const void* dx;                          /* [rows, C] gradient w.r.t. the block's output (also the residual gradient); see ln1_dxn */
    const void* h_pre;                       /* [rows, FF] pre-activation saved by the forward */
    const void* x_mid;                       /* [rows, C] the branch's input (ln_2 input) */
    const float *mean2, *rstd2, *ln_g;
    const void *pwt_proj, *pwt_fc;
    void* dh;                                /* out [rows, FF] */
    void* dx2;                               /* out [rows, C] */
    float *g_b_fc, *g_ln_g, *g_ln_b, *g_b_out;   /* accumulated */
    /* Optional (ln1_dxn != NULL): `dx` is not read but PRODUCED first, as the backward of the NEXT block's ln_1 (tfm_model.py:43):
     *   dx = ln1_res + LayerNorm-backward(ln1_dxn; ln1_x, ln1_mean, ln1_rstd, ln1_g)      -> dx_out [rows, C] bf16 (and the panel)
     *   g_ln1_g += colsum(ln1_dxn o xhat), g_ln1_b += colsum(ln1_dxn), g_dx_colsum += colsum(dx)  (= this block's c_proj bias grad)
     * ln1_res may alias dx2 (a workgroup reads its 64 rows before it writes them) or be NULL (no residual: the stack's post-LayerNorm). */
    const void *ln1_dxn, *ln1_x, *ln1_res;
    const float *ln1_mean, *ln1_rstd, *ln1_g;
    float *g_ln1_g, *g_ln1_b, *g_dx_colsum;
    void* dx_out;
    /* Optional tail (pwt_out != NULL): d_o [rows, C] bf16 = dx2 W_out -- the dX GEMM of the block's attention out-projection
     * (tfm_model.py:21: attn.out_proj), whose input gradient dx2 is sitting in this kernel's LDS panel; pwt_out = tan_pack_weights
     * image of out_proj.weight^T with TN = 512, TK = 16.  Saves the 8192 x 512 x 512 launch of tan_encoder_bwd. */
    const void* pwt_out;
    void* d_o;
    /* Optional head (pwt_in != NULL; needs the ln_1 fields above, ln1_dxn = NULL): ln1_dxn is PRODUCED first, in LDS only:
     *   ln1_dxn = dqkv W_in (+ dstage)  -- the dX GEMM of the NEXT block's attention in-projection (tfm_model.py:21: attn.in_proj) with
     * dqkv [rows, 3C] bf16 (tan_attn_bwd's output), pwt_in = tan_pack_weights image of in_proj_weight^T [C][3C] (TN = 512, TK = 16),
     * dstage [rows, C] bf16 or NULL = the deep-supervision gradient that joins at that ln_1 output.  Saves the 8192 x 1536 x 512
     * launch per block of tan_encoder_bwd and the [rows, C] round trip of its result. */
    const void* dqkv;
    const void* pwt_in;
    const void* dstage;
} tan_mlp_bwd_desc;
int tan_mlp_bwd(const tan_mlp_bwd_desc* d, void* stream);

/* Small batches (round 6): the same two branches with the HIDDEN dimension split over tan_mlp_split_chunks() = 8 workgroups per 64-row
 * panel.  A whole-panel workgroup streams all of its block's weights (4 MiB) whatever the batch, so 16 panels take as long as 160; here
 * workgroup (panel, chunk) runs one 256-wide hidden chunk (the same prologue, c_fc / c_proj phases, chunk epilogue and side outputs) and
 * stores its [64 x 512] f32 term as plane `chunk` of `part` [8, rows, C] (scratch: nothing is expected in it, nothing is left in it
 * that matters); a second launch adds the eight planes in chunk order and applies the row epilogue -- forward: + b_proj + residual ->
 * x_out and the optional next LayerNorm (tfm_model.py:37,43); backward: LayerNorm-2 backward + residual -> dx2 and the column-sum
 * parameter gradients.  Same descriptors and results (to the order of eight f32 additions) as tan_mlp_fwd / tan_mlp_bwd; the optional
 * head / tail / ln_1-prologue fields must be NULL (tan_encoder_* issues those pieces as the launches they were before they were folded
 * in).                                                                                                                               */
int tan_mlp_split_chunks(void);
int tan_mlp_fwd_split(const tan_mlp_desc* d, float* part, void* stream);
int tan_mlp_bwd_split(const tan_mlp_bwd_desc* d, float* part, void* stream);

/* ---- input embeddings in ONE launch (bf16 throughput mode, C = 512) ------------------------------------------------------------
 * tan_embed_fwd: per problem (modality), rows r = (video v, position t) with t = r % T:
 *   proj = a W^T;  y = LayerNorm(proj; ln_g, ln_b);  out[d][(v*out_grp_rows[d] + out_off[d] + t)] = y + pos[d][t]   (d = 0, 1)
 *   xn1[d] = LayerNorm(out[d]; ln1_g[d], ln1_b[d])  with mean1 / rstd1 at the same row index        (optional)
 * replaces model/tan_model.py:155-167 + 187-203 (video_pre_proj, ln_video_init, ln_position_init(pos[p0:p0+T]) for the dual and the
 * joint offset, the torch.cat into the joint stack's input) resp. 231-234 / 212-228 (text) and the first block's ln_1 of the stack
 * that consumes out[d] (model/tfm_model.py:35).  a: [rows, K] f32 or bf16 (a_dtype); pw: tan_pack_weights image of W [512][K] with
 * TN = 512, TK = 16; pos[d]: f32 [T, 512] rows ALREADY normalised by ln_position_init (or NULL); a_bf16 (optional, when a is f32):
 * the bf16 copy of a the weight-gradient GEMM of the backward reads; proj / mean / rstd: saved for the LayerNorm backward.
 * pad_dst (optional): pad_dst[v*pad_grp_rows + pad_off + t] = pad_src[r] (or 0): the rows' key-padding flags in the joint [B, L] mask.
 * Up to two problems (video, text) share one launch. */
typedef struct tan_embed_desc {
    const void* a; int a_dtype; long rows; int K, T, C;
    const void* pw; const float *ln_g, *ln_b;
    void* a_bf16; void* proj; float *mean, *rstd;
    void* out[2]; long out_grp_rows[2], out_off[2]; const float* pos[2];
    const float *ln1_g[2], *ln1_b[2]; void* xn1[2]; float *mean1[2], *rstd1[2];
    const unsigned char* pad_src; unsigned char* pad_dst; long pad_grp_rows, pad_off;
} tan_embed_desc;
int tan_embed_fwd(const tan_embed_desc* d, int nprob, void* stream);

/* tan_embed_bwd: backward of tan_embed_fwd's LayerNorm + position add for up to two problems in one launch (autograd of
 * model/tan_model.py:155-167, 187-199, 212-234): dy = d_out[0] + d_out[1] (either may be NULL; rows addressed like tan_embed_fwd's out[]),
 *   d_proj [rows, C] bf16 = LayerNorm-backward(dy; proj, mean, rstd, ln_g)   (operand of the pre-projection's weight gradient)
 *   g_ln_g += colsum(dy o xhat), g_ln_b += colsum(dy);  d_pos[d] [ceil(videos / tan_embed_bwd_group())][T, C] f32 = per video GROUP,
 *   the sum over its videos of d_out[d] (plain stores, every plane written in full; NULL = not wanted; tan_pos_ln_bwd adds the planes)
 * tan_pos_ln_bwd: ln_position_init's backward for up to three used table slices in one launch: g_table rows += LayerNorm-backward(sum of
 *   the nparts planes d_pos [nparts][n][C];
 *   x = the raw table rows, mean, rstd, gamma) (g_table NULL: a fixed sine table), g_gamma += colsum(d_pos o xhat), g_beta += colsum(d_pos);
 *   f32 atomics (the slices of the dual and the joint offset overlap). */
typedef struct tan_embed_bwd_desc {
    long rows; int T, C;
    const void* d_out[2]; long d_out_grp_rows[2], d_out_off[2];
    const void* proj; const float *mean, *rstd, *ln_g;
    void* d_proj; float *g_ln_g, *g_ln_b; float* d_pos[2];
} tan_embed_bwd_desc;
int tan_embed_bwd(const tan_embed_bwd_desc* d, int nprob, void* stream);
typedef struct tan_pos_ln_bwd_use { const float *d_pos, *x, *mean, *rstd; float* g_table; int n, nparts; } tan_pos_ln_bwd_use;
int tan_embed_bwd_group(void);   /* videos per partial plane of tan_embed_bwd's d_pos */
int tan_pos_ln_bwd(const tan_pos_ln_bwd_use* u, int nuse, const float* gamma, float* g_gamma, float* g_beta, int C, void* stream);

/* ---- the attention branch of a block in ONE launch per direction (bf16, C = 512, H = 8, 48 < L <= 80 rows per video) --------
 * tan_attnblk_fwd: x_mid = x_in + out_proj(MHA(xn1))  with MHA = nn.MultiheadAttention(512, 8) as called at
 * model/tfm_model.py:30-36 (packed in_proj q|k|v, q scaled by 64^-0.5, key_padding_mask keys = -inf, softmax over keys, dropout 0,
 * need_weights=False).  One workgroup per video keeps the xn1 panel in LDS; replaces the in_proj GEMM, tan_attn_fwd and the
 * out_proj GEMM (+bias +residual) of tan_encoder_fwd, and the qkv / attn_o round trips between them.
 *   pw_qkv = tan_pack_weights image of attn.in_proj_weight [1536][512] with TN = 384, TK = 32 (the "qkv16" format: tile (head
 *            pair hp, k step) holds, per wave w and feature block fb < 3, the 16 x 32 fragment of in_proj rows
 *            which*512 + (2hp + j)*64 + fblk*16 .. +15 where p = 3w + fb, which = (p / 4) % 3, j = p / 12, fblk = p % 4)
 *   pw_out = tan_pack_weights image of attn.out_proj.weight [512][512] with TN = 512, TK = 16
 * qkv [B*L, 3C], attn_o [B*L, C], lse [B, H, L] are what tan_attn_bwd needs later: all three or none (NULL: the no-grad forward
 * writes nothing but x_mid).  Fully padded key rows give zeros (lse = -inf), like tan_attn_fwd.                              */
typedef struct tan_attnblk_desc {
    int B, L, C, H;
    const void* xn1;                         /* [B*L, C] bf16: ln_1 output */
    const void* x_in;                        /* [B*L, C] bf16: the residual stream entering the block */
    const unsigned char* key_padding_mask;   /* [B, L] bytes, 1 = ignore, or NULL */
    const void *pw_qkv, *pw_out;
    const float *b_qkv, *b_out;
    void *qkv, *attn_o; float* lse;          /* out, saved for backward, or all NULL */
    void* x_mid;                             /* out [B*L, C] */
} tan_attnblk_desc;
int tan_attnblk_supported(int L, int C, int H, int dtype);
int tan_attnblk_fwd(const tan_attnblk_desc* d, void* stream);
/* Small batches (round 6): the same branch with one workgroup per (video, head pair) -- B workgroups leave most of the chip idle for
 * 50-70 us per launch at B = 16 -- each storing its [L x 512] f32 term of the out-projection as plane `hp` of `part` (scratch
 * [4, B*L, C] f32); a second launch adds the four planes in order, + b_out + x_in -> x_mid.  Same descriptor, same side outputs. */
int tan_attnblk_fwd_split(const tan_attnblk_desc* d, float* part, void* stream);

/* tools/lab only: device buffer ([8 waves][64] long) that receives workgroup 0's shader clock at the phase boundaries of
 * tan_attnblk_fwd, or NULL (default): no instrumentation */
int tan_attnblk_lab_set_dbg(void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* TAN_HIP_H */
