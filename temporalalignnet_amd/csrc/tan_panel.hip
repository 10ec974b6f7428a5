// Row-panel kernels (gfx950): the MLP branch of a pre-LN block (model/tfm_model.py:23-27,37) as ONE launch per direction.
// See tan_panel.h for the scheme.  A workgroup = 8 waves = one 64-row panel; LDS (160 KiB) holds the normalised input panel
// (64 KiB), the current 256-wide chunk of the hidden activation (32 KiB) and a ring of streamed weight tiles (64 KiB).
//
//   forward, per panel:   xn2 = LN2(x_mid)                                  (prologue; xn2 / mean / rstd also go to HBM for backward)
//     for the 8 chunks c of the 2048 hidden features:
//        h_c  = quickgelu(xn2 W_fc[c]^T + b_fc[c])      64 x 256, K = 512   (h_pre / h_act also go to HBM: dW operands)
//        acc += h_c W_proj[:, c]^T                      64 x 512, K = 256
//     x_out = x_mid + acc + b_proj ;  xn_next = LN_next(x_out)              (epilogue: next block's ln_1, or the stack's post-LN)
//
// Replaces four launches (LayerNorm, c_fc GEMM, c_proj GEMM, next LayerNorm) and their three HBM round trips.
#include "tan_panel.h"

namespace tal {

// ------------------------------------------------------------------------------------------------------------------
// weight packer: table entry e packs the row-major matrix src[e.src_off ..] of shape [N][K] into tiles [TN][TK], tile order
// (n-block, k-block).  Inside a tile the data is FRAGMENT-MAJOR: wave w (of PN_WAVES) owns rows w*TN/PN_WAVES .. and its MFMA A-operand
// fragments follow each other, [row block of 32][k step of 16] -> 1 KiB each, lane l's 16 bytes = W[row0 + f(l & 31)][k0 + 8 * (l >> 5) ..]
// -- exactly what one global_load_dwordx4 per lane puts into the fragment registers.  f permutes the 32 features of a block over
// the MFMA's rows, f(rho) = (rho & 3) + 4 (rho >> 3) + 16 ((rho >> 2) & 1): the 32x32 accumulator gives a lane rows
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5), i.e. with f its 16 registers are the 16 CONSECUTIVE features 16 (lane >> 5) + r of one
// activation row (swapped orientation, D[feature][row]) -- two 16-byte runs without any cross-lane exchange.
__global__ __launch_bounds__(256) void pack_tiles_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                         const tan_pack_entry* __restrict__ table) {
    const tan_pack_entry e = table[blockIdx.y];
    const int tiles_k = e.K / e.TK, ntiles = (e.N / e.TN) * tiles_k;
    const int KS = e.TK / 16, RB = e.TN / (32 * PN_WAVES); // k steps, 32-row blocks per wave
    const int slots = e.TN * e.TK / 8;                    // 16-byte slots per tile
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int nb = t / tiles_k, kt = t % tiles_k;
        const bf16_t* s0 = src + e.src_off + (long)nb * e.TN * e.K + (long)kt * e.TK;
        bf16_t* d0 = dst + e.dst_off + (long)t * e.TN * e.TK;
        if (e.TN == 384) {
            // "qkv16" format (tan_attnblk.hip, GEMM-a): tile (head pair nb, k step kt of 32) = 24 fragments of 16 features x 32 k,
            // fragment p = 3 * wave + fb holds in_proj rows which*512 + (2 nb + j)*64 + fblk*16 .. (which = (p/4) % 3: q|k|v,
            // j = p / 12: head of the pair, fblk = p % 4); lane l: row l & 15, k = 8 (l >> 4) .. + 7 (v_mfma_f32_16x16x32_bf16 A operand)
            for (int s = threadIdx.x; s < slots; s += blockDim.x) {
                const int lane = s & 63, p = s >> 6;
                const int which = (p >> 2) % 3, j = p / 12, fblk = p & 3;
                const int row = which * 512 + (2 * nb + j) * 64 + fblk * 16 + (lane & 15), k = kt * 32 + 8 * (lane >> 4);
                *reinterpret_cast<uint4*>(d0 + (long)s * 8) = *reinterpret_cast<const uint4*>(src + e.src_off + (long)row * e.K + k);
            }
            continue;
        }
        for (int s = threadIdx.x; s < slots; s += blockDim.x) {
            const int lane = s & 63, frag = s >> 6;
            const int ks = frag % KS, rb = (frag / KS) % RB, w = frag / (KS * RB);
            const int rho = lane & 31, f = (rho & 3) + 4 * (rho >> 3) + 16 * ((rho >> 2) & 1);
            const int row = w * (e.TN / PN_WAVES) + rb * 32 + f, k = ks * 16 + 8 * (lane >> 5);
            *reinterpret_cast<uint4*>(d0 + (long)s * 8) = *reinterpret_cast<const uint4*>(s0 + (long)row * e.K + k);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct MlpFwdArgs {
    static constexpr bool bwd = false;
    const bf16_t* x_mid; const float* ln_g; const float* ln_b;
    const char* pw_fc; const char* pw_proj;
    const float* b_fc; const float* b_proj;
    bf16_t* xn2; float* mean2; float* rstd2;
    bf16_t* h_pre; bf16_t* h_act; bf16_t* x_out;
    const float* nln_g; const float* nln_b; bf16_t* xn_next; float* nmean; float* nrstd;
    float eps;
    // optional head (MODE & 2048): x_mid is not read but PRODUCED first, x_mid = x_in + attn_o W_out^T + b_out -- the attention
    // out-projection + bias + residual of the block (tfm_model.py:30-36) where the fused attention launch does not run (L > 80)
    const bf16_t* attn_o; const char* pw_out; const float* b_out; const bf16_t* x_in; bf16_t* x_mid_w;
    float* part; long part_plane;   // SPLIT (MODE & 4096): [8][rows][512] f32 partial c_proj sums, one plane per hidden chunk; plane stride
};
// Backward of the same branch, same schedule with the roles of the two weights exchanged (tan_mlp_bwd):
//   dh_c = (dx W_proj[:, c]) o quickgelu'(h_pre_c)      "c_fc-like": K = 512 over the resident dx panel, packed W_proj^T tiles
//   dxn += dh_c W_fc[c, :]                             "c_proj-like": K = 256 per chunk, packed W_fc^T tiles
//   dx2 = dx + LN2-backward(dxn; x_mid, mean2, rstd2, gamma)
// dh leaves as the side output (operand of the c_fc weight gradient); the four bias / LayerNorm parameter gradients are column
// sums over the panel's rows, added to the f32 gradients with atomics.
struct MlpBwdArgs {
    static constexpr bool bwd = true;
    const bf16_t* dx; const bf16_t* h_pre; const bf16_t* x_mid;
    const float* mean2; const float* rstd2; const float* ln_g;
    const char* pw_fc;          // packed W_proj^T [2048][512]  (tiles [256][32])
    const char* pw_proj;        // packed W_fc^T   [512][2048]  (tiles [512][16])
    bf16_t* h_act;              // dh out [rows][2048] (the "activation" side output of the shared schedule)
    bf16_t* dx2;
    float* g_b_fc; float* g_ln_g; float* g_ln_b; float* g_b_out;
    // optional prologue: dx = ln1_res + LN-backward(ln1_dxn; ln1_x, ...) of the NEXT block's ln_1 instead of reading dx
    const bf16_t* ln1_dxn; const bf16_t* ln1_x; const bf16_t* ln1_res;
    const float* ln1_mean; const float* ln1_rstd; const float* ln1_g;
    float* g_ln1_g; float* g_ln1_b; float* g_dx_colsum; bf16_t* dx_out;
    // optional tail: d_o = dx2 W_out (the attention out-projection's dX GEMM) on the resident dx2 panel
    const char* pwt_out;        // packed W_out^T [512][512]  (tiles [512][16])
    bf16_t* d_o;
    // optional head (MODE & 512, with the ln_1 prologue): ln1_dxn is not read but PRODUCED first, dxn1 = dqkv W_in (+ dstage) -- the
    // dX GEMM of the NEXT block's attention in-projection, a [64 x 1536] x [1536 x 512] row-local product
    const bf16_t* dqkv;         // [rows][1536]
    const char* pwt_in;         // packed W_in^T [512][1536]  (tiles [512][16])
    const bf16_t* dstage;       // [rows][512] or NULL: the deep-supervision gradient that joins at that ln_1 output
    float* part; long part_plane;   // SPLIT (MODE & 4096): [8][rows][512] f32 partial sums of dxn = dh W_fc, one plane per hidden chunk
};

// Tiles are 16 KiB: c_fc [256 features][32 k], c_proj [512 features][16 k].  A weight fragment is consumed by exactly ONE wave
// (the wave that owns those output features), so the weights never touch LDS: every wave streams its own fragments straight into
// registers, global_load_dwordx4 per lane = 1 KiB per wave-instruction of the fragment-major packed image, MLP_D steps (32 KiB per
// wave, 128 KiB per CU) ahead of their use -- the compiler's own counted vmcnt keeps that many loads in flight.  LDS holds only
// the activation panels (the normalised input, and the hidden chunk double-buffered), read by all waves.  FOUR waves, one per
// SIMD, each with the whole 512-entry register file: 192 accumulator registers (64 rows x 64 hidden features + 64 rows x 128
// output features) sit in AGPRs, the weight ring (128) and the activation fragments in VGPRs.
//
// A lone wave per SIMD overlaps nothing by itself, so the schedule does: the chunk epilogue of chunk c (bias, QuickGELU, bf16,
// stores: VALU / LDS / VMEM work) is cut into 16 half-units and dealt one per step into the c_proj steps of chunk c-1 (matrix
// work that does not depend on it):
//     body(c):  [16 steps: c_fc(c)]  [16 steps: c_proj(c-1) MFMAs || epilogue(c) half-units]  barrier
// with body(0) = c_fc(0) + epilogue(0), body(8) = c_proj(7); every step also reads the NEXT step's activation fragments from
// LDS and re-loads the ring slot it just consumed.  One barrier per body hands hidden chunk c over (written by all waves, read
// by all waves in body(c+1)); the two hidden-panel buffers alternate.
constexpr int MLP_KDF = 32, MLP_KDP = 16, MLP_TILE = 16384, MLP_TF = 512 / MLP_KDF, MLP_TP = 256 / MLP_KDP;
#ifndef TAN_HPRE_STEP
#define TAN_HPRE_STEP 2
#endif
#ifndef TAN_MLP_D
#define TAN_MLP_D 4
#endif
constexpr int MLP_D = TAN_MLP_D;                 // weight prefetch distance in steps (4: as fast as 8 once the weights are requested up front, 32 registers less)
constexpr int MLP_XN_OFF = 0, MLP_H_OFF = 65536, MLP_PRE_OFF = 131072, MLP_LDS = 163840;   // input panel | hidden x2 | pre-activation
static_assert(PN_WAVES == 8, "eight waves: two groups of four, one wave of each per SIMD");
constexpr int MLP_NBH = 256 / (32 * PN_WAVES);      // 32-feature blocks of a hidden chunk per wave (2 | 1)
constexpr int MLP_NBO = 512 / (32 * PN_WAVES);      // 32-feature blocks of the output per wave (4 | 2)
constexpr int MLP_WFR = MLP_NBO;                    // weight fragments per step: c_fc NBH x 2 k steps = c_proj NBO
constexpr int MLP_HU = 8 * MLP_NBH;                 // epilogue half-units per chunk (16 | 8), dealt over the 16 c_proj steps
constexpr int MLP_HU_STRIDE = 16 / MLP_HU;
static_assert(MLP_TF == 16 && MLP_TP == 16 && MLP_TF % MLP_D == 0, "step bookkeeping");

struct MlpXFrags { bf16x8 f[4]; };       // activation fragments of one step: c_fc [k step][row block], c_proj [row block]
struct MlpWFrags { bf16x8 f[MLP_WFR]; };       // weight fragments of one step: c_fc [feature block][k step], c_proj [feature block]

// Activation-fragment addresses.  Chunk c = cJ + hi (cJ = first 16-byte chunk of the k step, even; hi = lane >> 5) of row r sits at
// slot c ^ (r & 15) = (cJ & ~15) + ((cJ & 15) ^ t) with t = hi ^ (r & 15): the part that depends on the lane takes only EIGHT values
// per panel, held in registers (xa[e] for cJ & 15 = 2e), the rest is an immediate offset.  Left to the compiler, every step of the
// unrolled loop kept its own pre-computed address register (48 of them) and the weight ring spilled.
struct MlpXAddr { unsigned xn[8], h[8]; };       // LDS byte addresses, row block 0; row block 1 = + 32 rows
__device__ __forceinline__ void mlp_xaddr_init(MlpXAddr& A, const char* lds, int lane) {
    const int row = lane & 31, t = (lane >> 5) ^ (row & 15);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        A.xn[e] = (unsigned)(uintptr_t)(lds + MLP_XN_OFF + row * 1024 + (((2 * e) ^ t) << 4));
        A.h[e] = (unsigned)(uintptr_t)(lds + MLP_H_OFF + row * 512 + (((2 * e) ^ t) << 4));
    }
}
typedef __attribute__((address_space(3))) const bf16x8* mlp_lds_frag_t;
template <int OFF>
__device__ __forceinline__ bf16x8 mlp_lds_frag(unsigned addr) { return *(mlp_lds_frag_t)(uintptr_t)(addr + OFF); }

template <int KT>
__device__ __forceinline__ void mlp_load_x_fc(MlpXFrags& F, const MlpXAddr& A) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int cJ = KT * 4 + 2 * i;                       // 16-byte chunk of k = KT*32 + 16*i
        F.f[2 * i + 0] = mlp_lds_frag<0>(A.xn[(cJ & 15) >> 1] + (cJ & ~15) * 16);
        F.f[2 * i + 1] = mlp_lds_frag<32 * 1024>(A.xn[(cJ & 15) >> 1] + (cJ & ~15) * 16);
    }
}
// fragments of K step KT (k = 16 KT) of a [64 rows][512] panel with 1-KiB rows at byte offset POFF from the input panel
template <int KT, int POFF>
__device__ __forceinline__ void mlp_load_x_k16(bf16x8 (&f)[2], const MlpXAddr& A) {
    constexpr int cJ = KT * 2;
    const unsigned a0 = A.xn[(cJ & 15) >> 1] + (cJ & ~15) * 16;
    f[0] = mlp_lds_frag<POFF>(a0);
    f[1] = mlp_lds_frag<POFF + 32 * 1024>(a0);
}
template <int KT>
__device__ __forceinline__ void mlp_load_x_proj(MlpXFrags& F, const MlpXAddr& A, int hb) {
    constexpr int cJ = KT * 2;                               // k = KT*16
    const unsigned a0 = A.h[(cJ & 15) >> 1] + (cJ & ~15) * 16 + hb * 32768;
    F.f[0] = mlp_lds_frag<0>(a0);
    F.f[1] = mlp_lds_frag<32 * 512>(a0);
}

__device__ __forceinline__ void mlp_load_w(MlpWFrags& W, const char* tile, int wave, int lane) {
    const char* p = tile + wave * (MLP_TILE / PN_WAVES) + lane * 16;
#pragma unroll
    for (int i = 0; i < MLP_WFR; ++i) W.f[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
}

__device__ __forceinline__ void mlp_mma_fc(const MlpWFrags& W, const MlpXFrags& F, f32x16 (&acc_h)[MLP_NBH][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)          // W.f[2 * nb + i]: hidden features wave*64 + nb*32 .., k step i
#pragma unroll
        for (int nb = 0; nb < MLP_NBH; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                acc_h[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[2 * nb + i], F.f[2 * i + mb], acc_h[nb][mb], 0, 0, 0);
}
__device__ __forceinline__ void mlp_mma_proj(const MlpWFrags& W, const MlpXFrags& F, f32x16 (&acc_o)[MLP_NBO][2]) {
#pragma unroll
    for (int nb = 0; nb < MLP_NBO; ++nb)       // W.f[nb]: output features wave*128 + nb*32 .., one k step
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc_o[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], F.f[mb], acc_o[nb][mb], 0, 0, 0);
}

// Chunk epilogue, dealt over the 16 steps of the c_proj phase: a lane owns, per row block mb, the 16 consecutive hidden features
// 16 hi + r of row mb * 32 + (lane & 31) (packer's feature permutation); step J finishes registers 2 (J & 7), 2 (J & 7) + 1 of row
// block J >> 3 -- one packed bf16 pair of the activation and one of the pre-activation -- and every fourth step stores the two
// 16-byte runs (activation panel + pre-activation panel).  Three pieces per step, issued between the step's four MFMAs (in-order
// issue: the wave's own MFMAs then execute under its VALU work, and the quarter-rate v_exp_f32 / v_rcp_f32 results are consumed one
// MFMA later):  P1 scale, 2 x exp2;  P2 1 + e, 2 x rcp;  P3 x * r, two packs, stores.  The bias is NOT added here: the accumulator
// of a chunk starts from it (mlp_init_h).
struct MlpEpiState { float ce[2][2]; uint32_t pre[4], act[4]; };      // ce[unit & 1]: two half-units may be in flight (balanced slots)
struct MlpBias32 { float b[32]; };       // b_fc[c * 256 + wave * 32 ..]: wave-uniform, through scalar loads
__device__ __forceinline__ void mlp_bias32_load(MlpBias32& B, const float* b_fc, int c, int wave) {
    pn_cfptr_t bp = (pn_cfptr_t)(uintptr_t)b_fc + c * 256 + wave * 32;
#pragma unroll
    for (int e = 0; e < 32; ++e) B.b[e] = bp[e];
}
__device__ __forceinline__ void mlp_init_h(f32x16 (&acc_h)[MLP_NBH][2], const MlpBias32& B, int hi) {
    static_assert(MLP_NBH == 1, "eight waves: one 32-feature block of the hidden chunk per wave");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = hi ? B.b[16 + r] : B.b[r];
        acc_h[0][0][r] = v;
        acc_h[0][1][r] = v;
    }
}
template <int J>
__device__ __forceinline__ void mlp_epi_p1(MlpEpiState& E, const f32x16 (&acc_h)[MLP_NBH][2]) {
    constexpr int mb = J >> 3, r = 2 * (J & 7), u = J & 1;
    E.ce[u][0] = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * acc_h[0][mb][r]);
    E.ce[u][1] = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * acc_h[0][mb][r + 1]);
}
template <int J>
__device__ __forceinline__ void mlp_epi_p2(MlpEpiState& E) {
    constexpr int u = J & 1;
    E.ce[u][0] = __builtin_amdgcn_rcpf(1.0f + E.ce[u][0]);
    E.ce[u][1] = __builtin_amdgcn_rcpf(1.0f + E.ce[u][1]);
}
template <int J>
__device__ __forceinline__ void mlp_epi_p3(MlpEpiState& E, const f32x16 (&acc_h)[MLP_NBH][2], char* lds, int hb, int wave, int lane) {
    constexpr int mb = J >> 3, r = 2 * (J & 7), w = J & 3, u = J & 1;
    const float x0 = acc_h[0][mb][r], x1 = acc_h[0][mb][r + 1];
    E.pre[w] = f2bf2(x0, x1);
    E.act[w] = f2bf2(x0 * E.ce[u][0], x1 * E.ce[u][1]);       // QuickGELU: x * sigmoid(1.702 x)
    if constexpr (w == 3) {
        const int ch = (wave * 32 >> 3) + 2 * (lane >> 5) + ((J & 7) >> 2), m = mb * 32 + (lane & 31);
        *reinterpret_cast<uint4*>(pn_panel_slot<512>(lds + MLP_H_OFF + hb * 32768, m, ch)) = make_uint4(E.act[0], E.act[1], E.act[2], E.act[3]);
        // the pre-activation goes to its own LDS panel: both leave for HBM as whole half rows under the next phases (mlp_copy_step*)
        *reinterpret_cast<uint4*>(pn_panel_slot<512>(lds + MLP_PRE_OFF, m, ch)) = make_uint4(E.pre[0], E.pre[1], E.pre[2], E.pre[3]);
    }
}

// ---- backward chunk epilogue: dh = acc o quickgelu'(h_pre), step J = row block J & 1, register pair q = J >> 1 (both row blocks of
// a feature pair in adjacent steps: their sum is the lane's share of the c_fc bias gradient).  The pre-activations come straight
// from HBM in the accumulator's layout (16 consecutive features of a row = two 16-byte loads per row block), issued under the
// c_fc-like phase of the same chunk.
struct MlpHPre { uint4 q[2][2]; };       // [row block][8-feature half]
struct MlpBwdEpi { float x[2][2], ce[2][2], cs[16]; uint32_t w[2][4]; };      // x / ce [unit & 1]
template <int MB, int HALF>
__device__ __forceinline__ void mlp_hpre_load(MlpHPre& H, const bf16_t* h_pre, long row0, int c, int wave, int lane) {
    H.q[MB][HALF] = *reinterpret_cast<const uint4*>(h_pre + (row0 + MB * 32 + (lane & 31)) * 2048 + c * 256 + wave * 32 + 16 * (lane >> 5) + 8 * HALF);
}
__device__ __forceinline__ float pn_half32_sum(float v) {      // sum over the 32 lanes that share lane >> 5, in every one of them
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r16[0]) + __int_as_float(r16[1]);
}
template <int J>
__device__ __forceinline__ void mlp_bepi_p1(MlpBwdEpi& E, const MlpHPre& H) {
    constexpr int mb = J & 1, q = J >> 1, un = J & 1;
    const uint4 u = H.q[mb][q >> 2];
    const uint32_t wd = (q & 3) == 0 ? u.x : (q & 3) == 1 ? u.y : (q & 3) == 2 ? u.z : u.w;
    E.x[un][0] = __uint_as_float(wd << 16);
    E.x[un][1] = __uint_as_float(wd & 0xffff0000u);
    E.ce[un][0] = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * E.x[un][0]);
    E.ce[un][1] = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * E.x[un][1]);
}
template <int J>
__device__ __forceinline__ void mlp_bepi_p2(MlpBwdEpi& E) {
    constexpr int un = J & 1;
    E.ce[un][0] = __builtin_amdgcn_rcpf(1.0f + E.ce[un][0]);       // s = sigmoid(1.702 x)
    E.ce[un][1] = __builtin_amdgcn_rcpf(1.0f + E.ce[un][1]);
}
template <int J>
__device__ __forceinline__ void mlp_bepi_p3(MlpBwdEpi& E, const f32x16 (&acc_h)[MLP_NBH][2], char* lds, int hb, int wave, int lane) {
    constexpr int mb = J & 1, q = J >> 1, un = J & 1;
    // quickgelu'(x) = s + 1.702 x s (1 - s), two values per packed-f32 instruction (v_pk_mul_f32 / v_pk_fma_f32)
    typedef float pn_f2 __attribute__((ext_vector_type(2)));
    const pn_f2 xx = {E.x[un][0], E.x[un][1]}, sg = {E.ce[un][0], E.ce[un][1]}, av = {acc_h[0][mb][2 * q], acc_h[0][mb][2 * q + 1]};
    const pn_f2 t = (xx * 1.702f) * sg;
    const pn_f2 dd = av * (t * (1.0f - sg) + sg);
    const float d[2] = {dd[0], dd[1]};
    E.w[mb][q & 3] = f2bf2(d[0], d[1]);
    // the lane's share of the c_fc bias gradient (two rows per feature); reduced over the 32 lanes once per chunk (pn_colsum16)
    if constexpr (mb == 0) {
        E.cs[2 * q] = d[0]; E.cs[2 * q + 1] = d[1];
    } else {
        E.cs[2 * q] += d[0]; E.cs[2 * q + 1] += d[1];
    }
    if constexpr ((q & 3) == 3) {
        const int ch = (wave * 32 >> 3) + 2 * (lane >> 5) + (q >> 2), m = mb * 32 + (lane & 31);
        *reinterpret_cast<uint4*>(pn_panel_slot<512>(lds + MLP_H_OFF + hb * 32768, m, ch)) = make_uint4(E.w[mb][0], E.w[mb][1], E.w[mb][2], E.w[mb][3]);
    }
}

// The side outputs (pre-activation and activation chunk, the operands of backward) leave for HBM ONE row-instruction at a time,
// spread over BOTH phases of the body that follows the chunk's barrier: the four row-instructions of the pre-activation panel under
// c_fc(c+1) (the panel is rewritten by the next epilogue), the four of the activation panel under c_proj(c) || epilogue(c+1).  As a burst between two barriers (the first version)
// the 16 MiB that all CUs store at once take longer to drain than the weight ring covers, and stores retire in order with the
// ring's loads: 31 us of a 94 us launch (tools/lab/mlp_lab.py, cold buffers).
struct MlpCopy { uint4 v; uint32_t lds0, voff; };
// A wave group (waves 4g .. 4g+3) owns hidden features [128 g, 128 g + 128) of a chunk: its half of the two LDS panels is copied by
// its own waves, one instruction = four half rows (16 lanes x 16 B = 256 B each); wave-in-group wg takes rows 16 wg .. 16 wg + 15.
__device__ __forceinline__ void mlp_copy_init(MlpCopy& C, int wave, int lane) {
    static_assert(PN_WAVES == 8, "two wave groups of four");
    const int g = wave >> 2, row_b = (wave & 3) * 16 + (lane >> 4), chunk = g * 16 + (lane & 15);   // rows row_b + 4 i, i = 0..3
    C.lds0 = row_b * 512 + ((chunk ^ (row_b & 15)) << 4);
    C.voff = (row_b * 2048 + chunk * 8) * 2;
}
// step J of a 16-step phase moving the group's half of ONE panel (ACT: activation panel of chunk cprev, else the pre-activation
// panel): row-instruction i = J / 4 is read from LDS at J = 4 i and stored at J = 4 i + 1
template <int J, bool ACT>
__device__ __forceinline__ void mlp_copy_step4(MlpCopy& C, const char* lds, bf16_t* dst, long row0, int cprev, int cdata) {
    constexpr int i = J >> 2;
    if constexpr ((J & 3) == 0) {
        const char* panel = ACT ? lds + MLP_H_OFF + (cprev & 1) * 32768 : lds + MLP_PRE_OFF;
        C.v = *reinterpret_cast<const uint4*>(panel + ((C.lds0 ^ (i << 6)) + i * 2048));      // row & 15 gains 4 i: no carry
    } else if constexpr ((J & 3) == 1) {
        bf16_t* base = dst + row0 * 2048 + cdata * 256;                 // wave-uniform (cdata: the chunk's place in the hidden dimension)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + C.voff + i * 16384) = C.v;
    }
}
// both panels in one 16-step phase (the last one): pair q = J / 2
template <int J>
__device__ __forceinline__ void mlp_copy_step8(MlpCopy& C, const char* lds, bf16_t* dst_pre, bf16_t* dst_act, long row0, int cprev, int cdata) {
    constexpr int q = J >> 1, i = q & 3;
    if constexpr ((J & 1) == 0) {
        const char* panel = q < 4 ? lds + MLP_PRE_OFF : lds + MLP_H_OFF + (cprev & 1) * 32768;
        C.v = *reinterpret_cast<const uint4*>(panel + ((C.lds0 ^ (i << 6)) + i * 2048));
    } else {
        bf16_t* base = (q < 4 ? dst_pre : dst_act) + row0 * 2048 + cdata * 256;
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(base) + C.voff + i * 16384) = C.v;
    }
}

// LayerNorm backward of a panel whose upstream gradient dy sits in the f32 accumulators (+ an optional bf16 gradient `dadd` joining
// it), plus the residual gradient that sits in the LDS panel at MLP_XN_OFF: out = dres + LN-backward(dy; x, mean, rstd, gamma),
// written back into the same panel slots (every slot is owned by one lane) and left there for pn_panel_copy_out; the parameter
// gradients that are column sums over the panel's rows go to the f32 gradients with one atomic per column.
struct PnLnBwd {
    const bf16_t* x; const float* mean; const float* rstd; const float* gamma;
    const bf16_t* dadd;                 // optional [rows][512]: added to dy first (the deep-supervision gradient of this LayerNorm's output)
    float* g_gamma; float* g_beta; float* g_colsum;      // g_colsum optional: column sums of the OUTPUT
};
__device__ __forceinline__ void pn_ln_bwd_epilogue(f32x16 (&acc_o)[MLP_NBO][2], char* lds, const PnLnBwd& P, long row0, int wave, int lane) {
    const int hi = lane >> 5;
    if (P.dadd) {
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float dv[8];
                    pn_unpack8(*reinterpret_cast<const uint4*>(P.dadd + (row0 + mb * 32 + (lane & 31)) * 512 + wave * (32 * MLP_NBO) + nb * 32 + (2 * hi + p) * 8), dv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc_o[nb][mb][8 * p + e] += dv[e];
                }
    }
    // ---- backward epilogue: LayerNorm-2 backward of dxn (the f32 accumulator) + the residual gradient -> dx2; column sums
    // over the panel's rows -> g_ln_g, g_ln_b, g_b_out.  A lane owns rows mb * 32 + (lane & 31) and features nbase + 16 hi + r
    // (two 8-feature chunks per feature block and row block); its dx values sit in the input panel at the slots its dx2 values
    // take, so the panel is updated in place and leaves as whole rows.
    char* xo_panel = lds + MLP_XN_OFF;
    float* red = reinterpret_cast<float*>(lds + MLP_PRE_OFF);     // [2][PN_WAVES][64 rows]
    float mean[2], rstd[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        mean[mb] = P.mean[row0 + mb * 32 + (lane & 31)];
        rstd[mb] = P.rstd[row0 + mb * 32 + (lane & 31)];
    }
    uint4 xq[MLP_NBO][2][2];       // x_mid chunks, kept packed for the second pass
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float cg = 0.f, cb = 0.f;      // lane (l & 31) = 16 nb + r: column sums of feature nbase(nb) + 16 hi + r
#pragma unroll
    for (int nb = 0; nb < MLP_NBO; ++nb) {
        const int nbase = wave * (32 * MLP_NBO) + nb * 32;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                xq[nb][mb][p] = *reinterpret_cast<const uint4*>(P.x + (row0 + mb * 32 + (lane & 31)) * 512 + nbase + (2 * hi + p) * 8);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pn_cfptr_t gp = (pn_cfptr_t)(uintptr_t)(P.gamma) + nbase + 8 * p;
            float gam[8];
            pn_uniform8(gp, gp + 16, hi, gam);
            float dg[8], db[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float xv[8];
                pn_unpack8(xq[nb][mb][p], xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mean[mb]) * rstd[mb], dy = acc_o[nb][mb][8 * p + e], g = dy * gam[e];
                    s1[mb] += g;
                    s2[mb] += g * xh;
                    dg[e] += dy * xh;
                    db[e] += dy;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float tg = pn_half32_sum(dg[e]), tb = pn_half32_sum(db[e]);
                const bool mine = (lane & 31) == 16 * nb + 8 * p + e;
                cg = mine ? tg : cg;
                cb = mine ? tb : cb;
            }
        }
    }
    {
        const int n = wave * (32 * MLP_NBO) + ((lane & 31) >> 4) * 32 + 16 * hi + (lane & 15);
        unsafeAtomicAdd(P.g_gamma + n, cg);
        unsafeAtomicAdd(P.g_beta + n, cb);
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const float t1 = pn_half_sum(s1[mb]), t2 = pn_half_sum(s2[mb]);
        if (lane < 32) { red[wave * 64 + mb * 32 + lane] = t1; red[PN_WAVES * 64 + wave * 64 + mb * 32 + lane] = t2; }
    }
    __syncthreads();
    float m1[2], m2[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < PN_WAVES; ++w) { t1 += red[w * 64 + mb * 32 + (lane & 31)]; t2 += red[PN_WAVES * 64 + w * 64 + mb * 32 + (lane & 31)]; }
        m1[mb] = t1 * (1.0f / 512);
        m2[mb] = t2 * (1.0f / 512);
    }
    float co = 0.f;
#pragma unroll
    for (int nb = 0; nb < MLP_NBO; ++nb) {
        const int nbase = wave * (32 * MLP_NBO) + nb * 32;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pn_cfptr_t gp = (pn_cfptr_t)(uintptr_t)(P.gamma) + nbase + 8 * p;
            float gam[8], ds[8];
            pn_uniform8(gp, gp + 16, hi, gam);
#pragma unroll
            for (int e = 0; e < 8; ++e) ds[e] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int m = mb * 32 + (lane & 31), ch = (nbase >> 3) + 2 * hi + p;
                uint4* slot = reinterpret_cast<uint4*>(pn_panel_slot<1024>(xo_panel, m, ch));
                float xv[8], rv[8], o[8];
                pn_unpack8(xq[nb][mb][p], xv);
                pn_unpack8(*slot, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mean[mb]) * rstd[mb], g = acc_o[nb][mb][8 * p + e] * gam[e];
                    o[e] = rstd[mb] * (g - m1[mb] - xh * m2[mb]) + rv[e];
                    ds[e] += o[e];
                }
                *slot = pn_pack8(o);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = pn_half32_sum(ds[e]);
                co = (lane & 31) == 16 * nb + 8 * p + e ? t : co;
            }
        }
    }
    if (P.g_colsum) {
        const int n = wave * (32 * MLP_NBO) + ((lane & 31) >> 4) * 32 + 16 * hi + (lane & 15);
        unsafeAtomicAdd(P.g_colsum + n, co);
    }
    __syncthreads();
}

#ifdef TAN_PANEL_LAB
__device__ long long* g_panel_dbg = nullptr;      // tools/lab: phase clocks of workgroup 0 (tan_panel_lab_set_dbg)
#endif

// MODE 0: the kernel.  Timing experiments of tools/lab/mlp_lab.py (results undefined), bit mask: 1 no MFMAs, 2 no weight
// streaming (loaded once), 4 no activation-fragment reads in the loop, 16 no side-output copy-out (arithmetic, LDS panel writes and barriers stay), 64 phase clocks
// into nrstd[], 256 no up-front touch of the weights.  (Variants that drop the epilogue arithmetic also drop the c_fc MFMAs -- dead code --
// and measure nothing useful: removed.)
// MODE & 4096 = SPLIT (round 6, small batches): the grid is (panels, 8) and workgroup (p, c0) runs ONE hidden chunk of panel p -- the
// prologue, c_fc(c0) + its epilogue + side outputs, c_proj(c0) -- and stores its [64 x 512] f32 partial sum as plane c0 of `part`; the
// row epilogue (sum of the eight planes in a fixed order, bias + residual + next LayerNorm / LayerNorm-2 backward) is a launch of its own
// (mlp_split_finish_*).  (First version: f32 atomics into one plane -- 4 M device-scope atomics per launch, 3x SLOWER than the
// whole-panel kernels: 7.4 vs 2.56 ms per step at B = 16.)  A whole-panel
// workgroup streams all 4 MiB of the block's weights whatever the batch: 16 panels take as long as 160 (85 / 133 us per launch at
// B = 16 with 16 of 256 CUs busy); eight workgroups per panel stream 512 KiB each.
template <int MODE, typename ArgsT>
__global__ __launch_bounds__(64 * PN_WAVES, PN_WAVES / 4) void mlp_panel_kernel(ArgsT a) {
    constexpr bool BWD = ArgsT::bwd;
    constexpr bool SPLIT = (MODE & 4096) != 0;
    constexpr int TILE = MLP_TILE, D = MLP_D;
    constexpr int XN_OFF = MLP_XN_OFF, H_OFF = MLP_H_OFF;
    __shared__ __attribute__((aligned(1024))) char lds[MLP_LDS];

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * PN_ROWS;
    const int c0 = SPLIT ? (int)blockIdx.y : 0;          // SPLIT: the one hidden chunk this workgroup runs
    const bool first = !SPLIT || c0 == 0;                // (outputs every chunk's workgroup would write alike are written by chunk 0's)
    const char* const pfc = a.pw_fc;
    const char* const ppj = a.pw_proj;

    // First touch of the packed weights (4 MiB, used once per step and stack: cold in every L2).  Streamed cold, every tile's first
    // miss goes to HBM in the middle of the kernel's own 100 MB of side-output writes and the ring (eight steps) cannot cover that
    // latency: 97-104 us per launch against 72-77 us with the weights resident in the Infinity Cache (tools/lab/mlp_lab.py, E2/E3).
    // So the whole set is requested up front, one dword per 128-byte line spread over the first 64 workgroups, while HBM is quiet.
    if (!(MODE & 256) && !SPLIT) {
        const int L = blockIdx.x * (64 * PN_WAVES) + tid;
        if (L < 32768) {
            const char* q = L < 16384 ? pfc + (long)L * 128 : ppj + (long)(L - 16384) * 128;
            (void)*reinterpret_cast<const volatile uint32_t*>(q);
        }
    }
    constexpr bool INP = BWD && (MODE & 512) != 0;      // the next block's in_proj dX GEMM runs first (its packed W_in^T leads the ring)
    constexpr bool OUTP = !BWD && (MODE & 2048) != 0;   // forward: the block's out_proj + bias + residual runs first (packed W_out leads)
    static_assert(!(SPLIT && (INP || OUTP)), "the split kernels run the MLP branch only");
    const char* pin = pfc;
    if constexpr (OUTP) pin = a.pw_out;
    if constexpr (INP) {
        pin = a.pwt_in;
        const int L = blockIdx.x * (64 * PN_WAVES) + tid;          // first touch of its 1.5 MiB as well
        if (L < 12288) (void)*reinterpret_cast<const volatile uint32_t*>(pin + (long)L * 128);
    }
    // the weight stream does not depend on the activations: start it before anything else (ring slots 0..7 = c_fc(0) tiles 0..7)
    MlpWFrags WQ[D];
    pn_static_for<0, D>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        mlp_load_w(WQ[J], pin + (long)(((INP || OUTP) ? 0 : c0 * 16) + J) * TILE, wave, lane);
    });

    // ---- prologue.  Forward: LN2 of the panel, 64 / PN_WAVES rows per wave (batches of 8), one 16-byte chunk per lane.  Backward:
    // the dx panel as it is.
    if constexpr (OUTP) {
        // ---- head (forward): x_mid = x_in + attn_o W_out^T + b_out.  The attn_o panel goes HBM -> LDS by LDS-DMA into the hidden panel
        // space, 32 K-steps of 16 through the weight ring (which then runs into c_fc(0)'s tiles), the bf16 result lands in the input
        // panel space -- where LN2 below normalises it in place -- and leaves for HBM as whole rows (the backward and this kernel's
        // own residual read it from there).
        constexpr int RPW = PN_ROWS / PN_WAVES;
        char* pX = lds + XN_OFF;
        char* pH = lds + H_OFF;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave * RPW + r;
            const bf16_t* g = a.attn_o + (row0 + row) * 512 + ((lane ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((pn_gptr_t)g, (pn_lptr_t)(pH + row * 1024), 16, 0, 0);
        }
        uint4 rq[MLP_NBO][2][2];          // residual rows in the accumulator layout
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    rq[nb][mb][pp] = *reinterpret_cast<const uint4*>(a.x_in + (row0 + mb * 32 + (lane & 31)) * 512 + wave * (32 * MLP_NBO) +
                                                                     nb * 32 + 8 * pp + 16 * hi);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MlpXAddr XH;
        mlp_xaddr_init(XH, lds, lane);
        f32x16 acc_d[MLP_NBO][2];
#pragma unroll
        for (int i = 0; i < MLP_NBO; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc_zero(acc_d[i][j]);
        bf16x8 xf[2][2];
        mlp_load_x_k16<0, H_OFF - XN_OFF>(xf[0], XH);
        pn_static_for<0, 32>([&](auto jc) {
            constexpr int KT = decltype(jc)::value;
            MlpWFrags& W = WQ[KT % D];
            if constexpr (KT < 31) mlp_load_x_k16<KT + 1, H_OFF - XN_OFF>(xf[(KT + 1) & 1], XH);
            const bf16x8 x0 = xf[KT & 1][0], x1 = xf[KT & 1][1];
#pragma unroll
            for (int nb = 0; nb < MLP_NBO; ++nb) {
                acc_d[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x0, acc_d[nb][0], 0, 0, 0);
                acc_d[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x1, acc_d[nb][1], 0, 0, 0);
            }
            if constexpr (KT + D < 32) mlp_load_w(W, pin + (long)(KT + D) * TILE, wave, lane);
            else mlp_load_w(W, pfc + (long)((0) * 16 + KT + D - 32) * TILE, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb) {
            const int nbase = wave * (32 * MLP_NBO) + nb * 32;
            pn_cfptr_t bp = (pn_cfptr_t)(uintptr_t)(a.b_out) + nbase;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float bias[8], res[8], v[8];
                    pn_uniform8(bp + 8 * pp, bp + 16 + 8 * pp, hi, bias);
                    pn_unpack8(rq[nb][mb][pp], res);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc_d[nb][mb][8 * pp + e] + bias[e] + res[e];
                    *reinterpret_cast<uint4*>(pn_panel_slot<1024>(pX, mb * 32 + (lane & 31), ((nbase + 8 * pp) >> 3) + 2 * hi)) = pn_pack8(v);
                }
        }
        __syncthreads();
        pn_panel_copy_out<1024>(pX, a.x_mid_w + row0 * 512, 512, wave, lane);
    }
    if constexpr (INP) {
        // ---- head: dxn1 = dqkv W_in (+ dstage), K = 1536 as three [64 x 512] panels of dqkv staged in the (idle) input / hidden
        // panel space, "c_proj-like" (the wave owns 64 output features, 96 K-steps of 16); the result lands in the hidden panel space
        // as the [64 x 512] bf16 panel the ln_1 backward below reads instead of HBM rows.  Was a launch of its own per block:
        // 8192 x 1536 x 512 in 128 x 128 tiles, one workgroup of four waves per CU, 36 us of the whole chip at 14 % of its MFMA peak.
        constexpr int RPW = PN_ROWS / PN_WAVES;
        char* pX = lds + XN_OFF;
        char* pH = lds + H_OFF;
        // dqkv K panels go HBM -> LDS by LDS-DMA (no staging registers): one wave-instruction per 1-KiB panel row, the panel's chunk
        // swizzle applied on the global side
        auto stage_panel = [&](int kp, char* panel) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r;
                const bf16_t* g = a.dqkv + (row0 + row) * 1536 + kp * 512 + ((lane ^ (row & 15)) << 3);
                __builtin_amdgcn_global_load_lds((pn_gptr_t)g, (pn_lptr_t)(panel + row * 1024), 16, 0, 0);
            }
        };
        stage_panel(0, pX);
        stage_panel(1, pH);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc_d[MLP_NBO][2];
#pragma unroll
        for (int i = 0; i < MLP_NBO; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc_zero(acc_d[i][j]);
        MlpXAddr XH;
        mlp_xaddr_init(XH, lds, lane);
        auto gemm_panel = [&](auto kpc, auto poffc) __attribute__((always_inline)) {
            constexpr int KP = decltype(kpc)::value, POFF = decltype(poffc)::value;
            bf16x8 xf[2][2];                   // the activation fragments of the even / odd steps: read one step ahead of their MFMAs
            mlp_load_x_k16<0, POFF>(xf[0], XH);
            pn_static_for<0, 32>([&](auto jc) {
                constexpr int KT = decltype(jc)::value, T = KP * 32 + KT;
                MlpWFrags& W = WQ[T % D];
                if constexpr (KT < 31) mlp_load_x_k16<KT + 1, POFF>(xf[(KT + 1) & 1], XH);
                const bf16x8 x0 = xf[KT & 1][0], x1 = xf[KT & 1][1];
#pragma unroll
                for (int nb = 0; nb < MLP_NBO; ++nb) {
                    acc_d[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x0, acc_d[nb][0], 0, 0, 0);
                    acc_d[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x1, acc_d[nb][1], 0, 0, 0);
                }
                if constexpr (T + D < 96) mlp_load_w(W, pin + (long)(T + D) * TILE, wave, lane);
                else mlp_load_w(W, pfc + (long)(T + D - 96) * TILE, wave, lane);   // the ring ends on c_fc(0)'s first tiles
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        gemm_panel(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        __syncthreads();                       // every wave is done with K panel 0
        stage_panel(2, pX);                    // in flight under K panel 1
        gemm_panel(std::integral_constant<int, 1>{}, std::integral_constant<int, H_OFF - XN_OFF>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * MLP_WFR) : "memory");     // (in-order returns: everything older than the ring's last D steps)
        __syncthreads();                       // K panel 2 is in place, every wave is done with K panel 1
        uint4 dsq[MLP_NBO][2][2];
        const bf16_t* dsp = a.dstage ? a.dstage : a.ln1_x;                 // (unconditional loads; weighted below)
        const float dmul = a.dstage ? 1.0f : 0.0f;
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    dsq[nb][mb][pp] = *reinterpret_cast<const uint4*>(dsp + (row0 + mb * 32 + (lane & 31)) * 512 + wave * (32 * MLP_NBO) +
                                                                      nb * 32 + 8 * pp + 16 * hi);
        gemm_panel(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float v[8], d8[8];
                    pn_unpack8(dsq[nb][mb][pp], d8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc_d[nb][mb][8 * pp + e] + d8[e] * dmul;
                    const int nu = wave * (32 * MLP_NBO) + nb * 32 + 8 * pp;
                    *reinterpret_cast<uint4*>(pn_panel_slot<1024>(pH, mb * 32 + (lane & 31), (nu >> 3) + 2 * hi)) = pn_pack8(v);
                }
        __syncthreads();
    }
    if constexpr (BWD) {
        constexpr int RPW = PN_ROWS / PN_WAVES;
        if (INP || a.ln1_dxn) {
            // the next block's ln_1 backward, row by row (a wave owns whole rows, a lane 8 features: tan_norm.hip's arithmetic):
            // the result is this kernel's dx panel, and goes to HBM once for the weight-gradient launch that reads it later
            const f8 gm = ld8f(a.ln1_g + lane * 8);
            const float rmul = a.ln1_res ? 1.0f : 0.0f;
            f8 dg, db, ds;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dg.v[j] = 0.f; db.v[j] = 0.f; ds.v[j] = 0.f; }
#pragma unroll
            for (int half = 0; half < RPW / 4; ++half) {
                f8 xv[4], dv[4], rv[4];
                float mean[4], rstd[4], s1[4], s2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long row = row0 + wave * RPW + half * 4 + q;
                    xv[q] = ld8(a.ln1_x + row * 512 + lane * 8);
                    if constexpr (INP) dv[q] = ld8(reinterpret_cast<const bf16_t*>(pn_panel_slot<1024>(lds + H_OFF, wave * RPW + half * 4 + q, lane)));
                    else dv[q] = ld8(a.ln1_dxn + row * 512 + lane * 8);
                    rv[q] = ld8((a.ln1_res ? a.ln1_res : a.ln1_x) + row * 512 + lane * 8);       // (unconditional load; weighted below)
                    mean[q] = a.ln1_mean[row]; rstd[q] = a.ln1_rstd[row];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1[q] = 0.f; s2[q] = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xv[q].v[j] = (xv[q].v[j] - mean[q]) * rstd[q];
                        const float d = dv[q].v[j];
                        dv[q].v[j] = d * gm.v[j];
                        s1[q] += dv[q].v[j]; s2[q] += dv[q].v[j] * xv[q].v[j];
                        dg.v[j] += d * xv[q].v[j];
                        db.v[j] += d;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float m1 = wave_sum(s1[q]) * (1.0f / 512), m2 = wave_sum(s2[q]) * (1.0f / 512);
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        o[j] = rstd[q] * (dv[q].v[j] - m1 - xv[q].v[j] * m2) + rv[q].v[j] * rmul;
                        ds.v[j] += o[j];
                    }
                    const uint4 u = pn_pack8(o);
                    const int m = wave * RPW + half * 4 + q;
                    *reinterpret_cast<uint4*>(a.dx_out + (row0 + m) * 512 + lane * 8) = u;
                    *reinterpret_cast<uint4*>(pn_panel_slot<1024>(lds + XN_OFF, m, lane)) = u;
                }
            }
            // column sums over the panel's 64 rows: the eight waves meet in the (still unused) hidden-activation region
            if constexpr (INP) __syncthreads();       // (which held the dxn1 panel: every wave has read its rows)
            float* red = reinterpret_cast<float*>(lds + H_OFF);       // [PN_WAVES][3][512]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(wave * 3 + 0) * 512 + lane * 8 + j] = dg.v[j];
                red[(wave * 3 + 1) * 512 + lane * 8 + j] = db.v[j];
                red[(wave * 3 + 2) * 512 + lane * 8 + j] = ds.v[j];
            }
            __syncthreads();
            for (int idx = tid; idx < 3 * 512; idx += 64 * PN_WAVES) {
                const int which = idx >> 9, cc = idx & 511;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < PN_WAVES; ++w) sum += red[(w * 3 + which) * 512 + cc];
                float* out = which == 0 ? a.g_ln1_g : (which == 1 ? a.g_ln1_b : a.g_dx_colsum);
                if (out) unsafeAtomicAdd(out + cc, sum);
            }
        } else {
            uint4 v[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) v[r] = *reinterpret_cast<const uint4*>(a.dx + (row0 + wave * RPW + r) * 512 + lane * 8);
#pragma unroll
            for (int r = 0; r < RPW; ++r) *reinterpret_cast<uint4*>(pn_panel_slot<1024>(lds + XN_OFF, wave * RPW + r, lane)) = v[r];
        }
    } else
    {
        constexpr int RPW = PN_ROWS / PN_WAVES;
        const f8 g = ld8f(a.ln_g + lane * 8), b = ld8f(a.ln_b + lane * 8);
#pragma unroll
        for (int half = 0; half < RPW / 8; ++half) {
            f8 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if constexpr (OUTP) v[r] = ld8(reinterpret_cast<const bf16_t*>(pn_panel_slot<1024>(lds + XN_OFF, wave * RPW + half * 8 + r, lane)));
                else v[r] = ld8(a.x_mid + (row0 + wave * RPW + half * 8 + r) * 512 + lane * 8);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[r].v[j];
                const float mean = wave_sum(s) * (1.0f / 512);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[r].v[j] -= mean; q += v[r].v[j] * v[r].v[j]; }
                const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512) + a.eps);
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = v[r].v[j] * rstd * g.v[j] + b.v[j];
                const uint4 u = pn_pack8(o);
                const int m = wave * RPW + half * 8 + r;
                if (a.xn2 && first) *reinterpret_cast<uint4*>(a.xn2 + (row0 + m) * 512 + lane * 8) = u;       // (NULL: no backward will follow)
                *reinterpret_cast<uint4*>(pn_panel_slot<1024>(lds + XN_OFF, m, lane)) = u;
                if (lane == 0 && a.mean2 && first) { a.mean2[row0 + m] = mean; a.rstd2[row0 + m] = rstd; }
            }
        }
    }
    __syncthreads();

    f32x16 acc_o[MLP_NBO][2];      // [feature block of the wave's output features][row block]
#pragma unroll
    for (int i = 0; i < MLP_NBO; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc_o[i][j]);
    f32x16 acc_h[MLP_NBH][2];      // [feature block of the wave's hidden features][row block]
    MlpXFrags FA, FB;        // activation fragments of the even / odd steps
    MlpEpiState ES;
    MlpBias32 B32;           // bias of the next chunk: loaded before a slot barrier, consumed right after it (mlp_init_h)
    MlpHPre HP;              // backward: pre-activations of the chunk in the accumulator layout
    MlpBwdEpi BE;
    if constexpr (!BWD) mlp_bias32_load(B32, a.b_fc, c0, wave);
    static_assert(MLP_NBO == 2 && MLP_WFR == 2, "eight waves");
    MlpXAddr XA;
    mlp_xaddr_init(XA, lds, lane);
    MlpCopy CP;
    mlp_copy_init(CP, wave, lane);
    mlp_load_x_fc<0>(FA, XA);
    __builtin_amdgcn_sched_barrier(0);

    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    // the two 16-step phases; the flags are compile-time so that every step is ONE basic block (the scheduler interleaves the
    // epilogue half-unit with the MFMAs only inside a block)
    auto fc_phase = [&](int c, auto has_copy, auto then_proj) __attribute__((always_inline)) {            // c_fc(c) (|| side outputs of chunk c-1)
        constexpr bool COPY = decltype(has_copy)::value, THEN_PROJ = decltype(then_proj)::value;
        const int hb = c & 1;
        pn_static_for<0, 16>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            MlpXFrags& cur = (J & 1) ? FB : FA;
            MlpXFrags& nxt = (J & 1) ? FA : FB;
            if (!(MODE & 4)) {
                if constexpr (J < 15) mlp_load_x_fc<J + 1>(nxt, XA);     // (the next phase's first fragments: after the slot barrier)
                else if constexpr (THEN_PROJ) mlp_load_x_proj<0>(nxt, XA, hb ^ 1);   // (balanced: c_proj(c-1)'s first steps follow in this slot)
            }
            if (!(MODE & 1)) mlp_mma_fc(WQ[J % D], cur, acc_h);
            if constexpr (BWD) {        // pre-activations of THIS chunk, consumed by the epilogue a phase later
                // all four together: each is a cold HBM read in the in-order vmcnt queue in front of the weight ring, and the
                // ring stalls once per batch of them, not once per load
                if constexpr (J == TAN_HPRE_STEP) {
                    mlp_hpre_load<0, 0>(HP, a.h_pre, row0, (c + c0), wave, lane);
                    mlp_hpre_load<1, 0>(HP, a.h_pre, row0, (c + c0), wave, lane);
                    mlp_hpre_load<0, 1>(HP, a.h_pre, row0, (c + c0), wave, lane);
                    mlp_hpre_load<1, 1>(HP, a.h_pre, row0, (c + c0), wave, lane);
                }
            } else if constexpr (COPY) {
                if (!(MODE & (8 | 16))) mlp_copy_step4<J, false>(CP, lds, a.h_pre, row0, c - 1, (c - 1 + c0));
            }
            if (!(MODE & 2)) {      // the tile eight steps on: c_fc(c) J+8, else the first half of the next phase that streams
                const char* src = J + D < 16 ? pfc + (long)((c + c0) * 16 + J + D) * TILE
                                             : (SPLIT ? ppj + (long)(c0 * 16 + J + D - 16) * TILE           // SPLIT: c_proj(c0) is what streams next
                                             : (c == 0 ? pfc + (long)((1) * 16 + J + D - 16) * TILE      // body(0) has no c_proj phase
                                                       : ppj + (long)((c - 1) * 16 + J + D - 16) * TILE));
                mlp_load_w(WQ[J % D], src, wave, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // c_proj(c-1) steps [J0, J1) (k steps of 16 over hidden panel c-1) || epilogue(c) half-units.  UNITS = 2: half-units 2 (J - 8),
    // 2 (J - 8) + 1 at step J (J = 8 .. 15: the second slot of a body); UNITS = 0: none (the first slot, and body(8)).  (Round 2's
    // schedule -- c_fc(c) | c_proj(c-1) with half-unit J at step J -- is in the history: `git log -S"U1_" -- tan_panel.hip`.)
    auto proj_steps = [&](int c, auto has_proj, auto has_epi, auto j0, auto j1, auto units) __attribute__((always_inline)) {
        constexpr bool PROJ = decltype(has_proj)::value, EPI = decltype(has_epi)::value;
        constexpr int J0 = decltype(j0)::value, J1 = decltype(j1)::value, UNITS = decltype(units)::value;
        constexpr bool ARITH = EPI && UNITS > 0 && !(MODE & (8 | 32));
        const int hb = c & 1;
        pn_static_for<J0, J1>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int U0 = UNITS == 2 ? 2 * (J - 8) : 0, U1 = U0 + 1;       // half-units of this step
            MlpXFrags& cur = (J & 1) ? FB : FA;
            MlpXFrags& nxt = (J & 1) ? FA : FB;
            MlpWFrags& W = WQ[J % D];
            if constexpr (PROJ) {
                if (!(MODE & 4)) {
                    if constexpr (J + 1 < J1) mlp_load_x_proj<J + 1>(nxt, XA, hb ^ 1);     // (the next range's first fragments: after its barrier)
                }
                if constexpr (!EPI && !BWD) {       // body(8): the side outputs of chunk 7 under c_proj(7)
                    if (!(MODE & (8 | 16))) mlp_copy_step8<J>(CP, lds, a.h_pre, a.h_act, row0, SPLIT ? 0 : 7, SPLIT ? c0 : 7);
                } else {                    // the activation panel of chunk c-1 (c_proj(c-1) reads it too; rewritten in body c+1)
                    if (!(MODE & (8 | 16))) mlp_copy_step4<J, true>(CP, lds, a.h_act, row0, c - 1, (c - 1 + c0));
                }
            }
            // c_proj: W.f[nb] = output features wave*64 + nb*32 .., one k step; the epilogue pieces sit between the MFMAs.  Neither
            // the MFMAs nor the arithmetic have side effects, so sched_barrier alone does not keep them apart (instruction selection
            // linearises pure nodes freely): empty asm statements tie each piece's results to the accumulator the NEXT MFMA
            // continues (written four MFMAs ago: no hazard padding), which orders piece -> asm -> MFMA -> next piece.
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROJ) { if (!(MODE & 1)) acc_o[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[0], cur.f[0], acc_o[0][0], 0, 0, 0); }
            if constexpr (ARITH && BWD) {
                mlp_bepi_p1<U0>(BE, HP);
                asm volatile("" : "+v"(acc_o[0][1]), "+v"(BE.ce[U0 & 1][0]), "+v"(BE.ce[U0 & 1][1]));
            } else if constexpr (ARITH) {
                mlp_epi_p1<U0>(ES, acc_h);
                asm volatile("" : "+v"(acc_o[0][1]), "+v"(ES.ce[U0 & 1][0]), "+v"(ES.ce[U0 & 1][1]));
            }
            if constexpr (PROJ) { if (!(MODE & 1)) acc_o[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[0], cur.f[1], acc_o[0][1], 0, 0, 0); }
            if constexpr (ARITH && BWD) {
                mlp_bepi_p2<U0>(BE);
                if constexpr (UNITS == 2) mlp_bepi_p1<U1>(BE, HP);
                asm volatile("" : "+v"(acc_o[1][0]), "+v"(BE.ce[0][0]), "+v"(BE.ce[0][1]), "+v"(BE.ce[1][0]), "+v"(BE.ce[1][1]));
            } else if constexpr (ARITH) {
                mlp_epi_p2<U0>(ES);
                if constexpr (UNITS == 2) mlp_epi_p1<U1>(ES, acc_h);
                asm volatile("" : "+v"(acc_o[1][0]), "+v"(ES.ce[0][0]), "+v"(ES.ce[0][1]), "+v"(ES.ce[1][0]), "+v"(ES.ce[1][1]));
            }
            if constexpr (PROJ) { if (!(MODE & 1)) acc_o[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[1], cur.f[0], acc_o[1][0], 0, 0, 0); }
            if constexpr (ARITH && BWD) {
                mlp_bepi_p3<U0>(BE, acc_h, lds, hb, wave, lane);
                if constexpr (UNITS == 2) mlp_bepi_p2<U1>(BE);
                asm volatile("" : "+v"(acc_o[1][1]), "+v"(BE.w[U0 & 1][(U0 >> 1) & 3]), "+v"(BE.cs[2 * (U0 >> 1)]), "+v"(BE.cs[2 * (U0 >> 1) + 1]),
                             "+v"(BE.ce[1][0]), "+v"(BE.ce[1][1]));
            } else if constexpr (ARITH) {
                mlp_epi_p3<U0>(ES, acc_h, lds, hb, wave, lane);
                if constexpr (UNITS == 2) mlp_epi_p2<U1>(ES);
                asm volatile("" : "+v"(acc_o[1][1]), "+v"(ES.pre[U0 & 3]), "+v"(ES.act[U0 & 3]), "+v"(ES.ce[1][0]), "+v"(ES.ce[1][1]));
            }
            if constexpr (PROJ) {
                if (!(MODE & 1)) acc_o[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[1], cur.f[1], acc_o[1][1], 0, 0, 0);
            }
            if constexpr (ARITH && UNITS == 2) {      // the second half-unit's last piece: behind the step's last MFMA
                if constexpr (BWD) {
                    mlp_bepi_p3<U1>(BE, acc_h, lds, hb, wave, lane);
                    asm volatile("" : "+v"(BE.w[U1 & 1][(U1 >> 1) & 3]), "+v"(BE.cs[2 * (U1 >> 1)]), "+v"(BE.cs[2 * (U1 >> 1) + 1]));
                } else {
                    mlp_epi_p3<U1>(ES, acc_h, lds, hb, wave, lane);
                    asm volatile("" : "+v"(ES.pre[U1 & 3]), "+v"(ES.act[U1 & 3]));
                }
            }
            if constexpr (PROJ) {
                if (!(MODE & 2)) {  // c_proj(c-1) J+D, else the first steps of the next body's first phase
                    if constexpr (J + D < 16) mlp_load_w(W, ppj + (long)((c - 1 + c0) * 16 + J + D) * TILE, wave, lane);
                    else if constexpr (EPI)     // c <= 7: body(c+1) starts with c_fc(c+1), body(8) with c_proj(7)
                        mlp_load_w(W, c < 7 ? pfc + (long)((c + 1) * 16 + J + D - 16) * TILE : ppj + (long)((7) * 16 + J + D - 16) * TILE,
                                   wave, lane);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // One barrier per 16-step slot.  The two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the SAME phase sequence
    // one slot apart, so that in every slot one wave of a SIMD is in a pure-MFMA c_fc phase while the other interleaves c_proj
    // MFMAs with the VALU-heavy chunk epilogue and the side-output stores: in lock step (the first version) both waves of a SIMD
    // did their epilogues at the same time and the matrix pipe idled -- MFMA, epilogue arithmetic and store stalls simply added up
    // (40 + 22 + 35 us, tools/lab/mlp_lab.py).  Dependencies with the skew: c_proj(c-1) of the early group runs in slot 2c+1 and needs
    // the late group's epilogue(c-1), slot 2c; the late group's c_proj(c-1), slot 2c+2, needs the early group's, slot 2c-1; a hidden
    // buffer is rewritten two bodies later (slots 2c+5 / 2c+6), after its last readers (slots 2c+3 / 2c+4).  The pre-activation
    // panel and the copy-out of both panels are split by group halves, so they never cross groups.  The last phase, c_proj(7), has
    // to wait for the late group's epilogue(7): the early group idles in slot 16.
    // lab instrumentation (MODE & 64): workgroup 0 records the shader clock at phase boundaries
    int tick_i = 0;
    auto tick = [&]() __attribute__((always_inline)) {
#ifdef TAN_PANEL_LAB
        if constexpr ((MODE & 64) != 0) {
            const long long t = __builtin_readcyclecounter();
            if (blockIdx.x == 0 && lane == 0 && g_panel_dbg) g_panel_dbg[wave * 64 + tick_i] = t;
            ++tick_i;
        }
#endif
    };
    auto slot_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto start_h = [&]() __attribute__((always_inline)) {       // a chunk's accumulator starts from the bias (forward) / zero
        if constexpr (BWD) {
            acc_zero(acc_h[0][0]);
            acc_zero(acc_h[0][1]);
        } else {
            mlp_init_h(acc_h, B32, hi);
        }
    };
    auto after_epi = [&](int c) __attribute__((always_inline)) {   // after the epilogue of chunk c, before the slot barrier
        if constexpr (BWD) {       // c_fc bias gradient: column sums of the wave's 32 features over the panel's 64 rows
            const float tot = pn_colsum16(BE.cs, lane);
            if (!(lane & 2)) unsafeAtomicAdd(a.g_b_fc + (c + c0) * 256 + wave * 32 + 16 * hi + pn_colsum16_index(lane), tot);
        } else if constexpr (!SPLIT) {
            mlp_bias32_load(B32, a.b_fc, (min(c + 1, 7)), wave);
        }
    };
    tick();
    if (grp) slot_barrier();
    // body(0) and body(8) are peeled as straight-line code around the loop: as if / else arms INSIDE the loop every extra variant of
    // a phase cost ~300 spilled registers at the joins
    using C0 = std::integral_constant<int, 0>;
    using C8 = std::integral_constant<int, 8>;
    using C16 = std::integral_constant<int, 16>;
    using U0_ = std::integral_constant<int, 0>;
    using U2_ = std::integral_constant<int, 2>;
    // BALANCED SLOTS (round 5).  The round-2 schedule put c_fc(c) -- 64 MFMAs -- in one slot and c_proj(c-1) + epilogue(c) -- 64 MFMAs
    // interleaved with ~2 k cycles of QuickGELU / pack / LDS writes in the same wave's instruction stream -- in the other; with the two
    // wave groups one slot apart, every slot lasted as long as its epilogue wave (5.2 k cycles without any memory traffic) while the
    // c_fc wave of the same SIMD sat at the slot barrier for half of it: tools/lab/mlp_lab.py, `MODE` 22 / 86, DESIGN.md section 6.
    // Now the slot boundary sits behind the first HALF of c_proj(c-1): k steps 0 .. 7 contract over the EARLY group's half of hidden
    // panel c-1, which that group finished a whole slot ago, so they need no barrier behind c_fc(c):
    //     slot X: c_fc(c) + c_proj(c-1)[k 0..7]   96 MFMAs           slot Y: c_proj(c-1)[k 8..15] || epilogue(c), two half-units a step
    // The tile order of the weight ring, the hidden-panel buffers and the side-output copy steps are what they were (a step keeps its
    // index J); the epilogue moves entirely into slot Y because the (single) pre-activation panel is still being copied out by the
    // group's other waves during c_fc(c).
    start_h();
    tick(); fc_phase(0, F_{}, F_{}); tick();
    slot_barrier();
    tick(); proj_steps(0, F_{}, T_{}, C8{}, C16{}, U2_{}); tick();
    after_epi(0);
    slot_barrier();
    if constexpr (!SPLIT) {
    mlp_load_x_fc<0>(FA, XA);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 1; c < 8; ++c) {
        start_h();
        tick(); fc_phase(c, T_{}, T_{});
        proj_steps(c, T_{}, T_{}, C0{}, C8{}, U0_{}); tick();
        slot_barrier();
        mlp_load_x_proj<8>(FA, XA, (c & 1) ^ 1);                    // the late group's half of hidden panel c-1
        __builtin_amdgcn_sched_barrier(0);
        tick(); proj_steps(c, T_{}, T_{}, C8{}, C16{}, U2_{}); tick();
        after_epi(c);
        slot_barrier();
        if (c < 7) mlp_load_x_fc<0>(FA, XA);                        // fragments of the next body's first step
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    // the last c_proj phase: body(8) = c_proj(7) over hidden panel 7 & 1; SPLIT: "body(1)" = c_proj(c0) over hidden panel 0
    constexpr int CL = SPLIT ? 1 : 8, HBL = SPLIT ? 0 : 1;
    mlp_load_x_proj<0>(FA, XA, HBL);
    __builtin_amdgcn_sched_barrier(0);
    tick(); proj_steps(CL, T_{}, F_{}, C0{}, C8{}, U0_{});
    slot_barrier();
    mlp_load_x_proj<8>(FA, XA, HBL);
    __builtin_amdgcn_sched_barrier(0);
    proj_steps(CL, T_{}, F_{}, C8{}, C16{}, U0_{}); tick();
    if (!grp) slot_barrier();
    tick();
    if (MODE & 1) {          // stream-only experiment: keep the loaded fragments alive
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < MLP_WFR; ++j) asm volatile("" ::"v"(WQ[i].f[j]));
    }

    __syncthreads();         // every wave is done with the activation panels
    if constexpr (SPLIT) {
        // the workgroup's term of the [64 x 512] sum over the hidden chunks -> plane c0: a lane owns rows mb * 32 + (lane & 31) and the
        // 16 consecutive features wave * 64 + nb * 32 + 16 hi + r (four 16-byte stores)
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float* dst = a.part + c0 * a.part_plane + (row0 + mb * 32 + (lane & 31)) * 512 + wave * (32 * MLP_NBO) + nb * 32 + 16 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(acc_o[nb][mb][4 * q], acc_o[nb][mb][4 * q + 1], acc_o[nb][mb][4 * q + 2],
                                                                          acc_o[nb][mb][4 * q + 3]);
            }
        return;
    }
    if constexpr (!BWD) {
    // ---- epilogue: + bias + residual -> x_out; LayerNorm of the (bf16-rounded) output row -> xn_next.  Both leave through LDS
    // panels as whole 1-KiB rows (pn_panel_copy_out); the input panel's space takes x_out, the hidden panels' space xn_next.
    char* xo_panel = lds + MLP_XN_OFF;
    char* xn_panel = lds + MLP_H_OFF;
    float* red = reinterpret_cast<float*>(lds + MLP_PRE_OFF);     // [2][PN_WAVES][64 rows]
    float xr[MLP_NBO][2][2][8];    // [nb][mb][p][e], rounded like the stored x_out
    float rs[2] = {0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < MLP_NBO; ++nb) {
        const int nbase = wave * (32 * MLP_NBO) + nb * 32;
        pn_cfptr_t bp = (pn_cfptr_t)(uintptr_t)(a.b_proj) + nbase;
        uint4 resq[2][2];    // this feature block's residual chunks: four loads in flight
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                resq[mb][p] = *reinterpret_cast<const uint4*>(a.x_mid + (row0 + mb * 32 + (lane & 31)) * 512 + nbase + (2 * hi + p) * 8);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int m = mb * 32 + (lane & 31);       // the lane's registers: features nbase + 16 hi + r of row m
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int ch = 2 * hi + p;
                float bias[8], res[8];
                pn_uniform8(bp + 8 * p, bp + 16 + 8 * p, hi, bias);
                pn_unpack8(resq[mb][p], res);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc_o[nb][mb][8 * p + e] + bias[e] + res[e];
                const uint4 u = pn_pack8(v);
                *reinterpret_cast<uint4*>(pn_panel_slot<1024>(xo_panel, m, (nbase >> 3) + ch)) = u;
                pn_unpack8(u, xr[nb][mb][p]);
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[mb] += xr[nb][mb][p][e];
            }
        }
    }
    if (a.xn_next) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float s = pn_half_sum(rs[mb]);
            if (lane < 32) red[wave * 64 + mb * 32 + lane] = s;
        }
    }
    __syncthreads();
    pn_panel_copy_out<1024>(xo_panel, a.x_out + row0 * 512, 512, wave, lane);
    if (a.xn_next) {
        float mean[2], rstd[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < PN_WAVES; ++w) s += red[w * 64 + mb * 32 + (lane & 31)];
            mean[mb] = s * (1.0f / 512);
            float q = 0.f;
#pragma unroll
            for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = xr[nb][mb][p][e] - mean[mb]; xr[nb][mb][p][e] = d; q += d * d; }
            q = pn_half_sum(q);
            if (lane < 32) red[PN_WAVES * 64 + wave * 64 + mb * 32 + lane] = q;
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < PN_WAVES; ++w) q += red[PN_WAVES * 64 + w * 64 + mb * 32 + (lane & 31)];
            rstd[mb] = rsqrtf(q * (1.0f / 512) + a.eps);
            const int m = mb * 32 + (lane & 31);
            if (wave == 0 && lane < 32) { a.nmean[row0 + m] = mean[mb]; a.nrstd[row0 + m] = rstd[mb]; }
        }
#pragma unroll
        for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int nu = wave * (32 * MLP_NBO) + nb * 32 + 8 * p;          // wave-uniform: gamma / beta come through scalar loads
                pn_cfptr_t gp = (pn_cfptr_t)(uintptr_t)(a.nln_g) + nu;
                pn_cfptr_t bp = (pn_cfptr_t)(uintptr_t)(a.nln_b) + nu;
                float g[8], b[8];
                pn_uniform8(gp, gp + 16, hi, g);
                pn_uniform8(bp, bp + 16, hi, b);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = xr[nb][mb][p][e] * rstd[mb] * g[e] + b[e];
                    *reinterpret_cast<uint4*>(pn_panel_slot<1024>(xn_panel, mb * 32 + (lane & 31), (nu >> 3) + 2 * hi)) = pn_pack8(y);
                }
            }
        __syncthreads();
        pn_panel_copy_out<1024>(xn_panel, a.xn_next + row0 * 512, 512, wave, lane);
    }

    } else {
        PnLnBwd P;
        P.x = a.x_mid; P.mean = a.mean2; P.rstd = a.rstd2; P.gamma = a.ln_g; P.dadd = nullptr;
        P.g_gamma = a.g_ln_g; P.g_beta = a.g_ln_b; P.g_colsum = a.g_b_out;
        pn_ln_bwd_epilogue(acc_o, lds, P, row0, wave, lane);
        if (a.pwt_out) {
            // ---- tail: d_o = dx2 W_out, the out-projection's dX GEMM (a [64 x 512] x [512 x 512] row-local product on the panel that
            // is sitting in LDS: it was a launch of its own, 8192 x 512 x 512 in 128 x 128 tiles of 8 K-steps -- 17 us of which 9 are
            // launch, first-tile latency and drain).  "c_proj-like": the wave owns 64 output features, 32 K-steps of 16; W_out^T
            // streams through the (idle) weight ring, the dx2 rows leave for HBM under it.
            pn_static_for<0, D>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                mlp_load_w(WQ[J], a.pwt_out + (long)J * TILE, wave, lane);
            });
            pn_panel_copy_out<1024>(lds + MLP_XN_OFF, a.dx2 + row0 * 512, 512, wave, lane);
            f32x16 acc_d[MLP_NBO][2];
#pragma unroll
            for (int i = 0; i < MLP_NBO; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc_zero(acc_d[i][j]);
            bf16x8 xt[2][2];                   // activation fragments read one step ahead of their MFMAs
            mlp_load_x_k16<0, 0>(xt[0], XA);
            pn_static_for<0, 32>([&](auto jc) {
                constexpr int KT = decltype(jc)::value;
                MlpWFrags& W = WQ[KT % D];
                if constexpr (KT < 31) mlp_load_x_k16<KT + 1, 0>(xt[(KT + 1) & 1], XA);
                const bf16x8 x0 = xt[KT & 1][0], x1 = xt[KT & 1][1];
#pragma unroll
                for (int nb = 0; nb < MLP_NBO; ++nb) {
                    acc_d[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x0, acc_d[nb][0], 0, 0, 0);
                    acc_d[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], x1, acc_d[nb][1], 0, 0, 0);
                }
                if constexpr (KT + D < 32) mlp_load_w(W, a.pwt_out + (long)(KT + D) * TILE, wave, lane);
                __builtin_amdgcn_sched_barrier(0);
            });
            char* do_panel = lds + MLP_H_OFF;          // (the hidden panels are idle since the last c_proj-like phase)
#pragma unroll
            for (int nb = 0; nb < MLP_NBO; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc_d[nb][mb][8 * p + e];
                        const int nu = wave * (32 * MLP_NBO) + nb * 32 + 8 * p;
                        *reinterpret_cast<uint4*>(pn_panel_slot<1024>(do_panel, mb * 32 + (lane & 31), (nu >> 3) + 2 * hi)) = pn_pack8(v);
                    }
            __syncthreads();
            pn_panel_copy_out<1024>(do_panel, a.d_o + row0 * 512, 512, wave, lane);
        } else {
            pn_panel_copy_out<1024>(lds + MLP_XN_OFF, a.dx2 + row0 * 512, 512, wave, lane);
        }
    }
}


// ---- row epilogues of the SPLIT kernels (one wave per row, a lane owns 8 consecutive features): what the whole-panel kernel does from
// its accumulators, here from the eight f32 planes the chunk workgroups of a panel left in `part`, added in chunk order (deterministic).
__device__ __forceinline__ f8 split_sum8(const float* __restrict__ part, long plane, long row, int lane) {
    f8 acc = ld8f(part + row * 512 + lane * 8);
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        const f8 t = ld8f(part + c * plane + row * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] += t.v[j];
    }
    return acc;
}
// forward: x_out = x_mid + (sum + b_proj); xn_next = LayerNorm(x_out rounded to bf16) (the whole-panel kernel's arithmetic, in its order)
__global__ __launch_bounds__(256) void mlp_split_finish_fwd_kernel(const float* __restrict__ part, long plane, const bf16_t* __restrict__ x_mid,
                                                                   const float* __restrict__ b_proj, bf16_t* __restrict__ x_out,
                                                                   const float* __restrict__ nln_g, const float* __restrict__ nln_b,
                                                                   bf16_t* __restrict__ xn_next, float* __restrict__ nmean,
                                                                   float* __restrict__ nrstd, long rows, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const f8 acc = split_sum8(part, plane, row, lane);
    const f8 res = ld8(x_mid + row * 512 + lane * 8), bias = ld8f(b_proj + lane * 8);
    f8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] = acc.v[j] + bias.v[j] + res.v[j];
    st8(x_out + row * 512 + lane * 8, v);
    if (!xn_next) return;
    uint4 u;
    u.x = f2bf2(v.v[0], v.v[1]); u.y = f2bf2(v.v[2], v.v[3]); u.z = f2bf2(v.v[4], v.v[5]); u.w = f2bf2(v.v[6], v.v[7]);
    float xr[8];
    pn_unpack8(u, xr);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xr[j];
    const float mean = wave_sum(s) * (1.0f / 512);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { xr[j] -= mean; q += xr[j] * xr[j]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512) + eps);
    const f8 g = ld8f(nln_g + lane * 8), b = ld8f(nln_b + lane * 8);
    f8 y;
#pragma unroll
    for (int j = 0; j < 8; ++j) y.v[j] = xr[j] * rstd * g.v[j] + b.v[j];
    st8(xn_next + row * 512 + lane * 8, y);
    if (lane == 0) { nmean[row] = mean; nrstd[row] = rstd; }
}

// backward: dx2 = dx + LayerNorm-2 backward of dxn (the f32 sums); g_ln_g += colsum(dxn o xhat), g_ln_b += colsum(dxn), g_b_out +=
// colsum(dx2) -- 16 rows per workgroup, one atomic per column and workgroup
__global__ __launch_bounds__(256) void mlp_split_finish_bwd_kernel(const float* __restrict__ part, long plane, const bf16_t* __restrict__ x_mid,
                                                                   const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                                   const float* __restrict__ ln_g, const bf16_t* __restrict__ dx,
                                                                   bf16_t* __restrict__ dx2, float* __restrict__ g_ln_g,
                                                                   float* __restrict__ g_ln_b, float* __restrict__ g_b_out, long rows) {
    __shared__ float red[4][3][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f8 gam = ld8f(ln_g + lane * 8);
    float dg[8], db[8], ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; ds[j] = 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = (long)blockIdx.x * 16 + wave * 4 + r;
        if (row >= rows) continue;
        const f8 dy = split_sum8(part, plane, row, lane);
        const f8 xv = ld8(x_mid + row * 512 + lane * 8), rv = ld8(dx + row * 512 + lane * 8);
        const float mean = mean2[row], rstd = rstd2[row];
        float xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xh[j] = (xv.v[j] - mean) * rstd;
            g[j] = dy.v[j] * gam.v[j];
            s1 += g[j];
            s2 += g[j] * xh[j];
            dg[j] += dy.v[j] * xh[j];
            db[j] += dy.v[j];
        }
        const float m1 = wave_sum(s1) * (1.0f / 512), m2 = wave_sum(s2) * (1.0f / 512);
        f8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o.v[j] = rstd * (g[j] - m1 - xh[j] * m2) + rv.v[j];
            ds[j] += o.v[j];
        }
        st8(dx2 + row * 512 + lane * 8, o);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[wave][0][lane * 8 + j] = dg[j];
        red[wave][1][lane * 8 + j] = db[j];
        red[wave][2][lane * 8 + j] = ds[j];
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 512; idx += 256) {
        const int which = idx >> 9, cc = idx & 511;
        const float sum = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
        float* out = which == 0 ? g_ln_g : (which == 1 ? g_ln_b : g_b_out);
        unsafeAtomicAdd(out + cc, sum);
    }
}

}  // namespace tal

using namespace tal;

extern "C" int tan_panel_waves(void) { return PN_WAVES; }

extern "C" int tan_mlp_split_chunks(void) { return 8; }

extern "C" int tan_mlp_fwd_split(const tan_mlp_desc* d, float* part, void* stream) {
    TAN_REQUIRE(d && part && d->x_mid && d->ln_g && d->ln_b && d->pw_fc && d->pw_proj && d->b_fc && d->b_proj && d->x_out);
    TAN_REQUIRE((d->h_pre != nullptr) == (d->h_act != nullptr) && (d->mean2 != nullptr) == (d->rstd2 != nullptr));
    TAN_REQUIRE(d->rows > 0 && d->rows % PN_ROWS == 0 && d->C == 512 && d->FF == 2048 && !d->pw_out && d->variant == 0);
    TAN_REQUIRE(!d->xn_next || (d->nln_g && d->nln_b && d->nmean && d->nrstd));
    MlpFwdArgs a{};
    a.x_mid = (const bf16_t*)d->x_mid; a.ln_g = d->ln_g; a.ln_b = d->ln_b;
    a.pw_fc = (const char*)d->pw_fc; a.pw_proj = (const char*)d->pw_proj; a.b_fc = d->b_fc; a.b_proj = d->b_proj;
    a.xn2 = (bf16_t*)d->xn2; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.h_pre = (bf16_t*)d->h_pre; a.h_act = (bf16_t*)d->h_act; a.x_out = (bf16_t*)d->x_out;
    a.eps = d->eps; a.part = part; a.part_plane = d->rows * 512;
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(d->rows / PN_ROWS), 8);
    const int rec = prof_begin(st, TAN_PROF_PANEL, 2.0 * d->rows * 512.0 * 2048.0 * 2.0);
    if (!d->h_pre) hipLaunchKernelGGL((mlp_panel_kernel<4096 | 16, MlpFwdArgs>), grid, dim3(64 * PN_WAVES), 0, st, a);
    else hipLaunchKernelGGL((mlp_panel_kernel<4096, MlpFwdArgs>), grid, dim3(64 * PN_WAVES), 0, st, a);
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_split_finish_fwd_kernel, dim3(cdiv(d->rows, 4)), dim3(256), 0, st, part, d->rows * 512, (const bf16_t*)d->x_mid, d->b_proj,
                       (bf16_t*)d->x_out, d->nln_g, d->nln_b, (bf16_t*)d->xn_next, d->nmean, d->nrstd, d->rows, d->eps);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_mlp_bwd_split(const tan_mlp_bwd_desc* d, float* part, void* stream) {
    TAN_REQUIRE(d && part && d->rows > 0 && d->rows % PN_ROWS == 0 && d->C == 512 && d->FF == 2048);
    TAN_REQUIRE(d->dx && d->h_pre && d->x_mid && d->mean2 && d->rstd2 && d->ln_g && d->pwt_proj && d->pwt_fc && d->dh && d->dx2);
    TAN_REQUIRE(d->g_b_fc && d->g_ln_g && d->g_ln_b && d->g_b_out && !d->ln1_dxn && !d->pwt_in && !d->pwt_out);
    MlpBwdArgs a{};
    a.dx = (const bf16_t*)d->dx; a.h_pre = (const bf16_t*)d->h_pre; a.x_mid = (const bf16_t*)d->x_mid;
    a.mean2 = d->mean2; a.rstd2 = d->rstd2; a.ln_g = d->ln_g;
    a.pw_fc = (const char*)d->pwt_proj; a.pw_proj = (const char*)d->pwt_fc;
    a.h_act = (bf16_t*)d->dh; a.dx2 = (bf16_t*)d->dx2;
    a.g_b_fc = d->g_b_fc; a.g_ln_g = d->g_ln_g; a.g_ln_b = d->g_ln_b; a.g_b_out = d->g_b_out;
    a.part = part; a.part_plane = d->rows * 512;
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(d->rows / PN_ROWS), 8);
    const int rec = prof_begin(st, TAN_PROF_PANEL, 2.0 * d->rows * 512.0 * 2048.0 * 2.0);
    hipLaunchKernelGGL((mlp_panel_kernel<4096, MlpBwdArgs>), grid, dim3(64 * PN_WAVES), 0, st, a);
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_split_finish_bwd_kernel, dim3(cdiv(d->rows, 16)), dim3(256), 0, st, part, d->rows * 512, (const bf16_t*)d->x_mid, d->mean2,
                       d->rstd2, d->ln_g, (const bf16_t*)d->dx, (bf16_t*)d->dx2, d->g_ln_g, d->g_ln_b, d->g_b_out, d->rows);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_pack_weights(const void* src, void* dst, const tan_pack_entry* table, int n, int max_tiles, void* stream) {
    TAN_REQUIRE(src && dst && table && n > 0 && max_tiles > 0);
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(max_tiles < 64 ? max_tiles : 64, n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (bf16_t*)dst, table);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_mlp_fwd(const tan_mlp_desc* d, void* stream) {
    TAN_REQUIRE(d && d->x_mid && d->ln_g && d->ln_b && d->pw_fc && d->pw_proj && d->b_fc && d->b_proj && d->x_out);
    TAN_REQUIRE((d->h_pre != nullptr) == (d->h_act != nullptr));           // the two side outputs: both or neither
    TAN_REQUIRE((d->mean2 != nullptr) == (d->rstd2 != nullptr));
    const bool no_side = !d->h_pre;
    TAN_REQUIRE(d->rows > 0 && d->rows % PN_ROWS == 0 && d->C == 512 && d->FF == 2048);
    TAN_REQUIRE(!d->xn_next || (d->nln_g && d->nln_b && d->nmean && d->nrstd));
    MlpFwdArgs a;
    a.part = nullptr; a.part_plane = 0;
    a.x_mid = (const bf16_t*)d->x_mid; a.ln_g = d->ln_g; a.ln_b = d->ln_b;
    a.pw_fc = (const char*)d->pw_fc; a.pw_proj = (const char*)d->pw_proj; a.b_fc = d->b_fc; a.b_proj = d->b_proj;
    a.xn2 = (bf16_t*)d->xn2; a.mean2 = d->mean2; a.rstd2 = d->rstd2;
    a.h_pre = (bf16_t*)d->h_pre; a.h_act = (bf16_t*)d->h_act; a.x_out = (bf16_t*)d->x_out;
    a.nln_g = d->nln_g; a.nln_b = d->nln_b; a.xn_next = (bf16_t*)d->xn_next; a.nmean = d->nmean; a.nrstd = d->nrstd;
    a.eps = d->eps;
    a.attn_o = (const bf16_t*)d->attn_o; a.pw_out = (const char*)d->pw_out; a.b_out = d->b_out; a.x_in = (const bf16_t*)d->x_in;
    a.x_mid_w = (bf16_t*)d->x_mid;
    const bool outp = d->pw_out != nullptr;
    if (outp) TAN_REQUIRE(d->attn_o && d->b_out && d->x_in && d->variant == 0);
    const dim3 grid((unsigned)(d->rows / PN_ROWS));
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_PANEL, 2.0 * d->rows * 512.0 * 2048.0 * 2.0 + (outp ? 2.0 * d->rows * 512.0 * 512.0 : 0.0));
#define TAN_MLP_LAUNCH(M) hipLaunchKernelGGL((mlp_panel_kernel<M, MlpFwdArgs>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a)
    // inference / the EMA target's forward: the side outputs of the chunk epilogue (h_pre, h_act: 2 x 4 KiB per row... 67 MB per
    // 8192 rows) are never read -- the instantiation without their copy-out (everything else identical)
    if (outp) {
        if (no_side) TAN_MLP_LAUNCH(2048 | 16);
        else TAN_MLP_LAUNCH(2048);
    } else if (no_side && d->variant == 0) {
        TAN_MLP_LAUNCH(16);
    } else
    switch (d->variant) {
#ifdef TAN_PANEL_LAB
        case 1: TAN_MLP_LAUNCH(1); break;
        case 87: TAN_MLP_LAUNCH(87); break;
        case 86: TAN_MLP_LAUNCH(86); break;
        case 19: TAN_MLP_LAUNCH(19); break;
        case 23: TAN_MLP_LAUNCH(23); break;
        case 7: TAN_MLP_LAUNCH(7); break;
        case 4: TAN_MLP_LAUNCH(4); break;
        case 6: TAN_MLP_LAUNCH(6); break;
        case 22: TAN_MLP_LAUNCH(22); break;
        case 20: TAN_MLP_LAUNCH(20); break;
        case 17: TAN_MLP_LAUNCH(17); break;
        case 18: TAN_MLP_LAUNCH(18); break;
        case 2: TAN_MLP_LAUNCH(2); break;
        case 8: TAN_MLP_LAUNCH(8); break;
        case 32: TAN_MLP_LAUNCH(32); break;
        case 10: TAN_MLP_LAUNCH(10); break;
        case 72: TAN_MLP_LAUNCH(72); break;
        case 96: TAN_MLP_LAUNCH(96); break;
        case 16: TAN_MLP_LAUNCH(16); break;
        case 64: TAN_MLP_LAUNCH(64); break;
        case 80: TAN_MLP_LAUNCH(80); break;
        case 256: TAN_MLP_LAUNCH(256); break;
#endif
        default: TAN_MLP_LAUNCH(0);
    }
#undef TAN_MLP_LAUNCH
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_mlp_bwd(const tan_mlp_bwd_desc* d, void* stream) {
    TAN_REQUIRE(d && d->rows > 0 && d->rows % PN_ROWS == 0 && d->C == 512 && d->FF == 2048);
    TAN_REQUIRE((d->dx || d->ln1_dxn || d->pwt_in) && d->h_pre && d->x_mid && d->mean2 && d->rstd2 && d->ln_g && d->pwt_proj && d->pwt_fc && d->dh && d->dx2);
    TAN_REQUIRE(d->g_b_fc && d->g_ln_g && d->g_ln_b && d->g_b_out);
    if (d->ln1_dxn || d->pwt_in) TAN_REQUIRE(d->ln1_x && d->ln1_mean && d->ln1_rstd && d->ln1_g && d->dx_out);
    TAN_REQUIRE((d->pwt_in != nullptr) == (d->dqkv != nullptr) && !(d->pwt_in && d->ln1_dxn));
    MlpBwdArgs a;
    a.part = nullptr; a.part_plane = 0;
    a.ln1_dxn = (const bf16_t*)d->ln1_dxn; a.ln1_x = (const bf16_t*)d->ln1_x; a.ln1_res = (const bf16_t*)d->ln1_res;
    a.ln1_mean = d->ln1_mean; a.ln1_rstd = d->ln1_rstd; a.ln1_g = d->ln1_g;
    a.g_ln1_g = d->g_ln1_g; a.g_ln1_b = d->g_ln1_b; a.g_dx_colsum = d->g_dx_colsum; a.dx_out = (bf16_t*)d->dx_out;
    a.dx = (const bf16_t*)d->dx; a.h_pre = (const bf16_t*)d->h_pre; a.x_mid = (const bf16_t*)d->x_mid;
    a.mean2 = d->mean2; a.rstd2 = d->rstd2; a.ln_g = d->ln_g;
    a.pw_fc = (const char*)d->pwt_proj; a.pw_proj = (const char*)d->pwt_fc;
    a.h_act = (bf16_t*)d->dh; a.dx2 = (bf16_t*)d->dx2;
    a.g_b_fc = d->g_b_fc; a.g_ln_g = d->g_ln_g; a.g_ln_b = d->g_ln_b; a.g_b_out = d->g_b_out;
    TAN_REQUIRE((d->pwt_out != nullptr) == (d->d_o != nullptr));
    a.pwt_out = (const char*)d->pwt_out; a.d_o = (bf16_t*)d->d_o;
    a.dqkv = (const bf16_t*)d->dqkv; a.pwt_in = (const char*)d->pwt_in; a.dstage = (const bf16_t*)d->dstage;
    const dim3 grid((unsigned)(d->rows / PN_ROWS));
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_PANEL, 2.0 * d->rows * 512.0 * 2048.0 * 2.0 + (d->pwt_out ? 2.0 * d->rows * 512.0 * 512.0 : 0.0) +
                                                                         (d->pwt_in ? 2.0 * d->rows * 1536.0 * 512.0 : 0.0));
#ifdef TAN_PANEL_LAB
    if (getenv("TAN_PANEL_LAB_CLOCKS")) hipLaunchKernelGGL((mlp_panel_kernel<64, MlpBwdArgs>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a);
    else
#endif
    if (d->pwt_in) hipLaunchKernelGGL((mlp_panel_kernel<512, MlpBwdArgs>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_panel_kernel<0, MlpBwdArgs>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a);
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}


#ifdef TAN_PANEL_LAB
extern "C" int tan_panel_lab_set_dbg(void* p) {
    long long* q = (long long*)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_panel_dbg), &q, sizeof(q));
}
#endif
