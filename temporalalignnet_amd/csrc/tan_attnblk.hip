// Attention branch of a pre-LN block (model/tfm_model.py:30-36: x + out_proj(MHA(LN1(x)))) as ONE launch per direction.
//
// One workgroup (8 waves) = one VIDEO: attention never crosses videos, so the whole branch is row-panel local.  The panel of the
// LN1 output (L <= 80 rows, 1 KiB each) stays in LDS; the in_proj / out_proj weights stream past it from L2 straight into
// registers (fragment-major packed images, tan_pack_weights), exactly like the row-panel MLP kernels (tan_panel.hip):
//
//   for the 4 head pairs hp:
//     GEMM-a   q|k|v of heads 2hp, 2hp+1 = xn1 W_in[hp]^T + b      384 features x L rows, K = 512      (v_mfma_f32_16x16x32_bf16)
//              -> six [LP][64] head images in LDS (tan_attn_img.h layout; also the saved qkv rows for backward)
//     attention per (head, 32 queries) on one wave each: S^T = K Q^T, softmax in registers, O^T = V^T P^T  (tan_attn.hip's short
//              kernel, verbatim); O overwrites the wave's own q rows of the image
//     GEMM-b   acc += O[:, hp] W_out[:, hp]^T                      512 features x L rows, K = 128      (v_mfma_f32_32x32x16_bf16)
//   x_mid = x_in + acc + b_out
//
// Replaces three launches (in_proj GEMM, attention, out_proj GEMM) and the HBM round trip of qkv and attn_o between them; in the
// no-grad forward (EMA target, evaluation) qkv / attn_o / lse are not written at all.
//
// Why 16x16x32 for GEMM-a: a head pair is 24 blocks of 16 features = 3 per wave for 8 waves (12 blocks of 32 do not divide),
// every weight fragment is used for all row blocks, and L = 80 (T = 64 frames + 16 sentences, the joint stack of the headline
// configuration) is exactly 5 row blocks of 16 -- no padding work.  GEMM-b keeps the 32x32x16 tiling and the packer's feature
// permutation of the MLP kernels (a lane owns 16 consecutive output features of one row: 16-byte epilogue).
#include "tan_panel.h"
#include "tan_attn_img.h"

namespace tal {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct AbFwdArgs {
    const bf16_t* xn1; const bf16_t* x_in; const unsigned char* keypad;
    const char* pw_qkv;         // packed in_proj weight, qkv16 format (pack_tiles_kernel, TN = 384)
    const char* pw_out;         // packed out_proj weight [512][512], TN = 512, TK = 16
    const float* b_qkv; const float* b_out;
    bf16_t* qkv; bf16_t* attn_o; float* lse;      // saved for backward, or NULL (not written)
    bf16_t* x_mid;
    int L, H;
    long long* dbg;             // tools/lab only (tan_attnblk_lab_set_dbg), or NULL
    float* part; long part_plane;   // SPLIT: [4][B*L][512] f32, plane hp = head pair hp's term of the out-projection; plane stride
};

constexpr int AB_D = 4;                          // weight prefetch distance in steps
struct AbWFrags { bf16x8 f[3]; };                // GEMM-a: 3 feature blocks of 16; GEMM-b: f[0..1] = 2 feature blocks of 32

__device__ __forceinline__ void ab_load_wa(AbWFrags& W, const char* pw, int hp, int ks, int wave, int lane) {
    const char* p = pw + ((long)((hp * 16 + ks) * PN_WAVES + wave) * 3) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) W.f[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
}
__device__ __forceinline__ void ab_load_wb(AbWFrags& W, const char* pw, int kt, int wave, int lane) {      // kt = k / 16 (0..31)
    const char* p = pw + (long)kt * 16384 + wave * 2048 + lane * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) W.f[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
}

// One (head, 32-query block) unit of the short-sequence attention forward on head images (tan_attn.hip: attn_fwd_short_kernel).
// O (normalised, bf16) overwrites rows q0 .. q0+31 of the q image: wave-private rows.  Returns the row's log-sum-exp (lane c and
// c + 32 hold the same value: query q0 + c).
template <int NKB>
__device__ __forceinline__ float ab_attn_unit(char* Qi, const char* Ki, const char* Vi, const float* bias, int q0, int lane) {
    const int c = lane & 31, hh = lane >> 5;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = img_frag_kc(Qi, q0 + c, 2 * ks + hh);
    f32x16 s[NKB];
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        acc_zero(s[kb]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_kc(Ki, 32 * kb + c, 2 * ks + hh), qf[ks], s[kb], 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * kb + 8 * g4 + 4 * hh);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = s[kb][4 * g4 + e] * 0.125f + bb[e];
                s[kb][4 * g4 + e] = v;
                m = fmaxf(m, v);
            }
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const bool dead = (m == -INFINITY);       // every key padded: the reference yields NaN here; zeros, like tan_attn.hip
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = dead ? 0.f : __expf(s[kb][r] - m);
            s[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = dead ? 0.f : 1.0f / sum;
    f32x16 o[2];
    acc_zero(o[0]); acc_zero(o[1]);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] *= inv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 pf = acc_frag(s[kb], j);
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag_tr(Vi, 32 * kb + 16 * j, 32 * db, lane), pf, o[db], 0, 0, 0);
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int row = q0 + c, chunk = 4 * db + g4;
            st4((bf16_t*)(Qi + row * 128 + ((chunk ^ img_swz(row)) << 4) + hh * 8),
                make_float4(o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]));
        }
    return dead ? -INFINITY : m + logf(sum);
}

// rows r0 .. r0+31 of a head image -> global rows (row stride ld elements), 8 rows per wave-instruction, rows >= L skipped
__device__ __forceinline__ void ab_rows_out(const char* img, int r0, int lane, bf16_t* out, long ld, int L) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = r0 + it * 8 + (lane >> 3), chunk = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(img + row * 128 + ((chunk ^ img_swz(row)) << 4));
        if (row < L) *reinterpret_cast<uint4*>(out + (long)row * ld + chunk * 8) = v;
    }
}

// SPLIT (round 6, small batches): the grid is (videos, 4) and workgroup (v, hp) runs ONE head pair of video v -- the prologue, GEMM-a(hp),
// the two heads' attention, GEMM-b(hp), the pair's side outputs -- and stores its [L x 512] f32 term of the out-projection as plane hp
// of `part`; bias + residual are a second launch (attnblk_split_finish_kernel).  One workgroup per video leaves 240 of the 256 CUs idle at
// B = 16 for 53 / 70 us per launch: the chains are latency-bound there (tan_panel.hip's SPLIT kernels, same idea).
template <int NRB16, bool SPLIT = false>
__global__ __launch_bounds__(64 * PN_WAVES, PN_WAVES / 4) void attnblk_fwd_kernel(AbFwdArgs a) {
    static_assert(PN_WAVES == 8, "eight waves");
    constexpr int NKB = (NRB16 + 1) / 2, LP = 32 * NKB, XROWS = 16 * NRB16, D = AB_D;
    constexpr int XP_OFF = 0, IMG_OFF = XROWS * 1024, IMG_B = LP * 128, BIAS_OFF = IMG_OFF + 6 * IMG_B, BQ_OFF = BIAS_OFF + LP * 4, LDS_B = BQ_OFF + 1536 * 4;
    static_assert(LDS_B <= 163840, "LDS budget");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_B];

    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L, C = 512;
    const long row0 = (long)blockIdx.x * L;
    const char* const pwa = a.pw_qkv;
    const char* const pwb = a.pw_out;
    const int hp0 = SPLIT ? (int)blockIdx.y : 0, hp1 = SPLIT ? hp0 + 1 : 4;       // head pairs of this workgroup

    // (a staggered start of half the workgroups -- which pays in the attention backward, tan_attn.hip -- does nothing here: 46 / 71 us
    //  with and without; one workgroup per CU, the side outputs already leave at different times per wave)
    // the weight stream does not depend on the activations: start it first (ring slots 0..3 = GEMM-a(0) steps 0..3)
    AbWFrags WQ[D];
    pn_static_for<0, D>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        ab_load_wa(WQ[J], pwa, hp0, J, wave, lane);
    });

    // ---- prologue: the xn1 panel (rows >= L repeat the last row: finite, masked as keys, never stored as queries), key bias
    {
        constexpr int RPW = XROWS / PN_WAVES;       // 8 | 10
        uint4 v[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = wave * RPW + r;
            v[r] = *reinterpret_cast<const uint4*>(a.xn1 + (row0 + min(m, L - 1)) * C + lane * 8);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) *reinterpret_cast<uint4*>(pn_panel_slot<1024>(lds + XP_OFF, wave * RPW + r, lane)) = v[r];
        float* bias = reinterpret_cast<float*>(lds + BIAS_OFF);
        const unsigned char* kp = a.keypad ? a.keypad + (long)blockIdx.x * L : nullptr;
        for (int j = tid; j < LP; j += 64 * PN_WAVES) bias[j] = (j >= L || (kp && kp[j])) ? -INFINITY : 0.f;
        // the in_proj bias (6 KiB): every GEMM-a starts from it, and a global load there is an exposed L2 round trip per head pair
        for (int j = tid; j < 1536 / 4; j += 64 * PN_WAVES)
            reinterpret_cast<float4*>(lds + BQ_OFF)[j] = reinterpret_cast<const float4*>(a.b_qkv)[j];
        if constexpr (LP > XROWS) {                 // image rows GEMM-a never writes: zero once (V rows must be finite)
            constexpr int PER = (LP - XROWS) * 8;     // 16-byte chunks per image
            for (int i = tid; i < 6 * PER; i += 64 * PN_WAVES)
                *reinterpret_cast<uint4*>(lds + IMG_OFF + (i / PER) * IMG_B + XROWS * 128 + (i % PER) * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();

    f32x16 acc_o[2][NKB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NKB; ++j) acc_zero(acc_o[i][j]);

    // activation-fragment address of row block 0 (16-row blocks): row = lane & 15, chunk (ks * 4 + (lane >> 4)) ^ (row & 15)
    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_t;
    const float* bias = reinterpret_cast<const float*>(lds + BIAS_OFF);
    const bool saving = a.qkv != nullptr;

    // Side outputs (operands of the backward).  Every vector-memory operation of a wave completes IN ORDER, so a store issued between
    // the steps of a GEMM sits in front of the weight ring's next loads until the chip's write path has acknowledged it: the 256-320 KiB
    // a video saves cost 12-20 us of a 48-75 us launch dealt into GEMM-a's steps (round 3).  Now: the q rows leave from the attention
    // wave itself before O overwrites them (as before); the k and v images of a head pair -- final once GEMM-a's epilogue has written
    // them -- leave DURING THE ATTENTION PHASE from the waves that have no attention unit (4 of 8 at L <= 64, 2 of 8 at L <= 80: their
    // ring loads for GEMM-b are already in flight, and the next ones are issued a whole attention phase later); only the O images
    // (2 x LP rows) still leave one 1-KiB piece per GEMM-a step of the NEXT head pair, read from LDS one step before they are stored.
    constexpr int PPI = LP / 8, NPW = 2 * PPI / PN_WAVES;        // pieces per image, O pieces per wave (2 | 3)
    constexpr int NIDLE = PN_WAVES - 2 * NKB, NKV = 4 * PPI / NIDLE;           // waves without an attention unit, k / v pieces per such wave (8 | 24)
    static_assert(2 * PPI % PN_WAVES == 0 && 4 * PPI % NIDLE == 0, "piece split");
    uint4 cpv;
    auto copy_read = [&](int i, int ln) __attribute__((always_inline)) {
        const int g = wave + PN_WAVES * i, j = g / PPI, row = (g % PPI) * 8 + (ln >> 3), chunk = ln & 7;
        cpv = *reinterpret_cast<const uint4*>(lds + IMG_OFF + (j * 3) * IMG_B + row * 128 + ((chunk ^ img_swz(row)) << 4));
    };
    auto copy_store = [&](int i, int ln, int hp_prev) __attribute__((always_inline)) {
        const int g = wave + PN_WAVES * i, j = g / PPI, row = (g % PPI) * 8 + (ln >> 3), chunk = ln & 7;
        if (row < L) *reinterpret_cast<uint4*>(a.attn_o + (row0 + row) * C + (2 * hp_prev + j) * 64 + chunk * 8) = cpv;
    };

    f32x4_t acc_a[3][NRB16];
    // ---- GEMM-a(hp): the wave's 3 feature blocks (of 16) x NRB16 row blocks, K = 512 in 16 steps of 32; COPY: the side outputs of
    // head pair hp - 1 leave under it
    auto gemm_a = [&](int hp, auto copy_flag) __attribute__((always_inline)) {
        constexpr bool COPY = decltype(copy_flag)::value;
        // Everything derived from the lane id inside the head-pair loop is loop-invariant, and hipcc hoists ALL of it (~100 fragment /
        // store addresses) in front of the loop and spills it.  An opaque copy of the lane id per phase keeps the address arithmetic
        // (a few VALU operations each) where it is used.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int xrow = ln & 15, xq = ln >> 4;
        const unsigned xbase = (unsigned)(uintptr_t)(lds + XP_OFF + xrow * 1024);
#pragma unroll
        for (int fb = 0; fb < 3; ++fb) {
            const int p = 3 * wave + fb, which = (p >> 2) % 3, j = p / 12, fblk = p & 3;
            const float4 bv = *reinterpret_cast<const float4*>(lds + BQ_OFF + (which * C + (2 * hp + j) * 64 + fblk * 16 + 4 * xq) * 4);
#pragma unroll
            for (int rb = 0; rb < NRB16; ++rb) { acc_a[fb][rb][0] = bv.x; acc_a[fb][rb][1] = bv.y; acc_a[fb][rb][2] = bv.z; acc_a[fb][rb][3] = bv.w; }
        }
        bf16x8 X[NRB16];       // single-buffered: a row block's fragment of the next step is read right behind its last MFMA of this one
#pragma unroll
        for (int rb = 0; rb < NRB16; ++rb) X[rb] = *(lds_frag_t)(uintptr_t)(xbase + rb * 16384 + ((xq ^ xrow) << 4));
        __builtin_amdgcn_sched_barrier(0);
        pn_static_for<0, 16>([&](auto jc) {
            constexpr int KS = decltype(jc)::value;
            AbWFrags& W = WQ[KS % D];
            if constexpr (COPY) {
                if (saving) {
                    if constexpr (KS >= 1 && KS <= NPW) copy_store(KS - 1, ln, hp - 1);
                    if constexpr (KS < NPW) copy_read(KS, ln);
                }
            }
#pragma unroll
            for (int rb = 0; rb < NRB16; ++rb) {
#pragma unroll
                for (int fb = 0; fb < 3; ++fb)
                    acc_a[fb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.f[fb], X[rb], acc_a[fb][rb], 0, 0, 0);
                if constexpr (KS < 15) X[rb] = *(lds_frag_t)(uintptr_t)(xbase + rb * 16384 + ((((KS + 1) * 4 + xq) ^ xrow) << 4));
            }
            if constexpr (KS + D < 16) ab_load_wa(W, pwa, hp, KS + D, wave, ln);
            else ab_load_wb(W, pwb, hp * 8 + (KS + D - 16), wave, ln);            // GEMM-b(hp) steps 0..3
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    gemm_a(hp0, std::false_type{});
#pragma unroll 1
    for (int hp = hp0; hp < hp1; ++hp) {
        __syncthreads();        // every wave is done reading the images of the previous head pair (GEMM-b(hp-1), its copy-out)
        // ---- GEMM-a epilogue: bf16 q|k|v into the head images
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int erow = ln & 15, eq = ln >> 4;
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) {
                const int p = 3 * wave + fb, which = (p >> 2) % 3, j = p / 12, fblk = p & 3;
                char* img = lds + IMG_OFF + (j * 3 + which) * IMG_B;
#pragma unroll
                for (int rb = 0; rb < NRB16; ++rb) {
                    const int row = rb * 16 + erow, chunk = fblk * 2 + (eq >> 1);
                    uint2 u;
                    u.x = f2bf2(acc_a[fb][rb][0], acc_a[fb][rb][1]);
                    u.y = f2bf2(acc_a[fb][rb][2], acc_a[fb][rb][3]);
                    *reinterpret_cast<uint2*>(img + row * 128 + ((chunk ^ img_swz(row)) << 4) + (eq & 1) * 8) = u;
                }
            }
        }
        __syncthreads();

        // ---- attention: one (head, 32-query block) unit per wave; the other waves move this head pair's k and v rows out
        if (wave >= 2 * NKB) {
            if (saving) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll 4
                for (int i = 0; i < NKV; ++i) {
                    const int g = (wave - 2 * NKB) + NIDLE * i, im = g / PPI, j = im >> 1, which = 1 + (im & 1);
                    const int row = (g % PPI) * 8 + (ln >> 3), chunk = ln & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(lds + IMG_OFF + (j * 3 + which) * IMG_B + row * 128 + ((chunk ^ img_swz(row)) << 4));
                    if (row < L) *reinterpret_cast<uint4*>(a.qkv + (row0 + row) * (3 * C) + which * C + (2 * hp + j) * 64 + chunk * 8) = v;
                }
            }
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int j = wave / NKB, q0 = (wave % NKB) * 32, h = 2 * hp + j;
            char* Qi = lds + IMG_OFF + (j * 3) * IMG_B;
            if (saving) ab_rows_out(Qi, q0, ln, a.qkv + row0 * (3 * C) + h * 64, 3 * C, L);       // q rows, before O overwrites them
            const float l = ab_attn_unit<NKB>(Qi, Qi + IMG_B, Qi + 2 * IMG_B, bias, q0, ln);
            if (saving && ln < 32 && q0 + ln < L) a.lse[((long)blockIdx.x * a.H + h) * L + q0 + ln] = l;
        }
        __syncthreads();

        // ---- GEMM-b: acc_o += O[:, head pair] W_out[:, head pair]^T, K = 128 in 8 steps of 16
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            pn_static_for<0, 8>([&](auto jc) {
                constexpr int KB = decltype(jc)::value;
                const char* Oi = lds + IMG_OFF + ((KB >> 2) * 3) * IMG_B;
                AbWFrags& W = WQ[KB % D];
                bf16x8 of[NKB];
#pragma unroll
                for (int rb = 0; rb < NKB; ++rb) of[rb] = img_frag_kc(Oi, rb * 32 + (ln & 31), 2 * (KB & 3) + (ln >> 5));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int rb = 0; rb < NKB; ++rb)
                        acc_o[nb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.f[nb], of[rb], acc_o[nb][rb], 0, 0, 0);
                if constexpr (KB + D < 8) ab_load_wb(W, pwb, hp * 8 + KB + D, wave, ln);
                else if (hp + 1 < hp1) ab_load_wa(W, pwa, hp + 1, KB + D - 8, wave, ln);      // GEMM-a(hp+1) steps 0..3
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if (hp + 1 < hp1) gemm_a(hp + 1, std::true_type{});
    }
    if constexpr (SPLIT) {
        if (saving) {          // the head pair's O rows
#pragma unroll 1
            for (int i = 0; i < NPW; ++i) { copy_read(i, lane); copy_store(i, lane, hp0); }
        }
        // the head pair's term of the out-projection -> plane hp0: a lane owns the 16 consecutive features wave * 64 + nb * 32 + 16 hi + r
        // of row mb * 32 + (lane & 31)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < NKB; ++mb) {
                const int m = mb * 32 + (lane & 31);
                if (m < L) {
                    float* dst = a.part + hp0 * a.part_plane + (row0 + m) * C + wave * 64 + nb * 32 + 16 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(acc_o[nb][mb][4 * q], acc_o[nb][mb][4 * q + 1], acc_o[nb][mb][4 * q + 2],
                                                                              acc_o[nb][mb][4 * q + 3]);
                }
            }
        return;
    }
    // the residual rows of the epilogue: requested before the last head pair's side outputs leave
    uint4 resq[2][NKB][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < NKB; ++mb)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                resq[nb][mb][p] = *reinterpret_cast<const uint4*>(a.x_in + (row0 + min(mb * 32 + (lane & 31), L - 1)) * C + wave * 64 + nb * 32 + (2 * hi + p) * 8);
    if (saving) {          // the last head pair's O rows
#pragma unroll 1
        for (int i = 0; i < NPW; ++i) { copy_read(i, lane); copy_store(i, lane, 3); }
    }

    // ---- epilogue: x_mid = x_in + acc_o + b_out.  A lane owns features nbase + 16 hi + r of row mb * 32 + (lane & 31) (packer's
    // feature permutation); the rows leave through the panel's LDS space as whole 1-KiB rows.
    __syncthreads();
    char* xo_panel = lds + XP_OFF;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int nbase = wave * 64 + nb * 32;
        pn_cfptr_t bp = (pn_cfptr_t)(uintptr_t)(a.b_out) + nbase;
#pragma unroll
        for (int mb = 0; mb < NKB; ++mb) {
            const int m = mb * 32 + (lane & 31);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float bias8[8], res[8], v[8];
                pn_uniform8(bp + 8 * p, bp + 16 + 8 * p, hi, bias8);
                pn_unpack8(resq[nb][mb][p], res);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc_o[nb][mb][8 * p + e] + bias8[e] + res[e];
                if (m < XROWS) *reinterpret_cast<uint4*>(pn_panel_slot<1024>(xo_panel, m, (nbase >> 3) + 2 * hi + p)) = pn_pack8(v);
            }
        }
    }
    __syncthreads();
    for (int m = wave; m < L; m += PN_WAVES)
        *reinterpret_cast<uint4*>(a.x_mid + (row0 + m) * C + lane * 8) = *reinterpret_cast<const uint4*>(pn_panel_slot<1024>(xo_panel, m, lane));
}

// x_mid = x_in + (sum of the four head pairs' planes, in order) + b_out: one wave per row, a lane owns 8 consecutive features
__global__ __launch_bounds__(256) void attnblk_split_finish_kernel(const float* __restrict__ part, long plane, const bf16_t* __restrict__ x_in,
                                                                   const float* __restrict__ b_out, bf16_t* __restrict__ x_mid, long rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    f8 acc = ld8f(part + row * 512 + lane * 8);
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const f8 t = ld8f(part + c * plane + row * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] += t.v[j];
    }
    const f8 res = ld8(x_in + row * 512 + lane * 8), bias = ld8f(b_out + lane * 8);
    f8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] = acc.v[j] + bias.v[j] + res.v[j];
    st8(x_mid + row * 512 + lane * 8, v);
}

}  // namespace tal

using namespace tal;

static long long* g_ab_dbg = nullptr;
extern "C" int tan_attnblk_lab_set_dbg(void* p) { g_ab_dbg = (long long*)p; return 0; }      // tools/lab/attnblk_lab.py

// rows per video the fused attention-branch kernels accept (C = 512, H = 8, bf16)
extern "C" int tan_attnblk_supported(int L, int C, int H, int dtype) {
    return dtype == TAN_BF16 && C == 512 && H == 8 && L > 48 && L <= 80;
}

extern "C" int tan_attnblk_fwd(const tan_attnblk_desc* d, void* stream) {
    TAN_REQUIRE(d && d->xn1 && d->x_in && d->pw_qkv && d->pw_out && d->b_qkv && d->b_out && d->x_mid && d->B > 0);
    TAN_REQUIRE(tan_attnblk_supported(d->L, d->C, d->H, TAN_BF16));
    TAN_REQUIRE((d->qkv != nullptr) == (d->attn_o != nullptr) && (d->qkv != nullptr) == (d->lse != nullptr));
    AbFwdArgs a;
    a.xn1 = (const bf16_t*)d->xn1; a.x_in = (const bf16_t*)d->x_in; a.keypad = d->key_padding_mask;
    a.pw_qkv = (const char*)d->pw_qkv; a.pw_out = (const char*)d->pw_out; a.b_qkv = d->b_qkv; a.b_out = d->b_out;
    a.qkv = (bf16_t*)d->qkv; a.attn_o = (bf16_t*)d->attn_o; a.lse = d->lse; a.x_mid = (bf16_t*)d->x_mid;
    a.L = d->L; a.H = d->H; a.dbg = g_ab_dbg;
    a.part = nullptr; a.part_plane = 0;
    const dim3 grid((unsigned)d->B);
    const double rows = (double)d->B * d->L;
    const int rec = prof_begin((hipStream_t)stream, TAN_PROF_ATTNBLK, 2.0 * rows * 512.0 * 2048.0 + 4.0 * rows * d->L * 512.0);
    if (d->L <= 64) hipLaunchKernelGGL((attnblk_fwd_kernel<4>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attnblk_fwd_kernel<5>), grid, dim3(64 * PN_WAVES), 0, (hipStream_t)stream, a);
    prof_end((hipStream_t)stream, rec);
    TAN_LAUNCH_CHECK();
    return 0;
}

// Small batches: the same branch with one workgroup per (video, head pair); `part` = scratch [4, B*L, C] f32 (tan_hip.h)
extern "C" int tan_attnblk_fwd_split(const tan_attnblk_desc* d, float* part, void* stream) {
    TAN_REQUIRE(d && part && d->xn1 && d->x_in && d->pw_qkv && d->pw_out && d->b_qkv && d->b_out && d->x_mid && d->B > 0);
    TAN_REQUIRE(tan_attnblk_supported(d->L, d->C, d->H, TAN_BF16));
    TAN_REQUIRE((d->qkv != nullptr) == (d->attn_o != nullptr) && (d->qkv != nullptr) == (d->lse != nullptr));
    AbFwdArgs a;
    a.xn1 = (const bf16_t*)d->xn1; a.x_in = (const bf16_t*)d->x_in; a.keypad = d->key_padding_mask;
    a.pw_qkv = (const char*)d->pw_qkv; a.pw_out = (const char*)d->pw_out; a.b_qkv = d->b_qkv; a.b_out = d->b_out;
    a.qkv = (bf16_t*)d->qkv; a.attn_o = (bf16_t*)d->attn_o; a.lse = d->lse; a.x_mid = (bf16_t*)d->x_mid;
    a.L = d->L; a.H = d->H; a.dbg = nullptr;
    const long rows = (long)d->B * d->L;
    a.part = part; a.part_plane = rows * 512;
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)d->B, 4);
    const int rec = prof_begin(st, TAN_PROF_ATTNBLK, 2.0 * rows * 512.0 * 2048.0 + 4.0 * rows * d->L * 512.0);
    if (d->L <= 64) hipLaunchKernelGGL((attnblk_fwd_kernel<4, true>), grid, dim3(64 * PN_WAVES), 0, st, a);
    else hipLaunchKernelGGL((attnblk_fwd_kernel<5, true>), grid, dim3(64 * PN_WAVES), 0, st, a);
    prof_end(st, rec);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(attnblk_split_finish_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, part, rows * 512, (const bf16_t*)d->x_in, d->b_out,
                       (bf16_t*)d->x_mid, rows);
    TAN_LAUNCH_CHECK();
    return 0;
}
