// Logits-free similarity + multi-positive NCE (gfx950, bf16 features).
//
// The reference materialises cosine logits [B,S,T,B,N] (tan_model.py:118,138; 403 MB f32 per tensor at B=128,N=16) and then
// makes several passes over them (clone / masked fill / two logsumexp per direction, loss.py:240-275).  Here one workgroup
// owns a 128-row panel of one stage and sweeps ALL text columns: every 128x128 logit tile is produced by the direct-to-LDS
// MFMA pipeline of tan_gemm_glds.hip and consumed in registers.
//
//   MODE_STATS : e = exp(l/0.07 - 1/0.07); row sums (valid columns) stay in registers across the sweep, column sums go to LDS
//                atomics, positives (same-video entries with tgt=1) likewise  ->  rowsum, possum_v, colpart, pospart
//   MODE_DL    : recomputes the tile and writes d loss / d logit (bf16) for the two follow-up GEMMs (d v_hat, d t_hat)
//
// so the forward never writes logits, and the backward writes them once in bf16 instead of f32 + cast.
#include <type_traits>

#include "tan_mma.h"

namespace tal {

constexpr float S_TAU = 0.07f;
constexpr int S_TILE = 128 * 64 * 2;  // 16 KiB operand tile
constexpr int S_MAXCOLS = 2048;       // column limit of simnce_kernel (LDS column accumulators)
constexpr int S_MAXCOLS_RES = 8192;   // column limit of the resident sweep (column partials go to global rows: only the workspace grows)

typedef const void __attribute__((address_space(1)))* sgptr_t;
typedef void __attribute__((address_space(3)))* slptr_t;

struct FamPtrs { void* p[8]; };
struct SimArgs {
    const bf16_t* V;      // [S, R, C] unit video features
    const bf16_t* Tt;     // [S or 1, Mp, C] unit text features
    long t_stage_stride;  // Mp*C or 0
    const float* tgt;     // [B, T, N] {0,1}
    const unsigned char* col_invalid;  // [Mp]
    const unsigned char* row_leak;     // [R] or null
    float *rowsum, *possum_v;          // [S, R]
    float *colpart, *pospart;          // [panels, S, Mp]   (MODE_STATS out)
    const float *colsum, *possum_t;    // [S, Mp]           (MODE_DL in)
    const float *g_v, *g_t;            // [S, R], [S, Mp]   (MODE_DL in)
    bf16_t* dl;                        // [S, R, Mp]        (MODE_DL out)
    int S, B, T, N, C, R, Mp;
    bf16_t* ekeep;                     // [S, R, Mp] or null: simnce_res_kernel<0> also stores every e = exp((cos - 1)/tau) (bf16)
    // simnce_res_kernel<1> with the same-video corrections as its tail (instead of a simnce_diag_kernel<true> launch):
    const float* diag;                 // [S, B, T, N] same-video cosines, or null: no corrections in the sweep kernel
    const float* corr;                 // [S, R, N] same-video corrections of the d-logits (simnce_corr_kernel), or null (simnce_dl_dvn_kernel)
    const int* colmap;                 // padded column b*N+k -> column of the sweep, or -1; null: identity
    int npanel, nfull;                 // simnce_res_kernel: row panels per stage; items (stage, panel) that are not cut in column halves
    const char* Tp;                    // simnce_res_kernel: fragment-major image of the text features (simnce_pack_text_kernel)
    long tp_stage_stride;              // bytes, or 0 (text features shared by the stages)
    // simnce_res_kernel<0> normalising its frame panel itself (tan_simfam_fwd, TAN_SIMFAM_NORM_IN_SWEEP): the panel is staged from the
    // stack's RAW stage rows (frame r of stage s at xraw.p[s] + ((r / T) * x_grp_rows + x_off + r % T) * C), L2-normalised in the LDS with
    // l2n_fwd_kernel's arithmetic, and written to V / inv_v for the backward -- inv_v == null: V holds unit features already
    FamPtrs xraw; long x_grp_rows, x_off;
    float* inv_v;
};

// K-contiguous 128-row operand tile, same image as tan_gemm_glds.hip (slot = chunk ^ ((row >> 1) & 7))
__device__ __forceinline__ void s_stage(const bf16_t* __restrict__ P, long ld, int outer0, int OUT, int k0, char* lds_tile,
                                        int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int row = piece * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int gr = min(outer0 + row, OUT - 1);
        __builtin_amdgcn_global_load_lds((sgptr_t)(P + (long)gr * ld + k0 + chunk * 8), (slptr_t)(lds_tile + piece * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 s_frag(const char* lds_tile, int o0, int ks, int lane) {
    const int row = o0 + (lane & 31), chunk = (ks >> 3) + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + (chunk ^ ((row >> 1) & 7)) * 16);
}

struct TileCtx {
    const unsigned char* col_invalid;
    const float* colsum; const float* g_t; bf16_t* dl;
    int R, Mp, s, m0, wm, wn, lane;
    float inv_tau;
    int cinv[2];                  // simnce_res_kernel: pad flags of the wave's two 32-column blocks, requested before the tile's K loop, or -1
};

// consume one finished 128x128 logit tile (column tile ct) straight from the accumulators.  Positives and the padded-frame
// quirk only touch the same-video blocks and are handled by simnce_diag_kernel on separately computed [S,B,T,N] blocks, so
// this hot loop carries no target lookups.
template <int MODE>
__device__ __forceinline__ void tile_done(const TileCtx& c, f32x16 (&acc)[2][2], float (&rowacc)[2][16], float* colacc, int ct) {
    const int c0 = ct * 128, R = c.R, Mp = c.Mp, lane = c.lane;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = c0 + c.wn * 64 + j * 32 + acc_col(lane);
        const bool col_ok = col < Mp;
        const bool col_valid = col_ok && !c.col_invalid[min(col, Mp - 1)];
        float csum = 0.f, bc = 0.f;
        if (MODE == 1) {
            const long idx = (long)c.s * Mp + min(col, Mp - 1);
            const float gt = c.g_t[idx], cs = c.colsum[idx];
            bc = col_ok ? gt / cs * c.inv_tau : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * 64 + i * 32 + acc_row(r, lane);
                float e = __expf((acc[i][j][r] - 1.0f) * c.inv_tau);
                if (!col_ok || row >= R) e = 0.f;
                if (MODE == 0) {
                    if (col_valid) rowacc[i][r] += e;
                    csum += e;
                } else {
                    const float g = e * ((col_valid ? rowacc[i][r] : 0.f) + bc);
                    if (col_ok && row < R) c.dl[((long)c.s * R + row) * Mp + col] = f2bf(g);
                }
            }
        if (MODE == 0 && col_ok) atomicAdd(&colacc[col], csum);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void simnce_kernel(SimArgs a) {
    // ONE static LDS object (4 operand tiles, then colacc[S_MAXCOLS], posacc[S_MAXCOLS]): with a dynamic (extern) array hipcc
    // cannot tell the DMA destination from the fragment reads and drains the DMA before every ds_read.
    __shared__ __attribute__((aligned(1024))) char lds[4 * S_TILE + (MODE == 0 ? S_MAXCOLS * 4 : 0)];
    float* colacc = reinterpret_cast<float*>(lds + 4 * S_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup order: every workgroup of the launch is resident at once (S * R/128 <= 512 slots) and consecutive ids go
    // round-robin to the 8 XCDs, so id -> (id % 8) * ceil(n/8) + id / 8 (bijective form) gives each XCD a CONTIGUOUS run of
    // (stage, row panel) items: at most two stages' text features (2 MB each) per 4-MB L2 instead of all S thrashing it.
    const int npanel = gridDim.x;
    int wg = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const int nwg = gridDim.x * gridDim.y, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int s = wg / npanel, panel = wg - s * npanel, m0 = panel * 128;
    const int R = a.R, Mp = a.Mp, Cw = a.C, nS = a.S;
    const bf16_t* V = a.V + (long)s * R * Cw;
    const bf16_t* Tt = a.Tt + (long)s * a.t_stage_stride;
    const int nk = Cw / 64, nct = (Mp + 127) / 128, nsteps = nct * nk;
    const float inv_tau = 1.0f / S_TAU;

    if (MODE == 0) {
        for (int c = tid; c < S_MAXCOLS; c += 256) colacc[c] = 0.f;
    }
    // per-lane row bookkeeping: this lane's 32 rows are m0 + wm*64 + i*32 + acc_row(r, lane)
    float rowacc[2][16];      // MODE_STATS: running row sums; MODE_DL: gv/rowsum/tau
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            rowacc[i][r] = 0.f;
            if (MODE == 1) {      // clamped, unconditional loads (a conditional load costs a branch + s_waitcnt each)
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                const long idx = (long)s * R + min(row, R - 1);
                const float gv = a.g_v[idx], rs = a.rowsum[idx];
                rowacc[i][r] = row < R ? gv / rs * inv_tau : 0.f;
            }
        }
    TileCtx c;
    c.col_invalid = a.col_invalid; c.colsum = a.colsum; c.g_t = a.g_t; c.dl = a.dl;
    c.R = R; c.Mp = Mp; c.s = s; c.m0 = m0; c.wm = wm; c.wn = wn; c.lane = lane; c.inv_tau = inv_tau;

    f32x16 acc[2][2];
    // One K-step of the flattened (column tile, k) sweep.  Buffer offsets are compile-time (two copies of the body).
#define SIM_KSTEP(STEP, CUR, NXT)                                                                                          \
    {                                                                                                                      \
        const int step_ = (STEP);                                                                                          \
        const int ct_ = step_ / nk, kt_ = step_ - ct_ * nk;                                                                \
        if (step_ + 1 < nsteps) {                                                                                          \
            const int ct2 = (step_ + 1) / nk, kt2 = (step_ + 1) - ct2 * nk;                                                \
            s_stage(V, Cw, m0, R, kt2 * 64, lds + (NXT) * 2 * S_TILE, wave, lane);                                         \
            s_stage(Tt, Cw, ct2 * 128, Mp, kt2 * 64, lds + (NXT) * 2 * S_TILE + S_TILE, wave, lane);                       \
        }                                                                                                                  \
        if (kt_ == 0) {                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]); \
        }                                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 64; ks += 16) {                                                            \
            bf16x8 af[2], bfr[2];                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = s_frag(lds + (CUR) * 2 * S_TILE, wm * 64 + i * 32, ks, lane); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
                bfr[j] = s_frag(lds + (CUR) * 2 * S_TILE + S_TILE, wn * 64 + j * 32, ks, lane);                            \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);                    \
        }                                                                                                                  \
        if (kt_ == nk - 1) tile_done<MODE>(c, acc, rowacc, colacc, ct_);                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
        __syncthreads();                                                                                                   \
    }

    s_stage(V, Cw, m0, R, 0, lds, wave, lane);
    s_stage(Tt, Cw, 0, Mp, 0, lds + S_TILE, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int step = 0; step < nsteps; step += 2) {
        SIM_KSTEP(step, 0, 1)
        if (step + 1 < nsteps) SIM_KSTEP(step + 1, 1, 0)
    }
#undef SIM_KSTEP

    if (MODE == 0) {
        // row sums: reduce over the 32 lanes that share a row (same lane >> 5), then over the two column waves via atomics
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = rowacc[i][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if ((lane & 31) == 0 && row < R) unsafeAtomicAdd(a.rowsum + (long)s * R + row, v);
            }
        __syncthreads();
        float* cp = a.colpart + ((long)panel * nS + s) * Mp;
        for (int cc = tid; cc < Mp; cc += 256) cp[cc] = colacc[cc];
    }
}

// ---- the same sweep with the frame panel RESIDENT in the LDS (C = 512) ---------------------------------------------------------
// simnce_kernel re-stages its 128 x 512 frame panel for every column tile: 2.5 MB of L2->LDS traffic per workgroup, and the 48
// panels an XCD works on at once (6 MB) do not fit its 4-MB L2 next to the text features -- rocprofv3 counted ~500 MB of HBM reads
// per launch for 52 MB of operands.  Here the panel's eight 16-KiB K tiles are staged ONCE (128 KiB) and stay; EIGHT waves share
// them: two groups of four (each the 2 x 2 wave grid of simnce_kernel) take the even / the odd column tiles, each streaming its
// text tile through its own 2 x 8 KiB double buffer in 32-deep K steps -- 128 + 32 = 160 KiB, the whole LDS of a CU, two waves per
// SIMD as with two resident workgroups of the re-staging kernel.  (Four waves per CU -- the panel + one 2 x 16 KiB text buffer --
// were measured first: 234 / 224 us vs 184 / 157 us per launch; one compiler-scheduled wave per SIMD cannot cover its own stalls.
// Cutting each panel's columns over two workgroups, 768 shorter items instead of 384 = a round and a half: 235 us, 5.34 vs 5.25 ms.)
// Measured: HBM reads 502 -> 62 MB (statistics sweep) and 514 -> 83 MB (dlogits sweep) per launch; 177 / 182 us per launch in the
// step (re-staging kernel: 167 / 157 us), the step itself 5.243 vs 5.256 ms (ABBA x2) -- the sweep was never HBM-bound, its loop
// (a `vmcnt(0)` + barrier per K step, compiler-ordered) is.
// Column sums: the 128 columns of a tile are final when the tile is done (a column tile is visited once per workgroup), so the two
// row-waves store their halves straight to colpart rows (2 * panel + wm): no LDS accumulators, no atomics.
template <int MODE>
__device__ __forceinline__ void tile_done_res(const TileCtx& c, f32x16 (&acc)[2][2], float (&rowacc)[2][16], float* colrow, int ct) {
    const int c0 = ct * 128, R = c.R, Mp = c.Mp, lane = c.lane;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = c0 + c.wn * 64 + j * 32 + acc_col(lane);
        const bool col_ok = col < Mp;
        const bool col_valid = col_ok && !(c.cinv[j] >= 0 ? c.cinv[j] : (int)c.col_invalid[min(col, Mp - 1)]);
        float csum = 0.f, bc = 0.f, e_prev = 0.f;
        unsigned pk[4] = {0u, 0u, 0u, 0u};
        const float k1 = 1.4426950408889634f * c.inv_tau, cmask = col_ok ? 1.0f : 0.0f, vmask = col_valid ? 1.0f : 0.0f;
        const bool rows_full = c.m0 + 128 <= R;
        if (MODE == 1) {
            const long idx = (long)c.s * Mp + min(col, Mp - 1);
            const float gt = c.g_t[idx], cs = c.colsum[idx];
            bc = col_ok ? gt / cs * c.inv_tau : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * 64 + i * 32 + acc_row(r, lane);
                // exp((cos - 1) / tau) as one fma + v_exp_f32; columns past Mp by a 0/1 factor, rows past R only in the last panel
                float e = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], k1, -k1)) * cmask;
                if (!rows_full && row >= R) e = 0.f;
                if (MODE == 0) {
                    rowacc[i][r] = fmaf(e, vmask, rowacc[i][r]);
                    csum += e;
                    // kept for the backward, in ACCUMULATOR order (two rows per dword): every store instruction
                    // writes contiguous bytes.  Row-major 2-byte stores (64-byte pieces of lines) cost the sweep +110 us and
                    // 200 MB of read-modify-write traffic; simnce_dl_kept_kernel does the transposition, where the LDS is free.
                    if (c.dl) {       // [i][j][r / 8][lane][4 dwords]: 16 bytes per lane, 1 KiB per store instruction (as dword stores --
                        //                 256 B per instruction -- the 327 KB a workgroup keeps were store-ISSUE bound: a CU retires ~4 B/clk of those)
                        if (r & 1) pk[(r >> 1) & 3] = f2bf2(e_prev, e);
                        e_prev = e;
                        if ((r & 7) == 7)
                            reinterpret_cast<uint4*>(c.dl)[((i * 2 + j) * 2 + (r >> 3)) * 64 + lane] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    }
                } else {
                    const float g = e * ((col_valid ? rowacc[i][r] : 0.f) + bc);
                    if (col_ok && row < R) c.dl[((long)c.s * R + row) * Mp + col] = f2bf(g);
                }
            }
        if (MODE == 0) {
            csum += __shfl_xor(csum, 32, 64);
            if (lane < 32 && col_ok) colrow[col] = csum;
        }
    }
}

template <int J, int END, typename F>
__device__ __forceinline__ void pn_static_for_s(F&& f) {
    if constexpr (J < END) {
        f(std::integral_constant<int, J>{});
        pn_static_for_s<J + 1, END>(f);
    }
}

// Fragment-major image of the (compacted) unit text features for simnce_res_kernel: piece (column block cb of 32, k step ks of 16) is
// 1 KiB, lane l's 16 bytes = Tt[cb*32 + (l & 31)][16 ks + 8 (l >> 5) ..] -- the B operand of v_mfma_f32_32x32x16_bf16 as ONE coalesced
// wave-load.  Columns past Mp repeat the last one (their logits are masked by col_ok).  grid (column blocks, stages), 256 threads.
__global__ __launch_bounds__(256) void simnce_pack_text_kernel(const bf16_t* __restrict__ Tt, long t_stage_stride, char* __restrict__ Tp,
                                                               long tp_stage_stride, int Mp, float* __restrict__ zero, long nzero) {
    const int cb = blockIdx.x, st = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (zero)      // the sweep's row sums start from zero (they meet in f32 atomics): was a memset launch in front of every sweep
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nzero; i += (long)gridDim.x * gridDim.y * 256) zero[i] = 0.f;
    const bf16_t* src = Tt + (long)st * t_stage_stride + (long)min(cb * 32 + (lane & 31), Mp - 1) * 512 + 8 * (lane >> 5);
    char* dst = Tp + (long)st * tp_stage_stride + (long)cb * 32 * 1024 + lane * 16;
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(src + (w * 8 + i) * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(dst + (long)(w * 8 + i) * 1024) = v[i];
}

template <int MODE>
__global__ __launch_bounds__(512) void simnce_res_kernel(SimArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * S_TILE];      // the frame panel: 8 K tiles of [128 rows][64 channels]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3, wm = gw >> 1, wn = gw & 1;
    // 1-D grid over (stage, row panel) items, one workgroup per CU.  The items past the last full round of the chip (384 items on 256
    // CUs: 128) are cut in two column halves each -- two workgroups -- so that the last round keeps every CU busy for half an item
    // instead of half the CUs for a whole one (a.nfull = items that are not cut; blocks are dispatched in id order).
    const int npanel = a.npanel;
    int wg = blockIdx.x, half = -1;
    if (wg < a.nfull) {
        const int nwg = a.nfull, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;       // a contiguous run of items per XCD (see simnce_kernel)
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    } else {
        half = (wg - a.nfull) & 1;
        wg = a.nfull + ((wg - a.nfull) >> 1);
    }
    const int s = wg / npanel, panel = wg - s * npanel, m0 = panel * 128;
    const int R = a.R, Mp = a.Mp, Cw = a.C, nS = a.S;
    const bf16_t* V = a.V + (long)s * R * Cw;
    const int nct = (Mp + 127) / 128;
    const float inv_tau = 1.0f / S_TAU;

    float rowacc[2][16];      // MODE_STATS: running row sums (this group's column tiles); MODE_DL: gv/rowsum/tau
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            rowacc[i][r] = 0.f;
            if (MODE == 1) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                const long idx = (long)s * R + min(row, R - 1);
                const float gv = a.g_v[idx], rs = a.rowsum[idx];
                rowacc[i][r] = row < R ? gv / rs * inv_tau : 0.f;
            }
        }
    TileCtx c;
    c.col_invalid = a.col_invalid; c.colsum = a.colsum; c.g_t = a.g_t; c.dl = a.dl;
    c.R = R; c.Mp = Mp; c.s = s; c.m0 = m0; c.wm = wm; c.wn = wn; c.lane = lane; c.inv_tau = inv_tau;
    float* colrow = MODE == 0 ? a.colpart + ((long)(2 * panel + wm) * nS + s) * Mp : nullptr;

    f32x16 acc[2][2];
    // The frame panel (8 K tiles of 64) is staged once and stays.  The TEXT operand never touches LDS: every wave streams the B
    // fragments of its own 64 columns straight into registers from a fragment-major image (simnce_pack_text_kernel: one 1-KiB
    // wave-load per 32 columns x 16 channels), through a ring RD steps deep that runs across column tiles.  The first version staged
    // 32-deep text tiles through two 8-KiB LDS buffers per wave group -- all the LDS the panel leaves -- with `s_waitcnt vmcnt(0)` and
    // a workgroup barrier behind every 8 MFMAs: one K step of prefetch against ~1 us of L2 latency under eight waves' requests, ~2.5k
    // cycles per step where the matrix pipe needs 0.5k (184 us per sweep).  Now the only barrier is the one behind the panel.
    constexpr int RD = MODE == 0 ? 8 : 4;          // 8 steps x 2 KiB in flight per wave: a step is only 4 MFMAs (128 cycles), the L2 answers in ~1k
    //                                               (the recomputing d-logits sweep, MODE 1, has the registers for 4)
    struct BF { bf16x8 f[2]; };
    BF ring[RD];
    const char* Tp = a.Tp + (long)s * a.tp_stage_stride;
    auto load_b = [&](BF& dst, int ct, int ks) __attribute__((always_inline)) {       // columns ct*128 + wn*64 + j*32 .., k = 16 ks ..
        // (Timing-only ablation, round 5: the wm = 1 waves re-reading one L1-resident piece -- HALF the L2 -> CU text stream -- takes the
        //  sweep from 108.2 to 105.9 us stand-alone and leaves the step where it is: the loop is not bound by that stream, and a
        //  128 x 64 wave tile that halves it would buy nothing.)
        const char* pb = Tp + ((long)((ct * 4 + wn * 2) * 32 + ks)) * 1024 + lane * 16;
        dst.f[0] = *reinterpret_cast<const bf16x8*>(pb);
        dst.f[1] = *reinterpret_cast<const bf16x8*>(pb + 32 * 1024);
    };
    // Every workgroup walks the same text image; in the same order and at the same pace they all ask the same L2 channel for the same
    // lines at the same time.  Each panel starts its walk at a different column tile (TAN_SIM_ROT=0: all start at tile 0).
    const int nall = (nct - grp + 1) / 2;                      // column tiles of this wave group: grp, grp + 2, ..
    const int kbase = half == 1 ? (nall + 1) / 2 : 0;          // .. of which a half item takes the first or the second part
    const int ngrp = half < 0 ? nall : (half == 0 ? (nall + 1) / 2 : nall / 2);
    const int rot = ngrp > 0 ? (panel * 7 + s * 3) % ngrp : 0;
    if (ngrp > 0) {
        pn_static_for_s<0, RD>([&](auto jc) { constexpr int J = decltype(jc)::value; load_b(ring[J], 2 * (kbase + rot) + grp, J); });
    }
    if (MODE == 0 && a.inv_v) {
        // raw stage rows -> LDS (same image as s_stage; a row's address goes through the stack's row grouping)
        const bf16_t* X = reinterpret_cast<const bf16_t*>(a.xraw.p[s]);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            char* tile = lds + (grp * 4 + kt) * S_TILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = gw * 4 + i;
                const int row = piece * 8 + (lane >> 3), slot = lane & 7;
                const int chunk = slot ^ ((row >> 1) & 7);
                const int gr = min(m0 + row, R - 1);
                const long sr = (long)(gr / a.T) * a.x_grp_rows + a.x_off + gr % a.T;
                __builtin_amdgcn_global_load_lds((sgptr_t)(X + sr * Cw + (grp * 4 + kt) * 64 + chunk * 8), (slptr_t)(tile + piece * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // x / |x| in place, one wave per row, lane l = elements 8 l .. 8 l + 7 (one 16-byte LDS access each way: element e of row r lives
        // in K tile e / 64 at r * 128 + (((e % 64) / 8) ^ ((r >> 1) & 7)) * 16), and the unit row leaves for HBM as ONE 1-KiB store per
        // row -- 8-byte pieces (l2n_fwd_kernel's lane layout) made the 128 KiB a workgroup keeps store-issue bound: +40 us per sweep
        bf16_t* vout = const_cast<bf16_t*>(V);
#pragma unroll 1
        for (int i0 = 0; i0 < 16; i0 += 4) {
            uint4 raw[4];
            char* ap[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wave * 16 + i0 + j;
                ap[j] = lds + (lane >> 3) * S_TILE + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) * 16);
                raw[j] = *reinterpret_cast<const uint4*>(ap[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wave * 16 + i0 + j;
                const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
                float x[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
                float q = ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
                const float inv = 1.0f / sqrtf(wave_sum(q));
                uint4 y;
                y.x = f2bf2(x[0] * inv, x[1] * inv); y.y = f2bf2(x[2] * inv, x[3] * inv);
                y.z = f2bf2(x[4] * inv, x[5] * inv); y.w = f2bf2(x[6] * inv, x[7] * inv);
                *reinterpret_cast<uint4*>(ap[j]) = y;
                if (m0 + row < R) {
                    *reinterpret_cast<uint4*>(vout + (long)(m0 + row) * Cw + lane * 8) = y;
                    if (lane == 0) a.inv_v[(long)s * R + m0 + row] = inv;
                }
            }
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s_stage(V, Cw, m0, R, (grp * 4 + kt) * 64, lds + (grp * 4 + kt) * S_TILE, gw, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // (starting the second wave group 32 / 64 / 127 x 64 cycles late -- so that its tile epilogues would fall under the first group's K
    //  loops -- changes nothing: 0.168-0.18 ms per forward either way; the waves are not barrier-coupled and drift apart by themselves)
    bf16x8 afA[2], afB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) afA[i] = s_frag(lds, wm * 64 + i * 32, 0, lane);
    for (int it = 0; it < ngrp; ++it) {
        const int k_ = it + rot < ngrp ? it + rot : it + rot - ngrp, kn_ = k_ + 1 < ngrp ? k_ + 1 : 0;
        const int ct_ = 2 * (kbase + k_) + grp, ctn_ = 2 * (kbase + kn_) + grp;
        const bool more = it + 1 < ngrp;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);
        // the pad flags of this tile's columns: requested here, consumed by the tile epilogue (they were two dependent L2 round trips
        // at the END of every tile)
#pragma unroll
        for (int j = 0; j < 2; ++j) c.cinv[j] = a.col_invalid[min(ct_ * 128 + wn * 64 + j * 32 + (lane & 31), Mp - 1)];
        pn_static_for_s<0, 32>([&](auto jc) {
            constexpr int KS = decltype(jc)::value;
            BF& Bf = ring[KS % RD];
            bf16x8(&af)[2] = (KS & 1) ? afB : afA;          // frame fragments: read from LDS one step ahead
            bf16x8(&an)[2] = (KS & 1) ? afA : afB;
#ifndef TAN_SIM_LAB
#define TAN_SIM_LAB 0      // tools/lab timing ablations (results wrong, nothing dead): 1 no text ring loads, 2 no frame-fragment reads, 4 no kept-e stores, 8 no tile epilogue
#endif
            if constexpr (!(TAN_SIM_LAB & 2)) {
                constexpr int KN = (KS + 1) & 31;
                const char* vn_ = lds + (KN >> 2) * S_TILE;
#pragma unroll
                for (int i = 0; i < 2; ++i) an[i] = s_frag(vn_, wm * 64 + i * 32, (KN & 3) * 16, lane);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(an[i]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], Bf.f[j], acc[i][j], 0, 0, 0);
            if constexpr (!(TAN_SIM_LAB & 1)) {
                if constexpr (KS + RD < 32) load_b(Bf, ct_, KS + RD);
                else if (more) load_b(Bf, ctn_, KS + RD - 32);
            } else {
                asm volatile("" : "+v"(Bf.f[0]), "+v"(Bf.f[1]));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (MODE == 0 && a.ekeep && !(TAN_SIM_LAB & 4)) c.dl = a.ekeep + ((((long)s * npanel + panel) * nct + ct_) * 4 + gw) * 4096;
        if constexpr (!(TAN_SIM_LAB & 8)) tile_done_res<MODE>(c, acc, rowacc, colrow, ct_);
        else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
        }
    }

    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = rowacc[i][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if ((lane & 31) == 0 && row < R) unsafeAtomicAdd(a.rowsum + (long)s * R + row, v);
            }
    }
    if (MODE == 1 && a.diag) {
        // Same-video corrections (simnce_diag_kernel<true>'s arithmetic) on the 128 rows this workgroup has just written, as the
        // kernel's tail: a separate launch of 768 small blocks queued behind the other sweep's whole-CU workgroups for 50 us on the
        // loss's critical chain.  The stores above are this workgroup's own: drained and fenced, they are visible to its loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int T = a.T, N = a.N;
        const int nrow = min(128, R - m0);
        for (int i = tid; i < nrow * N; i += 512) {
            const int lr = i / N, k = i - lr * N, row = m0 + lr;
            const int b = row / T, t = row - b * T;
            const int m = a.colmap ? a.colmap[b * N + k] : b * N + k;
            if (m < 0) continue;
            bf16_t* out = a.dl + ((long)s * R + row) * Mp + m;
            if (a.row_leak && a.row_leak[row]) { *out = 0; continue; }
            if (a.tgt[((long)b * T + t) * N + k] == 0.f) continue;
            const long ri = (long)s * R + row, ci = (long)s * Mp + m;
            const float e = __expf((a.diag[(((long)s * a.B + b) * T + t) * N + k] - 1.0f) * inv_tau);
            const float pv = a.possum_v[ri], pt = a.possum_t[ci];
            float corr = 0.f;
            if (!a.col_invalid[m] && pv > 0.f) corr += a.g_v[ri] / pv;
            if (pt > 0.f) corr += a.g_t[ci] / pt;
            *out = f2bf(bf2f(*out) - e * corr * inv_tau);
        }
    }
}

// d loss / d logits from the exponentials the statistics sweep kept instead of a second sweep: dl = e * (g_v/rowsum [valid column] +
// g_t/colsum) / tau.  One block per 128 x 128 tile (stage, row panel, column tile): the four wave tiles are read in the accumulator
// order the sweep stored them in (1 KiB per instruction), scaled, parked row-major in the LDS, the same-video corrections
// (simnce_diag_kernel<true>'s arithmetic) applied there, and the tile stored with 16-byte row-contiguous vectors.  2 x 126 MB of HBM
// traffic against a 64-GFLOP recomputation (~180 us in the step).
__global__ __launch_bounds__(256) void simnce_dl_kept_kernel(SimArgs a, int npanel, int nct) {
    constexpr int LD = 136;                                   // bf16 per LDS row (128 + 8: 16-byte aligned, off the bank period)
    __shared__ __attribute__((aligned(16))) bf16_t tile[128 * LD];
    __shared__ float rf[128], cf[128];
    __shared__ unsigned char cv[128];
    __shared__ int crange[2];                                 // sweep columns that hold sentences of this panel's videos: [min, max]
    const int tid = threadIdx.x, s = blockIdx.y;
    const int panel = blockIdx.x / nct, ct = blockIdx.x - panel * nct;
    const int R = a.R, Mp = a.Mp, T = a.T, N = a.N, m0 = panel * 128, c0 = ct * 128;
    const float inv_tau = 1.0f / S_TAU;
    if (tid == 0) { crange[0] = 0x7fffffff; crange[1] = -1; }
    __syncthreads();
    if (a.diag) {
        const int b_lo = m0 / T, b_hi = min(m0 + 127, R - 1) / T;
        for (int p = b_lo * N + tid; p < (b_hi + 1) * N; p += 256) {
            const int m = a.colmap ? a.colmap[p] : p;
            if (m >= 0) { atomicMin(&crange[0], m); atomicMax(&crange[1], m); }
        }
    }
    if (tid < 128) {
        const long ri = (long)s * R + min(m0 + tid, R - 1);
        rf[tid] = a.g_v[ri] / a.rowsum[ri] * inv_tau;
    } else {
        const int c = tid - 128, col = min(c0 + c, Mp - 1);
        const long ci = (long)s * Mp + col;
        cf[c] = a.g_t[ci] / a.colsum[ci] * inv_tau;
        cv[c] = !a.col_invalid[col];
    }
    __syncthreads();
    // 16 bytes per load = one lane's four row pairs of one column (the sweep's store unit); all eight loads in flight at once
    const uint4* E = reinterpret_cast<const uint4*>(a.ekeep) + (((long)s * npanel + panel) * nct + ct) * 2048;
    uint4 ev[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) ev[n] = E[tid + 256 * n];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int v = tid + 256 * n;                         // 16-byte index: [wave][i][j][r / 8][lane] x 4 dwords (row pairs r, r + 1)
        const int gw = v >> 9, ij = (v >> 7) & 3, q = (v >> 6) & 1, ln = v & 63;
        const int col = (gw & 1) * 64 + (ij & 1) * 32 + (ln & 31);
        const unsigned w[4] = {ev[n].x, ev[n].y, ev[n].z, ev[n].w};
        const float c = cf[col];
        const bool ok = cv[col];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (gw >> 1) * 64 + (ij >> 1) * 32 + acc_row(8 * q + 2 * j, ln);
            tile[row * LD + col] = f2bf(__uint_as_float(w[j] << 16) * ((ok ? rf[row] : 0.f) + c));
            tile[(row + 1) * LD + col] = f2bf(__uint_as_float(w[j] & 0xffff0000u) * ((ok ? rf[row + 1] : 0.f) + c));
        }
    }
    __syncthreads();
    if (a.diag && crange[0] <= c0 + 127 && crange[1] >= c0) {          // (block-uniform) 1-2 of a panel's column tiles
        const int nrow = min(128, R - m0);
        for (int i = tid; i < nrow * N; i += 256) {
            const int lr = i / N, k = i - lr * N, row = m0 + lr;
            const int b = row / T, t = row - b * T;
            const int m = a.colmap ? a.colmap[b * N + k] : b * N + k;
            if (m < c0 || m >= c0 + 128 || m >= Mp) continue;
            bf16_t* out = tile + lr * LD + (m - c0);
            if (a.row_leak && a.row_leak[row]) { *out = 0; continue; }
            if (a.tgt[((long)b * T + t) * N + k] == 0.f) continue;
            const long ri = (long)s * R + row, ci = (long)s * Mp + m;
            const float e = __expf((a.diag[(((long)s * a.B + b) * T + t) * N + k] - 1.0f) * inv_tau);
            const float pv = a.possum_v[ri], pt = a.possum_t[ci];
            float corr = 0.f;
            if (!a.col_invalid[m] && pv > 0.f) corr += a.g_v[ri] / pv;
            if (pt > 0.f) corr += a.g_t[ci] / pt;
            *out = f2bf(bf2f(*out) - e * corr * inv_tau);
        }
        __syncthreads();
    }
    const bool vec = (Mp & 7) == 0;
    for (int q = tid; q < 128 * 16; q += 256) {
        const int lr = q >> 4, cc = (q & 15) * 8, row = m0 + lr, col = c0 + cc;
        if (row >= R || col >= Mp) continue;
        bf16_t* dst = a.dl + ((long)s * R + row) * Mp + col;
        if (vec) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(tile + lr * LD + cc);
        else for (int j = 0; j < 8 && col + j < Mp; ++j) dst[j] = tile[lr * LD + cc + j];
    }
}

// Same-video corrections of the d-logits as a dense [S, R, N] f32 array (entry (row b*T + t, sentence k) = column colmap[b*N + k] of
// the sweep): what simnce_diag_kernel<true> subtracts -- e (g_v / possum_v [valid column] + g_t / possum_t) / tau on positives -- 0
// where nothing changes, +inf where the entry is to be zeroed (leaked frames).  The one-pass kernels apply it to their tiles in the
// LDS from here: looked up entry by entry inside them, each of the ~8 dependent loads (target, cosine, sums, upstream gradients)
// paid a memory latency per tile -- 7 us a tile, measured with the workgroup's phase clocks.  One block per (video, stage).
__global__ __launch_bounds__(256) void simnce_corr_kernel(const float* __restrict__ diag, const float* __restrict__ tgt,
                                                          const unsigned char* __restrict__ col_invalid, const unsigned char* __restrict__ row_leak,
                                                          const float* __restrict__ possum_v, const float* __restrict__ possum_t,
                                                          const float* __restrict__ g_v, const float* __restrict__ g_t, float* __restrict__ corr,
                                                          int B, int T, int N, const int* __restrict__ colmap, int Mp,
                                                          float* __restrict__ zero = nullptr, long nzero = 0) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int R = B * T;
    const float inv_tau = 1.0f / S_TAU;
    const float* blk = diag + ((long)s * B + b) * T * N;
    const float* tg = tgt + (long)b * T * N;
    float* out = corr + ((long)s * B + b) * T * N;
    if (zero)      // (tan_simfam_bwd: the f32 accumulator of the text-feature gradient GEMM, a memset launch otherwise)
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < nzero; i += (long)gridDim.x * gridDim.y * 256) zero[i] = 0.f;
    constexpr int U = 4;
    for (int i0 = threadIdx.x; i0 < T * N; i0 += 256 * U) {
        int cc[U]; long r[U], c[U]; bool in[U], leak[U];
        float tgv[U], bl[U], pv[U], gv[U], pt[U], gt[U]; unsigned char inval[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i = min(i0 + j * 256, T * N - 1);
            const int t = i / N, k = i - t * N;
            const int m = colmap ? colmap[b * N + k] : b * N + k;
            in[j] = m >= 0 && m < Mp;
            cc[j] = min(max(m, 0), Mp - 1);
            r[j] = (long)s * R + b * T + t;
            c[j] = (long)s * Mp + cc[j];
            leak[j] = row_leak && row_leak[b * T + t];
            tgv[j] = tg[i]; bl[j] = blk[i]; inval[j] = col_invalid[cc[j]];
            pv[j] = possum_v[r[j]]; gv[j] = g_v[r[j]]; pt[j] = possum_t[c[j]]; gt[j] = g_t[c[j]];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (i0 + j * 256 >= T * N) continue;
            float v = 0.f;
            if (in[j]) {
                if (leak[j]) v = INFINITY;
                else if (tgv[j] != 0.f) {
                    const float e = __expf((bl[j] - 1.0f) * inv_tau);
                    float cr = 0.f;
                    if (!inval[j] && pv[j] > 0.f) cr += gv[j] / pv[j];
                    if (pt[j] > 0.f) cr += gt[j] / pt[j];
                    v = e * cr * inv_tau;
                }
            }
            out[i0 + j * 256] = v;
        }
    }
}

// ---- d logits AND d v_hat = dl . t_hat from the kept exponentials, one pass ------------------------------------------------------
// simnce_dl_kept_kernel + the GEMM behind it read the d-logits twice more than needed: the element-wise pass writes them (126 MB per
// family at B = 128), the [S*R, Mp] x [Mp, 512] GEMM reads them back and runs at ~0.16 of the MFMA peak inside the step (K = Mp is
// short, its 128 x 128 tiles stage both operands through the LDS).  Here ONE workgroup (8 waves) owns a 128-row panel of a stage and
// walks its column tiles: the 128 x 128 d-logits tile is built in the LDS exactly as simnce_dl_kept_kernel builds it (row-major, same
// rounding, same-video corrections applied there), leaves for HBM as whole rows (the text-feature gradient still contracts over the
// rows of ALL panels: a second kernel), and -- while it is in the LDS -- is the dl operand of the d v_hat MFMAs:
//   D^T[feature][row] += T^T[feature][col] dl[row][col]^T          (v_mfma_f32_32x32x16_bf16, A = text fragment, B = dl fragment)
// wave w owns features 64 w .. 64 w + 63 of all 128 rows (8 accumulator tiles), so every text element is pulled from the L2 once per
// workgroup: the text operand comes from a fragment-major image of the TRANSPOSED unit text features (simnce_pack_textT_kernel:
// piece (16 columns, 32 features) = one 1-KiB wave-load, lane l = 8 consecutive columns of feature l & 31) through a register ring
// that runs across tiles, like the sweep's.  The transposed accumulators (a lane owns 4 consecutive features of one row per
// register quad) leave through the LDS as whole 1-KiB rows of d v_hat.
// The next tile's exponentials are requested a tile ahead and converted between the MFMA steps of the current one (two tile
// buffers, ONE barrier per tile).
constexpr int DV_LD = 136;                          // bf16 per LDS row of a d-logits tile (as simnce_dl_kept_kernel)
constexpr int DV_TILE_B = 128 * DV_LD * 2;          // 34 KiB
constexpr int DV_RAW_B = 32768;                     // a tile of kept exponentials as the sweep stored it
constexpr int DV_OUT_LD = 520;                      // bf16 per LDS row of the d v_hat panel (epilogue)
constexpr int DV_MAX_N = 32;                        // sentences per video the LDS-resident correction arrays hold
constexpr int DV_OFF_RAW = 2 * DV_TILE_B, DV_OFF_RF = DV_OFF_RAW + 2 * DV_RAW_B, DV_OFF_CORR = DV_OFF_RF + 512;
constexpr int DV_OFF_MCOL = DV_OFF_CORR + 128 * DV_MAX_N * 4;
constexpr int DV_LDS_B = DV_OFF_MCOL + 128 * DV_MAX_N * 2;          // 156.5 KiB (the epilogue's 130-KiB panel overlays the tile buffers)
constexpr int DV_RD = 8;                            // text-fragment ring depth (steps of 16 columns): one tile

// a wave-uniform 64-bit value the compiler may hold in VGPRs -> SGPRs (the "s" operands of inline asm)
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

// Transposed fragment-major text image: piece (column block cb16 of 16, feature block fb of 32) = 1 KiB at ((cb16 * 16 + fb) * 1024),
// lane l's 16 bytes = Tt[cb16*16 + 8 (l >> 5) + e][fb*32 + (l & 31)], e = 0..7; columns >= Mp are ZERO (they are the K padding of
// the last tile).  grid (32-column blocks, stages), 256 threads.
__global__ __launch_bounds__(256) void simnce_pack_textT_kernel(const bf16_t* __restrict__ Tt, long t_stage_stride, char* __restrict__ TpT,
                                                                long tpt_stage_stride, int Mp) {
    constexpr int LDT = 520;
    __shared__ __attribute__((aligned(16))) bf16_t tl[32 * LDT];
    const int blk = blockIdx.x, st = blockIdx.y, tid = threadIdx.x;
    const bf16_t* src = Tt + (long)st * t_stage_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, r = q >> 6, ch = q & 63, col = blk * 32 + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col < Mp) v = *reinterpret_cast<const uint4*>(src + (long)col * 512 + ch * 8);
        *reinterpret_cast<uint4*>(tl + r * LDT + ch * 8) = v;
    }
    __syncthreads();
    char* dst = TpT + (long)st * tpt_stage_stride + (long)blk * 2 * 16 * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int o = tid + 256 * i, l = o & 63, fb = (o >> 6) & 15, k2 = o >> 10;
        const bf16_t* g = tl + (k2 * 16 + 8 * (l >> 5)) * LDT + fb * 32 + (l & 31);
        uint4 u;
        u.x = (unsigned)g[0] | ((unsigned)g[LDT] << 16);
        u.y = (unsigned)g[2 * LDT] | ((unsigned)g[3 * LDT] << 16);
        u.z = (unsigned)g[4 * LDT] | ((unsigned)g[5 * LDT] << 16);
        u.w = (unsigned)g[6 * LDT] | ((unsigned)g[7 * LDT] << 16);
        *reinterpret_cast<uint4*>(dst + (long)(k2 * 16 + fb) * 1024 + l * 16) = u;
    }
}

// `l2` (tan_simfam_bwd): the backward of the L2 normalisation v_hat = x / |x| as the epilogue -- dx = (d v_hat - v_hat <v_hat, d v_hat>) / |x|
// from the bf16 d v_hat panel in the LDS, stored straight into the stack's stage-gradient rows: d v_hat never exists in HBM and the
// l2n_bwd launch (read d v_hat + v_hat, write dx: 150 MB per family at B = 128) is gone.  Same per-lane element order and reduction as
// l2n_bwd_kernel (tan_norm.hip).
struct FamL2 { FamPtrs dx; const float* inv; int grp; long dst_grp_rows, dst_off; };
__global__ __launch_bounds__(512) void simnce_dl_dvn_kernel(SimArgs a, int npanel, int nct, const char* __restrict__ TpT, long tpt_stage_stride,
                                                            bf16_t* __restrict__ dvn, FamL2 l2) {
    extern __shared__ __attribute__((aligned(1024))) char dv_lds[];
    __shared__ int crange[2];                                 // sweep columns that hold sentences of this panel's videos: [min, max]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid over (stage, row panel) items, a contiguous run of items per XCD (they share a stage's text image behind one L2)
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int s = wg / npanel, panel = wg - s * npanel, m0 = panel * 128;
    const int R = a.R, Mp = a.Mp, T = a.T, N = a.N;
    const float inv_tau = 1.0f / S_TAU;
    float* const rfb = reinterpret_cast<float*>(dv_lds + DV_OFF_RF);           // [128] g_v / rowsum / tau of the panel's rows
    float* const corrL = reinterpret_cast<float*>(dv_lds + DV_OFF_CORR);       // [128][N] corrections of the panel's same-video entries (a.corr)
    short* const mL = reinterpret_cast<short*>(dv_lds + DV_OFF_MCOL);          // [128][N] their sweep columns, or -1
    if (tid == 0) { crange[0] = 0x7fffffff; crange[1] = -1; }
    __syncthreads();
    const int nrow = min(128, R - m0);
    if (a.corr) {
        // the panel's rows of the correction array and their columns, once: the per-tile pass below touches the LDS only
        const float* cp = a.corr + ((long)s * R + m0) * N;
        for (int i = tid; i < nrow * N; i += 512) {
            const int lr = i / N, k = i - lr * N, bv = (m0 + lr) / T;
            int m = a.colmap ? a.colmap[bv * N + k] : bv * N + k;
            const float v = cp[i];
            if (m < 0 || m >= Mp) m = -1;
            corrL[i] = v; mL[i] = (short)m;
            if (m >= 0 && v != 0.f) { atomicMin(&crange[0], m); atomicMax(&crange[1], m); }
        }
    }
    if (tid < 128) {
        const long ri = (long)s * R + min(m0 + tid, R - 1);
        rfb[tid] = a.g_v[ri] / a.rowsum[ri] * inv_tau;
    }
    // The kept exponentials of a tile travel HBM -> LDS by LDS-DMA (no registers, requested a whole tile ahead): thread `tid` owns the
    // four 16-byte units tid + 512 n of the tile as the sweep stored it (wave quadrant n of the sweep, its accumulator tile ij,
    // register half qh) = 2 column sets (n & 1) x 2 row sets (n >> 1) of 8 rows, and reads back exactly the bytes its own wave
    // requested: the wait for them is the wave's own vmcnt (every later load of the ring is behind them in the in-order queue).
    // (Issued from inline asm: told about an LDS-DMA write, hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read it cannot tell
    // apart from the destination -- the conversion's and the row pieces' -- which drains the ring at every step.  Loads the compiler
    // does not count only make its own vmcnt waits more conservative: the queue is in order.)
    const char* Eb = reinterpret_cast<const char*>(a.ekeep) + (((long)s * npanel + panel) * nct) * (long)DV_RAW_B;
    const unsigned raw_lds0 = (unsigned)(uintptr_t)(dv_lds + DV_OFF_RAW) + (unsigned)wave * 1024u;
    auto raw_request = [&](int ct) __attribute__((always_inline)) {
        const char* src = sgpr_ptr(Eb + (long)ct * DV_RAW_B);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(raw_lds0 + (unsigned)((ct & 1) * DV_RAW_B + n * 8192)));
            const unsigned voff = (unsigned)(tid * 16 + n * 8192);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
        }
    };
    const int ij = (tid >> 7) & 3, qh = (tid >> 6) & 1;
    const int colb = (ij & 1) * 32 + (lane & 31);                              // + 64 (n & 1)
    const int rowb = (ij >> 1) * 32 + 16 * qh + 4 * (lane >> 5);               // + 64 (n >> 1) + 8 (j >> 1) + (2 j & 3) + h
    // column factors of this thread's two column sets of a tile: requested a tile and a half before the conversion uses them (6 registers)
    float cgt[2], ccs[2];
    int cinv[2];
    auto cols_load = [&](int ct) __attribute__((always_inline)) {
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int col = min(ct * 128 + cs * 64 + colb, Mp - 1);
            const long ci = (long)s * Mp + col;
            cgt[cs] = a.g_t[ci]; ccs[cs] = a.colsum[ci]; cinv[cs] = a.col_invalid[col];
        }
    };
    // one 16-byte unit: 8 rows (two float4 of row factors) of one column -> the row-major bf16 tile
    auto convert_unit = [&](int ct, int n) __attribute__((always_inline)) {
        const int cs = n & 1, rs = n >> 1;
        const uint4 ev = *reinterpret_cast<const uint4*>(dv_lds + DV_OFF_RAW + (ct & 1) * DV_RAW_B + (n * 512 + tid) * 16);
        const float cf = cgt[cs] / ccs[cs] * inv_tau;
        const bool ok = !cinv[cs];
        const float4 r0 = *reinterpret_cast<const float4*>(rfb + rs * 64 + rowb), r1 = *reinterpret_cast<const float4*>(rfb + rs * 64 + rowb + 8);
        const float rf[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        const unsigned w[4] = {ev.x, ev.y, ev.z, ev.w};
        bf16_t* tp = reinterpret_cast<bf16_t*>(dv_lds + (ct & 1) * DV_TILE_B) + (rs * 64 + rowb) * DV_LD + cs * 64 + colb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ro = 8 * (j >> 1) + ((2 * j) & 3);
            const unsigned pk = f2bf2(__uint_as_float(w[j] << 16) * ((ok ? rf[2 * j] : 0.f) + cf),
                                       __uint_as_float(w[j] & 0xffff0000u) * ((ok ? rf[2 * j + 1] : 0.f) + cf));
            tp[ro * DV_LD] = (bf16_t)(pk & 0xffffu);
            tp[(ro + 1) * DV_LD] = (bf16_t)(pk >> 16);
        }
    };
    // same-video corrections of tile ct in the LDS (simnce_diag_kernel<true>'s arithmetic on the precomputed values; block-uniform
    // condition: 1-2 tiles of a panel)
    auto diag_tile = [&](int ct) __attribute__((always_inline)) {
        const int c0 = ct * 128;
        bf16_t* tb = reinterpret_cast<bf16_t*>(dv_lds + (ct & 1) * DV_TILE_B);
        if (a.corr && crange[0] <= c0 + 127 && crange[1] >= c0) {
            for (int i = tid; i < nrow * N; i += 512) {
                const float v = corrL[i];
                const int m = mL[i];
                if (v == 0.f || m < c0 || m >= c0 + 128) continue;
                bf16_t* out = tb + (i / N) * DV_LD + (m - c0);
                *out = v == INFINITY ? (bf16_t)0 : f2bf(bf2f(*out) - v);
            }
            __syncthreads();
        }
    };
    // one 16-byte row piece of tile ct -> dl (Mp % 8 == 0: the host checks)
    auto store_unit = [&](int ct, int i) __attribute__((always_inline)) {
        const int q = tid + 512 * i, lr = q >> 4, cc = (q & 15) * 8, row = m0 + lr, col = ct * 128 + cc;
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(dv_lds + (ct & 1) * DV_TILE_B) + lr * DV_LD + cc);
        if (a.dl && row < R && col < Mp) *reinterpret_cast<uint4*>(a.dl + ((long)s * R + row) * Mp + col) = v;
    };

    struct BT { bf16x8 f[2]; };
    BT ring[DV_RD];
    const char* Tps = TpT + (long)s * tpt_stage_stride + (long)wave * 2048 + lane * 16;
    const int nk = nct * 8;
    auto load_bt = [&](BT& dst, int kk) __attribute__((always_inline)) {
        const char* p = Tps + (long)kk * 16384;
        dst.f[0] = *reinterpret_cast<const bf16x8*>(p);
        dst.f[1] = *reinterpret_cast<const bf16x8*>(p + 1024);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);

    // ---- prologue: tile 0 converted, tile 1 requested, the ring filled behind them
    raw_request(0);
    if (nct > 1) raw_request(1);
    cols_load(0);
    pn_static_for_s<0, DV_RD>([&](auto jc) { constexpr int J = decltype(jc)::value; load_bt(ring[J], min(J, nk - 1)); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // rfb, corrL, mL, crange
#pragma unroll
    for (int n = 0; n < 4; ++n) convert_unit(0, n);
    if (nct > 1) cols_load(1);
    __syncthreads();
    diag_tile(0);

    typedef __attribute__((address_space(3))) const bf16x8* lds_frag_t;
    for (int ct = 0; ct < nct; ++ct) {
        const unsigned abase = (unsigned)(uintptr_t)(dv_lds + (ct & 1) * DV_TILE_B) + ((lane & 31) * DV_LD + 8 * (lane >> 5)) * 2;
        bf16x8 af[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) af[rb] = *(lds_frag_t)(uintptr_t)(abase + rb * 32 * DV_LD * 2);
        // Per tile and wave: [step 0] the exponentials of tile ct + 2 are requested IN FRONT of the step's ring load -- that load is
        // waited for at step 0 of the next tile, and the queue is in order, so the conversion of tile ct + 1 [steps 1-4] finds its
        // bytes landed (requested a tile ago) without a wait of its own; [steps 4-7] tile ct leaves for HBM one row piece per step.
        pn_static_for_s<0, 8>([&](auto jc) {
            constexpr int KS = decltype(jc)::value;
            BT& Bt = ring[KS % DV_RD];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bt.f[j], af[rb], acc[rb][j], 0, 0, 0);
                if constexpr (KS < 7) af[rb] = *(lds_frag_t)(uintptr_t)(abase + (rb * 32 * DV_LD + (KS + 1) * 16) * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KS == 0) {
                if (ct + 2 < nct) raw_request(ct + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (KS >= 1 && KS <= 4) { if (ct + 1 < nct) convert_unit(ct + 1, KS - 1); }
            if constexpr (KS >= 4) store_unit(ct, KS - 4);
            if constexpr (KS == 5) { if (ct + 2 < nct) cols_load(ct + 2); }
            load_bt(Bt, min(ct * 8 + KS + DV_RD, nk - 1));
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        if (ct + 1 < nct) diag_tile(ct + 1);
    }

    // ---- epilogue: acc[rb][j][r] = d v_hat[row rb*32 + (lane & 31)][feature 64 wave + 32 j + acc_row(r, lane)] -> LDS rows -> 16-byte stores
    // (with `l2`: the unit rows and norms the normalisation's backward needs are requested NOW, so that their HBM latency passes under
    // the parking of the accumulators; wave w finishes rows 16 w .. 16 w + 15, lane l = elements 8 l .. 8 l + 7)
    uint4 yv[16];
    float invl = 0.f;
    if (l2.inv) {
        const bf16_t* Y = a.V + ((long)s * R + m0) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) yv[i] = *reinterpret_cast<const uint4*>(Y + (long)min(wave * 16 + i, nrow - 1) * 512);
        invl = l2.inv[(long)s * R + m0 + min(wave * 16 + (lane & 15), nrow - 1)];
    }
    bf16_t* outp = reinterpret_cast<bf16_t*>(dv_lds);          // (every wave is behind the last tile's barrier: the tile buffers are free)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = f2bf2(acc[rb][j][4 * g], acc[rb][j][4 * g + 1]);
                u.y = f2bf2(acc[rb][j][4 * g + 2], acc[rb][j][4 * g + 3]);
                *reinterpret_cast<uint2*>(outp + (rb * 32 + (lane & 31)) * DV_OUT_LD + wave * 64 + j * 32 + 8 * g + 4 * (lane >> 5)) = u;
            }
    __syncthreads();
    if (l2.inv) {
        bf16_t* dxs = reinterpret_cast<bf16_t*>(l2.dx.p[s]);
        auto unpack = [](const uint4& u, float (&x)[8]) __attribute__((always_inline)) {
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = wave * 16 + i, row = m0 + lr;
            float d[8], y[8];
            unpack(*reinterpret_cast<const uint4*>(outp + lr * DV_OUT_LD + lane * 8), d);
            unpack(yv[i], y);
            float dot = ((d[0] * y[0] + d[1] * y[1]) + (d[2] * y[2] + d[3] * y[3])) + ((d[4] * y[4] + d[5] * y[5]) + (d[6] * y[6] + d[7] * y[7]));
            dot = wave_sum(dot);
            const float inv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invl), i));
            uint4 o;
            o.x = f2bf2((d[0] - y[0] * dot) * inv, (d[1] - y[1] * dot) * inv); o.y = f2bf2((d[2] - y[2] * dot) * inv, (d[3] - y[3] * dot) * inv);
            o.z = f2bf2((d[4] - y[4] * dot) * inv, (d[5] - y[5] * dot) * inv); o.w = f2bf2((d[6] - y[6] * dot) * inv, (d[7] - y[7] * dot) * inv);
            if (row < R)                                             // (wave-uniform)
                *reinterpret_cast<uint4*>(dxs + ((long)(row / l2.grp) * l2.dst_grp_rows + l2.dst_off + row % l2.grp) * 512 + lane * 8) = o;
        }
        return;
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int q = tid + 512 * i, lr = q >> 6, ch = q & 63;
        if (m0 + lr < R)
            *reinterpret_cast<uint4*>(dvn + ((long)s * R + m0 + lr) * 512 + ch * 8) = *reinterpret_cast<const uint4*>(outp + lr * DV_OUT_LD + ch * 8);
    }
}

// the frame-panel-resident sweep takes C = 512; other channel counts run the re-staging kernel
static bool res_enabled(const SimArgs& a) { return a.C == 512; }

// colsum[s,c] = sum over row panels of colpart
__global__ void simnce_col_finalize(const float* __restrict__ colpart, float* __restrict__ colsum, int npanel, long SM) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SM) return;
    float s = 0.f;
#pragma unroll 16
    for (int k = 0; k < npanel; ++k) s += colpart[(long)k * SM + i];      // (independent loads: sixteen in flight)
    colsum[i] = s;
}

// Same-video blocks diag[s,b,t,n] (separately computed cosines): positives (loss.py:73-76) and the padded-frame quirk
// (loss.py:96-101: those entries read -6e4, i.e. e = 0).  One block per (video, stage).
//   DL = false: possum_v / possum_t, and the leaked e's are taken back out of rowsum / colsum
//   DL = true : dl -= e (gv/possum_v [valid col] + gt/possum_t) / tau on positives; dl = 0 on leaked entries
template <bool DL>
__global__ __launch_bounds__(256) void simnce_diag_kernel(const float* __restrict__ diag, const float* __restrict__ tgt,
                                                          const unsigned char* __restrict__ col_invalid,
                                                          const unsigned char* __restrict__ row_leak, float* __restrict__ rowsum,
                                                          float* __restrict__ colsum, float* __restrict__ possum_v,
                                                          float* __restrict__ possum_t, const float* __restrict__ g_v,
                                                          const float* __restrict__ g_t, bf16_t* __restrict__ dl, int B, int T, int N,
                                                          const int* __restrict__ colmap, int Mp) {
    // colmap (optional): padded column b*N+k -> column of the COMPACTED text matrix the sweep ran on, or -1 (dropped pad column)
    const int b = blockIdx.x, s = blockIdx.y;
    const int R = B * T;
    const float inv_tau = 1.0f / S_TAU;
    const float* blk = diag + ((long)s * B + b) * T * N;
    const float* tg = tgt + (long)b * T * N;
    if (!DL) {
        // every thread takes entries of the T x N block; row / column sums meet in LDS (f32 atomics, like the sweep's column sums).
        // One thread per row with a serial loop over the sentences, then one per sentence over the rows, left 3/4 (then 15/16) of the
        // block idle behind chains of dependent loads: 31 us on the loss's critical chain.
        extern __shared__ float dsm[];
        float* accR = dsm; float* lostR = dsm + T; float* accC = dsm + 2 * T; float* lostC = accC + N;
        for (int i = threadIdx.x; i < 2 * (T + N); i += 256) dsm[i] = 0.f;
        __syncthreads();
        for (int i = threadIdx.x; i < T * N; i += 256) {
            const int t = i / N, k = i - t * N;
            const int cc = colmap ? colmap[b * N + k] : b * N + k;
            const bool leak = row_leak && row_leak[b * T + t];
            const bool pos = tg[i] != 0.f;
            if (!leak && !pos) continue;
            const float e = __expf((blk[i] - 1.0f) * inv_tau);
            const bool valid = cc >= 0 && !col_invalid[cc];
            if (leak) {
                if (valid) atomicAdd(&lostR[t], e);
                if (cc >= 0) atomicAdd(&lostC[k], e);
            } else {
                if (valid) atomicAdd(&accR[t], e);
                if (cc >= 0) atomicAdd(&accC[k], e);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += 256) {
            possum_v[(long)s * R + b * T + t] = accR[t];
            if (row_leak && row_leak[b * T + t]) rowsum[(long)s * R + b * T + t] -= lostR[t];
        }
        for (int k = threadIdx.x; k < N; k += 256) {
            const int cc = colmap ? colmap[b * N + k] : b * N + k;
            if (cc < 0) continue;
            possum_t[(long)s * Mp + cc] = accC[k];
            if (lostC[k] != 0.f) colsum[(long)s * Mp + cc] -= lostC[k];
        }
    } else {
        // four entries per thread with every load issued before the first use (clamped addresses, predicated stores): entry by
        // entry, each of the ~10 loads behind its own branch paid a full memory latency (27 us per launch)
        constexpr int U = 4;
        for (int i0 = threadIdx.x; i0 < T * N; i0 += 256 * U) {
            int cc[U]; long r[U], c[U]; bool in[U], leak[U];
            float tgv[U], bl[U], pv[U], gv[U], pt[U], gt[U]; unsigned char inval[U]; bf16_t cur[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i = min(i0 + j * 256, T * N - 1);
                const int t = i / N, k = i - t * N;
                const int m = colmap ? colmap[b * N + k] : b * N + k;
                in[j] = i0 + j * 256 < T * N && m >= 0;
                cc[j] = max(m, 0);
                r[j] = (long)s * R + b * T + t;
                c[j] = (long)s * Mp + cc[j];
                leak[j] = row_leak && row_leak[b * T + t];
                tgv[j] = tg[i]; bl[j] = blk[i]; inval[j] = col_invalid[cc[j]];
                pv[j] = possum_v[r[j]]; gv[j] = g_v[r[j]]; pt[j] = possum_t[c[j]]; gt[j] = g_t[c[j]];
                cur[j] = dl[r[j] * Mp + cc[j]];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (!in[j]) continue;
                bf16_t* out = dl + r[j] * Mp + cc[j];
                if (leak[j]) { *out = 0; continue; }
                if (tgv[j] == 0.f) continue;
                const float e = __expf((bl[j] - 1.0f) * inv_tau);
                float corr = 0.f;
                if (!inval[j] && pv[j] > 0.f) corr += gv[j] / pv[j];
                if (pt[j] > 0.f) corr += gt[j] / pt[j];
                *out = f2bf(bf2f(cur[j]) - e * corr * inv_tau);
            }
        }
    }
}

__global__ void simnce_terms(const float* __restrict__ allsum, const float* __restrict__ possum, float* __restrict__ terms, long n,
                             float log_count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float shift = 1.0f / S_TAU;
    const float den = logf(allsum[i]) + shift;
    const float num = possum[i] > 0.f ? logf(possum[i]) + shift : -6e4f + log_count;
    terms[i] = den - num;
}

// v_terms and t_terms in one launch (they were two launches on the loss's serial chain)
__global__ void simnce_terms2(const float* __restrict__ rowsum, const float* __restrict__ possum_v, float* __restrict__ v_terms, long SR,
                              float log_cols, const float* __restrict__ colsum, const float* __restrict__ possum_t,
                              float* __restrict__ t_terms, long SM, float log_rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SR + SM) return;
    const bool rows = i < SR;
    const long k = rows ? i : i - SR;
    const float all = rows ? rowsum[k] : colsum[k], pos = rows ? possum_v[k] : possum_t[k], lc = rows ? log_cols : log_rows;
    const float shift = 1.0f / S_TAU;
    const float den = logf(all) + shift;
    const float num = pos > 0.f ? logf(pos) + shift : -6e4f + lc;
    (rows ? v_terms : t_terms)[k] = den - num;
}

}  // namespace tal

using namespace tal;

extern "C" int tan_simnce_max_cols(void) {
    SimArgs a{};
    a.C = 512;
    return res_enabled(a) ? S_MAXCOLS_RES : S_MAXCOLS;       // (C = 512, what the aligner runs; other channel counts: 2048)
}

static long simnce_corr_floats(int S, int B, int T, int N) { return (long)S * B * T * N + 128 * 32 + 8; }

extern "C" long tan_simnce_ws_floats(int S, int B, int T, int N) {
    const long R = (long)B * T, Mp = (long)B * N;
    // column partials (two per row panel) + same-video blocks + the fragment-major text image of the resident sweep (bf16, per stage,
    // columns rounded up to 128)
    // ... + the same-video corrections of the one-pass backward kernels ([S, R, N] f32, simnce_corr_kernel; slack for its 1-KiB LDS-DMA pieces)
    return 2 * (long)cdiv(R, 128) * S * Mp + (long)S * B * T * N + (long)S * cdiv(Mp, 128) * 128 * 512 / 2 + 4 + simnce_corr_floats(S, B, T, N);
}
// where the corrections live in `ws` (behind the text image; 16-byte aligned)
static float* simnce_corr_ptr(float* ws, int S, int B, int T, int N) {
    const long R = (long)B * T, Mp = (long)B * N;
    float* p = ws + 2 * (long)cdiv(R, 128) * S * Mp + (long)S * B * T * N + (long)S * cdiv(Mp, 128) * 128 * 512 / 2 + 4;
    return (float*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
}
static int simnce_corr_launch(const SimArgs& a, const float* diag, float* corr, hipStream_t st) {
    hipLaunchKernelGGL(simnce_corr_kernel, dim3(a.B, a.S), dim3(256), 0, st, diag, a.tgt, a.col_invalid, a.row_leak, (const float*)a.possum_v,
                       a.possum_t, a.g_v, a.g_t, corr, a.B, a.T, a.N, a.colmap, a.Mp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static int simnce_common(SimArgs& a, int S, int B, int T, int N, int C) {
    a.S = S; a.B = B; a.T = T; a.N = N; a.C = C; a.R = B * T; a.Mp = B * N;
    if (S <= 0 || B <= 0 || T <= 0 || N <= 0 || C <= 0 || C % 64 != 0) return TAN_ERR_BAD_ARG;
    if (((uintptr_t)a.V % 16) || ((uintptr_t)a.Tt % 16)) return TAN_ERR_BAD_ARG;
    return 0;
}

// same-video cosine blocks diag[s,b,t,n] = <vn[s,b*T+t], tn[s,b*N+n]>, C = 512: one WAVE per 32 frame rows of a (video, stage),
// both MFMA operands straight from global memory in fragment layout (rows K-contiguous: a lane's 8 channels are one 16-byte load),
// 32 k-steps, the 32 x 32 result stored for the N real sentences.  S*B*T*N*C MACs = 0.4 G at B = 128: the 128 x 128-tile GEMM this
// replaces spent 73 us on it (one launch, 768 mostly-padding tiles) or 6 x 9 us (shared text features: one launch per stage), a
// VALU dot-product version 47 us, on the critical chain of the loss.
__global__ __launch_bounds__(256) void simnce_blocks_kernel(const bf16_t* __restrict__ V, const bf16_t* __restrict__ tn_blocks,
                                                            long tb_stage_stride, float* __restrict__ diag, int B, int T, int N, long R) {
    constexpr int C = 512;
    const int b = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = (blockIdx.z * 4 + wave) * 32;
    if (t0 >= T) return;
    const int hi8 = 8 * (lane >> 5);
    const bf16_t* arow = V + ((long)s * R + (long)b * T + min(t0 + (lane & 31), T - 1)) * C + hi8;
    float* out = diag + ((long)s * B + b) * T * N;
    for (int n0 = 0; n0 < N; n0 += 32) {
        const bf16_t* brow = tn_blocks + (long)s * tb_stage_stride + ((long)b * N + min(n0 + (lane & 31), N - 1)) * C + hi8;
        f32x16 acc;
        acc_zero(acc);
#pragma unroll 8
        for (int ks = 0; ks < C; ks += 16) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(arow + ks);
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(brow + ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        const int n = n0 + acc_col(lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + acc_row(r, lane);
            if (t < T && n < N) out[t * N + n] = acc[r];
        }
    }
}

// (other channel counts: through the ordinary GEMM, batch = S*B, or per stage when the text features are shared)
static int simnce_diag_blocks(const SimArgs& a, const bf16_t* tn_blocks, long tb_stage_stride, float* diag, hipStream_t st) {
    if (a.C == 512 && (((uintptr_t)a.V | (uintptr_t)tn_blocks) % 16) == 0 && tb_stage_stride % 8 == 0) {
        hipLaunchKernelGGL(simnce_blocks_kernel, dim3(a.B, a.S, cdiv(a.T, 128)), dim3(256), 0, st, a.V, tn_blocks, tb_stage_stride, diag,
                           a.B, a.T, a.N, (long)a.R);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    }
    // per-stage text features laid out back to back: (stage, video) is ONE batch index for all three operands
    const bool one_launch = tb_stage_stride == (long)a.B * a.N * a.C;
    for (int s = 0; s < (one_launch ? 1 : a.S); ++s) {
        tan_gemm_desc d{};
        d.dtype = TAN_BF16; d.out_dtype = TAN_F32;
        d.M = a.T; d.N = a.N; d.K = a.C; d.a_kc = 1; d.b_kc = 1;
        d.A = a.V + (long)s * a.R * a.C; d.lda = a.C;
        d.B = tn_blocks + (long)s * tb_stage_stride; d.ldb = a.C;
        d.C = diag + (long)s * a.B * a.T * a.N; d.ldc = a.N;
        d.split_k = 1; d.alpha = 1.0f;
        d.batch = one_launch ? a.S * a.B : a.B;
        d.sA = (long)a.T * a.C; d.sB = (long)a.N * a.C; d.sC = (long)a.T * a.N;
        int rc = tan_gemm(&d, st);
        if (rc) return rc;
    }
    return 0;
}

static int simnce_cus() {
    static const int n = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
    return n;
}

// the resident sweep's text image lives behind the same-video blocks in `ws` (tan_simnce_ws_floats); a.Mp = columns of the sweep
static int simnce_pack_text(SimArgs& a, float* ws_after_diag, hipStream_t st, float* zero = nullptr, long nzero = 0) {
    char* base = (char*)(((uintptr_t)ws_after_diag + 15) & ~(uintptr_t)15);
    const int nblk = cdiv(a.Mp, 128) * 4;
    const bool shared = a.t_stage_stride == 0;
    a.Tp = base;
    a.tp_stage_stride = shared ? 0 : (long)nblk * 32 * 1024;
    hipLaunchKernelGGL(simnce_pack_text_kernel, dim3(nblk, shared ? 1 : a.S), dim3(256), 0, st, a.Tt, a.t_stage_stride, base, a.tp_stage_stride, a.Mp, zero, nzero);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// v_terms / t_terms of loss.py:240-253 straight from unit features (no logits); sums are kept for tan_simnce_bwd_dl.
static int simnce_fwd_impl(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                           const unsigned char* row_leak, float* rowsum, float* colsum, float* possum_v, float* possum_t,
                           float* v_terms, float* t_terms, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                           long tb_stage_stride, const int* colmap, int Mc, int phases, void* ekeep, void* stream) {
    TAN_REQUIRE(vn && tn && tgt && col_invalid && rowsum && colsum && possum_v && possum_t && v_terms && t_terms && ws);
    TAN_REQUIRE(!colmap || (tn_blocks && Mc > 0 && Mc <= B * N));
    if (phases == 0) phases = TAN_SIM_SWEEP | TAN_SIM_DIAG | TAN_SIM_TERMS;
    SimArgs a{};
    a.V = (const bf16_t*)vn; a.Tt = (const bf16_t*)tn; a.t_stage_stride = t_stage_stride;
    a.tgt = tgt; a.col_invalid = col_invalid; a.row_leak = row_leak;
    a.rowsum = rowsum; a.possum_v = possum_v;
    int rc = simnce_common(a, S, B, T, N, C);
    if (rc) return rc;
    const int npanel = cdiv(a.R, 128);
    a.colpart = ws;
    float* diag = ws + 2 * (long)npanel * S * a.Mp;      // sized for the padded column count
    if (colmap) a.Mp = Mc;
    else { tn_blocks = tn; tb_stage_stride = t_stage_stride; }
    if (a.Mp > (res_enabled(a) ? S_MAXCOLS_RES : S_MAXCOLS)) return TAN_ERR_BAD_ARG;        // columns of the sweep (compacted or not)
    hipStream_t st = (hipStream_t)stream;
    const long SM = (long)S * a.Mp, SR = (long)S * a.R;
    if (phases & TAN_SIM_SWEEP) {
        const bool res = res_enabled(a);
        const bool zero_rows = !(phases & TAN_SIM_ACC_ROWS);
        if (zero_rows && !res) {
            hipError_t e = hipMemsetAsync(rowsum, 0, sizeof(float) * (size_t)S * a.R, st);
            if (e != hipSuccess) return (int)e;
        }
        const int prec = prof_begin(st, TAN_PROF_SIMNCE, 2.0 * S * a.R * (double)a.Mp * C);
        if (ekeep && !res) return TAN_ERR_BAD_ARG;          // tan_simnce_keeps() said no
        a.ekeep = (bf16_t*)ekeep;
        if (res && (rc = simnce_pack_text(a, diag + (long)S * B * T * N, st, zero_rows ? rowsum : nullptr, (long)S * a.R))) return rc;
        if (res) {
            const int items = npanel * S, ncu = simnce_cus(), rem = items % ncu;
            a.npanel = npanel;
            // (cutting the items of the last partial round in two column halves -- a.nfull = items - rem, the kernel supports it -- was
            // measured: 130 vs 125 us per sweep, no gain: off)
            a.nfull = items; (void)ncu; (void)rem;
            hipLaunchKernelGGL((simnce_res_kernel<0>), dim3(a.nfull + 2 * (items - a.nfull)), dim3(512), 0, st, a);
        }
        else hipLaunchKernelGGL((simnce_kernel<0>), dim3(npanel, S), dim3(256), 0, st, a);
        prof_end(st, prec);
        TAN_LAUNCH_CHECK();
        hipLaunchKernelGGL(simnce_col_finalize, dim3(cdiv(SM, 64)), dim3(64), 0, st, a.colpart, colsum, res ? 2 * npanel : npanel, SM);
    }
    if (phases & TAN_SIM_DIAG) {
        if ((rc = simnce_diag_blocks(a, (const bf16_t*)tn_blocks, tb_stage_stride, diag, st))) return rc;
        hipLaunchKernelGGL((simnce_diag_kernel<false>), dim3(B, S), dim3(256), sizeof(float) * 2 * (size_t)(T + N), st, diag, tgt, col_invalid, row_leak, rowsum, colsum,
                           possum_v, possum_t, (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, B, T, N, colmap, a.Mp);
    }
    if (phases & TAN_SIM_TERMS) {
        hipLaunchKernelGGL(simnce_terms2, dim3(cdiv(SR + SM, 256)), dim3(256), 0, st, rowsum, possum_v, v_terms, SR, logf((float)a.Mp), colsum,
                           possum_t, t_terms, SM, logf((float)a.R));
    }
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_simnce_fwd(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                              const unsigned char* row_leak, float* rowsum, float* colsum, float* possum_v, float* possum_t,
                              float* v_terms, float* t_terms, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                              long tb_stage_stride, const int* colmap, int Mc, int phases, void* stream) {
    return simnce_fwd_impl(vn, tn, t_stage_stride, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t, v_terms, t_terms, ws,
                           S, B, T, N, C, tn_blocks, tb_stage_stride, colmap, Mc, phases, nullptr, stream);
}

// 1 when tan_simnce_fwd_keep / tan_simnce_bwd_dl_kept are available for this channel count
extern "C" int tan_simnce_keeps(int C) {
    SimArgs a{}; a.C = C;
    return res_enabled(a) ? 1 : 0;
}

extern "C" long tan_simnce_keep_elems(int S, int R, int Mp) { return (long)S * cdiv(R, 128) * cdiv(Mp, 128) * 16384; }

extern "C" int tan_simnce_fwd_keep(const void* vn, const void* tn, long t_stage_stride, const float* tgt, const unsigned char* col_invalid,
                                   const unsigned char* row_leak, float* rowsum, float* colsum, float* possum_v, float* possum_t,
                                   float* v_terms, float* t_terms, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                                   long tb_stage_stride, const int* colmap, int Mc, int phases, void* e_keep, void* stream) {
    TAN_REQUIRE(e_keep);
    return simnce_fwd_impl(vn, tn, t_stage_stride, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t, v_terms, t_terms, ws,
                           S, B, T, N, C, tn_blocks, tb_stage_stride, colmap, Mc, phases, e_keep, stream);
}

// d loss / d logits [S, R, Mp] in bf16 from upstream g_v [S,R], g_t [S,Mp] (recomputes every logit tile); ws as in fwd
static int simnce_bwd_impl(const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                           const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum,
                           const float* colsum, const float* possum_v, const float* possum_t, const float* g_v, const float* g_t,
                           void* dl, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks, long tb_stage_stride,
                           const int* colmap, int Mc, int phases, const void* ekeep, void* dvn, void* stream) {
    TAN_REQUIRE(vn && tn && tgt && col_invalid && rowsum && colsum && possum_v && possum_t && g_v && g_t && (dl || dvn) && ws);
    TAN_REQUIRE(!colmap || (tn_blocks && Mc > 0 && Mc <= B * N));
    if (phases == 0) phases = TAN_SIM_SWEEP | TAN_SIM_DIAG;
    TAN_REQUIRE(!dvn || (ekeep && (phases & TAN_SIM_SWEEP) && (uintptr_t)dvn % 16 == 0));
    SimArgs a{};
    a.V = (const bf16_t*)vn; a.Tt = (const bf16_t*)tn; a.t_stage_stride = t_stage_stride;
    a.tgt = tgt; a.col_invalid = col_invalid; a.row_leak = row_leak;
    a.rowsum = (float*)rowsum; a.possum_v = (float*)possum_v; a.colsum = colsum; a.possum_t = possum_t;
    a.g_v = g_v; a.g_t = g_t; a.dl = (bf16_t*)dl;
    int rc = simnce_common(a, S, B, T, N, C);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* diag = ws + 2 * (long)cdiv(a.R, 128) * S * a.Mp;
    if (colmap) a.Mp = Mc;
    else { tn_blocks = tn; tb_stage_stride = t_stage_stride; }
    if (a.Mp > (res_enabled(a) ? S_MAXCOLS_RES : S_MAXCOLS)) return TAN_ERR_BAD_ARG;        // columns of the sweep (compacted or not)
    if (ekeep && (phases & TAN_SIM_SWEEP)) {        // the statistics sweep kept its exponentials: one element-wise pass, corrections as its tail
        const bool tail = (phases & TAN_SIM_DIAG) != 0;
        if (tail && !(phases & TAN_SIM_DIAG_KEEP) && (rc = simnce_diag_blocks(a, (const bf16_t*)tn_blocks, tb_stage_stride, diag, st))) return rc;
        a.ekeep = (bf16_t*)ekeep; a.diag = tail ? diag : nullptr; a.colmap = colmap;
        if (dvn && tail) {          // the corrections as a dense array (once per backward: TAN_SIM_CORR_KEEP = the other one-pass call made it)
            float* corr = simnce_corr_ptr(ws, S, B, T, N);
            if (!(phases & TAN_SIM_CORR_KEEP) && (rc = simnce_corr_launch(a, diag, corr, st))) return rc;
            a.corr = corr;
        }
        if (dvn && (a.Mp % 8 || N > DV_MAX_N || a.Mp > 32767)) return TAN_ERR_BAD_ARG;          // (16-byte row pieces of dl; corrections in the LDS)
        if (dvn) {          // d logits + d v_hat in one pass; the transposed text image takes the place of the sweep's (not read again)
            const int npanel = cdiv(a.R, 128), nct = cdiv(a.Mp, 128);
            char* TpT = (char*)(((uintptr_t)(diag + (long)S * B * T * N) + 15) & ~(uintptr_t)15);
            const bool shared = a.t_stage_stride == 0;
            const long tpt_stride = shared ? 0 : (long)nct * 128 * 1024;
            hipLaunchKernelGGL(simnce_pack_textT_kernel, dim3(nct * 4, shared ? 1 : S), dim3(256), 0, st, a.Tt, a.t_stage_stride, TpT, tpt_stride, a.Mp);
            TAN_LAUNCH_CHECK();
            static std::atomic<unsigned long long> lds_done{0};
        const hipError_t attr = ensure_dyn_lds((const void*)simnce_dl_dvn_kernel, DV_LDS_B, lds_done);
            if (attr != hipSuccess) return (int)attr;
            const int prec = prof_begin(st, TAN_PROF_GEMM_BF16 + 1, 2.0 * S * a.R * (double)a.Mp * C);
            hipLaunchKernelGGL(simnce_dl_dvn_kernel, dim3(npanel * S), dim3(512), DV_LDS_B, st, a, npanel, nct, (const char*)TpT, tpt_stride, (bf16_t*)dvn, FamL2{});
            prof_end(st, prec);
            TAN_LAUNCH_CHECK();
            return 0;
        }
        hipLaunchKernelGGL(simnce_dl_kept_kernel, dim3(cdiv(a.R, 128) * cdiv(a.Mp, 128), S), dim3(256), 0, st, a, cdiv(a.R, 128), cdiv(a.Mp, 128));
        TAN_LAUNCH_CHECK();
        return 0;
    }
    if (phases & TAN_SIM_SWEEP) {
        const bool res = res_enabled(a);
        const bool tail = res && (phases & TAN_SIM_DIAG);      // corrections as the sweep kernel's tail
        if (tail && !(phases & TAN_SIM_DIAG_KEEP) && (rc = simnce_diag_blocks(a, (const bf16_t*)tn_blocks, tb_stage_stride, diag, st))) return rc;
        const int prec = prof_begin(st, TAN_PROF_SIMNCE, 2.0 * S * a.R * (double)a.Mp * C);
        if (res) {
            a.diag = tail ? diag : nullptr; a.colmap = colmap;
            if ((rc = simnce_pack_text(a, diag + (long)S * B * T * N, st))) return rc;
            a.npanel = cdiv(a.R, 128); a.nfull = a.npanel * S;
            hipLaunchKernelGGL((simnce_res_kernel<1>), dim3(a.nfull), dim3(512), 0, st, a);
            if (tail) phases &= ~TAN_SIM_DIAG;
        } else hipLaunchKernelGGL((simnce_kernel<1>), dim3(cdiv(a.R, 128), S), dim3(256), 0, st, a);
        prof_end(st, prec);
        TAN_LAUNCH_CHECK();
    }
    if (phases & TAN_SIM_DIAG) {
        if (!(phases & TAN_SIM_DIAG_KEEP) && (rc = simnce_diag_blocks(a, (const bf16_t*)tn_blocks, tb_stage_stride, diag, st))) return rc;
        hipLaunchKernelGGL((simnce_diag_kernel<true>), dim3(B, S), dim3(256), 0, st, diag, tgt, col_invalid, row_leak, (float*)rowsum,
                           (float*)colsum, (float*)possum_v, (float*)possum_t, g_v, g_t, (bf16_t*)dl, B, T, N, colmap, a.Mp);
    }
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_simnce_bwd_dl(const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                                 const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum,
                                 const float* colsum, const float* possum_v, const float* possum_t, const float* g_v, const float* g_t,
                                 void* dl, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks, long tb_stage_stride,
                                 const int* colmap, int Mc, int phases, void* stream) {
    return simnce_bwd_impl(vn, tn, t_stage_stride, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t, g_v, g_t, dl, ws,
                           S, B, T, N, C, tn_blocks, tb_stage_stride, colmap, Mc, phases, nullptr, nullptr, stream);
}

extern "C" int tan_simnce_bwd_dl_kept(const void* e_keep, const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                                      const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum,
                                      const float* colsum, const float* possum_v, const float* possum_t, const float* g_v,
                                      const float* g_t, void* dl, float* ws, int S, int B, int T, int N, int C, const void* tn_blocks,
                                      long tb_stage_stride, const int* colmap, int Mc, int phases, void* stream) {
    TAN_REQUIRE(e_keep);
    return simnce_bwd_impl(vn, tn, t_stage_stride, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t, g_v, g_t, dl, ws,
                           S, B, T, N, C, tn_blocks, tb_stage_stride, colmap, Mc, phases, e_keep, nullptr, stream);
}

// tan_simnce_bwd_dl_kept that ALSO returns d v_hat [S, R, C] = dl . t_hat (bf16; C = 512): the d-logits tile feeds the MFMAs while it
// is in the LDS (simnce_dl_dvn_kernel), so the [S*R, Mp] x [Mp, C] GEMM and its read of the d-logits are gone.  dl is still written
// (row-major [S, R, Mp]) for the text-feature gradient.  Overwrites the sweep's text image in `ws` (tan_simnce_ws_floats).
extern "C" int tan_simnce_bwd_dl_dvn_kept(const void* e_keep, const void* vn, const void* tn, long t_stage_stride, const float* tgt,
                                          const unsigned char* col_invalid, const unsigned char* row_leak, const float* rowsum,
                                          const float* colsum, const float* possum_v, const float* possum_t, const float* g_v,
                                          const float* g_t, void* dl, void* d_vn, float* ws, int S, int B, int T, int N, int C,
                                          const void* tn_blocks, long tb_stage_stride, const int* colmap, int Mc, int phases, void* stream) {
    TAN_REQUIRE(e_keep && d_vn && C == 512);
    if (phases == 0) phases = TAN_SIM_SWEEP | TAN_SIM_DIAG;
    return simnce_bwd_impl(vn, tn, t_stage_stride, tgt, col_invalid, row_leak, rowsum, colsum, possum_v, possum_t, g_v, g_t, dl, ws,
                           S, B, T, N, C, tn_blocks, tb_stage_stride, colmap, Mc, phases, e_keep, d_vn, stream);
}

// =================================================================================================================================
// One feature family from the stacks' stage outputs to their stage gradients (tan_simfam_fwd / tan_simfam_bwd, include/tan_hip.h).
// What ran between a stack's forward and its backward in the training step was 18 launches per family (l2n_fwd x 2, a gather,
// pack_text, the sweep, col_finalize, blocks, diag, terms2 | corr, pack_textT, the one-pass d-logits + d v_hat kernel, fill, GEMM,
// cast, rows_gather, l2n_bwd x 2), eleven of them 5-35 us of latency each on the critical chain of the stack (~230 us per family,
// profiles/r04_*kernel_stats.csv).  Here: [l2n_fwd of the frame rows] -> text -> sweep -> finish | one-pass kernel with the
// normalisation's backward as its epilogue -> GEMM -> text gradient.
namespace tal {

struct FamText {
    FamPtrs xs;                       // raw text rows per text stage
    long grp_rows, off;               // padded sentence m = b*N + k -> row (m / N) * grp_rows + off + m % N
    int N, Mc;
    const long long* idx;             // [Mc] sweep column -> padded sentence, or null (identity)
    bf16_t* tn; float* inv_t;         // [St, Mc, 512], [St, Mc]
    char* Tp; char* TpT; long img_stage_stride;
    float* zero; long nzero;
};

// Unit text features of the sweep's columns in ONE launch: gather (column compaction) + L2 normalisation (l2n_fwd_kernel's arithmetic) +
// the fragment-major image of the statistics sweep (simnce_pack_text_kernel's format) + the transposed one of the one-pass backward
// (simnce_pack_textT_kernel's).  One block per 32 columns and text stage; also zeroes the sweep's row sums.
__global__ __launch_bounds__(256) void simfam_text_kernel(FamText a) {
    constexpr int LDT = 520;
    __shared__ __attribute__((aligned(16))) bf16_t tl[32 * LDT];
    const int blk = blockIdx.x, st = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (a.zero)
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < a.nzero; i += (long)gridDim.x * gridDim.y * 256) a.zero[i] = 0.f;
    const bf16_t* xs = reinterpret_cast<const bf16_t*>(a.xs.p[st]);
    float4 v[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cs = min(blk * 32 + w * 8 + i, a.Mc - 1);          // columns past Mc repeat the last one (Tp) / are zero (TpT)
        const long m = a.idx ? (long)a.idx[cs] : (long)cs;
        const bf16_t* x = xs + ((m / a.N) * a.grp_rows + a.off + m % a.N) * 512;
        v[i][0] = ld4(x + lane * 4);
        v[i][1] = ld4(x + 256 + lane * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) q += (v[i][j].x * v[i][j].x + v[i][j].y * v[i][j].y) + (v[i][j].z * v[i][j].z + v[i][j].w * v[i][j].w);
        const float inv = 1.0f / sqrtf(wave_sum(q));
        const int r = w * 8 + i, c = blk * 32 + r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 y = make_float4(v[i][j].x * inv, v[i][j].y * inv, v[i][j].z * inv, v[i][j].w * inv);
            st4(tl + r * LDT + j * 256 + lane * 4, y);
            if (c < a.Mc) st4(a.tn + ((long)st * a.Mc + c) * 512 + j * 256 + lane * 4, y);
        }
        if (lane == 0 && c < a.Mc) a.inv_t[(long)st * a.Mc + c] = inv;
    }
    __syncthreads();
    {   // piece (column block blk, k step ks): lane l = tn[blk*32 + (l & 31)][16 ks + 8 (l >> 5) ..]
        char* dst = a.Tp + (long)st * a.img_stage_stride + (long)blk * 32 * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = w * 8 + i;
            *reinterpret_cast<uint4*>(dst + (long)ks * 1024) = *reinterpret_cast<const uint4*>(tl + (lane & 31) * LDT + 16 * ks + 8 * (lane >> 5));
        }
    }
    {   // piece (column block of 16, feature block fb of 32): lane l = tn[cb16*16 + 8 (l >> 5) + e][fb*32 + (l & 31)], e = 0..7
        char* dst = a.TpT + (long)st * a.img_stage_stride + (long)blk * 2 * 16 * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = tid + 256 * i, l = o & 63, fb = (o >> 6) & 15, k2 = o >> 10;
            const int r0 = k2 * 16 + 8 * (l >> 5);
            const bf16_t* g = tl + r0 * LDT + fb * 32 + (l & 31);
            uint4 u = make_uint4(0, 0, 0, 0);
            if (blk * 32 + r0 < a.Mc) {                                  // (Mc % 8 == 0: eight columns are all in or all out)
                u.x = (unsigned)g[0] | ((unsigned)g[LDT] << 16);
                u.y = (unsigned)g[2 * LDT] | ((unsigned)g[3 * LDT] << 16);
                u.z = (unsigned)g[4 * LDT] | ((unsigned)g[5 * LDT] << 16);
                u.w = (unsigned)g[6 * LDT] | ((unsigned)g[7 * LDT] << 16);
            }
            *reinterpret_cast<uint4*>(dst + (long)(k2 * 16 + fb) * 1024 + l * 16) = u;
        }
    }
}

struct FamFin {
    const bf16_t* vn; const bf16_t* tn; long tn_stage_stride;        // [S, R, 512]; [St, Mc, 512], stride Mc*512 or 0
    const float* colpart; int nparts;                                // [nparts, S, Mc] column partials of the sweep (two per row panel)
    float* diag;                                                     // [S, B, T, N] same-video cosines (kept for the fallback kernels)
    const float* tgt; const unsigned char* col_invalid; const unsigned char* row_leak; const int* colmap;
    float *rowsum, *colsum, *possum_v, *possum_t, *v_terms, *t_terms;
    const float *g_v, *g_t; float* corr;                             // optional: the backward's same-video corrections [S, R, N]
    float* zero; long nzero;
    int S, B, T, N, Mc; long R;
    float log_cols, log_rows;
};

// Everything between the statistics sweep and the loss terms in ONE launch, one block per (video, stage): the same-video cosine block
// by MFMA from global fragments (simnce_blocks_kernel), the column sums of the video's sentences over the sweep's row-panel partials
// (simnce_col_finalize), positives and leaked frames (simnce_diag_kernel<false>), v_terms / t_terms (simnce_terms2) and -- when the
// upstream gradients of the terms are known already -- the dense correction array of the backward (simnce_corr_kernel).  The cosine
// block stays in the LDS: the four kernels this replaces each re-read it entry by entry behind dependent global loads.  Blocks past
// the videos take the sweep's filler columns (compaction pads up to a multiple of 64; no sentence owns them).
__global__ __launch_bounds__(256) void simfam_finish_kernel(FamFin a) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = blockIdx.y;
    const int B = a.B, T = a.T, N = a.N, Mc = a.Mc, S = a.S;
    const long R = a.R;
    const float inv_tau = 1.0f / S_TAU, shift = 1.0f / S_TAU;
    if (a.zero)
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < a.nzero; i += (long)gridDim.x * gridDim.y * 256) a.zero[i] = 0.f;
    if ((int)blockIdx.x >= B) {                 // filler columns: their sums must be defined (g_t = 0 there, but 0 / garbage may be NaN)
        const int c = ((int)blockIdx.x - B) * 256 + tid;
        if (c < Mc && a.col_invalid[c]) {
            float cs = 0.f;
            for (int p = 0; p < a.nparts; ++p) cs += a.colpart[((long)p * S + s) * Mc + c];
            a.colsum[(long)s * Mc + c] = cs;
            a.possum_t[(long)s * Mc + c] = 0.f;
            a.t_terms[(long)s * Mc + c] = (logf(cs) + shift) - (-6e4f + a.log_rows);
        }
        return;
    }
    const int b = blockIdx.x, TN = T * N;
    float* dg = fsm;                            // [T][N] cosines
    float* ev = dg + TN;                        // [T][N] e of the entries that count (positives; every entry of a leaked frame)
    float* accRl = ev + TN;                     // [T] possum_v
    float* parts = accRl + T;                   // [8][32] column partial sums
    float* colf = parts + 256;                  // [32] x {colsum, possum_t, g_t}
    int* ccs = reinterpret_cast<int*>(colf + 96);                     // [32] sweep column of sentence k, or -1
    unsigned char* validk = reinterpret_cast<unsigned char*>(ccs + 32);   // [32] column takes part in the row sums
    unsigned char* leakf = validk + 32;                                // [T]
    if (tid < 32) {
        int cc = -1;
        if (tid < N) { cc = a.colmap ? a.colmap[b * N + tid] : b * N + tid; if (cc >= Mc) cc = -1; }
        ccs[tid] = cc;
        validk[tid] = cc >= 0 && !a.col_invalid[cc];
    }
    for (int t = tid; t < T; t += 256) leakf[t] = a.row_leak && a.row_leak[(long)b * T + t];
    // ---- same-video cosines: one wave per (32 frames, 32 sentences) unit, both operands straight from global memory
    const int nrb = (T + 31) / 32, nnb = (N + 31) / 32;
    for (int u = wave; u < nrb * nnb; u += 4) {
        const int t0 = (u % nrb) * 32, n0 = (u / nrb) * 32, hi8 = 8 * (lane >> 5);
        const int n = n0 + (lane & 31);
        int ccn = -1;
        if (n < N) ccn = a.colmap ? a.colmap[b * N + n] : b * N + n;
        if (ccn >= Mc) ccn = -1;
        const bf16_t* arow = a.vn + ((long)s * R + (long)b * T + min(t0 + (lane & 31), T - 1)) * 512 + hi8;
        const bf16_t* brow = a.tn + (long)s * a.tn_stage_stride + (long)max(ccn, 0) * 512 + hi8;
        f32x16 acc;
        acc_zero(acc);
#pragma unroll 16
        for (int ks = 0; ks < 512; ks += 16) {
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(arow + ks);
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(brow + ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        float* out = a.diag + ((long)s * B + b) * TN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + acc_row(r, lane), nn = n0 + acc_col(lane);
            if (t < T && nn < N) { dg[t * N + nn] = acc[r]; out[t * N + nn] = acc[r]; }
        }
    }
    __syncthreads();
    // ---- column partial sums of this video's sentences (fixed order: deterministic)
    {
        const int k = tid & 31, p0 = tid >> 5;
        float cs = 0.f;
        if (k < N && ccs[k] >= 0)
            for (int p = p0; p < a.nparts; p += 8) cs += a.colpart[((long)p * S + s) * Mc + ccs[k]];
        parts[p0 * 32 + k] = cs;
    }
    // ---- e of the entries that count
    const float* tg = a.tgt + (long)b * TN;
    for (int i = tid; i < TN; i += 256) {
        const int t = i / N, k = i - t * N;
        const bool counts = ccs[k] >= 0 && (leakf[t] || tg[i] != 0.f);
        ev[i] = counts ? __expf((dg[i] - 1.0f) * inv_tau) : 0.f;
    }
    __syncthreads();
    // ---- rows: positives among the valid columns (or, for a leaked frame, everything the sweep added that reads -6e4 in the reference)
    for (int t = tid; t < T; t += 256) {
        float sum = 0.f;
        for (int k = 0; k < N; ++k) sum += validk[k] ? ev[t * N + k] : 0.f;
        const long ri = (long)s * R + (long)b * T + t;
        const bool leak = leakf[t];
        const float pv = leak ? 0.f : sum;
        float rs = a.rowsum[ri];
        if (leak) { rs -= sum; a.rowsum[ri] = rs; }
        a.possum_v[ri] = pv;
        accRl[t] = pv;
        a.v_terms[ri] = (logf(rs) + shift) - (pv > 0.f ? logf(pv) + shift : -6e4f + a.log_cols);
    }
    // ---- columns
    if (tid < N && ccs[tid] >= 0) {
        const int k = tid, cc = ccs[k];
        float cs = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) cs += parts[p * 32 + k];
        float acc = 0.f, lost = 0.f;
        for (int t = 0; t < T; ++t) { const float e = ev[t * N + k]; if (leakf[t]) lost += e; else acc += e; }
        cs -= lost;
        const long ci = (long)s * Mc + cc;
        a.colsum[ci] = cs;
        a.possum_t[ci] = acc;
        a.t_terms[ci] = (logf(cs) + shift) - (acc > 0.f ? logf(acc) + shift : -6e4f + a.log_rows);
        colf[k] = cs; colf[32 + k] = acc; colf[64 + k] = a.g_t ? a.g_t[ci] : 0.f;
    }
    if (!a.g_v) return;
    __syncthreads();
    // ---- the backward's same-video corrections (simnce_corr_kernel's values)
    float* out = a.corr + ((long)s * B + b) * TN;
    for (int i = tid; i < TN; i += 256) {
        const int t = i / N, k = i - t * N;
        float v = 0.f;
        if (ccs[k] >= 0) {
            if (leakf[t]) v = INFINITY;
            else if (ev[i] != 0.f) {
                const float pv = accRl[t], pt = colf[32 + k];
                float cr = 0.f;
                if (validk[k] && pv > 0.f) cr += a.g_v[(long)s * R + (long)b * T + t] / pv;
                if (pt > 0.f) cr += colf[64 + k] / pt;
                v = ev[i] * cr * inv_tau;
            }
        }
        out[i] = v;
    }
}

struct FamTB {
    const float* acc;                 // [St, Mc, 512] f32 text-feature gradient (split-K sums)
    const bf16_t* tn; const float* inv_t; const int* colmap;
    FamPtrs dx; long grp_rows, off;
    int N, Mc; long Mp;
};

// The text-feature gradient back in the padded sentence order (rows_gather_kernel), through the L2 normalisation's backward
// (l2n_bwd_kernel's arithmetic on the bf16-rounded gradient), into the text rows of the stage gradients; dropped sentences get zeros.
__global__ __launch_bounds__(256) void simfam_text_bwd_kernel(FamTB a) {
    const int lane = threadIdx.x & 63, st = blockIdx.y;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.Mp) return;
    const int cc = a.colmap ? a.colmap[m] : (int)m;
    bf16_t* dx = reinterpret_cast<bf16_t*>(a.dx.p[st]) + ((m / a.N) * a.grp_rows + a.off + m % a.N) * 512 + lane * 4;
    if (cc < 0 || cc >= a.Mc) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        st4(dx, z); st4(dx + 256, z);
        return;
    }
    const float* ap = a.acc + ((long)st * a.Mc + cc) * 512 + lane * 4;
    const bf16_t* yp = a.tn + ((long)st * a.Mc + cc) * 512 + lane * 4;
    float4 d[2], y[2];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 f = ld4(ap + j * 256);
        const unsigned lo = f2bf2(f.x, f.y), hi = f2bf2(f.z, f.w);          // (the gradient as the bf16 tensor it used to be)
        d[j] = make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
        y[j] = ld4(yp + j * 256);
        dot += (d[j].x * y[j].x + d[j].y * y[j].y) + (d[j].z * y[j].z + d[j].w * y[j].w);
    }
    dot = wave_sum(dot);
    const float inv = a.inv_t[(long)st * a.Mc + cc];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        st4(dx + j * 256, make_float4((d[j].x - y[j].x * dot) * inv, (d[j].y - y[j].y * dot) * inv, (d[j].z - y[j].z * dot) * inv,
                                      (d[j].w - y[j].w * dot) * inv));
}

struct FamWs { float* colpart; float* diag; float* corr; char* Tp; char* TpT; long img_stage_stride; long bytes; };
static FamWs simfam_ws(void* ws, int S, int St, int B, int T, int N, int Mc) {
    auto up = [](long v) { return (v + 255) & ~255L; };
    const long R = (long)B * T, npanel = cdiv(R, 128), nblk = (long)cdiv(Mc, 128) * 4;
    FamWs w;
    char* p = (char*)ws;
    long o = 0;
    w.colpart = (float*)(p + o); o += up(2 * npanel * S * (long)Mc * 4);
    w.diag = (float*)(p + o);    o += up((long)S * R * N * 4);
    w.corr = (float*)(p + o);    o += up(((long)S * R * N + 128 * 32 + 8) * 4);
    w.img_stage_stride = nblk * 32 * 1024;
    w.Tp = p + o;                o += up((long)St * w.img_stage_stride);
    w.TpT = p + o;               o += up((long)St * w.img_stage_stride);
    w.bytes = o;
    return w;
}

static int simfam_check(const tan_simfam_desc* d) {
    TAN_REQUIRE(d && d->S >= 1 && d->S <= 8 && (d->St == 1 || d->St == d->S) && d->B > 0 && d->T > 0 && d->N > 0 && d->N <= DV_MAX_N);
    TAN_REQUIRE(d->C == 512 && d->Mc > 0 && d->Mc % 8 == 0 && d->Mc <= S_MAXCOLS_RES && d->Mc <= 32767);
    TAN_REQUIRE((d->idx != nullptr) == (d->colmap != nullptr));
    TAN_REQUIRE(d->idx || d->Mc == d->B * d->N);
    TAN_REQUIRE(d->col_invalid && d->tgt && d->vn && d->inv_v && d->tn && d->inv_t && d->rowsum && d->colsum && d->possum_v && d->possum_t);
    TAN_REQUIRE(d->e_keep && d->ws && ((uintptr_t)d->ws % 256) == 0 && ((uintptr_t)d->vn % 16) == 0 && ((uintptr_t)d->tn % 16) == 0);
    for (int s = 0; s < d->S; ++s) TAN_REQUIRE(d->x_video.p[s] && ((uintptr_t)d->x_video.p[s] % 16) == 0);
    for (int s = 0; s < d->St; ++s) TAN_REQUIRE(d->x_text.p[s] && ((uintptr_t)d->x_text.p[s] % 16) == 0);
    return 0;
}

static void simfam_args(const tan_simfam_desc* d, const FamWs& w, SimArgs& a) {
    a.V = (const bf16_t*)d->vn; a.Tt = (const bf16_t*)d->tn; a.t_stage_stride = d->St == 1 ? 0 : (long)d->Mc * 512;
    a.tgt = d->tgt; a.col_invalid = d->col_invalid; a.row_leak = d->row_leak;
    a.rowsum = d->rowsum; a.possum_v = d->possum_v; a.colpart = w.colpart;
    a.S = d->S; a.B = d->B; a.T = d->T; a.N = d->N; a.C = 512; a.R = d->B * d->T; a.Mp = d->Mc;
    a.ekeep = (bf16_t*)d->e_keep; a.colmap = d->colmap;
    a.npanel = cdiv(a.R, 128); a.nfull = a.npanel * d->S;
    a.Tp = w.Tp; a.tp_stage_stride = d->St == 1 ? 0 : w.img_stage_stride;
}

}  // namespace tal

extern "C" long tan_simfam_ws_bytes(int S, int St, int B, int T, int N, int Mc) {
    return simfam_ws(nullptr, S, St, B, T, N, Mc).bytes;
}

extern "C" long tan_simfam_diag_offset(int S, int St, int B, int T, int N, int Mc, int s) {
    const FamWs w = simfam_ws(nullptr, S, St, B, T, N, Mc);
    return (long)((char*)w.diag - (char*)nullptr) + (long)s * B * T * N * 4;
}

extern "C" int tan_simfam_fwd(tan_simfam_desc* d, void* stream) {
    int rc = simfam_check(d);
    if (rc) return rc;
    TAN_REQUIRE(d->v_terms && d->t_terms && (!d->g_v == !d->g_t));
    const bool do_sweep = !(d->flags & TAN_SIMFAM_FINISH_ONLY), do_finish = !(d->flags & TAN_SIMFAM_SWEEP_ONLY);
    TAN_REQUIRE(do_sweep || do_finish);
    hipStream_t st = (hipStream_t)stream;
    const int S = d->S, St = d->St, B = d->B, T = d->T, N = d->N, Mc = d->Mc;
    const long R = (long)B * T;
    const FamWs w = simfam_ws(d->ws, S, St, B, T, N, Mc);
    d->flags &= ~TAN_SIMFAM_CORR_DONE;
    // ---- unit frame features (tan_model.py:116,136)
    if (do_sweep && !(d->flags & TAN_SIMFAM_NORM_IN_SWEEP)) {
        if ((rc = tan_l2norm_fwd_multi(&d->x_video, d->vn, d->inv_v, S, R, 512, T, (int)d->v_grp_rows, (int)d->v_off, TAN_BF16, stream))) return rc;
    }
    // ---- unit text features of the sweep's columns + both text images; zero the row sums
    if (do_sweep) {
        FamText t{};
        for (int s = 0; s < St; ++s) t.xs.p[s] = (void*)d->x_text.p[s];
        t.grp_rows = d->t_grp_rows; t.off = d->t_off; t.N = N; t.Mc = Mc; t.idx = (const long long*)d->idx;
        t.tn = (bf16_t*)d->tn; t.inv_t = d->inv_t; t.Tp = w.Tp; t.TpT = w.TpT; t.img_stage_stride = w.img_stage_stride;
        t.zero = d->rowsum; t.nzero = (long)S * R;
        hipLaunchKernelGGL(simfam_text_kernel, dim3(cdiv(Mc, 128) * 4, St), dim3(256), 0, st, t);
        TAN_LAUNCH_CHECK();
    }
    // ---- the statistics sweep, exponentials kept (simnce_res_kernel<0>)
    SimArgs a{};
    simfam_args(d, w, a);
    if (d->flags & TAN_SIMFAM_NORM_IN_SWEEP) {       // the sweep stages the RAW stage rows and normalises its panel in the LDS
        for (int s = 0; s < S; ++s) a.xraw.p[s] = (void*)d->x_video.p[s];
        a.x_grp_rows = d->v_grp_rows; a.x_off = d->v_off; a.inv_v = d->inv_v;
    }
    if (do_sweep) {
        const int prec = prof_begin(st, TAN_PROF_SIMNCE, 2.0 * S * R * (double)Mc * 512);
        hipLaunchKernelGGL((simnce_res_kernel<0>), dim3(a.nfull), dim3(512), 0, st, a);
        prof_end(st, prec);
        TAN_LAUNCH_CHECK();
    }
    // ---- same-video blocks, column sums, positives, terms (+ the backward's corrections when its upstream gradients are known)
    if (do_finish) {
        FamFin f{};
        f.vn = (const bf16_t*)d->vn; f.tn = (const bf16_t*)d->tn; f.tn_stage_stride = a.t_stage_stride;
        f.colpart = w.colpart; f.nparts = 2 * a.npanel; f.diag = w.diag;
        f.tgt = d->tgt; f.col_invalid = d->col_invalid; f.row_leak = d->row_leak; f.colmap = d->colmap;
        f.rowsum = d->rowsum; f.colsum = d->colsum; f.possum_v = d->possum_v; f.possum_t = d->possum_t;
        f.v_terms = d->v_terms; f.t_terms = d->t_terms;
        f.S = S; f.B = B; f.T = T; f.N = N; f.Mc = Mc; f.R = R;
        f.log_cols = logf((float)Mc); f.log_rows = logf((float)R);
        if (d->g_v) {
            TAN_REQUIRE(d->d_tn_acc);
            f.g_v = d->g_v; f.g_t = d->g_t; f.corr = w.corr;
            f.zero = d->d_tn_acc; f.nzero = (long)St * Mc * 512;
        }
        const size_t lds = sizeof(float) * (2 * (size_t)T * N + T + 256 + 96) + 4 * 32 + 32 + (size_t)T + 16;
        TAN_REQUIRE(lds <= 160 * 1024);
        static std::atomic<unsigned long long> lds_done{0};
        const hipError_t attr = ensure_dyn_lds((const void*)simfam_finish_kernel, 160 * 1024, lds_done);
        if (attr != hipSuccess) return (int)attr;
        const int nfill = d->colmap ? (int)cdiv(Mc, 256) : 0;
        hipLaunchKernelGGL(simfam_finish_kernel, dim3(B + nfill, S), dim3(256), lds, st, f);
        TAN_LAUNCH_CHECK();
        if (d->g_v) d->flags |= TAN_SIMFAM_CORR_DONE;
    }
    return 0;
}

extern "C" int tan_simfam_bwd(tan_simfam_desc* d, void* stream) {
    int rc = simfam_check(d);
    if (rc) return rc;
    TAN_REQUIRE(d->g_v && d->g_t && d->dl && d->d_tn_acc && ((uintptr_t)d->dl % 16) == 0);
    hipStream_t st = (hipStream_t)stream;
    const int S = d->S, St = d->St, B = d->B, T = d->T, N = d->N, Mc = d->Mc;
    const long R = (long)B * T;
    for (int s = 0; s < S; ++s) TAN_REQUIRE(d->d_video.p[s]);
    for (int s = 0; s < St; ++s) TAN_REQUIRE(d->d_text.p[s]);
    const FamWs w = simfam_ws(d->ws, S, St, B, T, N, Mc);
    SimArgs a{};
    simfam_args(d, w, a);
    a.colsum = d->colsum; a.possum_t = d->possum_t; a.g_v = d->g_v; a.g_t = d->g_t; a.dl = (bf16_t*)d->dl;
    a.diag = w.diag; a.corr = w.corr;
    if (!(d->flags & TAN_SIMFAM_CORR_DONE)) {
        hipLaunchKernelGGL(simnce_corr_kernel, dim3(B, S), dim3(256), 0, st, (const float*)w.diag, a.tgt, a.col_invalid, a.row_leak,
                           (const float*)a.possum_v, a.possum_t, a.g_v, a.g_t, w.corr, B, T, N, a.colmap, Mc,
                           (d->flags & TAN_SIMFAM_ACC_ZEROED) ? nullptr : d->d_tn_acc, (long)St * Mc * 512);
        TAN_LAUNCH_CHECK();
    }
    // ---- d logits + d v_hat in one pass, the normalisation's backward as the epilogue -> the stack's stage-gradient rows
    {
        const int npanel = a.npanel, nct = cdiv(Mc, 128);
        FamL2 l2{};
        for (int s = 0; s < S; ++s) l2.dx.p[s] = (void*)d->d_video.p[s];
        l2.inv = d->inv_v; l2.grp = T; l2.dst_grp_rows = d->v_grp_rows; l2.dst_off = d->v_off;
        static std::atomic<unsigned long long> lds_done{0};
        const hipError_t attr = ensure_dyn_lds((const void*)simnce_dl_dvn_kernel, DV_LDS_B, lds_done);
        if (attr != hipSuccess) return (int)attr;
        const int prec = prof_begin(st, TAN_PROF_GEMM_BF16 + 1, 2.0 * S * R * (double)Mc * 512);
        hipLaunchKernelGGL(simnce_dl_dvn_kernel, dim3(npanel * S), dim3(512), DV_LDS_B, st, a, npanel, nct, (const char*)w.TpT,
                           St == 1 ? 0L : w.img_stage_stride, (bf16_t*)nullptr, l2);
        prof_end(st, prec);
        TAN_LAUNCH_CHECK();
    }
    // ---- d t_hat[st] (+)= dl[s]^T v_hat[s]  (f32, K slices meet in atomics; the accumulator was zeroed with the corrections)
    // Per-stage text features (the joint family): S problems [Mc x 512] under an R-long contraction on the 256 x 256-tile kernel
    // (tan_gemm_atb), two K slices -- 4.154 vs 4.185 ms per step against the 128 x 128-tile GEMM (ABBA x2 of 60 steps; three slices
    // 4.163; the shared-text problem of the dual family, one [Mc x 512] output under S*R rows, is equal or slower there with 16 / 21).
    bool done = false;
    const int atb_split = d->dtn_split_k < 0 ? -d->dtn_split_k : (d->dtn_split_k == 0 && St > 1 ? 2 : 0);
    const long Kc = St == 1 ? (long)S * R : R;
    if (atb_split > 0 && Kc % 128 == 0 && Kc / 128 >= atb_split && Mc % 8 == 0) {
        const void* A[8]; const void* Bm[8]; void* Cm[8]; int lda[8], Mv[8], Ns[8];
        const int np = St == 1 ? 1 : S;
        for (int p = 0; p < np; ++p) {
            A[p] = (const bf16_t*)d->dl + (long)p * R * Mc; Bm[p] = (const bf16_t*)d->vn + (long)p * R * 512;
            Cm[p] = d->d_tn_acc + (long)p * Mc * 512; lda[p] = Mc; Mv[p] = Mc; Ns[p] = 512;
        }
        rc = tan_gemm_atb(np, A, Bm, Cm, lda, Mv, Ns, Kc, TAN_F32, 1, atb_split, stream);
        if (rc == 0) done = true;
        else if (rc != TAN_ERR_BAD_ARG) return rc;
    }
    if (!done) {
        tan_gemm_desc g{};
        g.dtype = TAN_BF16; g.out_dtype = TAN_F32;
        g.M = Mc; g.N = 512; g.a_kc = 0; g.b_kc = 0;
        g.A = d->dl; g.lda = Mc; g.B = d->vn; g.ldb = 512; g.C = d->d_tn_acc; g.ldc = 512;
        g.accumulate = 1; g.alpha = 1.0f;
        if (St == 1) {
            g.K = (int)(S * R); g.batch = 1;
            g.split_k = d->dtn_split_k > 0 ? d->dtn_split_k : (int)((S * R / 512) < 1 ? 1 : ((S * R / 512) > 8 ? 8 : (S * R / 512)));
        } else {
            g.K = (int)R; g.batch = S; g.sA = R * Mc; g.sB = R * 512; g.sC = (long)Mc * 512;
            g.split_k = d->dtn_split_k > 0 ? d->dtn_split_k : 1;
        }
        if ((rc = tan_gemm(&g, stream))) return rc;
    }
    // ---- back to the padded sentence order, through the normalisation's backward, into the text rows of the stage gradients
    {
        FamTB t{};
        t.acc = d->d_tn_acc; t.tn = (const bf16_t*)d->tn; t.inv_t = d->inv_t; t.colmap = d->colmap;
        for (int s = 0; s < St; ++s) t.dx.p[s] = (void*)d->d_text.p[s];
        t.grp_rows = d->t_grp_rows; t.off = d->t_off; t.N = N; t.Mc = Mc; t.Mp = (long)B * N;
        hipLaunchKernelGGL(simfam_text_bwd_kernel, dim3(cdiv(t.Mp, 4), St), dim3(256), 0, st, t);
        TAN_LAUNCH_CHECK();
    }
    return 0;
}
