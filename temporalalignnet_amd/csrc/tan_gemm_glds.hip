// bf16 MFMA GEMM, direct-to-LDS edition (gfx950).  Fast path of tan_gemm for aligned bf16 problems whose contraction
// length is a multiple of 64; everything else stays on the register-staged kernel in tan_gemm.hip.
//
//   * 128x128 block tile, 256 threads = 4 waves (2x2), 64x64 per wave = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.
//   * operands stream HBM -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round trip, no ds_write pass),
//     two LDS buffers, ONE barrier per 64-deep K-step: the DMA of tile t+1 is in flight while tile t is multiplied.
//   * the LDS destination of a DMA is lane-linear (wave base + lane*16), so bank conflicts are removed by permuting the
//     per-lane SOURCE address and applying the same XOR when reading (cdna_hip_programming.md rule 21):
//       K-contiguous operand  image [128 rows][8 x 16-B slots]   slot = chunk ^ ((row >> 1) & 7), read with ds_read_b128
//       K-strided operand     image [64 k][16 x 16-B slots]      slot = chunk ^ ((k & 3) << 2),   read with
//                             ds_read_b64_tr_b16 (hardware 4x16 transpose: 4 consecutive k of one column per lane)
//     both read patterns are conflict-free (each 16/32-lane service group touches every bank once).
//   * rows / columns past the edge of the problem are CLAMPED to a valid address instead of masked: they only feed
//     accumulator rows / columns that the guarded epilogue never stores.
#include <stdlib.h>
#include <type_traits>

#include "tan_mma.h"
#include <cstdlib>

namespace tal {

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;   // one operand tile, either orientation: 16 KiB

struct GemmArgs2 {
    const bf16_t* A; const bf16_t* B; void* C;
    const float* bias; const void* residual; void* aux;
    long lda, ldb, ldc, ldr, ldaux;
    long sA, sB, sC;
    int M, N, K;
    int act, accumulate, split_k, kchunk, vec_epi;
    float alpha;
    float* colsum;
    int plane_xcd;
};

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// DMA one operand tile (16 KiB = 16 wave-instructions of 1 KiB; each of the 4 waves issues 4)
template <bool KC>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ P, long ld, int outer0, int OUT, int k0, char* lds_tile,
                                           int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;             // 1-KiB piece index, wave-uniform
        const bf16_t* src;
        if (KC) {
            const int row = piece * 8 + (lane >> 3), slot = lane & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            const int gr = min(outer0 + row, OUT - 1);
            src = P + (long)gr * ld + k0 + chunk * 8;
        } else {
            const int k = piece * 4 + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ ((k & 3) << 2);
            const int go = min(outer0 + chunk * 8, OUT - 8);
            src = P + (long)(k0 + k) * ld + go;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// A- or B-operand fragment (lane: outer index o0 + (lane & 31), k = ks + 8 * (lane >> 5) .. + 7)
template <bool KC>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int o0, int ks, int lane) {
    if (KC) {
        const int row = o0 + (lane & 31), chunk = (ks >> 3) + (lane >> 5);
        const int slot = chunk ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + slot * 16);
    } else {
        const int g = lane >> 4, p = lane & 15, r = p >> 2, q = p & 3;
        const int col = o0 + 16 * (g & 1) + 4 * q;           // first of the 4 columns this lane's 8 bytes cover
        const int k = ks + 8 * (g >> 1) + r;                  // k & 3 == r
        const int slot = (col >> 3) ^ (r << 2);
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_v4;
        const char* p0 = lds_tile + k * 256 + slot * 16 + (q & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0 + 1024));   // +4 k-rows; (k+4)&3 == r: same slot
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = lo; u.h[1] = hi;
        return u.v;
    }
}

template <typename TC, bool GUARD>
__device__ __forceinline__ void epilogue2(const GemmArgs2& g, f32x16 (&acc)[2][2], TC* C, const TC* R, TC* AUX, int m0, int n0,
                                          int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + acc_col(lane);
            if (GUARD && col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (GUARD && row >= g.M) continue;
                float v = acc[i][j][r] * g.alpha + bv;
                if (g.act == TAN_ACT_QUICKGELU) {
                    if (AUX) st_f(AUX + (long)row * g.ldaux + col, v);
                    v = quick_gelu_t<TC>(v);
                } else if (g.act == TAN_ACT_QUICKGELU_GRAD) {
                    v *= quick_gelu_grad_t<TC>(ld_f(AUX + (long)row * g.ldaux + col));
                } else if (g.act == TAN_ACT_RELU) {
                    v = fmaxf(v, 0.0f);
                }
                if (R) v += ld_f(R + (long)row * g.ldr + col);
                TC* cp = C + (long)row * g.ldc + col;
                if (g.accumulate) unsafeAtomicAdd((float*)cp, v);
                else st_f(cp, v);
            }
        }
}

// Vectorised epilogue: the 128x128 f32 tile is parked in the (now idle) 64 KiB of LDS, then every thread finishes 8 rows x
// 8 adjacent columns with 16-byte loads of aux / residual and 16-byte stores -- instead of 64 scattered 2-byte accesses
// per lane, each behind its own branch and wait.  Needs N % 8 == 0 and 16-byte aligned C / residual / aux rows.
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    uint4 u;
    u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]);
    u.z = f2bf2(v[4], v[5]); u.w = f2bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <typename TC>
__device__ __forceinline__ void epilogue_vec(const GemmArgs2& g, f32x16 (&acc)[2][2], float* tile, TC* C, const TC* R, TC* AUX,
                                             int m0, int n0, int wm, int wn, int lane, int tid) {
    // phase 1: accumulators (+ alpha, bias) -> LDS tile [128][128] f32
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lc = wn * 64 + j * 32 + acc_col(lane);
            const int gc = min(n0 + lc, g.N - 1);
            const float bv = g.bias ? g.bias[gc] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(wm * 64 + i * 32 + acc_row(r, lane)) * 128 + lc] = acc[i][j][r] * g.alpha + bv;
        }
    __syncthreads();
    // phase 2: 8 rows x 8 columns per thread, 16-byte global accesses.  Branch-free by construction: out-of-range rows /
    // columns are CLAMPED for the loads and predicated only at the stores, and the residual / pre-activation vectors of all
    // eight rows are loaded up front under ONE wave-uniform condition -- a per-row `if (R) load` made hipcc branch around
    // every load and wait `vmcnt(0)` behind it (and behind the previous row's store): eight serialised round trips per tile.
    const int cv = (tid & 15) * 8, col = n0 + cv;
    const bool col_ok = col < g.N;
    const int colc = col_ok ? col : g.N - 8;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    float rv[8][8], av[8][8];
    const bool has_r = R != nullptr, grad = g.act == TAN_ACT_QUICKGELU_GRAD;
    if (has_r) {
#pragma unroll
        for (int p = 0; p < 8; ++p) ld8(R + (long)min(m0 + (tid >> 4) + 16 * p, g.M - 1) * g.ldr + colc, rv[p]);
    }
    if (grad) {
#pragma unroll
        for (int p = 0; p < 8; ++p) ld8(AUX + (long)min(m0 + (tid >> 4) + 16 * p, g.M - 1) * g.ldaux + colc, av[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int lr = (tid >> 4) + 16 * p, row = m0 + lr;
        const bool ok = col_ok && row < g.M;
        float v[8];
        {
            const float4 a = *reinterpret_cast<const float4*>(tile + lr * 128 + cv);
            const float4 b = *reinterpret_cast<const float4*>(tile + lr * 128 + cv + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        if (g.act == TAN_ACT_QUICKGELU) {
            if (AUX && ok) st8(AUX + (long)row * g.ldaux + col, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = quick_gelu_t<TC>(v[e]);
        } else if (grad) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= quick_gelu_grad_t<TC>(av[p][e]);
        } else if (g.act == TAN_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        if (has_r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[p][e];
        }
        if (ok) {
            st8(C + (long)row * g.ldc + col, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += v[e];
        }
    }
    if (g.colsum) {      // fused bias gradient: column sums of this tile -> one atomic per column
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[(tid >> 4) * 128 + cv + e] = cs[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < g.N) {
            float s = 0.f;
#pragma unroll
            for (int y = 0; y < 16; ++y) s += tile[y * 128 + tid];
            unsafeAtomicAdd(g.colsum + n0 + tid, s);
        }
    }
}


// Work order.  The dispatcher deals consecutive workgroup ids (x fastest, then y, then z) round-robin to the 8 XCDs, each
// with a private L2.  Within one plane (one batch item / K-slice) id -> (id % 8) * ceil(n/8) + id / 8 (bijective form for any
// n) hands each XCD a CONTIGUOUS run of tiles, column tile fastest, so the tiles sharing an A row-panel hit one L2 instead of
// eight.  With several planes (batched GEMM, the dW K-slices) whole planes are pinned to XCDs instead -- 8 | planes: XCD x
// runs planes x, x+8, ..; planes | 8: 8/planes XCDs split a plane's tiles in contiguous runs -- so the operand slices of a
// plane are fetched into one or two L2s rather than all eight (dW fc: 120 MB -> 52 MB of fabric reads per launch).
__device__ __forceinline__ void work_item(const GemmArgs2& g, int& tile, int& z) {
    const int T = gridDim.x * gridDim.y, nz = gridDim.z;
    if (g.plane_xcd && nz > 1) {
        const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int xcd = id & 7, j = id >> 3;
        if ((nz & 7) == 0) { z = xcd + 8 * (j / T); tile = j % T; return; }
        if (8 % nz == 0 && T % (8 / nz) == 0) { z = xcd % nz; tile = (xcd / nz) * (T / (8 / nz)) + j; return; }
        // any other plane count (the six stages of the per-stage text-feature gradient): ONE contiguous run of (plane, tile) items per
        // XCD, tile fastest, so an XCD works on one or two planes instead of a slice of every plane -- with a run per plane each of
        // the 8 L2s pulled every plane's B operand (646 MB of reads per launch for 176 MB of operands, PMC)
        const int total = T * nz, q = total >> 3, r = total & 7;
        const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        z = item / T; tile = item - z * T;
        return;
    }
    int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = wg & 7, q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    z = blockIdx.z;
}


// ---- hand-issued transposing fragment reads -----------------------------------------------------------------------
// hipcc cannot tell the address of a `__builtin_amdgcn_ds_read_tr16_b64` from the destination of the LDS-DMA still in flight
// for the NEXT K-tile and drains it (`s_waitcnt vmcnt(0)`) before the first such read of every K-tile -- the two-buffer
// pipeline then overlaps nothing inside a workgroup (visible in the .s of every K-strided instantiation; the ds_read_b128
// instantiation has no such wait).  Where BOTH operands are K-strided (dW = dY^T X) the reads are therefore issued from inline
// asm, which the compiler neither waits for nor counts: form (ii) of the guide's section 5.7 -- "=v" destinations, then one
// `s_waitcnt lgkmcnt(0)` statement naming every destination "+v" before the first consumer -- software-pipelined by hand
// across the four 16-deep steps of a K-tile: [wait step s] [issue step s+1] [MFMAs of step s].
typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
struct TrFrag { tr_s16x4 lo, hi; };
struct TrStep { TrFrag a[2], b[2]; };

template <int OFF>
__device__ __forceinline__ void tr_issue(TrFrag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=v"(f.lo), "=v"(f.hi) : "v"(addr), "i"(OFF), "i"(OFF + 1024));
}
template <int OFF>    // OFF = byte offset of (buffer, k-step) inside the LDS array; the B tile follows the A tile
__device__ __forceinline__ void tr_issue_step(TrStep& s, const unsigned (&a_addr)[2], const unsigned (&b_addr)[2]) {
    tr_issue<OFF>(s.a[0], a_addr[0]);
    tr_issue<OFF>(s.a[1], a_addr[1]);
    tr_issue<OFF + TILE_BYTES>(s.b[0], b_addr[0]);
    tr_issue<OFF + TILE_BYTES>(s.b[1], b_addr[1]);
}
__device__ __forceinline__ void tr_wait(TrStep& s) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(s.a[0].lo), "+v"(s.a[0].hi), "+v"(s.a[1].lo), "+v"(s.a[1].hi), "+v"(s.b[0].lo), "+v"(s.b[0].hi),
                   "+v"(s.b[1].lo), "+v"(s.b[1].hi));
}
__device__ __forceinline__ bf16x8 tr_frag(const TrFrag& f) {
    union { bf16x8 v; tr_s16x4 h[2]; } u;
    u.h[0] = f.lo; u.h[1] = f.hi;
    return u.v;
}
__device__ __forceinline__ void tr_mma(f32x16 (&acc)[2][2], const TrStep& s) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(s.a[i]), tr_frag(s.b[j]), acc[i][j], 0, 0, 0);
}
// per-lane byte address (LDS offset) of a K-strided fragment at k-step 0 of buffer 0: the arithmetic of load_frag<false>
__device__ __forceinline__ unsigned tr_lane_addr(const char* lds_tile, int o0, int lane) {
    const int g = lane >> 4, p = lane & 15, r = p >> 2, q = p & 3;
    const int col = o0 + 16 * (g & 1) + 4 * q;
    const int k = 8 * (g >> 1) + r;
    const int slot = (col >> 3) ^ (r << 2);
    return (unsigned)(uintptr_t)(lds_tile + k * 256 + slot * 16 + (q & 1) * 8);
}
template <int CUR>
__device__ __forceinline__ void tr_tile(f32x16 (&acc)[2][2], const unsigned (&a_addr)[2], const unsigned (&b_addr)[2]) {
    constexpr int BASE = CUR * 2 * TILE_BYTES;       // k-step s starts 16 k-rows = 4096 bytes further
    TrStep s0, s1;
    tr_issue_step<BASE>(s0, a_addr, b_addr);
    tr_wait(s0); tr_issue_step<BASE + 4096>(s1, a_addr, b_addr); tr_mma(acc, s0);
    tr_wait(s1); tr_issue_step<BASE + 8192>(s0, a_addr, b_addr); tr_mma(acc, s1);
    tr_wait(s0); tr_issue_step<BASE + 12288>(s1, a_addr, b_addr); tr_mma(acc, s0);
    tr_wait(s1); tr_mma(acc, s1);
}

// K-contiguous A (compiler-scheduled ds_read_b128) x K-strided B (hand-issued transposing reads, same pipelining)
struct TrStepB { TrFrag b[2]; };
template <int OFF>
__device__ __forceinline__ void tr_issue_b(TrStepB& s, const unsigned (&b_addr)[2]) {
    tr_issue<OFF + TILE_BYTES>(s.b[0], b_addr[0]);
    tr_issue<OFF + TILE_BYTES>(s.b[1], b_addr[1]);
}
__device__ __forceinline__ void tr_wait_b(TrStepB& s) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s.b[0].lo), "+v"(s.b[0].hi), "+v"(s.b[1].lo), "+v"(s.b[1].hi));
}
template <int CUR, int KS>
__device__ __forceinline__ void kc_tr_mma(f32x16 (&acc)[2][2], const char* lds, const TrStepB& s, int wm, int lane) {
    bf16x8 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = load_frag<true>(lds + CUR * 2 * TILE_BYTES, wm * 64 + i * 32, KS, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], tr_frag(s.b[j]), acc[i][j], 0, 0, 0);
}
template <int CUR>
__device__ __forceinline__ void kc_tr_tile(f32x16 (&acc)[2][2], const char* lds, const unsigned (&b_addr)[2], int wm, int lane) {
    constexpr int BASE = CUR * 2 * TILE_BYTES;
    TrStepB s0, s1;
    tr_issue_b<BASE>(s0, b_addr);
    tr_wait_b(s0); tr_issue_b<BASE + 4096>(s1, b_addr); kc_tr_mma<CUR, 0>(acc, lds, s0, wm, lane);
    tr_wait_b(s1); tr_issue_b<BASE + 8192>(s0, b_addr); kc_tr_mma<CUR, 16>(acc, lds, s1, wm, lane);
    tr_wait_b(s0); tr_issue_b<BASE + 12288>(s1, b_addr); kc_tr_mma<CUR, 32>(acc, lds, s0, wm, lane);
    tr_wait_b(s1); kc_tr_mma<CUR, 48>(acc, lds, s1, wm, lane);
}

template <typename TC, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs2 g) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * TILE_BYTES];   // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = gridDim.x;
    int wg, z;
    work_item(g, wg, z);
    const int n0 = (wg % ntn) * GBN, m0 = (wg / ntn) * GBM;
    const int batch = z / g.split_k, split = z % g.split_k;
    const bf16_t* A = g.A + (long)batch * g.sA;
    const bf16_t* B = g.B + (long)batch * g.sB;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg) / GBK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);

    // per-lane LDS addresses of the hand-issued transposing reads (K-strided x K-strided instantiation only)
    const unsigned tr_a[2] = {tr_lane_addr(lds, wm * 64, lane), tr_lane_addr(lds, wm * 64 + 32, lane)};
    const unsigned tr_b[2] = {tr_lane_addr(lds, wn * 64, lane), tr_lane_addr(lds, wn * 64 + 32, lane)};

    // One K-step: start the DMA of tile t+1 into the OTHER buffer, multiply tile t, then wait + barrier.  The two buffers
    // are addressed with compile-time offsets (loop unrolled by two) so that the compiler can tell the DMA destination
    // from the fragment reads; with a runtime buffer index it drains the DMA (s_waitcnt vmcnt(0)) before the first ds_read.
    auto kstep = [&](int t, auto cur_c, auto nxt_c) {
        constexpr int CUR = decltype(cur_c)::value, NXT = decltype(nxt_c)::value;
        if (t + 1 < nt) {
            const int k0 = kbeg + (t + 1) * GBK;
            stage_tile<A_KC>(A, g.lda, m0, g.M, k0, lds + NXT * 2 * TILE_BYTES, wave, lane);
            stage_tile<B_KC>(B, g.ldb, n0, g.N, k0, lds + NXT * 2 * TILE_BYTES + TILE_BYTES, wave, lane);
        }
        if constexpr (!A_KC && !B_KC) {
            tr_tile<CUR>(acc, tr_a, tr_b);
        } else if constexpr (A_KC && !B_KC) {
            kc_tr_tile<CUR>(acc, lds, tr_b, wm, lane);
        } else {
#pragma unroll
            for (int ks = 0; ks < GBK; ks += 16) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = load_frag<A_KC>(lds + CUR * 2 * TILE_BYTES, wm * 64 + i * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = load_frag<B_KC>(lds + CUR * 2 * TILE_BYTES + TILE_BYTES, wn * 64 + j * 32, ks, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    if (nt > 0) {
        stage_tile<A_KC>(A, g.lda, m0, g.M, kbeg, lds, wave, lane);
        stage_tile<B_KC>(B, g.ldb, n0, g.N, kbeg, lds + TILE_BYTES, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nt; t += 2) {
        kstep(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if (t + 1 < nt) kstep(t + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    }

    TC* C = (TC*)g.C + (long)batch * g.sC;
    const TC* R = g.residual ? (const TC*)g.residual + (long)batch * g.sC : nullptr;
    TC* AUX = g.aux ? (TC*)g.aux + (long)batch * g.sC : nullptr;
    if (g.vec_epi) epilogue_vec<TC>(g, acc, reinterpret_cast<float*>(lds), C, R, AUX, m0, n0, wm, wn, lane, tid);
    else if (m0 + GBM <= g.M && n0 + GBN <= g.N) epilogue2<TC, false>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
    else epilogue2<TC, true>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// 4-stage variant: K-step 32, four 16-KiB LDS stages, THREE tiles in flight per workgroup (counted s_waitcnt vmcnt(8),
// raw s_barrier) instead of one 32-KiB tile.  Used where it measured faster (see use_four_stage).
constexpr int T4_BYTES = 128 * 32 * 2;    // 8 KiB operand tile

template <bool KC>
__device__ __forceinline__ void stage_tile4(const bf16_t* __restrict__ P, long ld, int outer0, int OUT, int k0, char* lds_tile,
                                            int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;             // 8 pieces of 1 KiB
        const bf16_t* src;
        if (KC) {                                   // [128 rows][4 x 16-B slots], slot = chunk ^ ((row >> 2) & 3)
            const int row = piece * 16 + (lane >> 2), slot = lane & 3;
            const int chunk = slot ^ ((row >> 2) & 3);
            const int gr = min(outer0 + row, OUT - 1);
            src = P + (long)gr * ld + k0 + chunk * 8;
        } else {                                    // [32 k][16 x 16-B slots], slot = chunk ^ ((k & 3) << 2)
            const int k = piece * 4 + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ ((k & 3) << 2);
            const int go = min(outer0 + chunk * 8, OUT - 8);
            src = P + (long)(k0 + k) * ld + go;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

template <bool KC>
__device__ __forceinline__ bf16x8 load_frag4(const char* lds_tile, int o0, int ks, int lane) {
    if (KC) {
        const int row = o0 + (lane & 31), chunk = (ks >> 3) + (lane >> 5);
        return *reinterpret_cast<const bf16x8*>(lds_tile + row * 64 + (chunk ^ ((row >> 2) & 3)) * 16);
    } else {
        return load_frag<false>(lds_tile, o0, ks, lane);      // same [k][256 B] image as the 64-deep tile
    }
}

// K-strided x K-strided stage of the 4-stage kernel: two 16-deep steps, reads issued by hand (see tr_tile)
template <int OFF>
__device__ __forceinline__ void tr_issue_step4(TrStep& s, const unsigned (&a_addr)[2], const unsigned (&b_addr)[2]) {
    tr_issue<OFF>(s.a[0], a_addr[0]);
    tr_issue<OFF>(s.a[1], a_addr[1]);
    tr_issue<OFF + T4_BYTES>(s.b[0], b_addr[0]);
    tr_issue<OFF + T4_BYTES>(s.b[1], b_addr[1]);
}
template <int CUR>
__device__ __forceinline__ void tr_stage4(f32x16 (&acc)[2][2], const unsigned (&a_addr)[2], const unsigned (&b_addr)[2]) {
    constexpr int BASE = CUR * 2 * T4_BYTES;
    TrStep s0, s1;
    tr_issue_step4<BASE>(s0, a_addr, b_addr);
    tr_wait(s0); tr_issue_step4<BASE + 4096>(s1, a_addr, b_addr); tr_mma(acc, s0);
    tr_wait(s1); tr_mma(acc, s1);
}

template <typename TC, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_glds4_kernel(GemmArgs2 g) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * T4_BYTES];   // [stage][A|B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = gridDim.x;
    int wg, z;
    work_item(g, wg, z);
    const int n0 = (wg % ntn) * GBN, m0 = (wg / ntn) * GBM;
    const int batch = z / g.split_k, split = z % g.split_k;
    const bf16_t* A = g.A + (long)batch * g.sA;
    const bf16_t* B = g.B + (long)batch * g.sB;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg) / 32;
    const long lda = g.lda, ldb = g.ldb;
    const int M = g.M, N = g.N;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);

    const unsigned tr_a[2] = {tr_lane_addr(lds, wm * 64, lane), tr_lane_addr(lds, wm * 64 + 32, lane)};
    const unsigned tr_b[2] = {tr_lane_addr(lds, wn * 64, lane), tr_lane_addr(lds, wn * 64 + 32, lane)};

#define G4_STAGE(T_, S_)                                                                                    \
    {                                                                                                       \
        stage_tile4<A_KC>(A, lda, m0, M, kbeg + (T_) * 32, lds + (S_) * 2 * T4_BYTES, wave, lane);          \
        stage_tile4<B_KC>(B, ldb, n0, N, kbeg + (T_) * 32, lds + (S_) * 2 * T4_BYTES + T4_BYTES, wave, lane); \
    }
    // every step issues exactly 4 DMA instructions per wave (a dummy re-load of the last tile past the end keeps the count
    // uniform), so "tile t+1 has landed" is always vmcnt(8): the two younger tiles may still be in flight
#define G4_STEP(T_, CUR, PRE)                                                                               \
    {                                                                                                       \
        const int t_ = (T_);                                                                                \
        G4_STAGE(min(t_ + 3, nt - 1), PRE)                                                                  \
        if constexpr (!A_KC && !B_KC) tr_stage4<CUR>(acc, tr_a, tr_b);                                      \
        else _Pragma("unroll") for (int ks = 0; ks < 32; ks += 16) {                                        \
            bf16x8 a[2], b[2];                                                                              \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) a[i] = load_frag4<A_KC>(lds + (CUR) * 2 * T4_BYTES, wm * 64 + i * 32, ks, lane); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
                b[j] = load_frag4<B_KC>(lds + (CUR) * 2 * T4_BYTES + T4_BYTES, wn * 64 + j * 32, ks, lane); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);        \
        }                                                                                                   \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                                       \
    }

    if (nt > 0) {
        G4_STAGE(0, 0)
        G4_STAGE(min(1, nt - 1), 1)
        G4_STAGE(min(2, nt - 1), 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nt; t += 4) {
            G4_STEP(t, 0, 3)
            if (t + 1 < nt) G4_STEP(t + 1, 1, 0)
            if (t + 2 < nt) G4_STEP(t + 2, 2, 1)
            if (t + 3 < nt) G4_STEP(t + 3, 3, 2)
        }
    }
#undef G4_STEP
#undef G4_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dummy tail DMAs must land before the LDS is reused
    __syncthreads();

    TC* C = (TC*)g.C + (long)batch * g.sC;
    const TC* R = g.residual ? (const TC*)g.residual + (long)batch * g.sC : nullptr;
    TC* AUX = g.aux ? (TC*)g.aux + (long)batch * g.sC : nullptr;
    if (g.vec_epi) epilogue_vec<TC>(g, acc, reinterpret_cast<float*>(lds), C, R, AUX, m0, n0, wm, wn, lane, tid);
    else if (m0 + GBM <= g.M && n0 + GBN <= g.N) epilogue2<TC, false>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
    else epilogue2<TC, true>(g, acc, C, R, AUX, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight-gradient GEMM: up to four problems gw_p[M_p,N_p] (partials) = dy_p[rows,M_p]^T x_p[rows,N_p] (both operands
// K-strided, the four Linear layers of one encoder block) in ONE launch of (sum of tiles) x (K slices) workgroups.  Launched one
// by one, each of the four is split 4-16 ways just to fill the chip and pays launch, first-tile latency and drain per slice
// (9 of the ~12 us of a short slice); together 192 tiles x 2 slices fill it with 64-80 K-steps each.
// Same pipeline as gemm_glds4_kernel<float, false, false> (three tiles in flight, hand-issued transposing reads).
struct GroupedDwArgs {
    const bf16_t* A[4]; const bf16_t* B[4]; float* C[4];
    int accumulate;                  // 1: C += (single K slice straight into the gradient), 0: C = (partial planes)
    int M[4], N[4];
    int tile_end[4];                 // running sum of tiles
    long plane[4];                   // M_p * N_p: stride between the K-slice partial planes of problem p
    int nprob, kchunk, K;
};

__global__ __launch_bounds__(256, 2) void gemm_dw_grouped_kernel(GroupedDwArgs ga) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * T4_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-contiguous tile order inside one K slice (see work_item); slices are blockIdx.y
    const int T = gridDim.x;
    int t;
    {
        const int id = blockIdx.x, xcd = id & 7, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) p += (i + 1 < ga.nprob && t >= ga.tile_end[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int t0 = p ? ga.tile_end[p - 1] : 0;
    const int M = ga.M[p], N = ga.N[p];
    const int ntn = (N + GBN - 1) / GBN;
    const int lt = t - t0, n0 = (lt % ntn) * GBN, m0 = (lt / ntn) * GBM;
    const int split = blockIdx.y;
    const int kbeg = split * ga.kchunk, kend = min(ga.K, kbeg + ga.kchunk);
    const int nt = (kend - kbeg) / 32;
    const long lda = M, ldb = N;
    const bf16_t* A = ga.A[p];
    const bf16_t* B = ga.B[p];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);
    const unsigned tr_a[2] = {tr_lane_addr(lds, wm * 64, lane), tr_lane_addr(lds, wm * 64 + 32, lane)};
    const unsigned tr_b[2] = {tr_lane_addr(lds, wn * 64, lane), tr_lane_addr(lds, wn * 64 + 32, lane)};

#define GD_STAGE(T_, S_)                                                                                     \
    {                                                                                                        \
        stage_tile4<false>(A, lda, m0, M, kbeg + (T_) * 32, lds + (S_) * 2 * T4_BYTES, wave, lane);          \
        stage_tile4<false>(B, ldb, n0, N, kbeg + (T_) * 32, lds + (S_) * 2 * T4_BYTES + T4_BYTES, wave, lane); \
    }
#define GD_STEP(T_, CUR, PRE)                                                                                \
    {                                                                                                        \
        const int t_ = (T_);                                                                                 \
        GD_STAGE(min(t_ + 3, nt - 1), PRE)                                                                   \
        tr_stage4<CUR>(acc, tr_a, tr_b);                                                                     \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                          \
        __builtin_amdgcn_s_barrier();                                                                        \
    }
    if (nt > 0) {
        GD_STAGE(0, 0)
        GD_STAGE(min(1, nt - 1), 1)
        GD_STAGE(min(2, nt - 1), 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int tt = 0; tt < nt; tt += 4) {
            GD_STEP(tt, 0, 3)
            if (tt + 1 < nt) GD_STEP(tt + 1, 1, 0)
            if (tt + 2 < nt) GD_STEP(tt + 2, 2, 1)
            if (tt + 3 < nt) GD_STEP(tt + 3, 3, 2)
        }
    }
#undef GD_STEP
#undef GD_STAGE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    GemmArgs2 g{};                       // what the shared epilogue reads
    g.M = M; g.N = N; g.ldc = N; g.ldr = N; g.alpha = 1.0f; g.act = TAN_ACT_NONE; g.vec_epi = 1;
    float* C = ga.C[p] + (long)split * ga.plane[p];
    epilogue_vec<float>(g, acc, reinterpret_cast<float*>(lds), C, ga.accumulate ? C : nullptr, nullptr, m0, n0, wm, wn, lane, tid);
}

// host side: returns -2 when the group is not eligible (caller runs the problems one by one)
int gemm_dw_grouped(int nprob, const void* const* dy, const void* const* x, float* const* parts, const int* Ms, const int* Ns,
                    long rows, int split, int accumulate, hipStream_t st) {
    if (nprob < 1 || nprob > 4 || split < 1 || (accumulate && split != 1) || rows % split != 0 || (rows / split) % 32 != 0) return -2;
    GroupedDwArgs ga{};
    int tiles = 0;
    for (int p = 0; p < nprob; ++p) {
        if (Ms[p] % 8 || Ns[p] % 8 || ((uintptr_t)dy[p] | (uintptr_t)x[p] | (uintptr_t)parts[p]) % 16) return -2;
        ga.A[p] = (const bf16_t*)dy[p]; ga.B[p] = (const bf16_t*)x[p]; ga.C[p] = parts[p];
        ga.M[p] = Ms[p]; ga.N[p] = Ns[p];
        tiles += cdiv(Ms[p], GBM) * cdiv(Ns[p], GBN);
        ga.tile_end[p] = tiles;
        ga.plane[p] = (long)Ms[p] * Ns[p];
    }
    for (int p = nprob; p < 4; ++p) { ga.tile_end[p] = tiles; ga.A[p] = ga.A[0]; ga.B[p] = ga.B[0]; ga.C[p] = ga.C[0]; ga.M[p] = ga.M[0]; ga.N[p] = ga.N[0]; }
    ga.nprob = nprob; ga.kchunk = (int)(rows / split); ga.K = (int)rows; ga.accumulate = accumulate;
    hipLaunchKernelGGL(gemm_dw_grouped_kernel, dim3(tiles, split), dim3(256), 0, st, ga);
    TAN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 block tiles for the grouped weight gradient: four waves, 128 x 128 per wave (16 accumulator tiles = 256 AGPRs),
// K cut in TWO slices (96 workgroups for the four Linear layers of a block) that add into the f32 gradient with atomics.
// Why: the weight-gradient kernels are bound by what a CU can pull through its vector-memory path -- the 128 x 128 kernel above
// and this one both move ~50 GB/s per CU with the LDS-DMA (measured with the MFMAs ablated: 162 us of DMA alone for 48
// workgroups x 8192 rows, 176 us with the MFMAs) -- so the lever is bytes per FLOP: a 256 x 256 tile stages 32 KiB per 32 K-rows
// for 32 MFMAs per wave, half of the 128 x 128 tile's bytes per FLOP (and half its LDS fragment bytes: 8 KiB per 16 MFMAs).
// Stand-alone the two kernels tie (86 us for 8192 rows); inside the training step this one runs the same 122 us on HALF the CUs
// (96 instead of 192), and the other stack's kernels speed up by what it leaves them: 5.46 -> 5.33 ms per step (ABBA x2).
// Three slices: 5.43; four (planes + folds or atomics): 5.45-5.54; one (48 workgroups): 5.78.
// LDS: four 32-KiB stages [A rows 0-127][A rows 128-255][B 0-127][B 128-255], each an 8-KiB [32 k][256 B] image exactly as
// stage_tile4<false> writes it; three stages in flight behind the one being read.  One barrier per stage, placed BETWEEN the two
// 16-deep steps: when a wave's reads of step 1 have landed it has finished with the stage, so behind that barrier the stage
// is free for the DMA of tile t+4 and tile t+1 (complete: counted vmcnt) may be read.
constexpr int D256_STAGE = 4 * T4_BYTES;
struct TrStep4 { TrFrag f[8]; };          // a[0..3] = f[0..3], b[0..3] = f[4..7]

// hipcc orders (pure) MFMA builtins freely around asm statements -- it sinks the 16 MFMAs of a step below the wait for the NEXT
// step's fragments and the barrier -- and it issues a stage's eight LDS-DMA loads and sixteen fragment reads in one burst.  With
// one wave per SIMD and the four waves in lock step behind a barrier that is fatal: the vector-memory front end takes ~16
// cycles per 1-KiB DMA instruction and a wave cannot reach its MFMAs before ITS loads are accepted, so all four waves queue
// there (~450 cycles of a 1024-cycle stage, measured by ablation), then all four multiply while the memory path idles.  Every
// instruction of the main loop is therefore volatile asm, executed in source order: one fragment read behind each of the first
// eight MFMAs of a step, one DMA load behind every fourth MFMA, accumulators pinned to AGPRs ("+a").
// The hazard recognizer does not look inside asm: `s_nop 1` covers "VALU wrote a source VGPR (the register allocator's copies of
// fragment halves) -> MFMA reads it" (2 wait states; it issues while the matrix pipe is still busy with the previous MFMA).
__device__ __forceinline__ void tr_wait256(TrStep4& s) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(s.f[0].lo), "+v"(s.f[0].hi), "+v"(s.f[1].lo), "+v"(s.f[1].hi), "+v"(s.f[2].lo), "+v"(s.f[2].hi),
                   "+v"(s.f[3].lo), "+v"(s.f[3].hi), "+v"(s.f[4].lo), "+v"(s.f[4].hi), "+v"(s.f[5].lo), "+v"(s.f[5].hi),
                   "+v"(s.f[6].lo), "+v"(s.f[6].hi), "+v"(s.f[7].lo), "+v"(s.f[7].hi));
}
__device__ __forceinline__ void mma256(f32x16& c, const TrFrag& a, const TrFrag& b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(tr_frag(a)), "v"(tr_frag(b)));
}
// one 1-KiB LDS-DMA load (M0 = destination, written and restored in the statement that uses it).  Scalar tile base + a per-lane
// 32-bit offset that never changes: a VALU pointer increment behind the load would have to wait until the queued load has read its
// address registers -- with the four waves' loads colliding at the vector-memory front end that stalled each wave ~50 cycles per load
__device__ __forceinline__ void dma256(unsigned voff, const char* base, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}

// C_p [M_p x N_p] (=|+=) A_p^T B_p for up to eight problems that share the contraction length K: A_p [K x lda_p] and B_p [K x N_p]
// row-major bf16 (the contraction runs over ROWS: both operands "K-strided").  The weight gradients of a block (tan_encoder_bwd); the
// general form is exported as tan_gemm_atb.  (Round 4 ran the two feature-gradient GEMMs of the similarity loss on it -- d_tn = dl^T vn,
// d_vn = (dl^T)^T tn with the transpose written by the d-logits pass: 810 vs 690 TF/s on the GEMMs themselves, but the transposed
// write made the d-logits pass 50 us longer and the K-sliced d_tn launches quantise badly on 256 CUs: 4.77 vs 4.65 ms per step,
// removed again.  DESIGN_APPENDIX.md A.9.)
// mv = valid output rows (M may be ragged: the last 256-row tile reads past the matrix' columns -- into the next row, the caller pads
// the end of the buffer -- and stores nothing there); out: 0 plain f32 store, 1 f32 read-add-write, 2 f32 atomics (K slices), 3 bf16
// store through the (then idle) LDS stages as whole rows.
struct Dw256Args {
    const bf16_t* A[8]; const bf16_t* B[8]; void* C[8];
    int M[8], Mv[8], N[8];           // M = row stride of A (elements) = columns of the tile grid's base; Mv <= M valid rows of C
    int tile_end[8];                 // running sum of tiles
    int nprob, kchunk, K, out;
};

__global__ __launch_bounds__(256) void gemm_dw256_kernel(Dw256Args ga) {
    constexpr int NST = 4;
    extern __shared__ __attribute__((aligned(1024))) char lds[];          // NST * D256_STAGE = 128 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> (tile, K slice).  ga.kchunk = S > 0 (S | 8): a 1-D grid, XCD x = id % 8 works slice x / (8/S) and a contiguous
    // run of tiles (column tile fastest), so the workgroups that share operand columns of the SAME K rows sit behind one L2 --
    // a CU pulls ~20 B/clk of L2 misses (what bounds this kernel) and several times that of L2 hits.
    int t, split, nsplit;
    if (ga.kchunk > 0) {
        nsplit = ga.kchunk;
        const int id = blockIdx.x, xcd = id & 7, per = 8 / nsplit, T = (int)gridDim.x / nsplit, run = T / per;
        split = xcd / per;
        t = (xcd % per) * run + (id >> 3);
    } else {
        const int T = gridDim.x;
        const int id = blockIdx.x, xcd = id & 7, q = T >> 3, r = T & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        split = blockIdx.y; nsplit = gridDim.y;
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) p += (i + 1 < ga.nprob && t >= ga.tile_end[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int t0 = p ? ga.tile_end[p - 1] : 0;
    const int M = ga.M[p], N = ga.N[p], Mv = ga.Mv[p];
    const int ntn = N / 256;
    const int lt = t - t0, n0 = (lt % ntn) * 256, m0 = (lt / ntn) * 256;
    // K slices in groups of 128 rows (four stages), the first K/128 % slices of them one group longer
    const int groups = ga.K / 128, gbase = groups / nsplit, grem = groups % nsplit;
    const int kbeg = 128 * (split * gbase + min(split, grem));
    const int nt = 4 * (gbase + (split < grem ? 1 : 0));

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_zero(acc[i][j]);
    // per-lane fragment addresses inside stage 0 / stage 2 (the 16-bit offset field of ds_read reaches two stages)
    unsigned f_lo[8], f_hi[8], f_top[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f_lo[i] = tr_lane_addr(lds + wm * T4_BYTES, i * 32, lane);
        f_lo[4 + i] = tr_lane_addr(lds + (2 + wn) * T4_BYTES, i * 32, lane);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { f_hi[i] = f_lo[i] + 2 * D256_STAGE; f_top[i] = f_lo[i] + 4 * D256_STAGE; }
    // the eight 1-KiB pieces this thread's wave loads per stage: u = 2 * (A rows 0-127 | A 128-255 | B 0-127 | B 128-255) + piece
    unsigned voff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int h = u >> 1, piece = wave * 2 + (u & 1);
        const int k = piece * 4 + (lane >> 4), slot = lane & 15, chunk = slot ^ ((k & 3) << 2);
        voff[u] = (unsigned)(k * (h < 2 ? M : N) + 128 * (h & 1) + chunk * 8) * 2u;
    }
    const char* baseA = (const char*)(ga.A[p] + (long)kbeg * M + m0);       // the K tile to load next
    const char* baseB = (const char*)(ga.B[p] + (long)kbeg * N + n0);
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)wave * 2048u;
    const long strideA = 64L * M, strideB = 64L * N;          // 32 k-rows, in bytes

#define D2_DST(S_, U_) (lds0 + (S_) * D256_STAGE + ((U_) >> 1) * T4_BYTES + ((U_) & 1) * 1024)
#define D2_READ(S_, STEP_, DST_, M_)                                                                           \
    {                                                                                                          \
        if ((S_) < 2)      tr_issue<((S_) & 1) * D256_STAGE + (STEP_) * 4096>(DST_.f[M_], f_lo[M_]);           \
        else if ((S_) < 4) tr_issue<((S_) & 1) * D256_STAGE + (STEP_) * 4096>(DST_.f[M_], f_hi[M_]);           \
        else               tr_issue<((S_) & 1) * D256_STAGE + (STEP_) * 4096>(DST_.f[M_], f_top[M_]);          \
    }
    // Stage CUR holds tile T_ (landed; s0 = its step-0 fragments, in flight or landed).  PRV / NXT = (CUR -+ 1) mod NST.
    // Step 0: 16 MFMAs on s0; behind them the step-1 fragments and the second half of tile T_+NST-1's loads (into stage PRV, free
    // since the previous barrier).  Then the one barrier of the stage: every wave has READ all of stage CUR (its step-1 fragments
    // have landed) and tile T_+1 is complete (own loads counted, vmcnt: tiles T_+2 .. T_+NST-1 may be in flight).  Step 1: 16 MFMAs
    // on s1; behind them the step-0 fragments of stage NXT and the first half of tile T_+NST's loads (into stage CUR).
#define D2_STAGE(T_, PRV, CUR, NXT)                                                                            \
    {                                                                                                          \
        tr_wait256(s0);                                                                                        \
        _Pragma("unroll") for (int m_ = 0; m_ < 16; ++m_) {                                                    \
            mma256(acc[m_ >> 2][m_ & 3], s0.f[m_ >> 2], s0.f[4 + (m_ & 3)]);                                   \
            if (m_ < 8) D2_READ(CUR, 1, s1, (m_ == 0 ? 0 : m_ < 5 ? m_ + 3 : m_ - 4))                          \
            if ((m_ & 3) == 3) dma256(voff[4 + (m_ >> 2)], baseB, D2_DST(PRV, 4 + (m_ >> 2)));                         \
        }                                                                                                      \
        baseB += ((T_) + NST < nt) ? strideB : 0L;                                                             \
        tr_wait256(s1);                                                                                        \
        if (NST == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                        \
        else          asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                        \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int m_ = 0; m_ < 16; ++m_) {                                                    \
            mma256(acc[m_ >> 2][m_ & 3], s1.f[m_ >> 2], s1.f[4 + (m_ & 3)]);                                   \
            if (m_ < 8) D2_READ(NXT, 0, s0, (m_ == 0 ? 0 : m_ < 5 ? m_ + 3 : m_ - 4))                          \
            if ((m_ & 3) == 3) dma256(voff[m_ >> 2], baseA, D2_DST(CUR, m_ >> 2));                                     \
        }                                                                                                      \
        baseA += ((T_) + NST + 1 < nt) ? strideA : 0L;                                                         \
    }
    TrStep4 s0, s1;
    // prologue: tiles 0 .. NST-2 and the first half of tile NST-1 (nt >= 4)
#pragma unroll
    for (int tl = 0; tl < NST - 1; ++tl) {
#pragma unroll
        for (int u = 0; u < 8; ++u) dma256(voff[u], u < 4 ? baseA : baseB, D2_DST(tl, u));
        baseA += (tl + 1 < nt) ? strideA : 0L;
        baseB += (tl + 1 < nt) ? strideB : 0L;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) dma256(voff[u], baseA, D2_DST(NST - 1, u));
    baseA += (NST < nt) ? strideA : 0L;
    if (NST == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else          asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) D2_READ(0, 0, s0, m)
    for (int tt = 0; tt < nt; tt += 4) {           // nt % 4 == 0
        D2_STAGE(tt, 3, 0, 1)
        D2_STAGE(tt + 1, 0, 1, 2)
        D2_STAGE(tt + 2, 1, 2, 3)
        D2_STAGE(tt + 3, 2, 3, 0)
    }
#undef D2_STAGE
#undef D2_READ
#undef D2_ADV
#undef D2_DST
    tr_wait256(s0);                      // the read-ahead of a tile past the end: landed, unused
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");     // + the last MFMA's result before the AGPR reads

    if (ga.out == 3) {
        // bf16 result: the wave's 128 x 128 quarter goes through LDS (rows of 272 B: off the bank period; 34 KiB per wave, the launch
        // asks for 136 KiB) and leaves as 16-byte vectors, four 256-byte row pieces per instruction
        __syncthreads();                 // every wave is done reading the stages
        constexpr int LDW = 136;         // bf16 per LDS row
        bf16_t* wt = reinterpret_cast<bf16_t*>(lds) + wave * (128 * LDW);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) wt[(i * 32 + acc_row(r, lane)) * LDW + j * 32 + acc_col(lane)] = f2bf(acc[i][j][r]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        bf16_t* C16 = reinterpret_cast<bf16_t*>(ga.C[p]);
#pragma unroll 4
        for (int q = lane; q < 128 * 16; q += 64) {
            const int lr = q >> 4, cc = (q & 15) * 8, grow = m0 + wm * 128 + lr;
            if (grow < Mv) *reinterpret_cast<uint4*>(C16 + (long)grow * N + n0 + wn * 128 + cc) = *reinterpret_cast<const uint4*>(wt + lr * LDW + cc);
        }
        return;
    }
    float* C = reinterpret_cast<float*>(ga.C[p]) + (long)(m0 + wm * 128) * N + n0 + wn * 128 + acc_col(lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (m0 + wm * 128 + i * 32 + acc_row(r, lane) >= Mv) continue;
            float* row = C + (long)(i * 32 + acc_row(r, lane)) * N;
            if (ga.out == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(row + j * 32, acc[i][j][r]);
            } else if (ga.out == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) row[j * 32] = acc[i][j][r];
            } else {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = row[j * 32];
#pragma unroll
                for (int j = 0; j < 4; ++j) row[j * 32] = v[j] + acc[i][j][r];
            }
        }
}

// host side; returns -2 when the group is not eligible.  lda[p] = row stride of A_p (>= Mv[p], multiple of 8), out as Dw256Args.out
int gemm_atb256(int nprob, const void* const* A, const void* const* B, void* const* C, const int* lda, const int* Mv, const int* Ns,
                long rows, int out, int split, hipStream_t st) {
    if (nprob < 1 || nprob > 8 || split < 1 || rows % 128 != 0 || rows / 128 < split) return -2;
    if ((out == 0 || out == 1 || out == 3) && split != 1) return -2;
    Dw256Args ga{};
    int tiles = 0;
    for (int p = 0; p < nprob; ++p) {
        if (lda[p] % 8 || Mv[p] < 1 || Mv[p] > lda[p] || Ns[p] % 256 || ((uintptr_t)A[p] | (uintptr_t)B[p] | (uintptr_t)C[p]) % 16) return -2;
        ga.A[p] = (const bf16_t*)A[p]; ga.B[p] = (const bf16_t*)B[p]; ga.C[p] = C[p];
        ga.M[p] = lda[p]; ga.Mv[p] = Mv[p]; ga.N[p] = Ns[p];
        tiles += (int)cdiv(Mv[p], 256) * (Ns[p] / 256);
        ga.tile_end[p] = tiles;
    }
    for (int p = nprob; p < 8; ++p) { ga.tile_end[p] = tiles; ga.A[p] = ga.A[0]; ga.B[p] = ga.B[0]; ga.C[p] = ga.C[0]; ga.M[p] = ga.M[0]; ga.Mv[p] = ga.Mv[0]; ga.N[p] = ga.N[0]; }
    ga.nprob = nprob; ga.K = (int)rows; ga.out = out;
    constexpr int LDS_BF16_EPI = 4 * 128 * 136 * 2;          // the bf16 epilogue parks four 128 x 128 quarters in rows of 136
    constexpr int LDS_MAX = LDS_BF16_EPI > 4 * D256_STAGE ? LDS_BF16_EPI : 4 * D256_STAGE;
    static std::atomic<unsigned long long> lds_done{0};
        const hipError_t attr = ensure_dyn_lds((const void*)gemm_dw256_kernel, LDS_MAX, lds_done);
    if (attr != hipSuccess) return (int)attr;
    const int lds_bytes = out == 3 ? LDS_MAX : 4 * D256_STAGE;
    ga.kchunk = (8 % split == 0 && tiles % (8 / split) == 0) ? split : 0;                 // slices pinned to XCD groups
    if (ga.kchunk) hipLaunchKernelGGL(gemm_dw256_kernel, dim3(tiles * split), dim3(256), lds_bytes, st, ga);
    else hipLaunchKernelGGL(gemm_dw256_kernel, dim3(tiles, split), dim3(256), lds_bytes, st, ga);
    TAN_LAUNCH_CHECK();
    return 0;
}

// gw_p += dy_p^T x_p (the weight gradients of a block): one K slice adds in place, more add with atomics
int gemm_dw256_grouped(int nprob, const void* const* dy, const void* const* x, float* const* gw, const int* Ms, const int* Ns,
                       long rows, int split, hipStream_t st) {
    if (nprob < 1 || nprob > 4) return -2;
    for (int p = 0; p < nprob; ++p) if (Ms[p] % 256) return -2;
    void* C[4];
    for (int p = 0; p < nprob; ++p) C[p] = gw[p];
    return gemm_atb256(nprob, dy, x, C, Ms, Ms, Ns, rows, split == 1 ? 1 : 2, split, st);
}

// The 4-stage / K-step-32 pipeline (three tiles in flight) pays for the K-strided x K-strided weight-gradient GEMMs once the K
// slice is long (dW c_fc 35.7 -> 33.8 us, c_proj 35.8 -> 33.3 us at 8192 rows).  For K-contiguous operands it won 6-12 % stand-alone
// on the N=512 outputs and LOST 2.4 % of the training step: that instantiation was removed in round 2.
static int use_four_stage(const tan_gemm_desc* d, const GemmArgs2& a) { return !d->a_kc && !d->b_kc && a.kchunk >= 1024; }

template <typename TC>
static int launch2(const tan_gemm_desc* d, const GemmArgs2& a, dim3 grid, hipStream_t st) {
    if (use_four_stage(d, a)) {
        hipLaunchKernelGGL((gemm_glds4_kernel<TC, false, false>), grid, dim3(256), 0, st, a);
        TAN_LAUNCH_CHECK();
        return 0;
    }
    if (d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm_glds_kernel<TC, true, true>), grid, dim3(256), 0, st, a);
    else if (d->a_kc && !d->b_kc) hipLaunchKernelGGL((gemm_glds_kernel<TC, true, false>), grid, dim3(256), 0, st, a);
    else if (!d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm_glds_kernel<TC, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_glds_kernel<TC, false, false>), grid, dim3(256), 0, st, a);
    TAN_LAUNCH_CHECK();
    return 0;
}

// returns -2 when the problem is not eligible (caller falls back to the register-staged kernel)
int gemm_glds_try(const tan_gemm_desc* d, hipStream_t st) {
    if (d->dtype != TAN_BF16) return -2;
    auto al = [](const void* p, long ld, long bs) { return ((uintptr_t)p % 16 == 0) && (ld % 8 == 0) && (bs % 8 == 0); };
    if (!al(d->A, d->lda, d->sA) || !al(d->B, d->ldb, d->sB)) return -2;
    if (d->K % GBK != 0) return -2;
    if (!d->a_kc && (d->M % 8 != 0 || d->M < 8)) return -2;
    if (!d->b_kc && (d->N % 8 != 0 || d->N < 8)) return -2;
    GemmArgs2 a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B; a.C = d->C; a.bias = d->bias; a.residual = d->residual; a.aux = d->aux;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr; a.ldaux = d->ldaux;
    a.sA = d->sA; a.sB = d->sB; a.sC = d->sC;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.act = d->act; a.accumulate = d->accumulate; a.split_k = d->split_k; a.alpha = d->alpha;
    a.kchunk = (int)(((long)cdiv(cdiv(d->K, d->split_k), GBK)) * GBK);
    const int oe = d->out_dtype == TAN_F32 ? 4 : 2;
    auto al16 = [&](const void* p, long ld) { return !p || (((uintptr_t)p % 16 == 0) && ((ld * oe) % 16 == 0)); };
    a.vec_epi = !d->accumulate && d->N % 8 == 0 && al16(d->C, d->ldc) && al16(d->residual, d->ldr) && al16(d->aux, d->ldaux) &&
                ((d->sC * oe) % 16 == 0);
    a.colsum = a.vec_epi ? d->colsum : nullptr;
    a.plane_xcd = 1;
    dim3 grid(cdiv(d->N, GBN), cdiv(d->M, GBM), d->batch * d->split_k);
    if (d->colsum && !a.vec_epi) {            // cannot fuse: run the GEMM, caller adds a separate column-sum pass
        int rc = d->out_dtype == TAN_F32 ? launch2<float>(d, a, grid, st) : launch2<bf16_t>(d, a, grid, st);
        return rc ? rc : -3;
    }
    if (d->out_dtype == TAN_F32) return launch2<float>(d, a, grid, st);
    return launch2<bf16_t>(d, a, grid, st);
}

}  // namespace tal

// C_p [M_p x N_p] (=|+=) A_p^T B_p, p < n <= 8: the 256 x 256-tile kernel above behind the C ABI (see include/tan_hip.h)
extern "C" int tan_gemm_atb(int n, const void* const* A, const void* const* B, void* const* C, const int* lda, const int* M, const int* N,
                            long K, int out_dtype, int accumulate, int split, void* stream) {
    TAN_REQUIRE(n >= 1 && n <= 8 && A && B && C && lda && M && N && K > 0 && split >= 1);
    int out;
    if (out_dtype == TAN_BF16) { TAN_REQUIRE(!accumulate && split == 1); out = 3; }
    else if (split > 1) { TAN_REQUIRE(accumulate); out = 2; }
    else out = accumulate ? 1 : 0;
    double work = 0;
    for (int p = 0; p < n; ++p) work += 2.0 * (double)K * M[p] * N[p];
    const int rec = tal::prof_begin((hipStream_t)stream, TAN_PROF_GEMM_BF16 + 3, work);
    const int rc = tal::gemm_atb256(n, A, B, C, lda, M, N, K, out, split, (hipStream_t)stream);
    tal::prof_end((hipStream_t)stream, rec);
    return rc == -2 ? TAN_ERR_BAD_ARG : rc;
}
