// Row-panel building blocks (gfx950): one workgroup (4 waves, one workgroup per CU) owns a 64-row panel of the residual stream and keeps
// it in LDS while the layer's weights stream past it.
//
//   * activations live in LDS as "panels" [64 rows][K] bf16, 16-byte chunk c of row r stored at chunk slot c ^ (r & 15):
//     a fragment read (ds_read_b128, lane = row, 16-lane service groups) touches 16 different slots of the 256-B bank row,
//     and so do the 8-lane service groups of the ds_write_b128 that fills a panel row by row.
//   * weights are PRE-PACKED once per optimizer step (tan_pack_weights) into the exact LDS image of the tiles the kernels
//     consume, tile after tile in consumption order: a tile is [TN rows = output features][TK contraction] bf16 = 16 KiB,
//     chunk slot = chunk ^ ((row / (16 / slots)) & (slots - 1)).  Streaming a tile is then a LINEAR 16-KiB copy with
//     global_load_lds_dwordx4 (1 KiB per wave-instruction, full cache lines, no address arithmetic), several tiles in flight
//     behind a counted s_waitcnt vmcnt(N) and a raw s_barrier.
//   * MFMA orientation is swapped: D[n][m] = sum_k W[n][k] X[m][k] (A operand = weight rows, B operand = activation rows),
//     so a lane's 16 accumulator registers hold ONE activation row and 16 output features; two v_permlane32_swap per register
//     pair turn them into 8 consecutive features = one 16-byte store (global row-major, or the next GEMM's LDS panel).
#pragma once
#include <type_traits>
#include <utility>

#include "tan_mma.h"

namespace tal {

typedef const void __attribute__((address_space(1)))* pn_gptr_t;
typedef void __attribute__((address_space(3)))* pn_lptr_t;
typedef const float __attribute__((address_space(4)))* pn_cfptr_t;     // constant address space: uniform loads become s_load

constexpr int PN_ROWS = 64;        // rows per panel
#ifndef TAN_PN_WAVES
#define TAN_PN_WAVES 8
#endif
constexpr int PN_WAVES = TAN_PN_WAVES;   // 4: one per SIMD (512 registers each); 8: two per SIMD (the hardware interleaves them)

template <int KD>
__device__ __forceinline__ int pn_tile_swz(int row) {
    constexpr int SL = KD / 8, RPB = 16 / SL;
    return (row / RPB) & (SL - 1);
}

// weight fragment (MFMA A operand): rows n0 + (lane & 31), k = ks + 8 * (lane >> 5) .. + 7 of a [rows][KD] tile image
template <int KD>
__device__ __forceinline__ bf16x8 pn_wfrag(const char* tile, int n0, int ks, int lane) {
    const int row = n0 + (lane & 31), chunk = (ks >> 3) + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(tile + row * (KD * 2) + ((chunk ^ pn_tile_swz<KD>(row)) << 4));
}

// activation fragment (MFMA B operand): rows mb * 32 + (lane & 31), k .. k + 7 of a panel with ROWB bytes per row
template <int ROWB>
__device__ __forceinline__ bf16x8 pn_pfrag(const char* panel, int mb, int k, int lane) {
    const int row = mb * 32 + (lane & 31), c = (k >> 3) + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(panel + row * ROWB + ((c ^ (row & 15)) << 4));
}

template <int ROWB>
__device__ __forceinline__ char* pn_panel_slot(char* panel, int row, int chunk) {
    return panel + row * ROWB + ((chunk ^ (row & 15)) << 4);
}

// stream one tile of PPW * 8 KiB: pieces of 1 KiB (one wave-instruction each), PPW per wave
template <int PPW>
__device__ __forceinline__ void pn_issue_tile(const char* src_tile, char* lds_stage, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave * PPW + i;
        __builtin_amdgcn_global_load_lds((pn_gptr_t)(src_tile + piece * 1024 + lane * 16), (pn_lptr_t)(lds_stage + piece * 1024), 16, 0, 0);
    }
}

// One 32x32 accumulator tile D[n][m] (lane: m = lane & 31, n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) -> two runs of 8 consecutive
// n per lane: o[p][e] is feature 8 * (2 * p + (lane >> 5)) + e of row m.
__device__ __forceinline__ void pn_rows_from_acc(const f32x16& acc, float (&o)[2][8]) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(acc[8 * p + x]), __float_as_int(acc[8 * p + 4 + x]), false, false);
            o[p][x] = __int_as_float(r[0]);
            o[p][4 + x] = __int_as_float(r[1]);
        }
}

__device__ __forceinline__ uint4 pn_pack8(const float (&v)[8]) {
    uint4 u;
    u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]);
    u.z = f2bf2(v[4], v[5]); u.w = f2bf2(v[6], v[7]);
    return u;
}
__device__ __forceinline__ void pn_unpack8(const uint4 u, float (&v)[8]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}

// 8 consecutive floats selected by the half-wave: lanes 0-31 read lo[0..7], lanes 32-63 hi[0..7]; both uniform (scalar loads)
__device__ __forceinline__ void pn_uniform8(pn_cfptr_t lo, pn_cfptr_t hi, int hiflag, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float a = lo[e], b = hi[e]; v[e] = hiflag ? b : a; }
}

// 16-byte LDS store the compiler does not see: an ordinary ds_write into the array that is also the destination of the LDS-DMA
// ring makes hipcc drain the whole ring first (s_waitcnt vmcnt(0): it cannot tell the panel from the ring).  The caller orders
// it with its own s_waitcnt lgkmcnt(0) + barrier before anybody reads the panel.
typedef unsigned pn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pn_lds_store16_hidden(const char* p, const uint4 v) {
    pn_u32x4 d;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)p), "v"(d) : "memory");
}

// Cooperative copy of a swizzled [64 rows][ROWB bytes] LDS panel to global rows (row stride ld elements): every wave-instruction
// stores WHOLE rows (1 KiB = two 512-byte rows or one 1-KiB row, lane-contiguous).  The accumulator layout would give 16 bytes per
// lane at a row stride instead: 64 partial-line writes per instruction, which cost L2 request slots like full lines do and -- VMEM
// returns being counted in order -- hold up every later weight-load wait of the wave.
template <int ROWB>
__device__ __forceinline__ void pn_panel_copy_out(const char* panel, bf16_t* dst, long ld, int wave, int lane) {
    constexpr int CH = ROWB / 16, RPI = 64 / CH, NI = PN_ROWS / RPI / PN_WAVES;
#pragma unroll 1
    for (int i = 0; i < NI; i += 2) {           // two rows-instructions at a time: called where the register file is full
        uint4 v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * NI + i + j) * RPI + lane / CH, chunk = lane % CH;
            v[j] = *reinterpret_cast<const uint4*>(panel + row * ROWB + ((chunk ^ (row & 15)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * NI + i + j) * RPI + lane / CH, chunk = lane % CH;
            *reinterpret_cast<uint4*>(dst + (long)row * ld + chunk * 8) = v[j];
        }
    }
}

template <int J, int END, typename F>
__device__ __forceinline__ void pn_static_for(F&& f) {
    if constexpr (J < END) {
        f(std::integral_constant<int, J>{});
        pn_static_for<J + 1, END>(f);
    }
}

__device__ __forceinline__ float pn_half_sum(float v) {     // v(lane) + v(lane ^ 32)
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}

}  // namespace tal
