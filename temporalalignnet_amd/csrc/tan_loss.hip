// Loss-side kernels for train/loss.py:get_loss (gfx950): the symmetric multi-positive NCE over materialised
// logits, the self-labelling scan ("softmax over time" + sliding-window argmax), agreement / de-duplication of the
// self-labelled targets and a masked quantile.  Logits are raw cosines [S, R = B*T, Mp = B*N] f32 (stage-major);
// the temperature 0.07 is applied here exactly as the reference does (a true division).
#include "tan_common.h"

// The self-labelling indices and threshold masks must match the reference bit for bit, so this file is compiled
// without cross-statement FMA contraction (hipcc's default -ffp-contract=fast fused `q*(n-1) - floor(q*(n-1))` into an
// exact fma and turned a zero interpolation weight into 3.6e-7, flipping a `>= quantile` comparison).
#pragma clang fp contract(off)

namespace tal {

constexpr float TAU = 0.07f;
constexpr float FILL = -6e4f;

// ------------------------------------------------------------------------------------------------------
// NCE "all" sums (loss.py:246-247,251-252): rowsum[s,r] = sum_{valid c} e, colsum[s,c] = sum_r e,
// e = exp(l/0.07 - 1/0.07)  (|l| <= 1 so e <= ~1: no running max needed; the shift is added back in nce_terms).
// Block = 64 rows of one stage; each wave owns 16 rows, lanes stride the columns; column partials accumulate in
// LDS (ds_add_f32) and leave as colpart[row_chunk][s][Mp].
__global__ __launch_bounds__(256) void nce_stats_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ col_invalid,
                                                        const unsigned char* __restrict__ row_leak, float* __restrict__ rowsum,
                                                        float* __restrict__ colpart, int R, int Mp, int T, int N) {
    extern __shared__ float cs[];
    const int s = blockIdx.y, rc = blockIdx.x, S = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < Mp; c += 256) cs[c] = 0.f;
    __syncthreads();
    const float shift = 1.0f / TAU;
    for (int i = 0; i < 16; ++i) {
        const int r = rc * 64 + wave * 16 + i;
        if (r >= R) break;
        const float* row = logits + ((long)s * R + r) * Mp;
        float acc = 0.f;
        // reference quirk (loss.py:96-101 in place on the online logits, model='init' + learn_agreement): same-video
        // entries of padded frames read -6e4, i.e. contribute nothing
        const int leak_b = (row_leak && row_leak[r]) ? r / T : -1;
        for (int c = lane; c < Mp; c += 64) {
            const float e = (c / N == leak_b) ? 0.f : expf(row[c] / TAU - shift);
            if (!col_invalid[c]) acc += e;
            atomicAdd(&cs[c], e);
        }
        acc = wave_sum(acc);
        if (lane == 0) rowsum[(long)s * R + r] = acc;
    }
    __syncthreads();
    float* out = colpart + ((long)rc * S + s) * Mp;
    for (int c = threadIdx.x; c < Mp; c += 256) out[c] = cs[c];
}

__global__ void nce_colsum_finalize(const float* __restrict__ colpart, float* __restrict__ colsum, int nchunk, long SM) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SM) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += colpart[(long)k * SM + i];
    colsum[i] = s;
}

// positives live on the same-video blocks only (loss.py:73-76): possum_v[s,(b,t)] = sum_k tgt[b,t,k] e,
// possum_t[s,(b,k)] = sum_t tgt[b,t,k] e.  One block per (video, stage).
__global__ __launch_bounds__(256) void nce_pos_kernel(const float* __restrict__ logits, const float* __restrict__ tgt,
                                                      const unsigned char* __restrict__ col_invalid,
                                                      const unsigned char* __restrict__ row_leak, float* __restrict__ possum_v,
                                                      float* __restrict__ possum_t, int B, int T, int N) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int R = B * T, Mp = B * N;
    const float shift = 1.0f / TAU;
    const float* blk = logits + ((long)s * R + (long)b * T) * Mp + (long)b * N;
    const float* tg = tgt + (long)b * T * N;
    for (int t = threadIdx.x; t < T; t += 256) {
        float acc = 0.f;
        for (int k = 0; k < N; ++k)
            if (tg[t * N + k] != 0.f && !col_invalid[b * N + k] && !(row_leak && row_leak[b * T + t]))
                acc += expf(blk[(long)t * Mp + k] / TAU - shift);
        possum_v[(long)s * R + b * T + t] = acc;
    }
    for (int k = threadIdx.x; k < N; k += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t)
            if (tg[t * N + k] != 0.f && !(row_leak && row_leak[b * T + t])) acc += expf(blk[(long)t * Mp + k] / TAU - shift);
        possum_t[(long)s * Mp + b * N + k] = acc;
    }
}

// v_terms[s,r] = LSE_all - LSE_pos (loss.py:246-248), t_terms[s,c] likewise (loss.py:250-253).  LSE over an empty positive
// set reproduces the reference's -6e4 fill: log(sum_valid exp(-6e4)) = -6e4 + log(#cols).
__global__ void nce_terms_kernel(const float* __restrict__ allsum, const float* __restrict__ possum, float* __restrict__ terms,
                                 long n, float log_count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float shift = 1.0f / TAU;
    const float den = logf(allsum[i]) + shift;
    const float num = possum[i] > 0.f ? logf(possum[i]) + shift : FILL + log_count;
    terms[i] = den - num;
}

// d logits (loss.py:240-275 backward): dl = (1/0.07) * [ gv[s,r] (e/rowsum - pos e/possum_v) + gt[s,c] (e/colsum - pos e/possum_t) ]
template <typename TO>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ tgt,
                                                      const unsigned char* __restrict__ col_invalid,
                                                      const unsigned char* __restrict__ row_leak,
                                                      const float* __restrict__ rowsum, const float* __restrict__ colsum,
                                                      const float* __restrict__ possum_v, const float* __restrict__ possum_t,
                                                      const float* __restrict__ gv, const float* __restrict__ gt,
                                                      TO* __restrict__ dl, int S, int B, int T, int N) {
    const int R = B * T, Mp = B * N;
    const long total = (long)S * R * Mp;
    const float shift = 1.0f / TAU, inv_tau = 1.0f / TAU;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Mp);
        const long sr = i / Mp;
        const int r = (int)(sr % R), s = (int)(sr / R);
        float g = 0.f;
        const int b = r / T, bc = c / N;
        const float e = (b == bc && row_leak && row_leak[r]) ? 0.f : expf(logits[i] / TAU - shift);
        const float gvr = gv[sr], gtc = gt[(long)s * Mp + c];
        if (!col_invalid[c]) g += gvr * e / rowsum[sr];
        g += gtc * e / colsum[(long)s * Mp + c];
        if (b == bc && tgt[((long)b * T + (r - b * T)) * N + (c - bc * N)] != 0.f) {
            if (!col_invalid[c] && possum_v[sr] > 0.f) g -= gvr * e / possum_v[sr];
            if (possum_t[(long)s * Mp + c] > 0.f) g -= gtc * e / possum_t[(long)s * Mp + c];
        }
        st_f(dl + i, g * inv_tau);
    }
}

// ------------------------------------------------------------------------------------------------------
// Self-labelling scan (loss.py:88-143 / 146-179), one block per video:
//   z[t,n]  = logits[S-1, (b,t), (b,n)] / 0.07, -6e4 where frame t or text n is padding          (loss.py:91-101)
//   p1      = softmax_n z ; prob = softmax_t (p1 / 0.07)                                           (loss.py:104)
//   window i of text n covers [i, i+dur_n) if it fits in [0,T), minus frames 0 and T-1, uniform     (loss.py:112-131)
//   scan[i] = mean of prob over the window ; max_pos = first argmax_i ; max_logit = window mean of z (loss.py:133-141)
__global__ __launch_bounds__(256) void selflabel_kernel(const float* __restrict__ blocks, long sb, long st, const unsigned char* __restrict__ vpad,
                                                        const unsigned char* __restrict__ tpad, const float* __restrict__ dur,
                                                        int* __restrict__ max_pos, float* __restrict__ max_prob,
                                                        float* __restrict__ max_logit, unsigned char* __restrict__ self_tgt,
                                                        int B, int T, int N) {
    extern __shared__ float sm[];
    float* z = sm;               // [T][N]
    float* p = z + T * N;        // [T][N]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* blk = blocks + (long)b * sb;            // same-video block of the last stage: element (t, n) at blk[t*st + n]
    for (int i = threadIdx.x; i < T * N; i += 256) {
        const int t = i / N, n = i % N;
        float v = blk[(long)t * st + n] / TAU;
        if (vpad && vpad[b * T + t]) v = FILL;
        if (tpad[b * N + n]) v = FILL;
        z[i] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {  // softmax over texts
        float m = -INFINITY;
        for (int n = 0; n < N; ++n) m = fmaxf(m, z[t * N + n]);
        float s = 0.f;
        for (int n = 0; n < N; ++n) { const float e = expf(z[t * N + n] - m); p[t * N + n] = e; s += e; }
        for (int n = 0; n < N; ++n) p[t * N + n] = (p[t * N + n] / s) / TAU;
    }
    __syncthreads();
    for (int n = wave; n < N; n += 4) {  // softmax over time, one wave per text
        float m = -INFINITY;
        for (int t = lane; t < T; t += 64) m = fmaxf(m, p[t * N + n]);
        m = wave_max(m);
        float s = 0.f;
        for (int t = lane; t < T; t += 64) { const float e = expf(p[t * N + n] - m); p[t * N + n] = e; s += e; }
        s = wave_sum(s);
        for (int t = lane; t < T; t += 64) p[t * N + n] = p[t * N + n] / s;
    }
    __syncthreads();
    for (int n = wave; n < N; n += 4) {
        const int d = (int)dur[b * N + n];
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int i = lane; i < T; i += 64) {
            float v = 0.f;
            if (d > 0 && i + d <= T) {
                const int lo = max(i, 1), hi = min(i + d, T - 1);  // members j in [lo, hi)
                const int cnt = hi - lo;
                if (cnt > 0) {
                    const float w = 1.0f / (float)cnt;
                    for (int j = lo; j < hi; ++j) v += p[j * N + n] * w;
                }
            }
            if (v > best) { best = v; best_i = i; }   // lane visits i in increasing order: keeps its first max
        }
        // wave arg-max, ties to the smallest index (torch.max returns the first maximal index)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        }
        const int i = best_i;
        int lo = 0, hi = 0;
        if (d > 0 && i + d <= T) { lo = max(i, 1); hi = min(i + d, T - 1); }
        const int cnt = max(hi - lo, 0);
        float ml = 0.f;
        if (cnt > 0) {
            const float w = 1.0f / (float)cnt;
            for (int j = lo + lane; j < hi; j += 64) ml += z[j * N + n] * w;
        }
        ml = wave_sum(ml);
        for (int t = lane; t < T; t += 64) self_tgt[((long)b * N + n) * T + t] = (t >= lo && t < hi) ? 1 : 0;
        if (lane == 0) {
            max_pos[b * N + n] = i;
            max_prob[b * N + n] = best;
            max_logit[b * N + n] = ml;
        }
    }
}

// per-text max over time of the last-stage same-video logits / 0.07 (loss.py:280,283)
__global__ void diag_max_kernel(const float* __restrict__ blocks, long sb, long st, const unsigned char* __restrict__ row_leak,
                                float* __restrict__ out, int B, int T, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i % N;
    const float* col = blocks + (long)b * sb + n;
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, (row_leak && row_leak[b * T + t]) ? FILL : col[(long)t * st] / TAU);
    out[i] = m;
}

// Agreement of the two self-labelled windows + exclusion principle (loss.py:181-226), one block per video.
//   kind 0 'i', 1 'u', 2 'keep', 3 'keep-joint'.  Writes tgt_out [B,T,N] f32, iou [B,N], conf [B,N].
__global__ __launch_bounds__(256) void agreement_kernel(const unsigned char* __restrict__ jt, const unsigned char* __restrict__ dt,
                                                        const unsigned char* __restrict__ yt, const float* __restrict__ ml_j,
                                                        const float* __restrict__ ml_d, const float* __restrict__ q_j,
                                                        const float* __restrict__ q_d, int kind, float* __restrict__ tgt_out,
                                                        float* __restrict__ iou_out, unsigned char* __restrict__ conf_out,
                                                        int B, int T, int N) {
    extern __shared__ unsigned char smb[];
    unsigned char* agree = smb;          // [N][T]
    unsigned char* dedup = agree + N * T;  // [N][T]
    __shared__ int conf_iou_s[64], lost_s[64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = wave; n < N; n += 4) {
        const long base = ((long)b * N + n) * T;
        int inter = 0, uni = 0;
        for (int t = lane; t < T; t += 64) {
            inter += (jt[base + t] & dt[base + t]) ? 1 : 0;
            uni += (jt[base + t] | dt[base + t]) ? 1 : 0;
        }
        inter = (int)wave_sum((float)inter);
        uni = (int)wave_sum((float)uni);
        const float iou = (float)inter / fmaxf((float)uni, 1e-5f);
        const bool c_iou = iou >= 0.5f;
        const bool conf = c_iou && (ml_d[b * N + n] >= q_d[0]) && (ml_j[b * N + n] >= q_j[0]);
        if (lane == 0) {
            iou_out[b * N + n] = iou;
            conf_out[b * N + n] = conf ? 1 : 0;
            conf_iou_s[n] = c_iou;
        }
        for (int t = lane; t < T; t += 64) {
            const unsigned char j = jt[base + t], d = dt[base + t], y = yt[base + t];
            unsigned char a;
            if (kind == 0) a = conf ? (j & d) : 0;
            else if (kind == 1) a = conf ? (j | d) : 0;
            else if (kind == 2) a = c_iou ? (j | d) : y;
            else a = c_iou ? j : y;
            agree[n * T + t] = a ? 1 : 0;
            dedup[n * T + t] = 0;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {  // first text at every frame (argmax over n; 0 when none)
        int first = 0;
        for (int n = 0; n < N; ++n) if (agree[n * T + t]) { first = n; break; }
        dedup[first * T + t] = 1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) dedup[t] = agree[t];  // text 0 keeps its own row
    __syncthreads();
    for (int n = wave; n < N; n += 4) {
        int any = 0;
        for (int t = lane; t < T; t += 64) any += dedup[n * T + t];
        any = (int)wave_sum((float)any);
        if (lane == 0) lost_s[n] = (any == 0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * T; i += 256) {
        const int n = i / T, t = i % T;
        const unsigned char v = lost_s[n] ? yt[((long)b * N + n) * T + t] : dedup[i];
        tgt_out[((long)b * T + t) * N + n] = v ? 1.0f : 0.0f;
    }
}

// quantile (torch.quantile, 'linear') of the valid entries of x[n]; single block, bitonic sort in LDS.
__global__ __launch_bounds__(1024) void masked_quantile_kernel(const float* __restrict__ x, const unsigned char* __restrict__ invalid,
                                                               int n, float q, float* __restrict__ out, int npow2) {
    extern __shared__ float v[];
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        float val = INFINITY;
        if (i < n && !(invalid && invalid[i])) { val = x[i]; atomicAdd(&cnt, 1); }
        v[i] = val;
    }
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const float a = v[i], b2 = v[ixj];
                    if ((a > b2) == up) { v[i] = b2; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        const int m = cnt;
        if (m == 0) { out[0] = NAN; return; }
        const float rank = q * (float)(m - 1);
        const float lo = floorf(rank);
        const int il = (int)lo, ih = min(il + 1, m - 1);
        const float w = rank - lo, a = v[il], b2 = v[ih];
        out[0] = (w < 0.5f) ? a + w * (b2 - a) : b2 - (b2 - a) * (1.0f - w);  // at::lerp
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// NCE tail (loss.py:236-237,254-275): which rows / columns own a positive, and the four masked means.
// pos_masks: rows_pos[b*T+t] = any_k(tgt[b,t,k] != 0 && !tpad[b,k]); cols_pos[b*N+k] = any_t(tgt[b,t,k] != 0) && !tpad[b,k].
__global__ __launch_bounds__(256) void pos_masks_kernel(const float* __restrict__ tgt, const unsigned char* __restrict__ tpad,
                                                        float* __restrict__ rows_pos, float* __restrict__ cols_pos, int T, int N) {
    const int b = blockIdx.x;
    const float* tg = tgt + (long)b * T * N;
    const unsigned char* tp = tpad + (long)b * N;
    for (int t = threadIdx.x; t < T; t += 256) {
        bool any = false;
        for (int k = 0; k < N; ++k) any |= (tg[t * N + k] != 0.f) && !tp[k];
        rows_pos[(long)b * T + t] = any ? 1.f : 0.f;
    }
    for (int k = threadIdx.x; k < N; k += 256) {
        bool any = false;
        for (int t = 0; t < T; ++t) any |= tg[t * N + k] != 0.f;
        cols_pos[(long)b * N + k] = (any && !tp[k]) ? 1.f : 0.f;
    }
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    return s;
}

// out[0] = (mean(v_d | rmask) + mean(t_d | cmask)) / 2, out[1] likewise for the joint terms; mean(x | m) = sum_{s,k} x[s,k] m[k]
// / (S sum_k m[k])  (0/0 = NaN like an empty .mean()).  counts[0..1] = sum(rmask), sum(cmask), kept for the backward.
__global__ __launch_bounds__(1024) void nce_tail_fwd_kernel(const float* __restrict__ v_d, const float* __restrict__ t_d,
                                                            const float* __restrict__ v_j, const float* __restrict__ t_j,
                                                            const float* __restrict__ rmask, const float* __restrict__ cmask, int Sd,
                                                            int Sj, long R, long M, float* __restrict__ out,
                                                            float* __restrict__ counts, const float* __restrict__ counts_in) {
    __shared__ float red[16];
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // sum v_d, t_d, v_j, t_j, n_r, n_c
    // one block (deterministic sums); eight rows per thread with their loads issued together -- row by row, each of the
    // Sd + Sj dependent-looking loads paid a full memory latency (36 us for 8192 rows x 12 stages on the loss's critical path)
    constexpr int U = 8;
    auto masked_sums = [&](const float* __restrict__ xd, const float* __restrict__ xj, const float* __restrict__ mask, long n, float& sd,
                           float& sj, float& cnt) {
        for (long r0 = threadIdx.x; r0 < n; r0 += 1024 * U) {
            float m[U];
            long rr[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long r = r0 + (long)k * 1024;
                rr[k] = r < n ? r : n - 1;
                m[k] = r < n ? mask[rr[k]] : 0.f;
                cnt += m[k];
            }
#pragma unroll 6
            for (int st = 0; st < Sd; ++st) {          // (unrolled by the usual stage count: 48 loads in flight per thread)
#pragma unroll
                for (int k = 0; k < U; ++k) sd += xd[(long)st * n + rr[k]] * m[k];
            }
#pragma unroll 6
            for (int st = 0; st < Sj; ++st) {
#pragma unroll
                for (int k = 0; k < U; ++k) sj += xj[(long)st * n + rr[k]] * m[k];
            }
        }
    };
    masked_sums(v_d, v_j, rmask, R, a[0], a[2], a[4]);
    masked_sums(t_d, t_j, cmask, M, a[1], a[3], a[5]);
    float tot[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) tot[i] = block_sum_1024(a[i], red);
    if (threadIdx.x == 0) {
        const float nr = counts_in ? counts_in[0] : tot[4], nc = counts_in ? counts_in[1] : tot[5];    // global counts (row f3)
        out[0] = 0.5f * (tot[0] / (Sd * nr) + tot[1] / (Sd * nc));
        out[1] = 0.5f * (tot[2] / (Sj * nr) + tot[3] / (Sj * nc));
        out[2] = (out[0] + out[1]) / 2.0f;                       // loss.py:352 (the default total), one launch less each way
        counts[0] = nr; counts[1] = nc;
    }
}

__global__ void nce_tail_bwd_kernel(const float* __restrict__ g_d, const float* __restrict__ g_j, const float* __restrict__ g_m,
                                    const float* __restrict__ rmask, const float* __restrict__ cmask,
                                    const float* __restrict__ counts, int Sd, int Sj, long R, long M, float* __restrict__ g_v_d,
                                    float* __restrict__ g_t_d, float* __restrict__ g_v_j, float* __restrict__ g_t_j) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float gm = g_m ? 0.5f * g_m[0] : 0.f;
    const float gd = 0.5f * ((g_d ? g_d[0] : 0.f) + gm), gj = 0.5f * ((g_j ? g_j[0] : 0.f) + gm);
    if (i < R) {
        const float m = rmask[i] / counts[0];
        for (int s = 0; s < Sd; ++s) g_v_d[(long)s * R + i] = gd * m / Sd;
        for (int s = 0; s < Sj; ++s) g_v_j[(long)s * R + i] = gj * m / Sj;
    }
    if (i < M) {
        const float m = cmask[i] / counts[1];
        for (int s = 0; s < Sd; ++s) g_t_d[(long)s * M + i] = gd * m / Sd;
        for (int s = 0; s < Sj; ++s) g_t_j[(long)s * M + i] = gj * m / Sj;
    }
}

}  // namespace tal

using namespace tal;

extern "C" long tan_nce_ws_floats(int S, int B, int T, int N) {
    const long R = (long)B * T, Mp = (long)B * N;
    return (long)cdiv(R, 64) * S * Mp;
}

extern "C" int tan_nce_fwd(const float* logits, const float* tgt, const unsigned char* col_invalid, const unsigned char* row_leak,
                           float* rowsum, float* colsum,
                           float* possum_v, float* possum_t, float* v_terms, float* t_terms, float* ws, int S, int B, int T,
                           int N, int n_valid_cols, void* stream) {
    TAN_REQUIRE(logits && tgt && col_invalid && rowsum && colsum && possum_v && possum_t && v_terms && t_terms && ws);
    TAN_REQUIRE(S > 0 && B > 0 && T > 0 && N > 0);
    const int R = B * T, Mp = B * N;
    TAN_REQUIRE((size_t)Mp * 4 <= 64 * 1024);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = cdiv(R, 64);
    hipLaunchKernelGGL(nce_stats_kernel, dim3(nchunk, S), dim3(256), (size_t)Mp * 4, st, logits, col_invalid, row_leak, rowsum, ws, R, Mp, T, N);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_colsum_finalize, dim3(cdiv((long)S * Mp, 256)), dim3(256), 0, st, ws, colsum, nchunk, (long)S * Mp);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_pos_kernel, dim3(B, S), dim3(256), 0, st, logits, tgt, col_invalid, row_leak, possum_v, possum_t, B, T, N);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_terms_kernel, dim3(cdiv((long)S * R, 256)), dim3(256), 0, st, rowsum, possum_v, v_terms, (long)S * R,
                       logf((float)n_valid_cols));
    hipLaunchKernelGGL(nce_terms_kernel, dim3(cdiv((long)S * Mp, 256)), dim3(256), 0, st, colsum, possum_t, t_terms,
                       (long)S * Mp, logf((float)R));
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_nce_bwd(const float* logits, const float* tgt, const unsigned char* col_invalid, const unsigned char* row_leak,
                           const float* rowsum,
                           const float* colsum, const float* possum_v, const float* possum_t, const float* g_v, const float* g_t,
                           void* dlogits, int out_dtype, int S, int B, int T, int N, void* stream) {
    TAN_REQUIRE(logits && tgt && col_invalid && rowsum && colsum && possum_v && possum_t && g_v && g_t && dlogits);
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)S * B * T * B * N;
    const unsigned grid = (unsigned)min((long)8192, (long)cdiv(total, 256));
    if (out_dtype == TAN_F32)
        hipLaunchKernelGGL((nce_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, logits, tgt, col_invalid, row_leak, rowsum, colsum, possum_v,
                           possum_t, g_v, g_t, (float*)dlogits, S, B, T, N);
    else if (out_dtype == TAN_BF16)
        hipLaunchKernelGGL((nce_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, logits, tgt, col_invalid, row_leak, rowsum, colsum,
                           possum_v, possum_t, g_v, g_t, (bf16_t*)dlogits, S, B, T, N);
    else return TAN_ERR_BAD_ARG;
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_selflabel(const float* blocks, long block_stride, long row_stride, const unsigned char* video_pad,
                             const unsigned char* text_pad, const float* dur, int* max_pos, float* max_prob, float* max_logit,
                             unsigned char* self_tgt, int B, int T, int N, void* stream) {
    TAN_REQUIRE(blocks && text_pad && dur && max_pos && max_prob && max_logit && self_tgt && B > 0 && T > 0 && N > 0);
    const size_t sm = (size_t)2 * T * N * 4;
    TAN_REQUIRE(sm <= 64 * 1024);
    hipLaunchKernelGGL(selflabel_kernel, dim3(B), dim3(256), sm, (hipStream_t)stream, blocks, block_stride, row_stride, video_pad, text_pad, dur,
                       max_pos, max_prob, max_logit, self_tgt, B, T, N);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_diag_max(const float* blocks, long block_stride, long row_stride, const unsigned char* row_leak, float* out, int B,
                            int T, int N, void* stream) {
    TAN_REQUIRE(blocks && out && B > 0 && T > 0 && N > 0);
    hipLaunchKernelGGL(diag_max_kernel, dim3(cdiv((long)B * N, 128)), dim3(128), 0, (hipStream_t)stream, blocks, block_stride, row_stride, row_leak,
                       out, B, T, N);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_agreement(const unsigned char* joint_tgt, const unsigned char* dual_tgt, const unsigned char* youtube_tgt,
                             const float* max_logit_joint, const float* max_logit_dual, const float* q_joint, const float* q_dual,
                             int kind, float* tgt_out, float* iou, unsigned char* conf, int B, int T, int N, void* stream) {
    TAN_REQUIRE(joint_tgt && dual_tgt && youtube_tgt && max_logit_joint && max_logit_dual && q_joint && q_dual && tgt_out && iou && conf);
    TAN_REQUIRE(kind >= 0 && kind <= 3 && N <= 64 && (size_t)2 * N * T <= 64 * 1024);
    hipLaunchKernelGGL(agreement_kernel, dim3(B), dim3(256), (size_t)2 * N * T, (hipStream_t)stream, joint_tgt, dual_tgt, youtube_tgt,
                       max_logit_joint, max_logit_dual, q_joint, q_dual, kind, tgt_out, iou, conf, B, T, N);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_masked_quantile(const float* x, const unsigned char* invalid, int n, float q, float* out, void* stream) {
    TAN_REQUIRE(x && out && n > 0 && n <= 8192);
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    hipLaunchKernelGGL(masked_quantile_kernel, dim3(1), dim3(1024), (size_t)p2 * 4, (hipStream_t)stream, x, invalid, n, q, out, p2);
    TAN_LAUNCH_CHECK();
    return 0;
}

// ---- stage-2 glue of get_loss in ONE launch (train/loss.py:280-290, 309-328, 345): from the per-sentence maxima md / mj of the last-stage
// same-video logits (tan_diag_max): z-scores over the real sentences, metric = -(z_d + z_j), th = quantile(metric, q), the kept-sentence
// mask and the rows that still own a positive; the alignability labels (1: both maxima above their medians, 0: both below, 2: ignore;
// 0 near the video's ends), their selection / target / counts / pos_weight; and confidence-ratio.  ~70 tiny torch kernels before.
// One block; `buf` (dynamic LDS) holds npow2 floats for the three bitonic sorts.
// (m = number of valid entries: the caller has it from its block sums -- counting them here with one LDS atomic per entry serialised
//  2 048 atomics per sort, round 6)
__device__ float s2_quantile(float* v, const float* x, const unsigned char* invalid, int n, int npow2, float q, int m) {
    __syncthreads();
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) v[i] = (i < n && !invalid[i]) ? x[i] : INFINITY;
    __syncthreads();
    if (m > 0) {
        // Two order statistics by RADIX SELECT on order-preserving keys instead of a sort: four passes of a 256-bin histogram (LDS
        // atomics, 2-8 entries per thread) narrow the key of the k-th smallest entry byte by byte; exact (it IS the entry the sort puts at
        // position k).  ~2 us per pass where the 66 barrier-separated stages of the bitonic sort took ~23 us per quantile at n = 2 048 --
        // three quantiles were most of this one-workgroup launch on the stage-2 step's critical section.  (An n^2 rank count, tried
        // first, is 4 M comparisons on ONE CU: 800 us.)
        __shared__ unsigned hist[256];
        __shared__ unsigned sel_prefix, sel_k;
        const float rank = q * (float)(m - 1);
        const float lo = floorf(rank);
        const int il = (int)lo, ih = min(il + 1, m - 1);
        float res[2];
        for (int which = 0; which < 2; ++which) {
            if (which == 1 && ih == il) { res[1] = res[0]; break; }
            if (threadIdx.x == 0) { sel_prefix = 0u; sel_k = (unsigned)(which ? ih : il); }
            unsigned mask = 0u;
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
                __syncthreads();
                const unsigned prefix = sel_prefix;
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    if (invalid[i]) continue;
                    const unsigned b = __float_as_uint(v[i]);
                    const unsigned key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (threadIdx.x < 64) {          // wave 0: the bin that holds entry number sel_k of the surviving keys
                    const int l = threadIdx.x;
                    const unsigned h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
                    const unsigned tot = h0 + h1 + h2 + h3;
                    unsigned inc = tot;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned t = __shfl_up(inc, o, 64);
                        if (l >= o) inc += t;
                    }
                    const unsigned before = inc - tot, k = sel_k;
                    if (k >= before && k < inc) {          // exactly one lane
                        unsigned r = k - before, bin;
                        if (r < h0) bin = 0;
                        else if ((r -= h0) < h1) bin = 1;
                        else if ((r -= h1) < h2) bin = 2;
                        else { r -= h2; bin = 3; }
                        sel_prefix = prefix | ((unsigned)(4 * l + bin) << shift);
                        sel_k = r;
                    }
                }
                mask |= 255u << shift;
                __syncthreads();
            }
            const unsigned key = sel_prefix;
            res[which] = __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
            __syncthreads();
        }
        const float w = rank - lo, a = res[0], b2 = res[1];
        return (w < 0.5f) ? a + w * (b2 - a) : b2 - (b2 - a) * (1.0f - w);  // at::lerp
    }
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = ((i & k) == 0);
                    const float a = v[i], b2 = v[ixj];
                    if ((a > b2) == up) { v[i] = b2; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
    float out = NAN;
    if (m > 0) {
        const float rank = q * (float)(m - 1);
        const float lo = floorf(rank);
        const int il = (int)lo, ih = min(il + 1, m - 1);
        const float w = rank - lo, a = v[il], b2 = v[ih];
        out = (w < 0.5f) ? a + w * (b2 - a) : b2 - (b2 - a) * (1.0f - w);  // at::lerp
    }
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(1024) void stage2_masks_kernel(const float* __restrict__ md_g, const float* __restrict__ mj_g,
                                                            const unsigned char* __restrict__ tpad_g, const float* __restrict__ tgt,
                                                            const float* __restrict__ abs_pos, const unsigned char* __restrict__ conf,
                                                            float q_th, int use_align, int B, int T, int N, int npow2,
                                                            float* __restrict__ metric, unsigned char* __restrict__ th_mask,
                                                            float* __restrict__ th_f, float* __restrict__ rows_pos_th,
                                                            float* __restrict__ lab_out, float* __restrict__ sel, float* __restrict__ y,
                                                            float* __restrict__ scal) {
    extern __shared__ float buf[];
    __shared__ float red[16];
    const int tid = threadIdx.x, Mp = B * N;
    // ONE global round trip: the maxima and the pad flags live in LDS from here on (a dozen dependent phases, each behind a global
    // load of the same few KiB, were most of this launch's 80 us -- on the stage-2 step's critical section between the two chains)
    float* const mdS = buf + npow2;
    float* const mjS = mdS + Mp;
    float* const metS = mjS + Mp;
    unsigned char* const padS = reinterpret_cast<unsigned char*>(metS + Mp);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = tid; i < Mp; i += 1024) {
        const float d_ = md_g[i], j_ = mj_g[i];
        const unsigned char p_ = tpad_g[i];
        mdS[i] = d_; mjS[i] = j_; padS[i] = p_;
        const float v = p_ ? 0.f : 1.f;
        a0 += v; a1 += d_ * v; a2 += j_ * v; a3 += (conf ? (float)conf[i] : 0.f) * v;
    }
    __syncthreads();
    const float* const md = mdS;
    const float* const mj = mjS;
    const unsigned char* const tpad = padS;
    const float n_valid = block_sum_1024(a0, red), mean_d = block_sum_1024(a1, red) / n_valid, mean_j = block_sum_1024(a2, red) / n_valid;
    const float conf_ratio = block_sum_1024(a3, red) / n_valid;
    a0 = 0.f; a1 = 0.f;
    for (int i = tid; i < Mp; i += 1024) {
        const float v = tpad[i] ? 0.f : 1.f, dd = md[i] - mean_d, dj = mj[i] - mean_j;
        a0 += dd * dd * v; a1 += dj * dj * v;
    }
    const float sd = sqrtf(block_sum_1024(a0, red) / (n_valid - 1.0f)), sj = sqrtf(block_sum_1024(a1, red) / (n_valid - 1.0f));
    for (int i = tid; i < Mp; i += 1024) {
        const float mt = -((md[i] - mean_d) / sd + (mj[i] - mean_j) / sj);
        metric[i] = mt; metS[i] = mt;
    }
    __syncthreads();
    const int m_valid = (int)(n_valid + 0.5f);
    const float th = s2_quantile(buf, metS, tpad, Mp, npow2, q_th, m_valid);
    for (int i = tid; i < Mp; i += 1024) {
        const bool keep = (metS[i] <= th) && !tpad[i];
        th_mask[i] = keep; th_f[i] = keep ? 1.f : 0.f;
    }
    // (rows_pos_th: stage2_rows_kernel, a launch of its own behind this one -- B*T rows x N loads issued by ONE workgroup were most of
    //  this kernel's 147 us, on the stage-2 step's critical chain)
    float med_d = 0.f, med_j = 0.f, n_sel = 0.f, n_pos = 0.f;
    if (use_align) {
        med_d = s2_quantile(buf, md, tpad, Mp, npow2, 0.5f, m_valid);
        med_j = s2_quantile(buf, mj, tpad, Mp, npow2, 0.5f, m_valid);
        a0 = 0.f; a1 = 0.f;
        for (int i = tid; i < Mp; i += 1024) {
            float lab = 2.0f;
            if (md[i] > med_d && mj[i] > med_j) lab = 1.0f;
            if (md[i] < med_d && mj[i] < med_j) lab = 0.0f;
            if (abs_pos) {
                const float centre = (abs_pos[2 * i] + abs_pos[2 * i + 1]) / 2.0f;
                if (centre < 0.2f || centre > 0.8f) lab = 0.0f;
            }
            const float sl = (lab != 2.0f && !tpad[i]) ? 1.f : 0.f;
            sel[i] = sl; y[i] = lab * sl;
            lab_out[i] = tpad[i] ? NAN : lab;
            a0 += sl; a1 += lab * sl;
        }
        n_sel = block_sum_1024(a0, red);
        n_pos = block_sum_1024(a1, red);
    }
    if (tid == 0) {
        scal[0] = n_sel; scal[1] = n_pos; scal[2] = n_sel / n_pos - 1.0f; scal[3] = conf_ratio; scal[4] = n_valid; scal[5] = th;
        scal[6] = med_d; scal[7] = med_j;
    }
}

// rows that still own a positive among the kept, real sentences (loss.py:288-290)
__global__ __launch_bounds__(256) void stage2_rows_kernel(const float* __restrict__ tgt, const unsigned char* __restrict__ tpad,
                                                          const float* __restrict__ th_f, float* __restrict__ rows_pos_th, int B, int T, int N) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B * T) return;
    const int b = r / T;
    float acc = 0.f;
    for (int k = 0; k < N; ++k) acc += tgt[(long)r * N + k] * (tpad[b * N + k] ? 0.f : 1.f) * th_f[b * N + k];
    rows_pos_th[r] = acc > 0.f ? 1.f : 0.f;
}

// BCE-with-logits of the alignability head on the selected sentences (loss.py:345-350: pos_weight = 1 / mean(label) - 1) and its top-1
// agreement: out[0] = sum(bce * sel) / n_sel, out[1] = sum(((x > 0) == y) * sel) / n_sel; backward: dx = g * dbce/dx * sel / n_sel.
__global__ __launch_bounds__(1024) void bce_sel_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ sel,
                                                           const float* __restrict__ scal, int n, float* __restrict__ out) {
    __shared__ float red[16];
    const float pw = scal[2];
    float a0 = 0.f, a1 = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float xi = x[i], yi = y[i], w = 1.0f + (pw - 1.0f) * yi;
        const float l = (1.0f - yi) * xi + w * (log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.f));
        a0 += l * sel[i];
        a1 += (((xi > 0.f) ? 1.f : 0.f) == yi ? 1.f : 0.f) * sel[i];
    }
    const float s0 = block_sum_1024(a0, red), s1 = block_sum_1024(a1, red);
    if (threadIdx.x == 0) { out[0] = s0 / scal[0]; out[1] = s1 / scal[0]; }
}
__global__ void bce_sel_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ sel,
                                   const float* __restrict__ scal, const float* __restrict__ g, int n, float* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xi = x[i], yi = y[i], w = 1.0f + (scal[2] - 1.0f) * yi, sg = 1.0f / (1.0f + expf(-xi));
    dx[i] = g[0] * ((1.0f - yi) - w * (1.0f - sg)) * sel[i] / scal[0];
}

// ---- everything get_loss derives from the batch's masks in ONE launch (prepare_inputs of loss.py; train/loss.py:58-70):
//   tpad_u8 [B*N], vpad_u8 [B*T], valid (bool) and valid_f (f32) [B*N], tgt f32 [B,T,N] = transpose of the [B,N,T] bool start/end mask,
//   and the text-column compaction of the fused sweeps: idx [Mc] (int64: the real sentences in order, then padded columns in order --
//   what a stable sort of the pad flags gives), colmap [B*N] (int32: rank among the real sentences, -1 for padded columns),
//   ci_run [Mc] (pad flags of the compacted columns).  Block 0 does the flags and the compaction (ranks from a block-wide prefix sum
// over <= 1024-column pieces), the other blocks the element-wise outputs.
__global__ __launch_bounds__(1024) void loss_prep_kernel(const float* tpad_f, const unsigned char* tpad_b, const unsigned char* vpad_b,
                                                         const unsigned char* tgt_raw, unsigned char* tpad_u8, unsigned char* vpad_u8,
                                                         unsigned char* valid, float* valid_f, float* tgt, long long* idx, int* colmap,
                                                         unsigned char* ci_run, int B, int T, int N, int Mc) {
    __shared__ int scan[1024];
    __shared__ int base_valid, base_pad, n_valid_total;
    const int tid = threadIdx.x, Mp = B * N, R = B * T;
    if (blockIdx.x > 0) {          // blocks 1..: the element-wise part (video pad bytes, the transposed f32 target), grid-strided
        const long nb = gridDim.x - 1, g0 = (long)(blockIdx.x - 1) * 1024 + tid, gs = nb * 1024;
        for (long i = g0; i < R; i += gs) vpad_u8[i] = vpad_b[i] ? 1 : 0;
        for (long i = g0; i < (long)B * T * N; i += gs) {           // tgt[b][t][n] = tgt_raw[b][n][t]
            const int n = (int)(i % N), t = (int)((i / N) % T), b = (int)(i / ((long)N * T));
            tgt[i] = tgt_raw[((long)b * N + n) * T + t] ? 1.0f : 0.0f;
        }
        return;
    }
    // block 0: the text pad flags and the column compaction (one block: the ranks need a prefix sum over all B*N columns)
    // first pass: total number of real sentences (the padded columns' ranks start behind them)
    int local = 0;
    for (int c = tid; c < Mp; c += 1024) {
        const bool pad = tpad_f ? tpad_f[c] != 0.0f : tpad_b[c] != 0;
        tpad_u8[c] = pad; valid[c] = !pad; valid_f[c] = pad ? 0.0f : 1.0f;
        local += !pad;
    }
    scan[tid] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (tid < o) scan[tid] += scan[tid + o]; __syncthreads(); }
    if (tid == 0) { n_valid_total = scan[0]; base_valid = 0; base_pad = 0; }
    __syncthreads();
    if (!idx) return;
    for (int c0 = 0; c0 < Mp; c0 += 1024) {                            // pieces of 1024 columns, in order
        const int c = c0 + tid;
        const bool in = c < Mp, pad = in && tpad_u8[c];
        const int v = in && !pad;
        scan[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                           // inclusive Hillis-Steele scan of the valid flags
            const int add = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        const int incl = scan[tid], piece_valid = scan[1023];
        if (in) {
            if (!pad) {
                const int r = base_valid + incl - 1;
                colmap[c] = r;
                if (r < Mc) { idx[r] = c; ci_run[r] = 0; }
            } else {
                colmap[c] = -1;
                const int r = n_valid_total + base_pad + (tid + 1 - incl) - 1;     // pads before and including this one, in order
                if (r < Mc) { idx[r] = c; ci_run[r] = 1; }
            }
        }
        __syncthreads();
        if (tid == 0) { base_valid += piece_valid; base_pad += (Mp - c0 < 1024 ? Mp - c0 : 1024) - piece_valid; }
        __syncthreads();
    }
}

extern "C" int tan_pos_masks(const float* tgt, const unsigned char* text_pad, float* rows_pos, float* cols_pos, int B, int T, int N,
                             void* stream) {
    TAN_REQUIRE(tgt && text_pad && rows_pos && cols_pos && B > 0 && T > 0 && N > 0);
    hipLaunchKernelGGL(pos_masks_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tgt, text_pad, rows_pos, cols_pos, T, N);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_nce_tail_fwd(const float* v_d, const float* t_d, const float* v_j, const float* t_j, const float* rows_mask,
                                const float* cols_mask, int Sd, int Sj, long R, long M, float* out2, float* counts2,
                                const float* counts_in, void* stream) {
    TAN_REQUIRE(v_d && t_d && v_j && t_j && rows_mask && cols_mask && out2 && counts2 && Sd > 0 && Sj > 0 && R > 0 && M > 0);
    hipLaunchKernelGGL(nce_tail_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v_d, t_d, v_j, t_j, rows_mask, cols_mask, Sd,
                       Sj, R, M, out2, counts2, counts_in);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_nce_tail_bwd(const float* g_dual, const float* g_joint, const float* g_mean, const float* rows_mask,
                                const float* cols_mask, const float* counts2, int Sd, int Sj, long R, long M, float* g_v_d, float* g_t_d,
                                float* g_v_j, float* g_t_j, void* stream) {
    TAN_REQUIRE((g_dual || g_joint || g_mean) && rows_mask && cols_mask && counts2 && g_v_d && g_t_d && g_v_j && g_t_j && Sd > 0 && Sj > 0 && R > 0 && M > 0);
    const long n = R > M ? R : M;
    hipLaunchKernelGGL(nce_tail_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g_dual, g_joint, g_mean, rows_mask, cols_mask,
                       counts2, Sd, Sj, R, M, g_v_d, g_t_d, g_v_j, g_t_j);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_loss_prep(const float* text_pad_f32, const unsigned char* text_pad_u8, const unsigned char* video_pad_u8,
                             const unsigned char* tgt_raw, unsigned char* tpad_u8, unsigned char* vpad_u8, unsigned char* valid,
                             float* valid_f, float* tgt, long* idx, int* colmap, unsigned char* ci_run, int B, int T, int N, int Mc,
                             void* stream) {
    TAN_REQUIRE((text_pad_f32 != nullptr) != (text_pad_u8 != nullptr));
    TAN_REQUIRE(video_pad_u8 && tgt_raw && tpad_u8 && vpad_u8 && valid && valid_f && tgt && B > 0 && T > 0 && N > 0);
    TAN_REQUIRE(!idx || (colmap && ci_run && Mc > 0 && Mc <= B * N));
    const long nel = (long)B * T * N;
    const unsigned nblk = 1 + (unsigned)((nel + 4095) / 4096 < 64 ? (nel + 4095) / 4096 : 64);
    hipLaunchKernelGGL(loss_prep_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, text_pad_f32, text_pad_u8, video_pad_u8, tgt_raw,
                       tpad_u8, vpad_u8, valid, valid_f, tgt, (long long*)idx, colmap, ci_run, B, T, N, Mc);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_stage2_masks(const float* md, const float* mj, const unsigned char* text_pad, const float* tgt, const float* abs_text_pos,
                                const unsigned char* conf, float q_th, int use_align, int B, int T, int N, float* metric,
                                unsigned char* th_mask, float* th_f, float* rows_pos_th, float* lab, float* sel, float* y, float* scal8,
                                void* stream) {
    TAN_REQUIRE(md && mj && text_pad && tgt && metric && th_mask && th_f && rows_pos_th && scal8 && B > 0 && T > 0 && N > 0);
    TAN_REQUIRE(!use_align || (lab && sel && y));
    const int Mp = B * N;
    TAN_REQUIRE(Mp <= 8192);
    int p2 = 4;                      // (>= 4: the rank-counting quantile reads the buffer in 16-byte pieces)
    while (p2 < Mp) p2 <<= 1;
    const size_t lds = ((size_t)p2 + 3 * (size_t)Mp) * 4 + (size_t)Mp;          // sort buffer | md | mj | metric | pad flags
    static std::atomic<unsigned long long> lds_done{0};
    const hipError_t attr = ensure_dyn_lds((const void*)stage2_masks_kernel, 152 * 1024, lds_done);      // (+ ~1.2 KiB of static LDS; 8192 sentences need 139 KiB)
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(stage2_masks_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, md, mj, text_pad, tgt, abs_text_pos, conf,
                       q_th, use_align, B, T, N, p2, metric, th_mask, th_f, rows_pos_th, lab, sel, y, scal8);
    TAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(stage2_rows_kernel, dim3(cdiv((long)B * T, 256)), dim3(256), 0, (hipStream_t)stream, tgt, text_pad, (const float*)th_f,
                       rows_pos_th, B, T, N);
    TAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int tan_bce_sel_fwd(const float* x, const float* y, const float* sel, const float* scal8, int n, float* out2, void* stream) {
    TAN_REQUIRE(x && y && sel && scal8 && out2 && n > 0);
    hipLaunchKernelGGL(bce_sel_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, sel, scal8, n, out2);
    TAN_LAUNCH_CHECK();
    return 0;
}
extern "C" int tan_bce_sel_bwd(const float* x, const float* y, const float* sel, const float* scal8, const float* g, int n, float* dx,
                               void* stream) {
    TAN_REQUIRE(x && y && sel && scal8 && g && dx && n > 0);
    hipLaunchKernelGGL(bce_sel_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, sel, scal8, g, n, dx);
    TAN_LAUNCH_CHECK();
    return 0;
}
