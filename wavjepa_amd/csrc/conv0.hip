// Conv layer 0 (C_in*k taps, stride s, no bias) + GroupNorm(C, C) + erf-GELU, forward and backward, gfx950.
// C_in*k is tiny (10 taps for mono audio), so this is not a GEMM shape: it is an HBM-bound producer of the
// [N][P][C] channels-last bf16 activation (1.69 GB at N=256), written exactly once.  The conv is recomputed from
// the audio wherever it is needed (statistics pass, apply pass, both backward passes) instead of being stored.
//   thread = 4 consecutive channels x one time parity;  a wave stores 512 contiguous bytes per time step.
//
// Backward.  With dz_t = dact_t gelu'(z_t), A1 = sum_t dz_t, A2 = sum_t dz_t xh_t, GroupNorm gives
//   dy_t = rstd gamma (dz_t - A1/L - xh_t A2/L)   for EVERY t, and   dw[c][q] = sum_n sum_t dy_t x_{t,q}.
// Splitting the sum,  dw = sum_n rstd gamma [ S - (A1/L) X1 - (A2/L) rstd (YX - mean X1) ]  with
//   S[q]  = sum_t dz_t x_{t,q}      only rows with a non-zero output gradient contribute (the student's context: ~20 %),
//   X1[q] = sum_t x_{t,q},  YX[q] = sum_t y_t x_{t,q}      gradient-independent: accumulated by the FORWARD statistics pass.
// So the backward is ONE pass over the listed active rows (A1, A2, S) plus a tiny per-(n,c) finalize; the dense part of
// GroupNorm's backward never touches the activation-sized tensors.
#include <string.h>
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int TC = 256;   // output time steps per workgroup
constexpr int NTH = 256;  // threads

struct Geo {
    int N, C_in, L, C, k, stride, L_out, P;
    long clip_stride;   // elements between consecutive clips of the audio (C_in * L when packed; larger for one channel of a
                        // multi-channel batch: wavjepa/extractors/audio_channel_feature_extractor.py:163-172 runs x[:, [c]])
};

template <int TAPS>
__device__ __forceinline__ void load_weights(float (&w)[4][TAPS], const bf16_t* __restrict__ wsrc, int c4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) w[j][tp] = bf2f(wsrc[(long)(c4 + j) * TAPS + tp]);
}

// audio chunk for output steps [t0, t0+TC): xs[ci][i] = audio[n][ci][t0*stride + i], i < span
__device__ __forceinline__ int stage_audio(float* xs, const bf16_t* __restrict__ audio, const Geo& g, int n, int t0, int span_max) {
    const int span = min(span_max, g.L - t0 * g.stride);
    for (int ci = 0; ci < g.C_in; ++ci)
        for (int i = threadIdx.x; i < span_max; i += NTH)
            xs[ci * span_max + i] = i < span ? bf2f(audio[(long)n * g.clip_stride + (long)ci * g.L + (long)t0 * g.stride + i]) : 0.f;
    return span;
}

template <int TAPS>
__device__ __forceinline__ void tap_offsets(int (&off)[TAPS], const Geo& g, int span_max) {
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
        const int ci = tp / g.k, kk = tp - ci * g.k;
        off[tp] = ci * span_max + kk;
    }
}

template <int TAPS>
__device__ __forceinline__ void conv4(float (&y)[4], float (&x)[TAPS], const float (&w)[4][TAPS], const float* xs,
                                      const int (&off)[TAPS], int tl, int stride) {
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) x[tp] = xs[off[tp] + tl * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) a = fmaf(x[tp], w[j][tp], a);
        y[j] = bf2f(f2bf(a));  // the conv output is a bf16 tensor in the reference's autocast flow
    }
}

// floats per partial record of the forward statistics pass
template <int TAPS>
__host__ __device__ constexpr long fwd_record(int C, int want_yx) { return want_yx ? 2L * C + (long)C * TAPS + TAPS : 2L * C; }

// out[n][i] = sum over chunks (in chunk order) of part[n][chunk][i]; the record is split over up to three outputs
// ([0, n0) -> o0, [n0, n0 + n1) -> o1, the rest -> o2, each [N][.]).  cnt_off != NULL: clip n only has
// ceil((cnt_off[n+1] - cnt_off[n]) / per_chunk) chunks written (sparse backward), else `chunks`.
__global__ __launch_bounds__(256) void conv0_fold_kernel(const float* __restrict__ part, int chunks, long rec, float* __restrict__ o0,
                                                         long n0, float* __restrict__ o1, long n1, float* __restrict__ o2,
                                                         const int32_t* __restrict__ cnt_off, int per_chunk) {
    const int n = blockIdx.y;
    int nch = chunks;
    if (cnt_off) nch = min(chunks, (cnt_off[n + 1] - cnt_off[n] + per_chunk - 1) / per_chunk);
    const float* src = part + (long)n * chunks * rec;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < rec; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int ch = 0; ch < nch; ++ch) s += src[ch * rec + i];
        if (i < n0) o0[n * n0 + i] = s;
        else if (i < n0 + n1) o1[n * n1 + (i - n0)] = s;
        else o2[n * (rec - n0 - n1) + (i - n0 - n1)] = s;
    }
}

// ---- pass 1 (forward) on the matrix cores (C % 16 == 0): per-(n,c) sum and sum of squares of the bf16-rounded conv output;
//      when want_yx also YX[n][c][q] = sum_t y_t x_{t,q} and X1[n][q] = sum_t x_{t,q} (what the backward needs of the dense
//      time axis).  The conv is a K = taps (padded to 32) MFMA, its bf16-rounded output feeds a second MFMA with K = time that
//      accumulates YX, so the VALU only rounds and sums.
//      Every workgroup (one TCS-step chunk of one clip) STORES its partial record
//          part[n][chunk][ 2C sums | C*TAPS yx | TAPS x1 ]
//      and conv0_fold_kernel adds the chunks in chunk order: no float atomics, so the GroupNorm statistics -- and with them
//      every activation of the step -- are bit-reproducible from run to run.
//   mfma(P, Q): lane (i, g) gets  sum_k Q[row i][k] P[row 4g+r][k], r < 4;  operand lane (i, g) holds row i, k = 8g .. 8g+7.
//   conv : P = audio patches (rows = 16 time steps), Q = weights (rows = 16 channels)  -> lane: channel i, times 4g+r
//   YX   : Q' = that result for two 16-step blocks (row = channel i, logical k = 8g+j <-> time kappa(g,j) = j<4 ? 4g+j : 16+4g+j-4),
//          P' = audio patches with rows = taps and the same kappa order                  -> lane: channel i, taps 4g+r
// Time steps per workgroup of the statistics pass.  Round 4: a constant 1024 -- at 256 clips x 6430 steps 1792 workgroups for the 768 the
// chip holds (160 VGPRs: three 4-wave workgroups per CU), i.e. 2.33 rounds that cost three.  Now the chunk count is chosen so that the
// launch is about ONE round (256 clips: 3 chunks of 2144 steps = 768 workgroups; fewer clips: more, shorter chunks), a multiple of 32.
constexpr int STATS_SLOTS = 768;
inline int stats_tcs(int N, int L_out, int C_in) {
    static int forced = -1;
    if (forced < 0) forced = wj_lab_env_int("WJ_CONV0_STATS_TCS", 0);     // A/B runs (1024: the round-4 chunks)
    if (forced >= 32) return forced / 32 * 32;
    int chunks = (STATS_SLOTS + N / 2) / (N > 0 ? N : 1);
    const int most = (L_out + 255) / 256;
    chunks = chunks < 1 ? 1 : (chunks > most ? most : chunks);
    const int tcs = ((L_out + chunks - 1) / chunks + 31) / 32 * 32;
    const int cap = 2176 / (C_in > 0 ? C_in : 1) / 32 * 32;           // the chunk's audio samples live in LDS (bf16, C_in x ~5 tcs): <= ~44 KB
    return tcs > cap ? (cap < 32 ? 32 : cap) : tcs;
}
// ---- pass 1, algebraic form (round 5; WJ_CONV0_STATS=mfma selects the pass below instead).  Everything pass 1 produces is a sum over
// the time axis of products of the conv output y_t = sum_k w_k x_{t,k} with 1, y_t or a patch element x_{t,q}:
//     sum_t y_t = w . X1,   sum_t y_t^2 = w^T X2 w,   sum_t y_t x_{t,q} = (w^T X2)_q,   with X1 = sum_t x_t, X2 = sum_t x_t x_t^T
// -- a TAPS-vector and a TAPS x TAPS matrix per CLIP, independent of the 512 channels: 0.7 MFLOP per clip instead of a pass that computes
// the whole conv output a first time only to sum it (256 us of the step's exposed start, VALU-bound on rounding and summing 421 M values).
// In DOUBLE precision throughout: a high-pass filter on a smooth signal makes w^T X2 w a difference of terms 1e4-1e6 times its size
// (fp32 Grams lose the variance there; the products of bf16 samples are exact in either precision, the sums are not).
// What changes: these are the sums of the UNROUNDED conv output, the pass below sums the bf16-rounded one as the reference's GroupNorm
// sees it.  The rounding errors of 6430 values average out: the means differ by <= ~1.5e-4 of the channel's standard deviation, rstd by a
// few 1e-5 relative (tests/test_ops_gpu.py::test_conv0_fwd_bwd) -- 1/30 and 1/100 of one bf16 step of the activations they shift and
// scale, and on the fp32 side of the reference's own values.
// conv0_gram_kernel: one workgroup per (chunk of TCS time steps, clip) STORES part[n][chunk][TAPS | TAPS * TAPS]; conv0_fold_f64_kernel adds
// the chunks in order (no atomics: bit-reproducible).  conv0_stats_from_gram_kernel: one thread per (clip, channel).
template <int TAPS>
__global__ __launch_bounds__(256) void conv0_gram_kernel(const bf16_t* __restrict__ audio, double* __restrict__ part, Geo g, int span, int TCS) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xa[];   // [C_in][span] samples of this chunk (0 past the clip)
    constexpr int G = TAPS + TAPS * TAPS;
    __shared__ double red[4][G];
    const int n = blockIdx.y, t0 = blockIdx.x * TCS;
    for (int ci = 0; ci < g.C_in; ++ci)
        for (int i = threadIdx.x; i < span; i += 256) {
            const long src = (long)t0 * g.stride + i;
            xa[ci * span + i] = src < g.L ? audio[(long)n * g.clip_stride + (long)ci * g.L + src] : f2bf(0.f);
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tcount = min(TCS, g.L_out - t0);
    for (int p = lane; p < G; p += 64) {
        const bool one = p < TAPS;
        const int qa = one ? p : (p - TAPS) / TAPS, qb = one ? 0 : (p - TAPS) - qa * TAPS;
        const int ca = qa / g.k, cb = qb / g.k;
        const int offa = ca * span + (qa - ca * g.k), offb = cb * span + (qb - cb * g.k);
        double acc = 0.0;
        for (int t = wave; t < tcount; t += 4) {
            const float a = bf2f(xa[offa + t * g.stride]);
            const float b = one ? 1.0f : bf2f(xa[offb + t * g.stride]);
            acc += (double)(a * b);                    // the product of two bf16 values is exact in fp32
        }
        red[wave][p] = acc;
    }
    __syncthreads();
    double* o = part + ((long)n * gridDim.x + blockIdx.x) * G;
    for (int p = threadIdx.x; p < G; p += 256) o[p] = (red[0][p] + red[1][p]) + (red[2][p] + red[3][p]);
}

// out[n][i] = sum over chunks, in chunk order, of part[n][chunk][i]
__global__ __launch_bounds__(256) void conv0_fold_f64_kernel(const double* __restrict__ part, int chunks, int rec, double* __restrict__ out) {
    const int n = blockIdx.y;
    const double* src = part + (long)n * chunks * rec;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rec; i += gridDim.x * 256) {
        double s = 0.0;
        for (int ch = 0; ch < chunks; ++ch) s += src[(long)ch * rec + i];
        out[(long)n * rec + i] = s;
    }
}

template <int TAPS>
__global__ __launch_bounds__(256) void conv0_stats_from_gram_kernel(const double* __restrict__ gram, const bf16_t* __restrict__ wsrc,
                                                                    float* __restrict__ sums, float* __restrict__ yx, float* __restrict__ x1,
                                                                    Geo g) {
    constexpr int G = TAPS + TAPS * TAPS;
    __shared__ double gs[G];
    const int n = blockIdx.y;
    for (int p = threadIdx.x; p < G; p += 256) gs[p] = gram[(long)n * G + p];
    __syncthreads();
    if (x1 && blockIdx.x == 0 && threadIdx.x < TAPS) x1[(long)n * TAPS + threadIdx.x] = (float)gs[threadIdx.x];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.C) return;
    double w[TAPS];
#pragma unroll
    for (int q = 0; q < TAPS; ++q) w[q] = (double)bf2f(wsrc[(long)c * TAPS + q]);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < TAPS; ++q) s1 += w[q] * gs[q];
#pragma unroll
    for (int q = 0; q < TAPS; ++q) {
        double v = 0.0;                                  // (w^T X2)_q = sum_t y_t x_{t,q}
#pragma unroll
        for (int k = 0; k < TAPS; ++k) v += w[k] * gs[TAPS + k * TAPS + q];
        s2 += v * w[q];
        if (yx) yx[((long)n * g.C + c) * TAPS + q] = (float)v;
    }
    sums[((long)n * g.C + c) * 2] = (float)s1;
    sums[((long)n * g.C + c) * 2 + 1] = (float)fmax(s2, 0.0);
}

template <int TAPS>
__global__ __launch_bounds__(256) void conv0_stats_mfma_kernel(const bf16_t* __restrict__ audio, const bf16_t* __restrict__ wsrc,
                                                               float* __restrict__ part, int want_yx, Geo g, int span, int TCS) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xa[];   // [C_in][span] samples of this chunk (0 past the clip)
    constexpr int QT = (TAPS + 15) / 16;
    const int n = blockIdx.y, t0 = blockIdx.x * TCS;
    const long rec = fwd_record<TAPS>(g.C, want_yx);
    float* sums = part + ((long)n * gridDim.x + blockIdx.x) * rec;   // [C][2]
    float* yx = want_yx ? sums + 2L * g.C : nullptr;                 // [C][TAPS]
    float* x1 = want_yx ? yx + (long)g.C * TAPS : nullptr;           // [TAPS]
    for (int ci = 0; ci < g.C_in; ++ci)
        for (int i = threadIdx.x; i < span; i += 256) {
            const long src = (long)t0 * g.stride + i;
            xa[ci * span + i] = src < g.L ? audio[(long)n * g.clip_stride + (long)ci * g.L + src] : f2bf(0.f);
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, gq = lane >> 4;
    const int tcount = min(TCS, g.L_out - t0);
    const bf16_t zero = f2bf(0.f);
    int offc[8];
    bool okc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = 8 * gq + j, ci = q / g.k;
        okc[j] = q < TAPS;
        offc[j] = ci * span + (q - ci * g.k);
    }
    int offq[QT];
    bool okq[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = 16 * qt + i, ci = q / g.k;
        okq[qt] = q < TAPS;
        offq[qt] = ci * span + (q - ci * g.k);
    }
    const int ntile = g.C / 16;
    for (int ctg = wave * 8; ctg < ntile; ctg += 32) {            // 8 channel tiles per wave per pass
        bf16x8 wf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                wf[u][j] = (ctg + u < ntile && okc[j]) ? wsrc[(long)((ctg + u) * 16 + i) * TAPS + 8 * gq + j] : zero;
        f32x4 yxa[8][QT];
        float s1[8], s2[8], xs1[QT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s1[u] = s2[u] = 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) yxa[u][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) xs1[qt] = 0.f;
        for (int tb = 0; tb < tcount; tb += 32) {
            bf16x8 p0, p1, pq[QT];
            const bool v0 = tb + i < tcount, v1 = tb + 16 + i < tcount;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0[j] = (v0 && okc[j]) ? xa[offc[j] + (tb + i) * g.stride] : zero;
                p1[j] = (v1 && okc[j]) ? xa[offc[j] + (tb + 16 + i) * g.stride] : zero;
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tt = tb + (j < 4 ? 4 * gq + j : 12 + 4 * gq + j);
                    pq[qt][j] = (okq[qt] && tt < tcount) ? xa[offq[qt] + tt * g.stride] : zero;
                }
            if (yx && ctg == 0) {                                  // wave 0's first pass also sums the patches themselves
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs1[qt] += bf2f(pq[qt][j]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p0, wf[u], z4, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, wf[u], z4, 0, 0, 0);
                bf16x8 yq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    yq[r] = f2bf(d0[r]);                            // the conv output is a bf16 tensor in the reference flow
                    yq[4 + r] = f2bf(d1[r]);
                    const float a = bf2f(yq[r]), b = bf2f(yq[4 + r]);
                    s1[u] += a + b;
                    s2[u] = fmaf(a, a, fmaf(b, b, s2[u]));
                }
                if (yx) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) yxa[u][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pq[qt], yq, yxa[u][qt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ctg + u >= ntile) continue;
            const int c = (ctg + u) * 16 + i;
            float a = s1[u], b = s2[u];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (gq == 0) {
                sums[c * 2 + 0] = a;
                sums[c * 2 + 1] = b;
            }
            if (yx) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 16 * qt + 4 * gq + r;
                        if (q < TAPS) yx[(long)c * TAPS + q] = yxa[u][qt][r];
                    }
            }
        }
        if (yx && ctg == 0) {                                      // wave 0
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float v = xs1[qt];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (gq == 0 && okq[qt]) x1[16 * qt + i] = v;
            }
        }
    }
}

// ---- pass 2 (forward): normalise, GELU, write channels-last bf16; emit mean / rstd ---------------------------
template <int TAPS>
__global__ __launch_bounds__(NTH) void conv0_apply_kernel(const bf16_t* __restrict__ audio, const bf16_t* __restrict__ wsrc,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ sums, bf16_t* __restrict__ act,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o, Geo g,
                                                          int span_max, float eps) {
    extern __shared__ float xs[];
    const int n = blockIdx.y, t0 = blockIdx.x * TC;
    const int half = threadIdx.x >> 7, cl = threadIdx.x & 127;
    stage_audio(xs, audio, g, n, t0, span_max);
    __syncthreads();
    const int tmax = min(TC, g.P - t0);
    const float invL = 1.0f / (float)g.L_out;
    for (int cb = 0; cb < g.C; cb += 512) {
        const int c4 = cb + cl * 4;
        if (c4 >= g.C) continue;
        float w[4][TAPS];
        load_weights<TAPS>(w, wsrc, c4);
        float mu[4], rs[4], ga[4], be[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s1 = sums[((long)n * g.C + c4 + j) * 2], s2 = sums[((long)n * g.C + c4 + j) * 2 + 1];
            mu[j] = s1 * invL;
            const float var = fmaxf(s2 * invL - mu[j] * mu[j], 0.f);
            rs[j] = rsqrtf(var + eps);
            ga[j] = gamma[c4 + j];
            be[j] = beta[c4 + j];
            if (blockIdx.x == 0 && half == 0) {
                mean_o[(long)n * g.C + c4 + j] = mu[j];
                rstd_o[(long)n * g.C + c4 + j] = rs[j];
            }
        }
        int off[TAPS];
        tap_offsets<TAPS>(off, g, span_max);
        for (int tl = half; tl < tmax; tl += 2) {
            const int t = t0 + tl;
            bf16x4 o;
            if (t < g.L_out) {
                float y[4], x[TAPS];
                conv4<TAPS>(y, x, w, xs, off, tl, g.stride);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // this pass is VALU-bound: erf-GELU on the packed fp32 pipe
                    f32x2 z, gl, unused;
                    z.x = (y[j] - mu[j]) * rs[j] * ga[j] + be[j];
                    z.y = (y[j + 1] - mu[j + 1]) * rs[j + 1] * ga[j + 1] + be[j + 1];
                    gelu_pk<false>(z, gl, unused);
                    o[j] = f2bf(gl.x); o[j + 1] = f2bf(gl.y);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(0.f);
            }
            *reinterpret_cast<bf16x4*>(act + ((long)n * g.P + t) * g.C + c4) = o;
        }
    }
}

// ---- pass 2 on the matrix cores (C % 64 == 0, taps <= 32): the VALU form above spends 10 of its ~25 issue slots per output on the
//      convolution itself; here the conv is one K = 32 MFMA per 16 steps x 16 channels (as in the statistics pass, so both passes now
//      round the same fp32 sums to bf16) and the vector unit only normalises, applies the erf-GELU and packs.
//   mfma(P, Q): lane (i, g) gets sum_k Q[row i][k] P[row 4g+r][k];  P = weights, Q = audio patches  -> lane: time step i, and for
//   channel tile u the four P rows 4g .. 4g+3.  The P rows of tiles 2p, 2p+1 are ASSIGNED to channels so that a lane ends with eight
//   consecutive channels of one time step: row (4g'+r) of tile u  <->  channel 32 (u >> 1) + 8 g' + 4 (u & 1) + r.  A lane then
//   writes 16 B, the four lanes of a time step 64 contiguous bytes; a wave owns 64 channels (4 tiles, 2 such stores per 16 steps), a workgroup of eight waves 512.
//   GroupNorm is applied in its folded form z = y * (rstd gamma) + (beta - mean rstd gamma) (two constants per channel instead of four:
//   a lane serves 16 channels).
constexpr int TCA = 512;  // output time steps per workgroup of the MFMA form
// MINB: workgroups per CU the register allocation must allow (the second __launch_bounds__ argument counts WAVES PER SIMD: 4 for two
// 8-wave workgroups).  At 134 VGPRs (MINB = 1) a CU holds 12 waves = ONE 8-wave workgroup, two
// waves per SIMD under a loop of dependent transcendentals; MINB = 2 caps the kernel at 128 VGPRs: two workgroups, four waves per SIMD.
template <int TAPS, int MINB = 1>
__global__ __launch_bounds__(512, MINB == 2 ? 4 : 1) void conv0_apply_mfma_kernel(const bf16_t* __restrict__ audio, const bf16_t* __restrict__ wsrc,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ sums, bf16_t* __restrict__ act,
                                                               float* __restrict__ mean_o, float* __restrict__ rstd_o, Geo g, int span,
                                                               float eps) {
    extern __shared__ __attribute__((aligned(16))) bf16_t xa[];   // [C_in][span] samples of this chunk (0 past the clip)
    const int n = blockIdx.y, t0 = blockIdx.x * TCA;
    for (int ci = 0; ci < g.C_in; ++ci)
        for (int i = threadIdx.x; i < span; i += 512) {
            const long src = (long)t0 * g.stride + i;
            xa[ci * span + i] = src < g.L ? audio[(long)n * g.clip_stride + (long)ci * g.L + src] : f2bf(0.f);
        }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, gq = lane >> 4;
    const int tmax = min(TCA, g.P - t0);            // rows this workgroup writes (rows >= L_out are zero padding)
    const int tlive = min(TCA, g.L_out - t0);       // rows that carry a conv output
    const float invL = 1.0f / (float)g.L_out;
    const bf16_t zero = f2bf(0.f);
    int offc[8];
    bool okc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = 8 * gq + j, ci = q / g.k;
        okc[j] = q < TAPS;
        offc[j] = ci * span + (q - ci * g.k);
    }
    for (int cb = wave * 64; cb < g.C; cb += 512) {        // eight waves: 64 channels (four tiles) each
        bf16x8 wf[4];
        float ka[4][4], kb[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int crow = cb + 32 * (u >> 1) + 8 * (i >> 2) + 4 * (u & 1) + (i & 3);     // channel of P row i of tile u
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[u][j] = okc[j] ? wsrc[(long)crow * TAPS + 8 * gq + j] : zero;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = cb + 32 * (u >> 1) + 8 * gq + 4 * (u & 1) + r;                // channel of this lane's output r of tile u
                const float s1 = sums[((long)n * g.C + c) * 2], s2 = sums[((long)n * g.C + c) * 2 + 1];
                const float mu = s1 * invL;
                const float var = fmaxf(s2 * invL - mu * mu, 0.f);
                const float rs = rsqrtf(var + eps);
                ka[u][r] = rs * gamma[c];
                kb[u][r] = fmaf(-mu, ka[u][r], beta[c]);
                if (blockIdx.x == 0 && i == 0) {
                    mean_o[(long)n * g.C + c] = mu;
                    rstd_o[(long)n * g.C + c] = rs;
                }
            }
        }
        auto patch = [&](int t) {
            bf16x8 pf;
            const bool live = t < tlive;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (live && okc[j]) ? xa[offc[j] + t * g.stride] : zero;
            return pf;
        };
        bf16x8 pf = patch(i);
        for (int tb = 0; tb < tmax; tb += 16) {
            const int t = tb + i;
            const bool live = t < tlive;
            const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], pf, z4, 0, 0, 0);   // all four in flight
            pf = patch(t + 16);                                     // the next block's patches travel from LDS meanwhile
            bf16_t* dst = act + ((long)n * g.P + t0 + t) * g.C + cb + 8 * gq;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 za, zb, ga, gb, unused;
                    // the conv output is a bf16 tensor in the reference's autocast flow
                    za.x = fmaf(bf2f(f2bf(d[2 * p][r])), ka[2 * p][r], kb[2 * p][r]);
                    za.y = fmaf(bf2f(f2bf(d[2 * p][r + 1])), ka[2 * p][r + 1], kb[2 * p][r + 1]);
                    zb.x = fmaf(bf2f(f2bf(d[2 * p + 1][r])), ka[2 * p + 1][r], kb[2 * p + 1][r]);
                    zb.y = fmaf(bf2f(f2bf(d[2 * p + 1][r + 1])), ka[2 * p + 1][r + 1], kb[2 * p + 1][r + 1]);
                    gelu_pk<false>(za, ga, unused);
                    gelu_pk<false>(zb, gb, unused);
                    o[r] = f2bf(ga.x); o[r + 1] = f2bf(ga.y);
                    o[4 + r] = f2bf(gb.x); o[4 + r + 1] = f2bf(gb.y);
                }
                if (!live) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = zero;
                }
                if (t < tmax) *reinterpret_cast<bf16x8*>(dst + 32 * p) = o;
            }
        }
    }
}

// ---- backward, pass over the listed rows: A1 = sum dz, A2 = sum dz xh, S[q] = sum dz x_q  per (n, c) ------------------
// Every workgroup = (chunk of BR listed rows, clip) STORES its partial part[n][chunk][C][2 + TAPS]; conv0_fold_kernel adds a
// clip's chunks in order (no float atomics).  rows == NULL: every row t < L_out of every clip.
constexpr int BR = 256;
template <int TAPS>
__global__ __launch_bounds__(NTH) void conv0_bwd_rows_kernel(const bf16_t* __restrict__ audio, const bf16_t* __restrict__ wsrc,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const bf16_t* __restrict__ dact, const int32_t* __restrict__ rows,
                                                             const int32_t* __restrict__ row_off, float* __restrict__ ws, Geo g) {
    extern __shared__ float xs[];                      // [BR][TAPS] audio patches of this chunk's rows
    int* tl = reinterpret_cast<int*>(xs + BR * TAPS);  // [BR] their time steps
    float* red = xs + BR * TAPS + BR;                  // [(2 + TAPS) * 4][128] partials of the odd half
    const int n = blockIdx.y;
    const int first = rows ? row_off[n] : 0;
    const int cnt = rows ? row_off[n + 1] - first : g.L_out;
    const int j0 = blockIdx.x * BR;
    if (j0 >= cnt) return;
    const int jn = min(BR, cnt - j0);
    const int half = threadIdx.x >> 7, cl = threadIdx.x & 127;
    for (int j = threadIdx.x; j < jn; j += NTH) tl[j] = rows ? rows[first + j0 + j] - n * g.P : j0 + j;
    __syncthreads();
    for (int idx = threadIdx.x; idx < jn * TAPS; idx += NTH) {
        const int j = idx / TAPS, tp = idx - j * TAPS;
        const int ci = tp / g.k, kk = tp - ci * g.k;
        xs[idx] = bf2f(audio[(long)n * g.clip_stride + (long)ci * g.L + (long)tl[j] * g.stride + kk]);
    }
    __syncthreads();
    for (int cb = 0; cb < g.C; cb += 512) {
        const int c4 = cb + cl * 4;
        const bool live = c4 < g.C;
        float acc[4][TAPS], a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) acc[j][tp] = 0.f;
        if (live) {
            float w[4][TAPS];
            load_weights<TAPS>(w, wsrc, c4);
            float mu[4], rs[4], ga[4], be[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mu[j] = mean[(long)n * g.C + c4 + j]; rs[j] = rstd[(long)n * g.C + c4 + j];
                ga[j] = gamma[c4 + j]; be[j] = beta[c4 + j];
            }
            // Rows in batches of RB: the batch's gradient rows are requested together, ahead of the arithmetic.  (Round 6, SQ counters of the
            // one-row-at-a-time loop: 45 % of the wave cycles parked at s_waitcnt behind ONE 8-byte load per row and thread, 30 % VALU -- the
            // pass was latency-bound at three waves per SIMD, not VALU-bound.)  Same rows in the same order: the sums keep their bits.
            constexpr int RB = 4;
            for (int jj0 = half; jj0 < jn; jj0 += 2 * RB) {
                bf16x4 dv[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int jq = jj0 + 2 * u < jn ? jj0 + 2 * u : jj0;          // past the end: a valid row, not used
                    dv[u] = *reinterpret_cast<const bf16x4*>(dact + ((long)n * g.P + tl[jq]) * g.C + c4);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int jj = jj0 + 2 * u;
                    if (jj >= jn) break;
                    float x[TAPS];
#pragma unroll
                    for (int tp = 0; tp < TAPS; ++tp) x[tp] = xs[jj * TAPS + tp];
                    const bf16x4 d = dv[u];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a = 0.f;
#pragma unroll
                        for (int tp = 0; tp < TAPS; ++tp) a = fmaf(x[tp], w[j][tp], a);
                        const float y = bf2f(f2bf(a));
                        const float xh = (y - mu[j]) * rs[j];
                        const float dz = bf2f(d[j]) * gelu_grad_f(xh * ga[j] + be[j]);
                        a1[j] += dz;
                        a2[j] = fmaf(dz, xh, a2[j]);
#pragma unroll
                        for (int tp = 0; tp < TAPS; ++tp) acc[j][tp] = fmaf(dz, x[tp], acc[j][tp]);
                    }
                }
            }
        }
        __syncthreads();
        if (live && half == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[(j * (2 + TAPS) + 0) * 128 + cl] = a1[j];
                red[(j * (2 + TAPS) + 1) * 128 + cl] = a2[j];
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) red[(j * (2 + TAPS) + 2 + tp) * 128 + cl] = acc[j][tp];
            }
        }
        __syncthreads();
        if (live && half == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* o = ws + (((long)n * gridDim.x + blockIdx.x) * g.C + c4 + j) * (2 + TAPS);
                o[0] = a1[j] + red[(j * (2 + TAPS) + 0) * 128 + cl];
                o[1] = a2[j] + red[(j * (2 + TAPS) + 1) * 128 + cl];
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) o[2 + tp] = acc[j][tp] + red[(j * (2 + TAPS) + 2 + tp) * 128 + cl];
            }
        }
    }
}

// ---- backward finalize: fold the clips.  Block = 32 (c, q) pairs x 8 clip lanes.
template <int TAPS>
__global__ __launch_bounds__(256) void conv0_bwd_final_kernel(const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ ws,
                                                              const float* __restrict__ yx, const float* __restrict__ x1,
                                                              float* __restrict__ dw, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, Geo g) {
    __shared__ float r0[8][33], r1[8][33], r2[8][33];
    const int pl = threadIdx.x & 31, nl = threadIdx.x >> 5;
    const int pair = blockIdx.x * 32 + pl;
    const int c = pair / TAPS, tp = pair - c * TAPS;
    const bool live = c < g.C;
    const float invL = 1.0f / (float)g.L_out;
    float sw = 0.f, sg = 0.f, sb = 0.f;
    if (live) {
        const float ga = gamma[c];
        for (int n = nl; n < g.N; n += 8) {
            const float* o = ws + ((long)n * g.C + c) * (2 + TAPS);
            const float A1 = o[0], A2 = o[1], S = o[2 + tp];
            const float mu = mean[(long)n * g.C + c], rs = rstd[(long)n * g.C + c];
            const float X1 = x1[(long)n * TAPS + tp], YX = yx[((long)n * g.C + c) * TAPS + tp];
            sw += rs * ga * (S - A1 * invL * X1 - A2 * invL * rs * (YX - mu * X1));
            sg += A2;
            sb += A1;
        }
    }
    r0[nl][pl] = sw; r1[nl][pl] = sg; r2[nl][pl] = sb;
    __syncthreads();
    if (nl == 0 && live) {
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a += r0[i][pl]; b += r1[i][pl]; d += r2[i][pl]; }
        dw[(long)c * TAPS + tp] += a;            // one thread owns (c, tp): a plain accumulate into the gradient buffer
        if (tp == 0) {
            dgamma[c] += b;
            dbeta[c] += d;
        }
    }
}

inline Geo geo_of(int N, int C_in, int L, int C, int k, int stride, int L_out, int P, long clip_stride) {
    Geo g; g.N = N; g.C_in = C_in; g.L = L; g.C = C; g.k = k; g.stride = stride; g.L_out = L_out; g.P = P;
    g.clip_stride = clip_stride > 0 ? clip_stride : (long)C_in * L;
    return g;
}

template <int TAPS>
void launch_fwd(const wj_conv0_fwd_args* a, const Geo& g, int span_max, hipStream_t s) {
    const size_t lds = (size_t)a->C_in * span_max * sizeof(float);
    dim3 grid2((a->P + TC - 1) / TC, a->N), block(NTH);
    const int want_yx = a->yx != nullptr;
    const int TCS = stats_tcs(a->N, a->L_out, a->C_in);
    const int chunks = (a->L_out + TCS - 1) / TCS;
    const long rec = fwd_record<TAPS>(a->C, want_yx);
    float* sums = a->workspace;                       // [N][C][2] folded statistics
    float* part = a->workspace + 2L * a->N * a->C;    // [N][chunks][rec] partial records
    const int span = (TCS - 1) * a->stride + a->k;
    static const int gram_mode = [] { const char* e = getenv("WJ_CONV0_STATS"); return (e && !strcmp(e, "mfma")) ? 0 : 1; }();
    constexpr long G = TAPS + (long)TAPS * TAPS;
    // (the partial Grams -- doubles -- and their fold share the scratch the pass below sizes for its partial records: they fit unless
    // C < ~4 TAPS)
    if (gram_mode && 2 * G * (chunks + 1) <= (long)chunks * fwd_record<TAPS>(a->C, 1)) {
        double* gpart = reinterpret_cast<double*>(part);          // [N][chunks][G]   (part is 8-byte aligned: 2 N C floats past a 256-byte-aligned base)
        double* gfold = gpart + (long)a->N * chunks * G;          // [N][G]
        hipLaunchKernelGGL(conv0_gram_kernel<TAPS>, dim3(chunks, a->N), dim3(256), (size_t)a->C_in * span * sizeof(bf16_t), s,
                           (const bf16_t*)a->audio, gpart, g, span, TCS);
        hipLaunchKernelGGL(conv0_fold_f64_kernel, dim3(1, a->N), dim3(256), 0, s, (const double*)gpart, chunks, (int)G, gfold);
        hipLaunchKernelGGL(conv0_stats_from_gram_kernel<TAPS>, dim3((a->C + 255) / 256, a->N), dim3(256), 0, s, (const double*)gfold,
                           (const bf16_t*)a->w, sums, a->yx, a->x1, g);
    } else {
        hipLaunchKernelGGL(conv0_stats_mfma_kernel<TAPS>, dim3(chunks, a->N), dim3(256), (size_t)a->C_in * span * sizeof(bf16_t), s,
                           (const bf16_t*)a->audio, (const bf16_t*)a->w, part, want_yx, g, span, TCS);
        hipLaunchKernelGGL(conv0_fold_kernel, dim3((unsigned)((rec + 1023) / 1024), a->N), dim3(256), 0, s, (const float*)part, chunks, rec,
                           sums, 2L * a->C, a->yx, (long)a->C * TAPS, a->x1, (const int32_t*)nullptr, 1);
    }
    static const int use_mfma = wj_lab_env_int("WJ_CONV0_APPLY_MFMA", 1);   // 0: the VALU form (A/B runs)
    if (use_mfma && a->C % 64 == 0 && TAPS <= 32) {
        const int span_a = (TCA - 1) * a->stride + a->k;
        static const int occ = wj_lab_env_int("WJ_CONV0_APPLY_OCC", 1);      // 2: 128 VGPRs, two workgroups per CU -- measured 836 against 851 us alone and nothing on top of the re-chunked statistics pass (profiles/r05_conv0_variants.log): the pass is VALU-throughput-bound, not latency-bound
        if (occ >= 2)
            hipLaunchKernelGGL((conv0_apply_mfma_kernel<TAPS, 2>), dim3((a->P + TCA - 1) / TCA, a->N), dim3(512), (size_t)a->C_in * span_a * sizeof(bf16_t),
                               s, (const bf16_t*)a->audio, (const bf16_t*)a->w, a->gamma, a->beta, (const float*)sums, (bf16_t*)a->act, a->mean,
                               a->rstd, g, span_a, a->eps);
        else
            hipLaunchKernelGGL((conv0_apply_mfma_kernel<TAPS, 1>), dim3((a->P + TCA - 1) / TCA, a->N), dim3(512), (size_t)a->C_in * span_a * sizeof(bf16_t),
                               s, (const bf16_t*)a->audio, (const bf16_t*)a->w, a->gamma, a->beta, (const float*)sums, (bf16_t*)a->act, a->mean,
                               a->rstd, g, span_a, a->eps);
        return;
    }
    hipLaunchKernelGGL(conv0_apply_kernel<TAPS>, grid2, block, lds, s, (const bf16_t*)a->audio, (const bf16_t*)a->w,
                       a->gamma, a->beta, (const float*)sums, (bf16_t*)a->act, a->mean, a->rstd, g, span_max, a->eps);
}

template <int TAPS>
void launch_bwd(const wj_conv0_bwd_args* a, const Geo& g, hipStream_t s) {
    const size_t lds = (size_t)(BR * TAPS + BR + (2 + TAPS) * 4 * 128) * sizeof(float);
    const int max_rows = a->rows ? a->max_rows : a->L_out;
    if (max_rows <= 0) return;
    static int attr = hipFuncSetAttribute((const void*)conv0_bwd_rows_kernel<TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr;
    const int chunks = (max_rows + BR - 1) / BR;
    const long rec = (long)a->C * (2 + TAPS);
    float* folded = a->workspace;                     // [N][C][2 + TAPS]
    float* part = a->workspace + (long)a->N * rec;    // [N][chunks][C][2 + TAPS]
    dim3 grid(chunks, a->N), block(NTH);
    hipLaunchKernelGGL(conv0_bwd_rows_kernel<TAPS>, grid, block, lds, s, (const bf16_t*)a->audio, (const bf16_t*)a->w, a->gamma,
                       a->beta, a->mean, a->rstd, (const bf16_t*)a->dact, a->rows, a->row_off, part, g);
    hipLaunchKernelGGL(conv0_fold_kernel, dim3((unsigned)((rec + 1023) / 1024), a->N), dim3(256), 0, s, (const float*)part, chunks, rec,
                       folded, rec, (float*)nullptr, 0L, (float*)nullptr, a->rows ? a->row_off : (const int32_t*)nullptr, BR);
    hipLaunchKernelGGL(conv0_bwd_final_kernel<TAPS>, dim3((a->C * TAPS + 31) / 32), dim3(256), 0, s, a->gamma, a->mean, a->rstd,
                       (const float*)folded, a->yx, a->x1, a->dw, a->dgamma, a->dbeta, g);
}

}  // namespace

// scratch sizes (wj_workspace_bytes): folded statistics + one partial record per (clip, chunk)
int64_t wj_conv0_fwd_ws_bytes(const wj_conv0_fwd_args* a) {
    const int TCS = stats_tcs(a->N, a->L_out, a->C_in);
    const int taps = a->C_in * a->k, chunks = (a->L_out + TCS - 1) / TCS;
    const long rec = 2L * a->C + (long)a->C * taps + taps;      // sized for the training form (yx / x1 requested)
    return (2L * a->N * a->C + (long)a->N * chunks * rec) * 4;
}
int64_t wj_conv0_bwd_ws_bytes(const wj_conv0_bwd_args* a) {
    const int taps = a->C_in * a->k;
    const int max_rows = a->max_rows > 0 ? a->max_rows : a->L_out;   // 0: sized for the dense form (every row t < L_out)
    const int chunks = (max_rows + BR - 1) / BR;
    return (long)a->N * (1 + chunks) * a->C * (2 + taps) * 4;
}

extern "C" int wj_conv0_gn_gelu_fwd(const wj_conv0_fwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->audio || !a->w || !a->gamma || !a->beta || !a->act || !a->mean || !a->rstd || !a->workspace) return WJ_ERR_ARG;
    if (a->N <= 0 || a->C <= 0 || (a->C & 15) || a->L_out <= 0 || a->P < a->L_out) return WJ_ERR_ARG;
    if ((a->L_out - 1) * a->stride + a->k > a->L) return WJ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const Geo g = geo_of(a->N, a->C_in, a->L, a->C, a->k, a->stride, a->L_out, a->P, a->audio_clip_stride);
    const int span_max = (TC - 1) * a->stride + a->k;
    if ((a->yx == nullptr) != (a->x1 == nullptr)) return WJ_ERR_ARG;
    const int taps = a->C_in * a->k;
    switch (taps) {
        case 10: launch_fwd<10>(a, g, span_max, s); break;
        case 20: launch_fwd<20>(a, g, span_max, s); break;
        default: return WJ_ERR_UNSUPPORTED;
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_conv0_gn_gelu_bwd(const wj_conv0_bwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->audio || !a->w || !a->gamma || !a->beta || !a->mean || !a->rstd || !a->dact || !a->dw || !a->dgamma ||
        !a->dbeta || !a->workspace || !a->yx || !a->x1)
        return WJ_ERR_ARG;
    if (a->N <= 0 || a->C <= 0 || (a->C & 3) || a->L_out <= 0 || a->P < a->L_out) return WJ_ERR_ARG;
    if (a->rows && (!a->row_off || a->max_rows < 0)) return WJ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const Geo g = geo_of(a->N, a->C_in, a->L, a->C, a->k, a->stride, a->L_out, a->P, a->audio_clip_stride);
    const int taps = a->C_in * a->k;
    switch (taps) {
        case 10: launch_bwd<10>(a, g, s); break;
        case 20: launch_bwd<20>(a, g, s); break;
        default: return WJ_ERR_UNSUPPORTED;
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
