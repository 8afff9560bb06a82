// On-device scene augmentation for gfx950: batched room-impulse-response convolution and segmental-SNR mixing.
//
// Reference: data_modules/scene_module/generate_scenes_batch.py -- convolve_with_rir (:12-44, torchaudio fftconvolve "full" cut
// to the input length), aggregate_noise (:47-71), add_noise (:108-150).
//
// y[b][c][t] = sum_k x[b][t-k] * h[b][c][k],  t < T           (fp32; T ~ 320 000 samples, L ~ 10^4..10^5 taps)
//
// The reference takes ONE real FFT of length T + L - 1 per signal.  Here the convolution is uniformly partitioned overlap-save
// with a block FFT that lives entirely in LDS (no multi-pass global FFT, no library):
//   * Bk = N/2 new samples per block, FFT size N = 8192 (64 KB of complex fp32 per workgroup);
//   * pass 1  spectra of every signal block  X[b][j] = FFT(x[(j-1)Bk .. (j+1)Bk))   and of every RIR partition
//             H[b][c][p] = FFT(h[p Bk .. (p+1)Bk) | 0);  only bins 0..N/2 are kept (real input: Hermitian), and two real blocks
//             ride through one complex FFT;
//   * pass 2  per pair of output blocks  Y_j = sum_p X[b][j-p] * H[b][c][p]  (complex MAC with a sliding window over X: L2-bound),
//             Hermitian extension, ONE inverse FFT for the pair (y0 + i y1), the last Bk samples are the linear convolution;
//             written (or accumulated) to y.
// FFT: 8192 = 16 x 16 x 32, three passes of register FFTs (radix 16 / 16 / 32 per thread, compile-time twiddles) with two trips
// through LDS (unit-stride or padded rows: no bank conflicts), inter-pass twiddles from one sincospif per thread and pass (the
// argument 2n/N is exact in fp32) and powers by squaring.  512 threads, 68 KB of LDS, two workgroups per CU.
// Results agree with an fp64 convolution to ~1e-6 of the output RMS (tests/test_scene_gpu.py).
//
// Segmental-SNR mix: per (b, c) sums of squares of source and noise over the window [start, start + length) are reduced in a
// fixed order (per-chunk partials, then a serial fold: bit-reproducible), a = sqrt(Ex / (En + 1e-9) * 10^(-snr/10)),
// out = source + a * noise.
#include <string.h>
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int FFT_THREADS = 512;
constexpr int MIX_CHUNKS = 64;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// 32nd roots of unity, exp(+2 pi i j / 32), j < 16 (the sign of the imaginary part is chosen at use)
__device__ constexpr float C32[16] = {1.f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f,
                                      0.f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
__device__ constexpr float S32[16] = {0.f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f, 0.923879533f, 0.98078528f,
                                      1.f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};

constexpr int ilog2(int r) { return r <= 1 ? 0 : 1 + ilog2(r / 2); }
constexpr int brev(int x, int bits) { int r = 0; for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

// R-point FFT (R = 8 / 16 / 32) on registers: natural order in, natural order out.  Decimation in frequency, every loop unrolled
// (all indices and twiddles are compile-time constants), then a bit-reversal that is pure register renaming.
// Forward = exp(-2 pi i nk / R); INV = exp(+...), no 1/R.
template <int R, bool INV>
__device__ __forceinline__ void fft_reg(float2 (&v)[R]) {
#pragma unroll
    for (int len = R; len >= 2; len >>= 1) {
        const int half = len / 2;
#pragma unroll
        for (int blk = 0; blk < R; blk += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float2 x = v[blk + j], y = v[blk + j + half];
                v[blk + j] = cadd(x, y);
                const float2 d = csub(x, y);
                const int e = j * (32 / len);               // exponent in 32nds of a turn, < 16
                if (e == 0) v[blk + j + half] = d;
                else if (e == 8) v[blk + j + half] = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                else v[blk + j + half] = cmul(d, make_float2(C32[e], INV ? S32[e] : -S32[e]));
            }
        }
    }
    float2 o[R];
#pragma unroll
    for (int k = 0; k < R; ++k) o[k] = v[brev(k, ilog2(R))];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = o[k];
}

// v[k] *= w^k for k < R, w = exp(-+ 2 pi i * num / den): one sincospif, powers by squaring / one product each (<= 4 roundings)
template <int R, bool INV>
__device__ __forceinline__ void twiddle_powers(float2 (&v)[R], int num, int den) {
    float sn, cs;
    sincospif(2.0f * (float)num / (float)den, &sn, &cs);
    float2 w[R];
    w[1] = make_float2(cs, INV ? sn : -sn);
#pragma unroll
    for (int k = 2; k < R; ++k) {
        const int low = k & (-k);
        w[k] = (low == k) ? cmul(w[k / 2], w[k / 2]) : cmul(w[k - low], w[low]);
    }
#pragma unroll
    for (int k = 1; k < R; ++k) v[k] = cmul(v[k], w[k]);
}

// N = R1 * R2 * R3 point FFT of one workgroup, three register passes with two trips through LDS.
//   n = n1 + M1 a (a < R1, n1 < M1 = R2 R3),  n1 = n2 + R3 b (b < R2, n2 < R3),  k = k1 + R1 (k2 + R2 k3)
//   pass 1  thread n1        : R1-point FFT over a, times w_N^(n1 k1)            -> LDS B[k1][n1]             (unit-stride lanes)
//   pass 2  thread (k1, n2)  : R2-point FFT over b, times w_M1^(n2 k2)           -> LDS D[k1 R2 + k2][n2]     (rows padded to R3 + 1)
//   pass 3  thread (k1, k2)  : R3-point FFT over n2                              -> LDS X[k] natural order
// The caller provides pass 1's inputs in registers (v[a] = x[n1 + M1 a]) and reads the result from LDS after the last barrier.
template <int R1, int R2, int R3>
struct Fft3 {
    static constexpr int N = R1 * R2 * R3, M1 = R2 * R3, T1 = M1, T2 = R1 * R3, T3 = R1 * R2, PITCH = R3 + 1;
    static constexpr int LDS_ELEMS = (T3 * PITCH > N ? T3 * PITCH : N);
    static_assert(T1 <= FFT_THREADS && T2 <= FFT_THREADS && T3 <= FFT_THREADS, "one work item per thread and pass");

    template <bool INV>
    static __device__ __forceinline__ void run(float2 (&v)[R1], float2* lds) {
        const int t = threadIdx.x;
        if (t < T1) {
            fft_reg<R1, INV>(v);
            twiddle_powers<R1, INV>(v, t, N);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) lds[k1 * M1 + t] = v[k1];
        }
        __syncthreads();
        float2 u[R2];
        const int k1 = t / R3, n2 = t - k1 * R3;
        if (t < T2) {
#pragma unroll
            for (int b = 0; b < R2; ++b) u[b] = lds[k1 * M1 + n2 + R3 * b];
            fft_reg<R2, INV>(u);
            twiddle_powers<R2, INV>(u, n2, M1);
        }
        __syncthreads();
        if (t < T2) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) lds[(k1 * R2 + k2) * PITCH + n2] = u[k2];
        }
        __syncthreads();
        float2 z[R3];
        if (t < T3) {
#pragma unroll
            for (int n = 0; n < R3; ++n) z[n] = lds[t * PITCH + n];
            fft_reg<R3, INV>(z);
        }
        __syncthreads();
        if (t < T3) {
            const int kk1 = t / R2, kk2 = t - kk1 * R2;
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) lds[kk1 + R1 * (kk2 + R2 * k3)] = z[k3];
        }
        __syncthreads();
    }
};

// pass 1 of the convolution.  Real input: TWO blocks share one complex FFT (z = u + i v;  U[k] = (Z[k] + conj Z[N-k]) / 2,
// V[k] = (Z[k] - conj Z[N-k]) / 2i).  One workgroup per pair: blockIdx.x < n_sig = B * ceil(nb / 2): signal blocks (b, 2q), (b, 2q+1);
// otherwise RIR partitions (b, c, 2q), (b, c, 2q+1).
template <typename F>
__global__ __launch_bounds__(FFT_THREADS) void scene_fwd_fft_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                                    float2* __restrict__ X, float2* __restrict__ H, int B, int C, int T,
                                                                    int L, long hsb, long hsc, int nb, int P) {
    constexpr int N = F::N, Bk = N / 2, NB = N / 2 + 1, R1 = N / F::M1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    const int nbp = (nb + 1) / 2, Pp = (P + 1) / 2, n_sig = B * nbp;
    const float* src;
    long first;        // source index of buffer position 0 of the pair's FIRST member; the second starts Bk later
    int hi, take;      // valid source indices [0, hi); buffer positions [0, take) come from the source, the rest are zero
    float2* out;       // spectrum of the first member; the second follows at out + NB
    bool second;
    if ((int)blockIdx.x < n_sig) {
        const int b = blockIdx.x / nbp, q = blockIdx.x - b * nbp;
        src = x + (long)b * T; first = (long)(2 * q - 1) * Bk; hi = T; take = N;
        out = X + ((long)b * nb + 2 * q) * NB; second = 2 * q + 1 < nb;
    } else {
        const int r = blockIdx.x - n_sig;
        const int b = r / (C * Pp), c = (r / Pp) % C, q = r % Pp;
        src = h + (long)b * hsb + (long)c * hsc; first = (long)(2 * q) * Bk; hi = L; take = Bk;
        out = H + (((long)b * C + c) * P + 2 * q) * NB; second = 2 * q + 1 < P;
    }
    float2 v[R1];
#pragma unroll
    for (int a = 0; a < R1; ++a) {
        const int n = (int)threadIdx.x + F::M1 * a;
        const long i0 = first + n, i1 = i0 + Bk;
        const bool ok = threadIdx.x < F::M1 && n < take;
        v[a] = make_float2((ok && i0 >= 0 && i0 < hi) ? src[i0] : 0.f, (ok && second && i1 >= 0 && i1 < hi) ? src[i1] : 0.f);
    }
    F::template run<false>(v, lds);
    for (int k = threadIdx.x; k < NB; k += FFT_THREADS) {
        const float2 zk = lds[k], zm = lds[(N - k) & (N - 1)];
        out[k] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        if (second) out[NB + k] = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
    }
}

// pass 2.  One workgroup per PAIR of output blocks (b, c, 2q), (b, c, 2q+1): the two products share every H[p] and all but one
// X[j-p] load (sliding window), and the two real results come out of ONE complex inverse FFT (z = y0 + i y1).
template <typename F>
__global__ __launch_bounds__(FFT_THREADS) void scene_mac_ifft_kernel(const float2* __restrict__ X, const float2* __restrict__ H,
                                                                     float* __restrict__ y, int B, int C, int T, int nb, int P,
                                                                     int accumulate) {
    constexpr int N = F::N, Bk = N / 2, NB = N / 2 + 1, R1 = N / F::M1;
    static_assert(2 * NB <= F::LDS_ELEMS, "both half spectra are staged side by side");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    const int nbp = (nb + 1) / 2;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);        // the blocks of one source share its spectra: keep them on one XCD's L2
    const int q = wg % nbp, c = (wg / nbp) % C, b = wg / (nbp * C);
    const int j0 = 2 * q, j1 = j0 + 1;
    const bool second = j1 < nb;
    const float2* Xb = X + (long)b * nb * NB;
    const float2* Hb = H + ((long)b * C + c) * P * NB;
    const int pmax = min(P - 1, second ? j1 : j0);
    // every thread walks the partitions once for ALL its bins (k = t + 512 i): 2 * KPT independent loads in flight per step
    constexpr int KPT = (NB + FFT_THREADS - 1) / FFT_THREADS;
    float2 a0[KPT], a1[KPT], xn[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = min((int)threadIdx.x + FFT_THREADS * i, NB - 1);      // clamped lanes recompute the last bin; never stored
        a0[i] = a1[i] = make_float2(0.f, 0.f);
        xn[i] = second ? Xb[(long)j1 * NB + k] : a0[i];       // X[j1 - p]; becomes X[j0 - (p - 1)] on the next turn
    }
    for (int p = 0; p <= pmax; ++p) {
        const bool both = p <= j0;
        float2 w[KPT], xc[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = min((int)threadIdx.x + FFT_THREADS * i, NB - 1);
            w[i] = Hb[(long)p * NB + k];
            xc[i] = Xb[(long)(both ? j0 - p : 0) * NB + k];
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            a1[i].x = fmaf(xn[i].x, w[i].x, fmaf(-xn[i].y, w[i].y, a1[i].x));
            a1[i].y = fmaf(xn[i].x, w[i].y, fmaf(xn[i].y, w[i].x, a1[i].y));
            if (both) {
                a0[i].x = fmaf(xc[i].x, w[i].x, fmaf(-xc[i].y, w[i].y, a0[i].x));
                a0[i].y = fmaf(xc[i].x, w[i].y, fmaf(xc[i].y, w[i].x, a0[i].y));
                xn[i] = xc[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = (int)threadIdx.x + FFT_THREADS * i;
        if (k < NB) { lds[k] = a0[i]; lds[NB + k] = a1[i]; }
    }
    __syncthreads();
    float2 v[R1];
#pragma unroll
    for (int a = 0; a < R1; ++a) {                           // Hermitian extension of both, combined as a0 + i a1, into the registers
        const int k = (int)threadIdx.x + F::M1 * a;
        v[a] = make_float2(0.f, 0.f);
        if (threadIdx.x < F::M1) {
            if (k <= N / 2) { const float2 u = lds[k], w = lds[NB + k]; v[a] = make_float2(u.x - w.y, u.y + w.x); }
            else { const float2 u = lds[N - k], w = lds[NB + N - k]; v[a] = make_float2(u.x + w.y, w.x - u.y); }
        }
    }
    __syncthreads();
    F::template run<true>(v, lds);
    const float inv = 1.0f / (float)N;
    float* yo = y + ((long)b * C + c) * T + (long)j0 * Bk;
    const int nvalid = min(2 * Bk, T - j0 * Bk);             // block j1 follows j0 in y
    for (int n = threadIdx.x; n < nvalid; n += FFT_THREADS) {
        const float2 z = lds[Bk + (n & (Bk - 1))];
        const float r = (n < Bk ? z.x : z.y) * inv;
        yo[n] = accumulate ? yo[n] + r : r;
    }
}

typedef Fft3<16, 16, 32> Fft8192;
typedef Fft3<8, 8, 16> Fft1024;

__global__ __launch_bounds__(256) void snr_partial_kernel(const float* __restrict__ s, const float* __restrict__ nz, const int* __restrict__ start,
                                                          const int* __restrict__ length, float* __restrict__ part, int C, int T) {
    const int bc = blockIdx.x, b = bc / C, ch = blockIdx.y;
    const int lo = max(start[b], 0), hi = min(start[b] + length[b], T);
    const int span = (T + MIX_CHUNKS - 1) / MIX_CHUNKS;
    const int t0 = max(lo, ch * span), t1 = min(hi, (ch + 1) * span);
    float ex = 0.f, en = 0.f;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        const float a = s[(long)bc * T + t], c = nz[(long)bc * T + t];
        ex = fmaf(a, a, ex); en = fmaf(c, c, en);
    }
    __shared__ float red[2][8];
    ex = wave_sum(ex); en = wave_sum(en);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ex; red[1][threadIdx.x >> 6] = en; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long)bc * MIX_CHUNKS + ch) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[((long)bc * MIX_CHUNKS + ch) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void snr_apply_kernel(const float* __restrict__ s, const float* __restrict__ nz, const float* __restrict__ snr,
                                                        const float* __restrict__ part, float* __restrict__ out, int C, int T) {
    const int bc = blockIdx.x, b = bc / C;
    float ex = 0.f, en = 0.f;
    for (int ch = 0; ch < MIX_CHUNKS; ++ch) {      // every workgroup folds the same partials in the same order
        ex += part[((long)bc * MIX_CHUNKS + ch) * 2 + 0];
        en += part[((long)bc * MIX_CHUNKS + ch) * 2 + 1];
    }
    const float a = sqrtf(ex / (en + 1e-9f) * exp10f(-snr[b] * 0.1f));
    const int span = (T + gridDim.y - 1) / gridDim.y;
    const int t0 = blockIdx.y * span, t1 = min(T, t0 + span);
    for (int t = t0 + threadIdx.x; t < t1; t += 256) out[(long)bc * T + t] = fmaf(a, nz[(long)bc * T + t], s[(long)bc * T + t]);
}

struct ConvPlan {
    int logn, N, Bk, NB, nb, P;
    int64_t x_bytes, h_bytes;
};

bool plan(const wj_rir_conv_args* a, ConvPlan& p) {
    if (a->B <= 0 || a->C <= 0 || a->T <= 0 || a->L <= 0) return false;
    p.N = a->fft_size ? a->fft_size : 8192;
    if (p.N != 1024 && p.N != 8192) return false;
    p.logn = p.N == 1024 ? 10 : 13;
    p.Bk = p.N / 2; p.NB = p.N / 2 + 1;
    p.nb = (a->T + p.Bk - 1) / p.Bk;
    p.P = (a->L + p.Bk - 1) / p.Bk;
    p.x_bytes = (int64_t)a->B * p.nb * p.NB * 8;
    p.h_bytes = (int64_t)a->B * a->C * p.P * p.NB * 8;
    return true;
}

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

}  // namespace

int64_t wj_rir_conv_ws_bytes(const wj_rir_conv_args* a) {
    ConvPlan p;
    if (!plan(a, p)) return -1;
    return p.x_bytes + p.h_bytes;
}

int64_t wj_snr_mix_ws_bytes(const wj_snr_mix_args* a) {
    if (a->B <= 0 || a->C <= 0) return -1;
    return (int64_t)a->B * a->C * MIX_CHUNKS * 2 * 4;
}

extern "C" int wj_rir_convolve(const wj_rir_conv_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    ConvPlan p;
    if (!a || !a->x || !a->h || !a->y || !a->workspace || !plan(a, p)) return WJ_ERR_ARG;
    if (a->h_stride_c < a->L || a->h_stride_b < (int64_t)(a->C - 1) * a->h_stride_c + a->L) return WJ_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float2* X = reinterpret_cast<float2*>(a->workspace);
    float2* H = reinterpret_cast<float2*>(reinterpret_cast<char*>(a->workspace) + p.x_bytes);
    constexpr int LDS8K = Fft8192::LDS_ELEMS * 8, LDS1K = Fft1024::LDS_ELEMS * 8;
    static int once = set_lds(scene_fwd_fft_kernel<Fft8192>, LDS8K) | set_lds(scene_mac_ifft_kernel<Fft8192>, LDS8K);
    (void)once;
    const dim3 g1(a->B * ((p.nb + 1) / 2) + a->B * a->C * ((p.P + 1) / 2)), g2(a->B * a->C * ((p.nb + 1) / 2)), blk(FFT_THREADS);
    if (p.logn == 13) {
        hipLaunchKernelGGL((scene_fwd_fft_kernel<Fft8192>), g1, blk, LDS8K, st, a->x, a->h, X, H, a->B, a->C, a->T, a->L,
                           (long)a->h_stride_b, (long)a->h_stride_c, p.nb, p.P);
        hipLaunchKernelGGL((scene_mac_ifft_kernel<Fft8192>), g2, blk, LDS8K, st, X, H, a->y, a->B, a->C, a->T, p.nb, p.P, a->accumulate);
    } else {
        hipLaunchKernelGGL((scene_fwd_fft_kernel<Fft1024>), g1, blk, LDS1K, st, a->x, a->h, X, H, a->B, a->C, a->T, a->L,
                           (long)a->h_stride_b, (long)a->h_stride_c, p.nb, p.P);
        hipLaunchKernelGGL((scene_mac_ifft_kernel<Fft1024>), g2, blk, LDS1K, st, X, H, a->y, a->B, a->C, a->T, p.nb, p.P, a->accumulate);
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_snr_mix(const wj_snr_mix_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->source || !a->noise || !a->out || !a->snr || !a->start || !a->length || !a->workspace) return WJ_ERR_ARG;
    if (a->B <= 0 || a->C <= 0 || a->T <= 0) return WJ_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(snr_partial_kernel, dim3(a->B * a->C, MIX_CHUNKS), dim3(256), 0, st, a->source, a->noise, a->start, a->length,
                       a->workspace, a->C, a->T);
    const int chunks = max(1, min(256, (a->T + 8191) / 8192));
    hipLaunchKernelGGL(snr_apply_kernel, dim3(a->B * a->C, chunks), dim3(256), 0, st, a->source, a->noise, a->snr, a->workspace, a->out,
                       a->C, a->T);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
