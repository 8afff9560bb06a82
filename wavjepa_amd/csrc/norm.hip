// LayerNorm forward/backward (fp32 statistics, fused post-norm residual add) and bf16 column sums for gfx950.
// HBM-bound kernels: one 64-lane wave owns a row, 16-B (fp32) / 8-B (bf16) vector accesses, wave-shuffle
// reductions; column (parameter-gradient) partials are combined in LDS before one global atomic per column per
// workgroup.
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int MAXV = 4;  // float4 chunks per 64-lane row: D <= 1024

// token row m -> row of a padded (seg rows per clip, `valid` of them tokens) conv buffer; chan = S > 1: the buffer is
// channel-major (clip index c*nclips + n) while token m = (n*S + c)*valid + t
__device__ __forceinline__ long remap_row(int m, int seg, int valid, int chan = 1, int nclips = 0) {
    if (seg <= 0) return (long)m;
    const int q = m / valid, t = m - q * valid;
    if (chan <= 1) return (long)q * seg + t;
    const int n = q / chan, c = q - n * chan;
    return ((long)c * nclips + n) * seg + t;
}

__device__ __forceinline__ f32x4 load4(const void* base, long row, int D, int col, bool is_bf16) {
    if (is_bf16) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>((const bf16_t*)base + row * D + col);
        return f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
    }
    return *reinterpret_cast<const f32x4*>((const float*)base + row * D + col);
}

// LPR lanes own a row (64, or 32 when D is an odd multiple of 128 -- D = 384 would leave a quarter of a 64-lane row idle),
// V float4 chunks per lane: D <= 4 V LPR.  A wave works on 64 / LPR rows at once.
template <int V, int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int GS_SPLIT = WJ_GROUP_STATS_SPLIT;   // workgroups per row group when group_stats is requested

template <int V, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(wj_ln_fwd_args a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR, sr = lane / LPR;
    const int D = a.D;
    const float invD = 1.0f / (float)D;
    f32x4 gam[V], bet[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int col = li * 4 + LPR * 4 * j;
        gam[j] = bet[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < D) {
            gam[j] = *reinterpret_cast<const f32x4*>(a.gamma + col);
            bet[j] = *reinterpret_cast<const f32x4*>(a.beta + col);
        }
    }
    // Row assignment.  Default: rows interleaved over the grid.  With group_stats: a workgroup owns a contiguous quarter of ONE
    // group of rows, keeps its (sum y, sum y^2) in registers and STORES the pair at the end; the consumer adds the quarters in
    // order (no float atomics: the teacher targets are bit-reproducible).
    int m_begin = (blockIdx.x * 4 + wave) * RPW + sr, m_end = a.M, m_step = gridDim.x * 4 * RPW;
    if (a.group_stats) {
        const int grp = blockIdx.x / GS_SPLIT, part = blockIdx.x - grp * GS_SPLIT;
        const int rpp = (a.group_rows + GS_SPLIT - 1) / GS_SPLIT;
        m_begin = grp * a.group_rows + part * rpp + wave * RPW + sr;
        m_end = min(a.M, grp * a.group_rows + min(a.group_rows, (part + 1) * rpp));
        m_step = 4 * RPW;
    }
    float gs1 = 0.f, gs2 = 0.f;
    for (int m = m_begin; m < m_end; m += m_step) {
        const long xr = remap_row(m, a.in_seg, a.in_valid, a.in_chan, a.in_chan > 1 ? a.M / (a.in_chan * a.in_valid) : 0);
        f32x4 s[V];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < D) {
                s[j] = load4(a.x, xr, D, col, a.x_is_bf16);
                if (a.r) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.r + (long)m * D + col);
                    s[j] += f32x4{bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
                }
                sum += s[j][0] + s[j][1] + s[j][2] + s[j][3];
            }
        }
        const float mean = row_sum<V, LPR>(sum) * invD;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            if (col < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = s[j][e] - mean;
                    sq += d * d;
                }
            }
        }
        const float var = row_sum<V, LPR>(sq) * invD;
        const float rstd = rsqrtf(var + a.eps);
        if (li == 0) {
            if (a.mean) a.mean[m] = mean;
            if (a.rstd) a.rstd[m] = rstd;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            if (col < D) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (s[j][e] - mean) * rstd * gam[j][e] + bet[j][e];
                    gs1 += y[e];
                    gs2 = fmaf(y[e], y[e], gs2);
                }
                if (a.y_f32) *reinterpret_cast<f32x4*>(a.y_f32 + (long)m * D + col) = y;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(y[e]);
                if (a.y_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.y_bf16 + (long)m * D + col) = o;
                if (a.y_fp8) {
                    // MX fp8 of bf16(y), as wj_quantize_mxfp8 defines it: a 32-column block = 8 consecutive lanes of this chunk, the
                    // four block scales of a 128-column K tile = 32 lanes -> one dword [kt][row] (D % 128 == 0: every lane of a
                    // block / K tile is live together, so the shuffles below are uniform)
                    float f[4], amax = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f[e] = bf2f(o[e]); amax = fmaxf(amax, fabsf(f[e])); }
                    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                    int sc = 0;
                    if (amax > 0.f) {
                        int ex;
                        const float mm = frexpf(amax * (1.0f / 448.0f), &ex);
                        sc = (mm == 0.5f) ? ex - 1 : ex;
                        sc = max(-127, min(127, sc));
                    }
                    const float inv = __builtin_amdgcn_ldexpf(1.0f, -sc);
                    unsigned w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w, true);
                    *reinterpret_cast<unsigned*>((unsigned char*)a.y_fp8 + (long)m * D + col) = w;
                    const unsigned sb = (unsigned)(sc + 127);
                    const int base = lane & ~31;
                    const unsigned s0 = __shfl(sb, base, 64), s1 = __shfl(sb, base + 8, 64), s2 = __shfl(sb, base + 16, 64),
                                   s3 = __shfl(sb, base + 24, 64);
                    if ((lane & 31) == 0)
                        ((uint32_t*)a.y_fp8_scales)[(long)(col / 128) * a.ld_fp8_scale + m] = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
                }
            }
        }
    }
    if (a.group_stats) {                     // kernel-uniform: fold lanes, then waves, then one pair of stores per workgroup
        __shared__ float gred[4][2];
        gs1 = wave_sum(gs1);
        gs2 = wave_sum(gs2);
        if (lane == 0) { gred[wave][0] = gs1; gred[wave][1] = gs2; }
        __syncthreads();
        if (threadIdx.x < 2) {
            float* gsp = a.group_stats + (long)blockIdx.x * 2;         // [group][GS_SPLIT][2]
            gsp[threadIdx.x] = gred[0][threadIdx.x] + gred[1][threadIdx.x] + gred[2][threadIdx.x] + gred[3][threadIdx.x];
        }
    }
}

// The LEAN form (wj_ln_fwd_args.workgroups > 0): the same arithmetic in the same order -- bit-identical outputs -- from a kernel that fits
// beside a persistent GEMM workgroup of ANOTHER stream on the same CU.  Such a workgroup (csrc/gemm_persist.hip) holds 2 x 224-232 of a
// SIMD's 512 VGPRs and 150 of the CU's 160 KB of LDS for the whole launch: what is left is one wave of <= 48 registers per SIMD and a few
// KB of LDS.  ln_fwd_kernel (78-100 VGPRs) therefore never runs beside it: the forward's LayerNorms -- HBM-bound, matrix pipe idle -- and
// the other stream's GEMMs -- matrix-bound, HBM mostly idle -- take turns on the chip.  This form keeps gamma / beta in memory (3 KB,
// L1-resident; re-read per row), has no fp8 / group-statistics / row-remap paths, and is launched with a grid capped by the caller (one
// workgroup per CU = one wave per SIMD): it takes the bandwidth the GEMM leaves idle and never the CU slots the next persistent launch needs.
template <int V, int LPR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void ln_fwd_lean_kernel(wj_ln_fwd_args a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR, sr = lane / LPR;
    const int D = a.D;
    const float invD = 1.0f / (float)D;
    for (int m = (blockIdx.x * 4 + wave) * RPW + sr; m < a.M; m += gridDim.x * 4 * RPW) {
        f32x4 s[V];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < D) {
                s[j] = load4(a.x, (long)m, D, col, a.x_is_bf16);
                if (a.r) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.r + (long)m * D + col);
                    s[j] += f32x4{bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
                }
                sum += s[j][0] + s[j][1] + s[j][2] + s[j][3];
            }
        }
        const float mean = row_sum<V, LPR>(sum) * invD;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            if (col < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = s[j][e] - mean;
                    sq += d * d;
                }
            }
        }
        const float var = row_sum<V, LPR>(sq) * invD;
        const float rstd = rsqrtf(var + a.eps);
        if (li == 0) {
            if (a.mean) a.mean[m] = mean;
            if (a.rstd) a.rstd[m] = rstd;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            if (col < D) {
                const f32x4 gam = *reinterpret_cast<const f32x4*>(a.gamma + col);
                const f32x4 bet = *reinterpret_cast<const f32x4*>(a.beta + col);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (s[j][e] - mean) * rstd * gam[e] + bet[e];
                if (a.y_f32) *reinterpret_cast<f32x4*>(a.y_f32 + (long)m * D + col) = y;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(y[e]);
                if (a.y_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.y_bf16 + (long)m * D + col) = o;
            }
        }
    }
}

constexpr int BWD_THREADS = 256;

// LPR lanes per row, TWO row slots in flight per wave (independent load streams ahead of the shuffle reductions), V float4
// chunks per lane (D <= 4 V LPR).  Column partials (dgamma, dbeta, dbias) stay in registers across the row loop and are
// combined through LDS atomics, then one global atomic per column per workgroup (or a workspace row, folded afterwards).
template <int V, int LPR>
__global__ __launch_bounds__(BWD_THREADS) void ln_bwd_kernel(wj_ln_bwd_args a) {
    constexpr int RPW = 64 / LPR;
    constexpr int CW = LPR * 4 * V;                 // columns covered (>= D)
    constexpr int nw = BWD_THREADS / 64;
    __shared__ float cacc[nw][3][CW];               // per-wave column partials, added in wave order below
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR, sr = lane / LPR;
    const int D = a.D;
    const float invD = 1.0f / (float)D;

    f32x4 dg[V], db[V], dbi[V], gam[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        dg[j] = db[j] = dbi[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int col = li * 4 + LPR * 4 * j;
        gam[j] = col < D ? *reinterpret_cast<const f32x4*>(a.gamma + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int stride = gridDim.x * nw * RPW;
    for (int m0 = (blockIdx.x * nw + wave) * RPW + sr; m0 < a.M; m0 += 2 * stride) {
        int mrow[2] = {m0, m0 + stride};
        f32x4 xh[2][V], dy[2][V];
        float mean[2], rstd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool live = mrow[u] < a.M;
            const int m = live ? mrow[u] : m0;
            const long xr = remap_row(m, a.in_seg, a.in_valid, a.chan, a.chan > 1 ? a.M / (a.chan * a.in_valid) : 0);
            mean[u] = a.mean[m];
            rstd[u] = a.rstd[m];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int col = li * 4 + LPR * 4 * j;
                xh[u][j] = dy[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (col < D) {
                    f32x4 sx = load4(a.x, xr, D, col, a.x_is_bf16);
                    if (a.r) {
                        const bf16x4 r = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.r + (long)m * D + col);
                        sx += f32x4{bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
                    }
                    f32x4 d = *reinterpret_cast<const f32x4*>(a.dy + (long)m * D + col);
                    if (a.dy2) {
                        if (a.dy2_is_bf16) {
                            const bf16x4 e2 = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.dy2 + (long)m * D + col);
                            d += f32x4{bf2f(e2[0]), bf2f(e2[1]), bf2f(e2[2]), bf2f(e2[3])};
                        } else {
                            d += *reinterpret_cast<const f32x4*>((const float*)a.dy2 + (long)m * D + col);
                        }
                    }
                    xh[u][j] = sx;
                    dy[u][j] = d;
                }
            }
        }
        float c1[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int col = li * 4 + LPR * 4 * j;
                if (col < D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[u][j][e] = (xh[u][j][e] - mean[u]) * rstd[u];
                        const float g = dy[u][j][e] * gam[j][e];
                        c1[u] += g;
                        c2[u] += g * xh[u][j][e];
                    }
                }
            }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {      // four reductions interleaved
            c1[0] += __shfl_xor(c1[0], o, 64); c2[0] += __shfl_xor(c2[0], o, 64);
            c1[1] += __shfl_xor(c1[1], o, 64); c2[1] += __shfl_xor(c2[1], o, 64);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (mrow[u] >= a.M) continue;
            const int m = mrow[u];
            const float k1 = c1[u] * invD, k2 = c2[u] * invD;
            const long orow = remap_row(m, a.out_seg, a.out_valid, a.chan, a.chan > 1 ? a.M / (a.chan * a.out_valid) : 0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int col = li * 4 + LPR * 4 * j;
                if (col < D) {
                    f32x4 ds;
                    bf16x4 dsb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ds[e] = rstd[u] * (dy[u][j][e] * gam[j][e] - k1 - xh[u][j][e] * k2);
                        dsb[e] = f2bf(ds[e]);
                        dg[j][e] += dy[u][j][e] * xh[u][j][e];
                        db[j][e] += dy[u][j][e];
                        dbi[j][e] += bf2f(dsb[e]);
                    }
                    if (a.ds_f32) *reinterpret_cast<f32x4*>(a.ds_f32 + (long)m * D + col) = ds;
                    if (a.ds_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.ds_bf16 + orow * D + col) = dsb;
                }
            }
        }
    }
    if constexpr (RPW == 2) {                       // fold the two sub-rows of the wave before touching LDS
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dg[j][e] += __shfl_xor(dg[j][e], 32, 64);
                db[j][e] += __shfl_xor(db[j][e], 32, 64);
                dbi[j][e] += __shfl_xor(dbi[j][e], 32, 64);
            }
    }
    if (sr == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int col = li * 4 + LPR * 4 * j;
            if (col < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cacc[wave][0][col + e] = dg[j][e];
                    cacc[wave][1][col + e] = db[j][e];
                    cacc[wave][2][col + e] = dbi[j][e];
                }
            }
        }
    }
    __syncthreads();
    if (a.workspace) {   // plain coalesced stores of this workgroup's partials; a second kernel folds them
        float* ws = a.workspace + (long)blockIdx.x * 3 * D;
        for (int c = threadIdx.x; c < 3 * D; c += BWD_THREADS) {
            const int w = c / D, cc = c % D;
            ws[c] = cacc[0][w][cc] + cacc[1][w][cc] + cacc[2][w][cc] + cacc[3][w][cc];
        }
        return;
    }
    for (int c = threadIdx.x; c < D; c += BWD_THREADS) {
        if (a.dgamma) atomicAdd(a.dgamma + c, cacc[0][0][c] + cacc[1][0][c] + cacc[2][0][c] + cacc[3][0][c]);
        if (a.dbeta) atomicAdd(a.dbeta + c, cacc[0][1][c] + cacc[1][1][c] + cacc[2][1][c] + cacc[3][1][c]);
        if (a.dbias) atomicAdd(a.dbias + c, cacc[0][2][c] + cacc[1][2][c] + cacc[2][2][c] + cacc[3][2][c]);
    }
}

// out[c] += sum over rows of an f32 matrix: workgroup = 128 columns x a row range (float4 per thread, 8 row lanes)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, long ldx, int M, int N,
                                                         float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2,
                                                         int n_each, int rows_per_wg) {
    __shared__ float red[8][132];
    const int t = threadIdx.x, cc = t & 31, rl = t >> 5;
    const int col = blockIdx.x * 128 + cc * 4;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (col < N)
        for (int r = r0 + rl; r < r1; r += 8) acc += *reinterpret_cast<const f32x4*>(x + (long)r * ldx + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cc * 4 + e] = acc[e];
    __syncthreads();
    if (t < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][t];
        const int c = blockIdx.x * 128 + t;
        if (c < N) {
            // columns [0, n_each) -> o0, [n_each, 2 n_each) -> o1, ... (LayerNorm's three gradients share one launch)
            const int which = c / n_each, cc2 = c - which * n_each;
            float* o = which == 0 ? o0 : (which == 1 ? o1 : o2);
            if (o) atomicAdd(o + cc2, s);
        }
    }
}

void launch_colsum_f32(const float* x, long ldx, int M, int N, float* o0, float* o1, float* o2, int n_each, hipStream_t s) {
    const int gx = (N + 127) / 128;
    int gy = 1024 / gx;               // ~1000 workgroups: a thread sums only a handful of rows (latency-bound otherwise)
    if (gy < 1) gy = 1;
    if (gy > (M + 7) / 8) gy = (M + 7) / 8;
    int rows = (M + gy - 1) / gy;
    rows = (rows + 7) / 8 * 8;
    gy = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(gx, gy), dim3(256), 0, s, x, ldx, M, N, o0, o1, o2, n_each, rows);
}

// The same fold for up to WJ_COLSUM_GROUP_MAX matrices in ONE launch: the per-workgroup partials that the LayerNorm / attention backward
// kernels of a few layers left in their own scratch rows (wj_colsum_f32_group).  An item gets GROUP_CB x GROUP_RB workgroup slots
// (column blocks of 128 x row ranges); slots beyond its width return at once.
constexpr int GROUP_CB = 18, GROUP_RB = 8;      // up to 2304 columns (3 x 768), 8 row ranges
__global__ __launch_bounds__(256) void colsum_f32_group_kernel(wj_colsum_group_args a) {
    __shared__ float red[8][132];
    const int item = blockIdx.x / (GROUP_CB * GROUP_RB), rem = blockIdx.x - item * (GROUP_CB * GROUP_RB);
    const int cb = rem % GROUP_CB, rb = rem / GROUP_CB;
    const int M = a.M[item], N = a.N[item];
    if (cb * 128 >= N) return;
    const float* x = a.x[item];
    const long ldx = a.ldx[item];
    int rows = (M + GROUP_RB - 1) / GROUP_RB;
    rows = (rows + 7) / 8 * 8;
    const int t = threadIdx.x, cc = t & 31, rl = t >> 5;
    const int col = cb * 128 + cc * 4;
    const int r0 = rb * rows, r1 = min(M, r0 + rows);
    if (r0 >= M) return;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (col < N)
        for (int r = r0 + rl; r < r1; r += 8) acc += *reinterpret_cast<const f32x4*>(x + (long)r * ldx + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cc * 4 + e] = acc[e];
    __syncthreads();
    if (t < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][t];
        const int c = cb * 128 + t;
        if (c < N) {
            const int n_each = a.n_each[item];
            const int which = c / n_each, cc2 = c - which * n_each;
            float* o = which == 0 ? a.o0[item] : (which == 1 ? a.o1[item] : a.o2[item]);
            if (o) atomicAdd(o + cc2, s);
        }
    }
}

// column sums: workgroup = 64 columns x a row range; thread (cc = t&7 -> 8 columns, rl = t>>3 -> row lane of 32)
__global__ __launch_bounds__(256) void colsum_kernel(wj_colsum_args a, int rows_per_wg) {
    __shared__ float red[32][65];
    const int t = threadIdx.x, cc = t & 7, rl = t >> 3;
    const int col = blockIdx.x * 64 + cc * 8;
    const int r0 = blockIdx.y * rows_per_wg;
    const int r1 = min(a.M, r0 + rows_per_wg);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < a.N) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>((const bf16_t*)a.x + (long)r * a.ldx + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cc * 8 + e] = acc[e];
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) s += red[r][t];
        const int c = blockIdx.x * 64 + t;
        if (c < a.N) atomicAdd(a.out + c, s);
    }
}

}  // namespace

extern "C" int wj_layernorm_fwd(const wj_ln_fwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->gamma || !a->beta) return WJ_ERR_ARG;
    if (a->M <= 0 || a->D <= 0 || (a->D & 3) || a->D > 256 * MAXV) return WJ_ERR_ARG;
    if (a->in_seg > 0 && a->in_valid <= 0) return WJ_ERR_ARG;
    if (a->in_chan > 1 && (a->in_seg <= 0 || a->M % (a->in_chan * a->in_valid))) return WJ_ERR_ARG;
    if (a->group_stats && a->group_rows <= 0) return WJ_ERR_ARG;
    if (a->y_fp8 && (!a->y_fp8_scales || (a->D % 128) || a->ld_fp8_scale < a->M)) return WJ_ERR_ARG;
    const bool half = (a->D % 128 == 0) && (a->D % 256 != 0) && a->D <= 384;   // 128 / 384: 32 lanes per row
    const int rpw = half ? 2 : 1;
    int grid = (a->M + 4 * rpw - 1) / (4 * rpw);
    if (grid > 8192) grid = 8192;
    if (a->group_stats) grid = ((a->M + a->group_rows - 1) / a->group_rows) * GS_SPLIT;
    dim3 g(grid), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (a->workgroups > 0 && !a->group_stats && !a->y_fp8 && a->in_seg <= 0) {
        // the lean form on a capped grid (see ln_fwd_lean_kernel): same bits, a kernel that shares a CU with a persistent GEMM
        dim3 gl(grid < a->workgroups ? grid : a->workgroups);
        if (half) {
            if (a->D == 128) hipLaunchKernelGGL((ln_fwd_lean_kernel<1, 32>), gl, b, 0, s, *a);
            else hipLaunchKernelGGL((ln_fwd_lean_kernel<3, 32>), gl, b, 0, s, *a);
        } else {
            switch ((a->D + 255) / 256) {
                case 1: hipLaunchKernelGGL((ln_fwd_lean_kernel<1, 64>), gl, b, 0, s, *a); break;
                case 2: hipLaunchKernelGGL((ln_fwd_lean_kernel<2, 64>), gl, b, 0, s, *a); break;
                case 3: hipLaunchKernelGGL((ln_fwd_lean_kernel<3, 64>), gl, b, 0, s, *a); break;
                default: hipLaunchKernelGGL((ln_fwd_lean_kernel<4, 64>), gl, b, 0, s, *a); break;
            }
        }
        WJ_CHECK_LAUNCH();
        return WJ_OK;
    }
    if (half) {
        if (a->D == 128) hipLaunchKernelGGL((ln_fwd_kernel<1, 32>), g, b, 0, s, *a);
        else hipLaunchKernelGGL((ln_fwd_kernel<3, 32>), g, b, 0, s, *a);
    } else {
        switch ((a->D + 255) / 256) {
            case 1: hipLaunchKernelGGL((ln_fwd_kernel<1, 64>), g, b, 0, s, *a); break;
            case 2: hipLaunchKernelGGL((ln_fwd_kernel<2, 64>), g, b, 0, s, *a); break;
            case 3: hipLaunchKernelGGL((ln_fwd_kernel<3, 64>), g, b, 0, s, *a); break;
            default: hipLaunchKernelGGL((ln_fwd_kernel<4, 64>), g, b, 0, s, *a); break;
        }
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

// partial rows wj_layernorm_bwd leaves in its workspace ([rows][3][D]) for M token rows of width D: its grid
static int ln_bwd_one_pass_rows() {   // WJ_LN_BWD_ONE_PASS_ROWS: launches of at most this many row slots give every wave ONE pass (0 = never)
    static const int v = wj_lab_env_int("WJ_LN_BWD_ONE_PASS_ROWS", 16384);
    return v;
}
static int ln_bwd_grid(int M, int D) {
    const int nw = BWD_THREADS / 64;
    const bool half = (D % 128 == 0) && (D % 256 != 0) && D <= 384;
    const int rpw = half ? 2 : 1;
    // Large M: >= 8 rows per wave (4 passes of its two row slots) -- the per-workgroup epilogue (LDS fold, partials store) is paid once
    // per 32 rows and the chip is full anyway.  Small M (the ragged student's 10 k rows are 39 rows per CU): 314 such workgroups put
    // five waves on a CU, each running its four load -> reduce -> store round trips one after the other (51 us for 139 MB cold); with ONE
    // pass per wave there are 1256 workgroups, three resident per CU (155 VGPRs), and the launch takes 36 us (tools/ln_bench.py; 6 / 8
    // / 12-wave workgroups of one pass: 53 / 44 / 36 us -- what counts is how many waves of 155 VGPRs a CU holds, 12, and that they do
    // not all sit in the same phase).
    const int passes = (half ? (M + 1) / 2 : M) <= ln_bwd_one_pass_rows() ? 1 : 4;
    int grid = (M + 2 * passes * nw * rpw - 1) / (2 * passes * nw * rpw);
    return grid > 1536 ? 1536 : grid;
}
extern "C" int wj_ln_bwd_partial_rows(int M, int D) {
    if (M <= 0 || D <= 0) return -1;
    return ln_bwd_grid(M, D);
}

extern "C" int wj_colsum_f32_group(const wj_colsum_group_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || a->n < 1 || a->n > WJ_COLSUM_GROUP_MAX) return WJ_ERR_ARG;
    for (int x = 0; x < a->n; ++x) {
        if (!a->x[x] || a->M[x] <= 0 || a->N[x] <= 0 || (a->N[x] & 3) || (a->ldx[x] & 3) || a->n_each[x] <= 0) return WJ_ERR_ARG;
        if (a->N[x] > GROUP_CB * 128 || a->N[x] > 3 * a->n_each[x]) return WJ_ERR_ARG;
    }
    hipLaunchKernelGGL(colsum_f32_group_kernel, dim3(a->n * GROUP_CB * GROUP_RB), dim3(256), 0, (hipStream_t)stream, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_layernorm_bwd(const wj_ln_bwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->dy || !a->x || !a->gamma || !a->mean || !a->rstd) return WJ_ERR_ARG;
    if (a->M <= 0 || a->D <= 0 || (a->D & 3) || a->D > 256 * MAXV) return WJ_ERR_ARG;
    if ((a->in_seg > 0 && a->in_valid <= 0) || (a->out_seg > 0 && a->out_valid <= 0)) return WJ_ERR_ARG;
    if (a->chan > 1 && ((a->in_seg > 0 && a->M % (a->chan * a->in_valid)) || (a->out_seg > 0 && a->M % (a->chan * a->out_valid)))) return WJ_ERR_ARG;
    const bool half = (a->D % 128 == 0) && (a->D % 256 != 0) && a->D <= 384;
    const int grid = ln_bwd_grid(a->M, a->D);
    dim3 g(grid), b(BWD_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (half) {
        if (a->D == 128) hipLaunchKernelGGL((ln_bwd_kernel<1, 32>), g, b, 0, s, *a);
        else hipLaunchKernelGGL((ln_bwd_kernel<3, 32>), g, b, 0, s, *a);
    } else {
        switch ((a->D + 255) / 256) {
            case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, 64>), g, b, 0, s, *a); break;
            case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, 64>), g, b, 0, s, *a); break;
            case 3: hipLaunchKernelGGL((ln_bwd_kernel<3, 64>), g, b, 0, s, *a); break;
            default: hipLaunchKernelGGL((ln_bwd_kernel<4, 64>), g, b, 0, s, *a); break;
        }
    }
    if (a->workspace && (a->dgamma || a->dbeta || a->dbias))
        launch_colsum_f32(a->workspace, 3L * a->D, grid, 3 * a->D, a->dgamma, a->dbeta, a->dbias, a->D, (hipStream_t)stream);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_colsum_f32(const wj_colsum_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->out || a->M <= 0 || a->N <= 0 || (a->N & 3) || (a->ldx & 3)) return WJ_ERR_ARG;
    launch_colsum_f32((const float*)a->x, a->ldx, a->M, a->N, a->out, nullptr, nullptr, a->N, (hipStream_t)stream);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_colsum_bf16(const wj_colsum_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->out || a->M <= 0 || a->N <= 0 || (a->N & 7) || (a->ldx & 7)) return WJ_ERR_ARG;
    const int gx = (a->N + 63) / 64;
    int gy = 2048 / gx;
    if (gy < 1) gy = 1;
    int rows = (a->M + gy - 1) / gy;
    rows = (rows + 31) / 32 * 32;
    gy = (a->M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *a, rows);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
