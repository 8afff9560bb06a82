// HBM-bound plumbing kernels of the WavJEPA step for gfx950: token gather/scatter, positions, conv weight layouts,
// GELU backward, teacher targets (joint instance norm), masked MSE, EMA, AdamW (+ global-norm clip), casts and the
// crop/normalise batch preparation.  All are vectorised (8-16 B per lane) streaming kernels or small reductions.
#include <string.h>
#include "common.h"
#include "../../include/wavjepa_hip.h"
#ifdef WJ_LAB
#include "../../include/wavjepa_hip_lab.h"
#endif

namespace {

inline int grid_for(long n_items, int per_block, int cap = 8192) {
    long g = (n_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ------------------------------------------------------------------------------------------- GELU backward (bf16)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dpost, const bf16_t* __restrict__ pre,
                                                       bf16_t* __restrict__ dpre, long n8) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dpost + i * 8);
        const bf16x8 p = *reinterpret_cast<const bf16x8*>(pre + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(d[e]) * gelu_grad_f(bf2f(p[e])));
        *reinterpret_cast<bf16x8*>(dpre + i * 8) = o;
    }
}

// listed rows only (sparse conv backward); the consumed dpost rows are zeroed on the way (clear_dpost) so that the buffer
// is all-zero again for the next step
__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(bf16_t* __restrict__ dpost, const bf16_t* __restrict__ pre,
                                                            bf16_t* __restrict__ dpre, const int32_t* __restrict__ rows,
                                                            int n_rows, int row8, int clear_dpost) {
    const long n8 = (long)n_rows * row8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const int r = (int)(i / row8), c = (int)(i - (long)r * row8);
        const long off = ((long)rows[r] * row8 + c) * 8;
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dpost + off);
        const bf16x8 p = *reinterpret_cast<const bf16x8*>(pre + off);
        bf16x8 o, z;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = f2bf(bf2f(d[e]) * gelu_grad_f(bf2f(p[e]))); z[e] = f2bf(0.f); }
        *reinterpret_cast<bf16x8*>(dpre + off) = o;
        if (clear_dpost) *reinterpret_cast<bf16x8*>(dpost + off) = z;
    }
}

__global__ __launch_bounds__(256) void zero_rows_kernel(char* __restrict__ buf, const int32_t* __restrict__ rows, int n_rows,
                                                        int row16) {
    const long n = (long)n_rows * row16;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / row16), c = (int)(i - (long)r * row16);
        *reinterpret_cast<uint4*>(buf + ((long)rows[r] * row16 + c) * 16) = make_uint4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------- conv weight layouts
__global__ __launch_bounds__(256) void conv_w_kernel(wj_conv_w_args a) {
    const int Co = a.C_out, Ci = a.C_in, k = a.k;
    if (a.mode == 0) {  // wp[o][kk*Ci + c] = w[o][c][kk]
        const long n = (long)Co * Ci * k;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const int c = (int)(i % Ci);
            const int kk = (int)((i / Ci) % k);
            const int o = (int)(i / ((long)Ci * k));
            ((bf16_t*)a.dst)[i] = f2bf(((const float*)a.src)[((long)o * Ci + c) * k + kk]);
        }
    } else if (a.mode == 1) {  // wd[v*Co + o][c] = w[o][c][rho + stride*(U-1-v)]
        const long n = (long)a.U * Co * Ci;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const int c = (int)(i % Ci);
            const int o = (int)((i / Ci) % Co);
            const int v = (int)(i / ((long)Ci * Co));
            const int kk = a.rho + a.stride * (a.U - 1 - v);
            ((bf16_t*)a.dst)[i] = f2bf(((const float*)a.src)[((long)o * Ci + c) * k + kk]);
        }
    } else {  // dw[o][c][kk] += dwp[o][kk*Ci + c]
        const long n = (long)Co * Ci * k;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const int kk = (int)(i % k);
            const int c = (int)((i / k) % Ci);
            const int o = (int)(i / ((long)Ci * k));
            ((float*)a.dst)[i] += ((const float*)a.src)[(long)o * Ci * k + (long)kk * Ci + c];
        }
    }
}

// ------------------------------------------------------------------------------------------- + positions
__global__ __launch_bounds__(256) void add_pos_kernel(wj_add_pos_args a) {
    const int D4 = a.D / 4;
    const long n = (long)a.M * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int m = (int)(i / D4), c = (int)(i - (long)m * D4) * 4;
        const bf16x4 x = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.x + (long)m * a.D + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.pos + (long)(m % a.T) * a.D + c);
        f32x4 y;
        bf16x4 yb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] = bf2f(x[e]) + p[e]; yb[e] = f2bf(y[e]); }
        if (a.y_f32) *reinterpret_cast<f32x4*>(a.y_f32 + (long)m * a.D + c) = y;
        if (a.y_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.y_bf16 + (long)m * a.D + c) = yb;
    }
}

// ------------------------------------------------------------------------------------------- mask gather (bit-exact copy)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ x, const int32_t* __restrict__ idx,
                                                          char* __restrict__ out, int n_rows, int row_bytes) {
    const int chunks = row_bytes / 16;
    for (int j = blockIdx.x; j < n_rows; j += gridDim.x) {
        const char* s = x + (long)idx[j] * row_bytes;
        char* d = out + (long)j * row_bytes;
        for (int c = threadIdx.x; c < chunks; c += 256)
            *reinterpret_cast<uint4*>(d + c * 16) = *reinterpret_cast<const uint4*>(s + c * 16);
    }
}

// ------------------------------------------------------------------------------------------- predictor input
__global__ __launch_bounds__(256) void scatter_fill_kernel(wj_scatter_fill_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D, M = a.B * a.T;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const int b = m / a.T, t = m - b * a.T;
        const int src = a.inv[m];
        for (int c = lane * 4; c < D; c += 256) {
            f32x4 tok;
            if (src >= 0) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.ctx_feats + (long)src * D + c);
                tok = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
            } else {
                const f32x4 mt = *reinterpret_cast<const f32x4*>(a.mask_token + c);
                // mask_token.repeat(...).type_as(bf16 features): the token is rounded to bf16 first
                tok = f32x4{bf2f(f2bf(mt[0])), bf2f(f2bf(mt[1])), bf2f(f2bf(mt[2])), bf2f(f2bf(mt[3]))};
            }
            const f32x4 y = tok + *reinterpret_cast<const f32x4*>(a.pos + (long)t * D + c);
            bf16x4 yb;
#pragma unroll
            for (int e = 0; e < 4; ++e) yb[e] = f2bf(y[e]);
            for (int g = 0; g < a.G; ++g) {
                const long orow = ((long)b * a.G + g) * a.T + t;
                if (a.out_f32) *reinterpret_cast<f32x4*>(a.out_f32 + orow * D + c) = y;
                if (a.out_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.out_bf16 + orow * D + c) = yb;
            }
        }
    }
}

// ragged form: one packed output row per listed dense index (b*G+g)*T + t
__global__ __launch_bounds__(256) void scatter_fill_rows_kernel(wj_scatter_fill_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    for (int r = blockIdx.x * 4 + wave; r < a.n_rows; r += gridDim.x * 4) {
        const int dense = a.rows[r];
        const int bg = dense / a.T, t = dense - bg * a.T, b = bg / a.G;
        const int src = a.inv[b * a.T + t];
        for (int c = lane * 4; c < D; c += 256) {
            f32x4 tok;
            if (src >= 0) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.ctx_feats + (long)src * D + c);
                tok = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
            } else {
                const f32x4 mt = *reinterpret_cast<const f32x4*>(a.mask_token + c);
                tok = f32x4{bf2f(f2bf(mt[0])), bf2f(f2bf(mt[1])), bf2f(f2bf(mt[2])), bf2f(f2bf(mt[3]))};
            }
            const f32x4 y = tok + *reinterpret_cast<const f32x4*>(a.pos + (long)t * D + c);
            bf16x4 yb;
#pragma unroll
            for (int e = 0; e < 4; ++e) yb[e] = f2bf(y[e]);
            if (a.out_f32) *reinterpret_cast<f32x4*>(a.out_f32 + (long)r * D + c) = y;
            if (a.out_bf16) *reinterpret_cast<bf16x4*>((bf16_t*)a.out_bf16 + (long)r * D + c) = yb;
        }
    }
}

// Backward.  Round 4's form (64 rows per workgroup, a wave walking 16 rows one after the other: index load -> row-map load -> data
// load, three dependent round trips per row, then 384 global float atomics per workgroup into the same 384 addresses) ran at
// 1.4 TB/s.  Now a wave owns SFB_WROWS rows whose indices (inv, and the G row-map entries of each) are fetched up front by one load per
// lane and handed round with readlane; the G x SFB_WROWS data loads then have no address dependence on each other.  The mask-token
// gradient of a workgroup goes to a row of `partials` (no atomics; folded by wj_colsum_f32_group) when the caller gives one.
constexpr int SFB_WROWS = 4;                 // rows per wave
constexpr int SFB_ROWS = 4 * SFB_WROWS;      // rows per workgroup
// JMAX: 256-float column blocks a lane may own (2: D <= 512 -- half the registers, so that two waves per SIMD fit beside a weight-gradient
// workgroup of the other stream, which is what this kernel runs against on the main stream's critical path; 4: D <= 1024)
template <int JMAX>
__global__ __launch_bounds__(256) void scatter_fill_bwd_kernel(wj_scatter_fill_bwd_args a) {
    // [4 waves][D] floats, sized at launch: a fixed [4][1024] (16 KB) left room for two of these workgroups beside a weight-gradient
    // workgroup of the other stream (128 KB of a CU's 160) and the kernel, on the main stream's critical path, ran 405 us instead of 90
    extern __shared__ float macc_dyn[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = a.D, M = a.B * a.T, G = a.G;
    const int m0 = blockIdx.x * SFB_ROWS + wave * SFB_WROWS;
    // lane l < SFB_WROWS * (G + 1): row r = l / (G + 1), slot u = l % (G + 1): u == G -> inv[m], else the (packed) input row of group u
    int idx = -1;
    {
        const int r = lane / (G + 1), u = lane - r * (G + 1), m = m0 + r;
        if (r < SFB_WROWS && m < M) {
            if (u == G) {
                idx = a.inv[m];
            } else {
                const int b = m / a.T, t = m - b * a.T;
                const int row = (b * G + u) * a.T + t;
                idx = a.rowmap ? a.rowmap[row] : row;          // ragged: packed row of this token, -1 = not visible
            }
        }
    }
    f32x4 mt[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) mt[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < SFB_WROWS; ++r) {
        if (m0 + r >= M) break;
        const int dst = __builtin_amdgcn_readlane(idx, r * (G + 1) + G);
        f32x4 s[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // four groups at a time, branch-free: every load of the batch is issued before the first add (with a `continue` on invisible
        // rows the loads of successive groups were separated by control flow -- one memory round trip per group and row)
        for (int g0 = 0; g0 < G; g0 += 4) {
            f32x4 v[4][JMAX];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u;
                const int row = g < G ? __builtin_amdgcn_readlane(idx, r * (G + 1) + (g < G ? g : 0)) : -1;
                ok[u] = row >= 0;
                const float* src = a.d_in + (long)(ok[u] ? row : 0) * D;
#pragma unroll
                for (int j = 0; j < JMAX; ++j) {
                    const int c = lane * 4 + 256 * j;
                    v[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (c < D) v[u][j] = *reinterpret_cast<const f32x4*>(src + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < JMAX; ++j)
                    if (ok[u]) s[j] += v[u][j];
        }
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < D) {
                if (dst >= 0) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf(s[j][e]);
                    *reinterpret_cast<bf16x4*>((bf16_t*)a.d_ctx_feats + (long)dst * D + c) = o;
                } else {
                    mt[j] += s[j];
                }
            }
        }
    }
    if (!a.d_mask_token) return;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) *reinterpret_cast<f32x4*>(&macc_dyn[wave * D + c]) = mt[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        const float v = (macc_dyn[c] + macc_dyn[D + c]) + (macc_dyn[2 * D + c] + macc_dyn[3 * D + c]);
        if (a.partials) a.partials[(long)blockIdx.x * D + c] = v;
        else atomicAdd(a.d_mask_token + c, v);
    }
}

// dst[m] = inv[m] >= 0 ? src[inv[m]] : 0     (gradient of the mask gather, jepa.py:399; inv == NULL: identity)
__global__ __launch_bounds__(256) void unmask_rows_kernel(wj_unmask_rows_args a) {
    const int D4 = a.D / 4;
    const long n = (long)a.M * D4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int m = (int)(i / D4), c = (int)(i - (long)m * D4) * 4;
        const int src = a.inv ? a.inv[m] : m;
        f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
        if (src >= 0) {
            if (a.src_is_f32) {
                y = *reinterpret_cast<const f32x4*>((const float*)a.src + (long)src * a.D + c);
            } else {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.src + (long)src * a.D + c);
                y = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
            }
        }
        if (a.dst_is_bf16) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(y[e]);
            *reinterpret_cast<bf16x4*>((bf16_t*)a.dst + (long)m * a.D + c) = o;
        } else {
            *reinterpret_cast<f32x4*>((float*)a.dst + (long)m * a.D + c) = y;
        }
    }
}

// ------------------------------------------------------------------------------------------- teacher targets
__global__ __launch_bounds__(1024) void instnorm_kernel(wj_instnorm_args a) {
    __shared__ float red[16];
    const long base = (long)blockIdx.x * a.TD;
    const float* x = a.x + base;
    const int n4 = a.TD / 4;
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        s += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = block_sum(s, red) / (float)a.TD;
    float q = 0.f;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
    }
    const float var = block_sum(q, red) / (float)a.TD;
    const float k = rsqrtf(var + a.eps) * a.scale;
    float* t = a.targets + base;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * k;
        if (a.accumulate) o += *reinterpret_cast<const f32x4*>(t + i * 4);
        *reinterpret_cast<f32x4*>(t + i * 4) = o;
    }
}

// one pass over the K kept layer outputs: block = 1024 x float4 of one sample
__global__ __launch_bounds__(256) void instnorm_mean_kernel(wj_instnorm_mean_args a) {
    const int b = blockIdx.y;
    const float* xs[8] = {a.x0, a.x1, a.x2, a.x3, a.x4, a.x5, a.x6, a.x7};
    const float invn = 1.0f / (float)a.TD, invk = 1.0f / (float)a.K;
    float mu[8], sc[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        mu[l] = sc[l] = 0.f;
        if (l < a.K) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int p = 0; p < WJ_GROUP_STATS_SPLIT; ++p) {       // fixed order: bit-reproducible targets
                s1 += a.stats[(((long)l * a.B + b) * WJ_GROUP_STATS_SPLIT + p) * 2];
                s2 += a.stats[(((long)l * a.B + b) * WJ_GROUP_STATS_SPLIT + p) * 2 + 1];
            }
            mu[l] = s1 * invn;
            sc[l] = rsqrtf(fmaxf(s2 * invn - mu[l] * mu[l], 0.f) + a.eps) * invk;
        }
    }
    const long base = (long)b * a.TD;
    const int n4 = a.TD / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            if (l < a.K) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs[l] + base + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e] - mu[l], sc[l], o[e]);
            }
        }
        *reinterpret_cast<f32x4*>(a.targets + base + i * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------- masked MSE
__global__ __launch_bounds__(1024) void mse_count_kernel(const uint8_t* __restrict__ tgt, float* __restrict__ ws, long n) {
    __shared__ float red[16];
    // one workgroup, 16 mask bytes per load (a byte per load made this 88 us of pure latency); the count is an integer below
    // 2^24, so the float sum is exact in any order
    int k = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(tgt) & 15) == 0;
    const long nv = vec ? n >> 4 : 0;
    for (long i = threadIdx.x; i < nv; i += 1024) {
        const uint4 v = reinterpret_cast<const uint4*>(tgt)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t nz = (w[e] | (w[e] >> 1) | (w[e] >> 2) | (w[e] >> 3) | (w[e] >> 4) | (w[e] >> 5) | (w[e] >> 6) | (w[e] >> 7)) & 0x01010101u;
            k += __popc(nz);
        }
    }
    for (long i = (nv << 4) + threadIdx.x; i < n; i += 1024) k += tgt[i] ? 1 : 0;
    float c = block_sum((float)k, red);
    if (threadIdx.x == 0) ws[1] = c;
}

__global__ __launch_bounds__(256) void mse_rows_kernel(wj_mse_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    const long R = a.rows ? (long)a.n_rows : (long)a.B * a.G * a.T;
    const float count = a.workspace[1];
    const float gk = 2.0f / ((float)D * (count + 1e-8f)) * a.gscale * (a.gscale_ptr ? a.gscale_ptr[0] : 1.0f);
    for (long r = blockIdx.x * 4L + wave; r < R; r += gridDim.x * 4L) {
        const long dense = a.rows ? (long)a.rows[r] : r;     // (b*G+g)*T + t; preds / dpreds are indexed by r
        const int t = (int)(dense % a.T);
        const long bg = dense / a.T;
        const int b = (int)(bg / a.G);
        const bool on = a.tgt[dense] != 0;
        float err = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            bf16x4 dp;
#pragma unroll
            for (int e = 0; e < 4; ++e) dp[e] = f2bf(0.f);
            if (on) {
                const bf16x4 p = *reinterpret_cast<const bf16x4*>((const bf16_t*)a.preds + r * D + c);
                const f32x4 y = *reinterpret_cast<const f32x4*>(a.targets + ((long)b * a.T + t) * D + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = bf2f(p[e]) - y[e];
                    err += d * d;
                    dp[e] = f2bf(d * gk);
                }
            }
            if (a.dpreds) *reinterpret_cast<bf16x4*>((bf16_t*)a.dpreds + r * D + c) = dp;
        }
        err = wave_sum(err);
        if (lane == 0) a.workspace[2 + r] = on ? err / (float)D : 0.f;
    }
}

__global__ __launch_bounds__(1024) void mse_final_kernel(float* __restrict__ ws, float* __restrict__ loss, long R) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < R; i += 1024) s += ws[2 + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        loss[0] = s / (ws[1] + 1e-8f);
        loss[1] = ws[1];
    }
}

// ------------------------------------------------------------------------------------------- EMA / AdamW / norms / casts
__global__ __launch_bounds__(256) void ema_kernel(wj_ema_args a) {
    const long n4 = a.n / 4;
    const float r = a.r, q = 1.0f - a.r;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(a.student + i * 4);
        f32x4 t = *reinterpret_cast<const f32x4*>(a.teacher + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = t[e] * r + q * s[e];
        *reinterpret_cast<f32x4*>(a.teacher + i * 4) = t;
        if (a.teacher_bf16) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(t[e]);
            *reinterpret_cast<bf16x4*>((bf16_t*)a.teacher_bf16 + i * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ ws, long n) {
    __shared__ float red[16];
    const long n4 = n / 4;
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int n, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += ws[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

__global__ __launch_bounds__(256) void adamw_kernel(wj_adamw_args a) {
    const long n4 = a.n / 4;
    float coef = a.grad_scale;
    if (a.max_norm > 0.f && a.sumsq) {
        const float norm = sqrtf(a.sumsq[0]) * a.grad_scale;
        coef *= fminf(1.0f, a.max_norm / (norm + 1e-6f));
    }
    const float decay = 1.0f - a.lr * a.weight_decay;
    const float step = a.lr / a.bc1;
    const float rbc2 = rsqrtf(a.bc2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 p = *reinterpret_cast<const f32x4*>(a.p + i * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i * 4);
        if (a.zero_grad) *reinterpret_cast<f32x4*>(a.g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 m = *reinterpret_cast<const f32x4*>(a.m + i * 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(a.v + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = g[e] * coef;
            p[e] *= decay;
            m[e] = a.beta1 * m[e] + (1.0f - a.beta1) * ge;
            v[e] = a.beta2 * v[e] + (1.0f - a.beta2) * ge * ge;
            p[e] -= step * m[e] / (sqrtf(v[e]) * rbc2 + a.eps);
        }
        *reinterpret_cast<f32x4*>(a.p + i * 4) = p;
        *reinterpret_cast<f32x4*>(a.m + i * 4) = m;
        *reinterpret_cast<f32x4*>(a.v + i * 4) = v;
        if (a.p_bf16) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(p[e]);
            *reinterpret_cast<bf16x4*>((bf16_t*)a.p_bf16 + i * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long n4 = n / 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(src + i * 4);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(s[e]);
        *reinterpret_cast<bf16x4*>(dst + i * 4) = o;
    }
    for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = f2bf(src[i]);
}

// ------------------------------------------------------------------------------------------- batched bf16 transposes (W^T shadows)
// dst[off + c * rows + r] = src[off + r * cols + c] for every matrix of the table: the bf16 weight shadows of the transformer stacks,
// transposed once per step so that every dgrad dx = dy . W runs as a ROW-form GEMM against W^T (the persistent eight-phase kernel)
// instead of the col-form 256 x 128 schedule.  One workgroup per 64 x 64 tile: 16-byte reads along the source rows, a padded LDS
// tile (pitch 66 elements: the column walk of the transposed read touches 32 distinct banks), 16-byte writes along the destination
// rows.  table: int64 [n_mats][4] = (element offset, rows, cols, first tile of the matrix in the launch's tile list).
__global__ __launch_bounds__(256) void transpose_tiles_kernel(wj_transpose_args a) {
    __shared__ bf16_t tile[64][66];
    const int t = threadIdx.x;
    int q = 0;
#pragma unroll 1
    for (int x = 1; x < a.n_mats; ++x)
        if ((long)blockIdx.x >= a.table[4 * x + 3]) q = x;
    const long off = a.table[4 * q], rows = a.table[4 * q + 1], cols = a.table[4 * q + 2];
    const int local = (int)(blockIdx.x - a.table[4 * q + 3]);
    const int tiles_c = (int)(cols / 64);
    const int tr = local / tiles_c, tc = local - tr * tiles_c;
    const bf16_t* src = (const bf16_t*)a.src + off + (long)(tr * 64) * cols + tc * 64;
    bf16_t* dst = (bf16_t*)a.dst + off + (long)(tc * 64) * rows + tr * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (t >> 3) + 32 * u, c8 = (t & 7) * 8;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)r * cols + c8);
#pragma unroll
        for (int x = 0; x < 8; ++x) tile[r][c8 + x] = v[x];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = (t >> 3) + 32 * u, r8 = (t & 7) * 8;       // destination row c (a source column), 8 source rows r8 ..
        bf16x8 v;
#pragma unroll
        for (int x = 0; x < 8; ++x) v[x] = tile[r8 + x][c];
        *reinterpret_cast<bf16x8*>(dst + (long)c * rows + r8) = v;
    }
}

// ------------------------------------------------------------------------------------------- crop + normalise
__global__ __launch_bounds__(1024) void crop_kernel(wj_crop_args a) {
    __shared__ float red[16];
    const int bs = blockIdx.x, b = bs / a.S;
    const int start = a.starts[bs];
    const long n = (long)a.C * a.length;
    const float* src = a.src + (long)b * a.C * a.L_full;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const int c = (int)(i / a.length), l = (int)(i - (long)c * a.length);
        s += src[(long)c * a.L_full + start + l];
    }
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const int c = (int)(i / a.length), l = (int)(i - (long)c * a.length);
        const float d = src[(long)c * a.L_full + start + l] - mean;
        q += d * d;
    }
    const float stdv = sqrtf(block_sum(q, red) / (float)(n - 1));
    const float inv = 1.0f / (stdv + 1e-5f);
    const int orow = a.perm_inv ? a.perm_inv[bs] : bs;
    bf16_t* out = (bf16_t*)a.out + (long)orow * n;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const int c = (int)(i / a.length), l = (int)(i - (long)c * a.length);
        out[i] = f2bf((src[(long)c * a.L_full + start + l] - mean) * inv);
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int wj_abi_version(void) { return WJ_ABI_VERSION; }
extern "C" int wj_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int wj_struct_size(const char* name) {
    if (!name) return -1;
#define WJ_SZ(T) if (!strcmp(name, #T)) return (int)sizeof(T);
    WJ_SZ(wj_gemm_args) WJ_SZ(wj_ln_fwd_args) WJ_SZ(wj_ln_bwd_args) WJ_SZ(wj_colsum_args) WJ_SZ(wj_attn_fwd_args)
    WJ_SZ(wj_attn_bwd_args) WJ_SZ(wj_conv0_fwd_args) WJ_SZ(wj_conv0_bwd_args) WJ_SZ(wj_gelu_bwd_args) WJ_SZ(wj_conv_w_args)
    WJ_SZ(wj_add_pos_args) WJ_SZ(wj_gather_args) WJ_SZ(wj_scatter_fill_args) WJ_SZ(wj_scatter_fill_bwd_args)
    WJ_SZ(wj_unmask_rows_args) WJ_SZ(wj_instnorm_args) WJ_SZ(wj_mse_args) WJ_SZ(wj_ema_args) WJ_SZ(wj_sumsq_args)
    WJ_SZ(wj_adamw_args) WJ_SZ(wj_cast_args) WJ_SZ(wj_crop_args) WJ_SZ(wj_zero_rows_args) WJ_SZ(wj_instnorm_mean_args) WJ_SZ(wj_spin_args)
    WJ_SZ(wj_gemm_fp8_args) WJ_SZ(wj_quantize_fp8_args) WJ_SZ(wj_wgrad_group_args) WJ_SZ(wj_rir_conv_args) WJ_SZ(wj_snr_mix_args) WJ_SZ(wj_resample_args) WJ_SZ(wj_mse_groups_args) WJ_SZ(wj_transpose_args) WJ_SZ(wj_colsum_group_args) WJ_SZ(wj_rccl_init_args) WJ_SZ(wj_rccl_launch_args) WJ_SZ(wj_rccl_wait_args)
#ifdef WJ_LAB
    WJ_SZ(wj_collective_footprint_args)
#endif
#undef WJ_SZ
    return -1;
}

int64_t wj_gemm_ws_bytes(const wj_gemm_args* a);              // csrc/gemm.hip
int64_t wj_conv0_fwd_ws_bytes(const wj_conv0_fwd_args* a);   // csrc/conv0.hip
int64_t wj_conv0_bwd_ws_bytes(const wj_conv0_bwd_args* a);
int64_t wj_rir_conv_ws_bytes(const wj_rir_conv_args* a);      // csrc/scene.hip
int64_t wj_snr_mix_ws_bytes(const wj_snr_mix_args* a);
int64_t wj_mse_groups_ws_bytes(const wj_mse_groups_args* a);  // csrc/denoise.hip

extern "C" int64_t wj_workspace_bytes(const char* fn, const void* args) {
    if (!fn || !args) return -1;
    if (!strcmp(fn, "wj_gemm_bf16")) return wj_gemm_ws_bytes((const wj_gemm_args*)args);
    if (!strcmp(fn, "wj_mask_scatter_fill_pos_bwd")) {           // `partials`: one row of D floats per workgroup
        const wj_scatter_fill_bwd_args* a = (const wj_scatter_fill_bwd_args*)args;
        return (int64_t)wj_scatter_fill_bwd_partial_rows(a->B, a->T) * a->D * 4;
    }
    if (!strcmp(fn, "wj_layernorm_bwd")) return 1536LL * 3 * ((const wj_ln_bwd_args*)args)->D * 4;
    if (!strcmp(fn, "wj_attn_bwd")) {
        const wj_attn_bwd_args* a = (const wj_attn_bwd_args*)args;
        return (int64_t)a->B * 3 * a->H * a->hd * 4;
    }
    if (!strcmp(fn, "wj_conv0_gn_gelu_fwd")) return wj_conv0_fwd_ws_bytes((const wj_conv0_fwd_args*)args);
    if (!strcmp(fn, "wj_conv0_gn_gelu_bwd")) return wj_conv0_bwd_ws_bytes((const wj_conv0_bwd_args*)args);
    if (!strcmp(fn, "wj_masked_mse")) {
        const wj_mse_args* a = (const wj_mse_args*)args;
        return (2 + (int64_t)a->B * a->G * a->T) * 4;
    }
    if (!strcmp(fn, "wj_grad_sumsq")) return 1024 * 4;
    if (!strcmp(fn, "wj_rir_convolve")) return wj_rir_conv_ws_bytes((const wj_rir_conv_args*)args);
    if (!strcmp(fn, "wj_snr_mix")) return wj_snr_mix_ws_bytes((const wj_snr_mix_args*)args);
    if (!strcmp(fn, "wj_mse_groups")) return wj_mse_groups_ws_bytes((const wj_mse_groups_args*)args);
    static const char* const none[] = {"wj_layernorm_fwd", "wj_colsum_bf16", "wj_colsum_f32", "wj_attn_fwd", "wj_gelu_bwd_bf16",
        "wj_conv_weight_layout", "wj_add_pos", "wj_mask_gather_rows", "wj_mask_scatter_fill_pos",
        "wj_unmask_rows_f32", "wj_instnorm_accumulate", "wj_instnorm_mean", "wj_ema_update", "wj_adamw_step", "wj_cast_f32_to_bf16",
        "wj_crop_normalize_bf16", "wj_zero_rows", "wj_spin", "wj_gemm_mxfp8", "wj_quantize_mxfp8", "wj_wgrad_grouped", "wj_resample_fir", "wj_transpose_bf16", "wj_colsum_f32_group", "wj_rccl_bucket_allreduce_launch", "wj_rccl_bucket_allreduce_wait", "wj_collective_footprint"};
    for (const char* n : none)
        if (!strcmp(fn, n)) return 0;
    return -1;
}

extern "C" int wj_gelu_bwd_bf16(const wj_gelu_bwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (a && a->rows) {
        if (!a->dpost || !a->pre || !a->dpre || a->n_rows < 0 || a->row_elems <= 0 || (a->row_elems & 7)) return WJ_ERR_ARG;
        if (a->n_rows == 0) return WJ_OK;
        hipLaunchKernelGGL(gelu_bwd_rows_kernel, dim3(grid_for((long)a->n_rows * (a->row_elems / 8), 256)), dim3(256), 0, STREAM,
                           (bf16_t*)a->dpost, (const bf16_t*)a->pre, (bf16_t*)a->dpre, a->rows, a->n_rows, a->row_elems / 8,
                           a->clear_dpost);
        WJ_CHECK_LAUNCH();
        return WJ_OK;
    }
    if (!a || !a->dpost || !a->pre || !a->dpre || a->n <= 0 || (a->n & 7)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(a->n / 8, 256)), dim3(256), 0, STREAM, (const bf16_t*)a->dpost,
                       (const bf16_t*)a->pre, (bf16_t*)a->dpre, (long)(a->n / 8));
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

__global__ void spin_kernel(long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long)(__builtin_amdgcn_s_memtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int wj_spin(const wj_spin_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || a->ticks < 0 || a->ticks > (1LL << 32)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, STREAM, (long)a->ticks);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

#ifdef WJ_LAB
// the footprint of a collective on its GPU (see the header): `passes` in-place read + rewrite sweeps by a few resident workgroups,
// stretched to `min_ticks` of the 100 MHz clock by pacing every piece
typedef __attribute__((ext_vector_type(4))) unsigned cf_u32x4;
__global__ __launch_bounds__(512) void collective_footprint_kernel(cf_u32x4* __restrict__ buf, long n16, long min_ticks, int passes) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = min(n16, lo + per);
    constexpr long PIECE = 512 * 8;                        // 64 KiB per workgroup and piece
    const long npieces = ((hi > lo ? hi - lo : 0) + PIECE - 1) / PIECE * passes;
    long done = 0;
    for (int ps = 0; ps < passes; ++ps)
        for (long p0 = lo; p0 < hi; p0 += PIECE) {
            cf_u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long i = p0 + u * 512 + threadIdx.x;
                v[u] = cf_u32x4{0u, 0u, 0u, 0u};
                if (i < hi) v[u] = __builtin_nontemporal_load(buf + i);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long i = p0 + u * 512 + threadIdx.x;
                asm volatile("" : "+v"(v[u].x));           // (keeps the rewrite of an unchanged value from being dropped)
                if (i < hi) __builtin_nontemporal_store(v[u], buf + i);
            }
            ++done;
            if (min_ticks > 0) {
                const unsigned long long due = (unsigned long long)(min_ticks * done / npieces);
                while (__builtin_amdgcn_s_memrealtime() - t0 < due) __builtin_amdgcn_s_sleep(16);
            }
        }
}

extern "C" int wj_collective_footprint(const wj_collective_footprint_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->buf || a->bytes <= 0 || (a->bytes & 15) || ((uintptr_t)a->buf & 15) || a->workgroups < 1 || a->workgroups > 256 ||
        a->passes < 1 || a->passes > 8 || a->min_ticks < 0)
        return WJ_ERR_ARG;
    hipLaunchKernelGGL(collective_footprint_kernel, dim3(a->workgroups), dim3(512), 0, STREAM, (cf_u32x4*)a->buf, (long)(a->bytes / 16),
                       (long)a->min_ticks, a->passes);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
#endif

extern "C" int wj_zero_rows(const wj_zero_rows_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->buf || !a->rows || a->n_rows < 0 || a->row_bytes <= 0 || (a->row_bytes & 15)) return WJ_ERR_ARG;
    if (a->n_rows == 0) return WJ_OK;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for((long)a->n_rows * (a->row_bytes / 16), 256)), dim3(256), 0, STREAM,
                       (char*)a->buf, a->rows, a->n_rows, a->row_bytes / 16);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_conv_weight_layout(const wj_conv_w_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->src || !a->dst || a->C_out <= 0 || a->C_in <= 0 || a->k <= 0 || a->mode < 0 || a->mode > 2) return WJ_ERR_ARG;
    if (a->mode == 1 && (a->U <= 0 || a->rho < 0 || a->rho + a->stride * (a->U - 1) >= a->k)) return WJ_ERR_ARG;
    const long n = a->mode == 1 ? (long)a->U * a->C_out * a->C_in : (long)a->C_out * a->C_in * a->k;
    hipLaunchKernelGGL(conv_w_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_add_pos(const wj_add_pos_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->pos || a->M <= 0 || a->T <= 0 || a->D <= 0 || (a->D & 3)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(add_pos_kernel, dim3(grid_for((long)a->M * a->D / 4, 256)), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_mask_gather_rows(const wj_gather_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->idx || !a->out || a->n_rows < 0 || a->D <= 0) return WJ_ERR_ARG;
    if (a->elem_bytes != 2 && a->elem_bytes != 4) return WJ_ERR_ARG;
    const int row_bytes = a->D * a->elem_bytes;
    if (row_bytes & 15) return WJ_ERR_ARG;
    if (a->n_rows == 0) return WJ_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(a->n_rows, 1, 16384)), dim3(256), 0, STREAM, (const char*)a->x, a->idx,
                       (char*)a->out, a->n_rows, row_bytes);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_mask_scatter_fill_pos(const wj_scatter_fill_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->ctx_feats || !a->inv || !a->mask_token || !a->pos || a->B <= 0 || a->T <= 0 || a->D <= 0 || (a->D & 3) ||
        a->G <= 0)
        return WJ_ERR_ARG;
    if (a->rows) {
        if (a->n_rows < 0) return WJ_ERR_ARG;
        if (a->n_rows == 0) return WJ_OK;
        hipLaunchKernelGGL(scatter_fill_rows_kernel, dim3(grid_for((long)a->n_rows, 4)), dim3(256), 0, STREAM, *a);
        WJ_CHECK_LAUNCH();
        return WJ_OK;
    }
    hipLaunchKernelGGL(scatter_fill_kernel, dim3(grid_for((long)a->B * a->T, 4)), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_scatter_fill_bwd_partial_rows(int B, int T) { return (int)(((long)B * T + SFB_ROWS - 1) / SFB_ROWS); }

extern "C" int wj_mask_scatter_fill_pos_bwd(const wj_scatter_fill_bwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->d_in || !a->inv || !a->d_ctx_feats || a->B <= 0 || a->T <= 0 || a->D <= 0 || (a->D & 3) || a->D > 1024 ||
        a->G <= 0 || a->G > 15)              // (a wave's 4 rows x (G + 1) indices are fetched by one load per lane)
        return WJ_ERR_ARG;
    const int grid = (a->B * a->T + SFB_ROWS - 1) / SFB_ROWS;
    if (a->D <= 512) hipLaunchKernelGGL(scatter_fill_bwd_kernel<2>, dim3(grid), dim3(256), (size_t)4 * a->D * sizeof(float), STREAM, *a);
    else hipLaunchKernelGGL(scatter_fill_bwd_kernel<4>, dim3(grid), dim3(256), (size_t)4 * a->D * sizeof(float), STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_unmask_rows_f32(const wj_unmask_rows_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->src || !a->dst || a->M <= 0 || a->D <= 0 || (a->D & 3)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(unmask_rows_kernel, dim3(grid_for((long)a->M * a->D / 4, 256)), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_instnorm_accumulate(const wj_instnorm_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->targets || a->B <= 0 || a->TD <= 0 || (a->TD & 3)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(instnorm_kernel, dim3(a->B), dim3(1024), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_instnorm_mean(const wj_instnorm_mean_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->stats || !a->targets || a->B <= 0 || a->TD <= 0 || (a->TD & 3) || a->K < 1 || a->K > 8) return WJ_ERR_ARG;
    const float* xs[8] = {a->x0, a->x1, a->x2, a->x3, a->x4, a->x5, a->x6, a->x7};
    for (int l = 0; l < a->K; ++l)
        if (!xs[l]) return WJ_ERR_ARG;
    int gx = (a->TD / 4 + 255) / 256;
    if (gx > 32) gx = 32;
    hipLaunchKernelGGL(instnorm_mean_kernel, dim3(gx, a->B), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_masked_mse(const wj_mse_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->preds || !a->targets || !a->tgt || !a->loss || !a->workspace) return WJ_ERR_ARG;
    if (a->B <= 0 || a->G <= 0 || a->T <= 0 || a->D <= 0 || (a->D & 3)) return WJ_ERR_ARG;
    const long Rd = (long)a->B * a->G * a->T;          // dense positions (the target count runs over all of them)
    if (a->rows && a->n_rows < 0) return WJ_ERR_ARG;
    const long R = a->rows ? (long)a->n_rows : Rd;    // rows actually held by preds / dpreds
    hipLaunchKernelGGL(mse_count_kernel, dim3(1), dim3(1024), 0, STREAM, a->tgt, a->workspace, Rd);
    if (R > 0) hipLaunchKernelGGL(mse_rows_kernel, dim3(grid_for(R, 4)), dim3(256), 0, STREAM, *a);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(1024), 0, STREAM, a->workspace, a->loss, R);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_ema_update(const wj_ema_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->student || !a->teacher || a->n <= 0 || (a->n & 3)) return WJ_ERR_ARG;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(a->n / 4, 256)), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_grad_sumsq(const wj_sumsq_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->g || !a->out || !a->workspace || a->n <= 0 || (a->n & 3)) return WJ_ERR_ARG;
    int grid = grid_for(a->n / 4, 256, 1024);
    if (a->workgroups > 0 && a->workgroups < grid) grid = a->workgroups;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, STREAM, a->g, a->workspace, (long)a->n);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, STREAM, (const float*)a->workspace, a->out, grid, a->accumulate);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_adamw_step(const wj_adamw_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->p || !a->g || !a->m || !a->v || a->n <= 0 || (a->n & 3)) return WJ_ERR_ARG;
    int grid = grid_for(a->n / 4, 256);
    if (a->workgroups > 0 && a->workgroups < grid) grid = a->workgroups;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_transpose_bf16(const wj_transpose_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->src || !a->dst || !a->table || a->n_mats <= 0 || a->n_tiles <= 0) return WJ_ERR_ARG;
    if (((uintptr_t)a->src | (uintptr_t)a->dst) & 15) return WJ_ERR_ARG;
    hipLaunchKernelGGL(transpose_tiles_kernel, dim3(a->n_tiles), dim3(256), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_cast_f32_to_bf16(const wj_cast_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->src || !a->dst || a->n <= 0) return WJ_ERR_ARG;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(a->n / 4 + 1, 256)), dim3(256), 0, STREAM, a->src, (bf16_t*)a->dst, (long)a->n);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_crop_normalize_bf16(const wj_crop_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->src || !a->starts || !a->out || a->B <= 0 || a->S <= 0 || a->C <= 0 || a->length <= 1 || a->L_full < a->length)
        return WJ_ERR_ARG;
    hipLaunchKernelGGL(crop_kernel, dim3(a->B * a->S), dim3(1024), 0, STREAM, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
