// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of libwavjepa_hip.so.
// Wavefront = 64 lanes everywhere in this tree; nothing here is portable to 32-wide hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define WJ_OK 0
#define WJ_ERR_ARG (-1)
#define WJ_ERR_LAUNCH (-2)
#define WJ_ERR_UNSUPPORTED (-3)

// hipGetLastError() reports the last error of ANY HIP call of this thread -- including benign ones of the host framework (an
// event query returning hipErrorNotReady) -- so every entry point first discards what it did not cause.
#define WJ_CLEAR_STALE_ERROR() (void)hipGetLastError()

#define WJ_CHECK_LAUNCH()                                                                              \
    do {                                                                                               \
        hipError_t e__ = hipGetLastError();                                                            \
        if (e__ != hipSuccess) {                                                                       \
            if (getenv("WJ_DEBUG")) fprintf(stderr, "[wavjepa_hip] %s: %s\n", __func__, hipGetErrorString(e__)); \
            return WJ_ERR_LAUNCH;                                                                      \
        }                                                                                              \
    } while (0)

// Laboratory switches (A/B runs, diagnostics that change results or schedules): read from the environment ONLY in the -DWJ_LAB build
// (libwavjepa_hip_lab.so, include/wavjepa_hip_lab.h); the release library compiles each to its default -- a production process cannot be
// pushed onto a diagnostic path by a stray variable.  Switches that are safe in production use getenv directly and are listed in
// INTEGRATION.md (tests/test_host_cpu.py checks both lists against the sources).
#ifdef WJ_LAB
constexpr bool WJ_LAB_BUILD = true;
static inline int wj_lab_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline const char* wj_lab_env_str(const char* name) { return getenv(name); }
#else
constexpr bool WJ_LAB_BUILD = false;
static inline int wj_lab_env_int(const char*, int dflt) { return dflt; }
static inline const char* wj_lab_env_str(const char*) { return nullptr; }
#endif

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves).  `red` is >= 16 floats of LDS.  All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// erf-GELU (nn.GELU(approximate='none')) and its derivative, fp32.
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16 resolution): ~14 VALU ops instead of libm's
// branchy erff (~60), and gelu' reuses the same exponential: exp(-(x/sqrt2)^2) = exp(-x^2/2).
__device__ __forceinline__ void erf_parts(float x, float& erf_abs, float& e) {
    // returns erf(|x|/sqrt2) and e = exp(-x^2/2)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn is a ~10-instruction IEEE divide
    e = __expf(-z * z);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    erf_abs = fmaf(-p * t, e, 1.0f);
}
__device__ __forceinline__ float gelu_f(float x) {
    float ea, e;
    erf_parts(x, ea, e);
    return 0.5f * x * (1.0f + copysignf(ea, x));
}
// gelu(x) and gelu'(x) from one erf / one exp
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& gp) {
    float ea, e;
    erf_parts(x, ea, e);
    const float cdf = 0.5f * (1.0f + copysignf(ea, x));
    g = x * cdf;
    gp = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float ea, e;
    erf_parts(x, ea, e);
    return fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.0f + copysignf(ea, x)));
}

// The same erf-GELU written two elements at a time: left to itself hipcc packs part of the scalar form above and shuffles
// registers around it (206 VALU instructions per 8 outputs of gelu + gelu' in the GEMM epilogue, 103 written this way; 141 -> 90
// for gelu alone).  Measured: BIAS_GELU2 -6 %, CONV_GELU -4 %, conv0 apply -6 %, teacher BIAS_GELU unchanged -- on gfx950 a
// v_pk_fma_f32 issues at the cost of two v_fma_f32, so what is saved is the shuffling, not half of the arithmetic.
// Same A-S 7.1.26 erf; exp(-x^2/2) taken as exp2(x^2 * (-log2(e)/2)).
template <bool WANT_GRAD>
__device__ __forceinline__ void gelu_pk(const f32x2 x, f32x2& g, f32x2& gp) {
    constexpr float C = 0.3275911f * 0.70710678118654752440f;
    f32x2 t, e;
    t.x = __builtin_amdgcn_rcpf(fmaf(fabsf(x.x), C, 1.0f));            // |x| is a free source modifier of the scalar fma only
    t.y = __builtin_amdgcn_rcpf(fmaf(fabsf(x.y), C, 1.0f));
    const f32x2 u = (x * x) * (-0.5f * 1.44269504088896340736f);
    e.x = __builtin_amdgcn_exp2f(u.x);
    e.y = __builtin_amdgcn_exp2f(u.y);
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
    const f32x2 ea = __builtin_elementwise_fma(-(p * t), e, f32x2{1.0f, 1.0f});
    f32x2 s;
    s.x = __builtin_copysignf(ea.x, x.x);
    s.y = __builtin_copysignf(ea.y, x.y);
    const f32x2 cdf = __builtin_elementwise_fma(s, f32x2{0.5f, 0.5f}, f32x2{0.5f, 0.5f});
    g = x * cdf;
    if constexpr (WANT_GRAD) gp = __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}
// 8 bf16 in, gelu (and gelu') as 8 bf16 out
template <bool WANT_GRAD>
__device__ __forceinline__ void gelu_bf16x8(const bf16x8 h, bf16x8& g, bf16x8& gp) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
    const u32x4_ w = __builtin_bit_cast(u32x4_, h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 x, a, b;
        x.x = __uint_as_float(w[j] << 16);
        x.y = __uint_as_float(w[j] & 0xffff0000u);
        gelu_pk<WANT_GRAD>(x, a, b);
        g[2 * j] = f2bf(a.x); g[2 * j + 1] = f2bf(a.y);
        if constexpr (WANT_GRAD) { gp[2 * j] = f2bf(b.x); gp[2 * j + 1] = f2bf(b.y); }
    }
}

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous chunk of the logical id space (neighbouring tiles then share that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
