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

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous chunk of the logical id space (neighbouring tiles then share that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
