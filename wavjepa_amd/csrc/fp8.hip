// MX fp8 quantisation for gfx950: bf16 rows -> OCP e4m3 bytes + one E8M0 scale per 32 consecutive elements (the operand format of
// v_mfma_scale_f32_16x16x128_f8f6f4; see wj_gemm_mxfp8 in gemm.hip).  HBM-bound: 2 B read, 1 B + 1/32 B written per element.
//   thread = 8 consecutive elements (16-B load, 8-B store); a 32-element block = 4 lanes (two shuffles for its max);
//   a 128-element K tile = 16 lanes: their four scale bytes are packed into ONE dword scales[kt][row] by a lane of the group.
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(wj_quantize_fp8_args a) {
    const int per_row = a.K / 8;                     // threads per row (a multiple of 16)
    const long total = (long)a.M * per_row;
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += (long)gridDim.x * 256) {     // uniform trip count per wave: total % 16 == 0, 64 | 256
        const int row = (int)(t / per_row), c8 = (int)(t - (long)row * per_row);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>((const bf16_t*)a.x + (long)row * a.ldx + c8 * 8);
        float f[8], amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f[e] = bf2f(v[e]); amax = fmaxf(amax, fabsf(f[e])); }
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        // s = ceil(log2(amax / 448)): smallest power of two with amax * 2^-s <= 448 (e4m3 max)
        int s = 0;
        if (amax > 0.f) {
            int ex;
            const float m = frexpf(amax * (1.0f / 448.0f), &ex);      // amax / 448 = m * 2^ex, m in [0.5, 1)
            s = (m == 0.5f) ? ex - 1 : ex;
            s = max(-127, min(127, s));
        }
        const float inv = __builtin_amdgcn_ldexpf(1.0f, -s);
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
        *reinterpret_cast<uint2*>((unsigned char*)a.q + (long)row * a.ldq + c8 * 8) = make_uint2(lo, hi);
        // the four block scales of this K tile: lanes 0, 4, 8, 12 of the 16-lane group hold blocks 0..3
        const unsigned sbyte = (unsigned)(s + 127);
        const int lane = threadIdx.x & 63, base = lane & ~15;
        const unsigned s0 = __shfl(sbyte, base + 0, 64), s1 = __shfl(sbyte, base + 4, 64), s2 = __shfl(sbyte, base + 8, 64),
                       s3 = __shfl(sbyte, base + 12, 64);
        if ((lane & 15) == 0) ((uint32_t*)a.scales)[(long)(c8 / 16) * a.ld_scale + row] = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
    }
}

}  // namespace

extern "C" int wj_quantize_mxfp8(const wj_quantize_fp8_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->q || !a->scales || a->M <= 0 || a->K <= 0 || (a->K % 128) || (a->ldx & 7) || (a->ldq & 7) || a->ld_scale < a->M)
        return WJ_ERR_ARG;
    const long total = (long)a->M * (a->K / 8);
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(quantize_mxfp8_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
