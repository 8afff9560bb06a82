// Kernels of the denoiser stage (reference wavjepa/denoiser.py) that the JEPA step does not have:
//   * wj_resample_fir  -- polyphase windowed-sinc resampling as torchaudio.functional.resample applies it (denoiser.py:29-42,
//                         WebAudioDataModule.py:50-60): y[b][i*new + p] = sum_k kernel[p][k] * xpad[b][i*orig + k], xpad = x with
//                         `width` zeros in front and `width + orig` behind, cut to ceil(new * L_in / orig) samples.  The kernel table
//                         (kaiser-windowed sinc) is built on the host (wavjepa_amd/resample.py) exactly as torchaudio builds it.
//   * wj_mse_groups    -- loss = sum_g w[g] * mean((preds[g] - targets)^2) over G prediction sets against ONE target tensor
//                         (denoiser.py:350-355: alpha * mse(clean) + (1 - alpha) * mse(generated)), plus d loss / d preds.
// Both HBM-bound streaming kernels; reductions go through per-workgroup partials folded in a fixed order (bit-reproducible).
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int MSE_BLOCKS = 1024;

// One workgroup: RS_THREADS consecutive frames i of one clip, all `nw` phases.  The input window of the block
// (RS_THREADS * orig + taps samples) and the kernel table live in LDS.
__global__ __launch_bounds__(RS_THREADS) void resample_fir_kernel(wj_resample_args a) {
    extern __shared__ float sm[];
    float* win = sm;                                   // [RS_THREADS * orig + taps]
    float* ker = sm + RS_THREADS * a.orig + a.taps;    // [nw][taps]
    const int b = blockIdx.y;
    const long i0 = (long)blockIdx.x * RS_THREADS;
    const int nwin = RS_THREADS * a.orig + a.taps;
    const float* x = a.x + (long)b * a.L_in;
    for (int j = threadIdx.x; j < nwin; j += RS_THREADS) {
        const long src = i0 * a.orig + j - a.width;
        win[j] = (src >= 0 && src < a.L_in) ? x[src] : 0.f;
    }
    for (int j = threadIdx.x; j < a.nw * a.taps; j += RS_THREADS) ker[j] = a.kernel[j];
    __syncthreads();
    const long i = i0 + threadIdx.x;
    float* y = a.y + (long)b * a.L_out;
    for (int p = 0; p < a.nw; ++p) {
        const long o = i * a.nw + p;
        if (o >= a.L_out) break;
        const float* w = win + threadIdx.x * a.orig;
        const float* kp = ker + p * a.taps;
        float acc = 0.f;
        for (int k = 0; k < a.taps; ++k) acc = fmaf(kp[k], w[k], acc);
        y[o] = acc;
    }
}

__global__ __launch_bounds__(256) void mse_groups_partial_kernel(wj_mse_groups_args a) {
    const int g = blockIdx.y;
    const float* p = a.preds + (long)g * a.n;
    float s = 0.f;
    const long span = (a.n + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * span, hi = min(a.n, lo + span);
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float d = p[i] - a.targets[i];
        s = fmaf(d, d, s);
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.workspace[(long)g * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void mse_groups_final_kernel(wj_mse_groups_args a, int nblocks) {
    if (threadIdx.x != 0) return;
    float total = 0.f;
    for (int g = 0; g < a.G; ++g) {
        float s = 0.f;
        for (int i = 0; i < nblocks; ++i) s += a.workspace[(long)g * nblocks + i];     // fixed order
        const float l = s / (float)a.n;
        a.loss[1 + g] = l;
        total = fmaf(a.w[g], l, total);
    }
    a.loss[0] = total;
}

__global__ __launch_bounds__(256) void mse_groups_grad_kernel(wj_mse_groups_args a) {
    const int g = blockIdx.y;
    const float c = 2.0f * a.w[g] / (float)a.n * (a.gscale ? a.gscale[0] : 1.0f);
    const float* p = a.preds + (long)g * a.n;
    float* d = a.dpreds + (long)g * a.n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) d[i] = c * (p[i] - a.targets[i]);
}

}  // namespace

int64_t wj_mse_groups_ws_bytes(const wj_mse_groups_args* a) { return a->G > 0 ? (int64_t)a->G * MSE_BLOCKS * 4 : -1; }

extern "C" int wj_resample_fir(const wj_resample_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->x || !a->y || !a->kernel) return WJ_ERR_ARG;
    if (a->B <= 0 || a->L_in <= 0 || a->L_out <= 0 || a->orig <= 0 || a->nw <= 0 || a->width < 0 || a->taps != 2 * a->width + a->orig)
        return WJ_ERR_ARG;
    const long lds = ((long)RS_THREADS * a->orig + a->taps + (long)a->nw * a->taps) * 4;
    if (lds > 64 * 1024) return WJ_ERR_UNSUPPORTED;
    const long frames = (a->L_out + a->nw - 1) / a->nw;
    hipLaunchKernelGGL(resample_fir_kernel, dim3((unsigned)((frames + RS_THREADS - 1) / RS_THREADS), a->B), dim3(RS_THREADS), (size_t)lds,
                       (hipStream_t)stream, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_mse_groups(const wj_mse_groups_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->preds || !a->targets || !a->loss || !a->workspace || a->G <= 0 || a->G > 4 || a->n <= 0) return WJ_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_groups_partial_kernel, dim3(MSE_BLOCKS, a->G), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(mse_groups_final_kernel, dim3(1), dim3(64), 0, st, *a, MSE_BLOCKS);
    if (a->dpreds) hipLaunchKernelGGL(mse_groups_grad_kernel, dim3(2048, a->G), dim3(256), 0, st, *a);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
