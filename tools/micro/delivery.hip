// Micro-benchmark: per-CU operand delivery L2 -> CU on gfx950, (a) global_load_lds_dwordx4 (LDS-DMA), (b) global_load_dwordx4 into
// VGPRs (+ ds_write_b128), on a working set that is L2 / Infinity-Cache resident.  Build: hipcc --offload-arch=gfx950 -O3 delivery.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

// every workgroup streams `iters` tiles of 32 KiB (row segments of SEG bytes from 256 or 512 rows) from its own window of the buffer
template <int MODE, int SEG>
__global__ __launch_bounds__(256, 1) void stream_kernel(const char* __restrict__ buf, long window, long row_pitch, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const char* base = buf + (long)blockIdx.x % 64 * window;       // 64 distinct windows: neighbours share lines through L2
    constexpr int LPR = SEG / 16;                                    // lanes per row segment
    constexpr int RPI = 64 / LPR;                                    // rows per wave-instruction
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const long k_off = (long)(it % 16) * SEG;                    // walk along the row like a K loop
#pragma unroll
        for (int u = 0; u < 8; ++u) {                                // 8 x 1 KiB per wave = 32 KiB per workgroup per iteration
            const int j = wave * 8 + u;
            const int row = j * RPI + lane / LPR;
            const char* src = base + (long)row * row_pitch + k_off + (lane % LPR) * 16;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(smem + (it & 1) * 32768 + j * 1024), 16, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(src);
                *reinterpret_cast<uint4*>(smem + (it & 1) * 32768 + j * 1024 + lane * 16) = v;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep one tile in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<float*>(smem + t * 4);
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int SEG>
void run(const char* name, const char* buf, long window, long pitch, float* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)stream_kernel<MODE, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<MODE, SEG>), dim3(grid), dim3(256), 65536, 0, buf, window, pitch, 100, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, SEG>), dim3(grid), dim3(256), 65536, 0, buf, window, pitch, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 32768;
    printf("%-46s %7.2f TB/s aggregate  %6.1f KB/us per CU\n", name, bytes / ms / 1e9, bytes / grid / ms / 1e6);
}

int main() {
    const long pitch = 1536;                       // bytes per row (K = 768 bf16)
    const long window = 512 * pitch;               // 768 KiB per window, 64 windows = 48 MiB (beyond the L2s, inside the Infinity Cache)
    char* buf; float* sink;
    hipMalloc(&buf, 64 * window + (1 << 20));
    hipMemset(buf, 1, 64 * window + (1 << 20));
    hipMalloc(&sink, 4);
    run<0, 64>("LDS-DMA, 64-B row segments (BK = 32)", buf, window, pitch, sink);
    run<0, 128>("LDS-DMA, 128-B row segments (BK = 64)", buf, window, pitch, sink);
    run<1, 64>("global_load -> VGPR -> ds_write, 64-B segments", buf, window, pitch, sink);
    run<1, 128>("global_load -> VGPR -> ds_write, 128-B segments", buf, window, pitch, sink);
    return 0;
}
