// Micro-benchmark: the output side of a GEMM tile on gfx950.  Every workgroup (512 lanes) writes `tiles` output tiles of 256 rows x
// 512 B (bf16, 256 columns) into a row-major matrix of pitch `ldc` bytes, as whole 128-B lines (8 rows x 128 B per wave instruction),
// and waits for the acknowledgements (s_waitcnt vmcnt(0)) after every tile, like an epilogue in front of operand loads does.
// Question: is the ~4.4 us per tile a per-CU limit, or the chip-wide write rate shared by 256 CUs that store in the same instant?
// -> run with 8 ... 256 workgroups and compare the per-workgroup tile time.
// Build: hipcc --offload-arch=gfx950 -O3 store_path.hip -o store_path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int POLICY>
__global__ __launch_bounds__(512, 1) void store_kernel(char* __restrict__ C, long ldc, int tiles_n, int tiles, int grid_tiles, unsigned long long* stamps) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;                         // 2 x 4 waves, 128 x 64 per wave
    u32x4 v = {(unsigned)t, 1u, 2u, 3u};
    unsigned long long t0 = 0;
    for (int i = 0; i < tiles; ++i) {
        if (i == 1 && t == 0) t0 = __builtin_amdgcn_s_memrealtime();
        const int q = (blockIdx.x + i * gridDim.x) % grid_tiles;
        const int tm = q / tiles_n, tn = q % tiles_n;
        char* base = C + ((long)tm * 256 + wm * 128) * ldc + (long)tn * 512 + wn * 128;
#pragma unroll
        for (int s = 0; s < 16; ++s) {                               // 16 x (8 rows x 128 B)
            // POLICY 2: the same 8 rows x 128 B per instruction, but CONSECUTIVE LANES ON DIFFERENT ROWS (lane -> row lane & 7, 16-B chunk
            // lane >> 3), which is what a register epilogue gets from the MFMA result layout without a transpose
            char* p = POLICY == 2 ? base + (long)(s * 8 + (lane & 7)) * ldc + (lane >> 3) * 16
                                  : base + (long)(s * 8 + (lane >> 3)) * ldc + (lane & 7) * 16;
            if (POLICY == 0) *reinterpret_cast<u32x4*>(p) = v;
            else __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
            v.x += 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (t == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x] = t1 - t0;
    }
}

int main() {
    const long M = 51200, N = 2304;                                  // the teacher's qkv output
    const long ldc = N * 2;
    const int tiles_n = N / 256, grid_tiles = (int)(M / 256) * tiles_n;
    char* C;
    unsigned long long* st;
    hipMalloc(&C, M * ldc);
    hipMalloc(&st, 4096 * 8);
    unsigned long long h[4096];
    const int tiles = 33;
    for (int policy = 0; policy < 3; ++policy)
        for (int grid : {8, 16, 32, 64, 128, 256, 512}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (policy == 0) store_kernel<0><<<grid, 512>>>(C, ldc, tiles_n, tiles, grid_tiles, st);
                else if (policy == 1) store_kernel<1><<<grid, 512>>>(C, ldc, tiles_n, tiles, grid_tiles, st);
                else store_kernel<2><<<grid, 512>>>(C, ldc, tiles_n, tiles, grid_tiles, st);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, st, grid * 8, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0;
            for (int i = 0; i < grid; ++i) { const double us = h[i] / 100.0 / (tiles - 1); sum += us; if (us > mx) mx = us; }   // 100 MHz
            printf("%s stores, %3d workgroups: %.2f us per 128-KB tile and workgroup (max %.2f) = %.1f GB/s per CU, %.2f TB/s in all\n",
                   policy == 2 ? "nt, row per lane" : policy ? "nontemporal" : "plain      ", grid, sum / grid, mx, 131072.0 / (sum / grid) * 1e-3, grid * 131072.0 / (sum / grid) * 1e-6);
        }
    return 0;
}
