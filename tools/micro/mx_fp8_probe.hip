// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (OCP e4m3 operands, E8M0 block scales) on gfx950: which lane holds which (row, k)
// bytes and which (row, k-block) scale.  Standalone: hipcc --offload-arch=gfx950 -O2 mx_fp8_probe.hip -o mx_fp8_probe && ./mx_fp8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// hypothesis H: lane l -> row (l & 15), k = 32 * (l >> 4) + j for byte j (j = 0..31, dword j / 4, byte j % 4)
// scale: the lane's VGPR byte `opsel` scales its own 32-k block (row l & 15, block l >> 4)
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C, int variant) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    v8i a, b;
    for (int d = 0; d < 8; ++d) {
        uint32_t wa = 0, wb = 0;
        for (int e = 0; e < 4; ++e) {
            int k = variant == 0 ? 32 * g + 4 * d + e : (variant == 1 ? 16 * g + 4 * (d & 3) + e + 64 * (d >> 2) : 8 * g + 4 * (d & 1) + e + 32 * (d >> 1));
            wa |= (uint32_t)A[i * 128 + k] << (8 * e);
            wb |= (uint32_t)B[i * 128 + k] << (8 * e);
        }
        a[d] = (int)wa; b[d] = (int)wb;
    }
    const int sca = sa[i * 4 + g] * 0x01010101, scb = sb[i * 4 + g] * 0x01010101;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sca, 0, scb);
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}

static float e4m3(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128), sa(64), sb(64);
    srand(3);
    for (auto& v : A) { int e = 5 + rand() % 5, m = rand() % 8; v = (uint8_t)(((rand() & 1) << 7) | (e << 3) | m); }
    for (auto& v : B) { int e = 5 + rand() % 5, m = rand() % 8; v = (uint8_t)(((rand() & 1) << 7) | (e << 3) | m); }
    for (int pass = 0; pass < 2; ++pass) {
        for (int x = 0; x < 64; ++x) { sa[x] = pass ? 124 + rand() % 7 : 127; sb[x] = pass ? 125 + rand() % 5 : 127; }
        // reference: A rows (m) x B rows (n), block scales per (row, k / 32)
        std::vector<double> ref(256);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double s = 0;
                for (int k = 0; k < 128; ++k)
                    s += (double)e4m3(A[m * 128 + k]) * ldexp(1.0, sa[m * 4 + k / 32] - 127) * e4m3(B[n * 128 + k]) * ldexp(1.0, sb[n * 4 + k / 32] - 127);
                ref[m * 16 + n] = s;
            }
        uint8_t *dA, *dB, *dsa, *dsb; float* dC;
        hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 1024);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 3; ++variant) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, variant);
            std::vector<float> C(256);
            hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
            // candidate output maps: (x) lane l holds D[row = 4 (l >> 4) + r][col = l & 15] with D = A-operand rows x B-operand rows
            double e0 = 0, e1 = 0, nrm = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (l >> 4) + r, col = l & 15;
                    e0 += fabs(C[l * 4 + r] - ref[row * 16 + col]);     // A rows -> output rows
                    e1 += fabs(C[l * 4 + r] - ref[col * 16 + row]);     // transposed
                    nrm += fabs(ref[row * 16 + col]);
                }
            printf("scales %s, operand layout variant %d: |err| A-rows-as-rows %.4g, transposed %.4g (|ref| %.4g)\n", pass ? "random" : "unit", variant, e0, e1, nrm);
        }
    }
    return 0;
}
