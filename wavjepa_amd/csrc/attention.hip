// Key-masked multi-head self-attention for short sequences (T <= 224, head dim 32/64; 16 in the 32-wide geometry) on gfx950.
//
// One workgroup (4 waves) per (batch, head).  The whole K/V (forward) or K,V then Q,dO (backward) of that head
// lives in LDS as ONE padded image per matrix ([rows][hd*2+32 bytes]) that serves both the row reads
// (ds_read_b128 -> MFMA operand with k = head dim) and the transposed reads (ds_read_b64_tr_b16 -> MFMA operand
// with k = sequence), conflict-free for both.  Scores are computed TRANSPOSED (S^T = K Q^T) so that the softmax
// reduction runs over registers + two cross-lane shuffles, and the fp32 accumulator tile is converted in place
// to the bf16 B-operand of the following MFMA (O^T = V^T P^T): P never goes through LDS.
//   forward : S^T -> mask/softmax -> O^T, LSE
//   backward: phase A (per query tile)  dQ^T = K^T dS^T        with S^T, dP^T = V dO^T recomputed
//             phase B (per key tile)    dV^T = dO^T P, dK^T = Q^T dS   with S = Q K^T, dP = dO V^T recomputed
#include <stdlib.h>
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int MAX_TILES = 14;  // 16-row tiles of the default instantiation: T <= 224
constexpr int MAX_TILES_LONG = 26;  // T <= 416 (4.01 s clips -> 400 tokens): K + V images of a 64-wide head = 133 KB of LDS
constexpr int NWB64 = 4, NWB32 = 4;   // backward waves per workgroup (more waves measured slower: 339 -> 439 us at hd 64)
constexpr int NWF_LONG = 7;    // forward, T > 128: 13 query tiles over 7 waves (2,2,2,2,2,2,1) instead of 4 (4,3,3,3)
constexpr float LOG2E = 1.4426950408889634f;
constexpr int NWF_SHORT = 4;   // forward, T <= 128 (ragged student / predictor): <= 2 tiles per wave, twice the workgroups per CU

template <int HD> struct Img {
    static constexpr int RS = HD * 2 + 32;  // padded row stride in bytes
    static constexpr int CH = HD / 8;       // 16-B chunks per row
};

// Copy rows [0,T) of TWO [T][ld] bf16 matrices (HD columns each) into their padded LDS images; rows [T,KP) are zero.
// All global loads of a thread are issued before its first LDS store (a load->store loop would serialise one HBM/L2
// round trip per iteration: ~13 of them for a 200 x 64 head).
template <int HD, int NWAVES, int MT = MAX_TILES, int HG = HD>
struct RowRegs {
    static constexpr int CH = Img<HD>::CH, RS = Img<HD>::RS;
    static constexpr int MAXI = (MT * 16 * CH + NWAVES * 64 - 1) / (NWAVES * 64);
    uint4 v0[MAXI], v1[MAXI];

    __device__ __forceinline__ void load(const bf16_t* __restrict__ src0, long ld0, const bf16_t* __restrict__ src1, long ld1, int T) {
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int idx = threadIdx.x + it * (NWAVES * 64);
            const int row = idx / CH, c = idx - row * CH;
            v0[it] = v1[it] = make_uint4(0, 0, 0, 0);
            if (row < T && c * 8 < HG) {     // (HG < HD: a 16-wide head in the 32-wide geometry, its upper half zeros)
                v0[it] = *reinterpret_cast<const uint4*>(src0 + (long)row * ld0 + c * 8);
                v1[it] = *reinterpret_cast<const uint4*>(src1 + (long)row * ld1 + c * 8);
            }
        }
    }
    __device__ __forceinline__ void store(char* img0, char* img1, int KP) const {
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int idx = threadIdx.x + it * (NWAVES * 64);
            const int row = idx / CH, c = idx - row * CH;
            if (row < KP) {
                *reinterpret_cast<uint4*>(img0 + row * RS + c * 16) = v0[it];
                *reinterpret_cast<uint4*>(img1 + row * RS + c * 16) = v1[it];
            }
        }
    }
};

template <int HD, int NWAVES, int MT = MAX_TILES, int HG = HD>
__device__ __forceinline__ void fill_images2(char* img0, const bf16_t* __restrict__ src0, long ld0, char* img1,
                                             const bf16_t* __restrict__ src1, long ld1, int T, int KP) {
    RowRegs<HD, NWAVES, MT, HG> r;
    r.load(src0, ld0, src1, ld1, T);
    r.store(img0, img1, KP);
}

// MFMA operand with k = head-dim: lane (i,g) gets row (rbase+i), d = ks*32 + 8g .. +7, from the LDS image.
template <int HD>
__device__ __forceinline__ bf16x8 row_frag(const char* img, int rbase, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(img + (rbase + i) * Img<HD>::RS + (ks * 4 + g) * 16);
}
// Same operand straight from global memory (each wave needs its own 16 rows exactly once).
__device__ __forceinline__ bf16x8 row_frag_global(const bf16_t* __restrict__ src, long ld, int rbase, int T, int ks, int lane, int hg) {
    const int i = lane & 15, g = lane >> 4;
    const int row = rbase + i;
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = f2bf(0.f);
    if (row < T && ks * 32 + 8 * g < hg) z = *reinterpret_cast<const bf16x8*>(src + (long)row * ld + ks * 32 + 8 * g);
    return z;
}
// MFMA A-operand with k = sequence (32-row chunk c) and rows = head-dim slice [d0, d0+16), matched to a B operand
// built from two accumulator tiles: element j of lane group g is sequence row c*32 + (j<4 ? 4g+j : 16+4g+j-4).
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int c, int d0, int lane) {
    const int i = lane & 15, g = lane >> 4, q = i >> 2, p = i & 3;
    const char* a0 = img + (c * 32 + 4 * g + q) * Img<HD>::RS + ((d0 + 4 * p) << 1);
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a0 + 16 * Img<HD>::RS));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ bf16x8 pack_tiles(f32x4 lo, f32x4 hi) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = f2bf(lo[e]); r[4 + e] = f2bf(hi[e]); }
    return r;
}
__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups (same lane&15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
// Sum over the 16 lanes of a DPP row (same lane >> 4), every lane gets the total: four v_add_f32 with DPP operands (quad swaps,
// then half-row and row mirrors -- once a quad holds its sum in all four lanes any pairing of quads will do) instead of four
// ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------ forward
// MT = most 16-row tiles a sequence may have (14: T <= 224; 8: T <= 128, fewer live registers -> more waves per SIMD)
template <int HD, int NWF, int MT, int HG = HD>
__global__ __launch_bounds__(NWF * 64, MT == 14 ? 4 : (MT == 8 ? (HD == 64 ? 4 : 6) : (MT == 12 ? 4 : 2))) void attn_fwd_kernel(wj_attn_fwd_args a) {
    constexpr int RS = Img<HD>::RS, KS = HD / 32, DT = HG / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, D = H * HG;        // HG: the head width in memory (16 runs in the 32-wide geometry)
    // the heads of one sequence read interleaved 2*HD-byte slices of the same rows: keep them on ONE XCD so that the
    // other half of every 128-B line is an L2 hit (round-robin dispatch would spread them over all eight L2s: PMC showed
    // the hd = 32 predictor fetching 1.8x (fwd) / 2.6x (bwd) its algorithmic bytes)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / H, h = wg - b * H;
    int T = a.T;
    long row0 = (long)b * a.T;
    if (a.seq_off) {                     // ragged: this sequence's rows in the packed buffers
        row0 = a.seq_off[b];
        T = min(a.seq_off[b + 1] - (int)row0, a.T);
    }
    const int nkt = (T + 15) / 16, KP = ((T + 31) / 32) * 32, nch = KP / 32;
    char* kimg = smem;
    char* vimg = smem + KP * RS;
    float* madd = reinterpret_cast<float*>(smem + 2 * KP * RS);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const long ld = 3L * D;
    const bf16_t* base = (const bf16_t*)a.qkv + row0 * ld + h * HG;
    bf16x8 qf[KS];                       // this wave's first query tile: in flight while K / V are staged
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = row_frag_global(base, ld, wave * 16, T, ks, lane, HG);
    fill_images2<HD, NWF, MT, HG>(kimg, base + D, ld, vimg, base + 2 * D, ld, T, KP);
    const uint8_t* km = a.key_mask ? a.key_mask + (long)(b / a.mask_group) * T : nullptr;
    for (int k = threadIdx.x; k < KP; k += blockDim.x)
        madd[k] = (k < T && !(km && km[k])) ? 0.f : -INFINITY;
    __syncthreads();

    // softmax in the exp2 domain on the RAW scores: max over s, then p = exp2(s * (scale * log2 e) - max * (scale * log2 e)) -- one
    // fma + v_exp_f32 per score.  (A wave-uniform branch that skipped the mask on tiles without one put a taken branch between an
    // MFMA and the first VALU read of its result; hipcc left one wait state there and the kernel returned run-dependent sums.)
    const float scale = rsqrtf((float)HG), scale2 = scale * LOG2E;
    for (int qt = wave; qt < nkt; qt += NWF) {
        bf16x8 qn[KS];                   // next tile's fragments: issued now, consumed at the end of this iteration
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qn[ks] = row_frag_global(base, ld, (qt + NWF) * 16, (qt + NWF < nkt) ? T : 0, ks, lane, HG);
        f32x4 s[MT];
        float mx = -INFINITY;
        // key tiles go in PAIRS (one 32-row chunk of the image; the second tile of the last chunk may be all padding: zero K rows
        // under a -inf mask): half the branches, and two independent MFMA chains for the scheduler to interleave
#pragma unroll
        for (int c = 0; c < MT / 2; ++c) {
            s[2 * c] = s[2 * c + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nch) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = 2 * c + u;
                    s[kt] = *reinterpret_cast<const f32x4*>(madd + kt * 16 + 4 * g);   // 0 / -inf: the mask rides in as the MFMA's C operand
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(kimg, kt * 16, ks, lane), qf[ks], s[kt], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[2 * c + u][r]);
            }
        }
        mx = group_max(mx);
        const float msafe = (mx == -INFINITY) ? 0.f : mx;  // fully masked row: all p = 0 (the reference yields NaN)
        const float m2 = msafe * scale2;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MT / 2; ++c) {
            if (c < nch) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[2 * c + u][r], scale2, -m2));
                        s[2 * c + u][r] = p;
                        sum += p;
                    }
            }
        }
        sum = group_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MT / 2; ++c) {
            if (c < nch) {
                // probabilities are normalised BEFORE the bf16 rounding (as a materialised softmax would be)
                const bf16x8 pf = pack_tiles(s[2 * c] * inv, s[2 * c + 1] * inv);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(vimg, c, dt * 16, lane), pf, o[dt], 0, 0, 0);
            }
        }
        const int q = qt * 16 + i;
        if (q < T) {
            bf16_t* op = (bf16_t*)a.out + (row0 + q) * D + h * HG;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[dt][r]);
                if (dt * 16 < HG) *reinterpret_cast<bf16x4*>(op + dt * 16 + 4 * g) = ov;
            }
            if (a.lse && g == 0)
                a.lse[a.seq_off ? (row0 + q) * H + h : ((long)b * H + h) * T + q] = sum > 0.f ? fmaf(msafe, scale, __logf(sum)) : INFINITY;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int HD, int NWB, int MT, int HG = HD>
__global__ __launch_bounds__(NWB * 64, MT > 14 ? 2 : (HD == 64 ? 3 : 4)) void attn_bwd_kernel(wj_attn_bwd_args a) {
    constexpr int RS = Img<HD>::RS, KS = HD / 32, DT = HG / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, D = H * HG;        // HG: the head width in memory (16 runs in the 32-wide geometry)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);   // heads of one sequence on one XCD (see the forward)
    const int b = wg / H, h = wg - b * H;
    int T = a.T;
    long row0 = (long)b * a.T;
    if (a.seq_off) {
        row0 = a.seq_off[b];
        T = min(a.seq_off[b + 1] - (int)row0, a.T);
    }
    const int nt = (T + 15) / 16, KP = ((T + 31) / 32) * 32, nch = KP / 32;
    char* img0 = smem;                // phase A: K      phase B: Q
    char* img1 = smem + KP * RS;      // phase A: V      phase B: dO
    float* lse_s = reinterpret_cast<float*>(smem + 2 * KP * RS);  // [KP]  (+inf for rows >= T)
    float* delta = lse_s + KP;                                     // [KP]
    float* kvalid = delta + KP;                                    // [KP]  1 = key attended, 0 = masked / padding
    float* bsum = kvalid + KP;                                     // [3*HD] column sums of dq | dk | dv (in_proj_bias grad)
    for (int x = threadIdx.x; x < 3 * HD; x += blockDim.x) bsum[x] = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const long ld = 3L * D;
    const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * ld + h * HG;
    const bf16_t* dO = (const bf16_t*)a.dout + row0 * D + h * HG;
    const bf16_t* O = (const bf16_t*)a.out + row0 * D + h * HG;
    bf16_t* dqkv = (bf16_t*)a.dqkv + row0 * ld + h * HG;
    const uint8_t* km = a.key_mask ? a.key_mask + (long)(b / a.mask_group) * T : nullptr;

    fill_images2<HD, NWB, MT, HG>(img0, qkv + D, ld, img1, qkv + 2 * D, ld, T, KP);
    // short sequences: phase B's Q / dO rows are fetched NOW (a few registers per thread) and only parked in LDS once phase A
    // is done with the K / V images -- their global latency hides behind the statistics loop and phase A
    constexpr bool EARLY = MT <= 8 && HD == 32;   // (the 64-wide head has no registers to spare at 3 waves per SIMD)
    RowRegs<HD, NWB, MT, HG> nxt;
    if constexpr (EARLY) nxt.load(qkv, ld, dO, D, T);
    for (int r = threadIdx.x; r < KP; r += blockDim.x) {
        float l = INFINITY, dl = 0.f, kv = 0.f;
        if (r < T) {
            l = a.lse[a.seq_off ? (row0 + r) * H + h : ((long)b * H + h) * T + r];
            kv = (km && km[r]) ? 0.f : 1.f;
#pragma unroll
            for (int c = 0; c < HG / 8; ++c) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(dO + (long)r * D + c * 8);
                const bf16x8 y = *reinterpret_cast<const bf16x8*>(O + (long)r * D + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += bf2f(x[e]) * bf2f(y[e]);
            }
        }
        lse_s[r] = l; delta[r] = dl; kvalid[r] = kv;
    }
    __syncthreads();
    const float scale = rsqrtf((float)HG);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase A: dQ for 16 queries per wave iteration (queries on the lane, keys on the accumulator rows)
    f32x4 csq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) csq[dt] = zero4;
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = row_frag_global(qkv, ld, wave * 16, T, ks, lane, HG);
        dof[ks] = row_frag_global(dO, D, wave * 16, T, ks, lane, HG);
    }
    for (int qt = wave; qt < nt; qt += NWB) {
        bf16x8 qn[KS], don[KS];          // prefetch of the next query tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qn[ks] = row_frag_global(qkv, ld, (qt + NWB) * 16, (qt + NWB < nt) ? T : 0, ks, lane, HG);
            don[ks] = row_frag_global(dO, D, (qt + NWB) * 16, (qt + NWB < nt) ? T : 0, ks, lane, HG);
        }
        const float my_lse = lse_s[qt * 16 + i], my_delta = delta[qt * 16 + i];
        f32x4 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = zero4;
#pragma unroll
        for (int c = 0; c < MT / 2; ++c) {
            if (c < nch) {
                f32x4 ds2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = 2 * c + u;
                    ds2[u] = zero4;
                    if (kt < nt) {
                        f32x4 s = zero4, dp = zero4;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img0, kt * 16, ks, lane), qf[ks], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img1, kt * 16, ks, lane), dof[ks], dp, 0, 0, 0);
                        }
                        const f32x4 kv = *reinterpret_cast<const f32x4*>(kvalid + kt * 16 + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = kv[r] * __expf(s[r] * scale - my_lse);
                            ds2[u][r] = p * (dp[r] - my_delta) * scale;
                        }
                    }
                }
                const bf16x8 dsf = pack_tiles(ds2[0], ds2[1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img0, c, dt * 16, lane), dsf, dq[dt], 0, 0, 0);
            }
        }
        const int q = qt * 16 + i;
        if (q < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ov[r] = f2bf(dq[dt][r]); csq[dt][r] += bf2f(ov[r]); }
                if (dt * 16 < HG) *reinterpret_cast<bf16x4*>(dqkv + (long)q * ld + dt * 16 + 4 * g) = ov;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; }
    }
    if (a.dbias) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = csq[dt][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (i == 0) atomicAdd(bsum + dt * 16 + 4 * g + r, v);
            }
    }
    __syncthreads();
    if constexpr (EARLY) nxt.store(img0, img1, KP);
    else fill_images2<HD, NWB, MT, HG>(img0, qkv, ld, img1, dO, D, T, KP);
    __syncthreads();

    // ---- phase B: dK, dV for 16 keys per wave iteration (keys on the lane, queries on the accumulator rows)
    f32x4 csk[DT], csv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) csk[dt] = csv[dt] = zero4;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = row_frag_global(qkv + D, ld, wave * 16, T, ks, lane, HG);
        vf[ks] = row_frag_global(qkv + 2 * D, ld, wave * 16, T, ks, lane, HG);
    }
    for (int kt = wave; kt < nt; kt += NWB) {
        bf16x8 kn[KS], vn[KS];           // prefetch of the next key tile
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kn[ks] = row_frag_global(qkv + D, ld, (kt + NWB) * 16, (kt + NWB < nt) ? T : 0, ks, lane, HG);
            vn[ks] = row_frag_global(qkv + 2 * D, ld, (kt + NWB) * 16, (kt + NWB < nt) ? T : 0, ks, lane, HG);
        }
        const float my_kv = kvalid[kt * 16 + i];
        f32x4 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dk[dt] = dv[dt] = zero4;
#pragma unroll
        for (int c = 0; c < MT / 2; ++c) {
            if (c < nch) {
                f32x4 p2[2], ds2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qt = 2 * c + u;
                    p2[u] = ds2[u] = zero4;
                    if (qt < nt) {
                        f32x4 s = zero4, dp = zero4;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img0, qt * 16, ks, lane), kf[ks], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img1, qt * 16, ks, lane), vf[ks], dp, 0, 0, 0);
                        }
                        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
                        const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta + qt * 16 + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = my_kv * __expf(s[r] * scale - l4[r]);
                            p2[u][r] = p;
                            ds2[u][r] = p * (dp[r] - d4[r]) * scale;
                        }
                    }
                }
                const bf16x8 pf = pack_tiles(p2[0], p2[1]);
                const bf16x8 dsf = pack_tiles(ds2[0], ds2[1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img1, c, dt * 16, lane), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img0, c, dt * 16, lane), dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        const int key = kt * 16 + i;
        if (key < T) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ok, ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok[r] = f2bf(dk[dt][r]); ov[r] = f2bf(dv[dt][r]);
                    csk[dt][r] += bf2f(ok[r]); csv[dt][r] += bf2f(ov[r]);
                }
                if (dt * 16 < HG) {
                    *reinterpret_cast<bf16x4*>(dqkv + (long)key * ld + D + dt * 16 + 4 * g) = ok;
                    *reinterpret_cast<bf16x4*>(dqkv + (long)key * ld + 2 * D + dt * 16 + 4 * g) = ov;
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
    }
    if (a.dbias) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = csk[dt][r], u = csv[dt][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                u += __shfl_xor(u, 1, 64); u += __shfl_xor(u, 2, 64); u += __shfl_xor(u, 4, 64); u += __shfl_xor(u, 8, 64);
                if (i == 0) {
                    atomicAdd(bsum + HD + dt * 16 + 4 * g + r, v);
                    atomicAdd(bsum + 2 * HD + dt * 16 + 4 * g + r, u);
                }
            }
        __syncthreads();
        // per-(b, h) partials with plain stores; wj_attn_bwd folds the B rows afterwards (atomics from every workgroup
        // into the same 3*D addresses cost 70-80 us per launch)
        for (int x = threadIdx.x; x < 3 * HD; x += blockDim.x) {
            const int part = x / HD, d = x - part * HD;
            if (d < HG) a.dbias_ws[(long)b * 3 * D + part * D + h * HG + d] = bsum[x];
        }
    }
}

// Backward for SHORT sequences (T <= 128: ragged student / predictor), one global round trip per workgroup.
// The general kernel above walks four dependent fetches (K/V images -> statistics rows -> Q/dO fragments -> K/V fragments again),
// each a full HBM/L2 latency with only 4 workgroups per CU to hide it (stamps: a predictor workgroup lives ~20 us for ~2 us of
// issue).  Here every wave fetches, at entry, the MFMA row fragments of ITS OWN <= 2 tiles of K, V, Q, dO and O (lane (i,g) = row i,
// 16-B chunk g: exactly the B-operand layout) plus lse / key mask for those rows, and everything downstream is fed from them:
//   * the K / V fragments are written to the LDS images for phase A and stay in registers as phase B's own-tile operands,
//   * the Q / dO fragments are phase A's own-tile operands and are written to the images once phase A is done,
//   * delta = rowsum(dO . O) falls out of the dO / O fragments with two cross-lane adds.
template <int HD, int NWB, int MT, bool MASKED, int HG = HD>
__global__ __launch_bounds__(NWB * 64, HD == 64 ? 3 : 4) void attn_bwd_frag_kernel(wj_attn_bwd_args a) {
    constexpr int RS = Img<HD>::RS, KS = HD / 32, DT = HG / 16, TPW = MT / NWB;
    static_assert(MT % NWB == 0, "tiles are dealt to waves round-robin");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, D = H * HG;        // HG: the head width in memory (16 runs in the 32-wide geometry)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / H, h = wg - b * H;
    int T = a.T;
    long row0 = (long)b * a.T;
    if (a.seq_off) {
        row0 = a.seq_off[b];
        T = min(a.seq_off[b + 1] - (int)row0, a.T);
    }
    const int nt = (T + 15) / 16, KP = ((T + 31) / 32) * 32, nch = KP / 32;
    char* img0 = smem;                // phase A: K      phase B: Q
    char* img1 = smem + KP * RS;      // phase A: V      phase B: dO
    float* lse_s = reinterpret_cast<float*>(smem + 2 * KP * RS);
    float* delta = lse_s + KP;
    float* kvalid = delta + KP;
    float* bsum = kvalid + KP;
    for (int x = threadIdx.x; x < 3 * HD; x += blockDim.x) bsum[x] = 0.f;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const long ld = 3L * D;
    const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * ld + h * HG;
    const bf16_t* dO = (const bf16_t*)a.dout + row0 * D + h * HG;
    const bf16_t* O = (const bf16_t*)a.out + row0 * D + h * HG;
    bf16_t* dqkv = (bf16_t*)a.dqkv + row0 * ld + h * HG;
    const uint8_t* km = a.key_mask ? a.key_mask + (long)(b / a.mask_group) * T : nullptr;

    bf16x8 kfr[TPW][KS], vfr[TPW][KS], qfr[TPW][KS], dofr[TPW][KS];
    float lse_r[TPW], delta_r[TPW], kv_r[TPW];
    {
        bf16x8 ofr[TPW][KS];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {          // every fetch of the workgroup's life, back to back
            const int rb = (wave + t * NWB) * 16, row = rb + i;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kfr[t][ks] = row_frag_global(qkv + D, ld, rb, T, ks, lane, HG);
                vfr[t][ks] = row_frag_global(qkv + 2 * D, ld, rb, T, ks, lane, HG);
                qfr[t][ks] = row_frag_global(qkv, ld, rb, T, ks, lane, HG);
                dofr[t][ks] = row_frag_global(dO, D, rb, T, ks, lane, HG);
                ofr[t][ks] = row_frag_global(O, D, rb, T, ks, lane, HG);
            }
            lse_r[t] = INFINITY; kv_r[t] = 0.f;
            if (row < T) {
                lse_r[t] = a.lse[a.seq_off ? (row0 + row) * H + h : ((long)b * H + h) * T + row] * LOG2E;
                kv_r[t] = (km && km[row]) ? 0.f : 1.f;
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int rb = (wave + t * NWB) * 16, row = rb + i;
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += bf2f(dofr[t][ks][e]) * bf2f(ofr[t][ks][e]);
            delta_r[t] = group_sum(dl);
            if (rb < KP) {                        // rows [T, KP) carry zeros / +inf (fragments of rows >= T are zero)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    *reinterpret_cast<bf16x8*>(img0 + row * RS + (ks * 4 + g) * 16) = kfr[t][ks];
                    *reinterpret_cast<bf16x8*>(img1 + row * RS + (ks * 4 + g) * 16) = vfr[t][ks];
                }
                if (g == 0) { lse_s[row] = lse_r[t]; delta[row] = delta_r[t]; kvalid[row] = kv_r[t]; }
            }
        }
    }
    __syncthreads();
    // p = exp(s * scale - lse) = exp2(s * (scale * log2 e) - lse * log2 e): one fma + v_exp_f32 per score; the 1/sqrt(hd) of dS is
    // applied once to the dQ / dK accumulators (as the flash kernels do) instead of to every dS element.
    // Without a key mask nothing needs masking at all: K / V rows >= T are zero in the images, so a padding key adds 0 to dQ, and the
    // dK / dV rows of padding keys are never stored; padding QUERIES have lse = +inf, p = 0.
    const float scale = rsqrtf((float)HG), scale2 = scale * LOG2E;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase A: dQ of this wave's query tiles
    f32x4 csq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) csq[dt] = zero4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int qt = wave + t * NWB;
        if (qt < nt) {
            f32x4 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = zero4;
#pragma unroll
            for (int c = 0; c < MT / 2; ++c) {
                if (c < nch) {
                    f32x4 ds2[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = 2 * c + u;
                        ds2[u] = zero4;
                        if (kt < nt) {
                            f32x4 s = zero4, dp = zero4;
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img0, kt * 16, ks, lane), qfr[t][ks], s, 0, 0, 0);
                                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img1, kt * 16, ks, lane), dofr[t][ks], dp, 0, 0, 0);
                            }
                            f32x4 kv;
                            if constexpr (MASKED) kv = *reinterpret_cast<const f32x4*>(kvalid + kt * 16 + 4 * g);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -lse_r[t]));
                                if constexpr (MASKED) p *= kv[r];
                                ds2[u][r] = p * (dp[r] - delta_r[t]);
                            }
                        }
                    }
                    const bf16x8 dsf = pack_tiles(ds2[0], ds2[1]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img0, c, dt * 16, lane), dsf, dq[dt], 0, 0, 0);
                }
            }
            // (one output tile per accumulator set: its last MFMA sits right in front of the loop's exit branch, and hipcc leaves the VALU
            // read behind that branch one wait state short -- tools/mfma_hazard_scan.py)
            if constexpr (HG < HD) asm volatile("s_nop 7" ::: "memory");
            const int q = qt * 16 + i;
            if (q < T) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ov[r] = f2bf(dq[dt][r] * scale); csq[dt][r] += bf2f(ov[r]); }
                    if (dt * 16 < HG) *reinterpret_cast<bf16x4*>(dqkv + (long)q * ld + dt * 16 + 4 * g) = ov;
                }
            }
        }
    }
    if (a.dbias) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = row16_sum(csq[dt][r]);
                if (i == 0) atomicAdd(bsum + dt * 16 + 4 * g + r, v);
            }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int rb = (wave + t * NWB) * 16, row = rb + i;
        if (rb < KP) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                *reinterpret_cast<bf16x8*>(img0 + row * RS + (ks * 4 + g) * 16) = qfr[t][ks];
                *reinterpret_cast<bf16x8*>(img1 + row * RS + (ks * 4 + g) * 16) = dofr[t][ks];
            }
        }
    }
    __syncthreads();

    // ---- phase B: dK, dV of this wave's key tiles
    f32x4 csk[DT], csv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) csk[dt] = csv[dt] = zero4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int kt = wave + t * NWB;
        if (kt < nt) {
            f32x4 dk[DT], dv[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dk[dt] = dv[dt] = zero4;
#pragma unroll
            for (int c = 0; c < MT / 2; ++c) {
                if (c < nch) {
                    f32x4 p2[2], ds2[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qt = 2 * c + u;
                        p2[u] = ds2[u] = zero4;
                        if (qt < nt) {
                            f32x4 s = zero4, dp = zero4;
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img0, qt * 16, ks, lane), kfr[t][ks], s, 0, 0, 0);
                                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<HD>(img1, qt * 16, ks, lane), vfr[t][ks], dp, 0, 0, 0);
                            }
                            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
                            const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta + qt * 16 + 4 * g);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale2, -l4[r]));
                                if constexpr (MASKED) p *= kv_r[t];
                                p2[u][r] = p;
                                ds2[u][r] = p * (dp[r] - d4[r]);
                            }
                        }
                    }
                    const bf16x8 pf = pack_tiles(p2[0], p2[1]);
                    const bf16x8 dsf = pack_tiles(ds2[0], ds2[1]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img1, c, dt * 16, lane), pf, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(img0, c, dt * 16, lane), dsf, dk[dt], 0, 0, 0);
                    }
                }
            }
            if constexpr (HG < HD) asm volatile("s_nop 7" ::: "memory");
            const int key = kt * 16 + i;
            if (key < T) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x4 ok, ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ok[r] = f2bf(dk[dt][r] * scale); ov[r] = f2bf(dv[dt][r]);
                        csk[dt][r] += bf2f(ok[r]); csv[dt][r] += bf2f(ov[r]);
                    }
                    if (dt * 16 < HG) {
                        *reinterpret_cast<bf16x4*>(dqkv + (long)key * ld + D + dt * 16 + 4 * g) = ok;
                        *reinterpret_cast<bf16x4*>(dqkv + (long)key * ld + 2 * D + dt * 16 + 4 * g) = ov;
                    }
                }
            }
        }
    }
    if (a.dbias) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = row16_sum(csk[dt][r]), u = row16_sum(csv[dt][r]);
                if (i == 0) {
                    atomicAdd(bsum + HD + dt * 16 + 4 * g + r, v);
                    atomicAdd(bsum + 2 * HD + dt * 16 + 4 * g + r, u);
                }
            }
        __syncthreads();
        for (int x = threadIdx.x; x < 3 * HD; x += blockDim.x) {
            const int part = x / HD, d = x - part * HD;
            if (d < HG) a.dbias_ws[(long)b * 3 * D + part * D + h * HG + d] = bsum[x];
        }
    }
}

template <typename K>
int set_lds(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1;
}

}  // namespace

extern "C" int wj_attn_fwd(const wj_attn_fwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->qkv || !a->out) return WJ_ERR_ARG;
    if (a->B <= 0 || a->T <= 0 || a->T > MAX_TILES_LONG * 16 || a->H <= 0 || a->mask_group < 1) return WJ_ERR_ARG;
    if (a->seq_off && a->key_mask) return WJ_ERR_ARG;
    if (a->hd != 16 && a->hd != 32 && a->hd != 64) return WJ_ERR_UNSUPPORTED;
    if (a->hd == 16 && a->T > MAX_TILES * 16) return WJ_ERR_UNSUPPORTED;      // (16-wide heads: the tiny configuration, T <= 224)
    const int hdk = a->hd == 16 ? 32 : a->hd;             // a 16-wide head runs in the 32-wide geometry, the upper half of its K dimension zeros
    const int KP = ((a->T + 31) / 32) * 32;
    const int lds = 2 * KP * (hdk * 2 + 32) + KP * 4;
    dim3 grid(a->B * a->H);
    hipStream_t st = (hipStream_t)stream;
    static int once = set_lds(attn_fwd_kernel<64, NWF_LONG, 14>, 2 * 224 * 160 + 224 * 4) | set_lds(attn_fwd_kernel<32, NWF_LONG, 14>, 2 * 224 * 96 + 224 * 4) |
                      set_lds(attn_fwd_kernel<64, NWF_SHORT, 8>, 2 * 128 * 160 + 128 * 4) | set_lds(attn_fwd_kernel<32, NWF_SHORT, 8>, 2 * 128 * 96 + 128 * 4) |
                      set_lds(attn_fwd_kernel<64, NWF_LONG, 26>, 2 * 416 * 160 + 416 * 4) | set_lds(attn_fwd_kernel<32, NWF_LONG, 26>, 2 * 416 * 96 + 416 * 4);
    (void)once;
    const bool shortseq = a->T <= 128;
    if (a->hd == 16) {
        if (shortseq) hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_SHORT, 8, 16>), grid, dim3(NWF_SHORT * 64), lds, st, *a);
        else if (a->T <= 192) hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_SHORT, 12, 16>), grid, dim3(NWF_SHORT * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_LONG, 14, 16>), grid, dim3(NWF_LONG * 64), lds, st, *a);
    } else if (a->T > MAX_TILES * 16) {     // 225 .. 416 tokens
        if (a->hd == 64) hipLaunchKernelGGL((attn_fwd_kernel<64, NWF_LONG, 26>), grid, dim3(NWF_LONG * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_LONG, 26>), grid, dim3(NWF_LONG * 64), lds, st, *a);
    } else if (a->hd == 64) {
        if (shortseq) hipLaunchKernelGGL((attn_fwd_kernel<64, NWF_SHORT, 8>), grid, dim3(NWF_SHORT * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_fwd_kernel<64, NWF_LONG, 14>), grid, dim3(NWF_LONG * 64), lds, st, *a);
    } else {
        // 129 .. 192 tokens at head dim 32: the predictor's longest ragged sequences of a step now and then (one step in eight at the
        // AudioSet mask parameters) -- three tiles per wave in the short-sequence kernel instead of the general one (105 -> ~75 us)
        if (shortseq) hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_SHORT, 8>), grid, dim3(NWF_SHORT * 64), lds, st, *a);
        else if (a->T <= 192) hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_SHORT, 12>), grid, dim3(NWF_SHORT * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_fwd_kernel<32, NWF_LONG, 14>), grid, dim3(NWF_LONG * 64), lds, st, *a);
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_attn_bwd(const wj_attn_bwd_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->qkv || !a->out || !a->dout || !a->lse || !a->dqkv) return WJ_ERR_ARG;
    if (a->B <= 0 || a->T <= 0 || a->T > MAX_TILES_LONG * 16 || a->H <= 0 || a->mask_group < 1) return WJ_ERR_ARG;
    if (a->seq_off && a->key_mask) return WJ_ERR_ARG;
    if (a->hd != 16 && a->hd != 32 && a->hd != 64) return WJ_ERR_UNSUPPORTED;
    if (a->hd == 16 && a->T > MAX_TILES * 16) return WJ_ERR_UNSUPPORTED;
    if (a->dbias && !a->dbias_ws) return WJ_ERR_ARG;
    const int hdk = a->hd == 16 ? 32 : a->hd;
    const int KP = ((a->T + 31) / 32) * 32;
    const int lds = 2 * KP * (hdk * 2 + 32) + 3 * KP * 4 + 3 * hdk * 4;
    dim3 grid(a->B * a->H);
    hipStream_t st = (hipStream_t)stream;
    static int once = set_lds(attn_bwd_kernel<64, NWB64, 14>, 2 * 224 * 160 + 3 * 224 * 4 + 3 * 64 * 4) |
                      set_lds(attn_bwd_kernel<32, NWB32, 14>, 2 * 224 * 96 + 3 * 224 * 4 + 3 * 32 * 4) |
                      set_lds(attn_bwd_kernel<64, NWB64, 26>, 2 * 416 * 160 + 3 * 416 * 4 + 3 * 64 * 4) |
                      set_lds(attn_bwd_kernel<32, NWB32, 26>, 2 * 416 * 96 + 3 * 416 * 4 + 3 * 32 * 4);
    (void)once;
    if (a->hd == 16) {
        if (a->T <= 128) {
            if (a->key_mask) hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 8, true, 16>), grid, dim3(NWB32 * 64), lds, st, *a);
            else hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 8, false, 16>), grid, dim3(NWB32 * 64), lds, st, *a);
        } else if (a->T <= 192) {
            if (a->key_mask) hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 12, true, 16>), grid, dim3(NWB32 * 64), lds, st, *a);
            else hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 12, false, 16>), grid, dim3(NWB32 * 64), lds, st, *a);
        } else {
            hipLaunchKernelGGL((attn_bwd_kernel<32, NWB32, 14, 16>), grid, dim3(NWB32 * 64), lds, st, *a);
        }
    } else if (a->T > MAX_TILES * 16) {     // 225 .. 416 tokens
        if (a->hd == 64) hipLaunchKernelGGL((attn_bwd_kernel<64, NWB64, 26>), grid, dim3(NWB64 * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_bwd_kernel<32, NWB32, 26>), grid, dim3(NWB32 * 64), lds, st, *a);
    } else if (a->T <= 128) {          // ragged student / predictor: at most 8 tiles (6 or 8 waves per workgroup measured 1.5-2x slower)
        static const int frag = wj_lab_env_int("WJ_ATTN_BWD_FRAG", 3);   // bit 0: hd 32, bit 1: hd 64 (A/B switch)
        const bool masked = a->key_mask != nullptr;
        if (a->hd == 64) {
            if (!(frag & 2)) hipLaunchKernelGGL((attn_bwd_kernel<64, NWB64, 8>), grid, dim3(NWB64 * 64), lds, st, *a);
            else if (masked) hipLaunchKernelGGL((attn_bwd_frag_kernel<64, NWB64, 8, true>), grid, dim3(NWB64 * 64), lds, st, *a);
            else hipLaunchKernelGGL((attn_bwd_frag_kernel<64, NWB64, 8, false>), grid, dim3(NWB64 * 64), lds, st, *a);
        } else {
            if (!(frag & 1)) hipLaunchKernelGGL((attn_bwd_kernel<32, NWB32, 8>), grid, dim3(NWB32 * 64), lds, st, *a);
            else if (masked) hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 8, true>), grid, dim3(NWB32 * 64), lds, st, *a);
            else hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 8, false>), grid, dim3(NWB32 * 64), lds, st, *a);
        }
    } else if (a->hd == 64) {
        hipLaunchKernelGGL((attn_bwd_kernel<64, NWB64, 14>), grid, dim3(NWB64 * 64), lds, st, *a);
    } else if (a->T <= 192) {          // 129 .. 192 tokens, head dim 32: the single-round-trip kernel with three tiles per wave
        if (a->key_mask) hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 12, true>), grid, dim3(NWB32 * 64), lds, st, *a);
        else hipLaunchKernelGGL((attn_bwd_frag_kernel<32, NWB32, 12, false>), grid, dim3(NWB32 * 64), lds, st, *a);
    } else {
        hipLaunchKernelGGL((attn_bwd_kernel<32, NWB32, 14>), grid, dim3(NWB32 * 64), lds, st, *a);
    }
    if (a->dbias && !a->defer_fold) {
        wj_colsum_args c;
        c.x = a->dbias_ws; c.out = a->dbias; c.ldx = 3L * a->H * a->hd; c.M = a->B; c.N = 3 * a->H * a->hd;
        const int rc = wj_colsum_f32(&c, stream);
        if (rc != WJ_OK) return rc;
    }
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
