// bf16 MFMA GEMM for gfx950 with fused epilogues: the dense-contraction engine of the WavJEPA step
// (QKV / out-proj / MLP / mappers / predictor, their dgrad + wgrad, and conv layers 1..5 as implicit GEMM
// over a channels-last activation with overlapping rows: lda = stride*C, K = k*C).
//
//   C[M,N] = opA(A) . opB(B)      fp32 accumulate on v_mfma_f32_16x16x32_bf16
//     a_trans = 0 : A stored [M][K], K contiguous, row stride lda          ("row form")
//     a_trans = 1 : A stored [K][M], M contiguous, row stride lda          ("col form", read with ds_read_b64_tr_b16)
//     b_trans = 0 : B stored [N][K], K contiguous (nn.Linear weight)       ("row form")
//     b_trans = 1 : B stored [K][N], N contiguous                          ("col form")
//   forward  y = x W^T      : (0,0)      dgrad dx = dy W : (0,1)      wgrad dW = dy^T x : (1,1)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), wave tile 64x64 = 4x4 MFMA tiles.  Operands are staged
// global -> VGPR (16 B/lane) -> LDS with the loads of tile t+1 issued before the MFMAs of tile t and the LDS
// write after them (one barrier per K tile, two LDS stages).  Row-form tiles are XOR-swizzled on 16-B chunks
// (conflict-free ds_read_b128), col-form tiles are padded to 288-B rows (conflict-free tr reads).
// The MFMA is issued with the operands swapped (B fragment first) so that every lane ends up owning 4
// consecutive N-columns of one output row: 8-B (bf16) / 16-B (fp32) epilogue stores and float4 bias loads.
// Workgroup ids are remapped so that the N-tiles sharing an A panel run on one XCD (shared L2).
#include "common.h"
#include "../../include/wavjepa_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int ROW_TILE_BYTES = 128 * 128;          // [128 rows][64 k] bf16, 128-B rows
constexpr int COL_ROW_BYTES = 288;                 // [64 k][128 rows] bf16, 256-B rows padded by 32 B
constexpr int COL_TILE_BYTES = 64 * COL_ROW_BYTES;  // 18432
constexpr int OPER_BYTES = COL_TILE_BYTES;         // per operand per stage (max of the two forms)
constexpr int STAGE_BYTES = 2 * OPER_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;          // 73728

struct Chunk4 { uint4 v[4]; };

// ---- global -> registers -------------------------------------------------------------------------------
// Row form: tile rows r0..r0+127 (limit R), k range k0..k0+63 (limit kend).  Thread t: chunk c = t&7, rows (t>>3)+32i.
template <bool TRANS>
__device__ __forceinline__ void load_tile(Chunk4& out, const bf16_t* __restrict__ base, long ld, int r0, int R,
                                          int k0, int kend, int t) {
    if constexpr (!TRANS) {
        const int c = t & 7, rr = t >> 3;
        const int k = k0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + rr + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < R && k < kend) v = *reinterpret_cast<const uint4*>(base + (long)row * ld + k);
            out.v[i] = v;
        }
    } else {
        // Col form: memory rows are k, 128 tile-rows contiguous (16 chunks of 8).  Thread t: chunk c = t&15, k rows (t>>4)+16i.
        const int c = t & 15, kk = t >> 4;
        const int row = r0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kk + 16 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < kend && row < R) v = *reinterpret_cast<const uint4*>(base + (long)k * ld + row);
            out.v[i] = v;
        }
    }
}

// ---- registers -> LDS ----------------------------------------------------------------------------------
template <bool TRANS>
__device__ __forceinline__ void store_tile(char* lds, const Chunk4& in, int t) {
    if constexpr (!TRANS) {
        const int c = t & 7, rr = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 32 * i;
            *reinterpret_cast<uint4*>(lds + row * 128 + ((c ^ (row & 7)) << 4)) = in.v[i];
        }
    } else {
        const int c = t & 15, kk = t >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kk + 16 * i;
            // rows k and k+8 would share banks (8 banks/row shift, period 8): flip the 128-B half for odd k/8
            *reinterpret_cast<uint4*>(lds + k * COL_ROW_BYTES + ((c ^ (((k >> 3) & 1) << 3)) << 4)) = in.v[i];
        }
    }
}

// ---- LDS -> MFMA fragment: lane (i = lane&15, g = lane>>4) gets tile-row (rbase+i), k = ks*32 + 8g + 0..7 ----
template <bool TRANS>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rbase, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if constexpr (!TRANS) {
        const int row = rbase + i;
        const int c = ks * 4 + g;
        return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        // two transposed 4x16 block reads: lane 4q+p of each 16-lane group addresses (k row q, columns 4p..4p+3)
        const int q = i >> 2, p = i & 3;
        const int k = ks * 32 + 8 * g + q;
        const char* a0 = lds + k * COL_ROW_BYTES + (((rbase + 4 * p) << 1) ^ ((g & 1) << 7));
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a0 + 4 * COL_ROW_BYTES));
        bf16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

struct EpiArgs {
    void* C;            // primary output
    void* C2;           // secondary output (EPI_BIAS_GELU2 / EPI_CONV_GELU: post-activation)
    const float* bias;  // [N] or null
    const void* aux;    // EPI_MUL_GELU_GRAD: pre-activation h (bf16, ldc); EPI_ADD_F32: addend (fp32, ldc)
    long ldc;
    int seg_rows;       // EPI_CONV_GELU: rows per clip segment (P) and valid rows (L): rows (m % P) >= L are written as 0
    int seg_valid;
    float alpha;
};

template <int EPI>
__device__ __forceinline__ void epilogue_store(const EpiArgs& e, int m, int n, f32x4 acc, int N) {
    // acc[r] belongs to C[m][n + r], r = 0..3 (n is a multiple of 4, N % 4 == 0)
    if (n >= N) return;
    const long off = (long)m * e.ldc + n;
    if constexpr (EPI == WJ_EPI_BF16 || EPI == WJ_EPI_BIAS_GELU2) {
        if (e.bias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n);
            acc += b;
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[r]);
        *reinterpret_cast<bf16x4*>((bf16_t*)e.C + off) = o;
        if constexpr (EPI == WJ_EPI_BIAS_GELU2) {
            bf16x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = f2bf(gelu_f(bf2f(o[r])));
            *reinterpret_cast<bf16x4*>((bf16_t*)e.C2 + off) = g;
            bf16x4 gp;
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[r] = f2bf(gelu_grad_f(bf2f(o[r])));
            *reinterpret_cast<bf16x4*>((bf16_t*)e.C + off) = gp;
        }
    } else if constexpr (EPI == WJ_EPI_MUL_GELU_GRAD) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>((const bf16_t*)e.aux + off);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(bf2f(f2bf(acc[r])) * bf2f(h[r]));
        *reinterpret_cast<bf16x4*>((bf16_t*)e.C + off) = o;
    } else if constexpr (EPI == WJ_EPI_ADD_F32) {
        f32x4 o = acc;
        if (e.aux) o += *reinterpret_cast<const f32x4*>((const float*)e.aux + off);
        *reinterpret_cast<f32x4*>((float*)e.C + off) = o;
    } else if constexpr (EPI == WJ_EPI_ATOMIC_F32) {
        float* c = (float*)e.C + off;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(c + r, acc[r] * e.alpha);
    } else if constexpr (EPI == WJ_EPI_CONV_GELU) {
        const bool valid = (m % e.seg_rows) < e.seg_valid;
        bf16x4 pre, post;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pre[r] = valid ? f2bf(acc[r]) : f2bf(0.f);
            post[r] = valid ? f2bf(gelu_f(bf2f(pre[r]))) : f2bf(0.f);
        }
        *reinterpret_cast<bf16x4*>((bf16_t*)e.C + off) = pre;
        *reinterpret_cast<bf16x4*>((bf16_t*)e.C2 + off) = post;
    }
}

template <bool ATRANS, bool BTRANS, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                     long lda, long ldb, int M, int N, int K, int tiles_n,
                                                     int split_k, int k_per_split, EpiArgs epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = wg / split_k, ksl = wg - tile * split_k;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = ksl * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nkt > 0) {
        Chunk4 ra, rb;
        load_tile<ATRANS>(ra, A, lda, m0, M, kbeg, kend, t);
        load_tile<BTRANS>(rb, B, ldb, n0, N, kbeg, kend, t);
        store_tile<ATRANS>(smem, ra, t);
        store_tile<BTRANS>(smem + OPER_BYTES, rb, t);
        __syncthreads();

        for (int kt = 0; kt < nkt; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            const bool more = (kt + 1) < nkt;
            if (more) {
                const int k0 = kbeg + (kt + 1) * BK;
                load_tile<ATRANS>(ra, A, lda, m0, M, k0, kend, t);
                load_tile<BTRANS>(rb, B, ldb, n0, N, k0, kend, t);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) af[x] = read_frag<ATRANS>(cur, wm * 64 + x * 16, ks, lane);
#pragma unroll
                for (int x = 0; x < 4; ++x) bfr[x] = read_frag<BTRANS>(cur + OPER_BYTES, wn * 64 + x * 16, ks, lane);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
            }
            if (more) {
                store_tile<ATRANS>(nxt, ra, t);
                store_tile<BTRANS>(nxt + OPER_BYTES, rb, t);
            }
            __syncthreads();
        }
    } else if (EPI == WJ_EPI_ATOMIC_F32) {
        return;  // empty K slice contributes nothing
    }

    // epilogue: lane (i, g) owns C[m0 + wm*64 + mi*16 + i][n0 + wn*64 + ni*16 + 4g .. +3]
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + i;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * g;
            epilogue_store<EPI>(epi, m, n, acc[mi][ni], N);
        }
    }
}

template <bool AT, bool BT, int EPI>
int launch(const wj_gemm_args* a, hipStream_t s) {
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
    int split = a->split_k < 1 ? 1 : a->split_k;
    int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
    split = (a->K + kps - 1) / kps;
    EpiArgs e;
    e.C = a->C; e.C2 = a->C2; e.bias = (const float*)a->bias; e.aux = a->aux; e.ldc = a->ldc;
    e.seg_rows = a->seg_rows > 0 ? a->seg_rows : 1; e.seg_valid = a->seg_rows > 0 ? a->seg_valid : 1;
    e.alpha = a->alpha;
    static bool attr_set = false;
    auto kern = gemm_kernel<AT, BT, EPI>;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int nwg = tiles_m * tiles_n * split;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), LDS_BYTES, s, (const bf16_t*)a->A, (const bf16_t*)a->B, (long)a->lda,
                       (long)a->ldb, a->M, a->N, a->K, tiles_n, split, kps, e);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

template <bool AT, bool BT>
int dispatch_epi(const wj_gemm_args* a, hipStream_t s) {
    switch (a->epilogue) {
        case WJ_EPI_BF16: return launch<AT, BT, WJ_EPI_BF16>(a, s);
        case WJ_EPI_BIAS_GELU2: return launch<AT, BT, WJ_EPI_BIAS_GELU2>(a, s);
        case WJ_EPI_MUL_GELU_GRAD: return launch<AT, BT, WJ_EPI_MUL_GELU_GRAD>(a, s);
        case WJ_EPI_ADD_F32: return launch<AT, BT, WJ_EPI_ADD_F32>(a, s);
        case WJ_EPI_ATOMIC_F32: return launch<AT, BT, WJ_EPI_ATOMIC_F32>(a, s);
        case WJ_EPI_CONV_GELU: return launch<AT, BT, WJ_EPI_CONV_GELU>(a, s);
        default: return WJ_ERR_ARG;
    }
}

}  // namespace

// Generation-1 kernel (128x128 tile, register staging), kept for A/B comparison: set WJ_GEMM_V1=1.
extern "C" int wj_gemm_bf16_v1(const wj_gemm_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (a && a->colsum) return WJ_ERR_UNSUPPORTED;  // the generation-1 kernel has no fused column sums
    if (!a || !a->A || !a->B || !a->C) return WJ_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return WJ_ERR_ARG;
    if ((a->N & 7) || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 3)) return WJ_ERR_ARG;
    if ((!a->a_trans || !a->b_trans) && (a->K & 7)) return WJ_ERR_ARG;  // row-form operands are read in 8-element K chunks
    if (a->a_trans && (a->M & 7)) return WJ_ERR_ARG;
    if ((a->epilogue == WJ_EPI_BIAS_GELU2 || a->epilogue == WJ_EPI_CONV_GELU) && !a->C2) return WJ_ERR_ARG;
    if (a->epilogue == WJ_EPI_MUL_GELU_GRAD && !a->aux) return WJ_ERR_ARG;
    if (a->split_k > 1 && a->epilogue != WJ_EPI_ATOMIC_F32) return WJ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!a->a_trans && !a->b_trans) return dispatch_epi<false, false>(a, s);
    if (!a->a_trans && a->b_trans) return dispatch_epi<false, true>(a, s);
    if (a->a_trans && a->b_trans) return dispatch_epi<true, true>(a, s);
    return dispatch_epi<true, false>(a, s);
}
