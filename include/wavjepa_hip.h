/* libwavjepa_hip.so -- C ABI of the MI355X-native (gfx950) kernels behind the WavJEPA pre-training step.
 *
 * The reference (labhamlet/wavjepa) has NO native/FFI boundary: its hot path is stock torch.nn modules called from
 * wavjepa/jepa.py.  This header is therefore the boundary a maintainer would bind to replace the ATen ops of that
 * path; each entry cites the reference call site (file:line under /root/reference)
 * whose computation it replaces.  INTEGRATION.md shows the ctypes stub that binds it from the reference side.
 *
 * Conventions
 *   - every entry:  int wj_xxx(const wj_xxx_args*, void* hip_stream)   -> 0 (WJ_OK) or a negative error code;
 *     never throws, never allocates, never synchronises; stream-ordered and re-entrant; caller owns all memory.
 *   - pointers are raw DEVICE pointers; "bf16" = 16-bit brain float, "f32" = IEEE float, "u8" = bool as bytes.
 *   - activations are token-major ("channels-last"): [rows][features], features contiguous.
 *   - row counts / leading dimensions are in ELEMENTS.
 */
#ifndef WAVJEPA_HIP_H
#define WAVJEPA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WJ_ABI_VERSION 16
int wj_abi_version(void);
/* Number of HIP devices visible (0 on a CPU-only host); never initialises a context beyond hipGetDeviceCount. */
int wj_device_count(void);
/* sizeof() of the argument struct named `name` (e.g. "wj_gemm_args"), or -1: lets a binding verify its struct mirror. */
int wj_struct_size(const char* name);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM (bf16 MFMA, fp32 accumulate) with fused epilogues.
 * Replaces nn.Linear fwd/bwd everywhere on the path (jepa.py:394,400,439; nn.TransformerEncoderLayer in_proj /
 * out_proj / linear1 / linear2, types/wavjepa_configs.py:28-47) and nn.Conv1d layers 1..5 as implicit GEMM over the
 * channels-last activation (extractors/audio_feature_extractor.py:66-70).
 *   C[M,N] = opA(A) . opB(B);  a_trans=0: A is [M][K] (K contiguous);  a_trans=1: A is stored [K][M] (M contiguous)
 *                              b_trans=0: B is [N][K] (K contiguous);  b_trans=1: B is stored [K][N] (N contiguous)
 * Requirements: N%8==0, lda%8==0, ldb%8==0, ldc%8==0, A/B/C 16-byte aligned; a_trans also needs M%8==0; K%8==0 unless both operands are
 * col form (wgrad: K = token rows, arbitrary).
 * -----------------------------------------------------------------------------------------------------------*/
enum {
    WJ_EPI_BF16 = 0,          /* C(bf16)  = acc (+ bias[n])                                                  */
    WJ_EPI_BIAS_GELU2 = 1,    /* h = bf16(acc + bias);  C(bf16) = gelu'(h) ;  C2(bf16) = gelu(h)   (linear1 + nn.GELU)  */
    WJ_EPI_MUL_GELU_GRAD = 2, /* C(bf16)  = bf16(acc) * aux(bf16), aux = the gelu'(h) saved by EPI 1 (backward through GELU) */
    WJ_EPI_ADD_F32 = 3,       /* C(f32)   = acc (+ aux(f32))                       (dgrad + residual-stream)   */
    WJ_EPI_ATOMIC_F32 = 4,    /* C(f32)  += alpha * acc   (atomic; split_k >= 1)   (wgrad into the grad buffer) */
    WJ_EPI_CONV_GELU = 5,     /* C(bf16)  = pre = bf16(acc); C2(bf16) = gelu(pre); rows (m % seg_rows) >= seg_valid -> 0 */
    WJ_EPI_BIAS_GELU = 6,     /* C(bf16)  = gelu(bf16(acc + bias))      (linear1 + nn.GELU where no backward follows: teacher) */
    WJ_EPI_BF16_ADD_POS = 8,  /* y = float(bf16(acc + bias)) + aux[(m % seg_rows)][n] (aux = f32 position table [seg_rows][N]);  C2(f32) = y,
                                  C(bf16) = bf16(y): the post-extraction mapper with the position add of jepa.py:394-396 in its epilogue
                                  (SURVEY K8 + K9) instead of a separate pass over the 51 200 x 768 token tensor.  C2 may be NULL. */
    WJ_EPI_MUL_GELU_GRAD_Z = 7 /* C(bf16) = bf16(acc) * gelu'(aux(bf16)), aux = the PRE-activation z at C's own rows / stride (evaluated here):
                                  the sparse conv dgrad (rowmap form, a_trans = 0, b_trans = 1) writes d(pre) of the layer below directly
                                  instead of d(post) + a wj_gelu_bwd_bf16 pass over the same rows */
};
typedef struct {
    const void* A;
    const void* B;
    void* C;
    void* C2;
    const void* bias; /* f32 [N] or NULL */
    const void* aux;
    float* colsum;    /* optional f32 [N]: += column sums of the bf16 output C (EPI_BF16 / EPI_MUL_GELU_GRAD): the bias
                         gradient of the Linear whose output-gradient this GEMM produces, fused instead of a re-read */
    const int32_t* rowmap; /* optional gather list (backward of the conv layers over the rows that carry gradient):
                         (a_trans=0, b_trans=1, EPI_BF16): int32 [M], logical row m of A and of C is storage row rowmap[m];
                         (a_trans=1, b_trans=1, EPI_ATOMIC_F32): int32 [K + 256] (padding readable), logical k of A and
                         of B is storage row rowmap[k].  Other combinations: WJ_ERR_UNSUPPORTED. */
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t a_trans, b_trans;
    int32_t epilogue;
    int32_t split_k;
    int32_t seg_rows, seg_valid;
    float alpha;
    void* workspace;         /* optional scratch, wj_workspace_bytes("wj_gemm_bf16", args) bytes, ZERO-FILLED ONCE by the caller and then */
    int64_t workspace_bytes; /* left to the library.  ONE pair-capable launch in flight per device: the two workgroups of a tile wait for each
                                other's flag, so both roles must become resident -- two such launches at once can fill an XCD with first
                                halves only (the engine hands the scratch to its main-stream launches only).  With it, row-form WJ_EPI_BF16
                                problems of 33-128 output tiles and K >= 1536 (the ragged student's N = 768 linears and dgrads: 117 tiles
                                for 256 CUs) run as K-split PAIRS: two workgroups per tile, half of K each, fp32 partial sums exchanged
                                through the scratch (csrc/gemm.hip).  NULL / too small: one workgroup per tile, as before. */
    int32_t schedule;        /* 0: the library picks the tile / schedule variant for the shape (csrc/gemm.hip: pick_variant).  1 + v: force variant
                                v = 0..6 (4 = the persistent eight-phase kernel, 5 = the row-panel kernel for N = 384, 6 = the persistent kernel
                                with a deferred GELU epilogue, 128 x 256 items); a variant that cannot run the shape falls back (6 -> 4 -> 3 -> 0).
                                Variants 0-3 and 5 give bit-identical outputs (same k order, bias added last); variants 4 and 6 start their
                                accumulators from the bias: a last-place difference of the bf16 output on <= 0.05 % of the elements (4 and 6
                                agree bit for bit).  Per call: the
                                library keeps no selection state (tests and tools/gemm_check.py force each schedule this way). */
    int32_t persist_cus;     /* resident workgroups per XCD of the persistent schedule, 1..32; 0: the default (32 = one per CU, or the
                                WJ_PERSIST_CUS environment variable read once at the first launch).  A data-parallel run (train.py:174-179:
                                DDP over 8 GPUs) passes 28 so that the RCCL channel kernels of the gradient all-reduce find free CUs while a
                                persistent GEMM of the backward is resident -- a 150-KiB-LDS workgroup on every CU leaves them none. */
} wj_gemm_args;
int wj_gemm_bf16(const wj_gemm_args*, void* stream);
/* Optional: tell the library that `stream` is being retired (call before hipStreamDestroy, with none of its GEMMs in flight).  The persistent
 * schedule keeps one set of tile counters per launching stream (64 sets; sets of idle streams are recycled least-recently-used anyway, which
 * asks the runtime whether the old owner is idle): a released set is reused without any query of a handle that may no longer exist.  Host
 * call: no stream argument beyond the handle, returns WJ_OK also for a stream the library has never seen. */
int wj_gemm_release_stream(void* stream);
/* ------------------------------------------------------------------------------------------------------------
 * MX fp8 GEMM (BASELINE config 5; build-defined numerics, the reference has no fp8): C[M,N] = A . B^T with A [M][K], B [N][K] OCP
 * e4m3 bytes (K contiguous, lda / ldb in BYTES, multiples of 16) and one E8M0 scale per 32 consecutive k of every row:
 *   x[r][k] = e4m3(q[r][k]) * 2^(s[r][k / 32] - 127).
 * Scales are stored [K / 128][ld_scale] dwords (dword (kt, r): byte b = scale of k block 4 kt + b of row r; ld_scale >= rows, and
 * the array carries 256 dwords of readable padding after the last row) -- what wj_quantize_mxfp8 writes.  fp32 accumulation on
 * v_mfma_scale_f32_16x16x128_f8f6f4; epilogues WJ_EPI_BF16 (+ bias), WJ_EPI_BIAS_GELU2, WJ_EPI_BIAS_GELU.  K % 256 == 0.
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const void* A;
    const void* B;
    const void* scale_a;
    const void* scale_b;
    void* C;            /* may be NULL for WJ_EPI_BIAS_GELU when q_out is given (the teacher keeps only the fp8 form of gelu(h)) */
    void* C2;
    const float* bias;
    void* q_out;        /* optional, GELU epilogues (N % 128 == 0): gelu(h) also as MX fp8, e4m3 bytes [M][ldc] ...               */
    void* q_scales;     /* ... + block scales [N / 128][ld_q_scale] dwords: the A operand of linear2 straight from linear1       */
    int64_t lda, ldb, ldc;
    int64_t ld_scale_a, ld_scale_b, ld_q_scale;
    int32_t M, N, K;
    int32_t epilogue;
} wj_gemm_fp8_args;
int wj_gemm_mxfp8(const wj_gemm_fp8_args*, void* stream);

/* q (e4m3 bytes, [M][ldq]) and block scales ([K / 128][ld_scale] dwords as above) of a bf16 matrix x [M][ldx], K % 128 == 0.
 * Per 32-element block: s = ceil(log2(amax / 448)) (no element saturates; an all-zero block gets s = 0), q = RNE(x * 2^-s). */
typedef struct {
    const void* x;
    void* q;
    void* scales;
    int64_t ldx, ldq, ld_scale;
    int32_t M, K;
} wj_quantize_fp8_args;
int wj_quantize_mxfp8(const wj_quantize_fp8_args*, void* stream);

/* Grouped weight gradients: for x < n (n <= 8):  C_x[M_x][N_x] (f32) += A_x^T . B_x  with A_x stored [K_x][M_x] (= dY, token-major)
 * and B_x stored [K_x][N_x] (= X, token-major) -- the autograd wgrad of nn.Linear, dW = dY^T X -- in ONE launch with ONE split-K
 * factor chosen for the group (atomic accumulation like WJ_EPI_ATOMIC_F32).  The four wgrads of a transformer layer together fill
 * the chip at a 3-8x smaller split than each alone: 3-4x fewer float-atomic bytes (csrc/gemm.hip).  M, N, lda, ldb % 8 == 0. */
typedef struct {
    const void* A[8];
    const void* B[8];
    void* C[8];
    int64_t lda[8], ldb[8], ldc[8];
    int32_t M[8], N[8], K[8];
    int32_t n;
} wj_wgrad_group_args;
int wj_wgrad_grouped(const wj_wgrad_group_args*, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * LayerNorm (fp32 statistics), optionally fused with the post-norm residual add.
 * Replaces nn.LayerNorm at jepa.py:392 (feature_norms) and norm1/norm2/final norm of every transformer layer
 * (x = norm(x + branch), torch TransformerEncoderLayer post-norm branch).
 *   s = x (+ r);  y = (s - mean) * rstd * gamma + beta
 *   x: f32, or bf16 when x_is_bf16;  r: bf16 or NULL;  outputs y_f32 / y_bf16 / mean / rstd are each optional.
 *   Input row m is read at row (m / in_valid) * in_seg + (m % in_valid) when in_seg > 0 (padded conv token buffer).
 *   With in_chan = S > 1 the source buffer is CHANNEL-major (conv clip index c*N + n, N = M / (S*in_valid)) while the tokens are
 *   clip-major with the channels of a clip back to back (token m = (n*S + c)*in_valid + t, the "B (C S)" flatten of
 *   audio_channel_feature_extractor.py:176-178): row m is read at ((c*N + n) * in_seg + t).
 *   group_stats (optional, f32 [ceil(M / group_rows)][WJ_GROUP_STATS_SPLIT][2], overwritten): partial (sum y, sum y^2) of the
 *   f32 output over every group of group_rows consecutive rows, one pair per quarter of the group; the consumer
 *   (wj_instnorm_mean; teacher targets, jepa.py:244-252) adds the quarters in order.  Plain stores, no float atomics:
 *   forward activations are bit-reproducible from run to run.
 * -----------------------------------------------------------------------------------------------------------*/
#define WJ_GROUP_STATS_SPLIT 4
typedef struct {
    const void* x;
    const void* r;
    const float* gamma;
    const float* beta;
    float* y_f32;
    void* y_bf16;
    float* mean;
    float* rstd;
    float* group_stats;
    void* y_fp8;        /* optional (D % 128 == 0): y as MX fp8 -- e4m3 bytes [M][D] ...                                              */
    void* y_fp8_scales; /* ... + E8M0 block scales [D / 128][ld_fp8_scale] dwords, exactly what wj_quantize_mxfp8 would produce   */
    int64_t ld_fp8_scale; /*   from bf16(y): the next GEMM's A operand without a separate quantisation pass (config 5)             */
    int32_t M, D;
    int32_t x_is_bf16;
    int32_t in_seg, in_valid;
    int32_t group_rows;
    int32_t in_chan;
    float eps;
    int32_t workgroups; /* 0: the full grid.  > 0 (no group_stats / y_fp8 / in_seg): at most this many workgroups of the LEAN form of the kernel
                           (<= 48 VGPRs, gamma / beta re-read per row; bit-identical outputs), which fits beside a persistent GEMM workgroup of
                           another stream on the same CU -- 256 = one per CU: the LayerNorm streams under that GEMM instead of taking turns
                           with it (csrc/norm.hip) */
} wj_ln_fwd_args;
int wj_layernorm_fwd(const wj_ln_fwd_args*, void* stream);

/* Backward of the above.  ds = d(x + r) = LN-backward(dy);  dgamma/dbeta (and dbias = column sums of bf16(ds), the
 * bias gradient of the Linear that produced r) are ACCUMULATED with atomics into f32 buffers.
 *   dy: f32 [M][D] (+ optional second addend dy2: f32, or bf16 with dy2_is_bf16 -- the grad_input a bf16 linear returns, added to the
 *   fp32 residual gradient here instead of by a read-modify-write GEMM epilogue);  outputs ds_f32 (may alias dy) / ds_bf16 optional;
 *   ds_bf16 row m is written at row (m / out_valid) * out_seg + (m % out_valid) when out_seg > 0. */
typedef struct {
    const float* dy;
    const void* dy2;
    const void* x;
    const void* r;
    const float* gamma;
    const float* mean;
    const float* rstd;
    float* ds_f32;
    void* ds_bf16;
    float* dgamma;
    float* dbeta;
    float* dbias;
    float* workspace; /* optional f32 [1536][3][D]: per-workgroup partials (plain stores) folded by a second kernel
                         instead of ~1500 contended atomics per column.  With dgamma = dbeta = dbias = NULL the partials are LEFT there
                         ([wj_ln_bwd_partial_rows(M, D)][3][D]: dgamma | dbeta | dbias) for a later wj_colsum_f32_group */
    int32_t M, D;
    int32_t x_is_bf16;
    int32_t in_seg, in_valid;
    int32_t out_seg, out_valid;
    int32_t chan;     /* S > 1: x (in_seg) and ds_bf16 (out_seg) buffers are channel-major, see wj_ln_fwd_args.in_chan */
    int32_t dy2_is_bf16;
} wj_ln_bwd_args;
int wj_layernorm_bwd(const wj_ln_bwd_args*, void* stream);

/* out[n] += sum_m bf16 X[m][n]   (bias gradients of in_proj / linear1 / mappers; atomic accumulate) */
typedef struct {
    const void* x;
    float* out;
    int64_t ldx;
    int32_t M, N;
} wj_colsum_args;
int wj_colsum_bf16(const wj_colsum_args*, void* stream);
/* same for an f32 matrix (x: f32 [M][N], ldx in elements): folds per-workgroup partial sums */
int wj_colsum_f32(const wj_colsum_args*, void* stream);

/* Deferred folds of parameter-gradient partials: up to WJ_COLSUM_GROUP_MAX matrices in one launch.  Item i: o0/o1/o2[c] += column sums
 * of x[i] (f32 [M][ldx], N columns; columns [0, n_each) -> o0, [n_each, 2 n_each) -> o1, the rest -> o2; NULL outputs are skipped).
 * The LayerNorm and attention backward of one transformer layer leave three such matrices (wj_layernorm_bwd with NULL gradient
 * outputs, wj_attn_bwd with defer_fold): folded per group of layers instead of by 75 launches of ~5 us per step
 * (autograd of nn.LayerNorm / in_proj_bias inside nn.TransformerEncoderLayer, jepa.py:125-131).  N <= 2304, N <= 3 * n_each. */
#define WJ_COLSUM_GROUP_MAX 16
typedef struct {
    const float* x[WJ_COLSUM_GROUP_MAX];
    float* o0[WJ_COLSUM_GROUP_MAX];
    float* o1[WJ_COLSUM_GROUP_MAX];
    float* o2[WJ_COLSUM_GROUP_MAX];
    int64_t ldx[WJ_COLSUM_GROUP_MAX];
    int32_t M[WJ_COLSUM_GROUP_MAX], N[WJ_COLSUM_GROUP_MAX], n_each[WJ_COLSUM_GROUP_MAX];
    int32_t n;
} wj_colsum_group_args;
int wj_colsum_f32_group(const wj_colsum_group_args*, void* stream);
/* rows of partials wj_layernorm_bwd writes for M token rows of width D (its workgroup count); no stream, no device work */
int wj_ln_bwd_partial_rows(int M, int D);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-head self-attention with a key-padding mask, forward and backward.
 * Replaces F.multi_head_attention_forward / scaled_dot_product_attention inside nn.TransformerEncoderLayer for the
 * student (jepa.py:397,452, mask = ctx_masks), the predictor (jepa.py:438, mask = ctx_and_target_masks) and the
 * teacher (jepa.py:256-258, no mask).   qkv: bf16 [B][T][3*H*hd] packed q|k|v;  key_mask: u8 [B][T], nonzero =
 * key NOT attended, or NULL;  out: bf16 [B][T][H*hd];  lse: f32 [B][H][T] (log-sum-exp of scaled scores).
 * hd in {32, 64};  T <= 416 (three instantiations: <= 128, <= 224, <= 416 tokens).  hd = 16 (T <= 224; BASELINE config 1's predictor,
 * 4 heads of 16): computed in the 32-wide geometry with the upper half of the head dimension zero, scale 1/sqrt(16).
 * Ragged form (seq_off != NULL): the B sequences are PACKED back to back, sequence b = rows [seq_off[b], seq_off[b+1])
 * of qkv / out / dout / dqkv, every key attended (key_mask must be NULL), T = upper bound of the lengths, and
 * lse: f32 [rows][H].  This is how the student / predictor run on their visible tokens only: a key-masked query row
 * that nobody reads (jepa.py:399 keeps ~ctx_masks rows; the loss keeps target rows, jepa.py:356) is never computed.
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const void* qkv;
    const uint8_t* key_mask;
    const int32_t* seq_off; /* optional int32 [B+1]: ragged form */
    void* out;
    float* lse;
    int32_t B, T, H, hd;
    int32_t mask_group; /* mask row used for batch b is b / mask_group (>=1); lets [B,T] masks serve B*G batches */
} wj_attn_fwd_args;
int wj_attn_fwd(const wj_attn_fwd_args*, void* stream);

typedef struct {
    const void* qkv;
    const uint8_t* key_mask;
    const int32_t* seq_off; /* optional int32 [B+1]: ragged form */
    const void* out;
    const void* dout; /* bf16 [B][T][H*hd] */
    const float* lse;
    void* dqkv;       /* bf16 [B][T][3*H*hd] */
    float* dbias;     /* optional f32 [3*H*hd]: += column sums of dqkv over all (b, t) = in_proj_bias gradient */
    float* dbias_ws;  /* f32 [B][3*H*hd] scratch, required with dbias: per-(b,h) partials, folded by a second kernel */
    int32_t B, T, H, hd;
    int32_t mask_group;
    int32_t defer_fold; /* 1: leave the partials in dbias_ws (rows = B, width 3*H*hd) for a later wj_colsum_f32_group into dbias */
} wj_attn_bwd_args;
int wj_attn_bwd(const wj_attn_bwd_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Conv layer 0 (C_in x k x stride, no bias) + GroupNorm(C, C) + erf-GELU, channels-last output.
 * Replaces cnn[0] = Conv1d -> Dropout(0) -> GroupNorm(dim, dim) -> GELU (audio_feature_extractor.py:90-96).
 *   audio: bf16 [N][C_in][L];  w: bf16 [C][C_in][k];  gamma/beta: f32 [C];
 *   act: bf16 [N][P][C] (rows >= L_out of every clip are written as 0);  mean/rstd: f32 [N][C] (of the bf16-rounded
 *   conv output over time, biased variance, eps 1e-5).  C % 16 == 0 (the statistics pass runs on the matrix cores); C_in*k in {10, 20}.
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const void* audio;
    const void* w;
    const float* gamma;
    const float* beta;
    void* act;
    float* mean;
    float* rstd;
    float* workspace; /* f32 scratch, wj_workspace_bytes("wj_conv0_gn_gelu_fwd") bytes: [N][C][2] folded (sum, sum of squares),
                         then one partial record per (clip, 1024-step chunk), stored and folded in chunk order (no atomics) */
    float* yx;        /* optional f32 [N][C][C_in*k]: sum_t y_t x_{t,q}  (kept for the backward; NULL on inference)  */
    float* x1;        /* optional f32 [N][C_in*k]:    sum_t x_{t,q}      (both or neither)                            */
    int64_t audio_clip_stride; /* elements between consecutive clips of `audio`; 0 = C_in*L (packed).  One channel c of an
                         [N][C_audio][L] batch as a mono conv (ConvChannelFeatureExtractor, audio_channel_feature_extractor.py:
                         163-172): audio = base + c*L, C_in = 1, audio_clip_stride = C_audio*L. */
    int32_t N, C_in, L, C, k, stride, L_out, P;
    float eps;
} wj_conv0_fwd_args;
int wj_conv0_gn_gelu_fwd(const wj_conv0_fwd_args*, void* stream);

/* Backward: dact bf16 [N][P][C] -> dw f32 [C][C_in][k], dgamma, dbeta (accumulated: += into the caller's gradient buffers).
 * GroupNorm spreads the gradient over the whole time axis, but that part only needs the forward's yx / x1 sums (see
 * csrc/conv0.hip); dact itself is read on the LISTED rows only: rows = int32 global row indices (n*P + t, ascending,
 * grouped by clip), row_off = int32 [N+1] offsets of every clip's rows, max_rows = the longest clip list.  rows == NULL
 * reads every row t < L_out.  workspace: wj_workspace_bytes("wj_conv0_gn_gelu_bwd") bytes for the same N / C / taps / L_out /
 * max_rows (max_rows = 0 sizes for the dense form): folded [N][C][2 + C_in*k] sums + per-(clip, 256-row chunk) partials. */
typedef struct {
    const void* audio;
    const void* w;
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* rstd;
    const void* dact;
    const float* yx;
    const float* x1;
    const int32_t* rows;
    const int32_t* row_off;
    float* dw;
    float* dgamma;
    float* dbeta;
    float* workspace;
    int64_t audio_clip_stride; /* as in wj_conv0_fwd_args */
    int32_t N, C_in, L, C, k, stride, L_out, P;
    int32_t max_rows;
} wj_conv0_bwd_args;
int wj_conv0_gn_gelu_bwd(const wj_conv0_bwd_args*, void* stream);

/* dpre(bf16) = dpost(bf16) * gelu'(pre(bf16)), elementwise over n elements (conv layers 1..5 backward through GELU;
 * rows that are padding hold pre = 0, dpost = 0 and stay 0). */
/* Listed-rows form (rows != NULL; sparse conv backward, DESIGN.md "active rows"): only rows[0..n_rows) of the
 * [.][row_elems] matrices are processed; with clear_dpost the consumed dpost rows are overwritten with zeros. */
typedef struct {
    void* dpost;
    const void* pre;
    void* dpre;
    const int32_t* rows;
    int64_t n;
    int32_t n_rows, row_elems, clear_dpost;
} wj_gelu_bwd_args;
int wj_gelu_bwd_bf16(const wj_gelu_bwd_args*, void* stream);

/* One wave that busy-waits for `ticks` s_memtime ticks.  Not compute: the engine uses two of these to find a second HIP
 * stream that really runs beside the main one (HIP maps streams onto a few hardware queues round-robin; two streams on
 * one queue serialise -- seen as soon as RCCL had created its own streams first). */
typedef struct {
    int64_t ticks;
} wj_spin_args;
int wj_spin(const wj_spin_args*, void* stream);


/* buf[rows[i]][0 .. row_bytes) = 0 for i < n_rows (restores the all-zero state of a sparse gradient buffer) */
typedef struct {
    void* buf;
    const int32_t* rows;
    int32_t n_rows, row_bytes;
} wj_zero_rows_args;
int wj_zero_rows(const wj_zero_rows_args*, void* stream);

/* Conv weight layout helpers (reference layout [C_out][C_in][k] f32  <->  GEMM layouts, bf16).
 *   mode 0: wp[o][kk*C_in + c]            = w[o][c][kk]                       (forward, B row form, K = k*C_in)
 *   mode 1: wd[(v*C_out + o)][c]          = w[o][c][rho + stride*(U-1-v)]     (dgrad phase rho, U taps, B col form)
 *   mode 2: dw[o][c][kk] += dwp[o][kk*C_in + c]   (f32 -> f32, un-permute the wgrad into the parameter gradient) */
typedef struct {
    const void* src;
    void* dst;
    int32_t C_out, C_in, k, stride, rho, U, mode;
} wj_conv_w_args;
int wj_conv_weight_layout(const wj_conv_w_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Token plumbing.
 * -----------------------------------------------------------------------------------------------------------*/
/* y_f32[m][d] = f32(x_bf16[m][d]) + pos[(m % T)][d];  y_bf16 = bf16(y_f32)          (jepa.py:396) */
typedef struct {
    const void* x;
    const float* pos;
    float* y_f32;
    void* y_bf16;
    int32_t M, T, D;
} wj_add_pos_args;
int wj_add_pos(const wj_add_pos_args*, void* stream);

/* Boolean-mask row gather:  out[j] = x[idx[j]] for j < n_rows (idx = flat (b*T+t) of rows with ~ctx_mask, ascending).
 * Pure copy, bit-exact (jepa.py:399).  elem_bytes in {2,4}.  Also the backward of the scatter below. */
typedef struct {
    const void* x;
    const int32_t* idx;
    void* out;
    int32_t n_rows, D, elem_bytes;
} wj_gather_args;
int wj_mask_gather_rows(const wj_gather_args*, void* stream);

/* Predictor input (jepa.py:425-435):  for group g, row (b,t):
 *   tok = inv[b*T+t] >= 0 ? ctx_feats[inv[b*T+t]] : mask_token        (bf16)
 *   out_f32[(b*G+g)][t] = f32(tok) + pos[t];  out_bf16 = bf16(out_f32)
 * Ragged form (rows != NULL): only the n_rows listed rows are produced, packed: out[r] = the row above for the dense
 * index rows[r] = (b*G+g)*T + t (the visible = context-or-target tokens of every (clip, group), ascending). */
typedef struct {
    const void* ctx_feats;   /* bf16 [n_ctx][D] */
    const int32_t* inv;      /* [B*T]: position in ctx_feats or -1 */
    const float* mask_token; /* f32 [D] */
    const float* pos;        /* f32 [T][D] */
    const int32_t* rows;     /* optional int32 [n_rows]: ragged form */
    float* out_f32;
    void* out_bf16;
    int32_t B, T, D, G;
    int32_t n_rows;
} wj_scatter_fill_args;
int wj_mask_scatter_fill_pos(const wj_scatter_fill_args*, void* stream);

/* Backward of the above: dtok[b*T+t] = sum_g d_in[(b*G+g)][t] (f32);  rows with inv >= 0 go to d_ctx_feats (bf16),
 * the others are summed into d_mask_token (f32, atomic) -- or, with `partials` (wj_workspace_bytes bytes: one row of D floats per
 * workgroup, wj_scatter_fill_bwd_partial_rows(B, T) rows), left as partial rows whose column sums the caller adds to d_mask_token
 * (wj_colsum_f32 / wj_colsum_f32_group): no float atomics, bit-reproducible.  G <= 15.
 * Ragged form (rowmap != NULL): d_in is packed; rowmap[(b*G+g)*T + t] = its row for that token or -1 (not visible:
 * contributes nothing). */
typedef struct {
    const float* d_in;
    const int32_t* inv;
    const int32_t* rowmap;   /* optional int32 [B*G*T]: ragged form */
    void* d_ctx_feats;
    float* d_mask_token;     /* NULL: the mask-token gradient is not wanted */
    float* partials;         /* optional, see above (d_mask_token must still be non-NULL to request the gradient) */
    int32_t B, T, D, G;
} wj_scatter_fill_bwd_args;
int wj_mask_scatter_fill_pos_bwd(const wj_scatter_fill_bwd_args*, void* stream);
/* rows of `partials` the launch writes; no stream, no device work */
int wj_scatter_fill_bwd_partial_rows(int B, int T);

/* dst[m][:] = inv[m] >= 0 ? src[inv[m]][:] : 0   for all M rows (dgrad of the gather: the gradient w.r.t. the
 * student encoder output / the local features is zero on non-context rows, jepa.py:399).  src is bf16 (or f32 when
 * src_is_f32), dst is f32 (or bf16 when dst_is_bf16); inv == NULL is the identity (a plain widening / narrowing copy). */
typedef struct {
    const void* src;
    const int32_t* inv;
    void* dst;
    int32_t M, D;
    int32_t src_is_f32, dst_is_bf16;
} wj_unmask_rows_args;
int wj_unmask_rows_f32(const wj_unmask_rows_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Teacher targets and loss.
 * -----------------------------------------------------------------------------------------------------------*/
/* targets[b] (+)= ((x[b] - mean_b) * rsqrt(var_b + eps)) * scale,  statistics JOINT over the T*D elements of sample b
 * (F.instance_norm on the reference's 4-D stack, jepa.py:244-252).  accumulate = 0 overwrites. */
typedef struct {
    const float* x;
    float* targets;
    int32_t B, TD;
    int32_t accumulate;
    float scale, eps;
} wj_instnorm_args;
int wj_instnorm_accumulate(const wj_instnorm_args*, void* stream);

/* The same targets in ONE pass over the K kept layer outputs (K <= 8), given their per-sample (sum, sum of squares):
 *   targets[b] = (1/K) sum_l (x_l[b] - mean_lb) * rsqrt(var_lb + eps),  mean = S1/TD, var = S2/TD - mean^2 (biased).
 * stats: f32 [K][B][WJ_GROUP_STATS_SPLIT][2] as written by wj_layernorm_fwd.group_stats.  Replaces K read-twice +
 * read-modify-write passes. */
typedef struct {
    const float* x0; const float* x1; const float* x2; const float* x3;
    const float* x4; const float* x5; const float* x6; const float* x7;
    const float* stats;
    float* targets;
    int32_t B, TD, K;
    float eps;
} wj_instnorm_mean_args;
int wj_instnorm_mean(const wj_instnorm_mean_args*, void* stream);

/* Masked MSE (jepa.py:335-362).  preds bf16 [B*G][T][D], targets f32 [B][T][D], tgt u8 [B][G][T].
 *   loss[0] = sum_{tgt} mean_d (p - y)^2 / (count + 1e-8);  loss[1] = count.
 *   If dpreds != NULL also writes dpreds (bf16) = tgt ? 2 (p - y) / (D (count + 1e-8)) * gscale : 0.
 * workspace: f32 [2 + B*G*T].
 * Ragged form (rows != NULL): preds / dpreds hold only the n_rows packed rows, row r standing for the dense index
 * rows[r] = (b*G+g)*T + t; tgt and targets stay dense. */
typedef struct {
    const void* preds;
    const float* targets;
    const uint8_t* tgt;
    const int32_t* rows;     /* optional int32 [n_rows]: ragged form */
    float* loss;
    void* dpreds;
    float* workspace;
    const float* gscale_ptr; /* optional DEVICE scalar multiplied into gscale (the upstream d(loss), no host sync) */
    int32_t B, G, T, D;
    int32_t n_rows;
    float gscale;
} wj_mse_args;
int wj_masked_mse(const wj_mse_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Workspace sizes.  The library never allocates: entry points that need scratch take a `workspace` pointer, and this
 * query tells the caller how many BYTES the call described by the SAME argument struct needs (pointers in it are
 * ignored).  `fn` is the entry point's name: "wj_layernorm_bwd", "wj_attn_bwd" (its dbias_ws), "wj_conv0_gn_gelu_fwd",
 * "wj_conv0_gn_gelu_bwd", "wj_masked_mse", "wj_grad_sumsq".  Returns 0 for entry points without scratch, -1 for an unknown
 * name or NULL arguments.
 * -----------------------------------------------------------------------------------------------------------*/
int64_t wj_workspace_bytes(const char* fn, const void* args);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser-side fused kernels over FLAT parameter storage.
 * -----------------------------------------------------------------------------------------------------------*/
/* teacher = r * teacher + (1 - r) * student (f32, jepa.py:193-198); also refreshes teacher_bf16 when non-NULL. */
typedef struct {
    const float* student;
    float* teacher;
    void* teacher_bf16;
    int64_t n;
    float r;
} wj_ema_args;
int wj_ema_update(const wj_ema_args*, void* stream);

/* out[0] = sum of squares of g[0..n) (f32).  workspace f32 [1024].  (torch clip_grad_norm_, train.py:177-178) */
typedef struct {
    const float* g;
    float* out;
    float* workspace;
    int64_t n;
    int32_t accumulate;  /* 0: out[0] = sum g^2;  1: out[0] += sum g^2 (the norm section by section, as the backward declares sections of the
                            flat gradient buffer final -- stream-ordered, so the result is deterministic) */
    int32_t workgroups;  /* 0: up to 1024.  > 0: at most this many (a launch beside the backward's matrix-bound kernels should take the idle
                            HBM, not their CU slots) */
} wj_sumsq_args;
int wj_grad_sumsq(const wj_sumsq_args*, void* stream);

/* AdamW with fused global-norm clipping (torch.optim.AdamW as configured at jepa.py:215-228 + gradient_clip_val=5):
 *   c = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6));  g = c * grad * grad_scale;
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
 * Also refreshes the bf16 shadow copy of p when p_bf16 != NULL.  max_norm <= 0 disables clipping. */
typedef struct {
    float* p;
    float* g;            /* read; with zero_grad also cleared */
    float* m;
    float* v;
    void* p_bf16;
    const float* sumsq;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2, max_norm, grad_scale;
    int32_t workgroups; /* ABI 15.  0: as many as the range fills (<= 8192).  > 0: at most this many (the kernel strides): an update that runs
                         * beside MFMA- / VALU-bound kernels of another stream (the training loop's overlapped update) should take the
                         * idle HBM, not the other kernels' CU slots */
    int32_t zero_grad;  /* ABI 16.  1: g is set to zero behind the read (optimizer.zero_grad() fused into the update: the next backward
                         * accumulates into a buffer that is already clear -- a 444-MB fill less on the step's critical path) */
} wj_adamw_args;
int wj_adamw_step(const wj_adamw_args*, void* stream);

/* Batched bf16 matrix transposes inside two flat buffers with the same layout: for every table entry (element offset, rows, cols,
 * first tile) dst[off + c*rows + r] = src[off + r*cols + c].  rows, cols multiples of 64, offsets multiples of 8; n_tiles = sum of
 * (rows/64)*(cols/64), first tile = running sum.  Keeps W^T shadows of the transformer weights so that the backward's dgrads
 * (autograd of nn.Linear inside nn.TransformerEncoderLayer, jepa.py:125-131) run as row-form GEMMs.  table is a DEVICE pointer. */
typedef struct {
    const void* src;
    void* dst;
    const int64_t* table; /* int64 [n_mats][4] */
    int32_t n_mats, n_tiles;
} wj_transpose_args;
int wj_transpose_bf16(const wj_transpose_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gradient-bucket all-reduce over RCCL (reference train.py:174-179: Lightning strategy="ddp" = NCCL bucket averages).
 * One communicator per process (= per GPU).  wj_rccl_unique_id on ONE rank -> its 128 bytes to every rank over any side channel ->
 * wj_rccl_bucket_allreduce_init on every rank (collective, blocks until all ranks called it) -> per bucket _launch on a stream ->
 * _wait makes the consuming stream wait for the bucket stream.  RCCL is resolved at run time (the copy already loaded in the
 * process, else librccl.so): WJ_ERR_UNSUPPORTED (-3) without it or before init.  The package's default transport remains
 * torch.distributed (backend "nccl" is this same RCCL); WJ_RCCL_DIRECT=1 routes the gradient buckets through these entries.
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const void* unique_id; /* 128 bytes from wj_rccl_unique_id (HOST memory) */
    int32_t rank, world;
} wj_rccl_init_args;
typedef struct {
    float* buf;            /* device, contiguous f32: reduced in place */
    int64_t count;
    int32_t average;       /* 1: mean over the ranks (DDP semantics), 0: sum */
} wj_rccl_launch_args;
typedef struct {
    void* on_stream;       /* the stream the buckets were launched on */
} wj_rccl_wait_args;
int wj_rccl_unique_id(void* out128);
int wj_rccl_bucket_allreduce_init(const wj_rccl_init_args*);
int wj_rccl_bucket_allreduce_launch(const wj_rccl_launch_args*, void* stream);
int wj_rccl_bucket_allreduce_wait(const wj_rccl_wait_args*, void* stream);
int wj_rccl_bucket_allreduce_finalize(void);

/* dst_bf16[i] = bf16(src_f32[i]) */
typedef struct {
    const float* src;
    void* dst;
    int64_t n;
} wj_cast_args;
int wj_cast_f32_to_bf16(const wj_cast_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batch preparation (jepa.py:291-316): crop `length` samples at starts[b][s] of source row b, normalise by the
 * crop's own mean / UNBIASED std over (C, length): (x - mean) / (std + 1e-5), cast to bf16, write at output row
 * perm_inv[b*S+s] (or b*S+s when perm_inv is NULL).   src f32 [B][C][L_full] -> out bf16 [B*S][C][length]
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const float* src;
    const int32_t* starts;
    const int32_t* perm_inv;
    void* out;
    int32_t B, S, C, L_full, length;
} wj_crop_args;
int wj_crop_normalize_bf16(const wj_crop_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Scene augmentation (data_modules/scene_module/generate_scenes_batch.py), fp32.
 *
 * wj_rir_convolve: batched room-impulse-response convolution, "full" convolution cut to the input length
 *   (convolve_with_rir, generate_scenes_batch.py:12-44 = torchaudio fftconvolve(waveform, rir[c])[..., :T] per channel c):
 *       y[b][c][t] = sum_{k < L, k <= t} x[b][t - k] * h[b][c][k]          t < T
 *   x f32 [B][T];  h f32, element (b, c, k) at h[b*h_stride_b + c*h_stride_c + k] (a channel slice of a [B][n][C'][L] stack needs
 *   no copy);  y f32 [B][C][T], overwritten, or added to when `accumulate` (aggregate_noise, :47-71: sum over noise sources).
 *   Uniformly partitioned overlap-save with an in-LDS block FFT (fft_size 8192; 1024 is accepted so that tests can cross many
 *   blocks / partitions at small sizes; 0 = 8192).  workspace: wj_workspace_bytes("wj_rir_convolve", args) bytes.
 *
 * wj_snr_mix: segmental-SNR mixing (add_noise, :108-150):
 *       Ex = sum_{t in [start_b, start_b + length_b)} source^2,  En likewise for noise   (per b, c)
 *       a = sqrt(Ex / (En + 1e-9) * 10^(-snr_b / 10));   out = source + a * noise
 *   source / noise / out f32 [B][C][T] (out may alias source); snr f32 [B] (dB); start / length int32 [B];
 *   workspace: wj_workspace_bytes("wj_snr_mix", args) bytes.  Sums are folded in a fixed order (bit-reproducible).
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const float* x;
    const float* h;
    float* y;
    void* workspace;
    int64_t h_stride_b, h_stride_c;
    int32_t B, C, T, L;
    int32_t accumulate;
    int32_t fft_size;
} wj_rir_conv_args;
int wj_rir_convolve(const wj_rir_conv_args*, void* stream);

typedef struct {
    const float* source;
    const float* noise;
    float* out;
    const float* snr;
    const int32_t* start;
    const int32_t* length;
    float* workspace;
    int32_t B, C, T;
} wj_snr_mix_args;
int wj_snr_mix(const wj_snr_mix_args*, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Denoiser stage (wavjepa/denoiser.py).
 *
 * wj_resample_fir: polyphase FIR resampling with a caller-supplied kernel table, as torchaudio.functional.resample applies its
 *   windowed-sinc kernel (denoiser.py:29-42: 32 kHz -> 16 kHz, kaiser window, lowpass_filter_width 64; WebAudioDataModule.py:50-60):
 *       xpad = [0] * width + x + [0] * (width + orig);   y[b][i * nw + p] = sum_{k < taps} kernel[p][k] * xpad[b][i * orig + k]
 *   for every i * nw + p < L_out (the caller passes L_out = ceil(new * L_in / orig); nw = new / gcd, orig = orig / gcd,
 *   taps = 2 * width + orig).  x f32 [B][L_in], kernel f32 [nw][taps], y f32 [B][L_out].
 *
 * wj_mse_groups: G prediction sets against one target tensor (denoiser.py:350-355):
 *       loss[1 + g] = mean((preds[g] - targets)^2);  loss[0] = sum_g w[g] * loss[1 + g]
 *       dpreds[g] = gscale[0] * w[g] * 2 (preds[g] - targets) / n          (optional; gscale NULL = 1)
 *   preds / dpreds f32 [G][n], targets f32 [n], w f32 [G] (device), G <= 4.  workspace: wj_workspace_bytes("wj_mse_groups").
 * -----------------------------------------------------------------------------------------------------------*/
typedef struct {
    const float* x;
    const float* kernel;
    float* y;
    int32_t B, L_in, L_out;
    int32_t orig, nw, width, taps;
} wj_resample_args;
int wj_resample_fir(const wj_resample_args*, void* stream);

typedef struct {
    const float* preds;
    const float* targets;
    const float* w;
    const float* gscale;
    float* loss;
    float* dpreds;
    float* workspace;
    int64_t n;
    int32_t G;
} wj_mse_groups_args;
int wj_mse_groups(const wj_mse_groups_args*, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVJEPA_HIP_H */
