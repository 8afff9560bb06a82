// Persistent form of the eight-phase bf16 GEMM (variant 4 of wj_gemm_bf16): C[M,N] = A[M,K] . B[N,K]^T, row-form operands,
// K % 128 == 0, M >= 256, N >= 256; epilogues: bias / erf-GELU / conv GELU (forward) and, since round 4, x gelu'(h) with the column sums
// of the result (the backward through linear2 + GELU; the predictor's other dgrads are plain BF16 launches against W^T shadows).
//
// Why.  In the one-tile-per-workgroup kernels (csrc/gemm.hip) a 256 x 256 tile with K = 768 spends half of its life outside the
// MFMA loop: the first operand pieces travel with nothing to hide them, the C tile goes accumulators -> LDS -> global between two
// __syncthreads (7.5-8.6 k cycles, and it needs the ring's LDS), the workgroup retires and the next one is dispatched.  With K = 384
// (the predictor) it is three quarters.  Here 256 workgroups (one per CU) stay resident and walk over the output tiles:
//   * the K-tile stream never stops at a tile boundary: while the last two K tiles of output tile i are multiplied, the LDS-DMA
//     pieces of the first two K tiles of tile i+1 are already on their way (same four-piece, one-piece-per-phase staging and the
//     same counted vmcnt waits as eight_phase_loop in gemm.hip);
//   * LDS-DMA addressing is scalar: `global_load_lds_dwordx4 voffset, sbase` from inline asm, where voffset (row-in-tile * ld +
//     swizzled chunk) is a per-lane CONSTANT of the whole kernel and sbase = matrix + tile origin + k is a wave-uniform SGPR pair
//     advanced by the scalar unit.  The K loop carries 4 address VGPRs instead of 16 and spends no VALU instruction on addresses
//     (while the SIMD partner runs its MFMA cluster the loading wave gets one vector-issue slot per MFMA);
//   * the epilogue works from the accumulator registers (the bias was the C operand of each accumulator's first MFMA): bf16 rounding
//     (-> GELU), then every 16-row block takes one trip through a PER-WAVE strip of LDS that turns the MFMA layout (a lane = a row)
//     into the store layout (8 consecutive lanes = one whole 128-B line of a row): the store path prices a wave store per cache line
//     touched (tools/micro/store_path.hip: 0.6 against 2.3 us per tile and CU).  No barrier: one wave's LDS operations execute in
//     order.  Both wave groups run their epilogues side by side (one extra barrier on each side keeps the loop's one-barrier lag);
//   * the stores are not waited for: the first two K tiles after an epilogue count them into their vmcnt thresholds (they are YOUNGER
//     than the pieces those waits retire); by the end of the second K tile they have drained;
//   * edge tiles are SHIFTED, not clipped: the last tile row / column starts at M - 256 / N - 256 and recomputes a strip its
//     neighbour also writes (same operands, same k order: the same bits).  No clamped rows, no predicated stores, every tile issues
//     the same instruction stream -- which is what lets the vmcnt arithmetic above count on the stores.  Round 4: when N % 256 == 128
//     the last item of a row panel is a HALF-WIDTH one instead (columns [N - 128, N), the B1 quadrants' MFMA clusters skipped): a
//     shifted tile there recomputed and rewrote a third (N = 384) of the launch's columns;
//   * tiles are PULLED: a workgroup's first tile is static (block id), every further one comes from an atomic counter of its XCD
//     label (blockIdx % 8: the tiles of one A panel stay on one L2), fetched a whole tile ahead by lane 0 of wave 0 and passed to the
//     other waves through an LDS word.  Workgroups that start late (a second stream holds their CU) simply pull fewer tiles.  Every
//     workgroup makes exactly one failing pull, so a launch draws exactly chunk_len values from each counter and the pull that draws
//     the last one resets it: no memset between launches, no host-side state besides a stream -> counter-slot table.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "../../include/wavjepa_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int NT = 512;
constexpr unsigned BUF = 65536u, BOFF = 32768u;   // LDS: two parities of [A 256 rows | B 256 rows] x 128 B
constexpr unsigned AUX = 131072u;                 // [2 tiles][8 waves][64 floats] bias of the wave's 64 columns
constexpr unsigned MAILBOX = AUX + 4096u;         // next-next tile index, written by wave 0
constexpr unsigned STAGE = MAILBOX + 256u;          // [8 waves][16 rows x STAGE_ROW B]: the epilogue's transpose (per wave, no barriers)
constexpr unsigned STAGE_ROW = 144u;               // 128 B of bf16 + 16: the 8-byte writes of a 32-lane pass and the 16-byte reads of a row hit distinct banks
constexpr int LDS_TOTAL = (int)(STAGE + 8u * 16u * STAGE_ROW);
constexpr int SLOTS = 64;                         // counter sets (one per stream that launches this kernel)
constexpr int CTR_STRIDE = 32;                    // dwords between the 8 counters of a set (one 128-B line each)

__device__ unsigned g_sched_ctr[SLOTS * 8 * CTR_STRIDE];
__device__ __attribute__((aligned(256))) unsigned char g_zero_bias[256];
constexpr int STAMP_N = 64;
#ifdef WJ_LAB
__device__ unsigned long long g_stamps[2 * 256 * STAMP_N];
#endif   // diagnostic (WJ_PERSIST_STAMPS=1): start, end of prologue, end of every tile

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct PArgs {
    const char* A;
    const char* B;
    char* C;
    char* C2;
    const float* bias;
    const char* aux;          // MUL_GELU_GRAD: gelu'(h), bf16 [M][ldc]
    float* colsum;            // MUL_GELU_GRAD: += column sums of C (the bias gradient of the Linear whose output gradient C is)
    unsigned* ctr;            // this launch's 8 counters
    long ldc_b;               // bytes
    unsigned lda_b, ldb_b;    // bytes
    int M, N, K, ntiles;
    int items_n;              // work items per 256-row panel: full tiles (the last one shifted inwards if N % 256 is neither 0 nor 128) ...
    int half_item;            // ... and, if 1, a last HALF-WIDTH item: columns [N - 128, N), the MFMA clusters of phases 1 / 2 skipped
    int wpx;                  // resident workgroups per XCD (grid = 8 wpx; 32 = every CU)
    int wb_panels, wb_cols;   // W blocking (WJ_PERSIST_WBLOCK): an XCD walks its panels in blocks of wb_panels x wb_cols items (0: panel by panel)
    int nostore;              // diagnostic (WJ_PERSIST_DIAG_NOSTORE=1): the epilogue computes but does not store -- what the store path costs
    int stagger;              // start-up de-phasing: workgroup j of an XCD starts j * stagger / wpx ticks of the 100 MHz clock late
    int seg_rows, seg_valid;
    int active;                   // diagnostic (with stamps): only the first `active` workgroups of every XCD work (32 = all)
    unsigned long long* stamps;   // diagnostic: [256 workgroups][STAMP_N] s_memrealtime values (100 MHz), or NULL
};

// A value hipcc cannot relate to its source: address arithmetic built on it is redone where it is written instead of being
// hoisted out of the tile loop and kept in registers across the MFMA phases (the loop runs at 128 accumulators + 64 fragment
// registers per lane; hoisted tables spill to scratch, and every scratch access is a vmcnt(0) in the LDS-DMA ring).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Counted wait behind an epilogue: the LAGF (after a full-width item) / LAGH (after a half-width item) VMEM operations that epilogue
// issued (stores, the column-sum atomic, the bias DMA) are YOUNGER than the pieces this wait retires and may stay in flight.
template <int BASE, int LAGF, int LAGH>
__device__ __forceinline__ void wait_lag(int lag) {
    if (lag == 0) wait_vmcnt<BASE>();
    else if (lag == 1) wait_vmcnt<BASE + LAGF>();
    else if (lag == 2) wait_vmcnt<BASE + LAGH>();
    else wait_vmcnt<BASE + 1>();                    // diagnostic (no stores issued): only the bias DMA sits in between
}

// One LDS-DMA instruction: 64 lanes x 16 B from sbase + voff (per lane) to LDS bytes [lds_wave + LDS_CONST + 16 lane).
// M0 is written here and nowhere else in this kernel (no builtin LDS-DMA is left in it).
// One dword from LDS byte address `addr` (the kernel's only LDS is the dynamic block at 0).  From asm: a `volatile` C++ read of the mailbox is
// not rewritten to the LDS address space by hipcc -- it became a FLAT load, and a flat load is waited for with vmcnt(0): every item boundary
// drained the epilogue's stores and the staged LDS-DMA pieces that the counted waits of the next K tiles are there to leave in flight
// (found in round 6 in the ISA of the round-3 kernel).
__device__ __forceinline__ unsigned lds_read_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <unsigned LDS_CONST>
__device__ __forceinline__ void dma(unsigned voff, const char* sbase, unsigned lds_wave) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 ::"v"(voff), "s"(sbase), "s"(lds_wave), "n"(LDS_CONST) : "memory", "m0", "scc");
}

// wave-uniform source bases of the next LDS-DMA of each piece (X = A rows 0-63 of each wave row, Y = rows 64-127, B0 = B rows
// 0-31 of each wave column, B1 = rows 32-63), advanced by 128 B per K tile
struct Bases {
    const char* x;
    const char* y;
    const char* b0;
    const char* b1;
};

// One K tile (64 deep) of the stream = four phases; see eight_phase_loop in gemm.hip for the phase / wait structure.
//   last1: this is the last K tile of its output tile (the Y / B1 pieces staged here belong to the next output tile);
//   last2: the next K tile is the last one (B0 / X staged here belong to the next output tile);
//   lag:   first K tile after an epilogue (1: of a full-width item, 2: of a half-width item; 0: none): its LAGF / LAGH VMEM operations
//          (stores, the column-sum atomic, 1 bias DMA) sit between the awaited pieces and the ones issued here.
//   half:  this output item is a half-width one (columns 0-31 of every wave column): the MFMA clusters of phases 1 / 2 (the B1
//          quadrants) are skipped; every LDS-DMA, wait and barrier stays, so the stream's vmcnt arithmetic is the same for both kinds
//          (the B1 pieces of a half item are staged from the B0 rows again: in bounds, never multiplied).
//   FIRST: first PAIR of K tiles of an output tile.  In its PAR == 0 tile every accumulator's first MFMA takes the bias as C (no
//          clearing, no bias add in the epilogue) and nothing is staged in phases 0 / 1: Y and B1 of the second K tile went out
//          BEFORE the previous tile's epilogue, so that the waits of the first K-tile pair only retire operations older than
//          that epilogue's stores (store acknowledgements take ~4 us under load; stores and LDS-DMA share one in-order vmcnt).
//          Its PAR == 1 tile hands the pulled tile index to the other waves.
template <int PAR, int LAGF, int LAGH, bool FIRST>
__device__ __forceinline__ void pp_tile(f32x4 (&acc)[8][4], char* smem, Bases& s, const char* nA, const char* nB, unsigned y_skip,
                                        unsigned b1_skip, const unsigned (&vx)[2], const unsigned (&vb)[2], const unsigned (&dx)[2],
                                        const unsigned (&db)[2], unsigned a_lo, unsigned b_lo, bool last1, bool last2, bool has_next,
                                        int lag, bool half, bool mail, const unsigned& pv, const f32x4 (&bv)[4],
                                        unsigned long long* st = nullptr) {
    constexpr unsigned CUR = PAR * BUF, OTH = (PAR ^ 1) * BUF;
    char* cur = smem + CUR;
    const unsigned a_hi = a_lo ^ 64u, b_hi = b_lo ^ 64u;
    if constexpr (PAR == 0) last1 = false;     // K tiles per output tile are even
    else last2 = false;
    const bool more1 = !last1 || has_next;
    const bool more2 = (!last1 && !last2) || has_next;
    bf16x8 af[8], b0f[4], b1f[4];
    auto lds = [&](unsigned off) { return *reinterpret_cast<const bf16x8*>(cur + off); };
    // ---- phase 0: X, B0 of this tile; stage Y(t+1); wait for B1(t)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b0f[2 * x] = lds(b_lo + x * 2048); b0f[2 * x + 1] = lds(b_hi + x * 2048); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[2 * x] = lds(a_lo + x * 2048); af[2 * x + 1] = lds(a_hi + x * 2048); }
    if constexpr (FIRST && PAR == 0) {
        // Y(t+1) went out before the previous epilogue (or in the prologue).  Younger than the awaited B1(t): B0, X, Y, B1 of t+1
        // (+ the epilogue's stores and the bias DMA)
        wait_lag<8, LAGF, LAGH>(lag);
    } else if (more1) {
        if constexpr (PAR == 1) {
            if (last1) s.y = nA + y_skip;
        }
        dma<OTH + 8192>(vx[0], s.y, dx[0]); dma<OTH + 8192>(vx[1], s.y, dx[1]);
        s.y += 128;
        if constexpr (FIRST) {                 // PAR == 1: the awaited Y(t), B1(t) are OLDER than the epilogue's stores
            wait_lag<6, LAGF, LAGH>(lag);
        } else {
            wait_vmcnt<6>();
        }
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni], af[2 * mi], (FIRST && PAR == 0) ? bv[ni] : acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni + 1], af[2 * mi + 1], acc[mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (FIRST) { if (st && threadIdx.x == 0) st[PAR * 4 + 0] = __builtin_amdgcn_s_memrealtime(); }
    // ---- phase 1: B1 of this tile; stage B1(t+1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b1f[2 * x] = lds(b_lo + 4096 + x * 2048); b1f[2 * x + 1] = lds(b_hi + 4096 + x * 2048); }
    if (more1 && !(FIRST && PAR == 0)) {
        if constexpr (PAR == 1) {
            if (last1) s.b1 = nB + b1_skip;
        }
        dma<OTH + BOFF + 4096>(vb[0], s.b1, db[0]); dma<OTH + BOFF + 4096>(vb[1], s.b1, db[1]);
        s.b1 += 128;
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!half) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni], af[2 * mi], (FIRST && PAR == 0) ? bv[2 + ni] : acc[mi][2 + ni], 0, 0, 0);
                acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni + 1], af[2 * mi + 1], acc[mi][2 + ni], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (FIRST) { if (st && threadIdx.x == 0) st[PAR * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
    // ---- phase 2: Y of this tile; stage B0(t+2) into THIS parity
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[2 * x] = lds(a_lo + 8192 + x * 2048); af[2 * x + 1] = lds(a_hi + 8192 + x * 2048); }
    if (more2) {
        if constexpr (PAR == 0) {
            if (last2) s.b0 = nB;
        }
        dma<CUR + BOFF>(vb[0], s.b0, db[0]); dma<CUR + BOFF>(vb[1], s.b0, db[1]);
        s.b0 += 128;
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!half) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[4 + mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni], af[2 * mi], (FIRST && PAR == 0) ? bv[2 + ni] : acc[4 + mi][2 + ni], 0, 0, 0);
                acc[4 + mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni + 1], af[2 * mi + 1], acc[4 + mi][2 + ni], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (FIRST) { if (st && threadIdx.x == 0) st[PAR * 4 + 2] = __builtin_amdgcn_s_memrealtime(); }
    // ---- phase 3: no fragment reads; stage X(t+2) into THIS parity; wait for X(t+1), B0(t+1)
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
        if constexpr (PAR == 0) {
            if (last2) s.x = nA;
        }
        dma<CUR>(vx[0], s.x, dx[0]); dma<CUR>(vx[1], s.x, dx[1]);
        s.x += 128;
        if constexpr (PAR == 0 && FIRST) {
            wait_lag<8, LAGF, LAGH>(lag);
        } else {
            wait_vmcnt<8>();
        }
    } else if (more1) {
        if constexpr (PAR == 0 && FIRST) {
            wait_lag<4, LAGF, LAGH>(lag);
        } else {
            wait_vmcnt<4>();
        }
    } else {
        wait_vmcnt<0>();
    }
    if constexpr (PAR == 1 && FIRST) {
        if (mail) {
            // The pull went out before this output tile's first K tile; every wait of this phase leaves at most the 8 pieces staged
            // since then in flight, so pv has landed.  (Phase 0's wait does NOT guarantee that: behind an epilogue it tolerates LAG
            // more operations, and in wave 0 the pull is one of them -- with a fast epilogue the stale register reached the mailbox
            // about once in 40 launches, and a tile was computed twice while another was skipped.)
            asm volatile("s_mov_b64 exec, 1\n\ts_nop 0\n\tds_write_b32 %1, %0\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                         ::"v"(pv), "v"(MAILBOX) : "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni], af[2 * mi], (FIRST && PAR == 0) ? bv[ni] : acc[4 + mi][ni], 0, 0, 0);
            acc[4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni + 1], af[2 * mi + 1], acc[4 + mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (FIRST) { if (st && threadIdx.x == 0) st[PAR * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 p;
    p[0] = f2bf(a);
    p[1] = f2bf(b);
    return __builtin_bit_cast(unsigned, p);
}

// VMEM operations one wave's epilogue leaves in flight per output item (global stores, + the column-sum atomic of MUL_GELU_GRAD); a
// half-width item stores 16 rows x 64 B per instruction, i.e. half as many instructions
template <int EPI, bool HALF> struct StoreCount {
    static constexpr int OUTS = (EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_CONV_GELU) ? 2 : 1;
    static constexpr int N = OUTS * (HALF ? 8 : 16) + (EPI == WJ_EPI_MUL_GELU_GRAD ? 1 : 0);
};
// LDS-DMA instructions per wave in the block that precedes an output item's first K tile: the bias.  (Tried for MUL_GELU_GRAD: two more
// that pull one dword of every line of the item's gelu' tile towards the L2 a K loop ahead -- 182.5 against 183.3 us in the step:
// the tile is read at HBM rate either way, and what the epilogue waits for is the chip-wide burst, not the latency.)
template <int EPI> struct TopDmaCount { static constexpr int N = 1; };

// 16-byte non-temporal global store: the C tile is not re-read by this kernel, and kept out of the L2's way its operand panels
// stay resident (measured with tools/persist_stamps.py: 1.30 instead of 1.37 us per K tile on the teacher's QKV shape, and
// 0.6 us less store-acknowledge stall per tile; sc1 / sc0 sc1 write-through forms were slower than plain stores).
__device__ __forceinline__ void store16(char* p, const u32x4& v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}

// 8 bf16 x 8 bf16 -> 8 bf16 (fp32 product, RNE), and the products' fp32 values added to csum (what the stored bf16 values sum to)
__device__ __forceinline__ u32x4 mul_bf16x8(const u32x4& v, const u32x4& g, float (&csum)[8], bool count) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = __uint_as_float(v[j] << 16), a1 = __uint_as_float(v[j] & 0xffff0000u);
        const float g0 = __uint_as_float(g[j] << 16), g1 = __uint_as_float(g[j] & 0xffff0000u);
        const bf16_t p0 = f2bf(a0 * g0), p1 = f2bf(a1 * g1);
        bf16x2 pk;
        pk[0] = p0; pk[1] = p1;
        o[j] = __builtin_bit_cast(unsigned, pk);
        csum[2 * j] += count ? bf2f(p0) : 0.f;
        csum[2 * j + 1] += count ? bf2f(p1) : 0.f;
    }
    return o;
}

// accumulators (bias included: it was the C operand of their first MFMA) -> bf16 (-> GELU) -> transposed through LDS -> 16-byte stores.
// acc[mi][ni][r] = C[m0 + wm*128 + mi*16 + i][n0 + wn*64 + ni*16 + 4 g + r]   (i = lane & 15, g = lane >> 4)
// (half-width item: ni < 2 only, and the wave's columns are n0 + wn*32 + ni*16 + 4 g + r)
// In that layout consecutive lanes hold different ROWS, and a wave store whose consecutive lanes touch different cache lines is handled
// line by line: 2.3 us per 128-KB tile and CU however the lanes are permuted inside the wave, against 0.6 us when every 8 consecutive
// lanes write one whole 128-B line (tools/micro/store_path.hip).  So each 16-row block takes one trip through a per-wave LDS strip:
// four 8-byte writes in the MFMA layout, two 16-byte reads with lane -> (row lane >> 3, 16-B chunk lane & 7), two stores of 8 rows x
// 128 B (half-width: two writes, one read with lane -> (row lane >> 2, chunk lane & 3), one store of 16 rows x 64 B).  One wave's LDS
// operations execute in order, so the strip needs neither waits nor barriers between its uses.
//
// MUL_GELU_GRAD (the backward through linear2 + GELU: C = bf16(acc) * gelu'(h), and the column sums of C = linear1's bias gradient):
// the gelu' tile is read in the STORE layout (whole lines), all of its loads issued before anything else; pass 1 packs and transposes
// the accumulators while they travel, pass 2 multiplies and stores.  The column sums are folded over the wave's 128 rows in registers,
// transposed across the lane groups with seven shuffles (each lane ends with ONE column) and added with one 256-B atomic per wave.
// skip_rows: the first rows of a tile that was shifted inwards at the M edge belong to its neighbour as well -- stored twice with the
// same bits, but counted once.
template <int EPI, bool HALF>
__device__ __forceinline__ void epilogue_regs(f32x4 (&acc)[8][4], char* smem, const PArgs& a, int m0, int n0, int skip_rows, int wave,
                                              int lane) {
    constexpr int NI = HALF ? 2 : 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int ln = opaque(lane);
    const int i = ln & 15, g = ln >> 4;
    char* strip = smem + STAGE + wave * (16 * STAGE_ROW);
    char* wr = strip + i * STAGE_ROW + g * 8;                                  // + ni * 32
    // store layout: full width lane -> (row ln >> 3 [+ 8], 16-B chunk ln & 7); half width lane -> (row ln >> 2, chunk ln & 3)
    const int srow = HALF ? (ln >> 2) : (ln >> 3), schunk = HALF ? (ln & 3) : (ln & 7);
    const char* rd = strip + srow * STAGE_ROW + schunk * 16;                   // + 8 * STAGE_ROW for rows 8-15 (full width)
    const long lane_off = (long)srow * a.ldc_b + (long)(wn * (HALF ? 64 : 128) + schunk * 16);
    const long tile_off = (long)(m0 + wm * 128) * a.ldc_b + (long)n0 * 2;
    char* c1 = a.C + tile_off + lane_off;
    char* c2 = nullptr;
    if constexpr (EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_CONV_GELU) c2 = a.C2 + tile_off + lane_off;
    int rem = 0;                                       // CONV_GELU: row % seg_rows, carried from row to row + 16 (seg_rows > 16)
    if constexpr (EPI == WJ_EPI_CONV_GELU) rem = (m0 + wm * 128 + i) % a.seg_rows;
    const long row8 = 8 * a.ldc_b;
    auto transpose = [&](const u32x2 (&o)[NI], u32x4& lo, u32x4& hi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<u32x2*>(wr + ni * 32) = o[ni];
        // No instruction; keeps the compiler from moving the strip's writes / reads across these points.  The lanes exchange data
        // through the strip without a barrier hipcc knows of: with a branch around the stores (tried for half-width tiles) it sank the
        // second output's strip WRITES into the branch -- in one thread's view their only reader -- and the lanes outside never wrote.
        __builtin_amdgcn_wave_barrier();
        lo = *reinterpret_cast<const u32x4*>(rd);
        if constexpr (!HALF) hi = *reinterpret_cast<const u32x4*>(rd + 8 * STAGE_ROW);
        __builtin_amdgcn_wave_barrier();
    };
    auto through_strip = [&](const u32x2 (&o)[NI], char* dst) {
        u32x4 lo, hi;
        transpose(o, lo, hi);
        if (WJ_LAB_BUILD && a.nostore) {            // diagnostic (lab build): keep the values alive, issue no store
            asm volatile("" ::"v"(lo));
            if constexpr (!HALF) asm volatile("" ::"v"(hi));
            return;
        }
        store16(dst, lo);
        if constexpr (!HALF) store16(dst + row8, hi);
    };
    if constexpr (EPI == WJ_EPI_MUL_GELU_GRAD) {
        const char* ax = a.aux + tile_off + lane_off;
        u32x4 gl[8], gh[8], tl[8], th[8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            gl[mi] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ax + (long)(mi * 16) * a.ldc_b));
            if constexpr (!HALF) gh[mi] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ax + (long)(mi * 16) * a.ldc_b + row8));
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            u32x2 o1[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const f32x4 v = acc[mi][ni];
                o1[ni] = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
            transpose(o1, tl[mi], th[mi]);
        }
        float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int r0 = wm * 128 + srow;                 // tile row of this lane's first output row
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const long off = (long)(mi * 16) * a.ldc_b;
            store16(c1 + off, mul_bf16x8(tl[mi], gl[mi], csum, r0 + mi * 16 >= skip_rows));
            if constexpr (!HALF) store16(c1 + off + row8, mul_bf16x8(th[mi], gh[mi], csum, r0 + mi * 16 + 8 >= skip_rows));
        }
        // csum[x]: this lane's rows of column schunk * 8 + x.  Fold over the lanes that share a chunk (lane bits 3-5 full width, 2-5
        // half width), halving the number of columns a lane carries at every step.
        float c4[4], c2_[2], c1_;
        {
            const bool up = HALF ? (ln & 4) : (ln & 8);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float send = up ? csum[x] : csum[4 + x];
                const float keep = up ? csum[4 + x] : csum[x];
                c4[x] = keep + __shfl_xor(send, HALF ? 4 : 8, 64);
            }
        }
        {
            const bool up = HALF ? (ln & 8) : (ln & 16);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const float send = up ? c4[x] : c4[2 + x];
                const float keep = up ? c4[2 + x] : c4[x];
                c2_[x] = keep + __shfl_xor(send, HALF ? 8 : 16, 64);
            }
        }
        {
            const bool up = HALF ? (ln & 16) : (ln & 32);
            const float send = up ? c2_[0] : c2_[1];
            const float keep = up ? c2_[1] : c2_[0];
            c1_ = keep + __shfl_xor(send, HALF ? 16 : 32, 64);
        }
        int col;
        if constexpr (HALF) {
            c1_ += __shfl_xor(c1_, 32, 64);             // 16 row groups: one more fold, both halves of the wave end with the same sum
            col = schunk * 8 + ((ln >> 2) & 1) * 4 + ((ln >> 3) & 1) * 2 + ((ln >> 4) & 1);
            if (ln < 32) atomicAdd(a.colsum + n0 + wn * 32 + col, c1_);
        } else {
            col = schunk * 8 + ((ln >> 3) & 1) * 4 + ((ln >> 4) & 1) * 2 + ((ln >> 5) & 1);
            atomicAdd(a.colsum + n0 + wn * 64 + col, c1_);
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        bool valid = true;
        if constexpr (EPI == WJ_EPI_CONV_GELU) {
            valid = rem < a.seg_valid;
            rem += 16;
            rem = rem >= a.seg_rows ? rem - a.seg_rows : rem;
        }
        u32x2 o1[NI], o2[NI];                           // [ni]: 4 columns of the first output (C), of the second output (C2)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const f32x4 v = acc[mi][ni];
            if constexpr (EPI == WJ_EPI_BF16) {
                o1[ni] = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            } else {
                // h = bf16(acc + bias), as a bf16 linear returns it; gelu / gelu' of THAT value (nn.GELU on a bf16 tensor)
                f32x2 h[2], gl[2], gp[2];
                h[0] = f32x2{bf2f(f2bf(v[0])), bf2f(f2bf(v[1]))};
                h[1] = f32x2{bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (EPI == WJ_EPI_BIAS_GELU2) gelu_pk<true>(h[q], gl[q], gp[q]);
                    else gelu_pk<false>(h[q], gl[q], gp[q]);
                }
                if constexpr (EPI == WJ_EPI_CONV_GELU) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        h[q].x = valid ? h[q].x : 0.f; h[q].y = valid ? h[q].y : 0.f;
                        gl[q].x = valid ? gl[q].x : 0.f; gl[q].y = valid ? gl[q].y : 0.f;
                    }
                }
                const u32x2 og = u32x2{pack_bf16(gl[0].x, gl[0].y), pack_bf16(gl[1].x, gl[1].y)};
                if constexpr (EPI == WJ_EPI_BIAS_GELU) {
                    o1[ni] = og;
                } else if constexpr (EPI == WJ_EPI_BIAS_GELU2) {
                    o1[ni] = u32x2{pack_bf16(gp[0].x, gp[0].y), pack_bf16(gp[1].x, gp[1].y)};   // C  = gelu'(h)
                    o2[ni] = og;                                                                   // C2 = gelu(h)
                } else {   // CONV_GELU: C = pre, C2 = post
                    o1[ni] = u32x2{pack_bf16(h[0].x, h[0].y), pack_bf16(h[1].x, h[1].y)};
                    o2[ni] = og;
                }
            }
        }
        const long off = (long)(mi * 16) * a.ldc_b;
        through_strip(o1, c1 + off);
        if constexpr (EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_CONV_GELU) through_strip(o2, c2 + off);
    }
}

template <int EPI>
__global__ __launch_bounds__(NT, 1) void gemm_persist_kernel(PArgs a) {
    constexpr int LAGF = StoreCount<EPI, false>::N + TopDmaCount<EPI>::N;   // + the DMAs of the block that precedes an output item's first K tile
    constexpr int LAGH = StoreCount<EPI, true>::N + TopDmaCount<EPI>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n = a.K >> 6;                           // K tiles per output tile (even)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- this workgroup's queue: the logical tile ids [cstart, cstart + clen) of its XCD label (as xcd_remap deals them)
    const int xl = blockIdx.x & 7;
    const int qn = a.ntiles >> 3, qr = a.ntiles & 7;
    const int clen = qn + (xl < qr ? 1 : 0);
    const int cstart = xl < qr ? xl * (qn + 1) : qr * (qn + 1) + (xl - qr) * qn;
    // fewer items than resident workgroups on this XCD label: the spare workgroups leave before they pull, so the counter still sees
    // exactly clen pulls (one failing pull per WORKING workgroup) and the pull that draws clen - 1 resets it
    if ((int)(blockIdx.x >> 3) >= clen) return;
    unsigned* ctr = a.ctr + xl * CTR_STRIDE;
    unsigned pv = 0;
    auto pull = [&]() {
        // lane 0 of wave 0.  A returning atomic from inline asm: its result register is not tracked by hipcc's waitcnt pass (a
        // tracked one would drain the LDS-DMA ring with vmcnt(0) at first use); it is consumed, again from asm, behind a counted
        // wait that covers it (tools/asm_checks.py verifies that nothing touches the register in between).
        if (wave == 0)
            asm volatile("s_mov_b64 exec, 1\n\ts_nop 0\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_mov_b64 exec, -1"
                         : "+v"(pv) : "v"(0u), "v"(1u), "s"(ctr) : "memory");
    };
    int n_stamp = 0;
    auto stamp = [&]() {
        if (WJ_LAB_BUILD && a.stamps && t == 0 && n_stamp < STAMP_N - 2) {
            a.stamps[blockIdx.x * STAMP_N + n_stamp] = __builtin_amdgcn_s_memrealtime();
            if (n_stamp == 1) a.stamps[blockIdx.x * STAMP_N + STAMP_N - 2] = __builtin_amdgcn_s_memtime();      // shader clock after the prologue ...
            a.stamps[blockIdx.x * STAMP_N + STAMP_N - 1] = __builtin_amdgcn_s_memtime();                          // ... and at the latest stamp
        }
        ++n_stamp;
    };
    stamp();

    // ---- piece geometry of this wave.  Two 1-KiB instructions (8 rows x 128 B) per piece: instruction u covers tile rows
    // r_u + lane / 8, lane % 8 = LDS chunk position, holding source chunk (lane % 8) ^ ((row >> 1) & 7) (the swizzle the fragment
    // reads undo).  dx / db: wave-uniform LDS bytes of the instruction inside the A / B region; vx / vb: per-lane source bytes
    // relative to the tile origin (the Y / B1 pieces are 64 / 32 rows further: a scalar term).
    unsigned dx[2], db[2], vx[2], vb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rx = (wave < 4 ? 16 * wave : 128 + 16 * (wave - 4)) + 8 * u;
        const int ib = 16 * wave + 8 * u, rb = (ib >> 5) * 64 + (ib & 31);
        dx[u] = lds0 + (unsigned)rx * 128u;
        db[u] = lds0 + (unsigned)rb * 128u;
        const int rowx = rx + (lane >> 3), rowb = rb + (lane >> 3);
        vx[u] = (unsigned)rowx * a.lda_b + (unsigned)(((lane & 7) ^ ((rowx >> 1) & 7)) * 16);
        vb[u] = (unsigned)rowb * a.ldb_b + (unsigned)(((lane & 7) ^ ((rowb >> 1) & 7)) * 16);
    }
    const unsigned y_skip = 64u * a.lda_b, b1_full = 32u * a.ldb_b;
    // A half-width item covers columns [n0, n0 + 128): wave column wn owns n0 + 32 wn .. + 31, staged where the B0 rows of a full
    // tile go (LDS rows 64 wn .. + 31).  Its per-lane source offsets are those of a full tile minus 32 wn rows: a wave-uniform
    // term, folded into the scalar base.
    const long half_adj = (long)((wave >> 1) * 32) * a.ldb_b;
    auto tile_coords = [&](int q, int& m0, int& n0, bool& half, int& skip) {
        const int L = cstart + q;
        int tm = L / a.items_n;
        int j = L - tm * a.items_n;
        if (a.wb_panels > 0) {
            // blocked walk (the host only asks for it when this XCD's run of items is whole panels): blocks of wb_panels panels, inside a
            // block column groups of wb_cols items, inside a group panel by panel -- the ~32 items an XCD has in flight are then
            // wb_panels x wb_cols instead of 2.7 panels x every column: W is fetched per column group, A per group of a block
            const int pfirst = cstart / a.items_n, npan = clen / a.items_n;
            const int per = a.wb_panels * a.items_n;
            const int lb = q / per, rem = q - lb * per;
            const int psz = min(a.wb_panels, npan - lb * a.wb_panels);
            const int grp = psz * a.wb_cols;
            const int cb = rem / grp, r2 = rem - cb * grp;
            const int pp = r2 / a.wb_cols;
            tm = pfirst + lb * a.wb_panels + pp;
            j = cb * a.wb_cols + (r2 - pp * a.wb_cols);
        }
        m0 = min(tm * 256, a.M - 256);                 // edge tiles are shifted inwards
        skip = tm * 256 - m0;                          // rows of a shifted tile that its neighbour owns
        half = a.half_item && j == a.items_n - 1;
        n0 = half ? a.N - 128 : min(j * 256, a.N - 256);
    };
    const char* bias_src = a.bias ? reinterpret_cast<const char*>(a.bias) : reinterpret_cast<const char*>(g_zero_bias);
    auto bias_dma = [&](int n0, bool half, int slot) {
        // this wave's 64 (half-width item: 32) bias values -> its LDS slot: lanes 0-15 (0-7), 16 B each
        const int ln = opaque(lane);
        const unsigned bias_v = a.bias ? (unsigned)((wn * (half ? 32 : 64) + (ln & 15) * 4) * 4) : (unsigned)((ln & 15) * 16);
        const char* sb = a.bias ? bias_src + (long)n0 * 4 : bias_src;
        const unsigned dst = lds0 + AUX + (unsigned)(slot * 2048 + wave * 256);
        if (half)
            asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xff\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                         ::"v"(bias_v), "s"(sb), "s"(dst) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xffff\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                         ::"v"(bias_v), "s"(sb), "s"(dst) : "memory", "m0");
    };

    f32x4 acc[8][4];                                   // defined by the first K tile of every output tile (C = 0 there)

    if (WJ_LAB_BUILD && a.stamps && (int)(blockIdx.x >> 3) >= a.active) return;   // diagnostic: a partly idle chip (WJ_PERSIST_ACTIVE)
    if (WJ_LAB_BUILD && a.stagger > 0) {
        // All workgroups start together and every item takes the same time, so the whole chip alternates between a K-loop phase
        // (matrix pipe busy, HBM idle) and an epilogue phase (every CU reads / writes its 128-256 KB at once, matrix pipe idle).
        // Spreading the starts over one item time lets one workgroup's epilogue traffic travel while its neighbours multiply.
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)((long)(blockIdx.x >> 3) * a.stagger / a.wpx);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }

    // ---- prologue: first tile is static; pull the second
    int m0, n0, skip0;
    bool half0;
    tile_coords(blockIdx.x >> 3, m0, n0, half0, skip0);
    Bases s;
    s.x = a.A + (long)m0 * a.lda_b;
    s.y = s.x + y_skip;
    s.b0 = a.B + (long)n0 * a.ldb_b - (half0 ? half_adj : 0);
    s.b1 = s.b0 + (half0 ? 0u : b1_full);
    bias_dma(n0, half0, 0);
    pull();
    dma<0>(vx[0], s.x, dx[0]); dma<0>(vx[1], s.x, dx[1]);
    dma<BOFF>(vb[0], s.b0, db[0]); dma<BOFF>(vb[1], s.b0, db[1]);
    dma<8192>(vx[0], s.y, dx[0]); dma<8192>(vx[1], s.y, dx[1]);
    dma<BOFF + 4096>(vb[0], s.b1, db[0]); dma<BOFF + 4096>(vb[1], s.b1, db[1]);
    s.x += 128; s.b0 += 128; s.y += 128; s.b1 += 128;
    dma<BUF + BOFF>(vb[0], s.b0, db[0]); dma<BUF + BOFF>(vb[1], s.b0, db[1]);
    dma<BUF>(vx[0], s.x, dx[0]); dma<BUF>(vx[1], s.x, dx[1]);
    dma<BUF + 8192>(vx[0], s.y, dx[0]); dma<BUF + 8192>(vx[1], s.y, dx[1]);
    dma<BUF + BOFF + 4096>(vb[0], s.b1, db[0]); dma<BUF + BOFF + 4096>(vb[1], s.b1, db[1]);
    s.x += 128; s.b0 += 128; s.y += 128; s.b1 += 128;
    wait_vmcnt<8>();                                   // K tile 0's pieces, the bias and the pull (all older) have landed
    if (wave == 0)
        asm volatile("s_mov_b64 exec, 1\n\ts_nop 0\n\tds_write_b32 %1, %0\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                     ::"v"(pv), "v"(MAILBOX) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    unsigned mail_v = lds_read_u32(MAILBOX);
    mail_v = __builtin_amdgcn_readfirstlane(mail_v);
    if (mail_v == (unsigned)(clen - 1) && t == 0) atomicExch(ctr, 0u);   // the launch's last pull on this counter: reset it
    int q_next = a.wpx + (int)mail_v;
    bool has_next = q_next < clen;

    const int i = lane & 15, g = lane >> 4;
    const unsigned sw = (unsigned)((g ^ ((i >> 1) & 7)) << 4);
    const unsigned a_lo = (unsigned)((wm * 128 + i) * 128) + sw;
    const unsigned b_lo = BOFF + (unsigned)((wn * 64 + i) * 128) + sw;
    stamp();
    if (wm == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind

    int lag = 0;
    int slot = 0;
    int tile_iter = 0;                                // diagnostic (phase stamps)
    for (;;) {
        // ---- before the first K tile of an output tile: where the next one starts, its bias, and the pull for the one after
        int m1 = m0, n1 = n0, skip1 = skip0;
        bool half1 = half0;
        const bool pulled = has_next;
        const char* nA = a.A;
        const char* nB = a.B;
        if (has_next) {
            tile_coords(q_next, m1, n1, half1, skip1);
            nA = a.A + (long)m1 * a.lda_b;
            nB = a.B + (long)n1 * a.ldb_b - (half1 ? half_adj : 0);
        }
        const unsigned b1_skip = half1 ? 0u : b1_full;  // of the NEXT item: where its B1 pieces come from
        __builtin_amdgcn_sched_barrier(0);
        bias_dma(n1, half1, slot ^ 1);                 // always exactly TopDmaCount DMAs here (the LAG counts rely on it)
        if (pulled) pull();
        __builtin_amdgcn_sched_barrier(0);
        f32x4 bv[4];                                   // bias of this lane's 16 columns: the accumulators start from it
        {
            const int g = opaque(lane) >> 4;
            const char* bs = smem + AUX + slot * 2048 + wave * 256;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 64 + g * 16);
        }
        unsigned long long* st = (WJ_LAB_BUILD && a.stamps && tile_iter < 7) ? a.stamps + 256 * STAMP_N + blockIdx.x * STAMP_N + tile_iter * 8 : nullptr;
        ++tile_iter;
        pp_tile<0, LAGF, LAGH, true>(acc, smem, s, nA, nB, y_skip, b1_skip, vx, vb, dx, db, a_lo, b_lo, false, 2 == n, has_next, lag, half0,
                                     false, pv, bv, st);
        pp_tile<1, LAGF, LAGH, true>(acc, smem, s, nA, nB, y_skip, b1_skip, vx, vb, dx, db, a_lo, b_lo, 2 == n, false, has_next, lag, half0,
                                     pulled && wave == 0, pv, bv, st);
        stamp();                                       // diagnostic: end of the first two K tiles
        for (int kt = 2; kt < n; kt += 2) {
            pp_tile<0, LAGF, LAGH, false>(acc, smem, s, nA, nB, y_skip, b1_skip, vx, vb, dx, db, a_lo, b_lo, false, kt + 2 == n, has_next, 0,
                                          half0, false, pv, bv);
            pp_tile<1, LAGF, LAGH, false>(acc, smem, s, nA, nB, y_skip, b1_skip, vx, vb, dx, db, a_lo, b_lo, kt + 2 == n, false, has_next, 0,
                                          half0, false, pv, bv);
        }
        if (has_next) {
            // Y, B1 of the next tile's SECOND K tile, before this tile's stores (see FIRST above).  Their slots (odd parity) were last
            // read in phases 1 / 2 of the K tile just finished, by a wave group that is past those phases whichever group asks.
            dma<BUF + 8192>(vx[0], s.y, dx[0]); dma<BUF + 8192>(vx[1], s.y, dx[1]);
            dma<BUF + BOFF + 4096>(vb[0], s.b1, db[0]); dma<BUF + BOFF + 4096>(vb[1], s.b1, db[1]);
            s.y += 128; s.b1 += 128;
        }
        // ---- epilogue from the registers
        __builtin_amdgcn_sched_barrier(0);
        stamp();                                       // diagnostic: end of the K loop
        // Waves 4-7 run one barrier behind: left alone, waves 0-3 would run their epilogue while 4-7 wait at a barrier, and 4-7 theirs
        // while 0-3 wait at the next one -- two epilogues back to back (measured: 2.0 + 3.0 us per tile).  One extra barrier for waves
        // 0-3 here and one for waves 4-7 behind the epilogue keep the lag and put both epilogues side by side.
        if (wm == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA results of the last phase -> VALU readers behind the loop branch
        if (half0) epilogue_regs<EPI, true>(acc, smem, a, m0, n0, skip0, wave, lane);
        else epilogue_regs<EPI, false>(acc, smem, a, m0, n0, skip0, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 1) __builtin_amdgcn_s_barrier();
        stamp();
        if (!has_next) break;
        lag = (WJ_LAB_BUILD && a.nostore) ? 3 : (half0 ? 2 : 1);         // every item issues all of its stores (edge tiles are shifted, not clipped)
        m0 = m1; n0 = n1; half0 = half1; skip0 = skip1;
        slot ^= 1;
        if (pulled) {
            // the word wave 0 wrote in the second K tile of the tile just finished (>= 2 barriers ago for either wave group; the next
            // write is a K tile away)
            unsigned mv = lds_read_u32(MAILBOX);
            mv = __builtin_amdgcn_readfirstlane(mv);
            if (mv == (unsigned)(clen - 1) && t == 0) atomicExch(ctr, 0u);
            q_next = a.wpx + (int)mv;
            has_next = q_next < clen;
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger
}

struct SlotTable {
    std::mutex mu;
    std::unordered_map<hipStream_t, int> slots;
    hipStream_t owner[SLOTS] = {};
    unsigned long long last_use[SLOTS] = {};      // launch sequence number of the slot's latest launch (LRU order)
    bool released[SLOTS] = {};                    // handed back by wj_gemm_release_stream: free without asking the runtime about its old owner
    unsigned long long seq = 0;
    unsigned* base[32] = {};      // per device: g_sched_ctr is a __device__ symbol, every device has its own copy
    int used = 0;                 // counter sets handed out so far (never-used ones first, then released ones, then idle ones LRU)
};
SlotTable& table() {
    static SlotTable t;
    return t;
}

// Work items of one 256-row panel: N / 256 full tiles, then a half-width item when N % 256 == 128 (WJ_PERSIST_HALF=0: a shifted full
// tile instead, the round-3 behaviour), or a full tile shifted inwards for any other remainder.
struct Items { int items_n, half_item; };
Items items_of(const wj_gemm_args* a) {
    static const int half_ok = wj_lab_env_int("WJ_PERSIST_HALF", 1);
    Items it;
    it.half_item = (half_ok && a->N % 256 == 128) ? 1 : 0;
    it.items_n = it.half_item ? a->N / 256 + 1 : (a->N + 255) / 256;
    return it;
}

// Resident workgroups per XCD: wj_gemm_args.persist_cus (1..32) per call; 0 = the default, 32 = one per CU, or WJ_PERSIST_CUS=n read ONCE
// (thread-safe static initialisation; no state that a later call can change).  A data-parallel run passes a value below 32 so that an RCCL
// channel kernel finds free CUs while a persistent GEMM is resident (trainer.init_persist_cus).
}  // namespace
int wj_gemm_persist_wpx(const wj_gemm_args* a) {
    static const int dflt = [] {
        const char* v = getenv("WJ_PERSIST_CUS");
        const int w = v ? atoi(v) : 32;
        return w < 1 ? 1 : (w > 32 ? 32 : w);
    }();
    return a->persist_cus > 0 ? (a->persist_cus > 32 ? 32 : a->persist_cus) : dflt;
}
namespace {

// Start-up spread in ticks of the 100 MHz clock (lab build only).  WJ_PERSIST_STAGGER_US=<us> applies to every launch; unset: 0.
int persist_stagger(const wj_gemm_args* a) {
    static const int us = wj_lab_env_int("WJ_PERSIST_STAGGER_US", 0);
    static const int only = wj_lab_env_int("WJ_PERSIST_STAGGER_EPI", -1);   // only launches with that epilogue (A/B runs)
    if (only >= 0 && a->epilogue != only) return 0;
    return us * 100;
}

template <int EPI>
int launch_persist(const wj_gemm_args* a, hipStream_t s, unsigned* ctr, int dev) {
    PArgs p;
    p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C; p.C2 = (char*)a->C2; p.bias = (const float*)a->bias;
    p.aux = (const char*)a->aux; p.colsum = a->colsum;
    p.ctr = ctr;
    p.ldc_b = a->ldc * 2; p.lda_b = (unsigned)(a->lda * 2); p.ldb_b = (unsigned)(a->ldb * 2);
    p.M = a->M; p.N = a->N; p.K = a->K;
    const Items it = items_of(a);
    p.items_n = it.items_n; p.half_item = it.half_item;
    p.ntiles = ((a->M + 255) / 256) * p.items_n;
    p.wpx = wj_gemm_persist_wpx(a);
    p.stagger = persist_stagger(a);
    {
        // Blocked tile order for shapes with N >= 2304 whose per-XCD run of items is whole panels and whose item count per panel is a
        // multiple of <cols>: default 8 panels x 4 columns (the teacher's / student's linear1 + GELU, N = 3072: 278 -> 269 us per teacher
        // launch, same-box A/B profiles/r05_ab_wblock.txt; N = 2304 has 9 items per panel and keeps the panel-by-panel walk, where 8x3
        // measured nothing).  WJ_PERSIST_WBLOCK="<panels>x<cols>" overrides, "0" switches it off.
        static int wbp = -1, wbc = 0;
        if (wbp < 0) {
            wbp = 8; wbc = 4;
            const char* v = wj_lab_env_str("WJ_PERSIST_WBLOCK");
            if (v) { int x = 0, y = 0; wbp = 0; if (sscanf(v, "%dx%d", &x, &y) == 2 && x > 0 && y > 0) { wbp = x; wbc = y; } }
        }
        p.wb_panels = 0; p.wb_cols = 0;
        const int panels = (a->M + 255) / 256;
        if (wbp > 0 && a->N >= 2304 && !p.half_item && panels % 8 == 0 && p.items_n % wbc == 0) { p.wb_panels = wbp; p.wb_cols = wbc; }
    }
    {
        static const int ns = wj_lab_env_int("WJ_PERSIST_DIAG_NOSTORE", 0);
        p.nostore = (ns && a->epilogue != WJ_EPI_MUL_GELU_GRAD) ? 1 : 0;
    }
    p.seg_rows = a->seg_rows > 0 ? a->seg_rows : 1;
    p.seg_valid = a->seg_rows > 0 ? a->seg_valid : 1;
    {
        static const int stamps = wj_lab_env_int("WJ_PERSIST_STAMPS", 0);   // lab build, WJ_PERSIST_STAMPS=1: diagnostic time stamps (tools/persist_stamps.py)
        p.stamps = nullptr;
        p.active = 32;
        if (stamps) {
            const char* av = wj_lab_env_str("WJ_PERSIST_ACTIVE");
            if (av) p.active = atoi(av);
#ifdef WJ_LAB
            void* sp = nullptr;
            if (hipGetSymbolAddress(&sp, HIP_SYMBOL(g_stamps)) == hipSuccess) p.stamps = (unsigned long long*)sp;
#endif
        }
    }
    auto kern = gemm_persist_kernel<EPI>;
    static std::atomic<bool> lds_ok[32];            // once per kernel and device
    if (!lds_ok[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return WJ_ERR_LAUNCH;
        lds_ok[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(8 * p.wpx), dim3(NT), LDS_TOTAL, s, p);
    WJ_CHECK_LAUNCH();
    if (p.active < 32) (void)hipMemsetAsync(ctr, 0, 8 * CTR_STRIDE * sizeof(unsigned), s);   // diagnostic: idle workgroups made no pulls, the counters did not wrap
    return WJ_OK;
}

}  // namespace

bool wj_gemm_persist_eligible(const wj_gemm_args* a) {
    if (a->a_trans || a->b_trans || a->rowmap || a->split_k > 1) return false;
    if (a->K < 128 || (a->K % 128) || a->M < 256 || a->N < 256) return false;
    const int e = a->epilogue;
    if (e != WJ_EPI_BF16 && e != WJ_EPI_BIAS_GELU2 && e != WJ_EPI_BIAS_GELU && e != WJ_EPI_CONV_GELU && e != WJ_EPI_MUL_GELU_GRAD) return false;
    if ((e == WJ_EPI_BIAS_GELU2 || e == WJ_EPI_CONV_GELU) && !a->C2) return false;
    if (e == WJ_EPI_CONV_GELU && a->seg_rows > 0 && a->seg_rows <= 16) return false;
    // MUL_GELU_GRAD: the register epilogue always folds the column sums (one atomic per wave and item is part of its vmcnt
    // arithmetic), and columns must not be computed twice (no shifted last tile column)
    if (e == WJ_EPI_MUL_GELU_GRAD) {
        if (!a->aux || !a->colsum || ((uintptr_t)a->aux & 15) || a->bias) return false;
        if (a->N % 256 != 0 && !items_of(a).half_item) return false;      // a shifted last tile column would add its columns twice
    }
    if (e != WJ_EPI_MUL_GELU_GRAD && a->colsum) return false;
    // 16-byte vector stores / LDS-DMA at every tile origin (wj_gemm_bf16 already requires N, lda, ldb, ldc % 8 == 0 and 16-byte
    // aligned A / B / C; stated here as well because the shifted / half-width edge items start at N - 256 / N - 128)
    if ((a->N & 7) || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 7)) return false;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->C2 | (uintptr_t)a->bias) & 15) return false;
    static const int min_tiles = wj_lab_env_int("WJ_PERSIST_MIN_TILES", 256);   // smallest item count that takes the persistent kernel
    const long tiles = (long)((a->M + 255) / 256) * items_of(a).items_n;
    if (tiles < min_tiles) return false;
    if (a->lda * 2 * 256 >= (1l << 31) || a->ldb * 2 * 256 >= (1l << 31)) return false;   // 32-bit per-lane offsets inside a tile
    return true;
}

#ifdef WJ_LAB
// diagnostic: copy the stamp buffer to the host (synchronises the device)
extern "C" int wj_debug_persist_stamps(unsigned long long* out, int n) {
    if (!out || n <= 0) return WJ_ERR_ARG;
    if (n > 2 * 256 * STAMP_N) n = 2 * 256 * STAMP_N;       // second half: per-phase stamps of the first K-tile pair of tiles 0-6
    if (hipDeviceSynchronize() != hipSuccess) return WJ_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8) != hipSuccess) return WJ_ERR_LAUNCH;
    return WJ_OK;
}
#endif

// The caller retires a stream: its counter set goes back to the pool at once (no query of a handle that may be dead by the time another
// stream needs a set).  Safe only when no persistent GEMM of that stream is in flight -- the caller's statement, as destroying the stream is.
extern "C" int wj_gemm_release_stream(void* stream) {
    SlotTable& T = table();
    std::lock_guard<std::mutex> lk(T.mu);
    auto it = T.slots.find((hipStream_t)stream);
    if (it == T.slots.end()) return WJ_OK;                 // never launched a persistent GEMM: nothing to release
    T.owner[it->second] = nullptr;
    T.last_use[it->second] = 0;                            // first in line for the next newcomer
    T.released[it->second] = true;
    T.slots.erase(it);
    return WJ_OK;
}

// The counter set of a stream (8 counters, one 128-B line each, all zero between launches): shared by the persistent kernels of this file
// and of csrc/gemm_pde.hip -- launches of one stream run in order, and every launch leaves its counters at zero.
unsigned* wj_gemm_persist_counters(hipStream_t s, int* dev_out) {
    SlotTable& T = table();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    int slot;
    {
        std::lock_guard<std::mutex> lk(T.mu);
        if (!T.base[dev]) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sched_ctr)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            T.base[dev] = (unsigned*)p;
        }
        auto it = T.slots.find(s);
        if (it == T.slots.end()) {
            int fresh = -1;
            for (int x = 0; x < SLOTS && fresh < 0; ++x)
                if (T.released[x]) { fresh = x; T.released[x] = false; }      // a set its owner gave back (wj_gemm_release_stream)
            if (fresh < 0) fresh = T.used < SLOTS ? T.used++ : SLOTS;
            if (fresh >= SLOTS) {
                // Every counter set has an owner: hand the least recently used one whose stream is idle to the newcomer.  A set is back at
                // zero whenever no launch of its stream is in flight (the last pull of a launch resets it), so an idle stream's set can change
                // hands; a host that recycles streams (a 65th distinct stream used to fall back to the one-tile kernel for the life of the
                // process) now keeps the persistent schedule.  A retired stream handle answers hipStreamQuery with an error: idle as well.
                fresh = -1;
                unsigned long long best = ~0ull;
                for (int x = 0; x < SLOTS; ++x) {
                    if (T.last_use[x] >= best || !T.owner[x]) continue;
                    // (a stream the host destroyed WITHOUT wj_gemm_release_stream: ROCm's hipStreamQuery checks the handle against its list
                    // of live streams and answers with an error, which counts as idle here; a host that wants no query of a dead handle at
                    // all releases its streams)
                    const hipError_t q = hipStreamQuery(T.owner[x]);
                    (void)hipGetLastError();
                    if (q == hipErrorNotReady) continue;           // work in flight on that stream: its counters may be live
                    best = T.last_use[x];
                    fresh = x;
                }
                if (fresh < 0) return nullptr;                      // 64 streams with persistent GEMMs in flight at once
                T.slots.erase(T.owner[fresh]);
            }
            T.owner[fresh] = s;
            it = T.slots.emplace(s, fresh).first;
        }
        slot = it->second;
        T.last_use[slot] = ++T.seq;
    }
    if (dev_out) *dev_out = dev;
    return T.base[dev] + (size_t)slot * 8 * CTR_STRIDE;
}

int wj_gemm_persist_launch(const wj_gemm_args* a, hipStream_t s) {
    if (!wj_gemm_persist_eligible(a)) return WJ_ERR_UNSUPPORTED;
    int dev = 0;
    unsigned* ctr = wj_gemm_persist_counters(s, &dev);
    if (!ctr) return WJ_ERR_UNSUPPORTED;
    switch (a->epilogue) {
        case WJ_EPI_BF16: return launch_persist<WJ_EPI_BF16>(a, s, ctr, dev);
        case WJ_EPI_BIAS_GELU2: return launch_persist<WJ_EPI_BIAS_GELU2>(a, s, ctr, dev);
        case WJ_EPI_BIAS_GELU: return launch_persist<WJ_EPI_BIAS_GELU>(a, s, ctr, dev);
        case WJ_EPI_CONV_GELU: return launch_persist<WJ_EPI_CONV_GELU>(a, s, ctr, dev);
        case WJ_EPI_MUL_GELU_GRAD: return launch_persist<WJ_EPI_MUL_GELU_GRAD>(a, s, ctr, dev);
        default: return WJ_ERR_UNSUPPORTED;
    }
}
