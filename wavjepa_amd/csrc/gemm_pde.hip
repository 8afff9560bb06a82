// Persistent bf16 GEMM with a DEFERRED epilogue (variant 6 of wj_gemm_bf16): C[M,N] = gelu(A[M,K] . B[N,K]^T + bias), row-form operands,
// K % 128 == 0, K >= 256, N % 256 == 0, M >= 128; epilogues BIAS_GELU (teacher linear1), BIAS_GELU2 (student / predictor linear1: gelu and
// gelu'), CONV_GELU (sparse conv layers: pre-activation and GELU, rows outside the active segments zeroed).
//
// Why.  In the persistent 256 x 256 kernel (csrc/gemm_persist.hip) an erf-GELU epilogue is pure vector arithmetic -- 14 issue slots per
// output, 18-21 with gelu' -- during which the matrix pipe idles, and the K loop is pure matrix work during which the vector pipe idles:
// 7 us of a 29.5-us item on the teacher's linear1 (K = 768), 13 of 22.4 on the predictor's (K = 384).  All eight waves of a workgroup
// walk in lockstep (they share the staged operand tiles), so the two never overlap, and the 128 accumulators + 64 fragment registers of
// a 128 x 64 wave tile leave no room to keep a finished tile around.  Here the work item is 128 x 256 (wave tile 64 x 64: 64 accumulators):
//   * at the end of an item the accumulators (bias included: it was the C operand of their first MFMA) are rounded to bf16 -- the value
//     GELU is taken of, as nn.GELU sees a bf16 linear's output -- and PARKED, packed, in 32 registers; the K loop of the next item starts
//     at once;
//   * the parked tile is finished in four blocks of 16 rows, one per K tile of the next item's first four K tiles, each inside the load
//     segment of that K tile's second phase: GELU, one trip through the wave's LDS strip (MFMA layout -> whole 128-B lines, as in
//     gemm_persist.hip), 16-byte non-temporal stores.  While one wave group does that, the other runs its MFMA cluster, and vice versa;
//   * the stores sit INSIDE the K-tile stream, between LDS-DMA instructions that share their in-order vmcnt counter: every wait counts
//     them exactly (stores of this K tile and of the previous one are younger than the pieces the wait retires; a store acknowledgement
//     takes ~4 us under load, waiting for one by accident costs more than the block);
//   * K tile = two phases (X x B0, X x B1), three pieces (X: 128 A rows; B0 / B1: the two 32-row halves of every wave column), two LDS
//     parities of 48 KB; piece schedule: B1(t+1) in phase 0 of K tile t, X(t+2) and B0(t+2) in phase 1 (into the parity whose X / B0
//     were read one barrier pair earlier);
//   * tiles are pulled from the per-XCD counters of gemm_persist.hip (same counter sets, same mailbox protocol); edge panels are shifted
//     inwards (M - 128), never clipped.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include "../../include/wavjepa_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int NT = 512;
constexpr unsigned BUF = 49152u, BOFF = 16384u;   // LDS: two parities of [A 128 rows | B 256 rows] x 128 B
constexpr unsigned AUX = 2u * BUF;                // [2 items][8 waves][64 floats] bias of the wave's 64 columns
constexpr unsigned MAILBOX = AUX + 4096u;         // next-next item index, written by wave 0
constexpr unsigned STAGE = MAILBOX + 256u;        // [8 waves][16 rows x STAGE_ROW B]: the blocks' transpose (per wave, no barriers)
constexpr unsigned STAGE_ROW = 144u;
constexpr int LDS_TOTAL = (int)(STAGE + 8u * 16u * STAGE_ROW);
constexpr int CTR_STRIDE = 32;

__device__ __attribute__((aligned(256))) unsigned char g_zero_bias_pde[256];

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct DArgs {
    const char* A;
    const char* B;
    char* C;
    char* C2;
    const float* bias;
    unsigned* ctr;
    long ldc_b;
    unsigned lda_b, ldb_b;
    int M, N, K, ntiles, items_n, wpx;
    int seg_rows, seg_valid;
    int diag;                 // lab build (WJ_PDE_DIAG): bit 0 = never park a tile (the raw K loop: no GELU, no stores, wrong results)
};

__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// s_waitcnt vmcnt(n), n a run-time value in 0..16 (the thresholds depend on which K tiles carry a block of the parked tile)
__device__ __forceinline__ void wait_cnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
}

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

struct Bases {
    const char* x;
    const char* b0;
    const char* b1;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 p;
    p[0] = f2bf(a);
    p[1] = f2bf(b);
    return __builtin_bit_cast(unsigned, p);
}

__device__ __forceinline__ void store16(char* p, const u32x4& v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}

template <int EPI> struct Outs {
    static constexpr int N = (EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_CONV_GELU) ? 2 : 1;
    static constexpr int S = 2 * N;                 // global stores one wave issues per block of the parked tile
};

// the parked tile of one wave: bf16(acc + bias) of its 64 x 64 outputs in the MFMA layout, and where it goes
struct Parked {
    u32x2 h[4][4];      // [mi][ni]: columns 4 g .. 4 g + 3 of rows mi * 16 + i  (i = lane & 15, g = lane >> 4)
    char* c1;           // C  + this lane's first store address (row srow of block 0, 16-B chunk schunk of the wave's 128-B line)
    char* c2;           // C2 likewise (two-output epilogues)
    int rem;            // CONV_GELU: (row of this lane in block 0) % seg_rows
};

// One 16-row block of the parked tile: GELU, through the strip, stores.  Issues exactly Outs<EPI>::S global stores.
template <int EPI, int MI>
__device__ __forceinline__ void parked_block(const Parked& pk, char* wr, const char* rd, long ldc_b, int seg_rows, int seg_valid) {
    bool valid = true;
    if constexpr (EPI == WJ_EPI_CONV_GELU) {
        int r = pk.rem + 16 * MI;                    // seg_rows > 16: at most MI wraps
#pragma unroll
        for (int x = 0; x < MI; ++x) r = r >= seg_rows ? r - seg_rows : r;
        valid = r < seg_valid;
    }
    u32x2 o1[4], o2[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const unsigned w0 = pk.h[MI][ni][0], w1 = pk.h[MI][ni][1];
        f32x2 h[2], gl[2], gp[2];
        h[0] = f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
        h[1] = f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (EPI == WJ_EPI_BIAS_GELU2) gelu_pk<true>(h[q], gl[q], gp[q]);
            else gelu_pk<false>(h[q], gl[q], gp[q]);
        }
        u32x2 og = u32x2{pack_bf16(gl[0].x, gl[0].y), pack_bf16(gl[1].x, gl[1].y)};
        if constexpr (EPI == WJ_EPI_BIAS_GELU) {
            o1[ni] = og;
        } else if constexpr (EPI == WJ_EPI_BIAS_GELU2) {
            o1[ni] = u32x2{pack_bf16(gp[0].x, gp[0].y), pack_bf16(gp[1].x, gp[1].y)};   // C  = gelu'(h)
            o2[ni] = og;                                                                   // C2 = gelu(h)
        } else {                                                                           // CONV_GELU: C = pre, C2 = post
            o1[ni] = u32x2{valid ? w0 : 0u, valid ? w1 : 0u};
            o2[ni] = u32x2{valid ? og[0] : 0u, valid ? og[1] : 0u};
        }
    }
    const long off = (long)(MI * 16) * ldc_b;
    const long row8 = 8 * ldc_b;
    auto through_strip = [&](const u32x2 (&o)[4], char* dst) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<u32x2*>(wr + ni * 32) = o[ni];
        __builtin_amdgcn_wave_barrier();             // the lanes exchange data through the strip without a barrier hipcc knows of
        const u32x4 lo = *reinterpret_cast<const u32x4*>(rd);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(rd + 8 * STAGE_ROW);
        __builtin_amdgcn_wave_barrier();
        store16(dst, lo);
        store16(dst + row8, hi);
    };
    through_strip(o1, pk.c1 + off);
    if constexpr (Outs<EPI>::N == 2) through_strip(o2, pk.c2 + off);
}

// One K tile (64 deep) of the stream = two phases.
//   TOP:      first K tile of an item: every accumulator's first MFMA takes the bias as C; top_n VMEM operations were issued in front of it
//             (the NEXT item's bias DMA; in wave 0 also the pull)
//   BLK:      0-3: this K tile carries block BLK of the parked tile (in phase 1's load segment); -1: none
//   prev_s:   global stores the PREVIOUS K tile issued (its block, behind that K tile's LDS-DMAs): younger than the pieces both phases wait for
//   stage_b1: B1(t+1) exists (false only in the last K tile of a workgroup's last item); sw_b1: it is the next item's first
//   stage_x:  X(t+2), B0(t+2) exist (false only in the last two K tiles of the last item); sw_x: they are the next item's first
template <int PAR, bool TOP, int BLK, int EPI>
__device__ __forceinline__ void kstep(f32x4 (&acc)[4][4], char* smem, Bases& s, const char* nA, const char* nB, unsigned b1_full,
                                      const unsigned (&vx)[2], const unsigned (&vb)[2], const unsigned (&dx)[2], const unsigned (&db)[2],
                                      unsigned a_lo, unsigned b_lo, bool sw_b1, bool stage_b1, bool sw_x, bool stage_x, int prev_s,
                                      int top_n, const f32x4 (&bv)[4], const Parked& pk, char* wr, const char* rd, const DArgs& a, bool mail,
                                      const unsigned& pv) {
    constexpr unsigned CUR = PAR * BUF, OTH = (PAR ^ 1) * BUF;
    constexpr int S = BLK >= 0 ? Outs<EPI>::S : 0;
    char* cur = smem + CUR;
    const unsigned a_hi = a_lo ^ 64u, b_hi = b_lo ^ 64u;
    bf16x8 af[8], b0f[4], b1f[4];
    auto lds = [&](unsigned off) { return *reinterpret_cast<const bf16x8*>(cur + off); };
    // ---- phase 0: X, B0 of this K tile; stage B1(t+1) into the other parity; wait for B1(t)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b0f[2 * x] = lds(b_lo + x * 2048); b0f[2 * x + 1] = lds(b_hi + x * 2048); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[2 * x] = lds(a_lo + x * 2048); af[2 * x + 1] = lds(a_hi + x * 2048); }
    if (stage_b1) {
        if (sw_b1) s.b1 = nB + b1_full;
        dma<OTH + BOFF + 4096>(vb[0], s.b1, db[0]); dma<OTH + BOFF + 4096>(vb[1], s.b1, db[1]);
        s.b1 += 128;
        // younger than B1(t): X(t+1), B0(t+1) (4), the previous K tile's block stores, what went out in front of a first K tile, B1(t+1) (2)
        wait_cnt(6 + prev_s + (TOP ? top_n : 0));
    } else {
        wait_cnt(0);
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni], af[2 * mi], TOP ? bv[ni] : acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni + 1], af[2 * mi + 1], acc[mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: B1 of this K tile; stage X(t+2), B0(t+2) into THIS parity; the parked tile's block; wait for X(t+1), B0(t+1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b1f[2 * x] = lds(b_lo + 4096 + x * 2048); b1f[2 * x + 1] = lds(b_hi + 4096 + x * 2048); }
    if (stage_x) {
        if (sw_x) { s.x = nA; s.b0 = nB; }
        dma<CUR>(vx[0], s.x, dx[0]); dma<CUR>(vx[1], s.x, dx[1]);
        dma<CUR + BOFF>(vb[0], s.b0, db[0]); dma<CUR + BOFF>(vb[1], s.b0, db[1]);
        s.x += 128; s.b0 += 128;
    }
    if constexpr (BLK >= 0) parked_block<EPI, BLK>(pk, wr, rd, a.ldc_b, a.seg_rows, a.seg_valid);
    __builtin_amdgcn_sched_barrier(0);
    if (stage_x) {
        // younger than X(t+1), B0(t+1): the previous K tile's block stores, what went out in front of a first K tile, B1(t+1) (2), X(t+2),
        // B0(t+2) (4), this block's stores
        wait_cnt(6 + S + prev_s + (TOP ? top_n : 0));
    } else {
        wait_cnt(0);
    }
    if (mail) {
        // the pull went out in front of this item's first K tile: 13 younger operations + two blocks' stores by now, 6 + two blocks' stores in flight
        asm volatile("s_mov_b64 exec, 1\n\ts_nop 0\n\tds_write_b32 %1, %0\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
                     ::"v"(pv), "v"(MAILBOX) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni], af[2 * mi], TOP ? bv[2 + ni] : acc[mi][2 + ni], 0, 0, 0);
            acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni + 1], af[2 * mi + 1], acc[mi][2 + ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

template <int EPI>
__global__ __launch_bounds__(NT, 1) void gemm_pde_kernel(DArgs a) {
    constexpr int S = Outs<EPI>::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n = a.K >> 6;                           // K tiles per item (even, >= 4)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- this workgroup's queue: the logical item ids [cstart, cstart + clen) of its XCD label
    const int xl = blockIdx.x & 7;
    const int qn = a.ntiles >> 3, qr = a.ntiles & 7;
    const int clen = qn + (xl < qr ? 1 : 0);
    const int cstart = xl < qr ? xl * (qn + 1) : qr * (qn + 1) + (xl - qr) * qn;
    if ((int)(blockIdx.x >> 3) >= clen) return;
    unsigned* ctr = a.ctr + xl * CTR_STRIDE;
    unsigned pv = 0;
    auto pull = [&]() {
        if (wave == 0)
            asm volatile("s_mov_b64 exec, 1\n\ts_nop 0\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_mov_b64 exec, -1"
                         : "+v"(pv) : "v"(0u), "v"(1u), "s"(ctr) : "memory");
    };

    // ---- piece geometry of this wave: two 1-KiB instructions (8 rows x 128 B) per piece
    unsigned dx[2], db[2], vx[2], vb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rx = 16 * wave + 8 * u;
        const int ib = 16 * wave + 8 * u, rb = (ib >> 5) * 64 + (ib & 31);
        dx[u] = lds0 + (unsigned)rx * 128u;
        db[u] = lds0 + (unsigned)rb * 128u;
        const int rowx = rx + (lane >> 3), rowb = rb + (lane >> 3);
        vx[u] = (unsigned)rowx * a.lda_b + (unsigned)(((lane & 7) ^ ((rowx >> 1) & 7)) * 16);
        vb[u] = (unsigned)rowb * a.ldb_b + (unsigned)(((lane & 7) ^ ((rowb >> 1) & 7)) * 16);
    }
    const unsigned b1_full = 32u * a.ldb_b;
    auto item_coords = [&](int q, int& m0, int& n0) {
        const int L = cstart + q;
        const int tm = L / a.items_n;
        m0 = min(tm * 128, a.M - 128);                 // the last panel is shifted inwards
        n0 = (L - tm * a.items_n) * 256;
    };
    const char* bias_src = a.bias ? reinterpret_cast<const char*>(a.bias) : reinterpret_cast<const char*>(g_zero_bias_pde);
    auto bias_dma = [&](int n0, int slot) {
        const int ln = opaque(lane);
        const unsigned bias_v = a.bias ? (unsigned)((wn * 64 + (ln & 15) * 4) * 4) : (unsigned)((ln & 15) * 16);
        const char* sb = a.bias ? bias_src + (long)n0 * 4 : bias_src;
        const unsigned dst = lds0 + AUX + (unsigned)(slot * 2048 + wave * 256);
        asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xffff\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                     ::"v"(bias_v), "s"(sb), "s"(dst) : "memory", "m0");
    };

    f32x4 acc[4][4];

    // ---- prologue: the first item is static; pull the second
    int m0, n0;
    item_coords(blockIdx.x >> 3, m0, n0);
    Bases s;
    s.x = a.A + (long)m0 * a.lda_b;
    s.b0 = a.B + (long)n0 * a.ldb_b;
    s.b1 = s.b0 + b1_full;
    bias_dma(n0, 0);
    pull();
    dma<0>(vx[0], s.x, dx[0]); dma<0>(vx[1], s.x, dx[1]);
    dma<BOFF>(vb[0], s.b0, db[0]); dma<BOFF>(vb[1], s.b0, db[1]);
    dma<BOFF + 4096>(vb[0], s.b1, db[0]); dma<BOFF + 4096>(vb[1], s.b1, db[1]);
    s.x += 128; s.b0 += 128; s.b1 += 128;
    dma<BUF>(vx[0], s.x, dx[0]); dma<BUF>(vx[1], s.x, dx[1]);
    dma<BUF + BOFF>(vb[0], s.b0, db[0]); dma<BUF + BOFF>(vb[1], s.b0, db[1]);
    s.x += 128; s.b0 += 128;
    wait_cnt(4);                                       // K tile 0's pieces, the bias and the pull (all older) have landed
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
    const unsigned a_lo = (unsigned)((wm * 64 + i) * 128) + sw;
    const unsigned b_lo = BOFF + (unsigned)((wn * 64 + i) * 128) + sw;
    // the strip of this wave, in the MFMA layout (write side) and in the store layout (read side: row lane >> 3 [+ 8], chunk lane & 7)
    const int ln = opaque(lane);
    char* strip = smem + STAGE + wave * (16 * STAGE_ROW);
    char* wr = strip + (ln & 15) * STAGE_ROW + (ln >> 4) * 8;
    const int srow = ln >> 3, schunk = ln & 7;
    const char* rd = strip + srow * STAGE_ROW + schunk * 16;
    const long lane_off = (long)srow * a.ldc_b + (long)(wn * 128 + schunk * 16);

    if (wm == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind

    Parked pk;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) pk.h[mi][ni] = u32x2{0u, 0u};
    pk.c1 = a.C; pk.c2 = a.C2; pk.rem = 0;
    bool have_parked = false;
    int slot = 0;
    int last_s = 0;                                    // stores issued by the last K tile of the previous item (n == 4 only)
    for (;;) {
        int m1 = m0, n1 = n0;
        const bool pulled = has_next;
        const char* nA = a.A;
        const char* nB = a.B;
        if (has_next) {
            item_coords(q_next, m1, n1);
            nA = a.A + (long)m1 * a.lda_b;
            nB = a.B + (long)n1 * a.ldb_b;
        }
        __builtin_amdgcn_sched_barrier(0);
        bias_dma(n1, slot ^ 1);                        // always exactly one DMA here (the first K tile's waits count it)
        if (pulled) pull();
        __builtin_amdgcn_sched_barrier(0);
        f32x4 bv[4];
        {
            const int g = opaque(lane) >> 4;
            const char* bs = smem + AUX + slot * 2048 + wave * 256;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 64 + g * 16);
        }
        const bool hn = has_next;
        const int top_n = (pulled && wave == 0) ? 2 : 1;
        // K tile t: sw_b1 = (t == n - 1), stage_b1 = (t < n - 1 || hn), sw_x = (t + 2 == n), stage_x = (t + 2 < n || hn)
#define KSTEP(PAR, TOP, BLK, T, PREV_S, MAIL)                                                                                                  \
    kstep<PAR, TOP, BLK, EPI>(acc, smem, s, nA, nB, b1_full, vx, vb, dx, db, a_lo, b_lo, (T) == n - 1, (T) < n - 1 || hn, (T) + 2 == n,     \
                              (T) + 2 < n || hn, PREV_S, top_n, bv, pk, wr, rd, a, MAIL, pv)
        if (have_parked) {
            KSTEP(0, true, 0, 0, last_s, false);
            KSTEP(1, false, 1, 1, S, pulled && wave == 0);
            KSTEP(0, false, 2, 2, S, false);
            KSTEP(1, false, 3, 3, S, false);
        } else {
            KSTEP(0, true, -1, 0, 0, false);
            KSTEP(1, false, -1, 1, 0, pulled && wave == 0);
            KSTEP(0, false, -1, 2, 0, false);
            KSTEP(1, false, -1, 3, 0, false);
        }
        int prev = have_parked ? S : 0;
        for (int kt = 4; kt < n; kt += 2) {
            KSTEP(0, false, -1, kt, prev, false);
            KSTEP(1, false, -1, kt + 1, 0, false);
            prev = 0;
        }
#undef KSTEP
        last_s = (n == 4 && have_parked) ? S : 0;
        // ---- park the item: bf16(acc) in the MFMA layout, and where its blocks go
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA results of the last phase -> VALU readers behind the loop branch
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const f32x4 v = acc[mi][ni];
                pk.h[mi][ni] = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
        {
            const long tile_off = (long)(m0 + wm * 64) * a.ldc_b + (long)n0 * 2;
            pk.c1 = a.C + tile_off + lane_off;
            if constexpr (Outs<EPI>::N == 2) pk.c2 = a.C2 + tile_off + lane_off;
            if constexpr (EPI == WJ_EPI_CONV_GELU) pk.rem = (m0 + wm * 64 + (ln & 15)) % a.seg_rows;
        }
        have_parked = !(WJ_LAB_BUILD && (a.diag & 1));
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        m0 = m1; n0 = n1;
        slot ^= 1;
        if (pulled) {
            unsigned mv = lds_read_u32(MAILBOX);
            mv = __builtin_amdgcn_readfirstlane(mv);
            if (mv == (unsigned)(clen - 1) && t == 0) atomicExch(ctr, 0u);
            q_next = a.wpx + (int)mv;
            has_next = q_next < clen;
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger
    // ---- the last item's parked tile: nothing left to hide it under
    if (WJ_LAB_BUILD && (a.diag & 1)) return;
    parked_block<EPI, 0>(pk, wr, rd, a.ldc_b, a.seg_rows, a.seg_valid);
    parked_block<EPI, 1>(pk, wr, rd, a.ldc_b, a.seg_rows, a.seg_valid);
    parked_block<EPI, 2>(pk, wr, rd, a.ldc_b, a.seg_rows, a.seg_valid);
    parked_block<EPI, 3>(pk, wr, rd, a.ldc_b, a.seg_rows, a.seg_valid);
}

template <int EPI>
int launch_pde(const wj_gemm_args* a, hipStream_t s, unsigned* ctr, int dev) {
    DArgs p;
    p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C; p.C2 = (char*)a->C2; p.bias = (const float*)a->bias;
    p.ctr = ctr;
    p.ldc_b = a->ldc * 2; p.lda_b = (unsigned)(a->lda * 2); p.ldb_b = (unsigned)(a->ldb * 2);
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.items_n = a->N / 256;
    p.ntiles = ((a->M + 127) / 128) * p.items_n;
    p.wpx = wj_gemm_persist_wpx(a);
    p.seg_rows = a->seg_rows > 0 ? a->seg_rows : 1;
    p.seg_valid = a->seg_rows > 0 ? a->seg_valid : 1;
    {
        static const int diag = wj_lab_env_int("WJ_PDE_DIAG", 0);
        p.diag = diag;
    }
    auto kern = gemm_pde_kernel<EPI>;
    static std::atomic<bool> lds_ok[32];            // once per kernel and device
    if (!lds_ok[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return WJ_ERR_LAUNCH;
        lds_ok[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(8 * p.wpx), dim3(NT), LDS_TOTAL, s, p);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

}  // namespace

bool wj_gemm_pde_eligible(const wj_gemm_args* a) {
    if (a->a_trans || a->b_trans || a->rowmap || a->split_k > 1 || a->colsum || a->aux) return false;
    if (a->K < 256 || (a->K % 128) || a->M < 128 || a->N < 256 || (a->N % 256)) return false;
    const int e = a->epilogue;
    if (e != WJ_EPI_BIAS_GELU2 && e != WJ_EPI_BIAS_GELU && e != WJ_EPI_CONV_GELU) return false;
    if ((e == WJ_EPI_BIAS_GELU2 || e == WJ_EPI_CONV_GELU) && !a->C2) return false;
    if (e == WJ_EPI_CONV_GELU && a->seg_rows > 0 && a->seg_rows <= 16) return false;
    if ((a->lda & 7) || (a->ldb & 7) || (a->ldc & 7)) return false;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->C2 | (uintptr_t)a->bias) & 15) return false;
    const long items = (long)((a->M + 127) / 128) * (a->N / 256);
    if (items < 512) return false;                    // two rounds of 256 workgroups at least: a parked tile needs a next item to hide under
    if (a->lda * 2 * 128 >= (1l << 31) || a->ldb * 2 * 256 >= (1l << 31)) return false;   // 32-bit per-lane offsets inside an item
    return true;
}

int wj_gemm_pde_launch(const wj_gemm_args* a, hipStream_t s) {
    if (!wj_gemm_pde_eligible(a)) return WJ_ERR_UNSUPPORTED;
    int dev = 0;
    unsigned* ctr = wj_gemm_persist_counters(s, &dev);
    if (!ctr) return WJ_ERR_UNSUPPORTED;
    switch (a->epilogue) {
        case WJ_EPI_BIAS_GELU2: return launch_pde<WJ_EPI_BIAS_GELU2>(a, s, ctr, dev);
        case WJ_EPI_BIAS_GELU: return launch_pde<WJ_EPI_BIAS_GELU>(a, s, ctr, dev);
        case WJ_EPI_CONV_GELU: return launch_pde<WJ_EPI_CONV_GELU>(a, s, ctr, dev);
        default: return WJ_ERR_UNSUPPORTED;
    }
}
