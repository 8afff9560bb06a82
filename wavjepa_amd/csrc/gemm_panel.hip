// Row-panel form of the bf16 GEMM for THIN outputs (variant 5 of wj_gemm_bf16): C[M, 384] = A[M, K] . B[384, K]^T (+ bias), row-form
// operands, K % 128 == 0 -- the predictor's d = 384 shapes (out_proj, linear2 and the row-form dgrads into d = 384: K = 384 / 1152 /
// 1536; autograd of reference jepa.py:129-131,422-440).
//
// Why another schedule.  On the persistent 256 x 256 kernel (csrc/gemm_persist.hip) N = 384 is a full item plus a HALF-width item per
// 256-row panel; the half item keeps every barrier, wait and LDS-DMA of a full one and costs 0.86 of it for half of the arithmetic, the
// launches are 2.67 rounds of items, and the A panels -- the only operand that comes from HBM: W is 0.3-1.2 MB and lives in the L2 --
// arrive with two K tiles in flight (all the LDS a 256 x 256 x 64 ring leaves): 551-908 TFLOP/s where the teacher's shapes reach 1000+.
//
// Here a work item is a FULL ROW of the output: 128 rows x all 384 columns (no half items), and the K loop is cut into STEPS of one
// 128-column third of W per 64-deep K tile:
//   * LDS = two independent rings: A, four slots of one K tile (128 rows x 128 B = 16 KB), filled four K TILES ahead (~2.5 us: an HBM
//     miss is covered); W, four slots of one step (128 rows of W x 128 B = 16 KB), filled four STEPS ahead (L2 latency).  48 KB of A
//     in flight per workgroup in 144 KB of LDS;
//   * 8 waves as 2 x 4: a wave owns 64 rows x 32 columns of every third = 96 accumulator registers; its A fragments (64 x 64) are read
//     once per K tile and serve the three steps, its W fragments once per step.  Both fragment sets are DOUBLE-BUFFERED in registers:
//     step j multiplies what step j - 1 read, and reads what step j + 1 multiplies, so the LDS latency sits under the MFMAs and ONE
//     barrier per step suffices (the eight-phase loop needs eight per K tile);
//   * the two operands are staged by DIFFERENT waves: waves 0-3 issue every A piece (4 LDS-DMA instructions each per K tile), waves 4-7 every W
//     piece (4 each per step).  A wave's vmcnt retires in issue order, so in a wave that issued both, a W piece (L2, needed three steps
//     later) could not be counted done before the A piece issued just ahead of it (HBM, needed four K tiles later): the deep A ring would
//     buy nothing -- exactly what bounds the N = 384 shapes on the eight-phase loop, where every wave stages both.  Split by role, an A
//     wave waits once per K tile with three younger tiles (12 instructions) in flight, a W wave once per step with two younger steps (8);
//   * every step issues the same LDS-DMA instructions whatever the position in the item, also across item boundaries and past the last
//     item (there the A source is clamped to the last panel: valid memory, never multiplied) -- so ONE counted vmcnt value per role is
//     right for every step (+ the 12 stores of an epilogue while they are younger than the awaited piece: three steps / three K tiles);
//   * the items of a workgroup are static (XCD label x resident index, strided): the rows of one A panel are read by one workgroup only,
//     so there is nothing to share and no queue to pull from;
//   * the last panel is SHIFTED inwards (M - 128) like the persistent kernel's edge tiles: same operands, same k order, same bits.
// The accumulation order per output element is that of variants 0-3 (k ascending, 32 per MFMA, bias added last in fp32): bit-identical
// to them; variant 4 starts from the bias and differs in the last place on <= 0.05 % of the elements.
#include <atomic>
#include "common.h"
#include "../../include/wavjepa_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int NT = 512;
constexpr int PN = 384;                           // output width this build serves
constexpr int THIRDS = PN / 128;
constexpr unsigned SLOT = 16384u;                 // one K tile of A (128 rows) or one step of W (128 rows), 128 B per row
constexpr unsigned A_RING = 0u, B_RING = 4u * SLOT;
constexpr unsigned BIAS_OFF = 8u * SLOT;          // PN floats
constexpr unsigned STRIP_OFF = BIAS_OFF + 2048u;  // [8 waves][16 rows x STRIP_ROW]: the epilogue's transpose, per wave, no barriers
constexpr unsigned STRIP_ROW = 144u;
constexpr int LDS_TOTAL = (int)(STRIP_OFF + 8u * 16u * STRIP_ROW);
constexpr int EPI_STORES = 4 * THIRDS;            // global stores one wave's epilogue issues per item

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct QArgs {
    const char* A;
    const char* B;
    char* C;
    const float* bias;
    long ldc_b;
    unsigned lda_b, ldb_b;
    int M, K, nitems, wpx;
    int diag;                 // lab build (WJ_PANEL_DIAG): 1 = no W pieces, 2 = no A pieces, 4 = no MFMAs, 8 = no fragment reads, 16 = no stores (timing only: wrong results); 32 = K rotation per workgroup (correct results, other summation order)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA instruction: 64 lanes x 16 B from sbase + voff (per lane) to LDS bytes [lds_dst + 16 lane).  M0 is written here only.
__device__ __forceinline__ void dma(unsigned voff, const char* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 p;
    p[0] = f2bf(a);
    p[1] = f2bf(b);
    return __builtin_bit_cast(unsigned, p);
}

// The running state of the two prefetch streams (wave-uniform).
struct Cursor {
    // W: (K tile, third) four steps ahead of the multiply; periodic in the item, so it never runs out
    int b_kt, b_sub;
    // A: (item index of this workgroup, K tile) four K tiles ahead; clamped to the last item when the workgroup's items are exhausted
    int a_it, a_kt;
};

template <int SUB, int PB, int PA>
__device__ __forceinline__ void step(f32x4 (&acc)[4][2 * THIRDS], bf16x8 (&af)[2][4][2], bf16x8 (&bf)[2][2][2], char* smem, const QArgs& a,
                                     Cursor& cur, unsigned& slot_b, unsigned& slot_a, int nk, int n_my, int q0, int cstart,
                                     const unsigned (&voff)[4], const unsigned (&dpiece)[4], unsigned a_rd, unsigned b_rd, bool a_wave,
                                     int& lag_w, int& lag_a, int rot) {
    // ---- wait for the pieces the NEXT step multiplies.  W waves: W step j + 1 (issued three steps ago; two younger steps = 8 instructions
    // may stay in flight).  A waves, in the last step of a K tile only: the A tile of the next K tile (issued four K tiles ago; three
    // younger tiles = 12).  Behind an epilogue its 12 stores are younger than the awaited piece for three steps / three K tiles.
    // lgkmcnt(0): the fragment reads of the previous step have returned, so the slot they came from may be refilled behind the barrier.
    __builtin_amdgcn_sched_barrier(0);
    if (!a_wave) {
        if (lag_w > 0) { wait_vmcnt<8 + EPI_STORES>(); --lag_w; }
        else wait_vmcnt<8>();
    } else if constexpr (SUB == THIRDS - 1) {
        if (lag_a > 0) { wait_vmcnt<12 + EPI_STORES>(); --lag_a; }
        else wait_vmcnt<12>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage W step j + 4 into the slot step j occupied (its fragments were read during step j - 1), and in the first step of a K
    // tile the A tile four K tiles ahead into the slot whose fragments were read at the end of the previous K tile
    {
        const int bk = cur.b_kt + rot < nk ? cur.b_kt + rot : cur.b_kt + rot - nk;   // this workgroup's K order starts at K tile `rot`
        const char* sb = a.B + (long)(cur.b_sub * 128) * a.ldb_b + (long)bk * 128;
        const unsigned dst = B_RING + slot_b * SLOT;
        if (!a_wave) {
            // (lab build, timing diagnostics: the pieces are re-read from ONE line so that the vmcnt arithmetic stays what it is)
            const char* src = (WJ_LAB_BUILD && (a.diag & 1)) ? a.B : sb;
            const unsigned mask = (WJ_LAB_BUILD && (a.diag & 1)) ? 0u : ~0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) dma(voff[u] & mask, src, dpiece[u] + dst);
        }
        slot_b = (slot_b + 1) & 3;
        if (++cur.b_sub == THIRDS) { cur.b_sub = 0; if (++cur.b_kt == nk) cur.b_kt = 0; }
    }
    if constexpr (SUB == 0) {
        const int it = cur.a_it < n_my ? cur.a_it : n_my - 1;                // past the last item: the last panel again (never multiplied)
        const int m0 = min((cstart + q0 + it * a.wpx) * 128, a.M - 128);
        const int ak = cur.a_kt + rot < nk ? cur.a_kt + rot : cur.a_kt + rot - nk;
        const char* sa = a.A + (long)m0 * a.lda_b + (long)ak * 128;
        const unsigned dst = A_RING + slot_a * SLOT;
        if (a_wave) {
            const char* src = (WJ_LAB_BUILD && (a.diag & 2)) ? a.A : sa;
            const unsigned mask = (WJ_LAB_BUILD && (a.diag & 2)) ? 0u : ~0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) dma(voff[u] & mask, src, dpiece[u] + dst);
        }
        slot_a = (slot_a + 1) & 3;
        if (++cur.a_kt == nk) { cur.a_kt = 0; ++cur.a_it; }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- fragments of step j + 1 into the other register set: W (slot_b now names step j + 1's slot) ...
    if (!(WJ_LAB_BUILD && (a.diag & 8))) {
        const char* bs = smem + B_RING + slot_b * SLOT;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            bf[PB ^ 1][ni][0] = *reinterpret_cast<const bf16x8*>(bs + b_rd + ni * 2048);
            bf[PB ^ 1][ni][1] = *reinterpret_cast<const bf16x8*>(bs + (b_rd ^ 64u) + ni * 2048);
        }
    }
    // ... and, in the last step of a K tile, the A fragments of the next K tile (slot_a, advanced in the K tile's first step, names the NEXT K tile's slot)
    if (SUB == THIRDS - 1 && !(WJ_LAB_BUILD && (a.diag & 8))) {
        const char* as = smem + A_RING + slot_a * SLOT;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[PA ^ 1][mi][0] = *reinterpret_cast<const bf16x8*>(as + a_rd + mi * 2048);
            af[PA ^ 1][mi][1] = *reinterpret_cast<const bf16x8*>(as + (a_rd ^ 64u) + mi * 2048);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply step j from the registers the previous step filled
    if (WJ_LAB_BUILD && (a.diag & 4)) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][SUB * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[PB][ni][0], af[PA][mi][0], acc[mi][SUB * 2 + ni], 0, 0, 0);
            acc[mi][SUB * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[PB][ni][1], af[PA][mi][1], acc[mi][SUB * 2 + ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
}

// accumulators + bias -> bf16 -> transposed through the wave's LDS strip -> 16-byte stores of 16 rows x 64 B
// acc[mi][t * 2 + ni][r] = C[m0 + wm * 64 + mi * 16 + i][t * 128 + wn * 32 + ni * 16 + 4 g + r]   (i = lane & 15, g = lane >> 4)
__device__ __forceinline__ void epilogue(f32x4 (&acc)[4][2 * THIRDS], char* smem, const QArgs& a, int m0, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int i = lane & 15, g = lane >> 4;
    char* strip = smem + STRIP_OFF + wave * (16 * STRIP_ROW);
    char* wr = strip + i * STRIP_ROW + g * 8;                                   // + ni * 32
    const int srow = lane >> 2, schunk = lane & 3;
    const char* rd = strip + srow * STRIP_ROW + schunk * 16;
    char* c0 = a.C + (long)(m0 + wm * 64 + srow) * a.ldc_b + (long)(wn * 32) * 2 + schunk * 16;
    const float* bias = reinterpret_cast<const float*>(smem + BIAS_OFF) + wn * 32 + 4 * g;
#pragma unroll
    for (int t = 0; t < THIRDS; ++t) {
        f32x4 bv[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(bias + t * 128 + ni * 16);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const f32x4 v = acc[mi][t * 2 + ni] + bv[ni];
                *reinterpret_cast<u32x2*>(wr + ni * 32) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
            }
            __builtin_amdgcn_wave_barrier();
            const u32x4 o = *reinterpret_cast<const u32x4*>(rd);
            __builtin_amdgcn_wave_barrier();
            // (lab build, diag 16: the stores go to ONE line per lane group -- same count, no write traffic)
            char* dst = (WJ_LAB_BUILD && (a.diag & 16)) ? a.C + (lane & 7) * 16 : c0 + (long)(mi * 16) * a.ldc_b + (long)(t * 128) * 2;
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
        }
    }
}

__global__ __launch_bounds__(NT, 1) void gemm_panel_kernel(QArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool a_wave = wave < 4;                     // waves 0-3 stage A, waves 4-7 stage W (see the header)
    const int nk = a.K >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- this workgroup's items: the panels [cstart, cstart + clen) of its XCD label, every wpx-th from q0
    const int xl = blockIdx.x & 7, q0 = blockIdx.x >> 3;
    const int qn = a.nitems >> 3, qr = a.nitems & 7;
    const int clen = qn + (xl < qr ? 1 : 0);
    const int cstart = xl < qr ? xl * (qn + 1) : qr * (qn + 1) + (xl - qr) * qn;
    if (q0 >= clen) return;
    const int n_my = (clen - q0 + a.wpx - 1) / a.wpx;

    // ---- bias -> LDS (zeros when there is none); visible behind the prologue's barrier
    if (t < PN) reinterpret_cast<float*>(smem + BIAS_OFF)[t] = a.bias ? a.bias[t] : 0.f;

    // ---- piece geometry: a slot is 16 instructions of 8 rows x 128 B; staging wave w (= wave & 3) issues rows 32 w + 8 u + lane / 8
    // (u = 0..3); lane % 8 is the LDS chunk position, holding source chunk (lane % 8) ^ ((row >> 1) & 7) (the swizzle the fragment reads undo)
    unsigned dpiece[4], voff[4];
    const unsigned ld_b = a_wave ? a.lda_b : a.ldb_b;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r0 = 32 * (wave & 3) + 8 * u;
        dpiece[u] = lds0 + (unsigned)r0 * 128u;
        const int row = r0 + (lane >> 3);
        voff[u] = (unsigned)row * ld_b + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    // fragment read offsets inside a slot: lane (i, g) reads row (block + i), chunk g (k 0-31) and g ^ 4 (k 32-63)
    const int i = lane & 15, g = lane >> 4;
    const unsigned sw = (unsigned)((g ^ ((i >> 1) & 7)) << 4);
    const unsigned a_rd = (unsigned)((wm * 64 + i) * 128) + sw;
    const unsigned b_rd = (unsigned)((wn * 32 + i) * 128) + sw;

    // K rotation (lab build, WJ_PANEL_DIAG bit 32): workgroup w walks the K tiles of every item from tile rot(w) round to rot(w) - 1, so
    // that at any instant the chip's workgroups read DIFFERENT 128-byte columns of their A rows (all of them start together and move at
    // the same pace: unrotated, every row piece in flight has the same address bits 7-10 -- the same few memory channels)
    const int rot = (WJ_LAB_BUILD && (a.diag & 32)) ? (q0 + 3 * xl) % nk : 0;
    auto phys = [&](int kt) { return kt + rot < nk ? kt + rot : kt + rot - nk; };
    // ---- prologue: A tiles 0-3 of the first item (nk >= 4) by the A waves, W steps 0-3 by the W waves
    Cursor cur;
    cur.a_it = 0; cur.a_kt = 4; cur.b_kt = 1; cur.b_sub = 1;          // the cursors behind the prologue: A tile 4, W step 4 = (K tile 1, third 1)
    if (cur.a_kt == nk) { cur.a_kt = 0; cur.a_it = 1; }
    {
        const int m0 = min((cstart + q0) * 128, a.M - 128);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (a_wave) {
                const char* sa = a.A + (long)m0 * a.lda_b + (long)phys(k) * 128;
#pragma unroll
                for (int u = 0; u < 4; ++u) dma(voff[u], sa, dpiece[u] + A_RING + k * SLOT);
            } else {
                const int kt = k / THIRDS, sub = k - kt * THIRDS;       // W step k
                const char* sb = a.B + (long)(sub * 128) * a.ldb_b + (long)phys(kt) * 128;
#pragma unroll
                for (int u = 0; u < 4; ++u) dma(voff[u], sb, dpiece[u] + B_RING + k * SLOT);
            }
        }
    }
    f32x4 acc[4][2 * THIRDS];
    bf16x8 af[2][4][2], bf[2][2][2];
    // A waves: tile 0 has landed with tiles 1-3 (12 instructions) in flight; W waves: step 0 with steps 1-3 in flight
    wait_vmcnt<12>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the bias words
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        bf[0][ni][0] = *reinterpret_cast<const bf16x8*>(smem + B_RING + b_rd + ni * 2048);
        bf[0][ni][1] = *reinterpret_cast<const bf16x8*>(smem + B_RING + (b_rd ^ 64u) + ni * 2048);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        af[0][mi][0] = *reinterpret_cast<const bf16x8*>(smem + A_RING + a_rd + mi * 2048);
        af[0][mi][1] = *reinterpret_cast<const bf16x8*>(smem + A_RING + (a_rd ^ 64u) + mi * 2048);
    }
    // slot_b: the slot the next W refill goes to = the slot of the step being multiplied; slot_a likewise for K tiles
    unsigned slot_b = 0, slot_a = 0;
    int lag_w = 0, lag_a = 0;

    for (int it = 0; it < n_my; ++it) {
        const int m0 = min((cstart + q0 + it * a.wpx) * 128, a.M - 128);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int c = 0; c < 2 * THIRDS; ++c) acc[mi][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; kt += 2) {
            step<0, 0, 0>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
            step<1, 1, 0>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
            step<2, 0, 0>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
            step<0, 1, 1>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
            step<1, 0, 1>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
            step<2, 1, 1>(acc, af, bf, smem, a, cur, slot_b, slot_a, nk, n_my, q0, cstart, voff, dpiece, a_rd, b_rd, a_wave, lag_w, lag_a, rot);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // MFMA results of the last step -> VALU readers behind the loop branch
        epilogue(acc, smem, a, m0, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        lag_w = 3;                                    // steps / K tiles whose awaited piece is older than this epilogue's stores
        lag_a = 3;
    }
    wait_vmcnt<0>();                                  // the clamped prefetches and the last stores
    __builtin_amdgcn_s_barrier();
}


}  // namespace

bool wj_gemm_panel_eligible(const wj_gemm_args* a) {
    if (a->a_trans || a->b_trans || a->rowmap || a->split_k > 1 || a->colsum || a->aux) return false;
    if (a->epilogue != WJ_EPI_BF16) return false;
    if (a->N != PN || a->M < 128) return false;
    if (a->K < 256 || (a->K % 128)) return false;                   // K tiles come in pairs (register double-buffering), at least four
    if ((a->lda & 7) || (a->ldb & 7) || (a->ldc & 7)) return false;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->bias) & 15) return false;
    if (a->lda * 2 * 128 >= (1l << 31) || a->ldb * 2 * 128 >= (1l << 31)) return false;   // 32-bit per-lane offsets inside a slot
    return true;
}

int wj_gemm_panel_launch(const wj_gemm_args* a, hipStream_t s) {
    if (!wj_gemm_panel_eligible(a)) return WJ_ERR_UNSUPPORTED;
    QArgs p;
    p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C; p.bias = (const float*)a->bias;
    p.ldc_b = a->ldc * 2; p.lda_b = (unsigned)(a->lda * 2); p.ldb_b = (unsigned)(a->ldb * 2);
    p.M = a->M; p.K = a->K;
    p.nitems = (a->M + 127) / 128;
    p.wpx = a->persist_cus > 0 ? (a->persist_cus > 32 ? 32 : a->persist_cus) : 32;
    p.diag = wj_lab_env_int("WJ_PANEL_DIAG", 0);              // (read per launch in the lab build: a timing tool sweeps it)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return WJ_ERR_UNSUPPORTED;
    static std::atomic<bool> lds_ok[32];
    if (!lds_ok[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)gemm_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return WJ_ERR_LAUNCH;
        lds_ok[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(gemm_panel_kernel, dim3(8 * p.wpx), dim3(NT), LDS_TOTAL, s, p);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}
