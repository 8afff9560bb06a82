// bf16 MFMA GEMM for gfx950, generation 3: the dense-contraction engine of the WavJEPA step
// (QKV / out-proj / MLP / mappers / predictor, their dgrad + wgrad, and conv layers 1..5 as implicit GEMM
// over a channels-last activation with overlapping rows: lda = stride*C, K = k*C).
//
//   C[M,N] = opA(A) . opB(B)      fp32 accumulate on v_mfma_f32_16x16x32_bf16
//     a_trans = 0 : A stored [M][K], K contiguous ("row form")      a_trans = 1 : A stored [K][M], M contiguous ("col form")
//     b_trans = 0 : B stored [N][K], K contiguous (nn.Linear)       b_trans = 1 : B stored [K][N], N contiguous
//   forward  y = x W^T : (0,0)      dgrad dx = dy W : (0,1)      wgrad dW = dy^T x : (1,1)
//
// Measured on MI355X (PMC): with a 256x128 tile the loop is bound by operand delivery L2 -> LDS (MFMA pipe 40 %
// busy, LDS 21 %), so the tile is as wide as N allows:
//   * BN = 256: 256 x 256 x 32 tile (128 FLOP per operand byte), 8 waves 2(M) x 4(N), 128x64 per wave (8x4 MFMA
//     tiles, 128 accumulator VGPRs), 4-stage LDS ring (128 KiB), one workgroup per CU.
//   * BN = 128: 256 x 128 x 32 tile, 8 waves 4(M) x 2(N), 64x64 per wave, 3-stage ring (72 KiB), TWO workgroups per CU
//     (one's prologue / VALU-heavy epilogue overlaps the other's MFMA loop) - used when N is not a multiple of 256.
// Common structure:
//   * Operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR staging);
//     tile t+STAGES-1 is issued while tile t is computed; the only VMEM wait in the loop is a COUNTED s_waitcnt
//     vmcnt(n) that leaves the younger tiles in flight, followed by one raw s_barrier per K tile.
//   * The LDS image is lane-linear (what LDS-DMA can write); bank-conflict-free reads come from an XOR swizzle
//     applied to the per-lane SOURCE address and again on the read (row-form 64-B rows: 16-B chunk ^ h[(row>>2)&3];
//     col-form: 32-B column ^ f(k)).  Col-form fragments are read with ds_read_b64_tr_b16 from inline asm (through
//     the intrinsic hipcc inserts vmcnt(0) before every read, draining the ring).
//   * K tails / out-of-range columns read a 256-B zero page instead of being predicated; out-of-range rows are
//     clamped (their outputs are never stored).
//   * MFMA operands are swapped (B fragment first) so a lane owns 4 consecutive output columns; the accumulators
//     are staged through LDS in row chunks and written as whole rows: 16-B coalesced stores, row-contiguous
//     256-B float atomics for the split-K wgrad.
//   * Workgroup ids are remapped so that the N-tiles sharing an A panel run on one XCD (shared L2).
#include <stdlib.h>
#include "common.h"
#include "../../include/wavjepa_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int BM = 256, BK = 32, NT = 512;
constexpr long PAIR_TILE_BYTES = 2L * 4 * 32 * 64 * 16;   // K-split pairs: scratch per output tile (two roles x four waves' accumulators)

// BMT: tile rows.  256 everywhere except the 384 x 128 tile of the predictor's grouped weight gradients (every dimension of those
// products is a multiple of 384: with 256-row tiles 14 % of their MFMA work fell on rows past the edge).
template <int BN, int BMT = 256> struct Cfg {
    // (384 x 128: a 3-stage ring -- 96 instead of 128 KB of LDS -- measured the same, 311.5 / 313.9 us per launch and 45.88 / 45.90 ms per
    // two-stream step, interleaved on one box)
    static constexpr int STAGES = (BN == 256 || BMT == 384) ? 4 : 3;
    static constexpr int A_BYTES = BMT * BK * 2;                       // 16384 (24576 for 384 rows)
    static constexpr int B_BYTES = BN * BK * 2;                        // 16384 / 8192
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;              // 32768 / 24576
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;            // 131072 / 73728
    static constexpr int LOADS_PER_TILE = STAGE_BYTES / 1024 / 8;      // LDS-DMA instructions per wave per K tile: 4 / 3
    static constexpr int WAVES_N = BN / 64;                            // 4 / 2
    static constexpr int WAVES_M = 8 / WAVES_N;                        // 2 / 4
    static constexpr int MI = BMT / WAVES_M / 16;                      // m-fragments per wave: 8 / 4 (6 for 384 x 128)
    static constexpr int CP_BF16 = BN * 2 + 16;                        // C staging pitch, bf16
    static constexpr int CP_F32 = BN * 4 + 16;                         // fp32
    // C staging: rows per chunk.  The 256-wide tile asks for 132 KiB of LDS (still one workgroup per CU) so that a bf16 tile is
    // staged in ONE pass -- all eight waves write at once, one barrier, whole-row stores -- and an fp32 tile in two
    // (stamps: the two-pass / four-pass forms took 8.6 k cycles per tile, 18-24 % of a K = 768 / 384 tile's lifetime)
    static constexpr int RC_BF16 = 256;
    static constexpr int RC_F32 = 128;
    static constexpr int EPI_BYTES = RC_BF16 * CP_BF16 > RC_F32 * CP_F32 ? RC_BF16 * CP_BF16 : RC_F32 * CP_F32;   // 135168 / 69632
    static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
};

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

// Branch-free pointer select (a ?: on pointers becomes EXEC-masked duplicate LDS-DMA instructions).
__device__ __forceinline__ const bf16_t* sel_ptr(bool ok, const bf16_t* p, const bf16_t* z) {
    const unsigned long long m = 0ull - (unsigned long long)ok;
    return reinterpret_cast<const bf16_t*>((reinterpret_cast<unsigned long long>(p) & m) |
                                           (reinterpret_cast<unsigned long long>(z) & ~m));
}

__device__ __forceinline__ int row_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }  // h = {0,2,3,1}

// ---- HBM -> LDS: this wave's share of one operand tile (ROWS x 32 k) ----------------------------------------
// Gather forms (sparse conv backward): `gidx` replaces the natural row index of the operand -- for a row-form operand the
// PER_WAVE storage rows this lane stages (looked up once per workgroup), for a col-form operand the 4 storage k-rows this
// wave stages of this K tile (wave-uniform scalars).
template <bool TRANS, int ROWS, bool GATHERED = false>
__device__ __forceinline__ void stage_tile(char* lds_tile, const bf16_t* __restrict__ base, long ld, int r0, int R, int k0,
                                           int kend, int wave, int lane, const bf16_t* zero, const int* gidx = nullptr) {
    constexpr int PER_WAVE = ROWS * BK * 2 / 1024 / 8;  // 1-KiB wave-instructions per wave: 2 (256 rows) / 1 (128 rows)
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int j = wave * PER_WAVE + u;
        const bf16_t* src;
        if constexpr (!TRANS) {
            // [ROWS][32 k] image, 64-B rows: LDS position (row, chunk p) holds source chunk p ^ h[(row >> 2) & 3]
            const int row = j * 16 + (lane >> 2);
            const int c = (lane & 3) ^ row_swz(row);
            int grow = r0 + row;
            grow = grow < R ? grow : R - 1;
            if constexpr (GATHERED) grow = gidx[u];
            const int k = k0 + c * 8;
            src = sel_ptr(k < kend, base + (long)grow * ld + k, zero);
        } else if constexpr (GATHERED) {
            static_assert(!GATHERED || ROWS == 256, "col-form gather is built for 256-wide operand tiles");
            constexpr int CPR = ROWS / 8;
            const int krow = j * 2 + (lane >> 5);
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int c = (lane % CPR) ^ (f << 1);
            const int k = k0 + krow;
            const int gk = (lane >> 5) ? gidx[2 * u + 1] : gidx[2 * u];
            const int row = r0 + c * 8;
            src = sel_ptr(k < kend && row < R, base + (long)gk * ld + row, zero);
        } else {
            // [32 k][ROWS] image: LDS position (k row, chunk p) holds source chunk p ^ (f(k) << 1),
            // f(k) = (k & 3) | ((k >> 3) & 1) << 2  -> the 8 k-rows one transposed read touches hit 8 distinct 32-B columns
            constexpr int CPR = ROWS / 8;   // 16-B chunks per k row (16 / 32 / 48: the image is chunk-linear, a wave-instruction = 64 chunks)
            const int q = j * 64 + lane;
            const int krow = q / CPR;
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int c = (q % CPR) ^ (f << 1);        // the XOR stays inside a group of 16 chunks
            const int k = k0 + krow;
            const int row = r0 + c * 8;
            src = sel_ptr(k < kend && row < R, base + (long)k * ld + row, zero);
        }
        __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)(lds_tile + j * 1024), 16, 0, 0);
    }
}

// Loop-invariant form of stage_tile for K tiles that lie entirely inside [kbeg, kend): the per-lane source pointers are
// computed once and advanced by a constant per tile.  Why it matters: while the partner wave of a SIMD runs its MFMA cluster,
// the loading wave gets ONE vector-issue slot per 16-cycle MFMA, so every VALU instruction of address arithmetic in the LOAD
// segment costs ~16 cycles (s_memtime stamps: 730 cycles of LOAD against 557 of COMPUTE with the arithmetic inline).
template <bool TRANS, int ROWS>
struct TilePtrs {
    static constexpr int PER_WAVE = ROWS * BK * 2 / 1024 / 8;
    const char* p[PER_WAVE];
    unsigned step[PER_WAVE];   // bytes per K tile; 0 for lanes parked on the zero page (col form, column out of range)

    __device__ __forceinline__ void init(const bf16_t* __restrict__ base, long ld, int r0, int R, int kbeg, int wave, int lane,
                                         const bf16_t* zero) {
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int j = wave * PER_WAVE + u;
            if constexpr (!TRANS) {
                const int row = j * 16 + (lane >> 2);
                const int c = (lane & 3) ^ row_swz(row);
                int grow = r0 + row;
                grow = grow < R ? grow : R - 1;
                p[u] = reinterpret_cast<const char*>(base + (long)grow * ld + kbeg + c * 8);
                step[u] = BK * 2;
            } else {
                constexpr int CPR = ROWS / 8;
                const int q = j * 64 + lane;
                const int krow = q / CPR;
                const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
                const int c = (q % CPR) ^ (f << 1);
                const int row = r0 + c * 8;
                const bool ok = row < R;
                p[u] = reinterpret_cast<const char*>(sel_ptr(ok, base + (long)(kbeg + krow) * ld + row, zero));
                step[u] = ok ? (unsigned)(BK * ld * 2) : 0u;
            }
        }
    }
    __device__ __forceinline__ void issue_and_advance(char* lds_tile, int wave) {
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) {
            const int j = wave * PER_WAVE + u;
            __builtin_amdgcn_global_load_lds((gl_void*)p[u], (lds_void*)(lds_tile + j * 1024), 16, 0, 0);
            p[u] += step[u];
        }
    }
};

// ---- LDS -> 4 MFMA fragments (tile rows rbase0 + 16x + i, k = 8g + 0..7; i = lane&15, g = lane>>4) ---------------
template <bool TRANS, int ROWS>
__device__ __forceinline__ void load_frags4(bf16x8* f, const char* tile, int rbase0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if constexpr (!TRANS) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int row = rbase0 + x * 16 + i;
            f[x] = *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((g ^ row_swz(row)) << 4));
        }
    } else {
        const int q = i >> 2, p = i & 3;
        const int k = 8 * g + q;
        const int fx = (q | ((g & 1) << 2)) << 5;   // f(k) == f(k + 4), as a byte XOR on the 32-B column
        const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)tile + k * (ROWS * 2);
        const unsigned a0 = base + ((((rbase0 + 0) + 4 * p) << 1) ^ fx);
        const unsigned a1 = base + ((((rbase0 + 16) + 4 * p) << 1) ^ fx);
        const unsigned a2 = base + ((((rbase0 + 32) + 4 * p) << 1) ^ fx);
        const unsigned a3 = base + ((((rbase0 + 48) + 4 * p) << 1) ^ fx);
        bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\t"
            "ds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
            "ds_read_b64_tr_b16 %2, %9\n\t"
            "ds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
            "ds_read_b64_tr_b16 %4, %10\n\t"
            "ds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
            "ds_read_b64_tr_b16 %6, %11\n\t"
            "ds_read_b64_tr_b16 %7, %11 offset:%12\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(4 * ROWS * 2)
            : "memory");
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
        f[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// two fragments (tile rows rbase0 + 16x + i, x < 2): the tail of a wave's six m-fragments in the 384-row tile
template <bool TRANS, int ROWS>
__device__ __forceinline__ void load_frags2(bf16x8* f, const char* tile, int rbase0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if constexpr (!TRANS) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int row = rbase0 + x * 16 + i;
            f[x] = *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((g ^ row_swz(row)) << 4));
        }
    } else {
        const int q = i >> 2, p = i & 3;
        const int k = 8 * g + q;
        const int fx = (q | ((g & 1) << 2)) << 5;
        const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)tile + k * (ROWS * 2);
        const unsigned a0 = base + ((((rbase0 + 0) + 4 * p) << 1) ^ fx);
        const unsigned a1 = base + ((((rbase0 + 16) + 4 * p) << 1) ^ fx);
        bf16x4 l0, h0, l1, h1;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %4\n\t"
            "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
            "ds_read_b64_tr_b16 %2, %5\n\t"
            "ds_read_b64_tr_b16 %3, %5 offset:%6\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
            : "v"(a0), "v"(a1), "n"(4 * ROWS * 2)
            : "memory");
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

struct EpiArgs {
    void* C;
    void* C2;
    const float* bias;
    const void* aux;
    float* colsum;
    long ldc;
    int seg_rows, seg_valid;
    float alpha;
    const int32_t* rowmap;   // gather forms: storage row of logical row m (GATHER 1) / of logical k (GATHER 2)
    const uint32_t* sa;      // MX fp8 path: E8M0 block scales of A, [K / 128][lds_a] dwords (byte b = k block 4 kt + b)
    const uint32_t* sb;      //              ... and of B, [K / 128][lds_b]
    long lds_a, lds_b;       //              rows (dwords) per K tile in those arrays
    unsigned char* q_out;    // GELU epilogues, optional: gelu(h) also as MX fp8 (bytes [M][ldc]) + block scales [N / 128][ld_q] dwords
    uint32_t* q_scales;
    long ld_q;
    float* pair_ws;          // K-split pairs (eight-phase schedule, pair_split below): [tiles][2 roles][4 waves][32 regs][64 lanes] float4
    unsigned* pair_flags;    //   [tiles][2]: role r's half of the partial sums is posted (reset to 0 by the partner that consumed it)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- eight-phase main loop (256 x 256 x 64 tile, row-form operands, K % 128 == 0) ----------------------------------------
// Follows the 256^2 8-phase structure of the CDNA4 guide (cdna_hip_programming.md section 5), re-derived for this kernel's
// fragment maps:
//   * K tile = 64 deep: LDS rows are 128 B, so every global -> LDS request is a FULL 128-B line (the 32-deep tiles of the other
//     schedules fetch 64-B half lines, which cost the L2 -> LDS path as much as whole lines: DESIGN_EXPERIMENTS.md, round 2).
//   * LDS = 2 parities x [A 256 rows | B 256 rows] x 128 B = 128 KiB; 16-B chunk c of row r sits at chunk position
//     c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups (swizzle on the per-lane SOURCE address of the LDS-DMA
//     and again on the read).
//   * a K tile is four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each (16 MFMAs); every phase =
//     [fragment reads + LDS-DMA issue] -> barrier -> [MFMA cluster] -> barrier.  Waves 4-7 run one barrier behind waves 0-3,
//     so on every SIMD one wave's MFMA cluster overlaps its partner's reads / DMA issue.
//   * B fragments are read in phases 0 / 1 only (the first ones stay in registers for phase 3), A fragments in phases 0 / 2; a slot
//     is re-staged >= 2 phases after its last read, and data is first read one phase after the counted wait that retires it (the
//     staggered group's wait sits one barrier later, still before that read).
// Staging by USAGE PHASE ("one 16-KiB piece per phase").  The first version of this loop staged 128-row HALF tiles: the A rows of tile
// t+1 went out in phases 0 / 1 of tile t and were waited for in phase 3 -- two to three phases (~1 k cycles) of flight, less than an L2
// miss takes -- because a half tile of A is read in phases 0 AND 2 and may only be restaged two phases after its last read (8192^3:
// 1045 TFLOP/s; the loop stalled on that wait).  Now an operand tile is cut by the phase that READS it (8192^3: 1359 TFLOP/s, the
// teacher's in_proj 852 -> 937, linear2 1026 -> 1082; outputs bit-identical):
//     X  = A rows 0-63   of each wave row  (read in phase 0)        Y  = A rows 64-127 (phase 2)
//     B0 = B rows 0-31   of each wave col  (phase 0, kept for 3)    B1 = B rows 32-63  (phase 1)
// each 128 rows x 128 B = 16 KiB = 2 LDS-DMA instructions per wave, so a piece's slot is free two phases after ITS read and every
// phase stages exactly one piece:   P0: Y(t+1)   P1: B1(t+1)   P2: B0(t+2)   P3: X(t+2)      (flight: 6 / 4 / 6 / 5 phases)
// Two counted waits per tile: phase 3 retires X(t+1), B0(t+1) (vmcnt(8): Y(t+1), B1(t+1), B0(t+2), X(t+2) stay in flight), phase 0
// retires B1(t) and with it the older Y(t) (vmcnt(6)); each is followed by a barrier before the phase that reads the data.
template <int PAR>
__device__ __forceinline__ void ep_tile(f32x4 (&acc)[8][4], char* smem, const char* (&px)[2], const char* (&py)[2], const char* (&pb0)[2],
                                         const char* (&pb1)[2], const unsigned (&dx)[2], const unsigned (&db)[2], unsigned a_lo, unsigned b_lo,
                                         bool more1, bool more2) {
    constexpr unsigned BUF = 65536u, BOFF = 32768u;
    char* cur = smem + PAR * BUF;
    char* oth = smem + (PAR ^ 1) * BUF;
    const unsigned a_hi = a_lo ^ 64u, b_hi = b_lo ^ 64u;
    bf16x8 af[8], b0f[4], b1f[4];
    auto lds = [&](unsigned off) { return *reinterpret_cast<const bf16x8*>(cur + off); };
    auto dma = [&](const char*& src, char* dst) {
        __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)dst, 16, 0, 0);
        src += 128;
    };
    // ---- phase 0: X, B0 of this tile; stage Y(t+1); wait for B1(t)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b0f[2 * x] = lds(b_lo + x * 2048); b0f[2 * x + 1] = lds(b_hi + x * 2048); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[2 * x] = lds(a_lo + x * 2048); af[2 * x + 1] = lds(a_hi + x * 2048); }
    if (more1) {
        dma(py[0], oth + dx[0] + 8192); dma(py[1], oth + dx[1] + 8192);
        wait_vmcnt<6>();
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
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni], af[2 * mi], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni + 1], af[2 * mi + 1], acc[mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: B1 of this tile; stage B1(t+1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b1f[2 * x] = lds(b_lo + 4096 + x * 2048); b1f[2 * x + 1] = lds(b_hi + 4096 + x * 2048); }
    if (more1) { dma(pb1[0], oth + BOFF + db[0] + 4096); dma(pb1[1], oth + BOFF + db[1] + 4096); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni], af[2 * mi], acc[mi][2 + ni], 0, 0, 0);
            acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni + 1], af[2 * mi + 1], acc[mi][2 + ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: Y of this tile; stage B0(t+2) into THIS parity (its B0 rows were read two phases ago)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[2 * x] = lds(a_lo + 8192 + x * 2048); af[2 * x + 1] = lds(a_hi + 8192 + x * 2048); }
    if (more2) { dma(pb0[0], cur + BOFF + db[0]); dma(pb0[1], cur + BOFF + db[1]); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc[4 + mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni], af[2 * mi], acc[4 + mi][2 + ni], 0, 0, 0);
            acc[4 + mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[2 * ni + 1], af[2 * mi + 1], acc[4 + mi][2 + ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: no fragment reads; stage X(t+2) into THIS parity (read in phase 0); wait for X(t+1), B0(t+1)
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
        dma(px[0], cur + dx[0]); dma(px[1], cur + dx[1]);
        wait_vmcnt<8>();
    } else if (more1) {
        wait_vmcnt<4>();
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
            acc[4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni], af[2 * mi], acc[4 + mi][ni], 0, 0, 0);
            acc[4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[2 * ni + 1], af[2 * mi + 1], acc[4 + mi][ni], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void eight_phase_loop(f32x4 (&acc)[8][4], char* smem, const bf16_t* __restrict__ A,
                                                  const bf16_t* __restrict__ B, long lda, long ldb, int m0, int n0, int M, int N, int K,
                                                  int wave, int lane) {
    constexpr unsigned BUF = 65536u, BOFF = 32768u;
    const int nkt = K / 64;                       // even (K % 128 == 0)
    // this wave's two instructions of a piece: 8 rows each.  X rows: wave rows 0-63 of wave row 0 (waves 0-3) / 1 (waves 4-7);
    // B0 rows: 32 of each wave column, waves pair up on a column.  Y = X + 64 rows, B1 = B0 + 32 rows.
    const char* px[2]; const char* py[2]; const char* pb0[2]; const char* pb1[2];
    unsigned dx[2], db[2];                        // LDS byte offsets (inside the A / B region) of the X / B0 instructions
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rx = (wave < 4 ? 16 * wave : 128 + 16 * (wave - 4)) + 8 * u;         // first row of the 8-row group
        const int ib = 16 * wave + 8 * u, rb = (ib >> 5) * 64 + (ib & 31);
        dx[u] = (unsigned)rx * 128u;
        db[u] = (unsigned)rb * 128u;
        auto src = [&](const bf16_t* base, long ld, int r0, int R, int row) {
            const int c = (lane & 7) ^ ((row >> 1) & 7);                             // LDS chunk position lane % 8 holds source chunk c
            int gr = r0 + row;
            gr = gr < R ? gr : R - 1;                                                // past the edge: clamped, never stored
            return reinterpret_cast<const char*>(base + (long)gr * ld + c * 8);
        };
        px[u] = src(A, lda, m0, M, rx + (lane >> 3));
        py[u] = src(A, lda, m0, M, rx + 64 + (lane >> 3));
        pb0[u] = src(B, ldb, n0, N, rb + (lane >> 3));
        pb1[u] = src(B, ldb, n0, N, rb + 32 + (lane >> 3));
    }
    auto dma = [&](const char*& s_, char* dst) {
        __builtin_amdgcn_global_load_lds((gl_void*)s_, (lds_void*)dst, 16, 0, 0);
        s_ += 128;
    };
    // prologue: tile 0 (parity 0) in the order its phases need it, then what "phases 2 / 3 of tile -1" would have staged of tile 1
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(px[u], smem + dx[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(pb0[u], smem + BOFF + db[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(py[u], smem + dx[u] + 8192);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(pb1[u], smem + BOFF + db[u] + 4096);
    if (nkt > 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) dma(pb0[u], smem + BUF + BOFF + db[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) dma(px[u], smem + BUF + dx[u]);
        wait_vmcnt<4>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const int i = lane & 15, g = lane >> 4, wm = wave >> 2, wn = wave & 3;
    const unsigned sw = (unsigned)((g ^ ((i >> 1) & 7)) << 4);
    const unsigned a_lo = (unsigned)((wm * 128 + i) * 128) + sw;
    const unsigned b_lo = BOFF + (unsigned)((wn * 64 + i) * 128) + sw;
    if (wm == 1) __builtin_amdgcn_s_barrier();    // waves 4-7 run one barrier behind
    for (int t = 0; t < nkt; t += 2) {
        ep_tile<0>(acc, smem, px, py, pb0, pb1, dx, db, a_lo, b_lo, t + 1 < nkt, t + 2 < nkt);
        ep_tile<1>(acc, smem, px, py, pb0, pb1, dx, db, a_lo, b_lo, t + 2 < nkt, t + 3 < nkt);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();    // balance the stagger
}

// ---- MX fp8 form of the eight-phase loop (256 x 256 x 128 tile) --------------------------------------------------------------
// Operands are OCP e4m3 bytes with one E8M0 scale per 32 consecutive k (block-scaled v_mfma_scale_f32_16x16x128_f8f6f4: twice the
// bf16 MFMA rate per byte of operand).  A K tile is 128 deep = the SAME 128-B LDS rows, swizzle, staging and phase structure as
// the bf16 loop above; the two 16-B chunks a lane reads per fragment (chunk g and chunk 4 + g) are the two halves of the MX
// operand (measured with tools/micro/mx_fp8_probe.hip: lane (i, g) supplies k = 16 g .. 16 g + 15 in registers 0-3 and
// k = 64 + 16 g .. in registers 4-7, and the scale byte of k block g of row i).  One MFMA per 16 x 16 fragment and K tile.
// Block scales: [K / 128][rows] dwords (byte b = block 4 kt + b), 1 KiB per operand and K tile, staged by LDS-DMA next to the
// ring (every wave moves 128 B of each, so the counted waits stay uniform) one K tile ahead.
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// The builtin form of the scaled MFMA is allocated with a destination DISTINCT from its accumulator input (early-clobber), which
// at 128 accumulator registers per lane spills ~150 of them to scratch (and every scratch access is a vmcnt(0) in the LDS-DMA
// ring).  In hardware vdst == srcC is the ordinary accumulate form, so the instruction is written out with the two tied.
// "s_nop 1": VALU-written scale registers feed the MFMA (hipcc pads nothing inside an asm statement).
__device__ __forceinline__ void mx_mfma(f32x4& acc, const v8i32& a, const v8i32& b, int scale_a, int scale_b) {
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                 : "+v"(acc) : "v"(a), "v"(b), "v"(scale_a), "v"(scale_b));
}

constexpr unsigned F8_SCALE_OFF = 131072u;   // LDS: [parity][A 1 KiB | B 1 KiB] after the two 64-KiB operand parities

template <int PAR>
__device__ __forceinline__ void ep8_tile(f32x4 (&acc)[8][4], char* smem, const char* __restrict__ A, const char* __restrict__ B,
                                         const char* __restrict__ SA, const char* __restrict__ SB, unsigned (&ox)[2], unsigned (&oy)[2],
                                         unsigned (&ob0)[2], unsigned (&ob1)[2], const unsigned (&dx)[2], const unsigned (&db)[2],
                                         unsigned& osa, unsigned& osb, unsigned sa_step, unsigned sb_step, unsigned a_lo, unsigned b_lo,
                                         int wave, int lane, bool more1, bool more2) {
    constexpr unsigned BUF = 65536u, BOFF = 32768u;
    char* cur = smem + PAR * BUF;
    char* oth = smem + (PAR ^ 1) * BUF;
    const unsigned a_hi = a_lo ^ 64u, b_hi = b_lo ^ 64u;
    const int i = lane & 15, g = lane >> 4, wm = wave >> 2, wn = wave & 3;
    const char* sc = smem + F8_SCALE_OFF + PAR * 2048;          // this tile's scales
    char* sc_o = smem + F8_SCALE_OFF + (PAR ^ 1) * 2048;        // next tile's
    v8i32 af[4], b0f[2], b1f[2];             // one MX operand = chunk g (k 16 g ..) | chunk 4 + g (k 64 + 16 g ..) of a 128-B row
    int sa[4], sb0[2], sb1[2];
    auto lds8 = [&](unsigned lo, unsigned hi) {
        const v4i32 x = *reinterpret_cast<const v4i32*>(cur + lo), y = *reinterpret_cast<const v4i32*>(cur + hi);
        return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto scale = [&](const char* base, int row) { return (int)(*reinterpret_cast<const uint32_t*>(base + row * 4) >> (8 * g)); };
    // LDS-DMA source = uniform matrix base + 32-bit per-lane offset (one register per stream instead of a 64-bit pointer)
    auto dma = [&](const char* base, unsigned& off, char* dst) {
        __builtin_amdgcn_global_load_lds((gl_void*)(base + off), (lds_void*)dst, 16, 0, 0);
        off += 128;
    };
    // Staging by usage phase as in the bf16 loop (X / Y / B0 / B1 pieces, one per phase), plus the next tile's block scales, issued
    // FIRST in phase 0 (their buffer was last read in phase 2 of the previous tile).  Stream order per tile:
    //   P0: scales(t+1), Y(t+1)   P1: B1(t+1)   P2: B0(t+2)   P3: X(t+2)
    // phase 0's wait retires B1(t) (vmcnt(8): B0(t+1), X(t+1), scales(t+1), Y(t+1) stay in flight), phase 3's retires X(t+1),
    // B0(t+1) and scales(t+1) (vmcnt(8): Y(t+1), B1(t+1), B0(t+2), X(t+2) stay in flight).
    // ---- phase 0: A rows 0-63, B cols 0-31
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b0f[x] = lds8(b_lo + x * 2048, b_hi + x * 2048); sb0[x] = scale(sc + 1024, wn * 64 + x * 16 + i); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[x] = lds8(a_lo + x * 2048, a_hi + x * 2048); sa[x] = scale(sc, wm * 128 + x * 16 + i); }
    if (more1) {
        if (lane < 8) {                       // this wave's 128 B of the next tile's A / B block scales (rows 32 wave .. +31)
            __builtin_amdgcn_global_load_lds((gl_void*)(SA + osa), (lds_void*)(sc_o + wave * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gl_void*)(SB + osb), (lds_void*)(sc_o + 1024 + wave * 128), 16, 0, 0);
        }
        osa += sa_step; osb += sb_step;
        dma(A, oy[0], oth + dx[0] + 8192); dma(A, oy[1], oth + dx[1] + 8192);
        wait_vmcnt<8>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            mx_mfma(acc[mi][ni], b0f[ni], af[mi], sb0[ni], sa[mi]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: B cols 32-63
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 2; ++x) { b1f[x] = lds8(b_lo + 4096 + x * 2048, b_hi + 4096 + x * 2048); sb1[x] = scale(sc + 1024, wn * 64 + 32 + x * 16 + i); }
    if (more1) { dma(B, ob1[0], oth + BOFF + db[0] + 4096); dma(B, ob1[1], oth + BOFF + db[1] + 4096); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            mx_mfma(acc[mi][2 + ni], b1f[ni], af[mi], sb1[ni], sa[mi]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: A rows 64-127
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) { af[x] = lds8(a_lo + 8192 + x * 2048, a_hi + 8192 + x * 2048); sa[x] = scale(sc, wm * 128 + 64 + x * 16 + i); }
    if (more2) { dma(B, ob0[0], cur + BOFF + db[0]); dma(B, ob0[1], cur + BOFF + db[1]); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            mx_mfma(acc[4 + mi][2 + ni], b1f[ni], af[mi], sb1[ni], sa[mi]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: no fragment reads
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
        dma(A, ox[0], cur + dx[0]); dma(A, ox[1], cur + dx[1]);
        wait_vmcnt<8>();
    } else if (more1) {
        wait_vmcnt<4>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            mx_mfma(acc[4 + mi][ni], b0f[ni], af[mi], sb0[ni], sa[mi]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

// A, B: e4m3 bytes, [M][K] / [N][K] (K contiguous, leading dimensions in BYTES, matrices < 4 GiB); K % 256 == 0
__device__ __forceinline__ void eight_phase_loop_fp8(f32x4 (&acc)[8][4], char* smem, const char* __restrict__ A, const char* __restrict__ B,
                                                     long lda, long ldb, int m0, int n0, int M, int N, int K, const EpiArgs& e, int wave,
                                                     int lane) {
    constexpr unsigned BUF = 65536u, BOFF = 32768u;
    const int nkt = K / 128;
    unsigned ox[2], oy[2], ob0[2], ob1[2], dx[2], db[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int rx = (wave < 4 ? 16 * wave : 128 + 16 * (wave - 4)) + 8 * u;
        const int ib = 16 * wave + 8 * u, rb = (ib >> 5) * 64 + (ib & 31);
        dx[u] = (unsigned)rx * 128u;
        db[u] = (unsigned)rb * 128u;
        auto off = [&](long ld, int r0, int R, int row) {
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int gr = r0 + row;
            gr = gr < R ? gr : R - 1;
            return (unsigned)((long)gr * ld + c * 16);
        };
        ox[u] = off(lda, m0, M, rx + (lane >> 3));
        oy[u] = off(lda, m0, M, rx + 64 + (lane >> 3));
        ob0[u] = off(ldb, n0, N, rb + (lane >> 3));
        ob1[u] = off(ldb, n0, N, rb + 32 + (lane >> 3));
    }
    // block scales: this wave's 32 rows of the tile, 4 rows (16 B) per lane of lanes 0-7; rows past the edge read the padding
    const char* SA = reinterpret_cast<const char*>(e.sa);
    const char* SB = reinterpret_cast<const char*>(e.sb);
    unsigned osa = (unsigned)(m0 + wave * 32 + (lane & 7) * 4) * 4u, osb = (unsigned)(n0 + wave * 32 + (lane & 7) * 4) * 4u;
    const unsigned sa_step = (unsigned)(e.lds_a * 4), sb_step = (unsigned)(e.lds_b * 4);
    auto dma = [&](const char* base, unsigned& off, char* dst) {
        __builtin_amdgcn_global_load_lds((gl_void*)(base + off), (lds_void*)dst, 16, 0, 0);
        off += 128;
    };
    if (lane < 8) {
        __builtin_amdgcn_global_load_lds((gl_void*)(SA + osa), (lds_void*)(smem + F8_SCALE_OFF + wave * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_void*)(SB + osb), (lds_void*)(smem + F8_SCALE_OFF + 1024 + wave * 128), 16, 0, 0);
    }
    osa += sa_step; osb += sb_step;
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(A, ox[u], smem + dx[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(B, ob0[u], smem + BOFF + db[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(A, oy[u], smem + dx[u] + 8192);
#pragma unroll
    for (int u = 0; u < 2; ++u) dma(B, ob1[u], smem + BOFF + db[u] + 4096);
    if (nkt > 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) dma(B, ob0[u], smem + BUF + BOFF + db[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) dma(A, ox[u], smem + BUF + dx[u]);
        wait_vmcnt<4>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const int i = lane & 15, g = lane >> 4, wm = wave >> 2, wn = wave & 3;
    const unsigned sw = (unsigned)((g ^ ((i >> 1) & 7)) << 4);
    const unsigned a_lo = (unsigned)((wm * 128 + i) * 128) + sw;
    const unsigned b_lo = BOFF + (unsigned)((wn * 64 + i) * 128) + sw;
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nkt; t += 2) {
        ep8_tile<0>(acc, smem, A, B, SA, SB, ox, oy, ob0, ob1, dx, db, osa, osb, sa_step, sb_step, a_lo, b_lo, wave, lane, t + 1 < nkt, t + 2 < nkt);
        ep8_tile<1>(acc, smem, A, B, SA, SB, ox, oy, ob0, ob1, dx, db, osa, osb, sa_step, sb_step, a_lo, b_lo, wave, lane, t + 2 < nkt, t + 3 < nkt);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA results -> compiler-scheduled readers (the epilogue)
    if (wm == 0) __builtin_amdgcn_s_barrier();
}

// GATHER 0: dense.  1: logical row m of A (row form) and of C lives at storage row rowmap[m] (conv dgrad over the active
// rows).  2: logical k of A and B (both col form) lives at storage row rowmap[k] (conv wgrad over the active rows;
// the list is padded with >= 256 readable entries).
// The body of one workgroup: `bid` of `nwg` workgroups of ONE problem (the plain kernel passes blockIdx / gridDim; the grouped
// kernel below passes the workgroup's position inside its problem of the group).
template <bool AT, bool BT, int EPI, int BN, int SCHED, int GATHER = 0, int BMT = 256>
__device__ __forceinline__ void gemm3_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, long lda, long ldb, int M, int N,
                                           int K, int tiles_n, int split_k, int k_per_split, const EpiArgs& e, int bid, int nwg,
                                           bool remapped = false) {
    // remapped: `bid` already is the logical (XCD-grouped) position inside the problem (grouped launches remap over the whole group)
    using C_ = Cfg<BN, BMT>;
    constexpr int BM = BMT;              // (shadows the file-level 256 inside this body)
    static_assert(BMT == 256 || (BMT == 384 && BN == 128 && SCHED < 2 && GATHER == 0), "384-row tiles: 384 x 128, plain / ping-pong schedule");
    constexpr bool STAGGER = SCHED == 1;
    static_assert(SCHED == 0 || BN == 256 || (SCHED == 1 && BMT == 384), "the ping-pong / eight-phase schedules are built for one 8-wave workgroup per CU");
    static_assert(GATHER == 0 || SCHED == 0, "gather forms use the plain schedule");
    static_assert(SCHED < 2 || (!AT && !BT), "eight-phase schedules: row-form operands");
    static_assert(SCHED != 3 || EPI == WJ_EPI_BF16 || EPI == WJ_EPI_BIAS_GELU2, "MX fp8 loop: forward epilogues");
    static_assert(GATHER != 1 || (!AT && (EPI == WJ_EPI_BF16 || EPI == WJ_EPI_MUL_GELU_GRAD_Z)), "row gather: row-form A, bf16 output");
    static_assert(EPI != WJ_EPI_MUL_GELU_GRAD_Z || GATHER == 1, "gelu'(z) epilogue: built for the sparse conv dgrad");
    static_assert(GATHER != 2 || (AT && BT && BN == 256), "k gather: col-form A and B, 256-wide tiles");
    constexpr int S = C_::STAGES, A_BYTES = C_::A_BYTES, STAGE_BYTES = C_::STAGE_BYTES, LPT = C_::LOADS_PER_TILE, MI = C_::MI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / C_::WAVES_N, wn = wave % C_::WAVES_N;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

    // K-split PAIRS (eight-phase schedule, SCHED 2, split_k == 2; launch_pair below): problems with 64-128 output tiles leave half of
    // the 256 CUs idle (the ragged student: M ~ 10 k rows x N = 768 = 117 tiles).  Two workgroups share a tile, each multiplies half of
    // K, then they exchange HALF of their fp32 partial sums through `pair_ws` -- role r keeps tile rows [128 r, 128 r + 128), i.e. its
    // waves with wm == r, and posts the accumulators of its other four waves -- and each finishes its half of the tile.  The exchange
    // is layout-free: a wave stores its accumulator registers as they are (register x lane, 16-byte coalesced) and the partner's wave
    // with the same index reads them back into the same registers.
    // Mapping: an XCD (bid % 8) owns a contiguous run of tiles; its workgroups j = bid / 8 are first the role-0 then the role-1
    // workgroups of those tiles, so both halves of an A panel stay on one L2 and the tiles of a panel run side by side.  Posting never
    // blocks and a pair's workgroups are `len` dispatch slots apart on one XCD: a workgroup that waits for its partner cannot starve it.
    constexpr bool PAIR_OK = SCHED == 2 && EPI == WJ_EPI_BF16;
    const bool pair = PAIR_OK && split_k == 2;
    int wg = 0, role = 0;
    if (pair) {
        const int nt = tiles_n * ((M + BM - 1) / BM);      // tiles of the problem
        const int x = bid & 7, j = bid >> 3, tq = nt >> 3, tr = nt & 7;
        const int len = tq + (x < tr ? 1 : 0), start = x < tr ? x * (tq + 1) : tr * (tq + 1) + (x - tr) * tq;
        if (j >= 2 * len) return;                 // grid padding
        role = j >= len ? 1 : 0;
        wg = role * nt + start + (role ? j - len : j);
    } else {
        wg = remapped ? bid : xcd_remap(bid, nwg);
    }
    // split-K work order: K-slice major, tile minor -> the tiles that stream the same K-slice of A / B are neighbours on one
    // XCD and share it through L2 (tile-major order fetched ~3x the algorithmic bytes: profiles/r01_pmc_traffic.json)
    const int ntiles = pair ? tiles_n * ((M + BM - 1) / BM) : nwg / split_k;
    const int ksl = wg / ntiles, tile = wg - ksl * ntiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = ksl * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (EPI == WJ_EPI_ATOMIC_F32 && nkt <= 0) return;

    f32x4 acc[MI][4];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // gather indices
    int arow[2] = {0, 0};                 // GATHER 1: storage rows of the two A rows this lane stages
    if constexpr (GATHER == 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lr = m0 + (wave * 2 + u) * 16 + (lane >> 2);
            arow[u] = e.rowmap[lr < M ? lr : M - 1];
        }
    }
    int kidx[4] = {0, 0, 0, 0};           // GATHER 2: storage k-rows 4*wave .. +3 of the tile about to be staged (scalars)
    auto load_kidx = [&](int k0) {
        // wave-uniform address in the CONSTANT address space: an s_load_dwordx4 (lgkmcnt), not a vector load whose
        // vmcnt wait would drain the LDS-DMA ring
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(4))) const i32x4 const_i32x4;
        const i32x4 v = *reinterpret_cast<const_i32x4*>(reinterpret_cast<unsigned long long>(e.rowmap + k0 + 4 * wave));
        kidx[0] = v[0]; kidx[1] = v[1]; kidx[2] = v[2]; kidx[3] = v[3];
    };

    // dense forms: loop-invariant source pointers (tiles inside [kbeg, kend)); the K tail tile, if any, takes stage_tile
    TilePtrs<AT, BM> pa;
    TilePtrs<BT, BN> pb;
    if constexpr (GATHER == 0) {
        pa.init(A, lda, m0, M, kbeg, wave, lane, zero);
        pb.init(B, ldb, n0, N, kbeg, wave, lane, zero);
    }
    auto stage_dense = [&](char* st, int k0) {
        if (k0 + BK <= kend) {
            pa.issue_and_advance(st, wave);
            pb.issue_and_advance(st + A_BYTES, wave);
        } else {
            stage_tile<AT, BM>(st, A, lda, m0, M, k0, kend, wave, lane, zero);
            stage_tile<BT, BN>(st + A_BYTES, B, ldb, n0, N, k0, kend, wave, lane, zero);
        }
    };

    if constexpr (SCHED == 3) {
        eight_phase_loop_fp8(acc, smem, reinterpret_cast<const char*>(A), reinterpret_cast<const char*>(B), lda, ldb, m0, n0, M, N, K, e, wave, lane);
    } else if constexpr (SCHED == 2) {
        eight_phase_loop(acc, smem, A + kbeg, B + kbeg, lda, ldb, m0, n0, M, N, kend - kbeg, wave, lane);
        if constexpr (PAIR_OK) {
            if (pair) {
                // post the accumulators of the waves whose rows the partner finishes, take the partner's for the rows kept here.
                // Coherence between the two workgroups (different CUs, possibly different XCDs = different L2s) is carried by the
                // ACCESSES, not by fences: agent-scope (sc1) stores write through the L2, agent-scope loads do not hit stale lines.
                // (Release / acquire fences -- buffer_wbl2 / buffer_inv of the whole L2 from 234 workgroups -- cost ~90 us per launch.)
                char* ws = reinterpret_cast<char*>(e.pair_ws) + (size_t)tile * PAIR_TILE_BYTES;
                unsigned* flags = e.pair_flags + tile * 2;
                const long slot = ((long)((wave & 3) * 32) * 64 + lane) * 16;
                if (wm != role) {
                    char* dst = ws + (size_t)role * (PAIR_TILE_BYTES / 2) + slot;
#pragma unroll
                    for (int a = 0; a < 8; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + (a * 4 + b) * 1024), "v"(acc[a][b]) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partials have reached the coherence point ...
                __syncthreads();
                if (t == 0) {
                    __hip_atomic_store(flags + role, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ... before the flag
                    while (__hip_atomic_load(flags + (role ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
                    __hip_atomic_store(flags + (role ^ 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // consumed: ready for the next launch
                }
                __syncthreads();
                if (wm == role) {
                    const char* src = ws + (size_t)(role ^ 1) * (PAIR_TILE_BYTES / 2) + slot;
                    // two batches of 16 loads, each one round trip (a batch per accumulator row made eight dependent round trips of
                    // ~2 us while all 234 workgroups exchange at once)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 p[16];
                        asm volatile(
                            "global_load_dwordx4 %0, %16, off sc1\n\t"
                            "global_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"
                            "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\t"
                            "global_load_dwordx4 %3, %16, off offset:3072 sc1\n\t"
                            "global_load_dwordx4 %4, %17, off sc1\n\t"
                            "global_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"
                            "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\t"
                            "global_load_dwordx4 %7, %17, off offset:3072 sc1\n\t"
                            "global_load_dwordx4 %8, %18, off sc1\n\t"
                            "global_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"
                            "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\t"
                            "global_load_dwordx4 %11, %18, off offset:3072 sc1\n\t"
                            "global_load_dwordx4 %12, %19, off sc1\n\t"
                            "global_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"
                            "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\t"
                            "global_load_dwordx4 %15, %19, off offset:3072 sc1\n\t"
                            "s_waitcnt vmcnt(0)"
                            : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]),
                              "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13]), "=&v"(p[14]), "=&v"(p[15])
                            : "v"(src + (h * 4 + 0) * 4096), "v"(src + (h * 4 + 1) * 4096), "v"(src + (h * 4 + 2) * 4096), "v"(src + (h * 4 + 3) * 4096)
                            : "memory");
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b) acc[h * 4 + a][b] += p[a * 4 + b];
                    }
                }
            }
        }
    } else if (nkt > 0) {
#pragma unroll
        for (int p = 0; p < S - 1; ++p) {
            if (p < nkt) {
                if constexpr (GATHER == 2) {
                    load_kidx(kbeg + p * BK);
                    stage_tile<AT, BM, true>(smem + p * STAGE_BYTES, A, lda, m0, M, kbeg + p * BK, kend, wave, lane, zero, kidx);
                    stage_tile<BT, BN, true>(smem + p * STAGE_BYTES + A_BYTES, B, ldb, n0, N, kbeg + p * BK, kend, wave, lane, zero, kidx);
                } else if constexpr (GATHER == 1) {
                    stage_tile<AT, BM, true>(smem + p * STAGE_BYTES, A, lda, m0, M, kbeg + p * BK, kend, wave, lane, zero, arow);
                    stage_tile<BT, BN>(smem + p * STAGE_BYTES + A_BYTES, B, ldb, n0, N, kbeg + p * BK, kend, wave, lane, zero);
                } else {
                    stage_dense(smem + p * STAGE_BYTES, kbeg + p * BK);
                }
            }
        }
        if constexpr (GATHER == 2) load_kidx(kbeg + (S - 1) * BK);   // indices of the tile staged in iteration 0
        int cur = 0;                      // stage holding tile kt
        if constexpr (!STAGGER) {
            for (int kt = 0; kt < nkt; ++kt) {
                // tile kt has landed once all but this wave's loads of the younger in-flight tiles are done; the barrier makes
                // every wave's share visible and proves every wave is past its reads of the stage refilled below (tile kt-1).
                const int younger = min(S - 2, nkt - 1 - kt);
                if (younger >= 2) wait_vmcnt<2 * LPT>();
                else if (younger == 1) wait_vmcnt<LPT>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (kt + S - 1 < nkt) {
                    const int nxt = cur == 0 ? S - 1 : cur - 1;   // (cur + S - 1) % S
                    char* st = smem + nxt * STAGE_BYTES;
                    const int k0 = kbeg + (kt + S - 1) * BK;
                    if constexpr (GATHER == 2) {
                        stage_tile<AT, BM, true>(st, A, lda, m0, M, k0, kend, wave, lane, zero, kidx);
                        stage_tile<BT, BN, true>(st + A_BYTES, B, ldb, n0, N, k0, kend, wave, lane, zero, kidx);
                        load_kidx(k0 + BK);                       // next iteration's tile: the scalar load has a whole tile to land
                    } else if constexpr (GATHER == 1) {
                        stage_tile<AT, BM, true>(st, A, lda, m0, M, k0, kend, wave, lane, zero, arow);
                        stage_tile<BT, BN>(st + A_BYTES, B, ldb, n0, N, k0, kend, wave, lane, zero);
                    } else {
                        stage_dense(st, k0);
                    }
                }
                const char* sa = smem + cur * STAGE_BYTES;
                const char* sb = sa + A_BYTES;
                bf16x8 af[MI], bfr[4];
                load_frags4<BT, BN>(bfr, sb, wn * 64, lane);
#pragma unroll
                for (int h = 0; h < MI / 4; ++h) load_frags4<AT, BM>(af + 4 * h, sa, wm * (MI * 16) + h * 64, lane);
                if constexpr (MI % 4 == 2) load_frags2<AT, BM>(af + MI - 2, sa, wm * (MI * 16) + (MI - 2) * 16, lane);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
                cur = cur == S - 1 ? 0 : cur + 1;
            }
        } else {
            // Ping-pong: waves 0-3 (group 0) and 4-7 (group 1) share the four SIMDs pairwise and run the same
            // LOAD / COMPUTE segments one barrier apart, so on every SIMD one wave issues its 32-MFMA cluster while its
            // partner issues the next LDS-DMA tile and reads its fragments.  Two barriers per K tile; every wave waits
            // (counted vmcnt) for tile kt+1 in the segment before group 0 starts reading it.
            // (Interleaving the LDS-DMA issue into the MFMA cluster instead was measured slower: ceiling 1133 -> 843 TFLOP/s.)
            const int grp = wave >> 2;
            {   // tile 0 landed and visible
                const int younger = min(S - 2, nkt - 1);
                if (younger >= 2) wait_vmcnt<2 * LPT>();
                else if (younger == 1) wait_vmcnt<LPT>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 starts one segment later
            for (int kt = 0; kt < nkt; ++kt) {
                // ---- LOAD segment
                __builtin_amdgcn_sched_barrier(0);
#ifndef WJ_ABLATE_GLDS
                if (kt + S - 1 < nkt) {
                    const int nxt = cur == 0 ? S - 1 : cur - 1;   // stage of tile kt-1: both groups read it >= 1 barrier ago
                    char* st = smem + nxt * STAGE_BYTES;
                    stage_dense(st, kbeg + (kt + S - 1) * BK);
                }
#endif
                const char* sa = smem + cur * STAGE_BYTES;
                const char* sb = sa + A_BYTES;
                bf16x8 af[MI], bfr[4];
                load_frags4<BT, BN>(bfr, sb, wn * 64, lane);
#pragma unroll
                for (int h = 0; h < MI / 4; ++h) load_frags4<AT, BM>(af + 4 * h, sa, wm * (MI * 16) + h * 64, lane);
                if constexpr (MI % 4 == 2) load_frags2<AT, BM>(af + MI - 2, sa, wm * (MI * 16) + (MI - 2) * 16, lane);
                const int younger = min(S - 2, nkt - 2 - kt);   // tiles younger than kt+1 still allowed in flight
                if (grp == 1 && kt + 1 < nkt) {
                    if (younger >= 2) wait_vmcnt<2 * LPT>();
                    else if (younger == 1) wait_vmcnt<LPT>();
                    else wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                // ---- COMPUTE segment
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if (grp == 0 && kt + 1 < nkt) {
                    if (younger >= 2) wait_vmcnt<2 * LPT>();
                    else if (younger == 1) wait_vmcnt<LPT>();
                    else wait_vmcnt<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                cur = cur == S - 1 ? 0 : cur + 1;
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();       // balance the stagger
        }
    }

    // ---- epilogue: accumulators -> LDS (row chunks) -> whole rows -----------------------------------------
    const int i = lane & 15, g = lane >> 4;
    // bias fragments: all four loads issued together BEFORE the barrier (inside the fragment loops they were serialised
    // L2 round trips: 25-35 us per launch on the QKV shape)
    f32x4 bv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == WJ_EPI_BF16 || EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_BF16_ADD_POS) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * g;
            if (e.bias && n < N) bv[ni] = *reinterpret_cast<const f32x4*>(e.bias + n);
        }
    }
    __syncthreads();
    constexpr bool F32_TILE = (EPI == WJ_EPI_ADD_F32 || EPI == WJ_EPI_ATOMIC_F32);
    static_assert(BMT == 256 || EPI == WJ_EPI_ATOMIC_F32, "384-row tiles: the split-K weight-gradient epilogue (three 128-row chunks)");
    constexpr int RC = F32_TILE ? C_::RC_F32 : C_::RC_BF16;
    constexpr int CP = F32_TILE ? C_::CP_F32 : C_::CP_BF16;
    constexpr int NCHUNK = BM / RC;
    constexpr int ROWS_PER_WAVE = MI * 16;

#pragma unroll 1
    for (int ch = 0; ch < NCHUNK; ++ch) {
        const int r_lo = ch * RC;
        // -- write this chunk's rows
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mrow = wm * ROWS_PER_WAVE + mi * 16;       // wave-uniform
            if (mrow >= r_lo && mrow < r_lo + RC && (!pair || wm == role)) {     // (a K-split pair: this workgroup finishes the rows of its role only)
                const int m = mrow - r_lo + i;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = wn * 64 + ni * 16 + 4 * g;
                    f32x4 v = acc[mi][ni];
                    if constexpr (F32_TILE) {
                        *reinterpret_cast<f32x4*>(smem + m * CP + n * 4) = v;
                    } else {
                        if constexpr (EPI == WJ_EPI_BF16 || EPI == WJ_EPI_BIAS_GELU2 || EPI == WJ_EPI_BF16_ADD_POS) v += bv[ni];
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
                        *reinterpret_cast<bf16x4*>(smem + m * CP + n * 2) = o;
                    }
                }
            }
        }
        __syncthreads();
        const int mh = m0 + r_lo;
        if constexpr (EPI == WJ_EPI_ATOMIC_F32) {
            // one float per lane: every atomic wave-instruction adds 256 contiguous bytes of one row
            constexpr int TPR = BN;                    // threads per row
            constexpr int RPP = NT / TPR;              // rows per pass: 2 / 4
            const int col = t % TPR, rr = t / TPR;
            const int n = n0 + col;
            if (n < N) {
#pragma unroll 8
                for (int r = rr; r < RC; r += RPP) {
                    const int m = mh + r;
                    if (m < M) {
                        const float v = *reinterpret_cast<const float*>(smem + r * CP + col * 4);
                        atomicAdd((float*)e.C + (long)m * e.ldc + n, v * e.alpha);
                    }
                }
            }
        } else if constexpr (EPI == WJ_EPI_ADD_F32) {
            constexpr int TPR = BN / 4;                // float4 chunks per row: 64 / 32
            constexpr int RPP = NT / TPR;              // 8 / 16
            constexpr int PASSES = RC / RPP;           // 8
            const int c4 = (t % TPR) * 4, rr = t / TPR;
            const int n = n0 + c4;
            const bool ncol = n < N;
            f32x4 ax[PASSES];
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {      // all addend loads in flight before the first use
                const int m = mh + rr + RPP * ps;
                ax[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e.aux && ncol && m < M) ax[ps] = *reinterpret_cast<const f32x4*>((const float*)e.aux + (long)m * e.ldc + n);
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int r = rr + RPP * ps, m = mh + r;
                if (ncol && m < M) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * CP + c4 * 4) + ax[ps];
                    *reinterpret_cast<f32x4*>((float*)e.C + (long)m * e.ldc + n) = v;
                }
            }
        } else {
            constexpr int TPR = BN / 8;                // 16-B chunks (8 bf16) per row: 32 / 16
            constexpr int RPP = NT / TPR;              // 16 / 32
            constexpr int PASSES = RC / RPP;           // 8
            const int c8 = (t % TPR) * 8, rr = t / TPR;
            const int n = n0 + c8;
            const bool ncol = n < N;
            bf16x8 hx[PASSES];
            int crow[PASSES];                          // GATHER 1: storage rows of this thread's output rows, loaded together
            if constexpr (GATHER == 1) {
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
                    const int m = mh + rr + RPP * ps;
                    crow[ps] = e.rowmap[m < M ? m : M - 1];
                }
            }
            float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            constexpr bool CS = (EPI == WJ_EPI_BF16 || EPI == WJ_EPI_MUL_GELU_GRAD);
            if constexpr (EPI == WJ_EPI_MUL_GELU_GRAD || EPI == WJ_EPI_MUL_GELU_GRAD_Z) {
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
                    const int m = mh + rr + RPP * ps;
#pragma unroll
                    for (int x = 0; x < 8; ++x) hx[ps][x] = f2bf(0.f);
                    long arow = m;
                    if constexpr (GATHER == 1) arow = crow[ps];          // aux lives at C's own (gathered) rows
                    if (ncol && m < M) hx[ps] = *reinterpret_cast<const bf16x8*>((const bf16_t*)e.aux + arow * e.ldc + n);
                }
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int r = rr + RPP * ps, m = mh + r;
                if (!(ncol && m < M)) continue;
                if (pair && ((r_lo + r) >> 7) != role) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + r * CP + c8 * 2);
                long orow = m;
                if constexpr (GATHER == 1) orow = crow[ps];
                const long off = orow * e.ldc + n;
                if constexpr (EPI == WJ_EPI_BF16) {
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = v;
                    if (e.colsum) {
#pragma unroll
                        for (int x = 0; x < 8; ++x) csum[x] += bf2f(v[x]);
                    }
                } else if constexpr (EPI == WJ_EPI_BIAS_GELU2) {
                    // h = v (bf16).  Stored: C = gelu'(h) (all the backward needs of h: one multiply there, no erf/exp),
                    // C2 = gelu(h); both from the same erf and exp.
                    // WJ_EPI_BIAS_GELU (forward only, e.C2 == NULL): C = gelu(h) and nothing else.
                    bf16x8 gl, gp;
                    if (e.q_out) {                                // kernel-uniform: MX fp8 of gelu(h) for the next GEMM (config 5)
                        if (e.C2) gelu_bf16x8<true>(v, gl, gp);
                        else gelu_bf16x8<false>(v, gl, gp);
                        float f[8], amax = 0.f;
#pragma unroll
                        for (int x = 0; x < 8; ++x) { f[x] = bf2f(gl[x]); amax = fmaxf(amax, fabsf(f[x])); }
                        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));          // 32 columns = 4 lanes of this row
                        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                        int sc = 0;
                        if (amax > 0.f) {
                            int ex;
                            const float mm = frexpf(amax * (1.0f / 448.0f), &ex);
                            sc = (mm == 0.5f) ? ex - 1 : ex;
                            sc = max(-127, min(127, sc));
                        }
                        const float inv = __builtin_amdgcn_ldexpf(1.0f, -sc);
                        unsigned lo = 0, hi = 0;
                        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
                        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
                        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
                        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
                        *reinterpret_cast<uint2*>(e.q_out + off) = make_uint2(lo, hi);
                        const unsigned sbyte = (unsigned)(sc + 127);
                        const int base = lane & ~15;                          // 128 columns = 16 lanes
                        const unsigned s0 = __shfl(sbyte, base, 64), s1 = __shfl(sbyte, base + 4, 64), s2 = __shfl(sbyte, base + 8, 64),
                                       s3 = __shfl(sbyte, base + 12, 64);
                        if ((lane & 15) == 0) e.q_scales[(long)(n >> 7) * e.ld_q + orow] = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
                        if (e.C2) {                               // training form: gelu'(h) and the bf16 gelu(h) as well
                            *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = gp;
                            *reinterpret_cast<bf16x8*>((bf16_t*)e.C2 + off) = gl;
                        } else if (e.C) {
                            *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = gl;
                        }
                    } else if (e.C2) {                            // kernel-uniform
                        gelu_bf16x8<true>(v, gl, gp);
                        *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = gp;
                        *reinterpret_cast<bf16x8*>((bf16_t*)e.C2 + off) = gl;
                    } else {
                        gelu_bf16x8<false>(v, gl, gp);
                        *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = gl;
                    }
                } else if constexpr (EPI == WJ_EPI_MUL_GELU_GRAD) {
                    bf16x8 o;
#pragma unroll
                    for (int x = 0; x < 8; ++x) o[x] = f2bf(bf2f(v[x]) * bf2f(hx[ps][x]));   // aux = gelu'(h) saved by the forward
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = o;
                    if (e.colsum) {
#pragma unroll
                        for (int x = 0; x < 8; ++x) csum[x] += bf2f(o[x]);
                    }
                } else if constexpr (EPI == WJ_EPI_BF16_ADD_POS) {
                    // y = float(bf16(acc + bias)) + pos[m % T][n]: what wj_add_pos makes of the mapper's bf16 output, without the round trip
                    const float* pr = reinterpret_cast<const float*>(e.aux) + (long)(m % e.seg_rows) * N + n;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pr), p1 = *reinterpret_cast<const f32x4*>(pr + 4);
                    f32x4 y0, y1;
                    bf16x8 o;
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        y0[x] = bf2f(v[x]) + p0[x];
                        y1[x] = bf2f(v[4 + x]) + p1[x];
                        o[x] = f2bf(y0[x]);
                        o[4 + x] = f2bf(y1[x]);
                    }
                    if (e.C2) {
                        *reinterpret_cast<f32x4*>((float*)e.C2 + off) = y0;
                        *reinterpret_cast<f32x4*>((float*)e.C2 + off + 4) = y1;
                    }
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = o;
                } else if constexpr (EPI == WJ_EPI_MUL_GELU_GRAD_Z) {
                    bf16x8 o;            // d(pre) = bf16(d(post)) * gelu'(pre): the bits of a bf16 d(post) tensor followed by wj_gelu_bwd_bf16
#pragma unroll
                    for (int x = 0; x < 8; ++x) o[x] = f2bf(bf2f(v[x]) * gelu_grad_f(bf2f(hx[ps][x])));
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = o;
                } else if constexpr (EPI == WJ_EPI_CONV_GELU) {
                    const bool valid = (m % e.seg_rows) < e.seg_valid;
                    bf16x8 pre, post, unused;
                    gelu_bf16x8<false>(v, post, unused);
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        pre[x] = valid ? v[x] : f2bf(0.f);
                        post[x] = valid ? post[x] : f2bf(0.f);
                    }
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C + off) = pre;
                    *reinterpret_cast<bf16x8*>((bf16_t*)e.C2 + off) = post;
                }
            }
            if constexpr (CS) {
                if (e.colsum) {   // kernel-uniform: fold this chunk's column partials (threads t, t+TPR, ... share a column chunk)
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
#pragma unroll
                        for (int o = TPR; o < 64; o <<= 1) csum[x] += __shfl_xor(csum[x], o, 64);
                    }
                    __syncthreads();                                  // all reads of the staged C tile are done
                    float* cs = reinterpret_cast<float*>(smem);       // [8 waves][BN]
                    if (lane < TPR && ncol) {
#pragma unroll
                        for (int x = 0; x < 8; ++x) cs[wave * BN + c8 + x] = csum[x];
                    }
                    __syncthreads();
                    if (t < BN && n0 + t < N) {
                        float tot = 0.f;
#pragma unroll
                        for (int wv = 0; wv < 8; ++wv) tot += cs[wv * BN + t];
                        atomicAdd(e.colsum + n0 + t, tot);
                    }
#pragma unroll
                    for (int x = 0; x < 8; ++x) csum[x] = 0.f;
                }
            }
        }
        if (ch + 1 < NCHUNK) __syncthreads();
    }
}

template <bool AT, bool BT, int EPI, int BN, int SCHED, int GATHER = 0>
__global__ __launch_bounds__(NT, BN == 256 ? 1 : 2) void gemm3_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                                      long lda, long ldb, int M, int N, int K, int tiles_n,
                                                                      int split_k, int k_per_split, EpiArgs e) {
    gemm3_body<AT, BT, EPI, BN, SCHED, GATHER>(A, B, lda, ldb, M, N, K, tiles_n, split_k, k_per_split, e, blockIdx.x, gridDim.x);
}

// ---- grouped split-K weight gradients -----------------------------------------------------------------------------------------
// Several C_p[M_p, N_p] += A_p^T B_p problems (both operands token-major: the wgrads of one transformer layer, or of two) in ONE
// launch.  Why: a lone wgrad has few output tiles (9-36), so filling 256 CUs needs split-K 7-28, and every K slice ends with a
// 256 x 256 fp32 atomic tile -- at the chip-wide float-atomic rate (~1.3 TB/s) that epilogue was 35-50 % of each launch.  Grouped,
// the tiles of all problems fill the chip with split-K 1-8: 3-4x fewer atomic bytes, K loops 3-8x longer.
struct GroupProblem {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    long lda, ldb, ldc;
    int M, N, K, tiles_n, split, kps, wg_begin, nwg;
};
constexpr int GROUP_MAX = 8;
struct GroupTable {
    GroupProblem p[GROUP_MAX];
    int n;
};

template <int BN, int BMT = 256, int SCHED = 0>
__global__ __launch_bounds__(NT, (BN == 256 || BMT == 384) ? 1 : 2) void gemm3_grouped_wgrad_kernel(GroupTable g) {
    // XCD-grouped order over the WHOLE group (blocks b, b + 8, ... share an XCD and get a contiguous run of the group's work list:
    // neighbours inside a problem, i.e. tiles that stream the same K slice).  No padding between problems: a group sized for the
    // chip's resident slots must not spill a few workgroups into a second round.
    const int gid = xcd_remap(blockIdx.x, gridDim.x);
    int q = 0;
#pragma unroll 1
    for (int x = 1; x < g.n; ++x)
        if (gid >= g.p[x].wg_begin) q = x;
    const GroupProblem& P = g.p[q];
    const int bid = gid - P.wg_begin;
    EpiArgs e;
    e.C = P.C; e.C2 = nullptr; e.bias = nullptr; e.aux = nullptr; e.colsum = nullptr; e.ldc = P.ldc; e.seg_rows = 1; e.seg_valid = 1;
    e.alpha = 1.f; e.rowmap = nullptr; e.sa = nullptr; e.sb = nullptr; e.lds_a = 0; e.lds_b = 0; e.q_out = nullptr; e.q_scales = nullptr; e.ld_q = 0;
    e.pair_ws = nullptr; e.pair_flags = nullptr;
    gemm3_body<true, true, WJ_EPI_ATOMIC_F32, BN, SCHED, 0, BMT>(P.A, P.B, P.lda, P.ldb, P.M, P.N, P.K, P.tiles_n, P.split, P.kps, e, bid, P.nwg, true);
}

// K-split pairs (gemm3_body): which problems, and the scratch they need: [tiles][2] flags (padded to 4 KiB), then per tile and role
// the accumulators of four waves (32 registers x 64 lanes x 16 B each).
int pair_min_k() {
    static int v = -1;
    if (v < 0) v = wj_lab_env_int("WJ_PAIR_MIN_K", 1536);     // lab build: WJ_PAIR_MIN_K=1000000 switches the pairs off (A/B runs)
    return v;
}
long pair_ws_need(int tiles) { return 4096L * ((tiles * 8 + 4095) / 4096) + (long)tiles * PAIR_TILE_BYTES; }
bool pair_shape(const wj_gemm_args* a) {
    if (a->a_trans || a->b_trans || a->rowmap || a->colsum || a->split_k > 1 || a->epilogue != WJ_EPI_BF16) return false;
    if (a->N % 256 != 0 || a->K % 256 != 0 || a->K < pair_min_k()) return false;
    const int tiles = ((a->M + BM - 1) / BM) * (a->N / 256);
    return tiles > 32 && tiles <= 128;            // 2 x tiles workgroups fit the chip in one round (<= 32 tiles would want a deeper split)
}

template <bool AT, bool BT, int EPI, int BN, int SCHED, int GATHER = 0>
int launch(const wj_gemm_args* a, hipStream_t s) {
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
    int split = a->split_k < 1 ? 1 : a->split_k;
    int kps = ((a->K + split - 1) / split + 63) / 64 * 64;
    split = (a->K + kps - 1) / kps;
    bool pair = false;
    if constexpr (SCHED == 2 && EPI == WJ_EPI_BF16 && GATHER == 0 && BN == 256) {
        pair = pair_shape(a) && a->workspace && a->workspace_bytes >= pair_ws_need(tiles_m * tiles_n) && !((uintptr_t)a->workspace & 255);
        if (pair) { split = 2; kps = a->K / 2; }
    }
    EpiArgs e;
    e.C = a->C; e.C2 = a->epilogue == WJ_EPI_BIAS_GELU ? nullptr : a->C2; e.bias = (const float*)a->bias; e.aux = a->aux; e.ldc = a->ldc; e.colsum = a->colsum;
    e.seg_rows = a->seg_rows > 0 ? a->seg_rows : 1; e.seg_valid = a->seg_rows > 0 ? a->seg_valid : 1;
    e.alpha = a->alpha;
    e.rowmap = a->rowmap;
    e.sa = nullptr; e.sb = nullptr; e.lds_a = 0; e.lds_b = 0; e.q_out = nullptr; e.q_scales = nullptr; e.ld_q = 0;
    e.pair_ws = nullptr; e.pair_flags = nullptr;
    if (pair) {
        e.pair_flags = (unsigned*)a->workspace;
        e.pair_ws = (float*)((char*)a->workspace + 4096L * ((tiles_m * tiles_n * 8 + 4095) / 4096));
    }
    auto kern = gemm3_kernel<AT, BT, EPI, BN, SCHED, GATHER>;
    constexpr int lds = Cfg<BN>::LDS_BYTES;
    static int attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return WJ_ERR_LAUNCH;
    // pairs: every XCD gets both roles of its run of tiles (ceil(tiles / 8) of each; spare workgroups leave at once)
    const int nwg = pair ? 16 * ((tiles_m * tiles_n + 7) / 8) : tiles_m * tiles_n * split;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, s, (const bf16_t*)a->A, (const bf16_t*)a->B, (long)a->lda,
                       (long)a->ldb, a->M, a->N, a->K, tiles_n, split, kps, e);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

// Variant selection, from tools/gemm_bench.py / tools/gemm_overhead.py on MI355X:
//   0 = 256x128 tile, 2 workgroups/CU    : short K, VALU-heavy epilogues (GELU), dgrad (col-form B), N % 256 != 0
//   1 = 256x256 tile, plain schedule     : split-K wgrad (long K loop; 850-1000 TFLOP/s)
//   2 = 256x256 tile, ping-pong schedule : forward / conv shapes with plain epilogues and K >= 512 (+9..17 % over variant 0)
//   3 = 256x256x64 tile, eight-phase     : row-form operands, K % 128 == 0 (+10..15 % over variant 2 at K = 768, +35 % at 8192^3)
//   4 = variant 3's loop, persistent     : >= 256 work items; forward epilogues, plain BF16 and MUL_GELU_GRAD (+ column sums) for the
//                                          row-form dgrads against W^T shadows (csrc/gemm_persist.hip)
//   5 = row panels, 128 x 384 items      : N = 384, row-form operands, plain BF16 (csrc/gemm_panel.hip)
//   6 = persistent, deferred epilogue    : 128 x 256 items, the GELU of item i under the K loop of item i + 1 (csrc/gemm_pde.hip): the
//                                          GELU epilogues (BIAS_GELU, BIAS_GELU2, CONV_GELU) with N % 256 == 0, K >= 256, >= 512 items
// wj_gemm_args.schedule = 1 + v forces variant v for that call (tests, tools/gemm_check.py; 1-3 need N % 256 == 0 to avoid wasted columns
// but stay correct; a variant that cannot run a shape falls back to 3, then 0); the lab build also honours WJ_GEMM_VARIANT=v for calls
// that leave the field 0.  The library keeps no selection state.
// The row-panel kernel is NOT picked automatically: on the predictor's shapes it measures what the persistent 256 x 256 kernel measures
// (cold 51-53 / 109-113 / 142-152 us at K = 384 / 1152 / 1536 either way; the step 46.04 against 45.96 ms interleaved) -- both are bound by
// the ISSUE of the LDS-DMA instructions, ~65 ns per 1-KB piece and wave (tools/panel_diag.py, DESIGN_EXPERIMENTS.md round 6).  It runs
// where wj_gemm_args.schedule asks for it; lab build: WJ_GEMM_PANEL=1 picks it for every eligible shape (A/B runs).
bool panel_auto() {
    static const int on = wj_lab_env_int("WJ_GEMM_PANEL", 0);
    return on != 0;
}

bool pde_auto() {
    static const int on = wj_lab_env_int("WJ_GEMM_PDE", 0);
    return on != 0;
}

int pick_variant(const wj_gemm_args* a) {
    static const int env_forced = wj_lab_env_int("WJ_GEMM_VARIANT", -1);
    const int forced = a->schedule > 0 ? a->schedule - 1 : env_forced;
    const bool ep_ok = !a->a_trans && !a->b_trans && a->K % 128 == 0 && a->split_k <= 1;   // eight-phase schedule (variant 3)
    if (forced == 6) return wj_gemm_pde_eligible(a) ? 6 : (wj_gemm_persist_eligible(a) ? 4 : (ep_ok ? 3 : 0));
    if (forced < 0 && pde_auto() && wj_gemm_pde_eligible(a)) return 6;
    if (forced == 5) return wj_gemm_panel_eligible(a) ? 5 : (ep_ok ? 3 : 0);
    if (forced == 4) return wj_gemm_persist_eligible(a) ? 4 : (ep_ok ? 3 : 0);
    if (forced == 3) return ep_ok ? 3 : 0;
    if (forced >= 0 && forced <= 2) return forced;
    // row panels (csrc/gemm_panel.hip): thin outputs (N = 384) with enough rows to fill the chip -- the predictor's out_proj / linear2 and
    // its dgrads into d = 384; full-row work items, A staged four K tiles ahead by waves of its own
    if (forced < 0 && panel_auto() && wj_gemm_panel_eligible(a) && a->M >= 128 * 256) return 5;
    // persistent eight-phase (csrc/gemm_persist.hip): the same K loop without the per-tile prologue / LDS-staged epilogue / dispatch gap
    // (its edge tiles are shifted inwards: a last tile column narrower than half a tile is mostly duplicate work)
    if (ep_ok && (a->N % 256 == 0 || a->N % 256 >= 128) && wj_gemm_persist_eligible(a)) return 4;
    // eight-phase (64-deep tiles, full-line fetches): every row-form shape whose N fills 256-wide tiles, and N = 384 with a long
    // K loop (the half-empty second tile still beats the 128-wide variant there); measured with tools/gemm_check.py
    if (ep_ok && (a->N % 256 == 0 || (a->N > 256 && a->K >= 1536))) return 3;
    if (a->N % 256 != 0) return 0;
    if (a->epilogue == WJ_EPI_ATOMIC_F32) return 1;
    if (!a->a_trans && !a->b_trans && a->K >= 512 && (a->epilogue == WJ_EPI_BF16 || a->epilogue == WJ_EPI_CONV_GELU)) return 2;
    return 0;
}

template <bool AT, bool BT, int EPI>
int launch_bn(const wj_gemm_args* a, hipStream_t s) {
    int v = pick_variant(a);
    if (v == 6) {
        const int rc = wj_gemm_pde_launch(a, s);
        if (rc != WJ_ERR_UNSUPPORTED) return rc;
        v = wj_gemm_persist_eligible(a) ? 4 : 3;
    }
    if (v == 5) {
        if constexpr (!AT && !BT && EPI == WJ_EPI_BF16) {
            const int rc = wj_gemm_panel_launch(a, s);
            if (rc != WJ_ERR_UNSUPPORTED) return rc;
        }
        v = wj_gemm_persist_eligible(a) ? 4 : 3;
    }
    if (v == 4) {
        const int rc = wj_gemm_persist_launch(a, s);
        if (rc != WJ_ERR_UNSUPPORTED) return rc;
        v = 3;                                   // no scheduling slot for this stream: the one-tile-per-workgroup form
    }
    switch (v) {
        case 1: return launch<AT, BT, EPI, 256, 0>(a, s);
        case 2: return launch<AT, BT, EPI, 256, 1>(a, s);
        case 3:
            if constexpr (!AT && !BT) return launch<AT, BT, EPI, 256, 2>(a, s);
            else return launch<AT, BT, EPI, 128, 0>(a, s);
        default: return launch<AT, BT, EPI, 128, 0>(a, s);
    }
}

template <bool AT, bool BT>
int dispatch_epi(const wj_gemm_args* a, hipStream_t s) {
    switch (a->epilogue) {
        case WJ_EPI_BF16: return launch_bn<AT, BT, WJ_EPI_BF16>(a, s);
        case WJ_EPI_BIAS_GELU2:
        case WJ_EPI_BIAS_GELU: return launch_bn<AT, BT, WJ_EPI_BIAS_GELU2>(a, s);   // one instantiation, C2 == NULL selects the single output
        case WJ_EPI_MUL_GELU_GRAD: return launch_bn<AT, BT, WJ_EPI_MUL_GELU_GRAD>(a, s);
        case WJ_EPI_ADD_F32: return launch_bn<AT, BT, WJ_EPI_ADD_F32>(a, s);
        case WJ_EPI_ATOMIC_F32: return launch_bn<AT, BT, WJ_EPI_ATOMIC_F32>(a, s);
        case WJ_EPI_CONV_GELU: return launch_bn<AT, BT, WJ_EPI_CONV_GELU>(a, s);
        case WJ_EPI_BF16_ADD_POS:
            if constexpr (!AT && !BT) return launch_bn<AT, BT, WJ_EPI_BF16_ADD_POS>(a, s);
            else return WJ_ERR_UNSUPPORTED;
        default: return WJ_ERR_ARG;
    }
}

}  // namespace

template <int EPI>
int launch_fp8(const wj_gemm_fp8_args* a, hipStream_t s) {
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + 255) / 256;
    EpiArgs e;
    e.C = a->C; e.C2 = a->epilogue == WJ_EPI_BIAS_GELU ? nullptr : a->C2; e.bias = a->bias; e.aux = nullptr; e.ldc = a->ldc; e.colsum = nullptr;
    e.seg_rows = 1; e.seg_valid = 1; e.alpha = 1.f; e.rowmap = nullptr;
    e.sa = (const uint32_t*)a->scale_a; e.sb = (const uint32_t*)a->scale_b; e.lds_a = a->ld_scale_a; e.lds_b = a->ld_scale_b;
    e.q_out = (unsigned char*)a->q_out; e.q_scales = (uint32_t*)a->q_scales; e.ld_q = a->ld_q_scale;
    e.pair_ws = nullptr; e.pair_flags = nullptr;
    auto kern = gemm3_kernel<false, false, EPI, 256, 3, 0>;
    constexpr int lds = Cfg<256>::LDS_BYTES;     // 135168 >= ring (128 KiB) + two parities of block scales (4 KiB)
    static_assert(lds >= 131072 + 4096, "LDS budget of the MX fp8 loop");
    static int attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return WJ_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(NT), lds, s, (const bf16_t*)a->A, (const bf16_t*)a->B, (long)a->lda, (long)a->ldb,
                       a->M, a->N, a->K, tiles_n, 1, a->K, e);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_gemm_mxfp8(const wj_gemm_fp8_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->A || !a->B || !a->scale_a || !a->scale_b) return WJ_ERR_ARG;
    if (!a->C && !(a->epilogue == WJ_EPI_BIAS_GELU && a->q_out)) return WJ_ERR_ARG;
    if (a->q_out && (!a->q_scales || (a->N % 128) || a->ld_q_scale < a->M || a->epilogue == WJ_EPI_BF16)) return WJ_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % 256) || (a->N & 7) || (a->ldc & 7) || (a->lda & 15) || (a->ldb & 15)) return WJ_ERR_ARG;
    if (a->ld_scale_a < a->M || a->ld_scale_b < a->N) return WJ_ERR_ARG;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->scale_a | (uintptr_t)a->scale_b | (uintptr_t)a->q_out) & 15) return WJ_ERR_ARG;
    if (a->epilogue == WJ_EPI_BIAS_GELU2 && !a->C2) return WJ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (a->epilogue) {
        case WJ_EPI_BF16: return launch_fp8<WJ_EPI_BF16>(a, s);
        case WJ_EPI_BIAS_GELU2:
        case WJ_EPI_BIAS_GELU: return launch_fp8<WJ_EPI_BIAS_GELU2>(a, s);
        default: return WJ_ERR_UNSUPPORTED;
    }
}

template <int BN, int BMT = 256, int SCHED = 0>
int launch_grouped(const wj_wgrad_group_args* a, hipStream_t s) {
    constexpr int BM = BMT;
    GroupTable g;
    g.n = a->n;
    long tiles_total = 0;
    for (int x = 0; x < a->n; ++x)
        tiles_total += (long)((a->M[x] + BM - 1) / BM) * ((a->N[x] + BN - 1) / BN);
    const int slots = (BN == 256 || BMT == 384) ? 256 : 512;         // co-resident workgroups of this tile variant
    int begin = 0;
    for (int x = 0; x < a->n; ++x) {
        GroupProblem& P = g.p[x];
        P.A = (const bf16_t*)a->A[x]; P.B = (const bf16_t*)a->B[x]; P.C = (float*)a->C[x];
        P.lda = a->lda[x]; P.ldb = a->ldb[x]; P.ldc = a->ldc[x]; P.M = a->M[x]; P.N = a->N[x]; P.K = a->K[x];
        const int tiles_m = (P.M + BM - 1) / BM;
        P.tiles_n = (P.N + BN - 1) / BN;
        const int tiles = tiles_m * P.tiles_n;
        // one split factor for the whole group: the group's tiles x split ~ the chip (rounded to nearest), K slices >= 1024 deep
        int split = (int)((slots + tiles_total / 2) / tiles_total);
        if (split < 1) split = 1;
        const int maxs = (P.K + 1023) / 1024;
        if (split > maxs) split = maxs;
        P.kps = ((P.K + split - 1) / split + 63) / 64 * 64;
        P.split = (P.K + P.kps - 1) / P.kps;
        P.nwg = tiles * P.split;
        P.wg_begin = begin;
        begin += P.nwg;
    }
    auto kern = gemm3_grouped_wgrad_kernel<BN, BMT, SCHED>;
    constexpr int lds = Cfg<BN, BMT>::LDS_BYTES;
    static int attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return WJ_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(begin), dim3(NT), lds, s, g);
    WJ_CHECK_LAUNCH();
    return WJ_OK;
}

extern "C" int wj_wgrad_grouped(const wj_wgrad_group_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || a->n < 1 || a->n > GROUP_MAX) return WJ_ERR_ARG;
    bool wide = true, m384 = true;
    for (int x = 0; x < a->n; ++x) {
        if (!a->A[x] || !a->B[x] || !a->C[x] || a->M[x] <= 0 || a->N[x] <= 0 || a->K[x] <= 0) return WJ_ERR_ARG;
        if ((a->M[x] & 7) || (a->N[x] & 7) || (a->lda[x] & 7) || (a->ldb[x] & 7) || (a->ldc[x] & 3)) return WJ_ERR_ARG;
        if (((uintptr_t)a->A[x] | (uintptr_t)a->B[x] | (uintptr_t)a->C[x]) & 15) return WJ_ERR_ARG;
        wide = wide && a->N[x] % 256 == 0;
        m384 = m384 && a->M[x] % 384 == 0 && a->N[x] % 128 == 0;
    }
    if (wide) return launch_grouped<256>(a, (hipStream_t)stream);
    // every problem a multiple of 384 rows x 128 columns (the predictor: d = 384): the 384 x 128 tile has no half-empty row tiles
    // (WJ_WGRAD_384=0: the 256 x 128 tile, round 4; =2: the ping-pong schedule on the 384 x 128 tile)
    static int m384_mode = -1;
    if (m384_mode < 0) m384_mode = wj_lab_env_int("WJ_WGRAD_384", 1);
    if (m384 && m384_mode == 1) return launch_grouped<128, 384, 0>(a, (hipStream_t)stream);
    if (m384 && m384_mode == 2) return launch_grouped<128, 384, 1>(a, (hipStream_t)stream);
    return launch_grouped<128>(a, (hipStream_t)stream);
}

// scratch bytes wj_gemm_bf16 can use for this problem (0: none) -- wj_workspace_bytes("wj_gemm_bf16", args)
int64_t wj_gemm_ws_bytes(const wj_gemm_args* a) {
    if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || !pair_shape(a)) return 0;
    return pair_ws_need(((a->M + BM - 1) / BM) * (a->N / 256));
}

extern "C" int wj_gemm_bf16(const wj_gemm_args* a, void* stream) {
    WJ_CLEAR_STALE_ERROR();
    if (!a || !a->A || !a->B || !a->C) return WJ_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return WJ_ERR_ARG;
    if ((a->N & 7) || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 7)) return WJ_ERR_ARG;
    if ((!a->a_trans || !a->b_trans) && (a->K & 7)) return WJ_ERR_ARG;  // row-form operands are read in 8-element K chunks
    if (a->a_trans && (a->M & 7)) return WJ_ERR_ARG;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return WJ_ERR_ARG;
    if ((a->epilogue == WJ_EPI_BIAS_GELU2 || a->epilogue == WJ_EPI_CONV_GELU) && !a->C2) return WJ_ERR_ARG;
    if (a->epilogue == WJ_EPI_MUL_GELU_GRAD && !a->aux) return WJ_ERR_ARG;
    if (a->epilogue == WJ_EPI_BF16_ADD_POS && (!a->aux || a->seg_rows <= 0 || ((uintptr_t)a->aux & 15) || ((uintptr_t)a->C2 & 15) || a->colsum)) return WJ_ERR_ARG;
    if (a->epilogue == WJ_EPI_MUL_GELU_GRAD_Z && (!a->aux || !a->rowmap || ((uintptr_t)a->aux & 15))) return WJ_ERR_ARG;
    if (a->split_k > 1 && a->epilogue != WJ_EPI_ATOMIC_F32) return WJ_ERR_ARG;
    if (a->colsum && a->epilogue != WJ_EPI_BF16 && a->epilogue != WJ_EPI_MUL_GELU_GRAD) return WJ_ERR_ARG;
    if (a->schedule < 0 || a->schedule > 7 || a->persist_cus < 0 || a->persist_cus > 32) return WJ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (a->rowmap) {
        // gather forms (sparse conv backward), one instantiation each
        if (!a->a_trans && a->b_trans && a->epilogue == WJ_EPI_BF16 && !a->colsum && !a->bias)
            return launch<false, true, WJ_EPI_BF16, 128, 0, 1>(a, s);
        if (!a->a_trans && a->b_trans && a->epilogue == WJ_EPI_MUL_GELU_GRAD_Z && !a->colsum && !a->bias && a->aux)
            return launch<false, true, WJ_EPI_MUL_GELU_GRAD_Z, 128, 0, 1>(a, s);
        if (a->a_trans && a->b_trans && a->epilogue == WJ_EPI_ATOMIC_F32) return launch<true, true, WJ_EPI_ATOMIC_F32, 256, 0, 2>(a, s);
        return WJ_ERR_UNSUPPORTED;
    }
    if (!a->a_trans && !a->b_trans) return dispatch_epi<false, false>(a, s);
    if (!a->a_trans && a->b_trans) return dispatch_epi<false, true>(a, s);
    if (a->a_trans && a->b_trans) return dispatch_epi<true, true>(a, s);
    return dispatch_epi<true, false>(a, s);
}
