/* Laboratory entry points of libwavjepa_hip_lab.so -- NOT part of the drop-in surface.
 *
 * The release library (libwavjepa_hip.so, include/wavjepa_hip.h) exports compute and query entries only and reads the environment
 * switches INTEGRATION.md lists as safe in production.  The same sources built with -DWJ_LAB (wavjepa_amd.build.build(lab=True) ->
 * wavjepa_amd/lib/libwavjepa_hip_lab.so) additionally
 *   * export the entries below (time stamps of the persistent GEMM; the all-reduce footprint rehearsal),
 *   * honour the result- or schedule-changing diagnostics of csrc/: WJ_PERSIST_DIAG_NOSTORE, WJ_PERSIST_STAMPS, WJ_PERSIST_ACTIVE,
 *     WJ_PERSIST_STAGGER_US / _EPI, WJ_PERSIST_HALF, WJ_PERSIST_WBLOCK, WJ_PERSIST_MIN_TILES, WJ_GEMM_VARIANT, WJ_GEMM_PANEL, WJ_PAIR_MIN_K, WJ_WGRAD_384,
 *     WJ_ATTN_BWD_FRAG, WJ_CONV0_STATS_TCS, WJ_CONV0_APPLY_MFMA, WJ_CONV0_APPLY_OCC, WJ_LN_BWD_ONE_PASS_ROWS
 *     (in the release build each of them is compiled to its default: csrc/common.h wj_lab_env_int).
 * tools/ and the tests that rehearse or dissect a kernel load this library (wavjepa_amd._abi.load_lab(), or WAVJEPA_HIP_LIB=<path>). */
#ifndef WAVJEPA_HIP_LAB_H
#define WAVJEPA_HIP_LAB_H
#include "wavjepa_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic (tools/persist_stamps.py; only filled when the process runs with WJ_PERSIST_STAMPS=1): copies the time stamps the last
 * persistent-GEMM launch wrote -- [2][256 workgroups][64] s_memrealtime values (100 MHz): start, end of prologue, per output tile the
 * end of its first K-tile pair / K loop / epilogue; second half: the eight phases of that first pair -- to `out` (n 64-bit words).
 * Synchronises the device. */
int wj_debug_persist_stamps(unsigned long long* out, int n);

/* MEASUREMENT AID, not a collective: the on-GPU footprint of an all-reduce of `bytes` bytes at `buf`, for rehearsing on ONE GPU what
 * the gradient all-reduce of a data-parallel run (train.py:174-179) takes away from the kernels that run beside it.  `workgroups`
 * resident workgroups of 512 threads (RCCL runs one per channel; the CUs a data-parallel run keeps free of persistent GEMM workgroups)
 * read and rewrite the buffer IN PLACE `passes` times (2 = the reduce-scatter and all-gather legs: 2 x bytes read and written; the
 * values are unchanged), paced so that the launch lasts at least `min_ticks` ticks of the 100 MHz clock (bytes over an assumed bus
 * bandwidth; 0 = as fast as those CUs go).  No data leaves the GPU; nothing about xGMI is measured. */
typedef struct {
    void* buf;
    int64_t bytes;       /* multiple of 16 */
    int64_t min_ticks;
    int32_t workgroups;  /* 1 .. 256 */
    int32_t passes;      /* 1 .. 8 */
} wj_collective_footprint_args;
int wj_collective_footprint(const wj_collective_footprint_args*, void* stream);

#ifdef __cplusplus
}
#endif
#endif
