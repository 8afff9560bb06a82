// Internal seam between csrc/gemm.hip (variant selection, the C ABI entry) and csrc/gemm_persist.hip (the persistent
// eight-phase kernel).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/wavjepa_hip.h"

// true when wj_gemm_persist_launch can run this problem (row-form operands, K % 128 == 0, >= 256 output tiles, a forward epilogue)
bool wj_gemm_persist_eligible(const wj_gemm_args* a);
// WJ_OK, or WJ_ERR_UNSUPPORTED (not eligible / no scheduling slot left for this stream: the caller takes another variant)
int wj_gemm_persist_launch(const wj_gemm_args* a, hipStream_t s);

// the stream's set of 8 tile counters (zero between launches), or NULL when no set is free; *dev_out = the current device
unsigned* wj_gemm_persist_counters(hipStream_t s, int* dev_out);
// resident workgroups per XCD for this call (wj_gemm_args.persist_cus, or the process default)
int wj_gemm_persist_wpx(const wj_gemm_args* a);

// csrc/gemm_pde.hip: the persistent kernel with a DEFERRED epilogue (variant 6: 128 x 256 items, the GELU of item i under the K loop of item
// i + 1); row-form operands, K % 128 == 0, K >= 256, N % 256 == 0, BIAS_GELU / BIAS_GELU2 / CONV_GELU
bool wj_gemm_pde_eligible(const wj_gemm_args* a);
int wj_gemm_pde_launch(const wj_gemm_args* a, hipStream_t s);

// csrc/gemm_panel.hip: the row-panel schedule for thin outputs (variant 5: N = 384, row-form operands, K % 128 == 0, K >= 256, WJ_EPI_BF16)
bool wj_gemm_panel_eligible(const wj_gemm_args* a);
int wj_gemm_panel_launch(const wj_gemm_args* a, hipStream_t s);
