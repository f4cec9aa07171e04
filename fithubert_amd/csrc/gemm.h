// Generic MFMA GEMM for gfx950:   C[i][j] = epilogue( alpha * sum_r X(i, r) * Y(j, r) )
//
// Every matrix product of the student (conv-as-GEMM over channels-last windows, the
// transformer/FFN/head linears, their data-gradients and weight-gradients) is one
// instance of this kernel.  An operand is either
//   R_CONTIG : element (idx, r) at  base(idx) + r           (reduction index contiguous)
//   R_STRIDED: element (idx, r) at  base(r)   + idx         (operand index contiguous)
// where base(.) maps a flat row index to (batch, t) so that overlapping conv windows
// (row stride s*Cin, window k*Cin) are addressed in place -- no im2col buffer.
#pragma once
#include <stdint.h>
#include "../../include/fithubert_hip.h"
#include "common.cuh"

typedef fh_drop DropCfgC;
typedef fh_gemm_operand GemmOperand;
typedef fh_gemm_epilogue GemmEpilogue;
typedef fh_gemm_desc GemmParams;

// dtype: 0 = float, 1 = bf16
bool fh_gemm_tuning();
int fh_gemm_launch(const GemmParams& p, int dtype, hipStream_t stream);

static inline GemmOperand fh_operand(const void* p, int layout, long ld) {
  GemmOperand o;
  o.p = p; o.layout = layout; o.ld = ld; o.rpb = 0; o.bstride = 0; o.gstride = 0; o.nseg = 1;
  o.tshift[0] = o.tshift[1] = 0; o.tvalid = 0; o.segoff[0] = o.segoff[1] = 0; o.seg1_limit = 0;
  return o;
}
static inline GemmEpilogue fh_epilogue() {
  GemmEpilogue e;
  e.alpha = 1.0f; e.bias = nullptr; e.bias_gstride = 0; e.bias_mod = 0; e.preact = nullptr; e.act = FH_ACT_NONE;
  e.drop.p = 0.f; e.drop.seed = 0; e.drop.site = 0; e.gelu_aux = nullptr; e.residual = nullptr; e.row_len = nullptr;
  e.out_f32 = 0; e.atomic = 0; e.idiv = 0; e.si1 = 0; e.si0 = 0; e.jdiv = 0; e.sj1 = 0; e.sj0 = 0; e.tail_cols = 0; e.aux_factor = 0;
  e.kd_teacher = nullptr; e.kd_tdtype = 1; e.kd_lstride = 0; e.kd_layer = nullptr; e.kd_w = nullptr; e.kd_Tt = 0; e.kd_P = 0; e.kd_type = 0;
  e.kd_grad_scale = 1.f; e.kd_count = 1.f; e.kd_partials = nullptr;
  e.c0_wave = e.c0_w = e.c0_gamma = e.c0_beta = e.c0_mean = e.c0_rstd = nullptr; e.c0_wpart = nullptr; e.c0_S = e.c0_K = e.c0_stride = 0;
  return e;
}
static inline GemmParams fh_gemm_params() {
  GemmParams p;
  p.X = fh_operand(nullptr, FH_R_CONTIG, 0);
  p.Y = fh_operand(nullptr, FH_R_CONTIG, 0);
  p.C = nullptr; p.M = p.N = p.Kseg = 0; p.ldc = 0; p.c_rpb = 0; p.c_bstride = 0; p.c_gstride = 0;
  p.groups = 1; p.splitk = 1; p.ep = fh_epilogue();
  return p;
}
