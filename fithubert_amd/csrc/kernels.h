// Internal launchers of the non-GEMM kernels (all stream-ordered, no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fithubert_hip.h"

// ---- conv0.hip : Conv1d(1, C0, k, s) + GroupNorm(C0, C0) + GELU
int fh_conv0_nchunk(int T0);
int fh_conv0_bwd_blocks(int B, int nchunk);
long fh_conv0_scratch_n(int B, int T0, int C0);  // floats of scratch for fh_conv0_fwd_launch / fh_conv0_bwd_launch (part | ab | backward partials)
int fh_conv0_fwd_launch(int dtype, const float* wave, const float* w, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, float2* part, int B, int S, int C0, int K, int stride, float eps,
                        hipStream_t st);
int fh_conv0_bwd_launch(int dtype, const float* wave, const float* w, const float* gamma, const float* beta,
                        const void* dy, const float* mean, const float* rstd, float2* part, float2* ab, float* dw,
                        float* dgamma, float* dbeta, int B, int S, int C0, int K, int stride, float grad_scale,
                        hipStream_t st, int stats_ready = 0, int ext_nblk = 0);
int fh_conv0_ab_width(int C0);   // float2 per utterance between the statistics partials and the backward's partial sums
long fh_conv0_fused_wpart_floats(int B, int C0);   // floats of the per-utterance sum blocks the fused conv-0 backward accumulates into (zeroed per step)
long fh_conv0_fused_scratch_n(int B, int T0, int C0);  // scratch when layer 1's data-gradient GEMM leaves the backward's partial sums (ext_nblk = fh_conv0_nchunk)

int fh_conv0_ln_gelu_launch(int dtype, const float* wave, const float* w, const float* bias, const float* gamma, const float* beta, void* y,
                            int B, int S, int C0, int K, int stride, float eps, hipStream_t st);

// ---- norm.hip
int fh_layernorm_fwd_launch(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                            float* rstd, int M, int C, float eps, fh_drop drop, hipStream_t st);
int fh_add_layernorm_fwd_launch(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                                int M, int C, float eps, hipStream_t st, int gelu = 0);
int fh_layernorm_bwd_launch(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                            const float* rstd, const float* gamma, void* dx, void* dx2, float* dgamma, float* dbeta,
                            int M, int C, fh_drop out_drop, fh_drop in_drop, hipStream_t st, float* dxsum = nullptr);
// (dxsum: optional [C] fp32, += the column sums of the last gradient written -- dx2 when given, else dx: the bias gradient of the
// Linear in front of the LayerNorm)
int fh_colsum_launch(int dtype, const void* X, float* out, int M, int N, long ldx, int groups, long x_gstride,
                     long out_gstride, int mod, float scale, hipStream_t st);
int fh_zero_rows_launch(int dtype, void* x, int B, int Tn, int C, int from, hipStream_t st);

// ---- posconv.hip : weight-normed grouped positional conv (+ GELU + residual), its backward, weight-norm fwd/bwd
int fh_posconv_cp(int cg);  // padded group width (32 | 64) the positional-conv kernels use, 0 = unsupported
int fh_posconv_prep_launch(int dtype, const float* g, const float* v, void* w_fwd, void* w_bwd, float* norm, int D,
                           int groups, int K, hipStream_t st);
int fh_posconv_fwd_launch(int dtype, const void* x, const void* w_fwd, const float* bias, void* zc, void* y, int B,
                          int Tn, int D, int groups, int K, hipStream_t st);
int fh_posconv_bwd_data_launch(int dtype, const void* dsum, const void* zc, const void* w_bwd, void* dzc, void* dx,
                               int B, int Tn, int D, int groups, int K, hipStream_t st);
int fh_posconv_bwd_weight_launch(int dtype, const void* dzc, const void* x, float* dw, float* dbias, int B, int Tn,
                                 int D, int groups, int K, hipStream_t st);
int fh_posconv_wn_bwd_launch(const float* g, const float* v, const float* norm, const float* dw, float* dg, float* dv,
                             int D, int groups, int K, hipStream_t st);

// ---- attention.hip : multi-head self-attention on a packed [B][T][3*D] q|k|v tensor
int fh_attention_fwd_launch(int dtype, const void* qkv, void* out, float* lse, const int* key_len, int B, int Tn, int H,
                            int dh, float scale, fh_drop drop, hipStream_t st);
// dq_scratch: fh_attention_bwd_scratch_floats(...) floats for the any-length fused backward (16-bit modes, Tn > 256: fp32 dQ partials
// of the key blocks), or null -> the two-pass form
long fh_attention_bwd_scratch_floats(int dtype, int B, int Tn, int H, int dh);
int fh_attention_bwd_launch(int dtype, const void* qkv, const void* out, const void* dout, const float* lse,
                            float* delta, void* dqkv, const int* key_len, int B, int Tn, int H, int dh, float scale,
                            fh_drop drop, hipStream_t st, float* dq_scratch = nullptr, float* dbias = nullptr);
// dbias (optional): [3 H dh] fp32, += the column sums of dqkv = the gradient of the q | k | v projection bias (out of the fused kernels'
// accumulators in the 16-bit modes; a pass over dqkv otherwise)

// ---- loss.hip : layer-wise KD loss (MSE / L1) fused with its gradient
int fh_kd_loss_launch(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred,
                      const float* layer_w, const int* teacher_layer, float* partials, double* out_losses, int L, int B,
                      int Tp, int Tt, int P, long teacher_lstride, int loss_type, float grad_scale, hipStream_t st);
// out[l] = sum(partials[l][0 .. nblk)) / count, out[L] = sum_l w[l] out[l]  (the tail of fh_kd_loss_launch; also behind the KD-loss GEMM epilogue)
int fh_kd_loss_final_launch(const float* partials, const float* layer_w, double* out_losses, int L, int nblk, double count, hipStream_t st);
int fh_kd_loss_rows_launch(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred,
                           const float* layer_w, const int* teacher_layer, float* partials, double* out_losses, int L,
                           int B, int Tp, int Tt, int P, long teacher_lstride, int loss_type, float rec_weight,
                           float sim_weight, hipStream_t st, float grad_scale = 1.0f);  // grad_scale: on dpred only (loss scaling)

// ---- elementwise.hip
// weight-layout jobs of fh_student_prepare, all in ONE launch (weight_layout_kernel): grouped 3-d permute + cast of fp32 master weights
constexpr int FH_MAX_LAYOUT_JOBS = 40;
struct fh_layout_job {
  long src, dst;                 // element offset into the flat fp32 parameters, BYTE offset into the compute-dtype buffer
  long src_gstride, dst_gstride; // elements
  int d[3], p[3], groups;
};
struct fh_layout_jobs {
  int n;
  int first_block[FH_MAX_LAYOUT_JOBS + 1];
  fh_layout_job job[FH_MAX_LAYOUT_JOBS];
  void add(long src, long dst, int d0, int d1, int d2, int p0, int p1, int p2, int groups = 1, long sg = 0, long dg = 0) {
    if (n >= FH_MAX_LAYOUT_JOBS) { n = FH_MAX_LAYOUT_JOBS + 1; return; }   // (reported by fh_layout_jobs_launch)
    fh_layout_job& q = job[n++];
    q.src = src; q.dst = dst; q.src_gstride = sg; q.dst_gstride = dg; q.groups = groups;
    q.d[0] = d0; q.d[1] = d1; q.d[2] = d2; q.p[0] = p0; q.p[1] = p1; q.p[2] = p2;
  }
};
int fh_layout_jobs_launch(int dtype, const float* params, void* wc, fh_layout_jobs& jobs, hipStream_t st);
int fh_cast_launch(int dtype, const float* src, void* dst, long n, hipStream_t st);
int fh_permute3_cast_launch(int dtype, const float* src, void* dst, int d0, int d1, int d2, int p0, int p1, int p2,
                            hipStream_t st, int groups = 1, long src_gstride = 0, long dst_gstride = 0);  // strides in elements
int fh_adamw_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                    float wd, float bc1, float bc2, float grad_scale, const float* decay_mask, hipStream_t st);
int fh_adamw_amp_launch(float* p, const float* g, float* m, float* v, long n, float lr_peak, int warmup_steps, int total_steps,
                        float beta1, float beta2, float eps, float wd, float grad_scale, fh_amp_state* state, hipStream_t st);
int fh_unary_launch(int dtype, const void* src, void* dst, long n, int mode, hipStream_t st);  // 0: dst = gelu(src), 1: dst += src
int fh_mul_gelu_grad_launch(int dtype, const void* src, const void* aux, void* dst, long n, float scale, hipStream_t st,
                            int factor = 0);
int fh_mask_drop_launch(int dtype, const void* src, void* dst, long rows, int C, int rpb, const int* row_len, fh_drop drop,
                        hipStream_t st);
