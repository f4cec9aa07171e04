// extern "C" entry points (include/fithubert_hip.h)
#include "gemm.h"
#include "kernels.h"

#define ST(s) ((hipStream_t)(s))

extern "C" {

int fh_gemm(const fh_gemm_desc* d, int dtype, fh_stream_t stream) {
  if (!d) return -1;
  return fh_gemm_launch(*d, dtype, ST(stream));
}

long fh_conv0_scratch_floats(int B, int S, int C0, int K, int stride) {
  const int T0 = (S - K) / stride + 1;
  return fh_conv0_scratch_n(B, T0, C0);
}
int fh_conv0_forward(int dtype, const float* wave, const float* w, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, float* scratch, int B, int S, int C0, int K, int stride, float eps,
                     fh_stream_t stream) {
  return fh_conv0_fwd_launch(dtype, wave, w, gamma, beta, y, mean, rstd, (float2*)scratch, B, S, C0, K, stride, eps,
                             ST(stream));
}
int fh_conv0_backward(int dtype, const float* wave, const float* w, const float* gamma, const float* beta,
                      const void* dy, const float* mean, const float* rstd, float* scratch, float* dw, float* dgamma,
                      float* dbeta, int B, int S, int C0, int K, int stride, float grad_scale, fh_stream_t stream) {
  const int T0 = (S - K) / stride + 1;
  float2* part = (float2*)scratch;
  float2* ab = part + (long)B * fh_conv0_nchunk(T0) * C0;
  return fh_conv0_bwd_launch(dtype, wave, w, gamma, beta, dy, mean, rstd, part, ab, dw, dgamma, dbeta, B, S, C0, K,
                             stride, grad_scale, ST(stream));
}

int fh_layernorm_forward(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                         float* rstd, int M, int C, float eps, fh_drop drop, fh_stream_t stream) {
  return fh_layernorm_fwd_launch(dtype, x, gamma, beta, y, mean, rstd, M, C, eps, drop, ST(stream));
}
int fh_add_layernorm_forward(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                             int M, int C, float eps, fh_stream_t stream) {
  return fh_add_layernorm_fwd_launch(dtype, x, res, gamma, beta, y, M, C, eps, ST(stream));
}
int fh_layernorm_gelu_forward(int dtype, const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps,
                              fh_stream_t stream) {
  return fh_add_layernorm_fwd_launch(dtype, x, nullptr, gamma, beta, y, M, C, eps, ST(stream), 1);
}
int fh_gelu_forward(int dtype, const void* x, void* y, long n, fh_stream_t stream) {
  return fh_unary_launch(dtype, x, y, n, 0, ST(stream));
}
int fh_layernorm_backward(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                          const float* rstd, const float* gamma, void* dx, void* dx2, float* dgamma, float* dbeta,
                          int M, int C, fh_drop out_drop, fh_drop in_drop, fh_stream_t stream) {
  return fh_layernorm_bwd_launch(dtype, dy, dy2, x, mean, rstd, gamma, dx, dx2, dgamma, dbeta, M, C, out_drop, in_drop,
                                 ST(stream));
}
int fh_colsum(int dtype, const void* X, float* out, int M, int N, long ldx, int groups, long x_gstride,
              long out_gstride, int mod, float scale, fh_stream_t stream) {
  return fh_colsum_launch(dtype, X, out, M, N, ldx, groups, x_gstride, out_gstride, mod, scale, ST(stream));
}

int fh_posconv_prepare(int dtype, const float* g, const float* v, void* w_fwd, void* w_bwd, float* norm, int D,
                       int groups, int K, fh_stream_t stream) {
  return fh_posconv_prep_launch(dtype, g, v, w_fwd, w_bwd, norm, D, groups, K, ST(stream));
}
int fh_posconv_forward(int dtype, const void* x, const void* w_fwd, const float* bias, void* zc, void* y, int B, int T,
                       int D, int groups, int K, fh_stream_t stream) {
  return fh_posconv_fwd_launch(dtype, x, w_fwd, bias, zc, y, B, T, D, groups, K, ST(stream));
}
int fh_posconv_backward_data(int dtype, const void* dsum, const void* zc, const void* w_bwd, void* dzc, void* dx, int B,
                             int T, int D, int groups, int K, fh_stream_t stream) {
  return fh_posconv_bwd_data_launch(dtype, dsum, zc, w_bwd, dzc, dx, B, T, D, groups, K, ST(stream));
}
int fh_posconv_backward_weight(int dtype, const void* dzc, const void* x, float* dws, float* dbias, int B, int T, int D,
                               int groups, int K, fh_stream_t stream) {
  return fh_posconv_bwd_weight_launch(dtype, dzc, x, dws, dbias, B, T, D, groups, K, ST(stream));
}
int fh_posconv_weightnorm_backward(const float* g, const float* v, const float* norm, const float* dws, float* dg,
                                   float* dv, int D, int groups, int K, fh_stream_t stream) {
  return fh_posconv_wn_bwd_launch(g, v, norm, dws, dg, dv, D, groups, K, ST(stream));
}

int fh_attention_forward(int dtype, const void* qkv, void* out, float* lse, const int* key_len, int B, int T, int H,
                         int dh, float scale, fh_drop drop, fh_stream_t stream) {
  return fh_attention_fwd_launch(dtype, qkv, out, lse, key_len, B, T, H, dh, scale, drop, ST(stream));
}
int fh_attention_backward(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                          void* dqkv, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop,
                          fh_stream_t stream) {
  return fh_attention_bwd_launch(dtype, qkv, out, dout, lse, delta, dqkv, key_len, B, T, H, dh, scale, drop,
                                 ST(stream));
}
long fh_attention_backward_scratch_floats(int dtype, int B, int T, int H, int dh) { return fh_attention_bwd_scratch_floats(dtype, B, T, H, dh); }
int fh_attention_backward_ws(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                             void* dqkv, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop, float* scratch,
                             fh_stream_t stream) {
  return fh_attention_bwd_launch(dtype, qkv, out, dout, lse, delta, dqkv, key_len, B, T, H, dh, scale, drop, ST(stream), scratch);
}
int fh_attention_backward_bias(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                               void* dqkv, float* dbias, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop,
                               float* scratch, fh_stream_t stream) {
  return fh_attention_bwd_launch(dtype, qkv, out, dout, lse, delta, dqkv, key_len, B, T, H, dh, scale, drop, ST(stream), scratch, dbias);
}

int fh_kd_loss(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
               const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
               long teacher_lstride, int loss_type, float grad_scale, fh_stream_t stream) {
  return fh_kd_loss_launch(dtype, teacher_dtype, pred, teacher, dpred, layer_w, teacher_layer, partials, out_losses, L,
                           B, Tp, Tt, P, teacher_lstride, loss_type, grad_scale, ST(stream));
}

int fh_kd_loss_rows(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
                    const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
                    long teacher_lstride, int loss_type, float rec_weight, float sim_weight, fh_stream_t stream) {
  return fh_kd_loss_rows_launch(dtype, teacher_dtype, pred, teacher, dpred, layer_w, teacher_layer, partials, out_losses, L,
                                B, Tp, Tt, P, teacher_lstride, loss_type, rec_weight, sim_weight, ST(stream));
}
int fh_kd_loss_rows_scaled(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
                           const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
                           long teacher_lstride, int loss_type, float rec_weight, float sim_weight, float grad_scale, fh_stream_t stream) {
  return fh_kd_loss_rows_launch(dtype, teacher_dtype, pred, teacher, dpred, layer_w, teacher_layer, partials, out_losses, L,
                                B, Tp, Tt, P, teacher_lstride, loss_type, rec_weight, sim_weight, ST(stream), grad_scale);
}

int fh_cast(int dtype, const float* src, void* dst, long n, fh_stream_t stream) {
  return fh_cast_launch(dtype, src, dst, n, ST(stream));
}
int fh_permute3_cast(int dtype, const float* src, void* dst, int d0, int d1, int d2, int p0, int p1, int p2,
                     fh_stream_t stream) {
  return fh_permute3_cast_launch(dtype, src, dst, d0, d1, d2, p0, p1, p2, ST(stream));
}
int fh_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
             float wd, float bc1, float bc2, float grad_scale, const float* decay_mask, fh_stream_t stream) {
  return fh_adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale, decay_mask, ST(stream));
}

int fh_adamw_amp(float* p, const float* g, float* m, float* v, long n, float lr_peak, int warmup_steps, int total_steps, float beta1,
                 float beta2, float eps, float wd, float grad_scale, fh_amp_state* state, fh_stream_t stream) {
  return fh_adamw_amp_launch(p, g, m, v, n, lr_peak, warmup_steps, total_steps, beta1, beta2, eps, wd, grad_scale, state, ST(stream));
}

int fh_conv1d_forward(int dtype, const void* x, const void* wf, const float* bias, void* y, int B, int T, int Cin, int Cout,
                      int k, int stride, int gelu, fh_stream_t stream) {
  if (T < k) return -561;
  const int Tout = (T - k) / stride + 1;
  GemmParams g = fh_gemm_params();
  g.X = fh_operand(x, FH_R_CONTIG, (long)stride * Cin);
  g.X.rpb = Tout;
  g.X.bstride = (long)T * Cin;
  g.Y = fh_operand(wf, FH_R_CONTIG, (long)k * Cin);
  g.C = y;
  g.M = B * Tout; g.N = Cout; g.Kseg = k * Cin; g.ldc = Cout;
  g.ep.bias = bias;
  g.ep.act = gelu ? FH_ACT_GELU : FH_ACT_NONE;
  return fh_gemm_launch(g, dtype, ST(stream));
}

// y = [gelu](x W^T + bias) [+ residual];  preact (optional) receives x W^T + bias
int fh_linear_forward(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* preact, void* y, int M,
                      int K, int N, int gelu, fh_stream_t stream) {
  GemmParams g = fh_gemm_params();
  g.X = fh_operand(x, FH_R_CONTIG, K);
  g.Y = fh_operand(w, FH_R_CONTIG, K);
  g.C = y;
  g.M = M; g.N = N; g.Kseg = K; g.ldc = N;
  g.ep.bias = bias;
  g.ep.act = gelu ? FH_ACT_GELU : FH_ACT_NONE;
  g.ep.residual = residual;
  g.ep.preact = preact;
  return fh_gemm_launch(g, dtype, ST(stream));
}

int fh_conv0_layernorm_gelu_forward(int dtype, const float* wave, const float* w, const float* bias, const float* gamma,
                                    const float* beta, void* y, int B, int S, int C0, int K, int stride, float eps, fh_stream_t stream) {
  return fh_conv0_ln_gelu_launch(dtype, wave, w, bias, gamma, beta, y, B, S, C0, K, stride, eps, ST(stream));
}

}  // extern "C"
