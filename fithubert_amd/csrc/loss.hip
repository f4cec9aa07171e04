// Layer-wise knowledge-distillation loss fused with its gradient.
//
//   reference: W2V2Distil.calculate_loss, train.py:250-267 (stack student projections / teacher hiddens),
//   :282 (narrow the teacher to the student's length), :284-293 (L1 / MSE, random-layer weights, per-layer
//   means, sum), :372-378 (total).  Closed form:  loss = sum_l w_l * mean_{b,t,d} f(pred_l - tgt_l).
//
// HBM-bound: one pass reads pred and teacher once (16-byte vectors), writes dL/dpred, and produces
// per-block fp32 partial sums; a one-block finisher adds the partials in a fixed order in fp64, so the
// loss is bitwise reproducible (no float atomics) and accurate to the 1e-4 parity target.
#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int LOSS_BLOCK = 256;
constexpr int CHUNKS_PER_THREAD = 8;

template <typename TT, int N> __device__ __forceinline__ void load_teacher(const TT* p, float* f);
template <> __device__ __forceinline__ void load_teacher<float, 4>(const float* p, float* f) { ld4(p, f); }
template <> __device__ __forceinline__ void load_teacher<float, 8>(const float* p, float* f) {
  ld4(p, f);
  ld4(p + 4, f + 4);
}
template <> __device__ __forceinline__ void load_teacher<bf16, 4>(const bf16* p, float* f) { ld4(p, f); }
template <> __device__ __forceinline__ void load_teacher<bf16, 8>(const bf16* p, float* f) {
  unpack16<bf16>(ld16(p), f);
}

template <typename T, typename TT>
__global__ void __launch_bounds__(LOSS_BLOCK) kd_loss_kernel(const T* __restrict__ pred, const TT* __restrict__ teacher,
                                                             T* __restrict__ dpred, const float* __restrict__ layer_w,
                                                             const int* __restrict__ teacher_layer,
                                                             float* __restrict__ partials, int B, int Tp, int Tt, int P,
                                                             long teacher_lstride, int loss_type, float grad_scale,
                                                             int nblk) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[LOSS_BLOCK / 64];
  const int l = blockIdx.y;
  const float w = layer_w[l];
  const int cpr = P / VEC;                       // chunks per row
  const long nchunks = (long)B * Tp * cpr;       // per layer
  const long per_layer = (long)B * Tp * P;
  const T* pl = pred + (long)l * per_layer;
  T* dl = dpred ? dpred + (long)l * per_layer : nullptr;
  const TT* tl = teacher + (long)teacher_layer[l] * teacher_lstride;
  const float gcoef = grad_scale * w / (float)per_layer * (loss_type == 0 ? 2.0f : 1.0f);
  float acc = 0.f;
  const long c0 = (long)blockIdx.x * (LOSS_BLOCK * CHUNKS_PER_THREAD);
#pragma unroll
  for (int u = 0; u < CHUNKS_PER_THREAD; ++u) {
    const long c = c0 + (long)u * LOSS_BLOCK + threadIdx.x;
    if (c >= nchunks) continue;
    const long row = c / cpr;
    const int pc = (int)(c - row * cpr);
    float pv[VEC], g[VEC];
    if (w != 0.f) {
      const long b = row / Tp, t = row - b * Tp;
      float tv[VEC];
      unpack16<T>(ld16(pl + c * VEC), pv);
      load_teacher<TT, VEC>(tl + (b * Tt + t) * P + (long)pc * VEC, tv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d = pv[e] - tv[e];
        if (loss_type == 0) {
          acc = fmaf(d, d, acc);
          g[e] = gcoef * d;
        } else {
          acc += fabsf(d);
          g[e] = d > 0.f ? gcoef : (d < 0.f ? -gcoef : 0.f);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] = 0.f;
    }
    if (dl) st16(dl + c * VEC, pack16<T>(g));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < LOSS_BLOCK / 64; ++i) s += red[i];
    partials[(long)l * nblk + blockIdx.x] = s;
  }
}

// out[l] = mean_l (unweighted), out[L] = sum_l w_l * mean_l
__global__ void kd_loss_final_kernel(const float* partials, const float* layer_w, double* out, int L, int nblk,
                                     double inv_count) {
  __shared__ double sh[256];
  double total = 0.0;
  for (int l = 0; l < L; ++l) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partials[(long)l * nblk + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double m = sh[0] * inv_count;
      out[l] = m;
      total += (double)layer_w[l] * m;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[L] = total;
}

}  // namespace

extern "C" int fh_kd_loss_nblk(int dtype, int B, int Tp, int P) {
  const int vec = dtype == 0 ? 4 : 8;
  const long nchunks = (long)B * Tp * (P / vec);
  return (int)((nchunks + LOSS_BLOCK * CHUNKS_PER_THREAD - 1) / (LOSS_BLOCK * CHUNKS_PER_THREAD));
}

int fh_kd_loss_launch(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred,
                      const float* layer_w, const int* teacher_layer, float* partials, double* out_losses, int L, int B,
                      int Tp, int Tt, int P, long teacher_lstride, int loss_type, float grad_scale, hipStream_t st) {
  const int vec = dtype == 0 ? 4 : 8;
  if (P % 8 || Tp > Tt) return -401;
  const int nblk = fh_kd_loss_nblk(dtype, B, Tp, P);
  dim3 grid(nblk, L), block(LOSS_BLOCK);
  (void)vec;
#define FH_LAUNCH_LOSS(T, TT)                                                                                       \
  hipLaunchKernelGGL((kd_loss_kernel<T, TT>), grid, block, 0, st, (const T*)pred, (const TT*)teacher, (T*)dpred,     \
                     layer_w, teacher_layer, partials, B, Tp, Tt, P, teacher_lstride, loss_type, grad_scale, nblk)
  if (dtype == 0 && teacher_dtype == 0) FH_LAUNCH_LOSS(float, float);
  else if (dtype == 0) FH_LAUNCH_LOSS(float, bf16);
  else if (teacher_dtype == 0) FH_LAUNCH_LOSS(bf16, float);
  else FH_LAUNCH_LOSS(bf16, bf16);
#undef FH_LAUNCH_LOSS
  FH_CHECK_LAUNCH();
  hipLaunchKernelGGL(kd_loss_final_kernel, dim3(1), dim3(256), 0, st, partials, layer_w, out_losses, L, nblk,
                     1.0 / ((double)B * Tp * P));
  FH_CHECK_LAUNCH();
  return 0;
}
