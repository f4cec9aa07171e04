// Element-wise kernels: dtype casts / weight re-layouts (per-step weight preparation) and fused AdamW.
//
//   reference: optimizer built at train.py:407-421 (s3prl AdamW_with_schedule, yaml optimizer section);
//   AMP weight casting is implicit in torch.cuda.amp (train.py:469,496).
#include "common.cuh"
#include "kernels.h"

namespace {

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float f[4];
    ld4(src + i4, f);
    st4(dst + i4, f);
  } else {
    for (long i = i4; i < n; ++i) dst[i] = from_f<T>(src[i]);
  }
}

// dst[a][b][c] (dims d[p0], d[p1], d[p2]) = src[i0][i1][i2] with (a, b, c) = (i[p0], i[p1], i[p2])
template <typename T>
__global__ void permute3_kernel(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1, int d2, int p0, int p1,
                                int p2, long src_gstride, long dst_gstride) {
  const int d[3] = {d0, d1, d2};
  const long n = (long)d0 * d1 * d2;
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  src += (long)blockIdx.y * src_gstride;   // blockIdx.y = group (one launch for the 12 projection heads)
  dst += (long)blockIdx.y * dst_gstride;
  const int e1 = d[p1], e2 = d[p2];
  int idx[3];
  idx[p2] = (int)(o % e2);
  idx[p1] = (int)((o / e2) % e1);
  idx[p0] = (int)(o / ((long)e1 * e2));
  dst[o] = from_f<T>(src[((long)idx[0] * d1 + idx[1]) * d2 + idx[2]]);
}

// ---- one launch for every "copy the fp32 master weights into a kernel layout" job of fh_student_prepare (round 5): the 16-bit cast
// of the flat buffer, the conv / TR / head weight permutes and the transposed layer weights used to be ~22 separate 5 - 25 us launches at
// the head of the student's stream, each of which could wait for a CU behind the teacher's persistent GEMM launches (rocprofv3 showed
// averages of 0.8 - 4.9 ms for kernels that run in 20 - 170 us alone).  A job = a grouped 3-d permute with cast:
//   dst[g][a][b][c] (dims d[p0], d[p1], d[p2]) = src[g][i0][i1][i2],  (a, b, c) = (i[p0], i[p1], i[p2])
// Blocks of 256 outputs are dealt over the jobs through the prefix table `first_block`.
template <typename T>
__global__ void __launch_bounds__(256) weight_layout_kernel(const float* __restrict__ params, unsigned char* __restrict__ wc, fh_layout_jobs jobs) {
  int j = 0;
#pragma unroll 1
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
  const fh_layout_job& q = jobs.job[j];
  const long per = (long)q.d[0] * q.d[1] * q.d[2];
  const long o_all = (long)((int)blockIdx.x - jobs.first_block[j]) * 256 + threadIdx.x;
  if (o_all >= per * q.groups) return;
  const int g = (int)(o_all / per);
  const long o = o_all - (long)g * per;
  const float* src = params + q.src + (long)g * q.src_gstride;
  T* dst = reinterpret_cast<T*>(wc + q.dst) + (long)g * q.dst_gstride;
  const int e1 = q.d[q.p[1]], e2 = q.d[q.p[2]];
  int idx[3];
  idx[q.p[2]] = (int)(o % e2);
  idx[q.p[1]] = (int)((o / e2) % e1);
  idx[q.p[0]] = (int)(o / ((long)e1 * e2));
  dst[o] = from_f<T>(src[((long)idx[0] * q.d[1] + idx[1]) * q.d[2] + idx[2]]);
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2, float gscale, const float* __restrict__ decay_mask) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int cnt = (int)min((long)4, n - i4);
  float pv[4], gv[4], mv[4], vv[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
  if (cnt == 4) {
    ld4(p + i4, pv); ld4(g + i4, gv); ld4(m + i4, mv); ld4(v + i4, vv);
    if (decay_mask) ld4(decay_mask + i4, dm);
  } else {
    for (int e = 0; e < cnt; ++e) {
      pv[e] = p[i4 + e]; gv[e] = g[i4 + e]; mv[e] = m[i4 + e]; vv[e] = v[i4 + e];
      if (decay_mask) dm[e] = decay_mask[i4 + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gr = gv[e] * gscale;
    mv[e] = b1 * mv[e] + (1.f - b1) * gr;
    vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
    const float mh = mv[e] / bc1, vh = vv[e] / bc2;
    pv[e] -= lr * (mh / (sqrtf(vh) + eps) + wd * dm[e] * pv[e]);
  }
  if (cnt == 4) {
    st4(p + i4, pv); st4(m + i4, mv); st4(v + i4, vv);
  } else {
    for (int e = 0; e < cnt; ++e) {
      p[i4 + e] = pv[e]; m[i4 + e] = mv[e]; v[i4 + e] = vv[e];
    }
  }
}

// ---- fp16 mode (the reference's AMP, train.py:469,496: GradScaler): the gradients carry a loss scale, a step whose gradient holds a
// non-finite value is skipped.  Detection, the skip and the step bookkeeping are device side (fh_amp_state), so no host sync:
//   amp_check   : found_inf |= any non-finite element of the flat gradient
//   amp_prepare : lr (linear warm-up / linear decay, distill.py lr_at) and the Adam bias corrections of step good_steps + 1
//   adamw_dev   : the AdamW update with those scalars read from the state; a no-op when found_inf
//   amp_update  : good_steps++ or skipped++, found_inf = 0
__global__ void amp_check_kernel(const float* __restrict__ g, long n, fh_amp_state* __restrict__ st) {
  bool bad = false;
  for (long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < n; i4 += (long)gridDim.x * blockDim.x * 4) {
    if (i4 + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i4);
      // x - x is 0 for finite x and NaN for inf / NaN
      const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
      bad |= !(t == 0.f);
    } else {
      for (long i = i4; i < n; ++i) bad |= !((g[i] - g[i]) == 0.f);
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&st->found_inf, 1);
}
__global__ void amp_prepare_kernel(fh_amp_state* st, float lr_peak, int warm, int total, float b1, float b2) {
  const int t = st->good_steps + 1, step = t - 1;
  double lr = lr_peak;
  if (total > 0) {
    const int w = warm > 1 ? warm : 1;
    if (step < w) lr = (double)lr_peak * (step + 1) / w;
    else {
      const int den = total - w > 1 ? total - w : 1;
      const double f = (double)(total - step) / den;
      lr = (double)lr_peak * (f > 0.0 ? f : 0.0);
    }
  }
  st->lr = (float)lr;
  st->bc1 = (float)(1.0 - pow((double)b1, (double)t));
  st->bc2 = (float)(1.0 - pow((double)b2, (double)t));
}
__global__ void amp_update_kernel(fh_amp_state* st) {
  if (st->found_inf) st->skipped += 1; else st->good_steps += 1;
  st->found_inf = 0;
}
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                 float b1, float b2, float eps, float wd, float gscale, const fh_amp_state* __restrict__ st) {
  if (st->found_inf) return;
  const float lr = st->lr, bc1 = st->bc1, bc2 = st->bc2;
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int cnt = (int)min((long)4, n - i4);
  float pv[4], gv[4], mv[4], vv[4];
  if (cnt == 4) {
    ld4(p + i4, pv); ld4(g + i4, gv); ld4(m + i4, mv); ld4(v + i4, vv);
  } else {
    for (int e = 0; e < cnt; ++e) { pv[e] = p[i4 + e]; gv[e] = g[i4 + e]; mv[e] = m[i4 + e]; vv[e] = v[i4 + e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gr = gv[e] * gscale;
    mv[e] = b1 * mv[e] + (1.f - b1) * gr;
    vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
    const float mh = mv[e] / bc1, vh = vv[e] / bc2;
    pv[e] -= lr * (mh / (sqrtf(vh) + eps) + wd * pv[e]);
  }
  if (cnt == 4) {
    st4(p + i4, pv); st4(m + i4, mv); st4(v + i4, vv);
  } else {
    for (int e = 0; e < cnt; ++e) { p[i4 + e] = pv[e]; m[i4 + e] = mv[e]; v[i4 + e] = vv[e]; }
  }
}

// dst = src * gelu'(aux) * scale   (factor != 0: aux already holds gelu' -- the forward saved it, fh_gemm_epilogue.aux_factor)
template <typename T>
__global__ void mul_gelu_grad_kernel(const T* __restrict__ src, const T* __restrict__ aux, T* __restrict__ dst, long nvec,
                                     float scale, int factor) {
  constexpr int VEC = ElemTraits<T>::VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float a[VEC], z[VEC];
    unpack16<T>(ld16(src + i * VEC), a);
    unpack16<T>(ld16(aux + i * VEC), z);
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] *= (factor ? z[e] : gelu_grad_t<T>(z[e])) * scale;
    st16(dst + i * VEC, pack16<T>(a));
  }
}

// mode 0: dst = gelu(src);  mode 1: dst += src.  Four 16-byte vectors per thread are in flight per trip (the loads are issued
// before the first use), which is what a pure streaming kernel needs to approach the HBM rate.
template <typename T>
__global__ void unary_kernel(const T* src, T* dst, long nvec, int mode) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int U = 4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += U * stride) {
    V16 raw[U], rawd[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * stride < nvec) raw[u] = ld16(src + (i0 + u * stride) * VEC);
    if (mode != 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u * stride < nvec) rawd[u] = ld16(dst + (i0 + u * stride) * VEC);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * stride >= nvec) break;
      float a[VEC];
      unpack16<T>(raw[u], a);
      if (mode == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = gelu_t<T>(a[e]);
      } else {
        float d[VEC];
        unpack16<T>(rawd[u], d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += d[e];
      }
      st16(dst + (i0 + u * stride) * VEC, pack16<T>(a));
    }
  }
}

// dst[row][c] = src[row][c] * dropmask(row*C + c) * (t(row) < row_len[b(row)])
template <typename T>
__global__ void mask_drop_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int C, int rpb,
                                 const int* __restrict__ row_len, DropCfg drop) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = C / VEC;
  const long nvec = rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    const int c0 = (int)(i - row * cpr) * VEC;
    float a[VEC];
    unpack16<T>(ld16(src + i * VEC), a);
    bool zero = false;
    if (row_len) {
      const long b = row / rpb;
      zero = (int)(row - b * rpb) >= row_len[b];
    }
    if (zero) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    } else if (drop.p > 0.f) {
      drop_scale_vec<VEC>(drop, (unsigned long long)row * C + c0, a);
    }
    st16(dst + i * VEC, pack16<T>(a));
  }
}

}  // namespace

int fh_unary_launch(int dtype, const void* src, void* dst, long n, int mode, hipStream_t st) {
  const int vec = dtype == 0 ? 4 : 8;
  if (n <= 0) return 0;
  if (n % vec) return -701;
  const long nvec = n / vec;
  dim3 grid((unsigned)min((long)4096, (nvec + 255) / 256)), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(unary_kernel<float>, grid, block, 0, st, (const float*)src, (float*)dst, nvec, mode);
  else if (dtype == 2)
    hipLaunchKernelGGL(unary_kernel<f16>, grid, block, 0, st, (const f16*)src, (f16*)dst, nvec, mode);
  else
    hipLaunchKernelGGL(unary_kernel<bf16>, grid, block, 0, st, (const bf16*)src, (bf16*)dst, nvec, mode);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_mul_gelu_grad_launch(int dtype, const void* src, const void* aux, void* dst, long n, float scale, hipStream_t st,
                            int factor) {
  const int vec = dtype == 0 ? 4 : 8;
  if (n <= 0) return 0;
  if (n % vec) return -701;
  const long nvec = n / vec;
  dim3 grid((unsigned)min((long)4096, (nvec + 255) / 256)), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(mul_gelu_grad_kernel<float>, grid, block, 0, st, (const float*)src, (const float*)aux, (float*)dst,
                       nvec, scale, factor);
  else if (dtype == 2)
    hipLaunchKernelGGL(mul_gelu_grad_kernel<f16>, grid, block, 0, st, (const f16*)src, (const f16*)aux, (f16*)dst, nvec,
                       scale, factor);
  else
    hipLaunchKernelGGL(mul_gelu_grad_kernel<bf16>, grid, block, 0, st, (const bf16*)src, (const bf16*)aux, (bf16*)dst, nvec,
                       scale, factor);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_mask_drop_launch(int dtype, const void* src, void* dst, long rows, int C, int rpb, const int* row_len, fh_drop drop,
                        hipStream_t st) {
  const int vec = dtype == 0 ? 4 : 8;
  if (rows <= 0) return 0;
  if (C % vec || (row_len && rpb <= 0)) return -702;
  const long nvec = rows * (C / vec);
  dim3 grid((unsigned)min((long)4096, (nvec + 255) / 256)), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(mask_drop_kernel<float>, grid, block, 0, st, (const float*)src, (float*)dst, rows, C, rpb, row_len,
                       drop);
  else if (dtype == 2)
    hipLaunchKernelGGL(mask_drop_kernel<f16>, grid, block, 0, st, (const f16*)src, (f16*)dst, rows, C, rpb, row_len, drop);
  else
    hipLaunchKernelGGL(mask_drop_kernel<bf16>, grid, block, 0, st, (const bf16*)src, (bf16*)dst, rows, C, rpb, row_len, drop);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_cast_launch(int dtype, const float* src, void* dst, long n, hipStream_t st) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)((n / 4 + 1 + 255) / 256)), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(cast_kernel<float>, grid, block, 0, st, src, (float*)dst, n);
  else if (dtype == 2)
    hipLaunchKernelGGL(cast_kernel<f16>, grid, block, 0, st, src, (f16*)dst, n);
  else
    hipLaunchKernelGGL(cast_kernel<bf16>, grid, block, 0, st, src, (bf16*)dst, n);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_permute3_cast_launch(int dtype, const float* src, void* dst, int d0, int d1, int d2, int p0, int p1, int p2,
                            hipStream_t st, int groups, long src_gstride, long dst_gstride) {
  const long n = (long)d0 * d1 * d2;
  if (n <= 0 || groups <= 0) return 0;
  dim3 grid((unsigned)((n + 255) / 256), groups), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(permute3_kernel<float>, grid, block, 0, st, src, (float*)dst, d0, d1, d2, p0, p1, p2, src_gstride, dst_gstride);
  else if (dtype == 2)
    hipLaunchKernelGGL(permute3_kernel<f16>, grid, block, 0, st, src, (f16*)dst, d0, d1, d2, p0, p1, p2, src_gstride, dst_gstride);
  else
    hipLaunchKernelGGL(permute3_kernel<bf16>, grid, block, 0, st, src, (bf16*)dst, d0, d1, d2, p0, p1, p2, src_gstride, dst_gstride);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_layout_jobs_launch(int dtype, const float* params, void* wc, fh_layout_jobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return 0;
  if (jobs.n > FH_MAX_LAYOUT_JOBS) return -705;
  int blocks = 0;
  for (int j = 0; j < jobs.n; ++j) {
    jobs.first_block[j] = blocks;
    const long n = (long)jobs.job[j].d[0] * jobs.job[j].d[1] * jobs.job[j].d[2] * jobs.job[j].groups;
    blocks += (int)((n + 255) / 256);
  }
  jobs.first_block[jobs.n] = blocks;
  if (blocks <= 0) return 0;
  if (dtype == 0)
    hipLaunchKernelGGL(weight_layout_kernel<float>, dim3(blocks), dim3(256), 0, st, params, (unsigned char*)wc, jobs);
  else if (dtype == 2)
    hipLaunchKernelGGL(weight_layout_kernel<f16>, dim3(blocks), dim3(256), 0, st, params, (unsigned char*)wc, jobs);
  else
    hipLaunchKernelGGL(weight_layout_kernel<bf16>, dim3(blocks), dim3(256), 0, st, params, (unsigned char*)wc, jobs);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_adamw_launch(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                    float wd, float bc1, float bc2, float grad_scale, const float* decay_mask, hipStream_t st) {
  if (n <= 0) return 0;
  dim3 grid((unsigned)(((n + 3) / 4 + 255) / 256)), block(256);
  hipLaunchKernelGGL(adamw_kernel, grid, block, 0, st, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, grad_scale,
                     decay_mask);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_adamw_amp_launch(float* p, const float* g, float* m, float* v, long n, float lr_peak, int warmup_steps, int total_steps,
                        float beta1, float beta2, float eps, float wd, float grad_scale, fh_amp_state* state, hipStream_t st) {
  if (n <= 0 || !state) return n <= 0 ? 0 : -703;
  hipLaunchKernelGGL(amp_check_kernel, dim3(2048), dim3(256), 0, st, g, n, state);
  FH_CHECK_LAUNCH();
  hipLaunchKernelGGL(amp_prepare_kernel, dim3(1), dim3(1), 0, st, state, lr_peak, warmup_steps, total_steps, beta1, beta2);
  FH_CHECK_LAUNCH();
  dim3 grid((unsigned)(((n + 3) / 4 + 255) / 256)), block(256);
  hipLaunchKernelGGL(adamw_dev_kernel, grid, block, 0, st, p, g, m, v, n, beta1, beta2, eps, wd, grad_scale, state);
  FH_CHECK_LAUNCH();
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, st, state);
  FH_CHECK_LAUNCH();
  return 0;
}

// Stand-in for a collective's reduction kernel (tools/rccl_standin.py): `workgroups` workgroups of 512 threads add src into dst,
// 16 bytes per lane and pass -- RCCL's ring all-reduce runs a few tens of workgroups of this kind next to the step's kernels, and
// what matters for the data-parallel step is whether they find CUs while a persistent GEMM launch holds the chip.
namespace {
__global__ void __launch_bounds__(512) copy_reduce_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(src)[i], b = reinterpret_cast<float4*>(dst)[i];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
    reinterpret_cast<float4*>(dst)[i] = b;
  }
}
}  // namespace
extern "C" int fh_debug_copy_reduce(float* dst, const float* src, long n, int workgroups, fh_stream_t stream) {
  if (!dst || !src || n < 4 || workgroups < 1) return -1;
  hipLaunchKernelGGL(copy_reduce_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, dst, src, n / 4);
  FH_CHECK_LAUNCH();
  return 0;
}
