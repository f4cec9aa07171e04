// LayerNorm forward / backward over the channel dimension and column sums (bias gradients).
//
//   reference: fairseq LayerNorm == nn.LayerNorm(eps=1e-5) at modules/model.py:321,447 (512-wide, after
//   the conv frontend), modules/module.py:251,281 (encoder.layer_norm), :513,518,569,580 (per layer).
//
// HBM-bound: one wave per row, 16-byte vector loads, the row lives in registers, two-pass statistics
// (mean, then centred sum of squares) reduced with wavefront shuffles -- no LDS, no atomics in forward.
// Backward keeps dgamma/dbeta partials in registers over a strip of rows and issues one fp32 atomic per
// column per wave.
#include <cstdlib>

// Every 16-byte load of this unit is NON-TEMPORAL (round 6): LayerNorm, the column sums and the row fills stream tensors of 61 - 200 MB
// once per launch; same box, alternating, B = 256: student step 50.30 -> 49.99 ms, teacher forward 42.34 -> 41.91 (25 add + LayerNorm passes).
#define FH_LD_NT 1
#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int LN_RPW = 1;  // rows per wave, loaded together (4x the bytes in flight of a row-at-a-time loop)

template <typename T, int NCH>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     int C, float eps, DropCfg drop, const T* __restrict__ res, int gelu) {
  // res != nullptr: normalises T(x + res) -- the post-LN residual add of the frozen teacher's layers, rounded to the
  // storage type first so that the result equals the unfused add-then-LayerNorm
  constexpr int VEC = ElemTraits<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
  if (row0 >= M) return;
  const int nchunk = C / VEC;
  float v[LN_RPW][NCH][VEC];
#pragma unroll
  for (int rr = 0; rr < LN_RPW; ++rr) {
    const int row = min(row0 + rr, M - 1);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunk) unpack16<T>(ld16(x + (long)row * C + ch * VEC), v[rr][u]);
    }
  }
  if (res) {
#pragma unroll
    for (int rr = 0; rr < LN_RPW; ++rr) {
      const int row = min(row0 + rr, M - 1);
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int ch = lane + 64 * u;
        if (ch < nchunk) {
          float a[VEC];
          unpack16<T>(ld16(res + (long)row * C + ch * VEC), a);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[rr][u][e] = (float)(T)(v[rr][u][e] + a[e]);
        }
      }
    }
  }
  float gm[NCH][VEC], bt[NCH][VEC];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int ch = lane + 64 * u;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        ld4(gamma + ch * VEC + e, gm[u] + e);
        ld4(beta + ch * VEC + e, bt[u] + e);
      }
    }
  }
  float mean[LN_RPW], rstd[LN_RPW];
#pragma unroll
  for (int rr = 0; rr < LN_RPW; ++rr) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NCH; ++u)
      if (lane + 64 * u < nchunk) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[rr][u][e];
      }
    mean[rr] = wave_sum(s) / (float)C;
  }
#pragma unroll
  for (int rr = 0; rr < LN_RPW; ++rr) {
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < NCH; ++u)
      if (lane + 64 * u < nchunk) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = v[rr][u][e] - mean[rr];
          ss = fmaf(d, d, ss);
        }
      }
    rstd[rr] = rsqrtf(wave_sum(ss) / (float)C + eps);
  }
#pragma unroll
  for (int rr = 0; rr < LN_RPW; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean[rr];
      if (rstd_out) rstd_out[row] = rstd[rr];
    }
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunk) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = fmaf((v[rr][u][e] - mean[rr]) * rstd[rr], gm[u][e], bt[u][e]);
        if (gelu) {  // conv -> LayerNorm -> GELU blocks of a layer_norm-mode feature extractor (HuBERT Large)
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = gelu_t<T>(o[e]);
        }
        if (drop.p > 0.f) drop_scale_vec<VEC>(drop, (unsigned long long)row * C + ch * VEC, o);
        st16(y + (long)row * C + ch * VEC, pack16<T>(o));
      }
    }
  }
}

#ifndef LN_BWD_UNROLL
#define LN_BWD_UNROLL 2
#endif
// dy_total = (dy [+ dy2]) [* dropmask(out_drop)]  is the gradient w.r.t. the LayerNorm output.
// dx -> gradient w.r.t. the LayerNorm input; dx2 (optional) = dx * dropmask(in_drop).
template <typename T, int NCH>
__global__ void __launch_bounds__(NCH == 1 ? 1024 : (NCH == 2 ? 512 : 256)) ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                     const T* __restrict__ x, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, T* __restrict__ dx2, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int C, int rows_per_wave,
                                                     DropCfg out_drop, DropCfg in_drop, float* __restrict__ dxsum) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int nwav = blockDim.x >> 6;
  const int wave_global = blockIdx.x * nwav + (threadIdx.x >> 6);
  const int nchunk = C / VEC;
  // dxsum (optional): column sums of the LAST gradient this kernel writes (dx2 when given, else dx) -- the bias gradient of the Linear
  // whose output the LayerNorm consumed (out-proj / fc2: d bias = sum over rows of the dropout-scaled LayerNorm input gradient), which
  // used to be a separate colsum pass over that tensor (round 5)
  float dg[NCH][VEC], db[NCH][VEC], gm[NCH][VEC], dxs[NCH][VEC];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int ch = lane + 64 * u;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      dg[u][e] = 0.f;
      db[u][e] = 0.f;
      dxs[u][e] = 0.f;
      gm[u][e] = ch < nchunk ? gamma[ch * VEC + e] : 0.f;
    }
  }
  const int r0 = wave_global * rows_per_wave;
  const int r1 = min(M, r0 + rows_per_wave);
#pragma unroll LN_BWD_UNROLL
  for (int row = r0; row < r1; ++row) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float g[NCH][VEC], xh[NCH][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunk) {
        float d[VEC], xv[VEC];
        unpack16<T>(ld16(dy + (long)row * C + ch * VEC), d);
        if (dy2) {
          float d2[VEC];
          unpack16<T>(ld16(dy2 + (long)row * C + ch * VEC), d2);
#pragma unroll
          for (int e = 0; e < VEC; ++e) d[e] += d2[e];
        }
        unpack16<T>(ld16(x + (long)row * C + ch * VEC), xv);
        if (out_drop.p > 0.f) drop_scale_vec<VEC>(out_drop, (unsigned long long)row * C + ch * VEC, d);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[u][e] = (xv[e] - mean) * rstd;
          dg[u][e] = fmaf(d[e], xh[u][e], dg[u][e]);
          db[u][e] += d[e];
          g[u][e] = d[e] * gm[u][e];
          s1 += g[u][e];
          s2 = fmaf(g[u][e], xh[u][e], s2);
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)C;
    const float c2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunk) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rstd * (g[u][e] - c1 - xh[u][e] * c2);
        st16(dx + (long)row * C + ch * VEC, pack16<T>(o));
        if (dx2) {
          if (in_drop.p > 0.f) drop_scale_vec<VEC>(in_drop, (unsigned long long)row * C + ch * VEC, o);
          st16(dx2 + (long)row * C + ch * VEC, pack16<T>(o));
        }
        if (dxsum) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) dxs[u][e] += o[e];
        }
      }
    }
  }
  // dgamma / dbeta: reduce the waves of the block through LDS, then ONE atomic per column per block
  extern __shared__ float red[];  // [waves][2][C]
  float* my = red + (threadIdx.x >> 6) * 2 * C;
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int ch = lane + 64 * u;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        my[ch * VEC + e] = dg[u][e];
        my[C + ch * VEC + e] = db[u][e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float v = 0.f;
    for (int w = 0; w < nwav; ++w) v += red[w * 2 * C + i];
    atomicAdd((i < C ? dgamma : dbeta - C) + i, v);
  }
  if (dxsum) {   // a second round through the same LDS rows (a third [C] block per wave would halve the workgroups a CU holds)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = lane + 64 * u;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) my[ch * VEC + e] = dxs[u][e];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
      float v = 0.f;
      for (int w = 0; w < nwav; ++w) v += red[w * 2 * C + i];
      atomicAdd(dxsum + i, v);
    }
  }
}

// out[n % mod] += sum_m X[m][n]   (fp32 atomics).  Block = 64 column chunks x 4 row lanes over a 512-row strip;
// the 4 row lanes are reduced through LDS so every block issues one atomic per column.
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ X, float* __restrict__ out, int M, int N,
                                                     long ldx, long x_gstride, long out_gstride, int mod, float scale) {
  constexpr int VEC = ElemTraits<T>::VEC;
  __shared__ float red[4][64 * VEC];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cx;
  const bool cv = ch * VEC < N;
  const T* xp = X + (long)blockIdx.z * x_gstride + (long)ch * VEC;
  const int r0 = blockIdx.y * 512, r1 = min(M, r0 + 512);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (cv) {
    int r = r0 + ry;
    for (; r + 28 < r1; r += 32) {  // 8 independent 16-byte loads in flight (the loop is latency-, not bandwidth-bound)
      V16 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ld16(xp + (long)(r + 4 * u) * ldx);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float v[VEC];
        unpack16<T>(t[u], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[e];
      }
    }
    for (; r < r1; r += 4) {
      float v[VEC];
      unpack16<T>(ld16(xp + (long)r * ldx), v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[ry][cx * VEC + e] = acc[e];
  __syncthreads();
  float* o = out + (long)blockIdx.z * out_gstride;
  for (int i = threadIdx.x; i < 64 * VEC; i += 256) {
    const int n = blockIdx.x * 64 * VEC + i;
    if (n < N) atomicAdd(o + (mod > 0 ? n % mod : n), (red[0][i] + red[1][i] + red[2][i] + red[3][i]) * scale);
  }
}

// zero rows [from, T) of every batch item of a [B][T][C] tensor
template <typename T>
__global__ void zero_rows_kernel(T* x, int B, int Tn, int C, int from) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const long per_b = (long)(Tn - from) * C / VEC;
  const long total = per_b * B;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / per_b, r = i - b * per_b;
    st16(x + ((long)b * Tn + from) * C + r * VEC, v16_zero());
  }
}

template <typename T>
int ln_fwd_t(const T* x, const float* g, const float* b, T* y, float* mean, float* rstd, int M, int C, float eps,
             DropCfg drop, hipStream_t st, const T* res = nullptr, int gelu = 0) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return -301;
  const int nch = (C / VEC + 63) / 64;
  dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW)), block(256);
  if (nch == 1)
    hipLaunchKernelGGL((ln_fwd_kernel<T, 1>), grid, block, 0, st, x, g, b, y, mean, rstd, M, C, eps, drop, res, gelu);
  else if (nch == 2)
    hipLaunchKernelGGL((ln_fwd_kernel<T, 2>), grid, block, 0, st, x, g, b, y, mean, rstd, M, C, eps, drop, res, gelu);
  else if (nch <= 4)
    hipLaunchKernelGGL((ln_fwd_kernel<T, 4>), grid, block, 0, st, x, g, b, y, mean, rstd, M, C, eps, drop, res, gelu);
  else
    return -302;
  FH_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int ln_bwd_t(const T* dy, const T* dy2, const T* x, const float* mean, const float* rstd, const float* g, T* dx, T* dx2,
             float* dgamma, float* dbeta, int M, int C, DropCfg out_drop, DropCfg in_drop, hipStream_t st, float* dxsum = nullptr) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return -301;
  const int nch = (C / VEC + 63) / 64;
  // 16 waves x 8 rows per workgroup: the per-column dgamma / dbeta sums leave a workgroup as ONE atomic per column, so rows per
  // workgroup set the atomic count (128 rows: 249 x 960 atomics for the student's 31872 x 480; 64 rows as 4 waves x 16 was 20 %
  // slower), while 8 rows per wave keep ~4000 waves in flight for the load -> shuffle-reduce -> store chain
  // few rows (the reference's own batch: 747 per layer): one dependent 8-row walk per wave was the whole 25 us of the launch;
  // fewer rows per wave shorten the chain (the extra per-column atomics are nothing at that size)
  const int rpw = M >= 16384 ? 8 : (M >= 4096 ? 4 : (M >= 1024 ? 2 : 1));
  const int wpb = nch == 1 ? 16 : (nch == 2 ? 8 : 4);  // = the kernel's launch bound; [waves][2][C] floats of LDS stay <= 64 KiB
  const int waves = (M + rpw - 1) / rpw;
  dim3 grid((waves + wpb - 1) / wpb), block(64 * wpb);
  const size_t smem = (size_t)2 * wpb * C * sizeof(float);
  if (nch == 1)
    hipLaunchKernelGGL((ln_bwd_kernel<T, 1>), grid, block, smem, st, dy, dy2, x, mean, rstd, g, dx, dx2, dgamma, dbeta, M, C,
                       rpw, out_drop, in_drop, dxsum);
  else if (nch == 2)
    hipLaunchKernelGGL((ln_bwd_kernel<T, 2>), grid, block, smem, st, dy, dy2, x, mean, rstd, g, dx, dx2, dgamma, dbeta, M, C,
                       rpw, out_drop, in_drop, dxsum);
  else if (nch <= 4)
    hipLaunchKernelGGL((ln_bwd_kernel<T, 4>), grid, block, smem, st, dy, dy2, x, mean, rstd, g, dx, dx2, dgamma, dbeta, M, C,
                       rpw, out_drop, in_drop, dxsum);
  else
    return -302;
  FH_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int fh_layernorm_fwd_launch(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                            float* rstd, int M, int C, float eps, fh_drop drop, hipStream_t st) {
  if (M <= 0) return 0;
  return dtype == 0 ? ln_fwd_t<float>((const float*)x, gamma, beta, (float*)y, mean, rstd, M, C, eps, drop, st)
                    : dtype == 2 ? ln_fwd_t<f16>((const f16*)x, gamma, beta, (f16*)y, mean, rstd, M, C, eps, drop, st)
                    : ln_fwd_t<bf16>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, M, C, eps, drop, st);
}

int fh_add_layernorm_fwd_launch(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                                int M, int C, float eps, hipStream_t st, int gelu) {
  if (M <= 0) return 0;
  fh_drop nodrop{0.f, 0u, 0u};
  return dtype == 0 ? ln_fwd_t<float>((const float*)x, gamma, beta, (float*)y, nullptr, nullptr, M, C, eps, nodrop, st, (const float*)res, gelu)
                    : dtype == 2 ? ln_fwd_t<f16>((const f16*)x, gamma, beta, (f16*)y, nullptr, nullptr, M, C, eps, nodrop, st, (const f16*)res, gelu)
                    : ln_fwd_t<bf16>((const bf16*)x, gamma, beta, (bf16*)y, nullptr, nullptr, M, C, eps, nodrop, st, (const bf16*)res, gelu);
}

int fh_layernorm_bwd_launch(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                            const float* rstd, const float* gamma, void* dx, void* dx2, float* dgamma, float* dbeta,
                            int M, int C, fh_drop out_drop, fh_drop in_drop, hipStream_t st, float* dxsum) {
  if (M <= 0) return 0;
  return dtype == 0 ? ln_bwd_t<float>((const float*)dy, (const float*)dy2, (const float*)x, mean, rstd, gamma,
                                      (float*)dx, (float*)dx2, dgamma, dbeta, M, C, out_drop, in_drop, st, dxsum)
                    : dtype == 2 ? ln_bwd_t<f16>((const f16*)dy, (const f16*)dy2, (const f16*)x, mean, rstd, gamma, (f16*)dx,
                                     (f16*)dx2, dgamma, dbeta, M, C, out_drop, in_drop, st, dxsum)
                    : ln_bwd_t<bf16>((const bf16*)dy, (const bf16*)dy2, (const bf16*)x, mean, rstd, gamma, (bf16*)dx,
                                     (bf16*)dx2, dgamma, dbeta, M, C, out_drop, in_drop, st, dxsum);
}

int fh_colsum_launch(int dtype, const void* X, float* out, int M, int N, long ldx, int groups, long x_gstride,
                     long out_gstride, int mod, float scale, hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  const int vec = dtype == 0 ? 4 : 8;
  if (N % vec || ldx % vec || x_gstride % vec) return -303;
  dim3 grid((N / vec + 63) / 64, (M + 511) / 512, groups), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, st, (const float*)X, out, M, N, ldx, x_gstride, out_gstride,
                       mod, scale);
  else if (dtype == 2)
    hipLaunchKernelGGL(colsum_kernel<f16>, grid, block, 0, st, (const f16*)X, out, M, N, ldx, x_gstride, out_gstride,
                       mod, scale);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, st, (const bf16*)X, out, M, N, ldx, x_gstride, out_gstride,
                       mod, scale);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_zero_rows_launch(int dtype, void* x, int B, int Tn, int C, int from, hipStream_t st) {
  if (from >= Tn || B <= 0) return 0;
  const int vec = dtype == 0 ? 4 : 8;
  if (C % vec) return -304;
  const long total = (long)B * (Tn - from) * C / vec;
  dim3 grid((unsigned)min((long)2048, (total + 255) / 256)), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(zero_rows_kernel<float>, grid, block, 0, st, (float*)x, B, Tn, C, from);
  else if (dtype == 2)
    hipLaunchKernelGGL(zero_rows_kernel<f16>, grid, block, 0, st, (f16*)x, B, Tn, C, from);
  else
    hipLaunchKernelGGL(zero_rows_kernel<bf16>, grid, block, 0, st, (bf16*)x, B, Tn, C, from);
  FH_CHECK_LAUNCH();
  return 0;
}
