// First conv layer of the waveform frontend, fused with its GroupNorm and GELU.
//
//   reference: modules/module.py:65-71 (Conv1d(1, C0, k, stride, bias=False) -> Dropout(0) ->
//              Fp32GroupNorm(C0 groups, C0 channels) -> GELU), applied at module.py:99-100.
//
// Cin = 1 makes this layer HBM-bound (reads S floats, writes T0*C0 elements per utterance), so it is
// not a GEMM: each thread owns one output channel (10 taps in registers) and walks time; the waveform
// chunk is staged once in LDS and broadcast.  GroupNorm with one channel per group normalises every
// (utterance, channel) over time, so the un-normalised conv output is never written: pass 1 computes
// chunk-local (mean, M2) with a pivot (no catastrophic cancellation), a tiny kernel merges chunks with
// Chan's formula in fp64, pass 2 recomputes the conv (10 FMA/elem) and writes GELU(GN(.)) once.
// Backward is ONE pass over dy (by linearity, below) + a tiny combine; the waveform needs no grad.
#include <cstdlib>

#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int TCH = 256;   // output time steps per block
constexpr int MAXK = 16;

struct Conv0Args {
  const float* wave;  // [B][S]
  const float* w;     // [C0][K]
  const float* gamma; // [C0]
  const float* beta;  // [C0]
  int B, S, T0, C0, K, stride, nchunk;
};

// stage the waveform chunk of block (b, chunk) in LDS; returns number of valid output steps
__device__ __forceinline__ int stage_wave(const Conv0Args& a, int b, int chunk, float* sw) {
  const int t0 = chunk * TCH;
  const int nt = min(TCH, a.T0 - t0);
  const int nsamp = (nt - 1) * a.stride + a.K;
  const float* src = a.wave + (long)b * a.S + (long)t0 * a.stride;
  for (int i = threadIdx.x; i < nsamp + MAXK; i += blockDim.x) sw[i] = i < nsamp ? src[i] : 0.f;  // zero tail: KT > K reads it
  __syncthreads();
  return nt;
}

__device__ __forceinline__ float conv_at(const float* sw, const float* w, int t, int K, int stride) {
  float v = 0.f;
  const float* x = sw + t * stride;
#pragma unroll
  for (int j = 0; j < MAXK; ++j)
    if (j < K) v = fmaf(w[j], x[j], v);
  return v;
}

// ---------------------------------------------------------------- forward pass 1: statistics WITHOUT evaluating the conv
// The conv output of channel c at frame t is v = w_c . x_t with x_t = wave[stride*t .. stride*t + K).  Its mean and variance
// over time are therefore  mean_c = w_c . m  and  var_c = w_c^T Cov w_c  with m = mean_t x_t (K numbers) and
// Cov = mean_t x_t x_t^T - m m^T (K x K) -- statistics of the WAVEFORM, shared by all C0 channels.  One cheap pass over the
// waveform (K*(K+1)/2 + K partial sums per 256-frame chunk) replaces a full conv pass over T0 x C0 outputs; the chunk
// partials are combined and the quadratic forms evaluated in fp64.
constexpr int NSTAT(int K) { return K + K * K; }  // m_j then R[j][j'] (full matrix, symmetric entries computed twice: simple)

__global__ void __launch_bounds__(128) conv0_stats_kernel(Conv0Args a, float* part /*[B][nchunk][K + K*K]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw = smem;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int nt = stage_wave(a, b, chunk, sw);
  const int ns = a.K + a.K * a.K;
  for (int s = threadIdx.x; s < ns; s += blockDim.x) {
    float acc = 0.f;
    if (s < a.K) {
      for (int t = 0; t < nt; ++t) acc += sw[t * a.stride + s];
    } else {
      const int j = (s - a.K) / a.K, jj = (s - a.K) - j * a.K;
      // pivot-free products are fine here: the combine is fp64 and Cov is formed from fp64 sums
      for (int t = 0; t < nt; ++t) acc = fmaf(sw[t * a.stride + j], sw[t * a.stride + jj], acc);
    }
    part[((long)b * a.nchunk + chunk) * ns + s] = acc;
  }
}

// combine chunk partials (fp64), then mean_c = w_c . m, var_c = w_c^T Cov w_c  ->  mean, rstd per (b, c)
__global__ void conv0_stats_final_kernel(const float* part, const float* w, float* mean, float* rstd, int B, int nchunk, int C0,
                                         int K, int T0, float eps, double* wsum /*[B][K + K*K] raw sums, kept for the backward, or null*/) {
  __shared__ double sm[MAXK + MAXK * MAXK];
  const int b = blockIdx.x;
  const int ns = K + K * K;
  for (int s = threadIdx.x; s < ns; s += blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < nchunk; ++k) acc += (double)part[((long)b * nchunk + k) * ns + s];
    if (wsum) wsum[(long)b * ns + s] = acc;
    sm[s] = acc / (double)T0;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < K * K; s += blockDim.x) {
    const int j = s / K, jj = s - j * K;
    sm[K + s] -= sm[j] * sm[jj];  // covariance (each entry owned by one thread; means are read-only here)
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C0; c += blockDim.x) {
    double mu = 0.0, var = 0.0;
    for (int j = 0; j < K; ++j) {
      const double wj = w[c * K + j];
      mu += wj * sm[j];
      double row = 0.0;
      for (int jj = 0; jj < K; ++jj) row += sm[K + j * K + jj] * (double)w[c * K + jj];
      var += wj * row;
    }
    mean[b * C0 + c] = (float)mu;
    rstd[b * C0 + c] = (float)(1.0 / sqrt(fmax(var, 0.0) + (double)eps));
  }
}

// ---------------------------------------------------------------- vectorised channel mapping (passes 2, bwd 1, bwd 2)
// A thread owns CV = 4 consecutive channels (8/16-byte loads and stores, 4x fewer LDS broadcasts per output) and walks time
// with stride `nlan`; a block's C0/4 channel groups x nlan time lanes = 256 threads.  These passes are latency-bound (one
// dependent load per output), so wider per-thread I/O is what buys time.
constexpr int CV = 4;

struct ChanMap {
  int ngrp, nlan, cg, tl, c0;
  __device__ __forceinline__ ChanMap(int C0, int cv = CV) {
    ngrp = C0 / cv;           // <= 256 and divides 256 (checked on the host)
    nlan = 256 / ngrp;
    cg = threadIdx.x % ngrp;
    tl = threadIdx.x / ngrp;
    c0 = cg * cv;
  }
};

// KT = compile-time tap count (10 for every shipped frontend; MAXK with zero weights past K otherwise): with a run-time
// bound hipcc kept the tap loop rolled, one LDS round trip + branch per tap
template <int KT>
__device__ __forceinline__ void load_w4(const Conv0Args& a, int c0, float (&w)[CV][KT]) {
#pragma unroll
  for (int e = 0; e < CV; ++e)
#pragma unroll
    for (int j = 0; j < KT; ++j) w[e][j] = j < a.K ? a.w[(c0 + e) * a.K + j] : 0.f;
}
template <int KT>
__device__ __forceinline__ void conv4_at(const float* sw, const float (&w)[CV][KT], int t, int stride, float* v) {
  const float* x = sw + t * stride;
  float xs[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) xs[j] = x[j];
#pragma unroll
  for (int e = 0; e < CV; ++e) v[e] = 0.f;
#pragma unroll
  for (int j = 0; j < KT; ++j)
#pragma unroll
    for (int e = 0; e < CV; ++e) v[e] = fmaf(w[e][j], xs[j], v[e]);
}

// ---------------------------------------------------------------- forward pass 2: conv + GN + GELU -> y
// CW channels per thread: 4, or 8 (bf16 16-byte stores, half the LDS broadcasts per output; an experiment, see the launcher).
template <typename T, int KT, int CW>
__global__ void __launch_bounds__(256) conv0_apply_kernel(Conv0Args a, const float* mean, const float* rstd, T* y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw = smem;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int nt = stage_wave(a, b, chunk, sw);
  const ChanMap m(a.C0, CW);
  float w[CW][KT];
#pragma unroll
  for (int e = 0; e < CW; ++e)
#pragma unroll
    for (int j = 0; j < KT; ++j) w[e][j] = j < a.K ? a.w[(m.c0 + e) * a.K + j] : 0.f;
  float sc[CW], sh[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    const int c = m.c0 + e;
    const float rs = rstd[b * a.C0 + c];
    sc[e] = rs * a.gamma[c];
    sh[e] = a.beta[c] - mean[b * a.C0 + c] * sc[e];
  }
  T* yp = y + ((long)b * a.T0 + (long)chunk * TCH) * a.C0 + m.c0;
#pragma unroll 2
  for (int t = m.tl; t < nt; t += m.nlan) {
    const float* x = sw + t * a.stride;
    float xs[KT], v[CW];
#pragma unroll
    for (int j = 0; j < KT; ++j) xs[j] = x[j];
#pragma unroll
    for (int e = 0; e < CW; ++e) v[e] = 0.f;
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int e = 0; e < CW; ++e) v[e] = fmaf(w[e][j], xs[j], v[e]);
#pragma unroll
    for (int e = 0; e < CW; ++e) v[e] = gelu_t<T>(fmaf(v[e], sc[e], sh[e]));
    if constexpr (CW == 8) st16(yp + (long)t * a.C0, pack16<T>(v));
    else st4(yp + (long)t * a.C0, v);
  }
}

// ---------------------------------------------------------------- forward pass 2 on the matrix pipe (16-bit outputs, K <= 10; round 6)
// The pass above is bound by vector-ALU issue -- 10 FMAs of the conv + ~10 instructions of GroupNorm / GELU / packing per output, the
// teacher's 512 channels at 0.85 of the issue slots (profiles/r05_kernel_pmc_shares.txt) -- while the matrix pipe idles.  Here the
// 10-tap dot products run as ONE 16 x 16 x 32 MFMA per (16 frames x 16 channels): the reduction index holds three K-long segments,
//     [ x_hi . w_hi | x_lo . w_hi | x_hi . w_lo ]      x = x_hi + x_lo,  w = w_hi + w_lo  split in the 16-bit type,
// i.e. the fp32 product up to the dropped x_lo w_lo term and the second-order split residue (~2^-16 relative to |w||x|; the output is
// rounded to 2^-9 / 2^-11 afterwards) with fp32 accumulation.  Per 256-frame chunk the waveform windows are expanded ONCE into
// [frame][32] rows in LDS (the A operand: one ds_read_b128 per lane and frame tile, shared by all channel tiles).  A wave owns a block
// of 128 channels as EIGHT B tiles whose columns are dealt (column n of tile m <-> channel 8 n + m): the accumulators then hold, per
// lane, the 8 consecutive channels 8 n .. 8 n + 7 of four frames -- one 16-byte store per frame, 16 lanes = 256 contiguous bytes of
// a row per instruction (with the channels of a tile in neighbouring COLUMNS a store instruction covered 16 rows in 16-byte pieces:
// 2.61 ms per B = 256 launch of the teacher's 512 channels against 2.38 this way; the vector pass 2.93, statistics included).  What is
// left on the vector ALU is affine + GELU + packing: the pass is bound by its 8.4 GB of stores now (3.9 TB/s; without them it takes
// 0.62 ms).  Teacher forward 44.6 -> 43.9 ms at B = 256 (same box, three alternating rounds); the student's 128 channels are neutral.
// The fp32 parity mode keeps the exact vector pass above; so does any K > 10 (3 K must fit the 32-deep MFMA).
template <typename T, int NCB>   // NCB = 128-channel blocks = C0 / 128: 4 (one per wave, every frame tile) or 1 (the waves split the frame tiles)
__global__ void __launch_bounds__(256) conv0_apply_mfma_kernel(Conv0Args a, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               T* __restrict__ y, int chunks_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw = smem;
  const int b = blockIdx.y;
  const int swn = (((TCH - 1) * a.stride + a.K + MAXK) + 3) / 4 * 4;
  T* xr = reinterpret_cast<T*>(smem + swn);   // [TCH][32]: x_hi(0..K) | x_lo(0..K) | x_hi(0..K) | zeros
  const int K = a.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
  const int cb0 = NCB == 4 ? wave * 128 : 0;                 // this wave's channel block
  const int ts0 = NCB == 4 ? 0 : wave, tss = NCB == 4 ? 1 : 4;   // ... and frame tiles ts0, ts0 + tss, ...
  // ---- once per workgroup (it walks chunks_per_wg chunks of ONE utterance): the split weight fragments and the GroupNorm affine.
  // The C0 x K weights pass through LDS (coalesced loads; per-element global loads under the segment tests were 64 branchy,
  // serialised round trips per workgroup)
  for (int i = threadIdx.x; i < a.C0 * K; i += 256) smem[i] = a.w[i];
  __syncthreads();
  // B fragments: tile m, column n = lane & 15 <-> channel cb0 + 8 n + m; this lane's k = 8 q .. 8 q + 7 of  w_hi | w_hi | w_lo | 0
  bf16x8 Wf[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int c = cb0 + 8 * n + m;
    union { bf16x8 v; T e[8]; } u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * q + e;
      const int seg = (k >= K) + (k >= 2 * K) + (k >= 3 * K), j = k - seg * K;
      const float w = smem[c * K + (seg < 3 ? j : 0)];
      const T hi = (T)w;
      const T lo = (T)(w - (float)hi);
      u.e[e] = seg < 2 ? hi : (seg == 2 ? lo : (T)0.f);
    }
    Wf[m] = u.v;
  }
  float sc[8], sh[8];   // this lane's 8 consecutive channels
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cb0 + 8 * n + e;
    const float rs = rstd[b * a.C0 + c];
    sc[e] = rs * a.gamma[c];
    sh[e] = a.beta[c] - mean[b * a.C0 + c] * sc[e];
  }
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(xr) + n * 64 + q * 16;
  const int chunk_end = min(a.nchunk, ((int)blockIdx.x + 1) * chunks_per_wg);
  for (int chunk = blockIdx.x * chunks_per_wg; chunk < chunk_end; ++chunk) {
    __syncthreads();   // the previous chunk's rows (or the staged weights) have been read
    const int nt = stage_wave(a, b, chunk, sw);
    for (int i = threadIdx.x; i < TCH * 32; i += 256) {
      const int t = i >> 5, k = i & 31;
      const int seg = (k >= K) + (k >= 2 * K) + (k >= 3 * K), j = k - seg * K;
      float x = 0.f;
      if (t < nt && seg < 3) x = sw[t * a.stride + j];
      const T hi = (T)x;
      xr[i] = seg == 1 ? (T)(x - (float)hi) : hi;
    }
    __syncthreads();
    T* ybase = y + ((long)b * a.T0 + (long)chunk * TCH + 4 * q) * a.C0 + cb0 + 8 * n;
    for (int ts = ts0; ts < TCH / 16; ts += tss) {
      if (16 * ts >= nt) break;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + ts * 16 * 64);   // A: row = frame 16 ts + n
      f32x4 acc[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = mfma16<T>(xf, Wf[m], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // acc[m][i] = frame 16 ts + 4 q + i, channel cb0 + 8 n + m
        float o[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          o[m] = gelu_t<T>(fmaf(acc[m][i], sc[m], sh[m]));
        }
        if (16 * ts + 4 * q + i < nt)
          st16_nt(ybase + (long)(16 * ts + i) * a.C0, pack16<T>(o));   // (non-temporal: 8.4 GB that nothing re-reads soon; 2.36 -> 2.26 ms)
      }
    }
  }
}

// ---------------------------------------------------------------- backward, one pass over dy
// With g[t] = dy[t] * gelu'(z[t]) and xh the normalised conv output, GroupNorm's backward is
//   dv[t] = rstd * (gamma g[t] - m1 - xh[t] m2),   m1 = gamma mean_t g,  m2 = gamma mean_t (g xh),
// and dW[c][j] = sum_t dv[t] x_j[t] with x_j[t] = wave[stride t + j].  m1 / m2 are only known after a full pass over t -- the
// two-pass form therefore evaluated the conv and gelu' twice per element.  But dW is LINEAR in dv:
//   dW[c][j] = rstd * ( gamma A_j - m1 Sx_j - m2 * rstd * ((w_c Rx)_j - mean Sx_j) ),   A_j = sum_t g[t] x_j[t],
// where Sx_j = sum_t x_j[t] and Rx = sum_t x[t] x[t]^T are the waveform sums the forward's statistics pass already uses
// (sum_t xh[t] x_j[t] = rstd (sum_i w_ci Rx_ij - mean Sx_j)).  So ONE pass accumulates A_j (K values), S1 = sum g and
// S2 = sum g xh per (utterance, channel, block); a small fp64 combine applies the formula.
// Partials: wpart[b][blk][c][KT + 2] = A_0 .. A_{KT-1}, S1, S2.
template <typename T, int KT>
__global__ void __launch_bounds__(256) conv0_bwd_onepass_kernel(Conv0Args a, const float* mean, const float* rstd, const T* dy,
                                                                float* wpart) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw = smem;
  float* red = smem + ((TCH - 1) * a.stride + a.K + MAXK + 3) / 4 * 4;  // [256][CV]
  const int b = blockIdx.y;
  const ChanMap m(a.C0);
  float w[CV][KT], acc[CV][KT + 2];
  load_w4<KT>(a, m.c0, w);
  float mu[CV], rs[CV], ga[CV], be[CV];
#pragma unroll
  for (int e = 0; e < CV; ++e) {
    const int c = m.c0 + e;
    mu[e] = mean[b * a.C0 + c]; rs[e] = rstd[b * a.C0 + c]; ga[e] = a.gamma[c]; be[e] = a.beta[c];
#pragma unroll
    for (int j = 0; j < KT + 2; ++j) acc[e][j] = 0.f;
  }
  // the block walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... keeping its partial sums in registers
  for (int chunk = blockIdx.x; chunk < a.nchunk; chunk += gridDim.x) {
    __syncthreads();
    const int nt = stage_wave(a, b, chunk, sw);
    const T* dyp = dy + ((long)b * a.T0 + (long)chunk * TCH) * a.C0 + m.c0;
    // dy of PF frames is requested before the first of them is used: the pass reads 2 bytes per ~40 VALU instructions, and with
    // two loads in flight per wave (the compiler's own unroll) three waves per SIMD kept ~12 KB per CU on the wire -- 0.8 TB/s
    constexpr int PF = 4;
    for (int t0 = m.tl; t0 < nt; t0 += PF * m.nlan) {
      float dq[PF][CV];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int t = t0 + u * m.nlan;
        if (t < nt) ld4(dyp + (long)t * a.C0, dq[u]);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int t = t0 + u * m.nlan;
        if (t >= nt) break;
        float v[CV];
        conv4_at<KT>(sw, w, t, a.stride, v);
        const float* x = sw + t * a.stride;
#pragma unroll
        for (int e = 0; e < CV; ++e) {
          const float xh = (v[e] - mu[e]) * rs[e];
          const float g = dq[u][e] * gelu_grad_t<T>(fmaf(xh, ga[e], be[e]));
          v[e] = g;
          acc[e][KT] += g;
          acc[e][KT + 1] = fmaf(g, xh, acc[e][KT + 1]);
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const float xj = x[j];
#pragma unroll
          for (int e = 0; e < CV; ++e) acc[e][j] = fmaf(v[e], xj, acc[e][j]);
        }
      }
    }
  }
  float* out = wpart + (((long)b * gridDim.x + blockIdx.x) * a.C0 + m.c0) * (KT + 2);
#pragma unroll
  for (int j = 0; j < KT + 2; ++j) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CV; ++e) red[threadIdx.x * CV + e] = acc[e][j];
    __syncthreads();
    if (m.tl == 0) {
#pragma unroll
      for (int e = 0; e < CV; ++e) {
        float s = acc[e][j];
        for (int l = 1; l < m.nlan; ++l) s += red[(l * m.ngrp + m.cg) * CV + e];
        out[e * (KT + 2) + j] = s;
      }
    }
  }
}

// per (utterance b, block of 64 channels): waveform sums from the statistics partials and the per-channel sums of the pass
// partials (fp64, every thread walking a CONTIGUOUS run of the partial arrays -- the first form had one thread per channel walk
// all blocks with a 48-byte stride: 95-180 us for a few KB of arithmetic), then for every channel the formula above
// waveform sums of one utterance from the statistics partials: Sx_j (K values) and Rx (K x K), fp64, [B][K + K*K].  One workgroup per
// utterance, every (sum, half of the chunks) pair a thread with four loads in flight -- the combine kernel used to walk all chunks of
// every sum serially in each of its blocks (125 dependent loads at 10 s: 0.6 ms under the concurrent weight-gradient stream)
__global__ void __launch_bounds__(512) conv0_wave_sums_kernel(const float* __restrict__ spart, double* __restrict__ sums, int nchunk, int ns) {
  __shared__ double part[2][MAXK + MAXK * MAXK];
  const int b = blockIdx.x, s = threadIdx.x & 255, half = threadIdx.x >> 8;
  if (s < ns) {
    const int k_lo = half ? nchunk / 2 : 0, k_hi = half ? nchunk : nchunk / 2;
    const float* pp = spart + (long)b * nchunk * ns + s;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = k_lo;
    for (; k + 4 <= k_hi; k += 4) {
      const float v0 = pp[(long)k * ns], v1 = pp[(long)(k + 1) * ns], v2 = pp[(long)(k + 2) * ns], v3 = pp[(long)(k + 3) * ns];
      a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; k < k_hi; ++k) a0 += (double)pp[(long)k * ns];
    part[half][s] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (threadIdx.x < ns) sums[(long)b * ns + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x];
}

constexpr int CBW = 64;
__global__ void __launch_bounds__(256) conv0_bwd_combine_kernel(const double* wsum, const float* wpart, const float* w,
                                                                const float* gamma, const float* mean, const float* rstd, float* dw,
                                                                float* dgamma, float* dbeta, int nchunk, int nblk, int C0, int K,
                                                                int KT, int T0, float scale) {
  __shared__ double sx[MAXK], rx[MAXK * MAXK];
  __shared__ double acc[CBW * (MAXK + 2)];
  // blockIdx.z: a share of the partial blocks (the formula below is linear in A_j, S1, S2, so every share adds its own contribution:
  // with one partial block per 256-frame tile -- the GEMM-epilogue form, 125 per 10 s utterance -- a single walk was 0.68 ms of
  // dependent loads)
  const int b = blockIdx.x, cb = blockIdx.y * CBW;
  const int k_lo = (int)((long)nblk * blockIdx.z / gridDim.z), k_hi = (int)((long)nblk * (blockIdx.z + 1) / gridDim.z);
  const int ns = K + K * K;
  for (int s = threadIdx.x; s < ns; s += blockDim.x) {
    const double a = wsum[(long)b * ns + s];
    if (s < K) sx[s] = a; else rx[s - K] = a;
  }
  const int per = KT + 2, items = CBW * per;   // (channel, slot) pairs of this block: contiguous in every partial block
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const float* pp = wpart + ((long)b * nblk * C0 + cb) * per + it;
    const long st = (long)C0 * per;
    int k = k_lo;
    for (; k + 4 <= k_hi; k += 4) {   // four independent loads in flight
      const float v0 = pp[k * st], v1 = pp[(k + 1) * st], v2 = pp[(k + 2) * st], v3 = pp[(k + 3) * st];
      a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; k < k_hi; ++k) a0 += (double)pp[k * st];
    acc[it] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  const int c = cb + threadIdx.x;
  if (threadIdx.x < CBW && c < C0) {
    const double* A = acc + threadIdx.x * per;
    const double s1 = A[KT], s2 = A[KT + 1];
    const double g = gamma[c], rs = rstd[b * C0 + c], mu = mean[b * C0 + c];
    const double m1 = g * s1 / T0, m2 = g * s2 / T0;
    for (int j = 0; j < K; ++j) {
      double q = 0.0;
      for (int i = 0; i < K; ++i) q += (double)w[c * K + i] * rx[i * K + j];
      const double sxh = rs * (q - mu * sx[j]);  // sum_t xhat[t] x_j[t]
      atomicAdd(dw + c * K + j, (float)(rs * (g * A[j] - m1 * sx[j] - m2 * sxh) * (double)scale));
    }
    atomicAdd(dbeta + c, (float)s1);
    atomicAdd(dgamma + c, (float)s2);
  }
}

size_t smem_bytes(int stride, int K, bool red2) {
  size_t sw = ((size_t)((TCH - 1) * stride + K + MAXK) + 3) / 4 * 4;  // staged samples + the zero tail
  return (sw + (red2 ? 2 : 1) * 256 * CV) * sizeof(float);
}

}  // namespace

int fh_conv0_nchunk(int T0) { return (T0 + TCH - 1) / TCH; }
int fh_conv0_bwd_blocks(int B, int nchunk) {
  const int per_b = (1024 + B - 1) / B;
  return nchunk < per_b ? nchunk : (per_b > 1 ? per_b : 1);
}
// scratch (floats) shared by forward and backward: float2 partials [B][nchunk][C0] | float2 [B][C0] | backward partial sums
// [B][blocks][C0][MAXK + 2]
// scratch when the backward's partial sums come from the GEMM epilogue: one block per 256-frame tile of every utterance
// width (in float2) of the per-utterance block between the statistics partials and the backward's partial sums: MAXK + MAXK^2 fp64
// waveform sums (or, historically, C0 (a, b) pairs)
int fh_conv0_ab_width(int C0) { return C0 > MAXK + MAXK * MAXK ? C0 : MAXK + MAXK * MAXK; }
// one block of (K + 2 sums per channel, laid out with the MAXK + 2 pitch the combine reads) per utterance: the GEMM epilogue of layer
// 1's data gradient accumulates into it, the engine zeroes exactly this many floats at the head of the backward
long fh_conv0_fused_wpart_floats(int B, int C0) { return (long)B * C0 * (MAXK + 2); }
long fh_conv0_fused_scratch_n(int B, int T0, int C0) {   // (one block of backward sums per utterance: the GEMM epilogue accumulates)
  const long nchunk = fh_conv0_nchunk(T0);
  return 2L * B * nchunk * C0 + 2L * B * fh_conv0_ab_width(C0) + fh_conv0_fused_wpart_floats(B, C0);
}
long fh_conv0_scratch_n(int B, int T0, int C0) {
  const long nchunk = fh_conv0_nchunk(T0);
  return 2L * B * nchunk * C0 + 2L * B * fh_conv0_ab_width(C0) + (long)B * fh_conv0_bwd_blocks(B, (int)nchunk) * C0 * (MAXK + 2);
}

// ---------------------------------------------------------------- layer_norm-mode frontend (HuBERT Large): conv + bias + LayerNorm + GELU
//   reference: modules/module.py:54-64 (Conv1d(1, C0, k, stride, bias=conv_bias) -> TransposeLast -> Fp32LayerNorm(C0) ->
//   TransposeLast -> GELU) as the teacher's first block (extractor_mode = layer_norm, utils/utils.py:102-149 loads such a model).
// One WAVE per output frame: lane l owns the C0 / 64 channels l, l + 64, ... (their taps in registers), the frame's K samples are
// wave-uniform LDS reads, mean / variance over the channels are two wave reductions, and the normalised, activated row is written
// once -- the un-normalised conv output (B x T0 x C0, 2.1 GB at B = 64) never exists.
template <typename T, int KT, int CPL>
__global__ void __launch_bounds__(256) conv0_ln_gelu_kernel(Conv0Args a, const float* __restrict__ bias, float eps, T* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sw = smem;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int nt = stage_wave(a, b, chunk, sw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[CPL][KT], bi[CPL], ga[CPL], be[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) {
    const int c = lane * CPL + e;    // CPL consecutive channels per lane: one 2 * CPL-byte store per frame
#pragma unroll
    for (int j = 0; j < KT; ++j) w[e][j] = j < a.K ? a.w[c * a.K + j] : 0.f;
    bi[e] = bias ? bias[c] : 0.f;
    ga[e] = a.gamma[c];
    be[e] = a.beta[c];
  }
  const float inv_c = 1.0f / (float)a.C0;
  for (int t = wave; t < nt; t += 4) {
    const float* x = sw + t * a.stride;
    float xs[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) xs[j] = x[j];
    float v[CPL], s1 = 0.f;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
      float acc = bi[e];
#pragma unroll
      for (int j = 0; j < KT; ++j) acc = fmaf(w[e][j], xs[j], acc);
      v[e] = acc;
      s1 += acc;
    }
    const float mu = wave_sum(s1) * inv_c;
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
      const float d = v[e] - mu;
      s2 = fmaf(d, d, s2);
    }
    const float rs = rsqrtf(wave_sum(s2) * inv_c + eps);
    float o[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) o[e] = gelu_t<T>(fmaf((v[e] - mu) * rs, ga[e], be[e]));
    T* dst = y + ((long)b * a.T0 + (long)chunk * TCH + t) * a.C0 + lane * CPL;
    if constexpr (sizeof(T) == 2 && CPL == 8) {
      st16(dst, pack16<T>(o));
    } else {
#pragma unroll
      for (int e = 0; e < CPL; e += 4) st4(dst + e, o + e);
    }
  }
}

int fh_conv0_fwd_launch(int dtype, const float* wave, const float* w, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, float2* part, int B, int S, int C0, int K, int stride, float eps,
                        hipStream_t st) {
  if (K > MAXK || C0 % 64 != 0 || C0 / CV > 256 || 256 % (C0 / CV) != 0) return -201;
  Conv0Args a;
  a.wave = wave; a.w = w; a.gamma = gamma; a.beta = beta;
  a.B = B; a.S = S; a.C0 = C0; a.K = K; a.stride = stride;
  a.T0 = (S - K) / stride + 1;
  a.nchunk = fh_conv0_nchunk(a.T0);
  dim3 grid(a.nchunk, B), block(256);
  if (K + K * K > 2 * C0) return -202;  // the statistics partials reuse the [B][nchunk][C0] float2 scratch
  hipLaunchKernelGGL(conv0_stats_kernel, grid, dim3(128), smem_bytes(stride, K, false), st, a, (float*)part);
  FH_CHECK_LAUNCH();
  // the utterance's raw waveform sums stay in the scratch (the [B][ab width] block behind the partials) for the backward's combine
  double* wsum = reinterpret_cast<double*>(part + (long)B * a.nchunk * C0);
  hipLaunchKernelGGL(conv0_stats_final_kernel, dim3(B), dim3(256), 0, st, (const float*)part, w, mean, rstd, B, a.nchunk, C0, K,
                     a.T0, eps, wsum);
  FH_CHECK_LAUNCH();
#define FH_C0_APPLY(T, KT, CW) \
  hipLaunchKernelGGL((conv0_apply_kernel<T, KT, CW>), grid, block, smem_bytes(stride, K, false), st, a, mean, rstd, (T*)y)
  // 8 channels per thread (16-byte bf16 stores) measured SLOWER: 674 vs 416 us (B = 128, C0 = 128) -- the pass is bound by the latency
  // of its dependent LDS reads, and the wider thread halves the waves that hide it; FH_CONV0_WIDE=1 selects it for experiments
  static const bool wide_env = getenv("FH_CONV0_WIDE") && getenv("FH_CONV0_WIDE")[0] == '1';
  const bool wide = wide_env && C0 / 8 <= 256 && 256 % (C0 / 8) == 0;
  // 16-bit outputs with K <= 10 and 128 or 512 channels: the conv on the matrix pipe (conv0_apply_mfma_kernel); FH_CONV0_MFMA=0 keeps
  // the vector pass (A/B comparisons)
  static const bool mfma_env = !(getenv("FH_CONV0_MFMA") && getenv("FH_CONV0_MFMA")[0] == '0');
  if (dtype != 0 && mfma_env && K <= 10 && (C0 == 128 || C0 == 512)) {
    size_t sm = smem_bytes(stride, K, false) - 256 * CV * sizeof(float) + (size_t)TCH * 32 * 2;
    if (sm < (size_t)C0 * K * sizeof(float)) sm = (size_t)C0 * K * sizeof(float);   // the weights pass through the same LDS first
    // a workgroup walks several chunks of one utterance (its weight fragments and GroupNorm affine are built once): as many as leave
    // ~12 workgroups per CU in the grid, at most 8
    int cpw = (int)((long)a.nchunk * B / (12L * 256));
    cpw = cpw < 1 ? 1 : (cpw > 8 ? 8 : cpw);
    const dim3 mgrid((a.nchunk + cpw - 1) / cpw, B);
#define FH_C0_MFMA(T, NT) hipLaunchKernelGGL((conv0_apply_mfma_kernel<T, NT>), mgrid, block, sm, st, a, mean, rstd, (T*)y, cpw)
    if (dtype == 2) { if (C0 == 128) FH_C0_MFMA(f16, 1); else FH_C0_MFMA(f16, 4); }
    else { if (C0 == 128) FH_C0_MFMA(bf16, 1); else FH_C0_MFMA(bf16, 4); }
#undef FH_C0_MFMA
    FH_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == 0) { if (K == 10) FH_C0_APPLY(float, 10, 4); else FH_C0_APPLY(float, MAXK, 4); }
  else if (dtype == 2) { if (K == 10) FH_C0_APPLY(f16, 10, 4); else FH_C0_APPLY(f16, MAXK, 4); }
  else if (wide) { if (K == 10) FH_C0_APPLY(bf16, 10, 8); else FH_C0_APPLY(bf16, MAXK, 8); }
  else { if (K == 10) FH_C0_APPLY(bf16, 10, 4); else FH_C0_APPLY(bf16, MAXK, 4); }
#undef FH_C0_APPLY
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_conv0_bwd_launch(int dtype, const float* wave, const float* w, const float* gamma, const float* beta,
                        const void* dy, const float* mean, const float* rstd, float2* part, float2* ab, float* dw,
                        float* dgamma, float* dbeta, int B, int S, int C0, int K, int stride, float grad_scale,
                        hipStream_t st, int stats_ready, int ext_nblk) {
  if (K > MAXK || C0 % 64 != 0 || C0 / CV > 256 || 256 % (C0 / CV) != 0) return -201;
  Conv0Args a;
  a.wave = wave; a.w = w; a.gamma = gamma; a.beta = beta;
  a.B = B; a.S = S; a.C0 = C0; a.K = K; a.stride = stride;
  a.T0 = (S - K) / stride + 1;
  a.nchunk = fh_conv0_nchunk(a.T0);
  dim3 grid(a.nchunk, B), block(256);
  // statistics partials of the waveform (the same pass as the forward's) -> `part`, read as floats [B][nchunk][K + K*K]
  if (K + K * K > 2 * C0) return -202;
  if (!stats_ready) {  // (the engine's backward finds the forward's partials of the same waveform still in `part`)
    hipLaunchKernelGGL(conv0_stats_kernel, grid, dim3(128), smem_bytes(stride, K, false), st, a, (float*)part);
    FH_CHECK_LAUNCH();
  }
  // ext_nblk > 0: the per-(utterance, block, channel) sums are already in `wpart` -- ext_nblk blocks of [C0][10 + 2] per utterance, left
  // there by the conv-0 weight-gradient epilogue of layer 1's data-gradient GEMM (gemm_loops.cuh) -- and only the combine runs
  const int nblk = ext_nblk > 0 ? ext_nblk : fh_conv0_bwd_blocks(B, a.nchunk);  // ~1024 blocks in total; each writes its partial sums once
  dim3 wgrid(nblk, B);
  float* wpart = (float*)(ab + (long)B * fh_conv0_ab_width(C0));
  if (ext_nblk <= 0) {
#define FH_C0_WG(T, KT) \
  hipLaunchKernelGGL((conv0_bwd_onepass_kernel<T, KT>), wgrid, block, smem_bytes(stride, K, false), st, a, mean, rstd, (const T*)dy, wpart)
    if (dtype == 0) { if (K == 10) FH_C0_WG(float, 10); else FH_C0_WG(float, MAXK); }
    else if (dtype == 2) { if (K == 10) FH_C0_WG(f16, 10); else FH_C0_WG(f16, MAXK); }
    else { if (K == 10) FH_C0_WG(bf16, 10); else FH_C0_WG(bf16, MAXK); }
#undef FH_C0_WG
    FH_CHECK_LAUNCH();
  } else if (K != 10) {
    return -204;
  }
  const int ksplit = nblk >= 32 ? 8 : 1;
  const int ns = K + K * K;
  double* wsum = reinterpret_cast<double*>(ab);   // [B][ns] fp64 raw waveform sums: left there by the forward's statistics kernel (stats_ready)
  if (!stats_ready) {
    hipLaunchKernelGGL(conv0_wave_sums_kernel, dim3(B), dim3(512), 0, st, (const float*)part, wsum, a.nchunk, ns);
    FH_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(conv0_bwd_combine_kernel, dim3(B, (C0 + CBW - 1) / CBW, ksplit), dim3(256), 0, st, (const double*)wsum, (const float*)wpart, w, gamma, mean,
                     rstd, dw, dgamma, dbeta, a.nchunk, nblk, C0, K, K == 10 ? 10 : MAXK, a.T0, grad_scale);
  FH_CHECK_LAUNCH();
  return 0;
}

int fh_conv0_ln_gelu_launch(int dtype, const float* wave, const float* w, const float* bias, const float* gamma, const float* beta, void* y,
                            int B, int S, int C0, int K, int stride, float eps, hipStream_t st) {
  if (K > MAXK || (C0 != 512 && C0 != 256)) return -203;   // 64 lanes x 8 (4) consecutive channels
  Conv0Args a;
  a.wave = wave; a.w = w; a.gamma = gamma; a.beta = beta;
  a.B = B; a.S = S; a.C0 = C0; a.K = K; a.stride = stride;
  a.T0 = (S - K) / stride + 1;
  a.nchunk = fh_conv0_nchunk(a.T0);
  dim3 grid(a.nchunk, B), block(256);
  const size_t smem = smem_bytes(stride, K, false);
#define FH_C0_LN(T, KT, CPL) hipLaunchKernelGGL((conv0_ln_gelu_kernel<T, KT, CPL>), grid, block, smem, st, a, bias, eps, (T*)y)
#define FH_C0_LN_T(T)                                                    \
  do {                                                                   \
    if (C0 == 512) { if (K == 10) FH_C0_LN(T, 10, 8); else FH_C0_LN(T, MAXK, 8); } \
    else { if (K == 10) FH_C0_LN(T, 10, 4); else FH_C0_LN(T, MAXK, 4); } \
  } while (0)
  if (dtype == 0) FH_C0_LN_T(float);
  else if (dtype == 2) FH_C0_LN_T(f16);
  else FH_C0_LN_T(bf16);
#undef FH_C0_LN_T
#undef FH_C0_LN
  FH_CHECK_LAUNCH();
  return 0;
}
