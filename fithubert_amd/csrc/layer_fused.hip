// Fused full-row kernels of the student's post-LN transformer layer (reference modules/module.py:557-582), bf16:
//
//   attention output :  s1 = x + drop1(attn Wo^T + bo)                         xa   = LN1(s1)
//   feed-forward     :  z1 = xa W1^T + b1    h = drop2(gelu(z1))
//                       f2 = h W2^T + b2     s2 = xa + drop3(f2)               xout = LN2(s2)
//
// As separate launches each K = N = 480 product runs at 0.13 of the MFMA peak (250 tiles of 256 x 256 on 256 CUs, every CU in
// load -> MFMA -> epilogue lock-step) and every LayerNorm re-reads what a GEMM just wrote.  Here a workgroup owns 64 COMPLETE rows
// (N <= 512 columns, 4 waves x 8 column tiles of 16): the weights stream through LDS in 64-deep K slices (global -> registers
// while the previous slice is in the MFMAs -> LDS), the 64 x F hidden tile h of the feed-forward never leaves LDS between the two
// products, and bias / GELU / dropout / residual / LayerNorm run on the accumulators -- the row statistics are a shuffle across
// the four lanes that share a row plus one LDS exchange between the four waves.  Outputs go through a row-major LDS staging
// tile so that every global store is a 16-byte access of a contiguous row.  HBM traffic per layer: feed-forward 6 row-tensors
// (read xa; write z1, h, f2, s2, xout) against 9 unfused, attention output 4 against 5.
//
// Arithmetic mirrors the unfused path (gemm.hip epilogue + norm.hip): dropout masks are the same function of (seed, site,
// element index), GELU is the bf16-mode fast form, the pre-LayerNorm sum is rounded to bf16 before its statistics are taken
// (the unfused LayerNorm reads the stored bf16 tensor; the backward re-derives x-hat from that tensor with these statistics).
#include <cstdlib>

#include "kernels.h"
#include "common.cuh"

namespace {

// Out-of-range rows / reduction indices LOAD from this chunk instead of being skipped: a conditional load compiles to a branch
// whose join waits for the load (s_waitcnt vmcnt(0)), which serialised the 16 loads of a weight slice -- each paying a full L2
// round trip -- and made a K slice cost 11 K cycles instead of 1.5 K.
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

constexpr int FM = 64;     // rows per workgroup
constexpr int FBK = 64;    // reduction slice
constexpr int FNT = 8;     // 16-column tiles per wave (4 waves x 8 x 16 = 512 columns at most)

struct FusedArgs {
  const bf16* A;        // streamed operand of the first product [M][K1]
  const bf16* W1;       // feed-forward: fc1 weight [F][K1]; null for the single-product form
  const float* b1;
  const bf16* W2;       // [N][K2]  (K2 = F for the feed-forward, K1 otherwise)
  const float* b2;
  const bf16* res;      // residual [M][N]
  const float* gamma;
  const float* beta;
  bf16* z1;             // [M][F] pre-activation of fc1 (saved for backward), feed-forward only
  bf16* h;              // [M][F] post-GELU, post-dropout (fc2's operand; saved for its weight gradient)
  bf16* pre;            // [M][N] product + bias before dropout / residual (the layer_result "fc2 out"), or null
  bf16* s;              // [M][N] pre-LayerNorm sum (saved for backward)
  bf16* y;              // [M][N] LayerNorm output
  float* mean;
  float* rstd;
  int M, K1, F, N;
  fh_drop drop_act;     // after the GELU (activation_dropout), index i * F + j
  fh_drop drop_out;     // on the second product before the residual, index i * N + j
  float eps;
  int dbg;              // timing ablations (FH_FUSED_DBG; results are then wrong): 1 no weight fetch, 2 no MFMA, 4 no global stores / fills, 8 no LDS commit
};

// ---- LDS images
// operand tiles: 16-row x 32-k sub-tiles of 1 KiB (row r = 64 B; 16-byte chunk c of row r stored at chunk c ^ ((r >> 3) << 1)),
// the layout of gemm.hip's ring loaders: a wave's ds_read_b128 of an MFMA fragment is conflict-free.
__device__ __forceinline__ int sub_off(int row, int kchunk, int KS) {  // kchunk = k / 8
  const int r = row & 15, c = kchunk & 3;
  return ((row >> 4) * KS + (kchunk >> 2)) * 1024 + r * 64 + ((c ^ ((r >> 3) << 1)) << 4);
}
__device__ __forceinline__ bf16x8 sub_frag(const unsigned char* tile, int row0, int ks, int KS, int lane) {
  const int r = lane & 15, q = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(tile + ((row0 >> 4) * KS + ks) * 1024 + r * 64 + ((q ^ ((r >> 3) << 1)) << 4));
}

struct Smem {
  unsigned char* wst;   // weight slice  [N rows][64 k]      N * 128 bytes
  unsigned char* xst;   // operand slice [64 rows][64 k]     8 KiB (directly behind wst: the staging tile may run into it)
  unsigned char* ht;    // hidden tile   [64 rows][F k]      F * 128 bytes (feed-forward only)
  float* red;           // [64 rows][4 waves] x 2
  int stg_stride;       // row-major staging tile over wst + xst: N * 2 + 16 bytes per row
  int dbg;
};

// one product: acc[n][m] += A(64 rows x K) . W(N x K)^T for this wave's column tiles.  A streams from global (a_glob) or sits in LDS
// (a_lds, sub-tile image with KSa sub-tiles per row tile).
template <bool A_LDS>
__device__ __forceinline__ void product(const Smem& sm, const bf16* __restrict__ a_glob, int lda, int row0, int M,
                                        const unsigned char* a_lds, int KSa, const bf16* __restrict__ W, int N, int K, int wave, int lane,
                                        int nt_w, f32x4 (&acc)[FNT][4], int dbg = 0) {
  const int tid = wave * 64 + lane;
  constexpr int WIT = 512 * 8 / 256;   // 16-byte chunks of the widest weight slice per thread
  V16 wreg[WIT], xreg[2];
  const int nwchunks = N * 8;
  const bf16* zc = reinterpret_cast<const bf16*>(g_zero16);
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int c = it * 256 + tid;
      const int n = c >> 3, kc = c & 7, k = k0 + kc * 8;
      const bf16* src = (c < nwchunks && k < K) ? W + (long)n * K + k : zc;
      wreg[it] = ld16(src);
    }
    if (!A_LDS) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int c = it * 256 + tid;
        const int r = c >> 3, kc = c & 7, k = k0 + kc * 8;
        const bf16* src = (row0 + r < M && k < K) ? a_glob + (long)(row0 + r) * lda + k : zc;
        xreg[it] = ld16(src);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int c = it * 256 + tid;
      if (c < nwchunks) st16(sm.wst + sub_off(c >> 3, c & 7, 2), wreg[it]);
    }
    if (!A_LDS) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int c = it * 256 + tid;
        st16(sm.xst + sub_off(c >> 3, c & 7, 2), xreg[it]);
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += FBK) {
    __syncthreads();   // the previous slice's fragment reads are done
    if (!(dbg & 8)) commit();
    __syncthreads();
    if (k0 + FBK < K && !(dbg & 1)) fetch(k0 + FBK);   // in flight while this slice is in the MFMAs
    if (dbg & 2) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (k0 + 32 * ks >= K) break;
      bf16x8 fx[4], fy[FNT];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        fx[m] = A_LDS ? sub_frag(a_lds, 16 * m, (k0 >> 5) + ks, KSa, lane) : sub_frag(sm.xst, 16 * m, ks, 2, lane);
      // (column tiles beyond N -- the last wave's when N < 512 -- multiply whatever lies behind the weight slice in LDS: branch-free,
      //  their accumulators are never read)
#pragma unroll
      for (int n = 0; n < FNT; ++n) fy[n] = sub_frag(sm.wst, (wave * FNT + n) * 16, ks, 2, lane);
#pragma unroll
      for (int n = 0; n < FNT; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[n][m], 0, 0, 0);
    }
  }
  __syncthreads();   // every wave is done with wst / xst: they become the staging tile
}

__device__ __forceinline__ uint2 pack4(const float* v) {
  uint2 p;
  p.x = bf16_bits(v[0]) | (bf16_bits(v[1]) << 16);
  p.y = bf16_bits(v[2]) | (bf16_bits(v[3]) << 16);
  return p;
}
__device__ __forceinline__ float bf16_round(float x) { return __uint_as_float(bf16_bits(x) << 16); }

// staging tile (row-major, 64 rows x ncols bf16) -> global rows [row0, row0 + 64): 16-byte chunks, contiguous rows
__device__ __forceinline__ void flush_rows(const Smem& sm, bf16* __restrict__ dst, int ldd, int ncols, int row0, int M, int tid) {
  if (sm.dbg & 4) return;
  const int cpr = ncols >> 3;
  for (int c = tid; c < FM * cpr; c += 256) {
    const int r = c / cpr, cc = c - r * cpr;
    if (row0 + r < M) st16(dst + (long)(row0 + r) * ldd + cc * 8, ld16(sm.wst + r * sm.stg_stride + cc * 16));
  }
}
// global rows -> staging tile
__device__ __forceinline__ void fill_rows(const Smem& sm, const bf16* __restrict__ src, int lds_, int ncols, int row0, int M, int tid) {
  if (sm.dbg & 4) return;
  const int cpr = ncols >> 3;
  for (int c = tid; c < FM * cpr; c += 256) {
    const int r = c / cpr, cc = c - r * cpr;
    const bf16* p = row0 + r < M ? src + (long)(row0 + r) * lds_ + cc * 8 : reinterpret_cast<const bf16*>(g_zero16);
    st16(sm.wst + r * sm.stg_stride + cc * 16, ld16(p));
  }
}

template <bool FFN>
__global__ void __launch_bounds__(256) layer_rows_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int N = a.N, F = a.F;
  const int wcols = FFN ? (F > N ? F : N) : N;
  Smem sm;
  sm.wst = smem;
  sm.xst = smem + wcols * 128;
  sm.ht = sm.xst + 8192;
  sm.red = reinterpret_cast<float*>(sm.ht + (FFN ? F * 128 : 0));
  sm.stg_stride = wcols * 2 + 16;
  sm.dbg = a.dbg;
  const int row0 = blockIdx.x * FM;

  f32x4 acc[FNT][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < FNT; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  if (FFN) {
    // ---------------- fc1: z1 = A W1^T + b1 ; h = drop(gelu(z1))
    const int nt_w = max(0, min(FNT, F / 16 - wave * FNT));
    zero_acc();
    product<false>(sm, a.A, a.K1, row0, a.M, nullptr, 0, a.W1, F, a.K1, wave, lane, nt_w, acc, a.dbg);
    const DropKeySpan dks = drop_key_span(a.drop_act, (unsigned long long)row0 * F);
    // pass 1: z1 (bf16) -> staging -> global ; h -> LDS hidden tile (operand image) and kept in acc (as floats) for pass 2
#pragma unroll
    for (int n = 0; n < FNT; ++n) {
      if (n >= nt_w) continue;
      const int j = (wave * FNT + n) * 16 + 4 * q;
      float b4[4];
      ld4(a.b1 + j, b4);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int rl = 16 * m + r16;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[n][m][e] + b4[e];
        *reinterpret_cast<uint2*>(sm.wst + rl * sm.stg_stride + j * 2) = pack4(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast_f(v[e]);
        if (a.drop_act.p > 0.f) {
          const unsigned long long didx = (unsigned long long)(row0 + rl) * F + j;
          const DropKey dk = drop_key_at(dks, didx);
          float s0, s1, s2, s3;
          drop_pair(dk, (uint32_t)didx, s0, s1);
          drop_pair(dk, (uint32_t)didx + 2u, s2, s3);
          v[0] *= s0; v[1] *= s1; v[2] *= s2; v[3] *= s3;
        }
        const uint2 hb = pack4(v);
        // hidden tile: operand image, column j = reduction index of fc2
        *reinterpret_cast<uint2*>(sm.ht + sub_off(rl, j >> 3, F >> 5) + (j & 7) * 2) = hb;
        acc[n][m] = f32x4{__uint_as_float(hb.x << 16), __uint_as_float(hb.x & 0xffff0000u), __uint_as_float(hb.y << 16),
                          __uint_as_float(hb.y & 0xffff0000u)};
      }
    }
    __syncthreads();
    flush_rows(sm, a.z1, F, F, row0, a.M, tid);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < FNT; ++n) {
      if (n >= nt_w) continue;
      const int j = (wave * FNT + n) * 16 + 4 * q;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v[4] = {acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]};
        *reinterpret_cast<uint2*>(sm.wst + (16 * m + r16) * sm.stg_stride + j * 2) = pack4(v);
      }
    }
    __syncthreads();
    flush_rows(sm, a.h, F, F, row0, a.M, tid);
    // (product() opens with a barrier before it rewrites wst)
  }

  // ---------------- second product + residual + LayerNorm
  const int nt_w = max(0, min(FNT, N / 16 - wave * FNT));
  zero_acc();
  if (FFN) product<true>(sm, nullptr, 0, row0, a.M, sm.ht, F >> 5, a.W2, N, F, wave, lane, nt_w, acc, a.dbg);
  else product<false>(sm, a.A, a.K1, row0, a.M, nullptr, 0, a.W2, N, a.K1, wave, lane, nt_w, acc, a.dbg);

  // product + bias; optional store of it (fc2's "layer result")
#pragma unroll
  for (int n = 0; n < FNT; ++n) {
    if (n >= nt_w) continue;
    const int j = (wave * FNT + n) * 16 + 4 * q;
    float b4[4];
    ld4(a.b2 + j, b4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[n][m][e] += b4[e];
      if (a.pre) {
        float v[4] = {acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]};
        *reinterpret_cast<uint2*>(sm.wst + (16 * m + r16) * sm.stg_stride + j * 2) = pack4(v);
      }
    }
  }
  if (a.pre) {
    __syncthreads();
    flush_rows(sm, a.pre, N, N, row0, a.M, tid);
    __syncthreads();
  }
  // residual rows -> staging (coalesced), then dropout + residual on the accumulators, rounded to bf16 as the stored sum is
  fill_rows(sm, a.res, N, N, row0, a.M, tid);
  __syncthreads();
  const DropKeySpan dko = drop_key_span(a.drop_out, (unsigned long long)row0 * N);
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < FNT; ++n) {
    if (n >= nt_w) continue;
    const int j = (wave * FNT + n) * 16 + 4 * q;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rl = 16 * m + r16;
      float v[4] = {acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]};
      if (a.drop_out.p > 0.f) {
        const unsigned long long didx = (unsigned long long)(row0 + rl) * N + j;
        const DropKey dk = drop_key_at(dko, didx);
        float s0, s1, s2, s3;
        drop_pair(dk, (uint32_t)didx, s0, s1);
        drop_pair(dk, (uint32_t)didx + 2u, s2, s3);
        v[0] *= s0; v[1] *= s1; v[2] *= s2; v[3] *= s3;
      }
      float rr[4];
      ld4(reinterpret_cast<const bf16*>(sm.wst + rl * sm.stg_stride) + j, rr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = bf16_round(v[e] + rr[e]);
        acc[n][m][e] = v[e];
        rsum[m] += v[e];
      }
    }
  }
  __syncthreads();   // residual reads done: the staging tile takes the sum
#pragma unroll
  for (int n = 0; n < FNT; ++n) {
    if (n >= nt_w) continue;
    const int j = (wave * FNT + n) * 16 + 4 * q;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float v[4] = {acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]};
      *reinterpret_cast<uint2*>(sm.wst + (16 * m + r16) * sm.stg_stride + j * 2) = pack4(v);
    }
  }
  // ---- row statistics: lanes q = 0..3 share a row, then the four waves
  float mean[4], rstd[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float s = rsum[m];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) sm.red[(16 * m + r16) * 4 + wave] = s;
  }
  __syncthreads();   // (also: the sum tile is complete in staging)
  flush_rows(sm, a.s, N, N, row0, a.M, tid);
  const float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 p = *reinterpret_cast<const float4*>(sm.red + (16 * m + r16) * 4);
    mean[m] = (p.x + p.y + p.z + p.w) * inv_n;
  }
  float vsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < FNT; ++n) {
    if (n >= nt_w) continue;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = acc[n][m][e] - mean[m];
        vsum[m] = fmaf(d, d, vsum[m]);
      }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float s = vsum[m];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) sm.red[256 + (16 * m + r16) * 4 + wave] = s;
  }
  __syncthreads();   // variance partials visible; every thread is past its flush of the sum tile
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 p = *reinterpret_cast<const float4*>(sm.red + 256 + (16 * m + r16) * 4);
    rstd[m] = rsqrtf((p.x + p.y + p.z + p.w) * inv_n + a.eps);
    if (wave == 0 && q == 0 && row0 + 16 * m + r16 < a.M) {
      a.mean[row0 + 16 * m + r16] = mean[m];
      a.rstd[row0 + 16 * m + r16] = rstd[m];
    }
  }
#pragma unroll
  for (int n = 0; n < FNT; ++n) {
    if (n >= nt_w) continue;
    const int j = (wave * FNT + n) * 16 + 4 * q;
    float g4[4], be4[4];
    ld4(a.gamma + j, g4);
    ld4(a.beta + j, be4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf((acc[n][m][e] - mean[m]) * rstd[m], g4[e], be4[e]);
      *reinterpret_cast<uint2*>(sm.wst + (16 * m + r16) * sm.stg_stride + j * 2) = pack4(v);
    }
  }
  __syncthreads();
  flush_rows(sm, a.y, N, N, row0, a.M, tid);
}

size_t fused_smem(int N, int F, bool ffn) {
  const int wcols = ffn ? (F > N ? F : N) : N;
  return (size_t)wcols * 128 + 8192 + (ffn ? (size_t)F * 128 : 0) + 2 * 256 * sizeof(float);
}

}  // namespace

// shapes the fused kernels take (bf16 only): full rows of at most 512 columns in 16-column tiles, 32-deep reduction steps
bool fh_layer_fused_ok(int dtype, int D, int Fd) {
  return dtype == 1 && D % 32 == 0 && Fd % 32 == 0 && D <= 512 && Fd <= 512 && D >= 64 && Fd >= 64;
}

// s = res + drop(A W^T + b);  y = LayerNorm(s)         (attention output projection + residual + LayerNorm)
int fh_rows_linear_ln_launch(const void* A, const void* W, const float* b, const void* res, const float* gamma, const float* beta, void* pre,
                             void* s, void* y, float* mean, float* rstd, int M, int K, int N, float eps, fh_drop drop, hipStream_t st) {
  if (!fh_layer_fused_ok(1, N, K)) return -701;
  FusedArgs a{};
  a.A = (const bf16*)A; a.W1 = nullptr; a.b1 = nullptr; a.W2 = (const bf16*)W; a.b2 = b; a.res = (const bf16*)res;
  a.gamma = gamma; a.beta = beta; a.z1 = nullptr; a.h = nullptr; a.pre = (bf16*)pre; a.s = (bf16*)s; a.y = (bf16*)y;
  a.mean = mean; a.rstd = rstd; a.M = M; a.K1 = K; a.F = K; a.N = N;
  a.drop_act.p = 0.f; a.drop_act.seed = 0; a.drop_act.site = 0;
  a.drop_out = drop; a.eps = eps;
  a.dbg = getenv("FH_FUSED_DBG") ? atoi(getenv("FH_FUSED_DBG")) : 0;
  const size_t smem = fused_smem(N, K, false);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)layer_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(layer_rows_kernel<false>, dim3((M + FM - 1) / FM), dim3(256), smem, st, a);
  FH_CHECK_LAUNCH();
  return 0;
}

// z1 = A W1^T + b1; h = drop_act(gelu(z1)); f2 = h W2^T + b2; s = res + drop_out(f2); y = LayerNorm(s)
int fh_rows_ffn_ln_launch(const void* A, const void* W1, const float* b1, const void* W2, const float* b2, const void* res, const float* gamma,
                          const float* beta, void* z1, void* h, void* f2, void* s, void* y, float* mean, float* rstd, int M, int D, int F,
                          float eps, fh_drop drop_act, fh_drop drop_out, hipStream_t st) {
  if (!fh_layer_fused_ok(1, D, F)) return -701;
  FusedArgs a{};
  a.A = (const bf16*)A; a.W1 = (const bf16*)W1; a.b1 = b1; a.W2 = (const bf16*)W2; a.b2 = b2; a.res = (const bf16*)res;
  a.gamma = gamma; a.beta = beta; a.z1 = (bf16*)z1; a.h = (bf16*)h; a.pre = (bf16*)f2; a.s = (bf16*)s; a.y = (bf16*)y;
  a.mean = mean; a.rstd = rstd; a.M = M; a.K1 = D; a.F = F; a.N = D;
  a.drop_act = drop_act; a.drop_out = drop_out; a.eps = eps;
  a.dbg = getenv("FH_FUSED_DBG") ? atoi(getenv("FH_FUSED_DBG")) : 0;
  const size_t smem = fused_smem(D, F, true);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)layer_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(layer_rows_kernel<true>, dim3((M + FM - 1) / FM), dim3(256), smem, st, a);
  FH_CHECK_LAUNCH();
  return 0;
}
