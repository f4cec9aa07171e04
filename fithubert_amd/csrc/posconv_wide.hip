// Forward-only positional conv for wide groups (up to 64 channels per group): the frozen teacher's
// Conv1d(768, 768, k=128, pad=64, groups=16) + SamePad + GELU + residual (fairseq HubertModel.encoder.pos_conv, used by
// the reference through utils/utils.py:84-88).  Same scheme as posconv.hip (window staged once in LDS, one tap = one
// MFMA K-slab over the group's input channels, weights streamed from L2) with the channel padding at 64.
#include <cstdlib>

#include "common.cuh"
#include "kernels.h"
#include "posconv_taps.cuh"

namespace {

constexpr int CPW = 64;
// frames per workgroup = 4 waves x 16 NTT frames: each tap's weight fragments (streamed from L2 by every wave) serve NTT frame
// tiles, so NTT sets the L2 traffic per MFMA

// LDS row strides (bytes), both = 32 (mod 64): the 16 rows x 4 sixteen-byte columns of one fragment read then cover the 64 banks
// exactly once (see PC<>::RS in posconv.hip).  The 48-channel form keeps only its 96 real bytes per row.
template <typename T> struct PW { static constexpr int RS = CPW * (int)sizeof(T) + 32, RS48 = 48 * (int)sizeof(T); };

__global__ void __launch_bounds__(256) wnw_norm2_kernel(const float* __restrict__ v, float* __restrict__ norm2, int rows, int K) {
  __shared__ float red[256];
  const int j = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float x = v[(long)r * K + j];
    s = fmaf(x, x, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) norm2[j] = red[0];
}

template <typename T>
__global__ void wnw_layout_kernel(const float* __restrict__ gsc, const float* __restrict__ v, const float* __restrict__ norm2,
                                  T* __restrict__ w_fwd, int G, int cg, int K) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)G * K * CPW * CPW;
  if (o >= n) return;
  const int ci = (int)(o % CPW), co = (int)((o / CPW) % CPW);
  const int j = (int)((o / (CPW * CPW)) % K), g = (int)(o / ((long)CPW * CPW * K));
  float w = 0.f;
  if (co < cg && ci < cg) w = gsc[j] * v[((long)(g * cg + co) * cg + ci) * K + j] * rsqrtf(norm2[j]);
  if constexpr (sizeof(T) == 2) {
    // bf16: MFMA-fragment order -- the 64 lanes x 8 elements of one (tap, k-slab, 16-channel block) A operand are contiguous
    // (1 KiB = 8 full cache lines per wave load; in [co][ci] order the same load touched 16 half lines and the CU's vector
    // memory path, not the MFMA, set the pace)
    const int ks = ci / 32, qq = (ci % 32) / 8, e = ci % 8, cb = co / 16, rr = co % 16;
    const long dst = (((((long)g * K + j) * 2 + ks) * 4 + cb) * 64 + (qq * 16 + rr)) * 8 + e;
    w_fwd[dst] = from_f<T>(w);
  } else {
    w_fwd[o] = from_f<T>(w);
  }
}

__device__ __forceinline__ void ld2w(const float* p, float* f) {
  float2 v = *reinterpret_cast<const float2*>(p);
  f[0] = v.x; f[1] = v.y;
}
__device__ __forceinline__ void ld2w(const bf16* p, float* f) {
  uint32_t v = *reinterpret_cast<const uint32_t*>(p);
  f[0] = __uint_as_float(v << 16);
  f[1] = __uint_as_float(v & 0xffff0000u);
}
__device__ __forceinline__ void ld2w(const f16* p, float* f) {
  union { uint32_t u; f16 h[2]; } c;
  c.u = *reinterpret_cast<const uint32_t*>(p);
  f[0] = (float)c.h[0];
  f[1] = (float)c.h[1];
}
__device__ __forceinline__ void st2w(f16* p, const float* f) { *reinterpret_cast<uint32_t*>(p) = pack2<f16>(f[0], f[1]); }
__device__ __forceinline__ void st2w(float* p, const float* f) { *reinterpret_cast<float2*>(p) = make_float2(f[0], f[1]); }
__device__ __forceinline__ void st2w(bf16* p, const float* f) {
  *reinterpret_cast<uint32_t*>(p) = bf16_bits(f[0]) | (bf16_bits(f[1]) << 16);
}

template <typename T, int NTT>
__global__ void __launch_bounds__(256) posconv_wide_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y, int B, int Tn,
                                                           int D, int cg, int K, int pad_left) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RS = PW<T>::RS;
  constexpr bool IS_BF = sizeof(T) == 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  constexpr int TTW = 64 * NTT;
  const int t0 = blockIdx.x * TTW, g = blockIdx.y, b = blockIdx.z;
  const int nrows = TTW + K - 1;
  const int units = CPW / 2;
  for (int idx = threadIdx.x; idx < nrows * units; idx += blockDim.x) {
    const int row = idx / units, u = idx - row * units;
    const int t = t0 - pad_left + row;
    float f[2] = {0.f, 0.f};
    if (t >= 0 && t < Tn && 2 * u + 1 < cg) ld2w(x + ((long)b * Tn + t) * D + g * cg + 2 * u, f);
    st2w(reinterpret_cast<T*>(lds + row * RS) + 2 * u, f);
  }
  __syncthreads();
  const int ncb = (cg + 15) / 16;  // output-channel blocks that hold real channels
  f32x4 acc[4][NTT];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* wg = w + (long)g * K * CPW * CPW;
  if constexpr (IS_BF) {
    // the weight fragments of tap j + 3 are requested (from L2, by every wave) while tap j's MFMAs run: a tap is 24 MFMAs
    // (~160 ns) and only two waves share a SIMD, so a single tap of look-ahead left the L2 latency exposed.  Ring of four
    // register buffers, indices resolved at compile time (K % 4 == 0, checked on the host).
    bf16x8 wr[4][2][4];
    auto load_tap = [&](int slot, int tap) {
      const T* wj = wg + (long)(tap < K ? tap : K - 1) * CPW * CPW;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (cb < ncb) wr[slot][ks][cb] = *reinterpret_cast<const bf16x8*>(wj + ((ks * 4 + cb) * 64 + lane) * 8);
    };
#pragma unroll
    for (int u = 0; u < 3; ++u) load_tap(u, u);
    for (int j0 = 0; j0 < K; j0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        load_tap((u + 3) & 3, j + 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 xf[NTT];
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt)
            xf[tt] = *reinterpret_cast<const bf16x8*>(lds + (wave * (16 * NTT) + tt * 16 + r + j) * RS + ks * 64 + q * 16);
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            if (cb < ncb) {
#pragma unroll
              for (int tt = 0; tt < NTT; ++tt)
                acc[cb][tt] = mfma16<T>(wr[u][ks][cb], xf[tt], acc[cb][tt]);
            }
          }
        }
      }
    }
  } else {
    for (int j = 0; j < K; ++j) {
      const T* wj = wg + (long)j * CPW * CPW;
#pragma unroll
      for (int kr = 0; kr < 4; ++kr) {
        f32x4 xf[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
          xf[tt] = *reinterpret_cast<const f32x4*>(lds + (wave * (16 * NTT) + tt * 16 + r + j) * RS + kr * 64 + q * 16);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          if (cb < ncb) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(wj + (long)(cb * 16 + r) * CPW + kr * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[e], xf[tt][e], acc[cb][tt], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int tl = wave * (16 * NTT) + tt * 16 + r;
    const int t = t0 + tl;
    if (t >= Tn) continue;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = cb * 16 + q * 4 + 2 * h;
        if (co >= cg) continue;
        const long off = ((long)b * Tn + t) * D + g * cg + co;
        float v[2] = {acc[cb][tt][2 * h] + bias[g * cg + co], acc[cb][tt][2 * h + 1] + bias[g * cg + co + 1]};
        float xv[2];
        ld2w(reinterpret_cast<const T*>(lds + (tl + pad_left) * RS) + co, xv);
        v[0] = xv[0] + gelu_t<T>(v[0]);
        v[1] = xv[1] + gelu_t<T>(v[1]);
        st2w(y + off, v);
      }
    }
  }
}

// ---- 48 channels per group (HuBERT / wav2vec 2.0 Base: 768 / 16), 16-bit: the reduction index FLATTENED over (tap, input channel).
// Per tap the padded form multiplies two 32-wide K slabs of which the second is half zeros (48 -> 64); two taps are 96 = 3 x 32
// reduction elements exactly: slab 0 = tap 2p channels 0..31, slab 1 = tap 2p channels 32..47 | tap 2p+1 channels 0..15, slab 2 = tap 2p+1
// channels 16..47 -- a lane's 8 consecutive elements never straddle a tap (48 % 8 == 0), so an X fragment is still one 16-byte LDS
// read, at row + (0 | 1) of the staged window.  36 instead of 48 MFMAs per tap pair and frame tile.
constexpr bool flat48(int dtype, int cg, int K) { return dtype != 0 && cg == 48 && K % 8 == 0; }
template <typename T>
__global__ void wnw_layout48_kernel(const float* __restrict__ gsc, const float* __restrict__ v, const float* __restrict__ norm2,
                                    T* __restrict__ w_fwd, int G, int K) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)G * (K / 2) * 9 * 512;
  if (o >= n) return;
  const int e = (int)(o % 8), lane = (int)((o / 8) % 64), cb = (int)((o / 512) % 3), sl = (int)((o / (512 * 3)) % 3);
  const int p = (int)((o / (512 * 9)) % (K / 2)), g = (int)(o / ((long)512 * 9 * (K / 2)));
  const int kk = 32 * sl + 8 * (lane >> 4) + e, j = 2 * p + kk / 48, ci = kk % 48, co = 16 * cb + (lane & 15);
  w_fwd[o] = from_f<T>(gsc[j] * v[((long)(g * 48 + co) * 48 + ci) * K + j] * rsqrtf(norm2[j]));
}

template <typename T, int NTT, bool TOEP>   // TOEP: K = 128 (64 tap pairs) in Toeplitz order
__global__ void __launch_bounds__(256) posconv_wide48_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ y, int B, int Tn,
                                                             int D, int K, int pad_left) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RS = PW<T>::RS48, cg = 48;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  constexpr int TTW = 64 * NTT;
  const int t0 = blockIdx.x * TTW, g = blockIdx.y, b = blockIdx.z;
  const int nrows = TTW + K - 1;
  // a row holds exactly its 48 channels = 6 sixteen-byte units (D and the group offset are multiples of 8 elements here: 768 / 48);
  // BATCH units' loads are issued before the first is stored (as 4-byte pairs, one per trip, staging cost as much as a third of the taps)
  constexpr int units = cg / 8, BATCH = 4;
  for (int i0 = threadIdx.x; i0 < nrows * units; i0 += blockDim.x * BATCH) {
    V16 v[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int idx = i0 + j * (int)blockDim.x;
      const int row = idx / units, u = idx - row * units;
      const int t = t0 - pad_left + row;
      v[j] = v16_zero();
      if (idx < nrows * units && t >= 0 && t < Tn) v[j] = ld16(x + ((long)b * Tn + t) * D + g * cg + 8 * u);
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int idx = i0 + j * (int)blockDim.x;
      if (idx >= nrows * units) continue;
      const int row = idx / units, u = idx - row * units;
      st16(lds + row * RS + u * 16, v[j]);
    }
  }
  __syncthreads();
  f32x4 acc[3][NTT];
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* wg = w + (long)g * (K / 2) * 9 * 512;
  const int np = K / 2;
  if constexpr (TOEP) {
    // Toeplitz order over the tap pairs (posconv_taps.cuh): pair p = rho + 8 i of tile tt reads the window rows 16 (tt + i) + 2 rho + r,
    // so the 8 pair classes rho x 3 reduction slabs are walked as 24 classes of NI = 8 taps; a class's window fragment of step m
    // serves tap i = m - tt of every tile (12 MFMAs per 1 KB LDS read), its 24 weight fragments are requested one class ahead
    const unsigned char* xbase = lds + (wave * (16 * NTT) + r) * RS + q * 16;
    toeplitz_taps<T, 3, NTT, 8>(24, [&](int c) { return xbase + (c / 3) * (2 * RS) + (c % 3) * 64; },
                                [&](int c) { return wg + (long)(c / 3) * (9 * 512) + (c % 3) * (3 * 512); }, 8L * 9 * 512, 16 * RS, lane, acc);
  } else {
  int xoff[3];   // byte offset of this lane's 8 reduction elements of slab sl inside the window rows of a tap pair
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    const int kk = 32 * sl + 8 * q;
    xoff[sl] = (kk / 48) * RS + (kk % 48) * 2;
  }
  // weight fragments of tap pair p + 3 are requested (from L2) while pair p's 36 MFMAs per frame tile run; ring of four register sets
  bf16x8 wr[4][3][3];
  auto load_pair = [&](int slot, int p) {
    const T* wp = wg + (long)(p < np ? p : np - 1) * 9 * 512;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) wr[slot][sl][cb] = *reinterpret_cast<const bf16x8*>(wp + ((sl * 3 + cb) * 64 + lane) * 8);
  };
#pragma unroll
  for (int u = 0; u < 3; ++u) load_pair(u, u);
  for (int p0 = 0; p0 < np; p0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u;
      load_pair((u + 3) & 3, p + 3);
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        bf16x8 xf[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
          xf[tt] = *reinterpret_cast<const bf16x8*>(lds + (wave * (16 * NTT) + tt * 16 + r + 2 * p) * RS + xoff[sl]);
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = mfma16<T>(wr[u][sl][cb], xf[tt], acc[cb][tt]);
      }
    }
  }
  }
  // a lane holds four consecutive channels of one frame: one 8-byte store (and one 8-byte LDS read of the residual) per tile
  float bq[3][4];
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) bq[cb][e] = bias[g * cg + cb * 16 + q * 4 + e];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int tl = wave * (16 * NTT) + tt * 16 + r;
    const int t = t0 + tl;
    if (t >= Tn) continue;
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
      const int co = cb * 16 + q * 4;
      const long off = ((long)b * Tn + t) * D + g * cg + co;
      float xv[4], v[4];
      ld4(reinterpret_cast<const T*>(lds + (tl + pad_left) * RS) + co, xv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xv[e] + gelu_t<T>(acc[cb][tt][e] + bq[cb][e]);
      st4(y + off, v);
    }
  }
}

}  // namespace

extern "C" int fh_posconv_wide_prepare(int dtype, const float* g, const float* v, void* w_fwd, float* norm2, int D, int groups,
                                       int K, fh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const int cg = D / groups;
  if (cg > CPW || cg % 2 || K > 1024) return -551;
  hipLaunchKernelGGL(wnw_norm2_kernel, dim3(K), dim3(256), 0, st, v, norm2, D * cg, K);
  FH_CHECK_LAUNCH();
  if (flat48(dtype, cg, K)) {   // (the flattened layout is smaller than the padded one: same buffer)
    const long nf = (long)groups * (K / 2) * 9 * 512;
    if (dtype == 2)
      hipLaunchKernelGGL(wnw_layout48_kernel<f16>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, g, v, norm2, (f16*)w_fwd, groups, K);
    else
      hipLaunchKernelGGL(wnw_layout48_kernel<bf16>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, g, v, norm2, (bf16*)w_fwd, groups, K);
    FH_CHECK_LAUNCH();
    return 0;
  }
  const long n = (long)groups * K * CPW * CPW;
  if (dtype == 0)
    hipLaunchKernelGGL(wnw_layout_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, v, norm2, (float*)w_fwd,
                       groups, cg, K);
  else if (dtype == 2)
    hipLaunchKernelGGL(wnw_layout_kernel<f16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, v, norm2, (f16*)w_fwd,
                       groups, cg, K);
  else
    hipLaunchKernelGGL(wnw_layout_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, v, norm2, (bf16*)w_fwd,
                       groups, cg, K);
  FH_CHECK_LAUNCH();
  return 0;
}

extern "C" int fh_posconv_wide_forward(int dtype, const void* x, const void* w_fwd, const float* bias, void* y, int B, int T, int D,
                                       int groups, int K, fh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const int cg = D / groups;
  if (cg > CPW || cg % 2 || K % 4) return -551;
  static int ntt = 0;
  if (!ntt) {
    const char* e = getenv("FH_POSCONV_WIDE_NTT");
    ntt = e ? atoi(e) : 4;
    if (ntt != 4 && ntt != 8) ntt = 4;
  }
  const int TTW = 64 * ntt;
  dim3 grid((T + TTW - 1) / TTW, groups, B), block(256);
  if (flat48(dtype, cg, K)) {
#define FH_PW48(TY, NT, TP)                                                                                                    \
  do {                                                                                                                       \
    const size_t smem = (size_t)(64 * NT + K - 1) * PW<TY>::RS48;                                                            \
    hipFuncSetAttribute((const void*)posconv_wide48_kernel<TY, NT, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((posconv_wide48_kernel<TY, NT, TP>), grid, block, smem, st, (const TY*)x, (const TY*)w_fwd, bias, (TY*)y, B, T, \
                       D, K, K / 2);                                                                                         \
  } while (0)
    static const bool toep_on = [] { const char* e = getenv("FH_POSCONV_WIDE_TOEPLITZ"); return !e || atoi(e) != 0; }();
    const bool toep = toep_on && K == 128 && ntt == 4;
    if (dtype == 2 && toep) FH_PW48(f16, 4, true);
    else if (toep) FH_PW48(bf16, 4, true);
    else if (dtype == 2 && ntt == 8) FH_PW48(f16, 8, false);
    else if (dtype == 2) FH_PW48(f16, 4, false);
    else if (ntt == 8) FH_PW48(bf16, 8, false);
    else FH_PW48(bf16, 4, false);
#undef FH_PW48
    FH_CHECK_LAUNCH();
    return 0;
  }
#define FH_PW_LAUNCH(TY, NT)                                                                                                   \
  do {                                                                                                                       \
    const size_t smem = (size_t)(64 * NT + K - 1) * PW<TY>::RS;                                                              \
    hipFuncSetAttribute((const void*)posconv_wide_kernel<TY, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);     \
    hipLaunchKernelGGL((posconv_wide_kernel<TY, NT>), grid, block, smem, st, (const TY*)x, (const TY*)w_fwd, bias, (TY*)y, B, T, D, \
                       cg, K, K / 2);                                                                                        \
  } while (0)
  if (dtype == 0) FH_PW_LAUNCH(float, 4);
  else if (dtype == 2 && ntt == 8) FH_PW_LAUNCH(f16, 8);
  else if (dtype == 2) FH_PW_LAUNCH(f16, 4);
  else if (ntt == 8) FH_PW_LAUNCH(bf16, 8);
  else FH_PW_LAUNCH(bf16, 4);
#undef FH_PW_LAUNCH
  FH_CHECK_LAUNCH();
  return 0;
}
