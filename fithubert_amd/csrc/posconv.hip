// Convolutional positional embedding: weight-normed grouped Conv1d(D, D, k=128, pad=64, groups=16) + SamePad +
// GELU + residual, with backward.
//
//   reference: built at modules/module.py:186-208 (weight_norm(dim=2): w = g * v / ||v||_(0,1) per tap, SamePad
//   drops the last frame, GELU), applied at module.py:276-278 (x = x + pos_conv(x)).
//
// Per group this is a [T x (cg*128)] x [(cg*128) x cg] product with cg = 30 channels (fithubert.yaml; 48 for the 768-wide
// ex.yaml student): every tap is one (CP = 32) or two (CP = 64) 16x16x32 MFMA K-steps over the zero-padded input
// channels of the group.  All kernels are templated on the padded group width CP.  The input window of a 128-frame tile
// (128 + 127 rows x 32 channels) is staged once in LDS and re-read for all 128 taps; the per-tap 32x32 weight
// blocks stream from L2.  Data-gradient = the same kernel on flipped/transposed weights with pad 63 and a fused
// GELU' on the staged operand; weight-gradient reduces over time with transposing LDS reads (ds_read_b64_tr_b16).
#include "common.cuh"
#include "kernels.h"
#include "posconv_taps.cuh"

namespace {

// output frames per block (fwd / dgrad) = 4 waves x 16 NTT frames: each tap's weight fragments (streamed from L2) serve a wave's NTT tiles

template <typename T, int CP> struct PC {
  static constexpr int RS = CP * (int)sizeof(T) + 16;  // LDS row stride in bytes
  static constexpr int CB = CP / 16;                    // 16-channel MFMA blocks per group
  static constexpr int TS = (sizeof(T) == 4 && CP == 64) ? 128 : 256;  // frames per staged tile (wgrad; LDS budget)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// ---------------------------------------------------------------- weight norm forward + re-layout
// ||v[:, :, j]||^2 over the (co, ci) rows, in two deterministic stages.  v is [rows][K] with the tap index j contiguous, so the
// threads of a wave walk j (coalesced) and a block owns a contiguous run of rows; one block per tap, as before, read every cache
// line of v K times over (57 us for 7 MB -- 1 % of a B = 4 step).  part: [gridDim.x][K] block partials (scratch).
constexpr int WN_NB = 120;
__global__ void __launch_bounds__(256) wn_norm2_part_kernel(const float* __restrict__ v, float* __restrict__ part, int rows, int K) {
  __shared__ float red[256];
  const int per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int j0 = 0; j0 < K; j0 += 128) {                     // (K <= 128 in every shipped model: one trip)
    const int j = j0 + (threadIdx.x & 127);
    float s = 0.f;
    if (j < K)
      for (int r = r0 + (threadIdx.x >> 7); r < r1; r += 2) {
        const float x = v[(long)r * K + j];
        s = fmaf(x, x, s);
      }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 128 && j < K) part[(long)blockIdx.x * K + j] = red[threadIdx.x] + red[threadIdx.x + 128];
    __syncthreads();
  }
}
__global__ void wn_norm2_final_kernel(const float* __restrict__ part, float* __restrict__ norm2, int nb, int K) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[(long)b * K + j];
  norm2[j] = s;
}

// w_fwd[g][j][co][ci] = w[g*cg+co][ci][j] ; w_bwd[g][j'][ci][co] = w[g*cg+co][ci][K-1-j']  (zero padded to CP; fp32 as written,
// bf16 with each tap's [CP][CP] block permuted into MFMA-fragment order)
template <typename T, int CP>
__global__ void wn_layout_kernel(const float* __restrict__ gsc, const float* __restrict__ v,
                                 const float* __restrict__ norm2, T* __restrict__ w_fwd, T* __restrict__ w_bwd,
                                 float* __restrict__ norm, int G, int cg, int K) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)G * K * CP * CP;
  if (o >= n) return;
  const int b = (int)(o % CP), a = (int)((o / CP) % CP);
  const int j = (int)((o / (CP * CP)) % K), g = (int)(o / ((long)CP * CP * K));
  if (g == 0 && a == 0 && b == 0) norm[j] = sqrtf(norm2[j]);
  float wf = 0.f, wb = 0.f;
  if (a < cg && b < cg) {
    // forward: a = co, b = ci, tap j
    wf = gsc[j] * v[((long)(g * cg + a) * cg + b) * K + j] * rsqrtf(norm2[j]);
    // backward: a = ci, b = co, tap K-1-j
    const int jj = K - 1 - j;
    wb = gsc[jj] * v[((long)(g * cg + b) * cg + a) * K + jj] * rsqrtf(norm2[jj]);
  }
  long dst = o;
  if constexpr (sizeof(T) == 2) {
    // bf16: MFMA-fragment order -- the 64 lanes x 8 elements of one (tap, k-slab, 16-row block) A operand are contiguous, so a
    // wave's weight load is 8 full cache lines (with 64-wide rows the [row][col] order made it 16 half lines; see posconv_wide.hip)
    const int ks = b / 32, qq = (b % 32) / 8, e = b % 8, cb = a / 16, rr = a % 16;
    dst = (((((long)g * K + j) * (CP / 32) + ks) * (CP / 16) + cb) * 64 + (qq * 16 + rr)) * 8 + e;
  }
  w_fwd[dst] = from_f<T>(wf);
  w_bwd[dst] = from_f<T>(wb);
}

// ---------------------------------------------------------------- fragment helpers
template <typename T, int CP> struct Frag {  // 16-bit element types (bf16 / f16)
  typedef bf16x8 type;
  static constexpr int NK = CP / 32;  // fragment loads per group reduction (32 channels each)
  static __device__ __forceinline__ type lds_row(const unsigned char* lds, int row, int lane, int kr) {
    return *reinterpret_cast<const bf16x8*>(lds + row * PC<T, CP>::RS + kr * 64 + (lane >> 4) * 16);
  }
  static __device__ __forceinline__ type gmem_frag(const T* wtap, int cb, int lane, int kr) {  // fragment-order layout
    return *reinterpret_cast<const bf16x8*>(wtap + ((kr * (CP / 16) + cb) * 64 + lane) * 8);
  }
  static __device__ __forceinline__ f32x4 mma(type a, type b, f32x4 c) { return mfma16<T>(a, b, c); }
};
template <int CP> struct Frag<float, CP> {
  typedef f32x4 type;
  static constexpr int NK = CP / 16;
  static __device__ __forceinline__ type lds_row(const unsigned char* lds, int row, int lane, int kr) {
    return *reinterpret_cast<const f32x4*>(lds + row * PC<float, CP>::RS + kr * 64 + (lane >> 4) * 16);
  }
  static __device__ __forceinline__ type gmem_frag(const float* wtap, int cb, int lane, int kr) {  // [row][col] layout
    return *reinterpret_cast<const f32x4*>(wtap + (long)(cb * 16 + (lane & 15)) * CP + kr * 16 + (lane >> 4) * 4);
  }
  static __device__ __forceinline__ f32x4 mma(type a, type b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
};

__device__ __forceinline__ void ld2(const float* p, float* f) {
  float2 v = *reinterpret_cast<const float2*>(p);
  f[0] = v.x; f[1] = v.y;
}
__device__ __forceinline__ void ld2(const bf16* p, float* f) {
  uint32_t v = *reinterpret_cast<const uint32_t*>(p);
  f[0] = __uint_as_float(v << 16);
  f[1] = __uint_as_float(v & 0xffff0000u);
}
__device__ __forceinline__ void ld2(const f16* p, float* f) {
  union { uint32_t u; f16 h[2]; } c;
  c.u = *reinterpret_cast<const uint32_t*>(p);
  f[0] = (float)c.h[0];
  f[1] = (float)c.h[1];
}
__device__ __forceinline__ void st2(f16* p, const float* f) { *reinterpret_cast<uint32_t*>(p) = pack2<f16>(f[0], f[1]); }
__device__ __forceinline__ void st2(float* p, const float* f) { *reinterpret_cast<float2*>(p) = make_float2(f[0], f[1]); }
__device__ __forceinline__ void st2(bf16* p, const float* f) {
  *reinterpret_cast<uint32_t*>(p) = bf16_bits(f[0]) | (bf16_bits(f[1]) << 16);
}

// Accesses at 4-byte alignment (the 30-channel groups start at multiples of 60 bytes; global memory takes the misaligned
// dwordx2 / x3 / x4).  ld4u / st4u: four consecutive 16-bit channels = 8 bytes
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <typename T> __device__ __forceinline__ void st4u(T* p, const float* f) {
  u32x2_a4 v = {pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3])};
  *reinterpret_cast<u32x2_a4*>(p) = v;
}
template <typename T> __device__ __forceinline__ void ld4u(const T* p, float* f) {
  const u32x2_a4 v = *reinterpret_cast<const u32x2_a4*>(p);
  union { uint32_t u[2]; T h[4]; } c;
  c.u[0] = v.x; c.u[1] = v.y;
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = (float)c.h[e];
}
// the first nch (0, 2, 4, 6 or >= 8) of 8 consecutive 16-bit channels, the rest zero: ONE access of 4 / 8 / 12 / 16 bytes
__device__ __forceinline__ V16 ld_unit(const void* p, int nch) {
  V16 v = v16_zero();
  if (nch >= 8) {
    const u32x4_a4 t = *reinterpret_cast<const u32x4_a4*>(p);
    v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z; v.w[3] = t.w;
  } else if (nch == 6) {
    const u32x3_a4 t = *reinterpret_cast<const u32x3_a4*>(p);
    v.w[0] = t.x; v.w[1] = t.y; v.w[2] = t.z;
  } else if (nch == 4) {
    const u32x2_a4 t = *reinterpret_cast<const u32x2_a4*>(p);
    v.w[0] = t.x; v.w[1] = t.y;
  } else if (nch == 2) {
    v.w[0] = *reinterpret_cast<const uint32_t*>(p);
  }
  return v;
}
__device__ __forceinline__ void st_unit(void* p, const V16& v, int nch) {
  if (nch >= 8) {
    *reinterpret_cast<u32x4_a4*>(p) = u32x4_a4{v.w[0], v.w[1], v.w[2], v.w[3]};
  } else if (nch == 6) {
    *reinterpret_cast<u32x3_a4*>(p) = u32x3_a4{v.w[0], v.w[1], v.w[2]};
  } else if (nch == 4) {
    *reinterpret_cast<u32x2_a4*>(p) = u32x2_a4{v.w[0], v.w[1]};
  } else if (nch == 2) {
    *reinterpret_cast<uint32_t*>(p) = v.w[0];
  }
}
// stage `nrows` window rows (frame t_first + row, group g) of `src` into LDS as [row][CP] (zero outside [0,Tn)
// and for padded channels).  GP: multiply by gelu'(zc) and, for rows in [own0, own1), also write the product to dz.
template <typename T, int CP, bool GP>
__device__ __forceinline__ void stage_rows(unsigned char* lds, const T* src, const T* zc, T* dz, int b, int Tn, int D,
                                           int cg, int g, int t_first, int nrows, int own0, int own1) {
  if constexpr (sizeof(T) == 2) {
    // 8 channels per thread and access (16 bytes; 12 for the ragged last unit of a 30-channel group, whose rows start at multiples of
    // 60 bytes: 4-byte aligned only).  As 4-byte pairs, 24 - 40 trips per thread with their index arithmetic, staging alone took
    // 86 us (forward) / 200 us (data gradient) of 640 at B = 256.  BATCH units' loads are issued before the first is used.
    constexpr int u8 = CP / 8, BATCH = 4;
    const int total8 = nrows * u8;
    for (int i0 = threadIdx.x; i0 < total8; i0 += blockDim.x * BATCH) {
      V16 v[BATCH], zv[BATCH];
      long off[BATCH];
      int nch[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int idx = i0 + j * (int)blockDim.x;
        const int row = idx / u8, u = idx - row * u8;
        const int t = t_first + row;
        const int rem = cg - 8 * u;
        nch[j] = (idx < total8 && t >= 0 && t < Tn && rem > 0) ? (rem < 8 ? rem : 8) : 0;
        off[j] = ((long)b * Tn + t) * D + g * cg + 8 * u;
        v[j] = ld_unit(src + off[j], nch[j]);
        if (GP) zv[j] = ld_unit(zc + off[j], nch[j]);
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int idx = i0 + j * (int)blockDim.x;
        if (idx >= total8) continue;
        const int row = idx / u8, u = idx - row * u8;
        if (GP) {
          float f[8], z[8];
          unpack16<T>(v[j], f);
          unpack16<T>(zv[j], z);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= gelu_grad_t<T>(z[e]);
          v[j] = pack16<T>(f);
          if (row >= own0 && row < own1) st_unit(dz + off[j], v[j], nch[j]);
        }
        st16(lds + row * PC<T, CP>::RS + u * 16, v[j]);
      }
    }
    return;
  }
  // 4-byte units (the 30-channel groups of fithubert.yaml start at 60-byte offsets).  Every thread first issues the loads of BATCH
  // units, then converts and stores them: one unit per trip left every trip's global-load round trip exposed (24 - 40 serial trips per
  // workgroup: ~0.3 of the forward's time at B = 256)
  constexpr int units = CP / 2, BATCH = 8;
  const int total = nrows * units;
  for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * BATCH) {
    float f[BATCH][2], z[BATCH][2];
    long off[BATCH];
    bool ok[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int idx = i0 + j * (int)blockDim.x;
      const int row = idx / units, u = idx - row * units;
      const int t = t_first + row;
      ok[j] = idx < total && t >= 0 && t < Tn && 2 * u + 1 < cg;
      off[j] = ((long)b * Tn + t) * D + g * cg + 2 * u;
      f[j][0] = f[j][1] = 0.f;
      if (ok[j]) {
        ld2(src + off[j], f[j]);
        if (GP) ld2(zc + off[j], z[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int idx = i0 + j * (int)blockDim.x;
      if (idx >= total) continue;
      const int row = idx / units, u = idx - row * units;
      if (GP && ok[j]) {
        f[j][0] *= gelu_grad_t<T>(z[j][0]);
        f[j][1] *= gelu_grad_t<T>(z[j][1]);
        if (row >= own0 && row < own1) st2(dz + off[j], f[j]);
      }
      st2(reinterpret_cast<T*>(lds + row * PC<T, CP>::RS) + 2 * u, f[j]);
    }
  }
}

// ---------------------------------------------------------------- forward / data-gradient
template <typename T, int CP, bool BWD, int NTT = 4>   // NTT: 16-frame tiles per wave (64 NTT output frames per block)
__global__ void __launch_bounds__(256) posconv_kernel(const T* __restrict__ in, const T* __restrict__ zc_in,
                                                      const T* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ out1, T* __restrict__ out2, int B, int Tn, int D,
                                                      int cg, int K, int pad_left) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef Frag<T, CP> F;
  constexpr int CB = PC<T, CP>::CB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int TT = 64 * NTT;
  // XCD-contiguous order of the (time block, group, utterance) grid: the 16 groups of one utterance read neighbouring 60-byte pieces
  // of the same rows (round 6, same-box A/B at B = 256: forward 287 -> 279 us, data gradient 273 -> 261; the teacher's 48-channel
  // kernel, whose 16 groups' weights outgrow one L2, measured 4 % SLOWER that way and keeps the dispatch order)
  const int lg_ = fh_xcd_logical_block(), bg_ = lg_ / (int)gridDim.x;
  const int t0 = (lg_ - bg_ * (int)gridDim.x) * TT, g = bg_ % (int)gridDim.y, b = bg_ / (int)gridDim.y;
  const int nrows = TT + K - 1;
  stage_rows<T, CP, BWD>(lds, in, zc_in, out1, b, Tn, D, cg, g, t0 - pad_left, nrows, pad_left, pad_left + TT);
  __syncthreads();

  f32x4 acc[CB][NTT];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* wg = w + (long)g * K * CP * CP;
  const int r = lane & 15;
  bool taps_done = false;
  if constexpr (sizeof(T) == 2 && CP == 32) {
    if (K == 128) {
      constexpr int RS = PC<T, CP>::RS;
      const unsigned char* xrow = lds + (wave * (16 * NTT) + r) * RS + (lane >> 4) * 16;
      toeplitz_taps<T, CB, NTT, 8>(16, [&](int c) { return xrow + c * RS; }, [&](int c) { return wg + (long)c * (CP * CP); }, 16L * CP * CP, 16 * RS,
                                   lane, acc);
      taps_done = true;
    }
  }
  if (!taps_done) {
  // weights stream from L2 (identical for every workgroup of the group): keep the NEXT tap's fragments in flight
  typename F::type wn[F::NK][CB];
#pragma unroll
  for (int kr = 0; kr < F::NK; ++kr)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wn[kr][cb] = F::gmem_frag(wg, cb, lane, kr);
#pragma unroll 2
  for (int j = 0; j < K; ++j) {
    typename F::type wf[F::NK][CB];
#pragma unroll
    for (int kr = 0; kr < F::NK; ++kr)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) wf[kr][cb] = wn[kr][cb];
    const int jn = j + 1 < K ? j + 1 : j;
#pragma unroll
    for (int kr = 0; kr < F::NK; ++kr)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) wn[kr][cb] = F::gmem_frag(wg + (long)jn * CP * CP, cb, lane, kr);
#pragma unroll
    for (int kr = 0; kr < F::NK; ++kr) {
      typename F::type xf[NTT];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) xf[tt] = F::lds_row(lds, wave * (16 * NTT) + tt * 16 + r + j, lane, kr);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) acc[cb][tt] = F::mma(wf[kr][cb], xf[tt], acc[cb][tt]);
    }
  }
  }

  const int q = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    // a lane holds four consecutive channels of one frame: one 8-byte access per tensor (pairs only for a group's ragged last quad).
    // As 4-byte pairs the epilogue alone took 175 us of the forward's 640 at B = 256 (16 rows x 4 lanes x 4 bytes per store instruction)
    float bq[CB][4];
    if (!BWD) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = cb * 16 + q * 4 + e;
          bq[cb][e] = co < cg ? bias[g * cg + co] : 0.f;
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const int tl = wave * (16 * NTT) + tt * 16 + r;
      const int t = t0 + tl;
      if (t >= Tn) continue;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int co = cb * 16 + q * 4;
        if (co >= cg) continue;
        const bool quad = co + 3 < cg;   // else channels co, co + 1 only (cg is even)
        const long off = ((long)b * Tn + t) * D + g * cg + co;
        float v[4] = {acc[cb][tt][0], acc[cb][tt][1], acc[cb][tt][2], acc[cb][tt][3]};
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bq[cb][e];
          if (quad) st4u(out1 + off, v);
          else st2(out1 + off, v);
          float xv[4];
          ld4u(reinterpret_cast<const T*>(lds + (tl + pad_left) * PC<T, CP>::RS) + co, xv);   // (padded channels are staged as zeros)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = xv[e] + gelu_t<T>(v[e]);
          if (quad) st4u(out2 + off, v);
          else st2(out2 + off, v);
        } else {
          float dv[4] = {0.f, 0.f, 0.f, 0.f};
          if (quad) ld4u(in + off, dv);
          else ld2(in + off, dv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += dv[e];
          if (quad) st4u(out2 + off, v);
          else st2(out2 + off, v);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const int tl = wave * (16 * NTT) + tt * 16 + r;
    const int t = t0 + tl;
    if (t >= Tn) continue;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = cb * 16 + q * 4 + 2 * h;
        if (co + 1 >= cg + 1) continue;  // co, co+1 < cg (cg is even)
        if (co >= cg) continue;
        const long off = ((long)b * Tn + t) * D + g * cg + co;
        float v[2] = {acc[cb][tt][2 * h], acc[cb][tt][2 * h + 1]};
        if (!BWD) {
          v[0] += bias[g * cg + co];
          v[1] += bias[g * cg + co + 1];
          st2(out1 + off, v);
          float xv[2];
          ld2(reinterpret_cast<const T*>(lds + (tl + pad_left) * PC<T, CP>::RS) + co, xv);
          v[0] = xv[0] + gelu_t<T>(v[0]);
          v[1] = xv[1] + gelu_t<T>(v[1]);
          st2(out2 + off, v);
        } else {
          float dv[2];
          ld2(in + off, dv);
          v[0] += dv[0];
          v[1] += dv[1];
          st2(out2 + off, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- weight gradient
// dws[g][j][co][ci] += sum_{b,t} dzc[b,t,g*cg+co] * x[b, t + j - pad_left, g*cg+ci]
template <int CP> struct WgFrag {
  // 8 consecutive frames k = base + 8*(lane>>4) + e for column (c0 + lane&15): two transposing reads
  static __device__ __forceinline__ bf16x8 load(const unsigned char* lds, int base_row, int c0, int lane) {
    const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
    const unsigned char* a0 = lds + (base_row + 8 * q + qq) * PC<bf16, CP>::RS + (c0 + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * PC<bf16, CP>::RS));
    union {
      struct { s16x4 lo, hi; } s;
      bf16x8 v;
    } u;
    u.s.lo = lo;
    u.s.hi = hi;
    return u.v;
  }
};

// CB = 16-channel blocks that hold real channels (3 for the 48-channel groups: the fourth block of the 64-wide padding is all
// zeros), JW = taps per wave (accumulators: JW * CB * CB tiles).  A workgroup re-stages the same dz / x tiles for its 4 * JW taps
// only, so JW sets how often every tile is staged: 2 taps per wave made the 64-wide build 5x slower per FLOP than the 32-wide one.
template <typename T, int CP, int CB, int JW, bool TOEP = false>
__global__ void __launch_bounds__(256) posconv_wgrad_kernel(const T* __restrict__ dzc, const T* __restrict__ x,
                                                            float* __restrict__ dws, int B, int Tn, int D, int cg, int K,
                                                            int pad_left, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RS = PC<T, CP>::RS, TS = PC<T, CP>::TS;
  unsigned char* ldsA = lds;              // dzc tile  [TS][CP]
  unsigned char* ldsB = lds + TS * RS;    // x window  [TS + K - 1][CP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const int jbase = blockIdx.x * (4 * JW) + wave * JW;
  // Toeplitz tap assignment (16-bit, 8 taps per wave, K = 128): a wave owns the two tap classes cls0, cls0 + 1 (mod 32), i.e. taps
  // cls + 32 i: at K-step ks tap (cls, i) needs the window frames [32 (ks + i) + cls, + 32), the fragment tap (cls, i + 1) read one
  // K-step earlier.  With 8 CONSECUTIVE taps per wave every K-step reads 2 + 16 fragments for its 32 MFMAs -- 4 waves x 18 KB per 512
  // MFMA cycles is more than the LDS port delivers (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5 on top); this order reads 2 + 4.
  static_assert(!TOEP || (sizeof(T) == 2 && JW == 8 && TS % 32 == 0), "Toeplitz taps: 16-bit, 8 taps per wave");
  constexpr bool toep = TOEP;   // (the launcher picks it for K = 128)
  const int cls0 = blockIdx.x * 8 + wave * 2;
  const int ntt = (Tn + TS - 1) / TS;
  f32x4 acc[JW][CB][CB];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int ib = 0; ib < CB; ++ib) acc[jj][cb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 16-bit, 32-channel groups: the (dzc | x window) rows of tile i + 1 are fetched into REGISTERS (10 sixteen-byte units per thread
  // at K = 128; the ragged last unit of a 30-channel row is 12 bytes, rows start at multiples of 60 bytes) before tile i is multiplied,
  // and stored to LDS behind it.  Staged synchronously, a tile cost 40 dependent load -> LDS-store round trips per thread against
  // 2 us of MFMA work with one workgroup per CU: 608 us per launch, four fifths of it waiting.
  constexpr int NPRE = 12, U8 = CP / 8;
  const int unitsA = TS * U8, total_units = unitsA + (TS + K - 1) * U8;
  const bool prefetch = sizeof(T) == 2 && total_units <= NPRE * 256;
  V16 pre[NPRE];
  auto fetch = [&](int tile) {
    const int b = tile / ntt, t0 = (tile - b * ntt) * TS;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int idx = (int)threadIdx.x + j * 256;
      const bool isb = idx >= unitsA;
      const int loc = isb ? idx - unitsA : idx, row = loc / U8, u = loc - row * U8;
      const int t = (isb ? t0 - pad_left : t0) + row;
      const int rem = cg - 8 * u;
      const int nch = (idx < total_units && t >= 0 && t < Tn && rem > 0) ? (rem < 8 ? rem : 8) : 0;
      pre[j] = ld_unit((isb ? x : dzc) + ((long)b * Tn + t) * D + g * cg + 8 * u, nch);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int idx = (int)threadIdx.x + j * 256;
      if (idx < total_units) {
        const bool isb = idx >= unitsA;
        const int loc = isb ? idx - unitsA : idx, row = loc / U8, u = loc - row * U8;
        st16((isb ? ldsB : ldsA) + row * RS + u * 16, pre[j]);
      }
    }
  };
  if (prefetch && (int)blockIdx.z < B * ntt) fetch(blockIdx.z);
  for (int tile = blockIdx.z; tile < B * ntt; tile += nsplit) {
    const int b = tile / ntt, t0 = (tile - b * ntt) * TS;
    __syncthreads();
    if (prefetch) {
      deposit();
    } else {
      stage_rows<T, CP, false>(ldsA, dzc, nullptr, nullptr, b, Tn, D, cg, g, t0, TS, 0, 0);
      stage_rows<T, CP, false>(ldsB, x, nullptr, nullptr, b, Tn, D, cg, g, t0 - pad_left, TS + K - 1, 0, 0);
    }
    __syncthreads();
    if (prefetch && tile + nsplit < B * ntt) fetch(tile + nsplit);
    if constexpr (TOEP) {
      {
        // tap (class c, i) at K-step ks reads the window fragment that starts at frame 32 (ks + i) + c: a sliding window of four
        // fragments per class, ONE new per K-step
        bf16x8 bw[2][4][CB];
#pragma unroll
        for (int cls = 0; cls < 2; ++cls)
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int ib = 0; ib < CB; ++ib) bw[cls][n][ib] = WgFrag<CP>::load(ldsB, 32 * n + cls0 + cls, ib * 16, lane);
#pragma unroll
        for (int ks = 0; ks < TS / 32; ++ks) {
          bf16x8 fa[CB];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) fa[cb] = WgFrag<CP>::load(ldsA, ks * 32, cb * 16, lane);
#pragma unroll
          for (int cls = 0; cls < 2; ++cls)
#pragma unroll
            for (int ib = 0; ib < CB; ++ib) bw[cls][(ks + 3) & 3][ib] = WgFrag<CP>::load(ldsB, 32 * (ks + 3) + cls0 + cls, ib * 16, lane);
#pragma unroll
          for (int cls = 0; cls < 2; ++cls)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int ib = 0; ib < CB; ++ib)
                  acc[cls * 4 + i][cb][ib] = mfma16<T>(fa[cb], bw[cls][(ks + i) & 3][ib], acc[cls * 4 + i][cb][ib]);
        }
        continue;
      }
    }
    if constexpr (sizeof(T) == 2 && !TOEP) {
#pragma unroll 1
      for (int ks = 0; ks < TS / 32; ++ks) {
        bf16x8 fa[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) fa[cb] = WgFrag<CP>::load(ldsA, ks * 32, cb * 16, lane);
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
          bf16x8 fb[CB];
#pragma unroll
          for (int ib = 0; ib < CB; ++ib) fb[ib] = WgFrag<CP>::load(ldsB, ks * 32 + jbase + jj, ib * 16, lane);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ib = 0; ib < CB; ++ib)
              acc[jj][cb][ib] = mfma16<T>(fa[cb], fb[ib], acc[jj][cb][ib]);
        }
      }
    } else if constexpr (sizeof(T) == 4) {
      const int r = lane & 15, q = lane >> 4;
#pragma unroll 1
      for (int kk = 0; kk < TS / 4; ++kk) {  // 4 frames per MFMA: frame = 4*kk + q
        float fa[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          fa[cb] = *reinterpret_cast<const float*>(ldsA + (4 * kk + q) * RS + (cb * 16 + r) * 4);
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
          float fb[CB];
#pragma unroll
          for (int ib = 0; ib < CB; ++ib)
            fb[ib] = *reinterpret_cast<const float*>(ldsB + (4 * kk + q + jbase + jj) * RS + (ib * 16 + r) * 4);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ib = 0; ib < CB; ++ib)
              acc[jj][cb][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cb], fb[ib], acc[jj][cb][ib], 0, 0, 0);
        }
      }
    }
  }
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int j = toep ? cls0 + (jj >> 2) + 32 * (jj & 3) : jbase + jj;
    if (j >= K) continue;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int ib = 0; ib < CB; ++ib)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = cb * 16 + 4 * q + e, ci = ib * 16 + r;
          if (co < cg && ci < cg) atomicAdd(dws + (((long)g * K + j) * CP + co) * CP + ci, acc[jj][cb][ib][e]);
        }
  }
}

// ---------------------------------------------------------------- weight norm backward
__global__ void __launch_bounds__(256) wn_dot_kernel(const float* __restrict__ dws, const float* __restrict__ v,
                                                     float* __restrict__ dot, int cg, int K, int CP) {
  __shared__ float red[4];
  const int j = blockIdx.x, g = blockIdx.y;
  float s = 0.f;
  for (int idx = threadIdx.x; idx < cg * cg; idx += 256) {
    const int co = idx / cg, ci = idx - co * cg;
    s = fmaf(dws[(((long)g * K + j) * CP + co) * CP + ci], v[((long)(g * cg + co) * cg + ci) * K + j], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dot + j, red[0] + red[1] + red[2] + red[3]);
}

__global__ void wn_apply_kernel(const float* __restrict__ gsc, const float* __restrict__ v, const float* __restrict__ norm,
                                const float* __restrict__ dws, const float* __restrict__ dot, float* __restrict__ dg,
                                float* __restrict__ dv, int G, int cg, int K, int CP) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)G * cg * cg * K;
  if (o >= n) return;
  const int j = (int)(o % K);
  const int ci = (int)((o / K) % cg);
  const int oc = (int)(o / ((long)K * cg));  // g*cg + co
  const int g = oc / cg, co = oc - g * cg;
  const float nj = norm[j], dj = dot[j];
  const float dw = dws[(((long)g * K + j) * CP + co) * CP + ci];
  dv[o] += gsc[j] / nj * (dw - v[o] * dj / (nj * nj));
  if (oc == 0 && ci == 0) dg[j] += dj / nj;
}

template <typename T, int CP> size_t fwd_lds(int K, int tt) { return (size_t)(tt + K - 1) * PC<T, CP>::RS; }
template <typename T, int CP> size_t wg_lds(int K) { return (size_t)(2 * PC<T, CP>::TS + K - 1) * PC<T, CP>::RS; }
template <typename Kern> void set_lds(Kern k, size_t bytes) {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T, int CP>
int prep_t(const float* g, const float* v, const float* norm2, void* w_fwd, void* w_bwd, float* norm, int groups, int cg, int K,
           hipStream_t st) {
  const long n = (long)groups * K * CP * CP;
  hipLaunchKernelGGL((wn_layout_kernel<T, CP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, v, norm2, (T*)w_fwd,
                     (T*)w_bwd, norm, groups, cg, K);
  FH_CHECK_LAUNCH();
  return 0;
}
template <typename T, int CP, bool BWD, int NTT>
int conv_launch(const void* in, const void* zc, const void* w, const float* bias, void* o1, void* o2, int B, int Tn, int D, int groups,
                int K, int pad_left, hipStream_t st) {
  constexpr int TT = 64 * NTT;
  dim3 grid((Tn + TT - 1) / TT, groups, B), block(256);
  const size_t smem = fwd_lds<T, CP>(K, TT);
  set_lds(posconv_kernel<T, CP, BWD, NTT>, smem);
  hipLaunchKernelGGL((posconv_kernel<T, CP, BWD, NTT>), grid, block, smem, st, (const T*)in, (const T*)zc, (const T*)w, bias, (T*)o1,
                     (T*)o2, B, Tn, D, D / groups, K, pad_left);
  FH_CHECK_LAUNCH();
  return 0;
}
template <typename T, int CP, bool BWD>
int conv_t(const void* in, const void* zc, const void* w, const float* bias, void* o1, void* o2, int B, int Tn, int D, int groups,
           int K, int pad_left, hipStream_t st) {
  // the Toeplitz tap loop (16-bit, 32-channel groups, K = 128) with 8 tiles per wave: half the weight-fragment traffic per MFMA
  if constexpr (sizeof(T) == 2 && CP == 32) {
    static const int wide = [] { const char* e = getenv("FH_POSCONV_NTT"); return e ? atoi(e) : 8; }();
    if (K == 128 && wide == 8 && Tn > 256) return conv_launch<T, CP, BWD, 8>(in, zc, w, bias, o1, o2, B, Tn, D, groups, K, pad_left, st);
  }
  return conv_launch<T, CP, BWD, 4>(in, zc, w, bias, o1, o2, B, Tn, D, groups, K, pad_left, st);
}
template <typename T, int CP, int CB, int JW, bool TOEP>
int wgrad_launch(const void* dzc, const void* x, float* dws, int B, int Tn, int D, int groups, int K, hipStream_t st) {
  constexpr int TS = PC<T, CP>::TS;
  if (K % (4 * JW)) return -502;
  const int ntt = (Tn + TS - 1) / TS;
  int nsplit = B * ntt;
  if (nsplit > 8) nsplit = 8;
  dim3 grid(K / (4 * JW), groups, nsplit), block(256);
  const size_t smem = wg_lds<T, CP>(K);
  set_lds(posconv_wgrad_kernel<T, CP, CB, JW, TOEP>, smem);
  hipLaunchKernelGGL((posconv_wgrad_kernel<T, CP, CB, JW, TOEP>), grid, block, smem, st, (const T*)dzc, (const T*)x, dws, B, Tn, D, D / groups, K,
                     K / 2, nsplit);
  FH_CHECK_LAUNCH();
  return 0;
}
template <typename T, int CP, int CB, int JW>
int wgrad_t(const void* dzc, const void* x, float* dws, int B, int Tn, int D, int groups, int K, hipStream_t st) {
  if constexpr (sizeof(T) == 2 && JW == 8) {
    static const bool toep = [] { const char* e = getenv("FH_POSCONV_WGRAD_TOEPLITZ"); return !e || atoi(e) != 0; }();
    if (K == 128 && toep) return wgrad_launch<T, CP, CB, JW, true>(dzc, x, dws, B, Tn, D, groups, K, st);
  }
  return wgrad_launch<T, CP, CB, JW, false>(dzc, x, dws, B, Tn, D, groups, K, st);
}

}  // namespace

// padded group width the kernels use for D / groups channels per group (0: not supported)
int fh_posconv_cp(int cg) { return cg <= 32 ? 32 : (cg <= 64 ? 64 : 0); }

int fh_posconv_prep_launch(int dtype, const float* g, const float* v, void* w_fwd, void* w_bwd, float* norm, int D,
                           int groups, int K, hipStream_t st) {
  const int cg = D / groups, cp = fh_posconv_cp(cg);
  if (!cp || cg % 2 || K > 1024 || K % 32) return -501;
  // norm doubles as the ||v||^2 accumulator: [0, K) = norm, [K, 2K) = norm2 scratch
  float* norm2 = norm + K;
  // the block partials park in w_bwd, which the layout kernel (stream-ordered behind) overwrites: groups x K x CP x CP elements >= WN_NB x K floats
  if ((size_t)groups * K * cp * cp * (dtype == 0 ? 4 : 2) < (size_t)WN_NB * K * sizeof(float)) return -502;
  hipLaunchKernelGGL(wn_norm2_part_kernel, dim3(WN_NB), dim3(256), 0, st, v, (float*)w_bwd, D * cg, K);
  FH_CHECK_LAUNCH();
  hipLaunchKernelGGL(wn_norm2_final_kernel, dim3((K + 127) / 128), dim3(128), 0, st, (const float*)w_bwd, norm2, WN_NB, K);
  FH_CHECK_LAUNCH();
  if (cp == 32) return dtype == 0 ? prep_t<float, 32>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st)
                                  : dtype == 2 ? prep_t<f16, 32>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st)
                                  : prep_t<bf16, 32>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st);
  return dtype == 0 ? prep_t<float, 64>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st)
                    : dtype == 2 ? prep_t<f16, 64>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st)
                    : prep_t<bf16, 64>(g, v, norm2, w_fwd, w_bwd, norm, groups, cg, K, st);
}

int fh_posconv_fwd_launch(int dtype, const void* x, const void* w_fwd, const float* bias, void* zc, void* y, int B,
                          int Tn, int D, int groups, int K, hipStream_t st) {
  const int cg = D / groups, cp = fh_posconv_cp(cg);
  if (!cp || cg % 2 || K % 2) return -501;
  if (cp == 32) return dtype == 0 ? conv_t<float, 32, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st)
                                  : dtype == 2 ? conv_t<f16, 32, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st)
                                  : conv_t<bf16, 32, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st);
  return dtype == 0 ? conv_t<float, 64, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st)
                    : dtype == 2 ? conv_t<f16, 64, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st)
                    : conv_t<bf16, 64, false>(x, nullptr, w_fwd, bias, zc, y, B, Tn, D, groups, K, K / 2, st);
}

int fh_posconv_bwd_data_launch(int dtype, const void* dsum, const void* zc, const void* w_bwd, void* dzc, void* dx,
                               int B, int Tn, int D, int groups, int K, hipStream_t st) {
  const int cg = D / groups, cp = fh_posconv_cp(cg);
  if (!cp || cg % 2 || K % 2) return -501;
  const int pl = K - 1 - K / 2;
  if (cp == 32) return dtype == 0 ? conv_t<float, 32, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st)
                                  : dtype == 2 ? conv_t<f16, 32, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st)
                                  : conv_t<bf16, 32, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st);
  return dtype == 0 ? conv_t<float, 64, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st)
                    : dtype == 2 ? conv_t<f16, 64, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st)
                    : conv_t<bf16, 64, true>(dsum, zc, w_bwd, nullptr, dzc, dx, B, Tn, D, groups, K, pl, st);
}

int fh_posconv_bwd_weight_launch(int dtype, const void* dzc, const void* x, float* dws, float* dbias, int B, int Tn,
                                 int D, int groups, int K, hipStream_t st) {
  const int cg = D / groups, cp = fh_posconv_cp(cg);
  if (!cp || cg % 2 || K % 32) return -501;
  int rc;
  if (cp == 32) rc = dtype == 0 ? wgrad_t<float, 32, 2, 8>(dzc, x, dws, B, Tn, D, groups, K, st) : dtype == 2 ? wgrad_t<f16, 32, 2, 8>(dzc, x, dws, B, Tn, D, groups, K, st) : wgrad_t<bf16, 32, 2, 8>(dzc, x, dws, B, Tn, D, groups, K, st);
  else if (cg <= 48) rc = dtype == 0 ? wgrad_t<float, 64, 3, 4>(dzc, x, dws, B, Tn, D, groups, K, st) : dtype == 2 ? wgrad_t<f16, 64, 3, 4>(dzc, x, dws, B, Tn, D, groups, K, st) : wgrad_t<bf16, 64, 3, 4>(dzc, x, dws, B, Tn, D, groups, K, st);
  else rc = dtype == 0 ? wgrad_t<float, 64, 4, 2>(dzc, x, dws, B, Tn, D, groups, K, st) : dtype == 2 ? wgrad_t<f16, 64, 4, 2>(dzc, x, dws, B, Tn, D, groups, K, st) : wgrad_t<bf16, 64, 4, 2>(dzc, x, dws, B, Tn, D, groups, K, st);
  if (rc) return rc;
  if (dbias) return fh_colsum_launch(dtype, dzc, dbias, B * Tn, D, D, 1, 0, 0, 0, 1.0f, st);
  return 0;
}

int fh_posconv_wn_bwd_launch(const float* g, const float* v, const float* norm, const float* dws, float* dg, float* dv,
                             int D, int groups, int K, hipStream_t st) {
  const int cg = D / groups, cp = fh_posconv_cp(cg);
  if (!cp) return -501;
  // dot scratch lives behind norm / norm2: [2K, 3K)
  float* dot = const_cast<float*>(norm) + 2 * K;
  hipMemsetAsync(dot, 0, sizeof(float) * K, st);
  hipLaunchKernelGGL(wn_dot_kernel, dim3(K, groups), dim3(256), 0, st, dws, v, dot, cg, K, cp);
  FH_CHECK_LAUNCH();
  const long n = (long)groups * cg * cg * K;
  hipLaunchKernelGGL(wn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, v, norm, dws, dot, dg, dv,
                     groups, cg, K, cp);
  FH_CHECK_LAUNCH();
  return 0;
}
