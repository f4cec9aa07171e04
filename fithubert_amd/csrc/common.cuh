// Common device helpers for the FitHuBERT gfx950 kernels.
// Element type T is float (exact-parity mode, f32 MFMA), bf16 (bf16 MFMA, f32 accumulate) or f16 (IEEE half -- the reference's
// own AMP dtype, train.py:469,496 -- on the f16 MFMA forms, which run at the bf16 rate; f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fithubert_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define FH_WAVE 64

#define FH_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int VEC = 4;   // elements per 16 bytes
  static constexpr int DT = 0;
};
template <> struct ElemTraits<bf16> {
  static constexpr int VEC = 8;
  static constexpr int DT = 1;
};
template <> struct ElemTraits<f16> {
  static constexpr int VEC = 8;
  static constexpr int DT = 2;
};
// dtype code -> element size / elements per 16 bytes (host side)
static inline int fh_esize(int dtype) { return dtype == 0 ? 4 : 2; }
static inline int fh_vec(int dtype) { return dtype == 0 ? 4 : 8; }

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f(f16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ f16 from_f<f16>(float x) { return (f16)x; }

// 16-byte vector of T, as 4 dwords
struct alignas(16) V16 {
  uint32_t w[4];
};
__device__ __forceinline__ V16 v16_zero() {
  V16 v;
  v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
  return v;
}
// Logical block index under the XCD-contiguous remap.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), so
// neighbours in the grid -- readers of neighbouring pieces of the same memory rows -- land on as many L2s as there are XCDs; here XCD k
// takes a contiguous range of the logical order instead (bijective for any grid size; speed only, never correctness).
__device__ __forceinline__ int fh_xcd_logical_block() {
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * (int)gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int per = nwg >> 3, rem = nwg & 7, xcd = lin & 7;
  return (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (lin >> 3);
}
#ifdef FH_LD_NT   // (per translation unit, defined in front of this header: every 16-byte load of the unit is non-temporal -- norm.hip, loss.hip)
__device__ __forceinline__ V16 ld16(const void* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  const u32x4_ q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(p));
  V16 v;
  v.w[0] = q[0]; v.w[1] = q[1]; v.w[2] = q[2]; v.w[3] = q[3];
  return v;
}
#else
__device__ __forceinline__ V16 ld16(const void* p) { return *reinterpret_cast<const V16*>(p); }
#endif
__device__ __forceinline__ void st16(void* p, const V16& v) { *reinterpret_cast<V16*>(p) = v; }
// non-temporal form (global_store_dwordx4 ... nt): for streamed outputs far larger than the caches that nothing re-reads soon
__device__ __forceinline__ void st16_nt(void* p, const V16& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  const u32x4_ q = {v.w[0], v.w[1], v.w[2], v.w[3]};
  __builtin_nontemporal_store(q, reinterpret_cast<u32x4_*>(p));
}

// unpack / pack a 16-byte vector into VEC floats
template <typename T> __device__ __forceinline__ void unpack16(const V16& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const V16& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16>(const V16& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v.w[i] << 16);
    f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack16<f16>(const V16& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    union { uint32_t u; f16 h[2]; } c;
    c.u = v.w[i];
    f[2 * i] = (float)c.h[0];
    f[2 * i + 1] = (float)c.h[1];
  }
}
__device__ __forceinline__ uint32_t bf16_bits(float x) {
  bf16 b = (bf16)x;
  return (uint32_t) * reinterpret_cast<uint16_t*>(&b);
}
// two floats -> one dword of 16-bit elements of type T (round to nearest even), low half first
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<bf16>(float a, float b) { return bf16_bits(a) | (bf16_bits(b) << 16); }
template <> __device__ __forceinline__ uint32_t pack2<f16>(float a, float b) {
  union { uint32_t u; f16 h[2]; } c;
  c.h[0] = (f16)a;
  c.h[1] = (f16)b;
  return c.u;
}
// 16 x 16 x 32 and 32 x 32 x 16 MFMA on 16-bit operands of type T (fragments travel as opaque 16-byte vectors typed bf16x8)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <typename T> __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (ElemTraits<T>::DT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x16_t mfma32(const bf16x8& a, const bf16x8& b, const f32x16_t& c) {
  if constexpr (ElemTraits<T>::DT == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ V16 pack16(const float* f);
template <> __device__ __forceinline__ V16 pack16<float>(const float* f) {
  V16 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
  return v;
}
template <> __device__ __forceinline__ V16 pack16<bf16>(const float* f) {
  V16 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = bf16_bits(f[2 * i]) | (bf16_bits(f[2 * i + 1]) << 16);
  return v;
}
template <> __device__ __forceinline__ V16 pack16<f16>(const float* f) {
  V16 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v.w[i] = pack2<f16>(f[2 * i], f[2 * i + 1]);
  return v;
}

// load / store 4 consecutive T elements as floats (8 B for bf16, 16 B for float)
__device__ __forceinline__ void ld4(const float* p, float* f) {
  float4 v = *reinterpret_cast<const float4*>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16* p, float* f) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  f[0] = __uint_as_float(v.x << 16);
  f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16);
  f[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void ld4(const f16* p, float* f) {
  union { uint2 v; f16 h[4]; } c;
  c.v = *reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = (float)c.h[i];
}
__device__ __forceinline__ void st4(f16* p, const float* f) {
  uint2 v;
  v.x = pack2<f16>(f[0], f[1]);
  v.y = pack2<f16>(f[2], f[3]);
  *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void st4(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void st4(bf16* p, const float* f) {
  uint2 v;
  v.x = bf16_bits(f[0]) | (bf16_bits(f[1]) << 16);
  v.y = bf16_bits(f[2]) | (bf16_bits(f[3]) << 16);
  *reinterpret_cast<uint2*>(p) = v;
}

// ---------------------------------------------------------------- math
// Exact-erf GELU (torch's default, modules/module.py:63-73) evaluated with the Abramowitz-Stegun 7.1.26 rational form:
// |erf error| <= 1.5e-7 (below fp32 round-off of the surrounding GEMMs), one v_rcp + one v_exp, no branches.
// libdevice erff() costs several times more VALU and dominated the epilogues of the short-K conv GEMMs.
// The same exponential also yields the Gaussian pdf needed by the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float u = fabsf(x) * 0.70710678118654752f;
  // v_rcp_f32 (1 ulp) instead of the correctly rounded __frcp_rn, which hipcc expands to a ~10-instruction IEEE division:
  // the 1e-7 relative error in t is below the 1.5e-7 absolute error of the 7.1.26 polynomial itself
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  const float e = __expf(-u * u);
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = fmaf(-p * t, e, 1.0f);
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}

// bf16 mode: a cheaper erf-GELU.  The outputs are rounded to bf16 (relative step 2^-8), so 1.5e-7 of absolute accuracy costs VALU
// that nothing downstream can see -- and the kernels that evaluate GELU per element (conv0 apply, the K = 128 conv epilogue, the
// teacher's GELU pass) are VALU-bound on it, not HBM-bound.  Form:  gelu(x) ~= x * sigmoid(w(x)),  w = x (a + b x^2 + c x^4)
// with (a, b, c) a minimax fit to the exact erf form over |x| <= 9: |error| <= 2.6e-5 absolute (the common tanh form, same
// structure with c = 0, is off by 4.7e-4), derivative error <= 1.1e-4.  |x| is clamped to 8 inside w only (the quartic term turns
// over beyond |x| ~ 10; sigmoid(w(8)) is 1 to 1e-12).  9 VALU issues, two of them transcendental, against 16 / 2 for the exact form.
// The fp32 parity mode keeps the exact form everywhere.
#define FH_GELU_A 1.59501577f
#define FH_GELU_B 7.40112923e-02f
#define FH_GELU_C (-7.03033628e-04f)
__device__ __forceinline__ float gelu_fast_sig(float x, float& x2c) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  x2c = xc * xc;
  // sigmoid(w) = 1 / (1 + 2^(-w log2 e)), with -log2 e folded into the polynomial's coefficients (one multiply less per element);
  // the raw v_exp_f32 / v_rcp_f32 (no denormal fix-up, 1 ulp)
  constexpr float L2E = -1.4426950408889634f;
  const float pw = fmaf(fmaf(FH_GELU_C * L2E, x2c, FH_GELU_B * L2E), x2c, FH_GELU_A * L2E);
  const float e = __builtin_amdgcn_exp2f(xc * pw);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_fast_f(float x) {
  float x2;
  return x * gelu_fast_sig(x, x2);
}
__device__ __forceinline__ float gelu_fast_grad_f(float x) {
  float x2;
  const float s = gelu_fast_sig(x, x2);
  // d/dx [x s(w)] = s + x s (1 - s) w'(x),  w' = a + 3 b x^2 + 5 c x^4   (in the clamped range s (1 - s) is 0 to fp32)
  const float wp = fmaf(fmaf(5.0f * FH_GELU_C, x2, 3.0f * FH_GELU_B), x2, FH_GELU_A);
  return fmaf(x * s * (1.0f - s), wp, s);
}
// element-type dispatch: exact for the fp32 parity mode, fast for bf16
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
  if constexpr (ElemTraits<T>::DT == 0) return gelu_f(x); else return gelu_fast_f(x);
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) {
  if constexpr (ElemTraits<T>::DT == 0) return gelu_grad_f(x); else return gelu_fast_grad_f(x);
}

// activation and derivative together (the forward of a layer that saves GELU' instead of its pre-activation: ep.aux_factor)
template <typename T> __device__ __forceinline__ void gelu_both_t(float x, float& y, float& g) {
  if constexpr (ElemTraits<T>::DT == 0) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    y = x * cdf;
    g = fmaf(x, pdf, cdf);
  } else {
    float x2;
    const float s = gelu_fast_sig(x, x2);
    const float wp = fmaf(fmaf(5.0f * FH_GELU_C, x2, 3.0f * FH_GELU_B), x2, FH_GELU_A);
    y = x * s;
    g = fmaf(y * (1.0f - s), wp, s);
  }
}

// ---------------------------------------------------------------- counter-based dropout
// mask(seed, site, idx) is a pure function, so backward regenerates the forward's mask.
__device__ __forceinline__ uint32_t fh_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
typedef fh_drop DropCfg;  // {p, seed, site}
// Split form: the per-site key (depends on seed, site and the HIGH index word only) is hoisted out of inner loops; per
// pair of elements it is one 32-bit mix + two integer compares (drop_apply / drop_pair below).
struct DropKey {
  uint32_t k;
  uint32_t thresh;
  float inv_keep;
};
__device__ __forceinline__ DropKey drop_key(const DropCfg& d, uint32_t idx_hi) {
  DropKey r;
  r.k = fh_mix32(d.seed * 0x9E3779B9u + d.site * 0x85ebca6bu + idx_hi * 0xc2b2ae35u + 0x27d4eb2fu);
  r.thresh = (uint32_t)(d.p * 16777216.0f);
  r.inv_keep = 1.0f / (1.0f - d.p);
  return r;
}
// Pair form: ONE hash word serves the two adjacent indices 2j and 2j + 1, 16 random bits each (keep <=> 16 bits >= p * 2^16).
// Every mask in the library is this function of (seed, site, index); vector callers hash once per pair.
__device__ __forceinline__ float drop_apply(const DropKey& k, uint32_t idx_lo) {
  const uint32_t h = fh_mix32((idx_lo & ~1u) ^ k.k);
  const uint32_t u = (idx_lo & 1u) ? (h >> 16) : (h & 0xffffu);
  return u >= (k.thresh >> 8) ? k.inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_scale(const DropCfg& d, unsigned long long idx) {
  return drop_apply(drop_key(d, (uint32_t)(idx >> 32)), (uint32_t)idx);
}
// Keys for an index RANGE of at most 2^32 elements (it touches at most two values of the high word): derived once per
// workgroup, then per element a compare + select instead of a hash and a division.
struct DropKeySpan {
  DropKey k;       // thresh / inv_keep shared; k.k = key of the range's first high word
  uint32_t k_next; // key of the following high word
  uint32_t hi0;
};
__device__ __forceinline__ DropKeySpan drop_key_span(const DropCfg& d, unsigned long long first_idx) {
  DropKeySpan s;
  s.hi0 = (uint32_t)(first_idx >> 32);
  s.k = drop_key(d, s.hi0);
  s.k_next = drop_key(d, s.hi0 + 1u).k;
  return s;
}
__device__ __forceinline__ DropKey drop_key_at(const DropKeySpan& s, unsigned long long idx) {
  DropKey k = s.k;
  k.k = (uint32_t)(idx >> 32) == s.hi0 ? s.k.k : s.k_next;
  return k;
}
// `idx_lo_even` must be even; s0 / s1 are the scales of idx_lo_even and idx_lo_even + 1.
__device__ __forceinline__ void drop_pair(const DropKey& k, uint32_t idx_lo_even, float& s0, float& s1) {
  const uint32_t h = fh_mix32(idx_lo_even ^ k.k);
  const uint32_t t16 = k.thresh >> 8;
  s0 = (h & 0xffffu) >= t16 ? k.inv_keep : 0.0f;
  s1 = (h >> 16) >= t16 ? k.inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_pair_one(const DropKey& k, uint32_t idx_lo) { return drop_apply(k, idx_lo); }
// N consecutive elements starting at an index that is a multiple of N (N even, <= 8): one key for all of them -- the low word
// cannot carry into the high word inside an aligned group -- and one hash per two elements.
template <int N> __device__ __forceinline__ void drop_scale_vec(const DropCfg& d, unsigned long long idx, float* v) {
  static_assert(N % 2 == 0, "pair form");
  const DropKey k = drop_key(d, (uint32_t)(idx >> 32));
  const uint32_t lo = (uint32_t)idx;
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    float s0, s1;
    drop_pair(k, lo + e, s0, s1);
    v[e] *= s0;
    v[e + 1] *= s1;
  }
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
