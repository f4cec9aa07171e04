// Fragment / tile helpers shared by the attention kernels (attention.hip) and the MFMA relation-transfer kernels (relation.hip):
// element traits of the LDS tiles, fragment loads from global rows and LDS tiles, the MFMA wrappers, the "accumulator tile as the next
// product's B operand" contraction (mma_transposed), tile staging, row exchanges over the four 16-lane rows of a wave.
#pragma once
#include "common.cuh"

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T> struct AT {           // 16-bit element types (bf16 / f16)
  static constexpr int DP = 64;             // padded head dim in LDS
  static constexpr int RS = DP * 2 + 16;    // LDS row stride (bytes)
  static constexpr int NK = 2;              // fragments per row over d (32 d-values each)
  static constexpr int KD = 32;
  typedef bf16x8 frag;
};
template <> struct AT<float> {
  static constexpr int DP = 64;
  static constexpr int RS = DP * 4 + 16;
  static constexpr int NK = 4;              // 16 d-values each (4 MFMA steps)
  static constexpr int KD = 16;
  typedef f32x4 frag;
};

__device__ __forceinline__ bf16x8 zero_frag(bf16x8*) {
  union { V16 v; bf16x8 f; } u;
  u.v = v16_zero();
  return u.f;
}
__device__ __forceinline__ f32x4 zero_frag(f32x4*) { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// fragment of a global row (d contiguous): d = KD*ks + VEC*(lane>>4) + e
template <typename T>
__device__ __forceinline__ typename AT<T>::frag gfrag(const T* row, bool valid, int ks, int lane, int dh) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int d0 = AT<T>::KD * ks + VEC * (lane >> 4);
  typename AT<T>::frag f = zero_frag((typename AT<T>::frag*)nullptr);
  if (valid && d0 < dh) {
    union { V16 v; typename AT<T>::frag f; } u;
    u.v = ld16(row + d0);
    f = u.f;
  }
  return f;
}
// fragment of an LDS tile row (d contiguous)
template <typename T>
__device__ __forceinline__ typename AT<T>::frag lfrag(const unsigned char* lds, int row, int ks, int lane) {
  return *reinterpret_cast<const typename AT<T>::frag*>(lds + row * AT<T>::RS + ks * 64 + (lane >> 4) * 16);
}
template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return mfma16<T>(a, b, c); }
template <typename T> __device__ __forceinline__ f32x4 mma(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
  return c;
}

// acc[dt] += Tile^T[d-tile dt][rows] * P   where `tile` is an LDS tile [64 rows][d], P0/P1 are two accumulator tiles
// covering tile rows [32*m2, 32*m2+16) and [32*m2+16, 32*m2+32) in C layout (row = 4*(lane>>4)+e, col = lane&15).
template <typename T, int NDT>
__device__ __forceinline__ void mma_transposed(const unsigned char* tile, int m2, const f32x4& P0, const f32x4& P1,
                                               f32x4 (&acc)[NDT], int lane) {
  const int i = lane & 15, q = lane >> 4;
  if constexpr (sizeof(T) == 2) {
    union { bf16x8 v; T e[8]; } pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pb.e[e] = (T)P0[e];
      pb.e[4 + e] = (T)P1[e];
    }
    const int qq = i >> 2, pp = i & 3;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const unsigned char* a0 = tile + (32 * m2 + 4 * q + qq) * AT<bf16>::RS + (16 * dt + 4 * pp) * 2;
      union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
      u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
      u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * AT<bf16>::RS));
      acc[dt] = mfma16<T>(u.v, pb.v, acc[dt]);
    }
  } else {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4& P = half ? P1 : P0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 32 * m2 + 16 * half + 4 * q + e;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const float a = *reinterpret_cast<const float*>(tile + row * AT<float>::RS + (16 * dt + i) * 4);
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, P[e], acc[dt], 0, 0, 0);
        }
      }
    }
  }
}

// stage 64 rows [row0, row0+64) of one head slice (column offset col) of a [B*T][ld] tensor into an LDS tile
template <typename T>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const T* base, long ld, int row0, int nrows_valid, int dh) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int cpr = dh / VEC;
  for (int idx = threadIdx.x; idx < 64 * cpr; idx += blockDim.x) {
    const int r = idx / cpr, c = idx - r * cpr;
    V16 v = v16_zero();
    if (row0 + r < nrows_valid) v = ld16(base + (long)(row0 + r) * ld + c * VEC);
    st16(lds + r * AT<T>::RS + c * 16, v);
  }
}
// Several 64-row tiles at once: every thread first issues ALL its 16-byte global loads (up to 4 per pass), then stores them.
// One tile per stage_tile call meant one exposed global-load round trip per tile -- staged that way, the teacher's forward (T = 499:
// 8 key + 8 value tiles) spent 34 K of its 87 K cycles per workgroup in 16 serialised round trips (s_memtime stamps).  The
// backward kernels stage their resident tiles through this.
// place(t, src, ld, dst, row0): tile t's source rows start at src + row0 * ld (head slice, d contiguous) and land at dst.
template <typename T, typename F>
__device__ __forceinline__ void stage_tiles(int ntiles, int nrows_valid, int dh, F&& place) {
  constexpr int VEC = ElemTraits<T>::VEC, BATCH = 4;
  const int cpr = dh / VEC, per_tile = 64 * cpr, total = ntiles * per_tile;
  const float inv_pt = 1.0f / (float)per_tile, inv_cpr = 1.0f / (float)cpr;  // exact for these ranges: (i + 0.5) / n is >= 0.5 / n off an integer
  for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * BATCH) {
    V16 v[BATCH];
    unsigned char* d[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int idx = i0 + j * (int)blockDim.x;
      d[j] = nullptr;
      v[j] = v16_zero();
      if (idx < total) {
        const int t = (int)(((float)idx + 0.5f) * inv_pt), rem = idx - t * per_tile;
        const int r = (int)(((float)rem + 0.5f) * inv_cpr), c = rem - r * cpr;
        const T* src;
        long sld;
        unsigned char* dst;
        int row0;
        place(t, src, sld, dst, row0);
        d[j] = dst + r * AT<T>::RS + c * 16;
        if (row0 + r < nrows_valid) v[j] = ld16(src + (long)(row0 + r) * sld + c * VEC);
      }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j)
      if (d[j]) st16(d[j], v[j]);
  }
}
template <typename T> __device__ __forceinline__ void zero_lds(unsigned char* lds, int bytes) {
  for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16) st16(lds + i, v16_zero());
}

// Dropout index of probability (b, h, query, key) = ((b H + h) Tn + query) * pitch + key with an EVEN row pitch: the pair form
// of the mask (common.cuh drop_pair: one hash word serves indices 2j and 2j + 1) needs every row to start on an even index,
// otherwise the forward's (row base + even offset) pairs and the dK/dV pass's (index & ~1) pairs disagree for odd Tn.
__device__ __forceinline__ unsigned long long drop_row_pitch(int Tn) { return (unsigned long long)((Tn + 1) & ~1); }
// v_exp_f32 without libm's denormal-range rescue (a compare, two selects and an ldexp per call): every exponent here is <= 0 up
// to rounding and a probability below 2^-126 may flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// butterfly over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 hold one query's partials).  gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows between two registers in ONE VALU issue: with both operands = v the pair
// comes back as (v of the even row | v of the odd row) in every lane of a row pair -- no ds_bpermute round trip through the LDS
// pipe (8 per 64-key tile and wave in the forward's softmax).
__device__ __forceinline__ void row_pairs16(float v, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void row_pairs32(float v, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor_max(float v) {
  float a, b;
  row_pairs16(v, a, b);
  v = fmaxf(a, b);
  row_pairs32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xor_sum(float v) {
  float a, b;
  row_pairs16(v, a, b);
  v = a + b;
  row_pairs32(v, a, b);
  return a + b;
}

}  // namespace
