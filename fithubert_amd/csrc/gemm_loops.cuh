// Device code and launchers of the generic MFMA GEMM (see gemm.h), shared by three translation units: gemm.hip (fp32 parity
// kernel, per-shape tuner, C entry points), gemm_bf16.hip and gemm_f16.hip (the 16-bit main loops instantiated for bf16 / IEEE half;
// same loops, same LDS images, v_mfma_f32_16x16x32_{bf16,f16} / 32x32x16 -- two units so that they compile side by side).
// 128x128 output tile, 4 waves (2x2), each wave a 64x64 sub-tile of 16x16 MFMA blocks; register-prefetched global->LDS staging;
// fp32 accumulate -- and the 256-tile LDS-DMA loops below.
//
//   T = bf16 / f16 : v_mfma_f32_16x16x32_{bf16,f16}, BK = 64
//   T = float      : v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), BK = 32
//
// MFMA orientation: the Y operand (output column j) is the MFMA "A" matrix and the X operand
// (output row i) the MFMA "B" matrix, so a lane's 4 accumulator registers are 4 CONSECUTIVE j
// of one output row -> 8/16-byte vector stores and vector bias/residual loads in the epilogue.
#pragma once
#include <cstdlib>
#include <unordered_map>
#include <utility>
#include <vector>

#include "gemm.h"

// host state shared by the translation units (defined in gemm.hip)
extern int fh_g_reserve_cus, fh_g_last_loop, fh_g_last_splitk;
int fh_stream_reserve_cus(hipStream_t st);   // CUs loop 8 leaves free on launches of this stream (gemm.hip: fh_gemm_stream_reserve_cus)

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int LDS_C_STRIDE = 144;  // R_CONTIG tile: 128 rows x (128 B of k + 16 B pad)
constexpr int TILE_BYTES = 18432;  // max over layouts

template <typename T> struct StridedCfg;
template <> struct StridedCfg<bf16> { static constexpr int STRIDE = 288; };   // 128 el * 2 B + 32
template <> struct StridedCfg<f16> { static constexpr int STRIDE = 288; };
template <> struct StridedCfg<float> { static constexpr int STRIDE = 528; };  // 128 el * 4 B + 16

struct RowMap {
  int b, t;
};
// segment 1 of an operand with a seg1_limit reads zeros at indices beyond it
__device__ __forceinline__ bool seg_live(const GemmOperand& o, int seg, int idx) {
  return seg == 0 || o.seg1_limit <= 0 || idx < o.seg1_limit;
}
// reduction segments of the output tile whose first column is j0 (tiles entirely beyond Y's seg1_limit have one)
__device__ __forceinline__ int tile_nseg(const GemmParams& p, int j0) {
  return (p.X.nseg == 2 && p.Y.seg1_limit > 0 && j0 >= p.Y.seg1_limit) ? 1 : p.X.nseg;
}
__device__ __forceinline__ RowMap map_row(int idx, int rpb) {
  RowMap r;
  if (rpb > 0) {
    r.b = idx / rpb;
    r.t = idx - r.b * rpb;
  } else {
    r.b = 0;
    r.t = idx;
  }
  return r;
}

// ------------------------------------------------------------------ global -> register -> LDS tile loaders
template <typename T, int LAYOUT> struct Loader;

template <typename T> struct Loader<T, FH_R_CONTIG> {
  static constexpr int VEC = ElemTraits<T>::VEC;
  const T* base;
  long off[4];
  long segdelta;
  unsigned vmask;  // bit (2*u + seg)
  int kc;
  V16 regs[4];

  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const T*>(o.p) + (long)g * o.gstride;
    kc = tid & 7;
    vmask = 0;
    segdelta = (long)(o.tshift[1] - o.tshift[0]) * o.ld + (o.segoff[1] - o.segoff[0]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + (tid >> 3) + 32 * u;
      const bool rv = idx < idx_limit;
      RowMap rm = map_row(rv ? idx : 0, o.rpb);
      off[u] = (long)rm.b * o.bstride + (long)(rm.t + o.tshift[0]) * o.ld + o.segoff[0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ts = rm.t + o.tshift[s];
        const bool v = rv && (s < o.nseg) && (o.tvalid == 0 || (ts >= 0 && ts < o.tvalid)) && seg_live(o, s, idx);
        vmask |= (v ? 1u : 0u) << (2 * u + s);
      }
    }
  }
  __device__ __forceinline__ void load(const GemmOperand& o, int seg, int k0, int Kseg) {
    const int k = k0 + kc * VEC;
    const bool kv = k < Kseg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (kv && ((vmask >> (2 * u + seg)) & 1u))
        regs[u] = ld16(base + off[u] + (long)seg * segdelta + k);
      else
        regs[u] = v16_zero();
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) st16(lds + ((tid >> 3) + 32 * u) * LDS_C_STRIDE + kc * 16, regs[u]);
  }
};

template <typename T> struct Loader<T, FH_R_STRIDED> {
  static constexpr int VEC = ElemTraits<T>::VEC;
  static constexpr int CPR = 128 / VEC;    // 16-byte chunks per tile row
  static constexpr int RSTEP = 256 / CPR;  // tile rows covered by one pass of the block
  static constexpr int STRIDE = StridedCfg<T>::STRIDE;
  const T* base;
  int col;
  bool colvalid;
  int rbase, ic;
  V16 regs[4];

  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const T*>(o.p) + (long)g * o.gstride;
    ic = tid & (CPR - 1);
    rbase = tid / CPR;
    col = idx0 + ic * VEC;
    colvalid = col < idx_limit;
  }
  __device__ __forceinline__ void load(const GemmOperand& o, int seg, int k0, int Kseg) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = k0 + rbase + u * RSTEP;
      if (colvalid && r < Kseg && seg_live(o, seg, col)) {
        RowMap rm = map_row(r, o.rpb);
        regs[u] = ld16(base + o.segoff[seg] + (long)rm.b * o.bstride + (long)rm.t * o.ld + col);
      } else {
        regs[u] = v16_zero();
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) st16(lds + (rbase + u * RSTEP) * STRIDE + ic * 16, regs[u]);
  }
  __device__ __forceinline__ void store_stride(unsigned char* lds, int stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) st16(lds + (rbase + u * RSTEP) * stride + ic * 16, regs[u]);
  }
};

// ------------------------------------------------------------------ LDS -> MFMA fragments
// bf16: 8 k-values per lane per k32-step, k = ks*32 + 8*(lane>>4) + e
template <int LAYOUT> __device__ __forceinline__ bf16x8 frag_bf16(const unsigned char* lds, int row0, int ks, int lane);
template <> __device__ __forceinline__ bf16x8 frag_bf16<FH_R_CONTIG>(const unsigned char* lds, int row0, int ks, int lane) {
  // k-order inside a k32 step (shared by both layouts, any order works as long as X and Y agree):
  //   fragment element e < 4 -> k = 4q + e ; e >= 4 -> k = 16 + 4q + (e - 4).
  // With this order the transposing reads of an R_STRIDED tile touch rows {4q+qq} and {16+4q+qq}: the two 16-lane
  // groups of a half-wave read 8 DIFFERENT consecutive rows -> conflict-free (rows 8 apart alias to the same banks).
  const unsigned char* a = lds + (row0 + (lane & 15)) * LDS_C_STRIDE + ks * 64 + (lane >> 4) * 8;
  union {
    struct { uint2 lo, hi; } s;
    bf16x8 v;
  } u;
  u.s.lo = *reinterpret_cast<const uint2*>(a);
  u.s.hi = *reinterpret_cast<const uint2*>(a + 32);
  return u.v;
}
template <> __device__ __forceinline__ bf16x8 frag_bf16<FH_R_STRIDED>(const unsigned char* lds, int row0, int ks, int lane) {
  // ds_read_b64_tr_b16: within a 16-lane group, lane 4*qq+p supplies the address of row qq, columns 4p..4p+3
  // of a 4x16 block; lane i receives column i of the 4 rows.
  constexpr int STRIDE = StridedCfg<bf16>::STRIDE;
  const int i = lane & 15, q = lane >> 4;
  const int qq = i >> 2, pp = i & 3;
  const unsigned char* a0 = lds + (ks * 32 + 4 * q + qq) * STRIDE + (row0 + 4 * pp) * 2;
  const unsigned char* a1 = a0 + 16 * STRIDE;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
  union {
    struct { s16x4 lo, hi; } s;
    bf16x8 v;
  } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}
// float: 4 k-values per lane per k16-round, k = kr*16 + 4*(lane>>4) + e ; MFMA step e uses element e
template <int LAYOUT> __device__ __forceinline__ f32x4 frag_f32(const unsigned char* lds, int row0, int kr, int lane);
template <> __device__ __forceinline__ f32x4 frag_f32<FH_R_CONTIG>(const unsigned char* lds, int row0, int kr, int lane) {
  const unsigned char* a = lds + (row0 + (lane & 15)) * LDS_C_STRIDE + kr * 64 + (lane >> 4) * 16;
  return *reinterpret_cast<const f32x4*>(a);
}
template <> __device__ __forceinline__ f32x4 frag_f32<FH_R_STRIDED>(const unsigned char* lds, int row0, int kr, int lane) {
  constexpr int STRIDE = StridedCfg<float>::STRIDE;
  const unsigned char* a = lds + (kr * 16 + 4 * (lane >> 4)) * STRIDE + (row0 + (lane & 15)) * 4;
  f32x4 v;
  v[0] = *reinterpret_cast<const float*>(a);
  v[1] = *reinterpret_cast<const float*>(a + STRIDE);
  v[2] = *reinterpret_cast<const float*>(a + 2 * STRIDE);
  v[3] = *reinterpret_cast<const float*>(a + 3 * STRIDE);
  return v;
}

template <typename T, int XL, int YL> struct TileMath {  // 16-bit element types
  static __device__ __forceinline__ void run(const unsigned char* ldsX, const unsigned char* ldsY, int xi0, int yj0,
                                             int lane, f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fx[4], fy[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) fx[m] = frag_bf16<XL>(ldsX, xi0 + 16 * m, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n) fy[n] = frag_bf16<YL>(ldsY, yj0 + 16 * n, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[n][m] = mfma16<T>(fy[n], fx[m], acc[n][m]);
    }
  }
};
template <int XL, int YL> struct TileMath<float, XL, YL> {
  static __device__ __forceinline__ void run(const unsigned char* ldsX, const unsigned char* ldsY, int xi0, int yj0,
                                             int lane, f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int kr = 0; kr < 2; ++kr) {
      f32x4 fx[4], fy[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) fx[m] = frag_f32<XL>(ldsX, xi0 + 16 * m, kr, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n) fy[n] = frag_f32<YL>(ldsY, yj0 + 16 * n, kr, lane);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[n][e], fx[m][e], acc[n][m], 0, 0, 0);
    }
  }
};

// ------------------------------------------------------------------ KD-loss epilogue (fh_gemm_epilogue.kd_*)
__device__ __forceinline__ void kd_load8(const GemmEpilogue& ep, long off, float* t) {
  if (ep.kd_tdtype == 0) {
    const float* q = reinterpret_cast<const float*>(ep.kd_teacher) + off;
    ld4(q, t);
    ld4(q + 4, t + 4);
  } else if (ep.kd_tdtype == 2) {
    unpack16<f16>(ld16(reinterpret_cast<const f16*>(ep.kd_teacher) + off), t);
  } else {
    unpack16<bf16>(ld16(reinterpret_cast<const bf16*>(ep.kd_teacher) + off), t);
  }
}
// teacher element offset of output (group g, row i, column c); c a multiple of 8, so its 8 columns share one sub-frame
__device__ __forceinline__ long kd_offset(const GemmParams& p, int g, int i, int c) {
  const GemmEpilogue& ep = p.ep;
  const int b = i / p.c_rpb, t = i - b * p.c_rpb, j = c / ep.kd_P, f = p.N / ep.kd_P;
  return (long)ep.kd_layer[g] * ep.kd_lstride + ((long)b * ep.kd_Tt + (long)f * t + j) * ep.kd_P + (c - j * ep.kd_P);
}
struct KdCoef {
  float g;     // d loss / d v per unit of (v - t) (mse) or of sign(v - t) (l1); the arithmetic of kd_loss_kernel, term by term
  bool live;   // the group's weight is not 0
};
__device__ __forceinline__ KdCoef kd_coef(const GemmEpilogue& ep, int g) {
  const float w = ep.kd_w[g];
  KdCoef c;
  c.g = ep.kd_grad_scale * w / ep.kd_count * (ep.kd_type == 0 ? 2.0f : 1.0f);
  c.live = w != 0.f;
  return c;
}
// a[8]: the prediction (already rounded to the output type) -> the gradient; returns this lane's share of the loss sum
__device__ __forceinline__ float kd_apply(const GemmEpilogue& ep, const KdCoef& kc, const float* t, float* a) {
  float acc = 0.f;
  if (kc.live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = a[e] - t[e];
      if (ep.kd_type == 0) {
        acc = fmaf(d, d, acc);
        a[e] = kc.g * d;
      } else {
        acc += fabsf(d);
        a[e] = d > 0.f ? kc.g : (d < 0.f ? -kc.g : 0.f);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
  }
  return acc;
}
// the loss sum of one 64-row x 64-column block (row0, col0 multiples of 64), summed over the wave that owns it
__device__ __forceinline__ void kd_write_slot(const GemmParams& p, int g, int row0, int col0, int lane, float v) {
  v = wave_sum(v);
  if (lane == 0 && row0 < p.M && col0 < p.N) {
    const int nrb = (p.M + 63) >> 6, ncb = (p.N + 63) >> 6;
    p.ep.kd_partials[((long)g * nrb + (row0 >> 6)) * ncb + (col0 >> 6)] = v;
  }
}

// ------------------------------------------------------------------ epilogue (shared by both main loops)
template <typename T, int NN = 4, int NM = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[NN][NM], int i0, int j0, int xi0, int yj0,
                                              int lane, int g) {
  const GemmEpilogue& ep = p.ep;
  const int jq = (lane >> 4) * 4;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int i = i0 + xi0 + 16 * m + (lane & 15);
    if (i >= p.M) continue;
    RowMap rm = map_row(i, p.c_rpb);
    const long crow = (long)g * p.c_gstride + (long)rm.b * p.c_bstride + (long)rm.t * p.ldc;
    bool zero_row = false;
    if (ep.row_len) zero_row = rm.t >= ep.row_len[rm.b];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const int j = j0 + yj0 + 16 * n + jq;
      if (j >= p.N) continue;
      if (ep.tail_cols > 0 && rm.t == p.c_rpb - 1 && j >= ep.tail_cols) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[n][m][e] * ep.alpha;
      if (ep.atomic) {
        float* Cf = reinterpret_cast<float*>(p.C) + (long)g * p.c_gstride;
        const long im = ep.idiv > 0 ? (long)(i / ep.idiv) * ep.si1 + (long)(i % ep.idiv) * ep.si0 : (long)i * p.ldc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int jj = j + e;
          const long jm = ep.jdiv > 0 ? (long)(jj / ep.jdiv) * ep.sj1 + (long)(jj % ep.jdiv) * ep.sj0 : (long)jj;
          atomicAdd(Cf + im + jm, v[e]);
        }
        continue;
      }
      if (ep.bias) {
        const float* bp = ep.bias + (long)g * ep.bias_gstride + (ep.bias_mod > 0 ? (j % ep.bias_mod) : j);
        float4 bv = *reinterpret_cast<const float4*>(bp);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      const long coff = crow + j;
      if (ep.preact && ep.aux_factor && ep.act == FH_ACT_GELU) {  // the saved tensor is GELU'(pre-activation)
        float gq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gelu_both_t<T>(v[e], v[e], gq[e]);
        st4(reinterpret_cast<T*>(ep.preact) + coff, gq);
      } else {
        if (ep.preact) st4(reinterpret_cast<T*>(ep.preact) + coff, v);
        if (ep.act == FH_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_t<T>(v[e]);
        }
      }
      if (ep.drop.p > 0.f) {
        const unsigned long long didx = ((unsigned long long)g * p.M + i) * (unsigned long long)p.N + j;
        drop_scale_vec<4>(ep.drop, didx, v);
      }
      if (ep.gelu_aux) {
        float a[4];
        ld4(reinterpret_cast<const T*>(ep.gelu_aux) + coff, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= ep.aux_factor ? a[e] : gelu_grad_t<T>(a[e]);
      }
      if (ep.residual) {
        float r[4];
        ld4(reinterpret_cast<const T*>(ep.residual) + coff, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      if (zero_row) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.f;
      }
      if (ep.out_f32)
        st4(reinterpret_cast<float*>(p.C) + coff, v);
      else
        st4(reinterpret_cast<T*>(p.C) + coff, v);
    }
  }
}

// ------------------------------------------------------------------ LDS-staged epilogue (non-atomic outputs)
// The MFMA C layout gives a lane 4 consecutive columns of one row, so direct stores touch 16 rows x 32 B per wave
// instruction: on the HBM-bound conv layers the L2 drowned in partial-line requests.  Instead each wave parks its 64x64
// fp32 sub-tile in LDS (two passes of 32 rows, 272-B padded rows) and re-reads it row-major, 8 consecutive columns per
// lane: bias / residual / GELU'-operand loads and every store are 16-byte accesses, 8 rows x 128 contiguous bytes per
// instruction.  All arithmetic stays fp32 (no extra rounding).
constexpr int EPI_ROW = 272;             // 64 fp32 columns + 16 B pad
constexpr int EPI_WAVE = 32 * EPI_ROW;   // 8704 B per wave per pass

template <typename T> __device__ __forceinline__ void ld8(const T* p, float* f);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* f) { ld4(p, f); ld4(p + 4, f + 4); }
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float* f) { unpack16<bf16>(ld16(p), f); }
template <> __device__ __forceinline__ void ld8<f16>(const f16* p, float* f) { unpack16<f16>(ld16(p), f); }
// 8 elements as loaded (conversion deferred to the point of use): one 16-byte vector of bf16, two of fp32
template <typename T> struct Raw8 {  // 16-bit element types
  V16 v;
  __device__ __forceinline__ void load(const T* p) { v = ld16(p); }
  __device__ __forceinline__ void get(float* f) const { unpack16<T>(v, f); }
};
template <> struct Raw8<float> {
  V16 v[2];
  __device__ __forceinline__ void load(const float* p) { v[0] = ld16(p); v[1] = ld16(p + 4); }
  __device__ __forceinline__ void get(float* f) const { unpack16<float>(v[0], f); unpack16<float>(v[1], f + 4); }
};
template <typename T> __device__ __forceinline__ void st8(T* p, const float* f);
template <> __device__ __forceinline__ void st8<float>(float* p, const float* f) { st4(p, f); st4(p + 4, f + 4); }
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float* f) { st16(p, pack16<bf16>(f)); }
template <> __device__ __forceinline__ void st8<f16>(f16* p, const float* f) { st16(p, pack16<f16>(f)); }

// read-back phase of the staged epilogue: `my` holds 32 rows x 64 fp32 columns of this wave (pass h)
template <typename T, bool KD = false>
__device__ __forceinline__ void epilogue_readback(const GemmParams& p, const unsigned char* my, int h, int i0, int j0, int xi0,
                                                  int yj0, int lane, int g, const float (&bv)[8], float* kd_acc = nullptr) {
  const GemmEpilogue& ep = p.ep;
  if (ep.atomic) {
    // weight gradients: one output row x 64 contiguous columns per wave instruction (256 B, the shape fp32 atomics like)
    float* Cf = reinterpret_cast<float*>(p.C) + (long)g * p.c_gstride;
    const int j = j0 + yj0 + lane;
    if (j < p.N) {
      const long jm = ep.jdiv > 0 ? (long)(j / ep.jdiv) * ep.sj1 + (long)(j % ep.jdiv) * ep.sj0 : (long)j;
      for (int rl = 0; rl < 32; ++rl) {
        const int i = i0 + xi0 + 32 * h + rl;
        if (i >= p.M) break;
        const long im = ep.idiv > 0 ? (long)(i / ep.idiv) * ep.si1 + (long)(i % ep.idiv) * ep.si0 : (long)i * p.ldc;
        atomicAdd(Cf + im + jm, *reinterpret_cast<const float*>(my + rl * EPI_ROW + lane * 4) * ep.alpha);
      }
    }
    return;
  }
  // A lane keeps ONE 8-column group j for all its rows: the column bound and the index bases are hoisted out of the row loop, and
  // the row -> (batch, frame) map, the output offset and the dropout index advance incrementally.
  // NO vector-memory LOAD may sit between the stores of a tile: vmcnt counts loads and stores in one in-order queue, so waiting
  // for any load also waits for every older store to be acknowledged (~2.5 us under load) -- with the bias loaded inside each
  // pass, a tile's four passes serialised behind each other's stores and the epilogue cost 10 us of a 256 x 256 tile.  The bias
  // therefore comes in registers (epilogue_bias, before the first pass); passes that do need per-row operands (GELU' operand,
  // residual, row lengths) issue all of them before their first store.
  const int j = j0 + yj0 + 8 * (lane & 7);
  if (j >= p.N) return;
  const int i_first = i0 + xi0 + 32 * h + (lane >> 3);
  if (i_first >= p.M) return;
  // Streamlined form for the common case -- rows contiguous in memory (no batch wrap), no row mask, bf16 / fp32 output as T, all
  // four rows of the lane inside M: per row 2 LDS reads, the option math, the conversion and the store(s), with the output /
  // operand offsets as base + constant stride.  The general form below executes ~150 instructions per row (option tests, 64-bit
  // index updates, row wrap), which made a pass instruction-bound (~3000 cycles per 32 x 64 block, measured with s_memtime)
  // rather than LDS-bound.
  const long wrap0 = p.c_rpb > 0 ? p.c_bstride - (long)p.c_rpb * p.ldc : 0;
  // (batch entries of fewer than 8 rows -- more than one wrap per row step -- stay with the general form)
  const bool simple = !ep.out_f32 && (p.c_rpb <= 0 || p.c_rpb >= 8) && !KD;   // (KD: the general form below carries the loss arithmetic)
  if (simple && __builtin_amdgcn_readfirstlane(i0 + xi0 + 32 * h + 31 < p.M)) {
    // offsets of the lane's four rows: constant stride, plus the batch-entry correction when a row step crosses into the next
    // entry (the conv data gradients write every other frame of a longer sequence: batch stride != rows x ldc)
    long co[4];
    bool dead[4] = {false, false, false, false};
    bool skip[4] = {false, false, false, false};
    {
      int rb = 0, rt = i_first;
      if (p.c_rpb > 0 && (wrap0 != 0 || ep.row_len || ep.tail_cols > 0)) {
        rb = i_first / p.c_rpb;
        rt = i_first - rb * p.c_rpb;
      }
      long o = (long)g * p.c_gstride + (long)i_first * p.ldc + (long)rb * wrap0 + j;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        co[it] = o;
        if (ep.tail_cols > 0) skip[it] = rt == p.c_rpb - 1 && j >= ep.tail_cols;  // the item's last row is narrower
        if (ep.row_len) dead[it] = rt >= ep.row_len[rb];  // rows past an utterance's valid frames are written as zeros
        o += 8L * p.ldc;
        rt += 8;
        if (p.c_rpb > 0 && rt >= p.c_rpb) {
          rt -= p.c_rpb;
          ++rb;
          o += wrap0;
        }
      }
    }
    const unsigned char* lp = my + (lane >> 3) * EPI_ROW + (lane & 7) * 32;
    const bool gelu = ep.act == FH_ACT_GELU, drop = ep.drop.p > 0.f;
    Raw8<T> aux[4], res[4];
    if (ep.gelu_aux) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (!skip[it]) aux[it].load(reinterpret_cast<const T*>(ep.gelu_aux) + co[it]);
    }
    if (ep.residual) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (!skip[it]) res[it].load(reinterpret_cast<const T*>(ep.residual) + co[it]);
    }
    if (__builtin_amdgcn_readfirstlane(ep.gelu_aux && ep.aux_factor && !ep.preact && !gelu && !drop && !ep.residual && !ep.row_len)) {
      // the conv data gradients (product x saved GELU'): their own straight-line instance of the row loop, without the option tests
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (skip[it]) continue;
        const f32x4 a = *reinterpret_cast<const f32x4*>(lp + it * 8 * EPI_ROW);
        const f32x4 b = *reinterpret_cast<const f32x4*>(lp + it * 8 * EPI_ROW + 16);
        float v[8], x[8];
        aux[it].get(x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fmaf(a[e], ep.alpha, bv[e]) * x[e];
          v[4 + e] = fmaf(b[e], ep.alpha, bv[4 + e]) * x[4 + e];
        }
        st8<T>(reinterpret_cast<T*>(p.C) + co[it], v);
      }
      return;
    }
    const unsigned long long d0 = ((unsigned long long)g * p.M + i_first) * (unsigned long long)p.N + j;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (skip[it]) continue;
      const f32x4 a = *reinterpret_cast<const f32x4*>(lp + it * 8 * EPI_ROW);
      const f32x4 b = *reinterpret_cast<const f32x4*>(lp + it * 8 * EPI_ROW + 16);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaf(a[e], ep.alpha, bv[e]);
        v[4 + e] = fmaf(b[e], ep.alpha, bv[4 + e]);
      }
      if (ep.preact && ep.aux_factor && gelu) {  // the saved tensor is GELU'(pre-activation)
        float gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_both_t<T>(v[e], v[e], gq[e]);
        st8<T>(reinterpret_cast<T*>(ep.preact) + co[it], gq);
      } else {
        if (ep.preact) st8<T>(reinterpret_cast<T*>(ep.preact) + co[it], v);
        if (gelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
        }
      }
      if (drop) drop_scale_vec<8>(ep.drop, d0 + (unsigned long long)(it * 8) * (unsigned long long)p.N, v);
      if (ep.gelu_aux) {
        float x[8];
        aux[it].get(x);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= ep.aux_factor ? x[e] : gelu_grad_t<T>(x[e]);
      }
      if (ep.residual) {
        float x[8];
        res[it].get(x);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += x[e];
      }
      if (dead[it]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      st8<T>(reinterpret_cast<T*>(p.C) + co[it], v);
    }
    return;
  }
  RowMap rm = map_row(i_first, p.c_rpb);
  long coff = (long)g * p.c_gstride + (long)rm.b * p.c_bstride + (long)rm.t * p.ldc + j;
  unsigned long long didx = ((unsigned long long)g * p.M + i_first) * (unsigned long long)p.N + j;
  const long wrap = p.c_bstride - (long)p.c_rpb * p.ldc;  // offset correction when a row step crosses into the next batch entry
  long co[4];
  unsigned long long di[4];
  bool live[4], dead[4];
  Raw8<T> aux[4], res[4];
  const bool has_loads = ep.gelu_aux || ep.residual || ep.row_len;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    live[it] = i_first + it * 8 < p.M;
    dead[it] = false;
    if (it > 0) {
      rm.t += 8;
      coff += 8L * p.ldc;
      didx += 8ULL * (unsigned long long)p.N;
      if (p.c_rpb > 0)
        while (rm.t >= p.c_rpb) {
          rm.t -= p.c_rpb;
          ++rm.b;
          coff += wrap;
        }
    }
    co[it] = coff;
    di[it] = didx;
    if (ep.tail_cols > 0 && rm.t == p.c_rpb - 1 && j >= ep.tail_cols) live[it] = false;  // (rows are only ever skipped, never reordered)
    if (has_loads && live[it]) {
      if (ep.gelu_aux) aux[it].load(reinterpret_cast<const T*>(ep.gelu_aux) + coff);
      if (ep.residual) res[it].load(reinterpret_cast<const T*>(ep.residual) + coff);
      if (ep.row_len) dead[it] = rm.t >= ep.row_len[rm.b];
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (!live[it]) continue;
    const int rl = it * 8 + (lane >> 3);
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(my + rl * EPI_ROW + (lane & 7) * 32);
      const f32x4 b = *reinterpret_cast<const f32x4*>(my + rl * EPI_ROW + (lane & 7) * 32 + 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaf(a[e], ep.alpha, bv[e]);
        v[4 + e] = fmaf(b[e], ep.alpha, bv[4 + e]);
      }
    }
    if (ep.preact && ep.aux_factor && ep.act == FH_ACT_GELU) {
      float gq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gelu_both_t<T>(v[e], v[e], gq[e]);
      st8<T>(reinterpret_cast<T*>(ep.preact) + co[it], gq);
    } else {
      if (ep.preact) st8<T>(reinterpret_cast<T*>(ep.preact) + co[it], v);
      if (ep.act == FH_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
      }
    }
    if (ep.drop.p > 0.f) drop_scale_vec<8>(ep.drop, di[it], v);
    if (ep.gelu_aux) {
      float a[8];
      aux[it].get(a);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= ep.aux_factor ? a[e] : gelu_grad_t<T>(a[e]);
    }
    if (ep.residual) {
      float rr[8];
      res[it].get(rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rr[e];
    }
    if (dead[it]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if constexpr (KD) {   // v is the prediction: round it as the stored `pred` would be, compare with the teacher, write the gradient
      float tv[8];
      kd_load8(ep, kd_offset(p, g, i_first + it * 8, j), tv);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = to_f(from_f<T>(v[e]));
      const float part = kd_apply(ep, kd_coef(ep, g), tv, v);
      if (kd_acc) *kd_acc += part;
    }
    if (ep.out_f32)
      st8<float>(reinterpret_cast<float*>(p.C) + co[it], v);
    else
      st8<T>(reinterpret_cast<T*>(p.C) + co[it], v);
  }
}
// bias of the 8 columns a lane owns in epilogue_readback (zeros without a bias): loaded once per tile and column block, before
// the first pass
__device__ __forceinline__ void epilogue_bias(const GemmParams& p, int j0, int yj0, int lane, int g, float (&bv)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  const int j = j0 + yj0 + 8 * (lane & 7);
  if (p.ep.bias && !p.ep.atomic && j < p.N) {
    const float* bp = p.ep.bias + (long)g * p.ep.bias_gstride + (p.ep.bias_mod > 0 ? (j % p.ep.bias_mod) : j);
    ld4(bp, bv);
    ld4(bp + 4, bv + 4);
  }
}

template <typename T, bool KD = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams& p, const f32x4 (&acc)[4][4], unsigned char* lds, int i0,
                                                     int j0, int xi0, int yj0, int lane, int wave, int g) {
  unsigned char* my = lds + wave * EPI_WAVE;
  const int r = lane & 15, q = lane >> 4;
  float bv[8];
  float kd_acc = 0.f;
  epilogue_bias(p, j0, yj0, lane, g, bv);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();  // main-loop LDS reads (h = 0) / previous pass (h = 1) are finished in every wave
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        *reinterpret_cast<f32x4*>(my + (mm * 16 + r) * EPI_ROW + (n * 16 + 4 * q) * 4) = acc[n][2 * h + mm];
    __syncthreads();
    epilogue_readback<T, KD>(p, my, h, i0, j0, xi0, yj0, lane, g, bv, &kd_acc);
  }
  if constexpr (KD) kd_write_slot(p, g, i0 + xi0, j0 + yj0, lane, kd_acc);   // this wave's 64 x 64 block
}

// bf16-staged epilogue for outputs without per-row operands (no GELU' operand, residual or row mask; bf16 output): bias, GELU and
// dropout are applied in the accumulator layout (a lane holds 4 consecutive columns of a row), the result is converted ONCE and
// staged as bf16 -- half the LDS bytes and half the LDS instructions of the fp32 staging -- and the read-back is a plain copy
// (16-byte LDS read -> 16-byte store, 8 rows x 128 contiguous bytes per wave instruction).  A pre-activation output is staged and
// copied the same way before the activation.  `acc` is this wave's [4 n-tiles][NM m-tiles] block, column block yj0 .. yj0 + 63.
// Per-row OPERANDS (round 3): a residual and / or a GELU' operand (the saved factor, or a pre-activation to differentiate) are applied in
// the read-back -- one coalesced 16-byte load per 16-byte store, issued for the whole pass before its first store -- on the staged
// 16-bit value, i.e. the result is rounded twice (as the reference's own AMP arithmetic rounds a Linear's output before the residual
// add); the fp32 parity mode never comes here.  That took the K = 480 layer launches with a residual from 32.9 to ~24 us stand-alone.
// Returns false (nothing written) when the tile does not qualify; the caller then runs the fp32-staged form.
constexpr int EPB_ROW = 144;  // 64 bf16 columns + 16 B pad: conflict-free 8-byte writes (16 rows) and 16-byte reads (8 lanes / row)
template <typename H, int NM, bool KD = false, typename AccT = void>
__device__ __forceinline__ bool epilogue_bf16(const GemmParams& p, const AccT& acc, int n_base, unsigned char* my, int i0, int j0,
                                              int xi0, int yj0, int lane, int g) {
  const GemmEpilogue& ep = p.ep;
  const long wrap0 = p.c_bstride - (long)p.c_rpb * p.ldc;
  const bool ok = !ep.atomic && !ep.row_len && !ep.out_f32 && !ep.tail_cols && (p.c_rpb <= 0 || wrap0 == 0) &&
                  i0 + xi0 + 16 * NM - 1 < p.M;
  if (!__builtin_amdgcn_readfirstlane(ok)) return false;
  const bool has_ops = KD || __builtin_amdgcn_readfirstlane(ep.gelu_aux != nullptr || ep.residual != nullptr);
  float kd_acc[NM / 4 > 0 ? NM / 4 : 1];   // loss sums of this wave's 64-row blocks (KD-loss epilogue)
#pragma unroll
  for (int sblk = 0; sblk < (NM / 4 > 0 ? NM / 4 : 1); ++sblk) kd_acc[sblk] = 0.f;
  KdCoef kc{0.f, false};
  if constexpr (KD) kc = kd_coef(ep, g);
  const int r = lane & 15, q = lane >> 4;
  float b4[4][4];
  bool cv[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int j = j0 + yj0 + 16 * n + 4 * q;
    cv[n] = j < p.N;
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[n][e] = 0.f;
    if (ep.bias && cv[n]) ld4(ep.bias + (long)g * ep.bias_gstride + (ep.bias_mod > 0 ? (j % ep.bias_mod) : j), b4[n]);
  }
  const bool gelu = ep.act == FH_ACT_GELU, drop = ep.drop.p > 0.f;
  const bool both = gelu && ep.preact && ep.aux_factor;  // the saved tensor is GELU'(pre-activation)
  // read-back side: row it * 8 + (lane >> 3) of the pass, 8 columns at (lane & 7) * 8
  const int jr = j0 + yj0 + 8 * (lane & 7);
  const bool jrv = jr < p.N;
  const long rstep = 8L * p.ldc;
  const unsigned char* lp = my + (lane >> 3) * EPB_ROW + (lane & 7) * 16;
  unsigned char* wp = my + r * EPB_ROW + 8 * q;
#pragma unroll
  for (int h = 0; h < NM / 2; ++h) {
    const long base = (long)g * p.c_gstride + (long)(i0 + xi0 + 32 * h + (lane >> 3)) * p.ldc + jr;
    float v[2][4][4];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[mm][n][e] = fmaf(acc[n_base + n][2 * h + mm][e], ep.alpha, b4[n][e]);
    if (ep.preact) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          float sv[4];  // what the saved tensor receives: the pre-activation, or GELU' of it (then v becomes the activation here)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sv[e] = v[mm][n][e];
            if (both) gelu_both_t<H>(v[mm][n][e], v[mm][n][e], sv[e]);
          }
          uint2 pk;
          pk.x = pack2<H>(sv[0], sv[1]);
          pk.y = pack2<H>(sv[2], sv[3]);
          *reinterpret_cast<uint2*>(wp + mm * 16 * EPB_ROW + n * 32) = pk;
        }
      if (jrv) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          st16(reinterpret_cast<bf16*>(ep.preact) + base + it * rstep, ld16(lp + it * 8 * EPB_ROW));
      }
    }
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (gelu && !both) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[mm][n][e] = gelu_fast_f(v[mm][n][e]);
        }
        if (drop) {
          const unsigned long long didx = ((unsigned long long)g * p.M + (i0 + xi0 + 16 * (2 * h + mm) + r)) * (unsigned long long)p.N +
                                          (j0 + yj0 + 16 * n + 4 * q);
          drop_scale_vec<4>(ep.drop, didx, v[mm][n]);
        }
        uint2 pk;
        pk.x = pack2<H>(v[mm][n][0], v[mm][n][1]);
        pk.y = pack2<H>(v[mm][n][2], v[mm][n][3]);
        *reinterpret_cast<uint2*>(wp + mm * 16 * EPB_ROW + n * 32) = pk;
      }
    if (jrv && !has_ops) {
#pragma unroll
      for (int it = 0; it < 4; ++it) st16(reinterpret_cast<bf16*>(p.C) + base + it * rstep, ld16(lp + it * 8 * EPB_ROW));
    } else if (jrv) {
      V16 ax[4], rs[4];
      float tv[KD ? 4 : 1][8];
      if constexpr (KD) {
#pragma unroll
        for (int it = 0; it < 4; ++it) kd_load8(ep, kd_offset(p, g, i0 + xi0 + 32 * h + (lane >> 3) + 8 * it, jr), tv[it]);
      }
      if (ep.gelu_aux) {
#pragma unroll
        for (int it = 0; it < 4; ++it) ax[it] = ld16(reinterpret_cast<const bf16*>(ep.gelu_aux) + base + it * rstep);
      }
      if (ep.residual) {
#pragma unroll
        for (int it = 0; it < 4; ++it) rs[it] = ld16(reinterpret_cast<const bf16*>(ep.residual) + base + it * rstep);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float a[8], o[8];
        unpack16<H>(ld16(lp + it * 8 * EPB_ROW), a);
        if (ep.gelu_aux) {
          unpack16<H>(ax[it], o);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] *= ep.aux_factor ? o[e] : gelu_grad_t<H>(o[e]);
        }
        if (ep.residual) {
          unpack16<H>(rs[it], o);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += o[e];
        }
        if constexpr (KD) kd_acc[h >> 1] += kd_apply(ep, kc, tv[it], a);   // a = the staged (rounded) prediction -> its gradient
        st16(reinterpret_cast<bf16*>(p.C) + base + it * rstep, pack16<H>(a));
      }
    }
  }
  if constexpr (KD) {
#pragma unroll
    for (int sblk = 0; sblk < NM / 4; ++sblk) kd_write_slot(p, g, i0 + xi0 + 64 * sblk, j0 + yj0, lane, kd_acc[sblk]);
  }
  return true;
}

// XCD-aware block remap (guide T1): blocks b and b+8 share an XCD (and its private L2).  Logical tiles are handed out so
// that the tiles_j column tiles of one row tile -- which re-read the same X rows -- run back to back on ONE XCD; without
// it every XCD fetched every X tile on its own (PMC: GEMM FETCH_SIZE was ~4x the algorithmic bytes).  Bijective for any
// grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// Output tile (ti, tj) of logical block lb of the 256 x 256 loops.  Row-major over a wide output (tj fastest) makes the 32 CUs of an
// XCD walk ALL column tiles of 2 - 3 row tiles at a time: every K step they pull one 32 KiB slice per column tile through the XCD's
// 4 MiB L2 and come back to it a full round later, when it is gone (N = 3072, K = 768: 4.7 MB of weights per round).  Column PANELS
// of cb tiles with the rows fastest inside a panel keep a panel's weights (<= 1.5 MiB, at least 4 tiles, panels of equal width)
// resident while the rows stream: teacher fc1 -4.5 %, the q|k|v shape -9 %, 8192^3 -7 % (1.44 PFLOP/s); outputs of up to cb
// column tiles keep the row-major order (narrower panels lose: fewer row tiles share a weight slice).
__device__ __forceinline__ void tile_coords(int lb, int tiles_i, int tiles_j, int kred, int& ti, int& tj) {
  int cbmax = 3072 / (kred > 0 ? kred : 1);
  cbmax = cbmax < 4 ? 4 : cbmax;
  if (tiles_j <= cbmax) {
    ti = lb / tiles_j;
    tj = lb - ti * tiles_j;
    return;
  }
  const int np = (tiles_j + cbmax - 1) / cbmax, cb = (tiles_j + np - 1) / np;
  const int panel = tiles_i * cb;
  const int pn = lb / panel, r = lb - pn * panel;
  const int w = min(cb, tiles_j - pn * cb);
  ti = r / w;
  tj = pn * cb + (r - ti * w);
}
// (logical tile, reduction split) of this workgroup.  A split-K launch is a (tiles, splits) grid dispatched x-fastest, so the
// tiles of ONE split -- which re-read the same reduction slab of an operand (a weight gradient with 3 column tiles read dY three
// times) -- would sit on different XCDs; remapping the flattened id keeps them back to back on one XCD's L2.
// It is a per-shape choice (bit 16 of fh_gemm_desc.splitk, set by the tuner): the layer weight gradients gain ~10 %, the long
// conv-layer reductions (one row tile, 171 splits) lose as much.
constexpr int FH_SPLIT_REMAP = 1 << 16;
struct LogicalBlock { int lb, split; };
__device__ __forceinline__ LogicalBlock logical_block(int splitk_field) {
  if (gridDim.y == 1 || !(splitk_field & FH_SPLIT_REMAP)) return {xcd_remap(blockIdx.x, gridDim.x), (int)blockIdx.y};
  const int l = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int sp = l / (int)gridDim.x;
  return {l - sp * (int)gridDim.x, sp};
}

// ------------------------------------------------------------------ bf16 main loop on 32x32x16 MFMA (register-staged tiles)
// Half as many MFMA and LDS-read instructions per K-tile as the 16x16x32 form (16 + 16 vs 32 + 32 per wave): the student's
// GEMMs are issue-bound between MFMAs, not MFMA-bound.  R_STRIDED tiles use a 320-byte row so that the two 16-lane groups of
// a half-wave (same 4 k-rows, column halves 0 / 1) and the 4 rows land on 64 distinct banks.
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int S32_STRIDE = 320;
constexpr int TILE32_BYTES = 64 * S32_STRIDE;  // 20480 >= 128 * 144

template <int LAYOUT> __device__ __forceinline__ bf16x8 frag32(const unsigned char* lds, int row0, int s, int lane);
template <> __device__ __forceinline__ bf16x8 frag32<FH_R_CONTIG>(const unsigned char* lds, int row0, int s, int lane) {
  // lane (r = lane & 31, h = lane >> 5) holds k = 16 s + 8 h + e of row r
  return *reinterpret_cast<const bf16x8*>(lds + (row0 + (lane & 31)) * LDS_C_STRIDE + s * 32 + (lane >> 5) * 16);
}
template <> __device__ __forceinline__ bf16x8 frag32<FH_R_STRIDED>(const unsigned char* lds, int row0, int s, int lane) {
  const int i = lane & 15, grp = lane >> 4, qq = i >> 2, pp = i & 3;
  const int k = s * 16 + 8 * (grp >> 1) + qq;
  const unsigned char* a0 = lds + k * S32_STRIDE + (row0 + (grp & 1) * 16 + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union {
    struct { s16x4 lo, hi; } v;
    bf16x8 f;
  } u;
  u.v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  u.v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * S32_STRIDE));
  return u.f;
}

template <typename H, int XL, int YL>
__global__ void __launch_bounds__(256) gemm32_kernel(const GemmParams p) {
  constexpr int BK = 64;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE32_BYTES];
  unsigned char* ldsX = lds;
  unsigned char* ldsY = lds + TILE32_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN - 1) / BN;
  const LogicalBlock lblk = logical_block(p.splitk);
  const int lb = lblk.lb;
  const int ti = lb / tiles_j, tj = lb - ti * tiles_j;
  const int i0 = ti * BM, j0 = tj * BN;
  const int tiles_per_seg = (p.Kseg + BK - 1) / BK;
  const int ntiles = tiles_per_seg * tile_nseg(p, j0);
  const int per_split = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = lblk.split * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;

  Loader<H, XL> lx;
  Loader<H, YL> ly;
  lx.init(p.X, g, i0, p.M, tid);
  ly.init(p.Y, g, j0, p.N, tid);
  f32x16 acc[2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][m][e] = 0.f;
  const int xi0 = (wave & 1) * 64, yj0 = (wave >> 1) * 64;
  auto store_tiles = [&]() {
    if constexpr (XL == FH_R_CONTIG) lx.store(ldsX, tid); else lx.store_stride(ldsX, S32_STRIDE);
    if constexpr (YL == FH_R_CONTIG) ly.store(ldsY, tid); else ly.store_stride(ldsY, S32_STRIDE);
  };
  {
    const int seg = t0 / tiles_per_seg, k0 = (t0 - seg * tiles_per_seg) * BK;
    lx.load(p.X, seg, k0, p.Kseg);
    ly.load(p.Y, seg, k0, p.Kseg);
  }
  int nseg_t = t0 / tiles_per_seg, nk0 = (t0 - nseg_t * tiles_per_seg) * BK;
  for (int t = t0; t < t1; ++t) {
    store_tiles();
    __syncthreads();
    nk0 += BK;
    if (nk0 >= tiles_per_seg * BK) { nk0 = 0; ++nseg_t; }
    if (t + 1 < t1) {
      lx.load(p.X, nseg_t, nk0, p.Kseg);
      ly.load(p.Y, nseg_t, nk0, p.Kseg);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 fx[2], fy[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) fx[m] = frag32<XL>(ldsX, xi0 + 32 * m, s, lane);
#pragma unroll
      for (int n = 0; n < 2; ++n) fy[n] = frag32<YL>(ldsY, yj0 + 32 * n, s, lane);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[n][m] = mfma32<H>(fy[n], fx[m], acc[n][m]);
    }
    __syncthreads();
  }
  // staged epilogue: D[j_local][i_local] with i_local = lane & 31, j_local = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  unsigned char* my = lds + wave * EPI_WAVE;
  float bv[8];
  epilogue_bias(p, j0, yj0, lane, g, bv);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 v = {acc[n][h][4 * q4], acc[n][h][4 * q4 + 1], acc[n][h][4 * q4 + 2], acc[n][h][4 * q4 + 3]};
        *reinterpret_cast<f32x4*>(my + (lane & 31) * EPI_ROW + (n * 32 + 8 * q4 + 4 * (lane >> 5)) * 4) = v;
      }
    __syncthreads();
    epilogue_readback<H>(p, my, h, i0, j0, xi0, yj0, lane, g, bv);
  }
}

// ------------------------------------------------------------------ the kernel
template <typename T, int XL, int YL, bool KD = false>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BK = 8 * VEC;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
  unsigned char* ldsX = lds;
  unsigned char* ldsY = lds + TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN - 1) / BN;
  const LogicalBlock lblk = logical_block(p.splitk);
  const int lb = lblk.lb;
  const int ti = lb / tiles_j, tj = lb - ti * tiles_j;
  const int i0 = ti * BM, j0 = tj * BN;

  const int tiles_per_seg = (p.Kseg + BK - 1) / BK;
  const int ntiles = tiles_per_seg * tile_nseg(p, j0);
  const int per_split = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = lblk.split * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;  // (only possible in atomic split-K mode)

  Loader<T, XL> lx;
  Loader<T, YL> ly;
  lx.init(p.X, g, i0, p.M, tid);
  ly.init(p.Y, g, j0, p.N, tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int xi0 = (wave & 1) * 64, yj0 = (wave >> 1) * 64;

  {
    const int seg = t0 / tiles_per_seg, k0 = (t0 - seg * tiles_per_seg) * BK;
    lx.load(p.X, seg, k0, p.Kseg);
    ly.load(p.Y, seg, k0, p.Kseg);
  }
  for (int t = t0; t < t1; ++t) {
    lx.store(ldsX, tid);
    ly.store(ldsY, tid);
    __syncthreads();
    if (t + 1 < t1) {
      const int seg = (t + 1) / tiles_per_seg, k0 = (t + 1 - seg * tiles_per_seg) * BK;
      lx.load(p.X, seg, k0, p.Kseg);
      ly.load(p.Y, seg, k0, p.Kseg);
    }
    TileMath<T, XL, YL>::run(ldsX, ldsY, xi0, yj0, lane, acc);
    __syncthreads();
  }

  gemm_epilogue_staged<T, KD>(p, acc, lds, i0, j0, xi0, yj0, lane, wave, g);
}

// ------------------------------------------------------------------ direct-to-LDS loads (LDS-DMA): shared pieces
// global_load_lds_dwordx4 moves 16 B per lane straight into LDS (no VGPR staging, no ds_write pass): one wave
// instruction fills 1 KiB of LINEAR LDS.  Tiles are therefore unpadded; bank conflicts are removed with an XOR swizzle
// applied to the per-lane SOURCE address and, identically, to the fragment reads (guide rule 21).  A lane whose row / column /
// reduction index is out of range fetches from a 16-byte zero chunk, so tails need no masking.
__device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int strided_key(int kl) { return ((kl & 3) << 1) | (((kl >> 3) & 1) << 3); }

// fragment of a [32 k][256 B] R_STRIDED LDS block (chunk c of k-row kl stored at chunk c ^ strided_key(kl)), read transposed
template <int LAYOUT> __device__ __forceinline__ bf16x8 frag_dma(const unsigned char* lds, int row0, int lane);
template <> __device__ __forceinline__ bf16x8 frag_dma<FH_R_STRIDED>(const unsigned char* lds, int row0, int lane) {
  const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
  const int kl = 8 * q + qq;
  const int phys = ((row0 >> 3) + (pp >> 1)) ^ strided_key(kl);
  const unsigned char* a0 = lds + kl * 256 + phys * 16 + 8 * (pp & 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union {
    struct { s16x4 lo, hi; } s;
    bf16x8 v;
  } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 256));
  return u.v;
}

// ------------------------------------------------------------------ bf16 main loop, large tiles (LDS-DMA ring, 16x16x32 MFMA)
// The 128x128 loops above re-read every operand element from LDS once per 64 output columns / rows of a wave and keep
// 3-4 small workgroups per CU; on the student's compute-bound shapes they sit at 0.4-0.6 PFLOP/s.  This loop follows the
// guide's large-tile recipe instead: a 256 x 256 (or 256 x 128) output tile, each wave owning 128 x 64 of it (128 fp32
// accumulators per lane, half the LDS bytes per flop), operands streamed HBM -> LDS by global_load_lds_dwordx4 into an
// NST-deep ring with a counted vmcnt and ONE raw s_barrier per K-tile, no staging VGPRs and no ds_write at all.
//   R_CONTIG image : 16-row x 32-k sub-tiles of 1 KiB, one per wave instruction, st_16x32 swizzle (16-B chunk c of row r at
//                    chunk c ^ ((r >> 3) << 1)): ds_read_b128 of an MFMA fragment touches 16 distinct 16-B bank slots per group.
//   R_STRIDED image: [32 k][128 cols] blocks of 8 KiB (256-B k-rows, chunk c at c ^ strided_key(k)), read transposed with
//                    ds_read_b64_tr_b16 exactly as in the 128-tile DMA loop above.
// Out-of-range rows / columns / reduction indices fetch from a 16-byte zero chunk, so the K tail needs no masking.
// LDS-DMA issued from an asm statement: hipcc does not model it, so it neither drains it (vmcnt(0)) in front of the
// transposing LDS reads -- which it does for the builtin form, treating every pending DMA as a may-alias LDS write -- nor
// counts it; the ring's own counted vmcnt + barrier protocol orders the data (guide 5.7: M0 is written and restored
// inside the statement, the destination base is wave-uniform).
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned char* lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lptr_t)lds_dst));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// the same with the destination as a wave-uniform LDS byte address (kept in an SGPR: no 64-bit generic pointer per destination)
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
// two pieces 1 KiB apart in LDS with ONE M0 set-up: the instruction offset is added to the LDS address AND to the global address, so
// the second source pointer is passed 1 KiB low
__device__ __forceinline__ void glds16x2(const unsigned char* g0, const unsigned char* g1, unsigned lds_addr) {
  unsigned keep;
  const unsigned char* g1m = g1 - 1024;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g0), "v"(g1m), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lptr_t)p)); }

template <int LAYOUT, int ROWS, int BK_, int NW> struct RingLoader;

template <int ROWS, int BK_, int NW> struct RingLoader<FH_R_CONTIG, ROWS, BK_, NW> {
  static constexpr int KS = BK_ / 32;             // 32-k sub-tiles per row tile
  static constexpr int NI = ROWS / 16 * KS / NW;  // wave instructions per K-tile
  static constexpr int NR = NI / KS;              // distinct rows per lane
  static constexpr int TILE = ROWS * BK_ * 2;
  static_assert(NI >= KS && NI % KS == 0, "tile too small for this wave count");
  const bf16* base;
  long off[NR];
  long segdelta;
  unsigned vmask;
  int kch;  // logical k-chunk (8 elements) of this lane inside a sub-tile
  int wave;
  const unsigned char* ptr[NI];
  int step[NI];
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = tid >> 6;
    const int lane = tid & 63, r = lane >> 2;
    kch = (lane & 3) ^ ((r >> 3) << 1);
    vmask = 0;
    segdelta = (long)(o.tshift[1] - o.tshift[0]) * o.ld + (o.segoff[1] - o.segoff[0]);
#pragma unroll
    for (int ur = 0; ur < NR; ++ur) {
      const int idx = idx0 + (wave * NR + ur) * 16 + r;
      const bool rv = idx < idx_limit;
      RowMap rm = map_row(rv ? idx : 0, o.rpb);
      off[ur] = (long)rm.b * o.bstride + (long)(rm.t + o.tshift[0]) * o.ld + o.segoff[0];
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        const int ts = rm.t + o.tshift[sg];
        const bool v = rv && (sg < o.nseg) && (o.tvalid == 0 || (ts >= 0 && ts < o.tvalid)) && seg_live(o, sg, idx);
        vmask |= (v ? 1u : 0u) << (2 * ur + sg);
      }
    }
  }
  __device__ __forceinline__ void set_tile(const GemmOperand& o, int seg, int k0, int Kseg) {  // slow path
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int ur = u / KS, ks = u % KS;
      const int k = k0 + ks * 32 + kch * 8;
      const bool v = (k < Kseg) && ((vmask >> (2 * ur + seg)) & 1u);
      ptr[u] = v ? (const unsigned char*)(base + off[ur] + (long)seg * segdelta + k) : (const unsigned char*)g_zero_chunk;
      step[u] = v ? BK_ * 2 : 0;
    }
  }
  __device__ __forceinline__ void set_zero() {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      ptr[u] = (const unsigned char*)g_zero_chunk;
      step[u] = 0;
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      glds16(ptr[u], tile + (wave * NI + u) * 1024);
      ptr[u] += step[u];
    }
  }
  __device__ __forceinline__ void issue_one(unsigned char* tile, int u) {
    glds16(ptr[u], tile + (wave * NI + u) * 1024);
    ptr[u] += step[u];
  }
  // fragment of row tile `row0 / 16`, k-step ks: lane (r = lane & 15, q = lane >> 4) gets k = 32 ks + 8 q .. + 7 of row r
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int row0, int ks, int lane) {
    const int r = lane & 15, q = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(tile + ((row0 >> 4) * KS + ks) * 1024 + r * 64 + ((q ^ ((r >> 3) << 1)) << 4));
  }
};

template <int ROWS, int BK_, int NW> struct RingLoader<FH_R_STRIDED, ROWS, BK_, NW> {
  static constexpr int KB = BK_ / 32;                    // 32-k blocks per K-tile
  static constexpr int NI = (ROWS / 128) * KB * 8 / NW;  // wave instructions per K-tile (4 k-rows x 256 B each)
  static constexpr int TILE = ROWS * BK_ * 2;
  static_assert(NI >= 1 && ((ROWS / 128) * KB * 8) % NW == 0, "tile too small for this wave count");
  const bf16* base;
  int col[NI];
  int kt[NI];  // reduction row inside the K-tile
  unsigned cvalid;
  int wave;
  const unsigned char* ptr[NI];
  int tcur[NI];
  long step;   // bytes per K-tile
  long wrap;   // extra bytes when the mapped row index crosses a batch item
  int rpb;
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = tid >> 6;
    const int lane = tid & 63;
    cvalid = 0;
    rpb = o.rpb > 0 ? o.rpb : 0x7fffffff;
    wrap = o.rpb > 0 ? (o.bstride - (long)o.rpb * o.ld) * 2 : 0;
    step = (long)BK_ * o.ld * 2;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int si = wave * NI + u, blk = si >> 3, within = si & 7;
      const int mp = blk / KB, kb = blk % KB;
      const int klb = within * 4 + (lane >> 4);
      kt[u] = kb * 32 + klb;
      const int c = (lane & 15) ^ strided_key(klb);
      col[u] = idx0 + mp * 128 + c * 8;
      cvalid |= (col[u] < idx_limit ? 1u : 0u) << u;
    }
  }
  __device__ __forceinline__ void set_tile(const GemmOperand& o, int seg, int k0, int Kseg) {  // slow path
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int r = k0 + kt[u];
      ptr[u] = (const unsigned char*)g_zero_chunk;
      tcur[u] = -1;  // < 0: this lane stays on the zero chunk
      if (((cvalid >> u) & 1u) && r < Kseg && seg_live(o, seg, col[u])) {
        RowMap rm = map_row(r, o.rpb);
        ptr[u] = (const unsigned char*)(base + o.segoff[seg] + (long)rm.b * o.bstride + (long)rm.t * o.ld + col[u]);
        tcur[u] = rm.t;
      }
    }
  }
  __device__ __forceinline__ void set_zero() {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      ptr[u] = (const unsigned char*)g_zero_chunk;
      tcur[u] = -1;
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile) {
    // branch-free pointer advance (a branch here splits the loop into blocks and hipcc then drains vmcnt(0) before the
    // LDS reads): rpb == 0 is stored as INT_MAX, so the batch wrap never fires for it
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      glds16(ptr[u], tile + (wave * NI + u) * 1024);
      const bool live = tcur[u] >= 0;
      const int tn = tcur[u] + BK_;
      const bool w = live && tn >= rpb;  // requires rpb >= BK (checked on the host)
      ptr[u] += (live ? step : 0) + (w ? wrap : 0);
      tcur[u] = live ? (w ? tn - rpb : tn) : -1;
    }
  }
  __device__ __forceinline__ void issue_one(unsigned char* tile, int u) {
    glds16(ptr[u], tile + (wave * NI + u) * 1024);
    const bool live = tcur[u] >= 0;
    const int tn = tcur[u] + BK_;
    const bool w = live && tn >= rpb;
    ptr[u] += (live ? step : 0) + (w ? wrap : 0);
    tcur[u] = live ? (w ? tn - rpb : tn) : -1;
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int row0, int ks, int lane) {
    return frag_dma<FH_R_STRIDED>(tile + ((row0 >> 7) * KB + ks) * 8192, row0 & 127, lane);
  }
};

// ------------------------------------------------------------------ conv-0 weight-gradient epilogue (fh_gemm_epilogue.c0_*)
// The accumulators of a 256-frame x 128-channel tile ARE d loss / d y0 of those frames (y0 = GELU(GroupNorm(conv0(wave))), layer 1 being
// a k = 1 conv): instead of storing them (1.05 GB at B = 128) for a second kernel to read back, each lane turns its 8 frames x 16
// channels into g = dy * GELU'(.), with conv0 + GroupNorm recomputed from the waveform window staged in LDS, and accumulates what
// conv0_bwd_onepass_kernel accumulates: A_j = sum_t g x_j[t], S1 = sum g, S2 = sum g xhat.  Two channels at a time (their taps and 12 sums
// in registers beside the 128 accumulators), the 16 frame-lanes of a channel reduced with four DPP adds, the two row-waves through LDS.
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a row (every lane receives it)
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1, 0, 3, 2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2, 3, 0, 1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}
template <typename H, int NT, int MT>
__device__ __forceinline__ void conv0_wgrad_epilogue(const GemmParams& p, const f32x4 (&acc)[NT][MT], unsigned char* lds, int i0, int xi0,
                                                     int yj0, int lane, int wave_row, int b, int ti, int ntile) {
  constexpr int KT = 10, PER = KT + 2;
  const GemmEpilogue& ep = p.ep;
  const int stride = ep.c0_stride, K = ep.c0_K, C0 = p.N;
  float* sw = reinterpret_cast<float*>(lds);       // the tile's waveform window: 256 * stride + K samples (zero past the utterance)
  float* red = sw + 2048;                          // [2 row-waves][C0][PER]
  float* wl = red + 2 * 128 * PER;                 // conv-0 taps [C0][KT] (zero past K)
  float* stl = wl + 128 * KT;                      // per channel: mean, rstd of this utterance, gamma, beta
  {
    const long s0 = (long)i0 * stride;
    const float* src = ep.c0_wave + (long)b * ep.c0_S;
    const int n = 256 * stride + 16;
    for (int i = threadIdx.x; i < n && i < 2048; i += blockDim.x) sw[i] = s0 + i < ep.c0_S ? src[s0 + i] : 0.f;
    for (int i = threadIdx.x; i < C0 * KT; i += blockDim.x) {
      const int c = i / KT, j = i - c * KT;
      wl[i] = j < K ? ep.c0_w[c * K + j] : 0.f;
    }
    for (int c = threadIdx.x; c < C0; c += blockDim.x) {
      stl[4 * c] = ep.c0_mean[b * C0 + c];
      stl[4 * c + 1] = ep.c0_rstd[b * C0 + c];
      stl[4 * c + 2] = ep.c0_gamma[c];
      stl[4 * c + 3] = ep.c0_beta[c];
    }
  }
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c0 = yj0 + 16 * n + 4 * q + 2 * half;
      float w[2][KT], mu[2], rs[2], ga[2], be[2], a12[2][PER];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = c0 + e;
#pragma unroll
        for (int j = 0; j < KT; ++j) w[e][j] = wl[c * KT + j];
        mu[e] = stl[4 * c]; rs[e] = stl[4 * c + 1]; ga[e] = stl[4 * c + 2]; be[e] = stl[4 * c + 3];
#pragma unroll
        for (int j = 0; j < PER; ++j) a12[e][j] = 0.f;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float* x = sw + (xi0 + 16 * m + r) * stride;
        float xs[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) xs[j] = x[j];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = 0.f;
#pragma unroll
          for (int j = 0; j < KT; ++j) v = fmaf(w[e][j], xs[j], v);
          const float xh = (v - mu[e]) * rs[e];
          const float gq = acc[n][m][2 * half + e] * gelu_grad_t<H>(fmaf(xh, ga[e], be[e]));
          a12[e][KT] += gq;
          a12[e][KT + 1] = fmaf(gq, xh, a12[e][KT + 1]);
#pragma unroll
          for (int j = 0; j < KT; ++j) a12[e][j] = fmaf(gq, xs[j], a12[e][j]);
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const float t = row16_sum(a12[e][j]);
          if (r == 0) red[(wave_row * C0 + c0 + e) * PER + j] = t;
        }
    }
  }
  __syncthreads();
  // ONE block of sums per utterance, accumulated with fp32 atomics (zeroed by the caller): a partial block per tile was 125 x 6 KiB per
  // utterance -- 98 MB at B = 128 -- which the fp64 combine behind it then read back at the end of the step (0.63 ms beside the
  // concurrent weight-gradient stream); the 1536 atomics of a tile leave as the tile retires, spread over the launch
  (void)ntile; (void)ti;
  float* out = ep.c0_wpart + (long)b * C0 * PER;
  for (int i = threadIdx.x; i < C0 * PER; i += blockDim.x) atomicAdd(out + i, red[i] + red[C0 * PER + i]);
}

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <typename H, int XL, int YL, int BM_, int BN_, int WM, int WN, int BK_, int NST, bool KD = false, bool C0B = false>
// (the 4-wave build is meant to run TWO workgroups per CU: without the second launch-bound argument the compiler took 364 registers
// per wave -- 236 + 128 accumulators -- and one workgroup was all a CU could hold)
__global__ void __launch_bounds__(WM* WN * 64, WM * WN == 4 ? 2 : 1) gemm_ring_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, TM = BM_ / WM, TN = BN_ / WN, MT = TM / 16, NT = TN / 16, KS = BK_ / 32;
  typedef RingLoader<XL, BM_, BK_, NW> LX;
  typedef RingLoader<YL, BN_, BK_, NW> LY;
  constexpr int STAGE = LX::TILE + LY::TILE;
  constexpr int LPT = LX::NI + LY::NI;  // LDS-DMA instructions per lane per K-tile
  static_assert(TM % 32 == 0 && TN % 64 == 0, "epilogue passes are 32 rows x 64 columns per wave");
  static_assert(NST * STAGE >= NW * EPI_WAVE, "ring smaller than the epilogue staging area");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN_ - 1) / BN_;
  const LogicalBlock lblk = logical_block(p.splitk);
  const int lb = lblk.lb;
  const int ti = lb / tiles_j, tj = lb - ti * tiles_j;
  const int i0 = ti * BM_, j0 = tj * BN_;
  const int tiles_per_seg = (p.Kseg + BK_ - 1) / BK_;
  const int ntiles = tiles_per_seg * tile_nseg(p, j0);
  const int per_split = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = lblk.split * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;
  const int nt = t1 - t0;

  LX lx;
  LY ly;
  lx.init(p.X, g, i0, p.M, tid);
  ly.init(p.Y, g, j0, p.N, tid);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xi0 = (wave % WM) * TM, yj0 = (wave / WM) * TN;

  int is_seg = t0 / tiles_per_seg;
  int is_k0 = (t0 - is_seg * tiles_per_seg) * BK_;
  int issued = 0;
  bool need_slow = true;
  auto issue_next = [&](int stage) {
    unsigned char* st = lds + stage * STAGE;
    if (issued == nt) {  // pipeline drain: harmless zero-chunk loads keep the vmcnt bookkeeping uniform
      lx.set_zero();
      ly.set_zero();
    } else if (issued < nt) {
      const bool partial = is_k0 + BK_ > p.Kseg;
      if (need_slow || partial) {
        lx.set_tile(p.X, is_seg, is_k0, p.Kseg);
        ly.set_tile(p.Y, is_seg, is_k0, p.Kseg);
      }
      is_k0 += BK_;
      need_slow = false;
      if (is_k0 >= p.Kseg) {
        is_k0 = 0;
        is_seg = min(is_seg + 1, 1);
        need_slow = true;
      }
    }
    if (issued < NST - 1) {  // prologue tiles go out as a burst; in the loop the DMA rides between the MFMA groups
      lx.issue(st);
      ly.issue(st + LX::TILE);
    }
    ++issued;
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue_next(s);

  int stage = 0;
  for (int it = 0; it < nt; ++it) {
    // this wave's loads of tile `it` have landed (NST-2 younger tiles may still be in flight) ...
    wait_vm_barrier<(NST - 2) * LPT>();
    // ... and after the barrier so have everyone's; the stage read in iteration it-1 is free again: refill it
    issue_next(stage == 0 ? NST - 1 : stage - 1);  // pointers only: the loads themselves are spread over the MFMA groups
    const unsigned char* bx = lds + stage * STAGE;
    const unsigned char* by = bx + LX::TILE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 fx[MT], fy[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) fy[n] = LY::frag(by, yj0 + 16 * n, ks, lane);
#pragma unroll
      for (int m = 0; m < MT; ++m) fx[m] = LX::frag(bx, xi0 + 16 * m, ks, lane);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n][m] = mfma16<H>(fy[n], fx[m], acc[n][m]);
        {
          // the tile's LPT LDS-DMA instructions, spread evenly behind the KS * MT MFMA groups: issued as one burst at the
          // top of the tile they held every wave's (and the SIMD's) issue for ~1k cycles while the MFMA pipe idled
          unsigned char* st = lds + (stage == 0 ? NST - 1 : stage - 1) * STAGE;
          constexpr int S = KS * MT;
          const int slot = ks * MT + m;
#pragma unroll
          for (int u = slot * LPT / S; u < (slot + 1) * LPT / S; ++u) {
            if (u < LX::NI) lx.issue_one(st, u);
            else ly.issue_one(st + LX::TILE, u - LX::NI);
          }
        }
      }
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the (zero-chunk) tail loads before LDS is reused
  __syncthreads();                                   // every wave is done reading the ring
  if constexpr (C0B) {   // the tile is consumed, not stored: conv-0 weight-gradient sums of its 256 frames
    static_assert(!C0B || (BM_ == 256 && BN_ == 128 && WM == 2 && TN == 64), "one column tile of 128 channels, two row-waves");
    conv0_wgrad_epilogue<H, NT, MT>(p, acc, lds, i0, xi0, yj0, lane, wave % WM, g, ti, (p.M + BM_ - 1) / BM_);
    return;
  }
  // staged epilogue, wave-local from here on (each wave owns its staging rows): passes of 32 rows x 64 columns
  unsigned char* my = lds + wave * EPI_WAVE;
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < TN / 64; ++cb) {
    // outputs without per-row operands: the 16-bit staged form (half the LDS traffic, bias / GELU / dropout in the accumulator layout) --
    // what loops 7 / 8 use; with it the 4-wave 256 x 128 build (two workgroups per CU: one stores while the other multiplies) is a
    // candidate for the forward launches too
    if (epilogue_bf16<H, MT, KD>(p, acc, cb * 4, my, i0, j0, xi0, yj0 + 64 * cb, lane, g)) continue;
    float bv[8];
    float kd_acc[TM / 64] = {};
    epilogue_bias(p, j0, yj0 + 64 * cb, lane, g, bv);
#pragma unroll
    for (int h = 0; h < TM / 32; ++h) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          *reinterpret_cast<f32x4*>(my + (mm * 16 + r) * EPI_ROW + (n * 16 + 4 * q) * 4) = acc[cb * 4 + n][2 * h + mm];
      epilogue_readback<H, KD>(p, my, h, i0, j0, xi0, yj0 + 64 * cb, lane, g, bv, &kd_acc[h >> 1]);
    }
    if constexpr (KD) {
#pragma unroll
      for (int sblk = 0; sblk < TM / 64; ++sblk) kd_write_slot(p, g, i0 + xi0 + 64 * sblk, j0 + yj0 + 64 * cb, lane, kd_acc[sblk]);
    }
  }
}

// ------------------------------------------------------------------ small-M loop (R_CONTIG x R_CONTIG): the whole reduction in one trip
// At the reference's own batch (3 utterances per GPU: M = 747 rows per layer GEMM) a launch has a few dozen tiles and its time
// is the LATENCY of its K loop: 7.5 dependent load -> LDS -> MFMA steps of ~2.5 us each = 22-27 us for 30 MFLOP, 110 such launches
// per step.  Here a workgroup of 4 waves owns a 64 x 64 tile and brings the WHOLE reduction extent (512 elements per pass: 64 KiB per
// operand) into LDS with one burst of LDS-DMA -- one memory round trip --, then runs all its MFMAs back to back; wave w owns the
// 16-column strip w (4 accumulators), the operand image and fragment reads are the ring loaders'.  ~4x more tiles than the
// 128-tile loops (so the few rows spread over the CUs) and ~3x shorter per launch.
template <typename H>
__global__ void __launch_bounds__(256) gemm_small_kernel(const GemmParams p) {
  typedef RingLoader<FH_R_CONTIG, 64, 512, 4> LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + 63) / 64;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int ti = lb / tiles_j, tj = lb - ti * tiles_j;
  const int i0 = ti * 64, j0 = tj * 64;
  LD lx, ly;
  lx.init(p.X, g, i0, p.M, tid);
  ly.init(p.Y, g, j0, p.N, tid);
  f32x4 acc[1][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nseg = tile_nseg(p, j0);
  for (int seg = 0; seg < nseg; ++seg)
    for (int k0 = 0; k0 < p.Kseg; k0 += 512) {
      lx.set_tile(p.X, seg, k0, p.Kseg);
      ly.set_tile(p.Y, seg, k0, p.Kseg);
      lx.issue(lds);
      ly.issue(lds + LD::TILE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int nks = min(16, (p.Kseg - k0 + 31) >> 5);
      for (int ks = 0; ks < nks; ++ks) {
        const bf16x8 fy = LD::frag(lds + LD::TILE, 16 * wave, ks, lane);
        bf16x8 fx[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) fx[m] = LD::frag(lds, 16 * m, ks, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[0][m] = mfma16<H>(fy, fx[m], acc[0][m]);
      }
      __syncthreads();  // the tiles are rewritten by the next pass
    }
  gemm_epilogue<H, 1, 4>(p, acc, i0, j0, 0, 16 * wave, lane, g);
}
bool small_ok(const GemmParams& p) {
  return p.X.layout == FH_R_CONTIG && p.Y.layout == FH_R_CONTIG && !p.ep.atomic && (p.splitk & 0xffff) <= 1;
}
template <typename H> int launch_small(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + 63) / 64, tiles_j = (p.N + 63) / 64;
  dim3 grid(tiles_i * tiles_j, 1, p.groups), block(256);
  const size_t smem = 2 * (size_t)64 * 512 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_small_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_small_kernel<H>, grid, block, smem, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}

template <typename T> int launch_t(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + BM - 1) / BM, tiles_j = (p.N + BN - 1) / BN;
  dim3 grid(tiles_i * tiles_j, p.splitk & 0xffff, p.groups), block(256);
  const int xl = p.X.layout, yl = p.Y.layout;
  if (p.ep.kd_teacher) {
    if (xl != FH_R_CONTIG || yl != FH_R_CONTIG) return -113;
    hipLaunchKernelGGL((gemm_kernel<T, FH_R_CONTIG, FH_R_CONTIG, true>), grid, block, 0, stream, p);
    FH_CHECK_LAUNCH();
    return 0;
  }
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_kernel<T, FH_R_CONTIG, FH_R_CONTIG>), grid, block, 0, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm_kernel<T, FH_R_CONTIG, FH_R_STRIDED>), grid, block, 0, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_kernel<T, FH_R_STRIDED, FH_R_CONTIG>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<T, FH_R_STRIDED, FH_R_STRIDED>), grid, block, 0, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}

template <typename H> int launch_32(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + BM - 1) / BM, tiles_j = (p.N + BN - 1) / BN;
  dim3 grid(tiles_i * tiles_j, p.splitk & 0xffff, p.groups), block(256);
  const int xl = p.X.layout, yl = p.Y.layout;
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm32_kernel<H, FH_R_CONTIG, FH_R_CONTIG>), grid, block, 0, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm32_kernel<H, FH_R_CONTIG, FH_R_STRIDED>), grid, block, 0, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm32_kernel<H, FH_R_STRIDED, FH_R_CONTIG>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((gemm32_kernel<H, FH_R_STRIDED, FH_R_STRIDED>), grid, block, 0, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}


// Ping-pong form of the ring (256 x 256 tile, 8 waves, BK = 32, 4 tile buffers): each K-tile is two phases of 16 MFMAs
// (the wave's upper / lower 64 output rows), every phase = { LDS reads of its fragments + two LDS-DMA issues | barrier |
// 16 MFMAs | barrier }.  Waves 4-7 run one barrier behind waves 0-3 (one extra s_barrier up front, made up at the end), and
// each SIMD hosts one wave of either half: while one wave of a SIMD reads LDS / issues DMA the other owns the MFMA pipe.
//   RAW: a wave waits vmcnt(8) for its own DMA of tile t+1 in the read part of tile t's second phase; the first read of
//        tile t+1 (waves 0-3, two barriers later) is behind a barrier that every wave passed after that wait.
//   WAR: tile t+3 goes to the buffer of tile t-1, whose last reads were retired (lgkmcnt(0) before the barrier) one
//        barrier before the earliest issue.
template <typename H, int XL, int YL>
__global__ void __launch_bounds__(512) gemm_pp_kernel(const GemmParams p) {
  constexpr int BM_ = 256, BN_ = 256, BK_ = 32, NW = 8, NST = 4, MT = 8, NT = 4;
  typedef RingLoader<XL, BM_, BK_, NW> LX;
  typedef RingLoader<YL, BN_, BK_, NW> LY;
  constexpr int STAGE = LX::TILE + LY::TILE;
  static_assert(LX::NI == 2 && LY::NI == 2, "vmcnt bookkeeping below assumes 4 LDS-DMA per lane per K-tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN_ - 1) / BN_;
  const LogicalBlock lblk = logical_block(p.splitk);
  const int lb = lblk.lb;
  const int ti = lb / tiles_j, tj = lb - ti * tiles_j;
  const int i0 = ti * BM_, j0 = tj * BN_;
  const int tiles_per_seg = (p.Kseg + BK_ - 1) / BK_;
  const int ntiles = tiles_per_seg * tile_nseg(p, j0);
  const int per_split = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = lblk.split * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;
  const int nt = t1 - t0;
  const bool late = __builtin_amdgcn_readfirstlane(tid) >= 256;  // waves 4-7

  LX lx;
  LY ly;
  lx.init(p.X, g, i0, p.M, tid);
  ly.init(p.Y, g, j0, p.N, tid);
  f32x4 acc[NT][MT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xi0 = (wave & 1) * 128, yj0 = (wave >> 1) * 64;

  int is_seg = t0 / tiles_per_seg;
  int is_k0 = (t0 - is_seg * tiles_per_seg) * BK_;
  int issued = 0;
  bool need_slow = true;
  auto advance = [&]() {  // source pointers of the next tile to issue (zero chunk past the end)
    if (issued == nt) {
      lx.set_zero();
      ly.set_zero();
    } else if (issued < nt) {
      const bool partial = is_k0 + BK_ > p.Kseg;
      if (need_slow || partial) {
        lx.set_tile(p.X, is_seg, is_k0, p.Kseg);
        ly.set_tile(p.Y, is_seg, is_k0, p.Kseg);
      }
      is_k0 += BK_;
      need_slow = false;
      if (is_k0 >= p.Kseg) {
        is_k0 = 0;
        is_seg = min(is_seg + 1, 1);
        need_slow = true;
      }
    }
    ++issued;
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    advance();
    lx.issue(lds + s * STAGE);
    ly.issue(lds + s * STAGE + LX::TILE);
  }
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // tile 0 has landed everywhere
  if (late) __builtin_amdgcn_s_barrier();                          // the stagger
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 fx[4], fy[NT];
  int rd = 0, wr = NST - 1;
  for (int it = 0; it < nt; ++it) {
    const unsigned char* bx = lds + rd * STAGE;
    const unsigned char* by = bx + LX::TILE;
    unsigned char* wx = lds + wr * STAGE;
    // ---- phase 1: rows 0-63 of the wave's tile; DMA the X half of tile it+3
#pragma unroll
    for (int n = 0; n < NT; ++n) fy[n] = LY::frag(by, yj0 + 16 * n, 0, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) fx[m] = LX::frag(bx, xi0 + 16 * m, 0, lane);
    advance();
    lx.issue(wx);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n][m] = mfma16<H>(fy[n], fx[m], acc[n][m]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: rows 64-127; DMA the Y half of tile it+3; tile it+1 must have landed when the next phase reads it
#pragma unroll
    for (int m = 0; m < 4; ++m) fx[m] = LX::frag(bx, xi0 + 64 + 16 * m, 0, lane);
    ly.issue(wx + LX::TILE);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n][4 + m] = mfma16<H>(fy[n], fx[m], acc[n][4 + m]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rd = rd == NST - 1 ? 0 : rd + 1;
    wr = wr == NST - 1 ? 0 : wr + 1;
  }
  if (!late) __builtin_amdgcn_s_barrier();  // make up the stagger
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tail DMA drained, every wave done with the ring
  unsigned char* my = lds + wave * EPI_WAVE;
  const int r = lane & 15, q = lane >> 4;
  float bv[8];
  epilogue_bias(p, j0, yj0, lane, g, bv);
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        *reinterpret_cast<f32x4*>(my + (mm * 16 + r) * EPI_ROW + (n * 16 + 4 * q) * 4) = acc[n][2 * h + mm];
    epilogue_readback<H>(p, my, h, i0, j0, xi0, yj0, lane, g, bv);
  }
}

template <typename H> int launch_pp(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + 255) / 256, tiles_j = (p.N + 255) / 256;
  dim3 grid(tiles_i * tiles_j, p.splitk & 0xffff, p.groups), block(512);
  const size_t smem = (size_t)4 * 512 * 32 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_pp_kernel<H, FH_R_CONTIG, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<H, FH_R_CONTIG, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<H, FH_R_STRIDED, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_pp_kernel<H, FH_R_STRIDED, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int xl = p.X.layout, yl = p.Y.layout;
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_pp_kernel<H, FH_R_CONTIG, FH_R_CONTIG>), grid, block, smem, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm_pp_kernel<H, FH_R_CONTIG, FH_R_STRIDED>), grid, block, smem, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_pp_kernel<H, FH_R_STRIDED, FH_R_CONTIG>), grid, block, smem, stream, p);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<H, FH_R_STRIDED, FH_R_STRIDED>), grid, block, smem, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ 8-phase form (R_CONTIG x R_CONTIG), after the guide's 256^2 template
// 256 x 256 tile, BK = 64, two 64-KiB tile buffers, 8 waves (2 x 4), each wave's 128 x 64 output in four quadrants.  A K-tile
// is four phases { LDS reads of the quadrant's fragments + ONE quarter of a later tile by LDS-DMA (2 instructions) | barrier |
// 16 MFMAs | barrier } over the quadrants (mh0, nh0) (mh0, nh1) (mh1, nh1) (mh1, nh0)   (mh / nh: upper / lower 64 rows, left /
// right 32 columns of every wave's tile).  The quarter a phase has just read is dead, so it is refilled at once with the tile TWO
// ahead (first form of this loop: the tile one ahead, issued during the iteration before its use -- its third quarter then had one
// phase, ~0.35 us, to arrive before the counted wait in front of it, less than a loaded L2 / HBM round trip, and every K-tile
// stalled there: -4 ... -7 % on K >= 768).  Waves 4-7 run one barrier behind waves 0-3, so each SIMD alternates one wave
// reading / issuing with one wave in its MFMA cluster.
template <int LAYOUT> struct QuarterLoader;
template <> struct QuarterLoader<FH_R_CONTIG> {  // 256 rows x 64 k, image = 16-row x 32-k sub-tiles (st_16x32), as RingLoader
  const bf16* base;
  long off[2];       // [half]
  long segdelta;
  unsigned vmask;
  int kch, wave, rt[2];
  const unsigned char* ptr[2][2];  // [half][ks]
  int step[2][2];
  // is_x: X operand (halves = upper / lower 64 rows of each of the 2 wave rows); else Y (halves = 32-column halves of 4 wave columns)
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid, bool is_x) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform values stay in SGPRs: the LDS destinations are scalar)
    const int lane = tid & 63, r = lane >> 2;
    kch = (lane & 3) ^ ((r >> 3) << 1);
    vmask = 0;
    segdelta = (long)(o.tshift[1] - o.tshift[0]) * o.ld + (o.segoff[1] - o.segoff[0]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      rt[h] = is_x ? (wave >> 2) * 8 + h * 4 + (wave & 3) : (wave >> 1) * 4 + h * 2 + (wave & 1);
      const int idx = idx0 + rt[h] * 16 + r;
      const bool rv = idx < idx_limit;
      RowMap rm = map_row(rv ? idx : 0, o.rpb);
      off[h] = (long)rm.b * o.bstride + (long)(rm.t + o.tshift[0]) * o.ld + o.segoff[0];
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        const int ts = rm.t + o.tshift[sg];
        const bool v = rv && (sg < o.nseg) && (o.tvalid == 0 || (ts >= 0 && ts < o.tvalid)) && seg_live(o, sg, idx);
        vmask |= (v ? 1u : 0u) << (2 * h + sg);
      }
    }
  }
  __device__ __forceinline__ void set_tile(const GemmOperand& o, int seg, int k0, int Kseg) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k = k0 + ks * 32 + kch * 8;
        const bool v = (k < Kseg) && ((vmask >> (2 * h + seg)) & 1u);
        ptr[h][ks] = v ? (const unsigned char*)(base + off[h] + (long)seg * segdelta + k) : (const unsigned char*)g_zero_chunk;
        step[h][ks] = v ? 128 : 0;
      }
  }
  __device__ __forceinline__ void set_zero() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ptr[h][ks] = (const unsigned char*)g_zero_chunk;
        step[h][ks] = 0;
      }
  }
  __device__ __forceinline__ void issue_half(unsigned tile, int h) {  // tile: LDS byte address of the operand's 32 KiB image
    glds16x2(ptr[h][0], ptr[h][1], tile + (unsigned)(rt[h] * 2 * 1024));
    ptr[h][0] += step[h][0];
    ptr[h][1] += step[h][1];
  }
  __device__ __forceinline__ void skip_half(int h) {  // the pointer advance of issue_half without the DMA
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ptr[h][ks] += step[h][ks];
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int row0, int ks, int lane) {
    return RingLoader<FH_R_CONTIG, 256, 64, 8>::frag(tile, row0, ks, lane);
  }
};

// R_STRIDED operand (reduction index is the memory row, 256 columns contiguous): image = eight 32-column panels of
// [64 k][64 B]; the 32-byte window w (16 columns) of k-row k sits at window w ^ ((k >> 3) & 1), so the eight k-rows a
// half-wave's ds_read_b64_tr_b16 touches (k0..k0+3 and k0+8..k0+11, 64-byte rows) land in eight different 32-byte bank
// slots.  One LDS-DMA instruction = 16 k-rows x 64 B of one panel; a quarter = four panels x four k-chunks.
template <> struct QuarterLoader<FH_R_STRIDED> {
  const bf16* base;
  int col[2][2];   // [half][u] first of this lane's 8 columns
  int kt[2][2];    // reduction row inside the K-tile
  unsigned cvalid;
  int wave;
  int dst[2][2];   // LDS byte offset of the instruction's 1 KiB
  const unsigned char* ptr[2][2];
  int tcur[2][2];
  long step, wrap;
  int rpb;
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid, bool is_x) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    cvalid = 0;
    rpb = o.rpb > 0 ? o.rpb : 0x7fffffff;
    wrap = o.rpb > 0 ? (o.bstride - (long)o.rpb * o.ld) * 2 : 0;
    step = (long)64 * o.ld * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = 2 * wave + u, pq = idx >> 2, kc = idx & 3;
        const int P = is_x ? (pq >> 1) * 4 + h * 2 + (pq & 1) : 2 * pq + h;
        const int klocal = 16 * kc + (lane >> 2);
        const int pc = lane & 3;
        const int w = (pc >> 1) ^ ((klocal >> 3) & 1);
        kt[h][u] = klocal;
        col[h][u] = idx0 + P * 32 + w * 16 + (pc & 1) * 8;
        dst[h][u] = P * 4096 + kc * 1024;
        cvalid |= (col[h][u] < idx_limit ? 1u : 0u) << (2 * h + u);
      }
  }
  __device__ __forceinline__ void set_tile(const GemmOperand& o, int seg, int k0, int Kseg) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = k0 + kt[h][u];
        ptr[h][u] = (const unsigned char*)g_zero_chunk;
        tcur[h][u] = -1;
        if (((cvalid >> (2 * h + u)) & 1u) && r < Kseg && seg_live(o, seg, col[h][u])) {
          RowMap rm = map_row(r, o.rpb);
          ptr[h][u] = (const unsigned char*)(base + o.segoff[seg] + (long)rm.b * o.bstride + (long)rm.t * o.ld + col[h][u]);
          tcur[h][u] = rm.t;
        }
      }
  }
  __device__ __forceinline__ void set_zero() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ptr[h][u] = (const unsigned char*)g_zero_chunk;
        tcur[h][u] = -1;
      }
  }
  __device__ __forceinline__ void issue_half(unsigned tile, int h) { step_half<true>(tile, h); }
  __device__ __forceinline__ void skip_half(int h) { step_half<false>(0u, h); }
  template <bool DMA> __device__ __forceinline__ void step_half(unsigned tile, int h) {
    if (DMA) glds16x2(ptr[h][0], ptr[h][1], tile + (unsigned)dst[h][0]);  // dst[h][1] = dst[h][0] + 1024 (k-chunks 2 w', 2 w' + 1 of one panel)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool live = tcur[h][u] >= 0;
      const int tn = tcur[h][u] + 64;
      const bool w = live && tn >= rpb;  // requires rpb >= 64 (checked on the host)
      ptr[h][u] += (live ? step : 0) + (w ? wrap : 0);
      tcur[h][u] = live ? (w ? tn - rpb : tn) : -1;
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int row0, int ks, int lane) {
    const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
    const int k = ks * 32 + 8 * q + qq;
    const int w = ((row0 >> 4) & 1) ^ ((k >> 3) & 1);
    const unsigned char* a0 = tile + (row0 >> 5) * 4096 + k * 64 + w * 32 + pp * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    union {
      struct { s16x4 lo, hi; } v;
      bf16x8 f;
    } u;
    u.v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 64));
    return u.f;
  }
};

// phase building blocks of the 8-phase loops (they name acc, fa, fb, bx, by, xi0, yj0, lane of the enclosing function):
// the DMA wait (this wave's own loads, RAW) goes BEFORE the barrier; the fragment reads are only waited for behind it, so
// their latency overlaps the wait for the other wave group (the buffers they read are rewritten a whole K-tile later)
// FH_PHASE_LGKM_WAIT: rounds 2 - 4 drained every fragment read (lgkmcnt(0)) behind the barrier before the cluster's first MFMA; the
// compiler tracks these reads itself (they are plain LDS loads) and emits a counted lgkmcnt in front of each MFMA pair, so without the
// drain the cluster starts as soon as its FIRST fragments have landed (-DFH_PHASE_DRAIN restores the drain for A/B runs)
#ifdef FH_PHASE_DRAIN
#define FH_PHASE_LGKM_WAIT asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define FH_PHASE_LGKM_WAIT
#endif
#define FH_PHASE_TAIL(WAITSTR)                                   \
  asm volatile(WAITSTR ::: "memory");                            \
  __builtin_amdgcn_sched_barrier(0);                             \
  __builtin_amdgcn_s_barrier();                                  \
  FH_PHASE_LGKM_WAIT                                             \
  __builtin_amdgcn_sched_barrier(0);                             \
  __builtin_amdgcn_s_setprio(1);
// Before a quarter that the previous phase read is handed to the LDS-DMA again: this wave's FIRST fragment read of the current
// phase has returned.  The waves of the other group issued their reads of that quarter before the barrier this wave has just
// passed; the LDS serves requests in arrival order, so a read issued behind them having returned means theirs have been served
// (the DMA data could only overtake them through an L1 hit on the zero chunk, ~150 cycles; N = this phase's LDS reads - 1,
// at most 15; scalar loads in flight only make the wait stricter).
#define FH_REFILL_GUARD(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((N) > 15 ? 15 : (N)) : "memory");
#define FH_PHASE_MMA(MH, NH)                                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int m = 0; m < 4; ++m)                       \
      _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[(NH) * 2 + n][(MH) * 4 + m] =                                  \
          mfma16<H>(fb[ks][n], fa[ks][m], acc[(NH) * 2 + n][(MH) * 4 + m]);     \
  __builtin_amdgcn_s_setprio(0);                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);
#define FH_READ_A(MH)                                                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int m = 0; m < 4; ++m) fa[ks][m] =           \
      QuarterLoader<XL>::frag(bx, xi0 + (MH) * 64 + 16 * m, ks, lane);
#define FH_READ_B(NH)                                                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int n = 0; n < 2; ++n) fb[ks][n] =           \
      QuarterLoader<YL>::frag(by, yj0 + (NH) * 32 + 16 * n, ks, lane);


// one 256 x 256 output tile (logical tile lb, reduction split lblk.split of nsplit) of the 8-phase loop
template <typename H, int XL, int YL, bool KD = false>
__device__ __forceinline__ void gemm_8ph_tile(const GemmParams& p, unsigned char* lds, const LogicalBlock lblk, int nsplit, int g) {
  constexpr int BK_ = 64, XT = 256 * 64 * 2, STAGE = 2 * XT;
  constexpr int RA = XL == FH_R_STRIDED ? 16 : 8, RB = YL == FH_R_STRIDED ? 8 : 4;  // LDS reads of an A / B fragment phase
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_j = (p.N + 255) / 256;
  const int lb = lblk.lb;
  int ti, tj;
  tile_coords(lb, (p.M + 255) / 256, tiles_j, p.Kseg * p.X.nseg, ti, tj);
  const int i0 = ti * 256, j0 = tj * 256;
  const int tiles_per_seg = (p.Kseg + BK_ - 1) / BK_;
  const int ntiles = tiles_per_seg * tile_nseg(p, j0);
  const int per_split = (ntiles + nsplit - 1) / nsplit;
  const int t0 = lblk.split * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;
  const int nt = t1 - t0;
  const bool late = __builtin_amdgcn_readfirstlane(tid) >= 256;

  QuarterLoader<XL> lx;
  QuarterLoader<YL> ly;
  lx.init(p.X, g, i0, p.M, tid, true);
  ly.init(p.Y, g, j0, p.N, tid, false);
  f32x4 acc[4][8];  // [n][m]
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xi0 = (wave >> 2) * 128, yj0 = (wave & 3) * 64;  // wave rows 0-1 (X), wave columns 0-3 (Y)

  int is_seg = t0 / tiles_per_seg;
  int is_k0 = (t0 - is_seg * tiles_per_seg) * BK_;
  int issued = 0;
  bool need_slow = true;
  auto advance = [&]() {
    if (issued == nt) {
      lx.set_zero();
      ly.set_zero();
    } else if (issued < nt) {
      const bool partial = is_k0 + BK_ > p.Kseg;
      if (need_slow || partial) {
        lx.set_tile(p.X, is_seg, is_k0, p.Kseg);
        ly.set_tile(p.Y, is_seg, is_k0, p.Kseg);
      }
      is_k0 += BK_;
      need_slow = false;
      if (is_k0 >= p.Kseg) {
        is_k0 = 0;
        is_seg = min(is_seg + 1, 1);
        need_slow = true;
      }
    }
    ++issued;
  };
  // prologue: K-tiles 0 and 1, each in need order; K-tile 1 may still be in flight when the loop starts
  const unsigned lds0 = lds_addr_of(lds);
  for (int k = 0; k < (nt > 1 ? 2 : 1); ++k) {
    const unsigned b = lds0 + k * STAGE;
    advance();
    lx.issue_half(b, 0);
    ly.issue_half(b + XT, 0);
    ly.issue_half(b + XT, 1);
    lx.issue_half(b, 1);
  }
  if (nt > 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 fa[2][4], fb[2][2];  // [ks][m-tile of the half], [ks][n-tile of the half]
  // a quarter of a stage is dead once the phase that reads it has passed, so K-tile it + 2 goes into the stage K-tile it is being
  // read from: A(mh0) in phase 2, B(nh1) in phase 3, A(mh1) in phase 4, B(nh0) -- which phase 4 reads again -- in phase 1 of the
  // next iteration; every piece has 3 to 7 phases to land and ONE counted wait per K-tile (phase 4: all but the six instructions
  // of this iteration's phases 2 - 4, i.e. the whole of K-tile it + 1) orders the data; the last two iterations issue nothing
  for (int it = 0; it < nt; ++it) {
    const unsigned char* bx = lds + (it & 1) * STAGE;
    const unsigned char* by = bx + XT;
    const unsigned wc = lds0 + (it & 1) * STAGE, wo = lds0 + ((it + 1) & 1) * STAGE;
    const bool more = it + 2 < nt;
    FH_READ_B(0)
    FH_READ_A(0)
    if (it >= 1 && it + 1 < nt) {
      FH_REFILL_GUARD(RA + RB - 1)
      ly.issue_half(wo + XT, 0);
    }
    FH_PHASE_TAIL("")
    FH_PHASE_MMA(0, 0)
    FH_READ_B(1)
    if (more) {
      advance();
      FH_REFILL_GUARD(RB - 1)
      lx.issue_half(wc, 0);
    }
    FH_PHASE_TAIL("")
    FH_PHASE_MMA(0, 1)
    FH_READ_A(1)
    if (more) {
      FH_REFILL_GUARD(RA - 1)
      ly.issue_half(wc + XT, 1);
    }
    FH_PHASE_TAIL("")
    FH_PHASE_MMA(1, 1)
    FH_READ_B(0)
    if (more) {
      FH_REFILL_GUARD(RB - 1)
      lx.issue_half(wc, 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FH_PHASE_TAIL("")
    FH_PHASE_MMA(1, 0)
  }
  if (!late) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned char* my = lds + wave * EPI_WAVE;
  const int r = lane & 15, q = lane >> 4;
  if (epilogue_bf16<H, 8, KD>(p, acc, 0, my, i0, j0, xi0, yj0, lane, g)) return;
  float bv[8];
  float kd_acc[2] = {0.f, 0.f};
  epilogue_bias(p, j0, yj0, lane, g, bv);
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        *reinterpret_cast<f32x4*>(my + (mm * 16 + r) * EPI_ROW + (n * 16 + 4 * q) * 4) = acc[n][2 * h + mm];
    epilogue_readback<H, KD>(p, my, h, i0, j0, xi0, yj0, lane, g, bv, &kd_acc[h >> 1]);
  }
  if constexpr (KD) {   // (a row tile cut by M: the 64-row blocks that exist)
    kd_write_slot(p, g, i0 + xi0, j0 + yj0, lane, kd_acc[0]);
    kd_write_slot(p, g, i0 + xi0 + 64, j0 + yj0, lane, kd_acc[1]);
  }
}

template <typename H, int XL, int YL, bool KD = false>
__global__ void __launch_bounds__(512) gemm_8ph_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  gemm_8ph_tile<H, XL, YL, KD>(p, lds, logical_block(p.splitk), (int)gridDim.y, (int)blockIdx.z);
}
// ------------------------------------------------------------------ loop 8: the 8-phase loop as a STREAMING persistent launch
// One workgroup per CU walks the tiles f, f + G, f + 2 G, ... (G = grid size, a multiple of 8, so a workgroup's tiles keep their XCD
// class and the column tiles of a row tile stay on one XCD's L2 as in the one-tile form).  In-kernel stamps (s_memtime, K = 768
// tiles) put the one-tile form's time outside the main loop at: first-load latency ~2.8 K cycles, and an epilogue that is
// STORE-ISSUE bound -- a CU retires its 128 KiB of bf16 output at ~14 B / cycle (~9.4 K cycles, every wave blocked in its store
// instructions, the other waves waiting for it at the next barrier), whatever else the epilogue does.  This form puts the next
// tile's start-up UNDER that store phase:
//   epilogue rows 0 - 63 (8 stores per wave)  |  next tile: loaders set up (VALU that runs while the store path drains), its bias
//   row and its first TWO K-tiles issued by LDS-DMA into the tile buffers (dead since the main loop ended)  |  epilogue rows 64 - 127
// so that the data lands while the second half of the stores is issued, and the next main loop starts on resident tiles:
//   * vmcnt counts loads, stores and LDS-DMA in ONE in-order queue: `s_waitcnt vmcnt(n)` with n = the operations issued AFTER the
//     piece waited for (K-tile 1's pieces, the second half's stores) retires it without waiting for those stores'
//     acknowledgement (~2.5 us under load); the first half's stores are older, but ~4 us old by then;
//   * the loop's own counted waits are younger than all stores: its first iteration (K-tiles 0 and 1 resident or on their way)
//     issues the first three quarters of K-tile 2 and waits once -- for K-tile 1, counting those six pieces and the stores behind
//     it --, and by the second iteration's counted wait the stores have been acknowledged;
//   * nothing the compiler can see may be a vector-memory LOAD between the LDS-DMA and the stores (it would be waited for with
//     a count that ignores the asm-issued DMA, i.e. drain them): the bias row comes in by LDS-DMA as well (256 B per wave into a
//     stash next to the staging rows), every other epilogue operand is a kernel argument (scalar loads); the kernels must not
//     spill (scratch accesses are vector memory) -- hence ONE set of loaders, set up when half the accumulators are dead.
// The epilogue is the bf16-staged form (epilogue_bf16) in 16-row passes, staged in the 32 KiB of LDS above the two tile buffers
// (8 waves x (16 rows x 144 B + 256 B bias stash)); launches whose epilogue needs per-row operands, fp32 / atomic output or a
// reduction split are not offered this loop (stream_ok); a row tile cut by M masks its stores and drains them before the next tile
// (an instruction without an active lane may be branched over, so their number is not known).
constexpr int EPS_WAVE = 16 * EPB_ROW + 256;
// Loop 8 writes its output with NON-TEMPORAL stores (global_store_dwordx4 ... nt; round 6): its tiles leave as 128 KB bursts that every CU
// issues at about the same time, nothing re-reads a line before the launch ends, and the consumers' reads come from HBM either way (outputs
// of 61 MB ... 4 GB against 32 MB of L2).  Same box, alternating, B = 256: student step 51.05 -> 50.18 ms, teacher forward 43.24 -> 42.00,
// bench 92.74 -> 91.20 ms per step.  The same switch in the epilogues of loops 3 / 4 / 5 / 7, LayerNorm, the KD loss and the positional conv
// measured neutral, in the attention kernels (whose outputs the next launch re-reads at once) +0.8 ms -- they keep plain stores; a non-temporal
// LDS-DMA for the X rows of one-column-tile products (read once) cost the student step +0.8 ms.
// (compile time on purpose: behind a run-time test the compiler merged the two stores and dropped the hint.)  -DFH_L8_PLAIN_STORES: A/B.
#ifdef FH_L8_PLAIN_STORES
#define FH_L8_ST(ptr, v) st16(ptr, v)
#else
#define FH_L8_ST(ptr, v) st16_nt(ptr, v)
#endif
__device__ __forceinline__ void glds4(const float* gsrc, unsigned dst) {  // 64 lanes x 4 B -> 256 contiguous bytes at LDS address dst
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
// waits until at most n (wave-uniform: 8 or 16 stores, 8 prefetch pieces, their sums; anything else drains) of this wave's
// vector-memory operations are pending
__device__ __forceinline__ void wait_vm_upto(int n) {
  switch (__builtin_amdgcn_readfirstlane(n)) {
#define FH_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    FH_W(6) FH_W(8) FH_W(14) FH_W(16) FH_W(17) FH_W(22) FH_W(24)
#undef FH_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// m-tiles M0 .. M1 - 1 (16 rows each) of this wave's 128 x 64 block: bias / GELU / dropout in the accumulator layout, staged as
// bf16 sixteen rows at a time.  Returns the number of store INSTRUCTIONS issued when that is known exactly (wave-uniform: 0 when
// the wave's 64 columns lie beyond N, 2 -- 4 with a pre-activation copy -- per m-tile for a block whose rows are all inside M:
// every instruction then has an active lane), -1 for a block cut by M (rows past M are masked out, and an instruction without
// active lanes may be branched over).
// OP = 1 (the conv data gradients): the product is multiplied by a per-element operand (the saved GELU', ep.gelu_aux with ep.aux_factor)
// that stream_opnd_issue() brought into the dead tile buffers by LDS-DMA -- read here in the ACCUMULATOR layout, so the output is rounded
// once as in the other loops --, and the output rows follow the batch wrap of an odd frame count (c_rpb / c_bstride, ep.tail_cols).
struct StreamOpnd {
  const unsigned char* opl;  // this wave's operand pieces of the half being written: [m-tile][row half][64 lanes x 16 B]
  long off0;                 // element offset of (group, batch entry of the wave's first row, row 0, column 0) in C / the operand
  long wrap;                 // c_bstride - c_rpb * ldc: added once a row index runs past the batch entry
  int t0, rpb, tail;         // the wave's first row inside its batch entry; rows per entry (0: no wrap); valid columns of an entry's last row (0: all)
};
// global element offset and store predicate of local row d (0 .. 127) of the wave's block, column jr
__device__ __forceinline__ long stream_row_off(const StreamOpnd& o, int d, long ldc, int jr, bool& keep) {
  const int t = o.t0 + d;
  const bool crossed = o.rpb > 0 && t >= o.rpb;
  const int tt = crossed ? t - o.rpb : t;
  keep = !(o.tail > 0 && tt == o.rpb - 1 && jr >= o.tail);
  return o.off0 + (long)t * ldc + (crossed ? o.wrap : 0L) + jr;
}
template <typename H, int M0, int M1, bool PRE, bool GELU, bool DROP, bool BOTH, int OP = 0, typename AccT>
__device__ __forceinline__ int epilogue_stream16_t(const GemmParams& p, const AccT& acc, const float (&b4)[4][4], unsigned char* my, int i0,
                                                   int j0, int xi0, int yj0, int lane, int g, const StreamOpnd* so = nullptr) {
  const GemmEpilogue& ep = p.ep;
  const int r = lane & 15, q = lane >> 4;
  const int jr = j0 + yj0 + 8 * (lane & 7);
  const bool jrv = jr < p.N;
  if (!__builtin_amdgcn_readfirstlane(j0 + yj0 < p.N && i0 + xi0 < p.M)) return 0;
  const bool full = __builtin_amdgcn_readfirstlane(i0 + xi0 + 128 <= p.M);
  const long rstep = 8L * p.ldc;
  const unsigned char* lp = my + (lane >> 3) * EPB_ROW + (lane & 7) * 16;
  unsigned char* wp = my + r * EPB_ROW + 8 * q;
#pragma unroll
  for (int m = M0; m < M1; ++m) {
    const int row = i0 + xi0 + 16 * m + (lane >> 3);
    long base = (long)g * p.c_gstride + (long)row * p.ldc + jr, base1 = base + rstep;
    bool sv0 = jrv && (full || row < p.M), sv1 = jrv && (full || row + 8 < p.M);
    if constexpr (OP != 0) {
      bool k0, k1;
      base = stream_row_off(*so, 16 * m + (lane >> 3), p.ldc, jr, k0);
      base1 = stream_row_off(*so, 16 * m + (lane >> 3) + 8, p.ldc, jr, k1);
      sv0 = sv0 && k0;
      sv1 = sv1 && k1;
    }
    float v[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[n][e] = fmaf(acc[n][m][e], ep.alpha, b4[n][e]);
    if constexpr (OP != 0) {
      // operand of (row 16 m + r, columns 16 n + 4 q ..): piece (m, r >> 3), slot (r & 7) * 8 + (chunk ^ (r & 7)), half q & 1
      const unsigned char* ob = so->opl + ((m - M0) * 2 + (r >> 3)) * 1024 + (r & 7) * 128 + (q & 1) * 8;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float x[4];
        ld4(reinterpret_cast<const H*>(ob + (((2 * n + (q >> 1)) ^ (r & 7)) << 4)), x);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[n][e] *= x[e];
      }
    }
    if (PRE) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float sv[4];  // the pre-activation, or (BOTH) GELU' of it while v becomes the activation
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sv[e] = v[n][e];
          if (BOTH) gelu_both_t<H>(v[n][e], v[n][e], sv[e]);
        }
        uint2 pk;
        pk.x = pack2<H>(sv[0], sv[1]);
        pk.y = pack2<H>(sv[2], sv[3]);
        *reinterpret_cast<uint2*>(wp + n * 32) = pk;
      }
      const V16 t0 = ld16(lp), t1 = ld16(lp + 8 * EPB_ROW);
      if (sv0) FH_L8_ST(reinterpret_cast<bf16*>(ep.preact) + base, t0);
      if (sv1) FH_L8_ST(reinterpret_cast<bf16*>(ep.preact) + base1, t1);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (GELU && !BOTH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[n][e] = gelu_fast_f(v[n][e]);
      }
      if (DROP) {
        const unsigned long long didx = ((unsigned long long)g * p.M + (i0 + xi0 + 16 * m + r)) * (unsigned long long)p.N +
                                        (j0 + yj0 + 16 * n + 4 * q);
        drop_scale_vec<4>(ep.drop, didx, v[n]);
      }
      uint2 pk;
      pk.x = pack2<H>(v[n][0], v[n][1]);
      pk.y = pack2<H>(v[n][2], v[n][3]);
      *reinterpret_cast<uint2*>(wp + n * 32) = pk;
    }
    const V16 t0 = ld16(lp), t1 = ld16(lp + 8 * EPB_ROW);
    if (sv0) FH_L8_ST(reinterpret_cast<bf16*>(p.C) + base, t0);
    if (sv1) FH_L8_ST(reinterpret_cast<bf16*>(p.C) + base1, t1);
  }
  return full ? (M1 - M0) * (PRE ? 4 : 2) : -1;
}
// the option tests hoisted out of the unrolled passes: each instance is straight-line code the scheduler can overlap across passes
template <typename H, int M0, int M1, int OP = 0, typename AccT>
__device__ __forceinline__ int epilogue_stream16(const GemmParams& p, const AccT& acc, const float (&b4)[4][4], unsigned char* my, int i0,
                                                 int j0, int xi0, int yj0, int lane, int g, const StreamOpnd* so = nullptr) {
  if constexpr (OP != 0) return epilogue_stream16_t<H, M0, M1, false, false, false, false, OP>(p, acc, b4, my, i0, j0, xi0, yj0, lane, g, so);
  int opt = (p.ep.preact ? 4 : 0) | (p.ep.act == FH_ACT_GELU ? 2 : 0) | (p.ep.drop.p > 0.f ? 1 : 0);
  if ((opt & 6) == 6 && p.ep.aux_factor) opt |= 8;  // the saved tensor is GELU'(pre-activation)
  switch (__builtin_amdgcn_readfirstlane(opt)) {
#define FH_E(O) case O: return epilogue_stream16_t<H, M0, M1, ((O) & 4) != 0, ((O) & 2) != 0, ((O) & 1) != 0, ((O) & 8) != 0>(p, acc, b4, my, i0, j0, xi0, yj0, lane, g);
    FH_E(1) FH_E(2) FH_E(3) FH_E(4) FH_E(5) FH_E(6) FH_E(7) FH_E(14) FH_E(15)
#undef FH_E
    default: return epilogue_stream16_t<H, M0, M1, false, false, false, false>(p, acc, b4, my, i0, j0, xi0, yj0, lane, g);
  }
}

template <typename H, int XL, int YL, int OP = 0>
__global__ void __launch_bounds__(512) gemm_8ph_stream_kernel(const GemmParams p, int tiles, int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int BK_ = 64, XT = 256 * 64 * 2, STAGE = 2 * XT;
  constexpr int RA = XL == FH_R_STRIDED ? 16 : 8, RB = YL == FH_R_STRIDED ? 8 : 4;  // LDS reads of an A / B fragment phase
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = wave >= 4;
  const int tiles_j = (p.N + 255) / 256;
  const int tiles_per_seg = (p.Kseg + BK_ - 1) / BK_;
  const int xi0 = (wave >> 2) * 128, yj0 = (wave & 3) * 64;
  const unsigned lds0 = lds_addr_of(lds);
  unsigned char* my = lds + 2 * STAGE + wave * EPS_WAVE;
  const float* stash = reinterpret_cast<const float*>(my + 16 * EPB_ROW);
  const unsigned stash_lds = lds0 + 2 * STAGE + wave * EPS_WAVE + 16 * EPB_ROW;
  const bool has_bias = p.ep.bias != nullptr;

  QuarterLoader<XL> lx;
  QuarterLoader<YL> ly;
  int i0 = 0, j0 = 0, g = 0, nt = 0;
  int is_seg = 0, is_k0 = 0, issued = 0;
  bool need_slow = true;
  auto advance = [&]() {
    if (issued == nt) {
      lx.set_zero();
      ly.set_zero();
    } else if (issued < nt) {
      const bool partial = is_k0 + BK_ > p.Kseg;
      if (need_slow || partial) {
        lx.set_tile(p.X, is_seg, is_k0, p.Kseg);
        ly.set_tile(p.Y, is_seg, is_k0, p.Kseg);
      }
      is_k0 += BK_;
      need_slow = false;
      if (is_k0 >= p.Kseg) {
        is_k0 = 0;
        is_seg = min(is_seg + 1, 1);
        need_slow = true;
      }
    }
    ++issued;
  };
  // tile f becomes the current tile: loaders set up, its bias row and its first two K-tiles issued (1 + 8 + 8 pieces at most,
  // oldest first; the quarters of a K-tile in the order the phases need them)
  // (OP: only K-tile 0 goes out between the epilogue halves -- the second half's operand pieces still sit in K-tile 1's buffer)
  auto start_tile = [&](int f, int nk) {
    g = f / tiles;
    const int lb = xcd_remap(f - g * tiles, tiles);
    int ti, tj;
    tile_coords(lb, tiles / tiles_j, tiles_j, p.Kseg * p.X.nseg, ti, tj);
    i0 = ti * 256;
    j0 = tj * 256;
    nt = tiles_per_seg * tile_nseg(p, j0);
    if (has_bias) {
      const int j = j0 + yj0 + lane;
      const float* bp = j < p.N ? p.ep.bias + (long)g * p.ep.bias_gstride + (p.ep.bias_mod > 0 ? (j % p.ep.bias_mod) : j)
                                : reinterpret_cast<const float*>(g_zero_chunk);
      glds4(bp, stash_lds);
    }
    lx.init(p.X, g, i0, p.M, tid, true);
    ly.init(p.Y, g, j0, p.N, tid, false);
    is_seg = 0;
    is_k0 = 0;
    issued = 0;
    need_slow = true;
    for (int k = 0; k < (nt > 1 ? nk : 1); ++k) {
      const unsigned b = lds0 + k * STAGE;
      advance();
      lx.issue_half(b, 0);
      ly.issue_half(b + XT, 0);
      ly.issue_half(b + XT, 1);
      lx.issue_half(b, 1);
    }
  };
  auto issue_ktile1 = [&]() {   // (OP) K-tile 1 of the current tile, behind the second epilogue half
    if (nt > 1) {
      const unsigned b = lds0 + STAGE;
      advance();
      lx.issue_half(b, 0);
      ly.issue_half(b + XT, 0);
      ly.issue_half(b + XT, 1);
      lx.issue_half(b, 1);
    }
  };

  int f = blockIdx.x;
  if (f >= total) return;
  start_tile(f, 2);
  int young1 = 0;   // (OP) operations younger than K-tile 1's last piece: none, it is issued behind the stores
  int younger = 0;  // vector-memory operations this wave issued AFTER the last prefetch piece (stores of the previous tile's second half); < 0: drain
  f32x4 acc[4][8];  // [n][m]
  bf16x8 fa[2][4], fb[2][2];
  for (;;) {
    // K-tile 0 (and the bias row) landed in every wave; K-tile 1 may still be in flight
    wait_vm_upto(younger < 0 ? 0 : younger + (nt > 1 ? 8 : 0));
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a quarter of a stage is dead once the phase that reads it has passed, so K-tile it + 2 goes into the stage K-tile it is being
    // read from: A(mh0) in phase 2, B(nh1) in phase 3, A(mh1) in phase 4, B(nh0) in phase 1 of the next iteration (phase 4 reads it
    // again) -- every piece has 3 to 7 phases to land, and ONE counted wait per K-tile (phase 4: all but the six instructions of
    // this iteration's phases 2 - 4, i.e. the whole of K-tile it + 1) orders the data; the last two iterations issue nothing
    {
      const unsigned char* bx = lds;
      const unsigned char* by = bx + XT;
      const bool more = nt > 2;
      FH_READ_B(0)
      FH_READ_A(0)
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(0, 0)
      FH_READ_B(1)
      if (more) {
        advance();
        FH_REFILL_GUARD(RB - 1)
        lx.issue_half(lds0, 0);
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(0, 1)
      FH_READ_A(1)
      if (more) {
        FH_REFILL_GUARD(RA - 1)
        ly.issue_half(lds0 + XT, 1);
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(1, 1)
      FH_READ_B(0)
      if (more) {
        FH_REFILL_GUARD(RB - 1)
        lx.issue_half(lds0, 1);
      }
      wait_vm_upto(younger < 0 ? 0 : (OP != 0 ? young1 : younger) + (more ? 6 : 0));
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(1, 0)
    }
    for (int it = 1; it < nt; ++it) {
      const unsigned char* bx = lds + (it & 1) * STAGE;
      const unsigned char* by = bx + XT;
      const unsigned wc = lds0 + (it & 1) * STAGE, wo = lds0 + ((it + 1) & 1) * STAGE;
      const bool more = it + 2 < nt;
      FH_READ_B(0)
      FH_READ_A(0)
      if (it + 1 < nt) {
        FH_REFILL_GUARD(RA + RB - 1)
        ly.issue_half(wo + XT, 0);
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(0, 0)
      FH_READ_B(1)
      if (more) {
        advance();
        FH_REFILL_GUARD(RB - 1)
        lx.issue_half(wc, 0);
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(0, 1)
      FH_READ_A(1)
      if (more) {
        FH_REFILL_GUARD(RA - 1)
        ly.issue_half(wc + XT, 1);
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(1, 1)
      FH_READ_B(0)
      if (more) {
        FH_REFILL_GUARD(RB - 1)
        lx.issue_half(wc, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      FH_PHASE_TAIL("")
      FH_PHASE_MMA(1, 0)
    }
    if (!late) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // both tile buffers are dead in every wave from here on
    const int fn = f + (int)gridDim.x;
    const bool has_next = fn < total;
    const int ci0 = i0, cj0 = j0, cg = g;
    float b4[4][4];
    {
      const int q = lane >> 4;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (has_bias) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(stash + 16 * n + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) b4[n][e] = t[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b4[n][e] = 0.f;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stash is rewritten by the next tile's bias piece
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (OP != 0) {
      // The operand tile of this wave's 128 x 64 block comes into the dead tile buffers by LDS-DMA: 16 pieces (m-tile, row half) of
      // 64 lanes x 16 B -- the first half's eight into stage 0, the second half's into stage 1, 8 KiB per wave each --, all of them
      // OLDER than any store of the tile, so waiting for them never waits for a store acknowledgement.  Lane i fetches row i >> 3,
      // column chunk (i & 7) ^ (i >> 3): the accumulator-layout reads of the epilogue then spread over the banks.
      StreamOpnd so;
      {
        const RowMap rm = map_row(ci0 + xi0, p.c_rpb);
        so.off0 = (long)cg * p.c_gstride + (long)rm.b * (p.c_rpb > 0 ? p.c_bstride : 0L);
        so.wrap = p.c_rpb > 0 ? p.c_bstride - (long)p.c_rpb * p.ldc : 0L;
        so.t0 = rm.t; so.rpb = p.c_rpb; so.tail = p.ep.tail_cols;
      }
      const bool live_wave = __builtin_amdgcn_readfirstlane(cj0 + yj0 < p.N && ci0 + xi0 < p.M);
      if (live_wave) {
        const int rho = lane >> 3, jr = cj0 + yj0 + 8 * ((lane & 7) ^ rho);
        const H* aux = reinterpret_cast<const H*>(p.ep.gelu_aux);
        const bool jv = jr < p.N;
        const int mrows = p.M - (ci0 + xi0);          // rows of the block inside M (>= 1)
        unsigned dst = lds0 + wave * 8192;
#pragma unroll 1
        for (int mh = 0; mh < 16; ++mh) {             // (rolled: sixteen unrolled 64-bit address chains cost the kernel its last registers)
          const int d = 8 * mh + rho;                 // local row of piece mh = 2 m + half
          bool keep;
          const long off = stream_row_off(so, d, p.ldc, jr, keep);
          const bool v = keep && jv && d < mrows;
          glds16(v ? reinterpret_cast<const unsigned char*>(aux + off) : reinterpret_cast<const unsigned char*>(g_zero_chunk), dst);
          dst += mh == 7 ? (unsigned)(STAGE - 7 * 1024) : 1024u;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the first half's pieces
      }
      so.opl = lds + wave * 8192;
      const int n1 = epilogue_stream16<H, 0, 4, OP>(p, acc, b4, my, ci0, cj0, xi0, yj0, lane, cg, &so);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave has read its stage-0 pieces: K-tile 0 may land there
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) start_tile(fn, 1);
      __builtin_amdgcn_sched_barrier(0);
      // the second half's pieces are older than the first half's stores and the K-tile-0 pieces issued since
      if (live_wave) wait_vm_upto(n1 < 0 ? 0 : n1 + (has_next ? 8 + (has_bias ? 1 : 0) : 0));
      so.opl = lds + STAGE + wave * 8192;
      const int n2 = epilogue_stream16<H, 4, 8, OP>(p, acc, b4, my, ci0, cj0, xi0, yj0, lane, cg, &so);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and its stage-1 pieces: K-tile 1 may land
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) issue_ktile1();
      __builtin_amdgcn_sched_barrier(0);
      // K-tile 0 is older than the second half's stores and K-tile 1's pieces; K-tile 1 is younger than every store (young1 = 0): the
      // first iteration's counted wait for it also waits for those stores' acknowledgements
      younger = (n1 < 0 || n2 < 0) ? -1 : n2;
    } else {
    const int n1 = epilogue_stream16<H, 0, 4>(p, acc, b4, my, ci0, cj0, xi0, yj0, lane, cg);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) start_tile(fn, 2);
    __builtin_amdgcn_sched_barrier(0);
    const int n2 = epilogue_stream16<H, 4, 8>(p, acc, b4, my, ci0, cj0, xi0, yj0, lane, cg);
    __builtin_amdgcn_sched_barrier(0);
    younger = (n1 < 0 || n2 < 0) ? -1 : n2;
    }
    if (!has_next) break;
    f = fn;
  }
}

template <typename H> int launch_8ph(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + 255) / 256, tiles_j = (p.N + 255) / 256;
  dim3 grid(tiles_i * tiles_j, p.splitk & 0xffff, p.groups), block(512);
  const size_t smem = 131072;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_8ph_kernel<H, FH_R_STRIDED, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_8ph_kernel<H, FH_R_STRIDED, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int xl = p.X.layout, yl = p.Y.layout;
  if (p.ep.kd_teacher) {   // the KD-loss epilogue is compiled into its own instance (32 more live registers in the read-back)
    if (xl != FH_R_CONTIG || yl != FH_R_CONTIG) return -113;
    static bool kd_attr = false;
    if (!kd_attr) {
      hipFuncSetAttribute((const void*)gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_CONTIG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      kd_attr = true;
    }
    hipLaunchKernelGGL((gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_CONTIG, true>), grid, block, smem, stream, p);
    FH_CHECK_LAUNCH();
    return 0;
  }
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_CONTIG>), grid, block, smem, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm_8ph_kernel<H, FH_R_CONTIG, FH_R_STRIDED>), grid, block, smem, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_8ph_kernel<H, FH_R_STRIDED, FH_R_CONTIG>), grid, block, smem, stream, p);
  else
    hipLaunchKernelGGL((gemm_8ph_kernel<H, FH_R_STRIDED, FH_R_STRIDED>), grid, block, smem, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}

// ... and its operand instance (OP = 1): the conv data gradients -- product x saved GELU' (ep.gelu_aux with ep.aux_factor), nothing else in
// the epilogue, both operands R_CONTIG; the output rows may wrap per batch entry (c_rpb / c_bstride) and an entry's last row may be
// narrower (ep.tail_cols), in whole 16-byte chunks; a wave's 128 rows cross one entry boundary at most
bool stream_opnd_ok(const GemmParams& p) {
  static const bool on = [] { const char* e = getenv("FH_GEMM_STREAM_OPND"); return !e || e[0] != '0'; }();
  const GemmEpilogue& e = p.ep;
  return on && e.gelu_aux && e.aux_factor && !e.residual && !e.preact && e.act == FH_ACT_NONE && e.drop.p <= 0.f && !e.bias && !e.atomic && !e.row_len &&
         !e.out_f32 && !e.kd_teacher && !e.c0_wpart && (p.splitk & 0xffff) <= 1 && p.X.layout == FH_R_CONTIG && p.Y.layout == FH_R_CONTIG &&
         (p.c_rpb <= 0 || p.c_rpb >= 128) && p.N % 8 == 0 && e.tail_cols % 8 == 0 && p.ldc % 8 == 0 && (p.c_rpb <= 0 || p.c_bstride % 8 == 0);
}
// launches loop 8 can take: the bf16-staged epilogue without per-row operands, one reduction split
bool stream_ok(const GemmParams& p) {
  // TIMING-ONLY debug switch (wrong results; compiled only with -DFH_DEBUG_BUILD): offer loop 8 to launches with a GELU' operand / row wrap / tail columns, ignoring them --
  // an upper bound of what a streaming persistent data-gradient launch could gain
#ifdef FH_DEBUG_BUILD   // never in the shipped library: an inherited environment variable must not be able to produce wrong gradients
  static const int force = [] { const char* e = getenv("FH_DEBUG_L8_FORCE"); return (e && e[0] == '1') ? 1 : 0; }();
  if (force) return !p.ep.atomic && !p.ep.out_f32 && !p.ep.kd_teacher && !p.ep.c0_wpart && !p.ep.residual && !p.ep.row_len && (p.splitk & 0xffff) <= 1;
#endif
  if (stream_opnd_ok(p)) return true;
  const long wrap0 = p.c_bstride - (long)p.c_rpb * p.ldc;
  return !p.ep.atomic && !p.ep.gelu_aux && !p.ep.residual && !p.ep.row_len && !p.ep.out_f32 && !p.ep.tail_cols && !p.ep.kd_teacher && !p.ep.c0_wpart &&
         (p.c_rpb <= 0 || wrap0 == 0) && (p.splitk & 0xffff) <= 1;
}
template <typename H> int launch_8ph_stream(const GemmParams& p, hipStream_t stream) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -121;
    ncu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount & ~7 : 8;
  }
  const int tiles_i = (p.M + 255) / 256, tiles_j = (p.N + 255) / 256, tiles = tiles_i * tiles_j;
  const long total = (long)tiles * p.groups;
  if (total > 0x7fffffffL) return -122;
  // a persistent workgroup owns its CU's registers and LDS for the whole launch (up to milliseconds): with fh_gemm_reserve_cus the
  // grid leaves CUs free for kernels that must make progress meanwhile (the collective kernels of a data-parallel run)
  const int sres = fh_stream_reserve_cus(stream);
  int pgrid = (ncu - (sres > fh_g_reserve_cus ? sres : fh_g_reserve_cus)) & ~7;
  if (pgrid < 8) pgrid = 8;
  dim3 grid((unsigned)(total < pgrid ? total : pgrid), 1, 1), block(512);
  const size_t smem = 131072 + 8 * EPS_WAVE;
  typedef void (*kern_t)(const GemmParams, int, int);
  static const kern_t kerns[2][2] = {{gemm_8ph_stream_kernel<H, FH_R_CONTIG, FH_R_CONTIG>, gemm_8ph_stream_kernel<H, FH_R_CONTIG, FH_R_STRIDED>},
                                     {gemm_8ph_stream_kernel<H, FH_R_STRIDED, FH_R_CONTIG>, gemm_8ph_stream_kernel<H, FH_R_STRIDED, FH_R_STRIDED>}};
  static bool attr_set = false;
  if (!attr_set) {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) hipFuncSetAttribute((const void*)kerns[a][b], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  if (stream_opnd_ok(p)) {
    static bool op_attr = false;
    if (!op_attr) {
      hipFuncSetAttribute((const void*)gemm_8ph_stream_kernel<H, FH_R_CONTIG, FH_R_CONTIG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      op_attr = true;
    }
    hipLaunchKernelGGL((gemm_8ph_stream_kernel<H, FH_R_CONTIG, FH_R_CONTIG, 1>), grid, block, smem, stream, p, tiles, (int)total);
    FH_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(kerns[p.X.layout == FH_R_STRIDED][p.Y.layout == FH_R_STRIDED], grid, block, smem, stream, p, tiles, (int)total);
  FH_CHECK_LAUNCH();
  return 0;
}

template <typename H, int BM_, int BN_, int WM, int WN, int BK_, int NST> int launch_ring(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + BM_ - 1) / BM_, tiles_j = (p.N + BN_ - 1) / BN_;
  dim3 grid(tiles_i * tiles_j, p.splitk & 0xffff, p.groups), block(WM * WN * 64);
  const size_t smem = (size_t)NST * (BM_ + BN_) * BK_ * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_CONTIG, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_CONTIG, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_STRIDED, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_STRIDED, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int xl = p.X.layout, yl = p.Y.layout;
  if (p.ep.c0_wpart) {   // conv-0 weight-gradient epilogue: its own instance of the 4-wave 256 x 128 build (X contiguous, Y strided)
    if constexpr (BM_ == 256 && BN_ == 128 && WM == 2 && WN == 2) {
      if (xl != FH_R_CONTIG || yl != FH_R_STRIDED) return -113;
      static bool c0_attr = false;
      if (!c0_attr) {
        hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_CONTIG, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        c0_attr = true;
      }
      hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_CONTIG, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST, false, true>), grid, block, smem, stream, p);
      FH_CHECK_LAUNCH();
      return 0;
    } else {
      return -113;
    }
  }
  if (p.ep.kd_teacher) {   // KD-loss epilogue: its own instance
    if (xl != FH_R_CONTIG || yl != FH_R_CONTIG) return -113;
    static bool kd_attr = false;
    if (!kd_attr) {
      hipFuncSetAttribute((const void*)gemm_ring_kernel<H, FH_R_CONTIG, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      kd_attr = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_CONTIG, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST, true>), grid, block, smem, stream, p);
    FH_CHECK_LAUNCH();
    return 0;
  }
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_CONTIG, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST>), grid, block, smem, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_CONTIG, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST>), grid, block, smem, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_STRIDED, FH_R_CONTIG, BM_, BN_, WM, WN, BK_, NST>), grid, block, smem, stream, p);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<H, FH_R_STRIDED, FH_R_STRIDED, BM_, BN_, WM, WN, BK_, NST>), grid, block, smem, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}
bool ring_ok(const GemmParams& p, int bk) {
  if (p.X.layout == FH_R_STRIDED && p.X.rpb > 0 && p.X.rpb < bk) return false;
  if (p.Y.layout == FH_R_STRIDED && p.Y.rpb > 0 && p.Y.rpb < bk) return false;
  return true;
}

// ------------------------------------------------------------------ 16-bit main loops by number (fh_gemm_set_path numbering)
template <typename H> int launch_path_h(const GemmParams& p, int path, hipStream_t stream) {
  fh_g_last_loop = 2;
  fh_g_last_splitk = p.splitk;
  if (path == 0 || (path == 3 && ring_ok(p, 64)) || (path >= 4 && path <= 6 && ring_ok(p, 32)) ||
      (path == 7 && ring_ok(p, 64)))
    fh_g_last_loop = path;
  if (path == 1 && small_ok(p)) { fh_g_last_loop = 1; return launch_small<H>(p, stream); }
  if (path == 0) return launch_t<H>(p, stream);
  if (path == 3 && ring_ok(p, 64)) return launch_ring<H, 256, 256, 2, 4, 64, 2>(p, stream);
  if (path == 4 && ring_ok(p, 32)) return launch_ring<H, 256, 128, 2, 2, 32, 3>(p, stream);
  if (path == 5 && ring_ok(p, 32)) return launch_ring<H, 256, 256, 2, 4, 32, 4>(p, stream);
  if (path == 6 && ring_ok(p, 32)) return launch_pp<H>(p, stream);
  if (path == 7 && ring_ok(p, 64)) return launch_8ph<H>(p, stream);
  if (path == 8 && ring_ok(p, 64) && stream_ok(p)) { fh_g_last_loop = 8; return launch_8ph_stream<H>(p, stream); }
  // a table entry "loop 8" reached by a launch the streaming loop cannot take (a neighbour class's choice): its non-persistent twin
  if (path == 8 && ring_ok(p, 64)) { fh_g_last_loop = 7; return launch_8ph<H>(p, stream); }
  return launch_32<H>(p, stream);
}

}  // namespace
