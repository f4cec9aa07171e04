// Generic MFMA GEMM for gfx950 (see gemm.h).  128x128 output tile, 4 waves (2x2), each wave a
// 64x64 sub-tile of 16x16 MFMA blocks; register-prefetched global->LDS staging; fp32 accumulate.
//
//   T = bf16 : v_mfma_f32_16x16x32_bf16, BK = 64
//   T = float: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), BK = 32
//
// MFMA orientation: the Y operand (output column j) is the MFMA "A" matrix and the X operand
// (output row i) the MFMA "B" matrix, so a lane's 4 accumulator registers are 4 CONSECUTIVE j
// of one output row -> 8/16-byte vector stores and vector bias/residual loads in the epilogue.
#include <cstdlib>
#include <utility>
#include <vector>

#include "gemm.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int LDS_C_STRIDE = 144;  // R_CONTIG tile: 128 rows x (128 B of k + 16 B pad)
constexpr int TILE_BYTES = 18432;  // max over layouts

template <typename T> struct StridedCfg;
template <> struct StridedCfg<bf16> { static constexpr int STRIDE = 288; };   // 128 el * 2 B + 32
template <> struct StridedCfg<float> { static constexpr int STRIDE = 528; };  // 128 el * 4 B + 16

struct RowMap {
  int b, t;
};
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
        const bool v = rv && (s < o.nseg) && (o.tvalid == 0 || (ts >= 0 && ts < o.tvalid));
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
      if (colvalid && r < Kseg) {
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
};

// ------------------------------------------------------------------ LDS -> MFMA fragments
// bf16: 8 k-values per lane per k32-step, k = ks*32 + 8*(lane>>4) + e
template <int LAYOUT> __device__ __forceinline__ bf16x8 frag_bf16(const unsigned char* lds, int row0, int ks, int lane);
template <> __device__ __forceinline__ bf16x8 frag_bf16<FH_R_CONTIG>(const unsigned char* lds, int row0, int ks, int lane) {
  const unsigned char* a = lds + (row0 + (lane & 15)) * LDS_C_STRIDE + ks * 64 + (lane >> 4) * 16;
  return *reinterpret_cast<const bf16x8*>(a);
}
template <> __device__ __forceinline__ bf16x8 frag_bf16<FH_R_STRIDED>(const unsigned char* lds, int row0, int ks, int lane) {
  // ds_read_b64_tr_b16: within a 16-lane group, lane 4*qq+p supplies the address of row qq, columns 4p..4p+3
  // of a 4x16 block; lane i receives column i of the 4 rows.
  constexpr int STRIDE = StridedCfg<bf16>::STRIDE;
  const int i = lane & 15, q = lane >> 4;
  const int qq = i >> 2, pp = i & 3;
  const unsigned char* a0 = lds + (ks * 32 + 8 * q + qq) * STRIDE + (row0 + 4 * pp) * 2;
  const unsigned char* a1 = a0 + 4 * STRIDE;
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

template <typename T, int XL, int YL> struct TileMath;
template <int XL, int YL> struct TileMath<bf16, XL, YL> {
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
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[n][m], 0, 0, 0);
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

// ------------------------------------------------------------------ epilogue (shared by both main loops)
template <typename T>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], int i0, int j0, int xi0, int yj0,
                                              int lane, int g) {
  const GemmEpilogue& ep = p.ep;
  const int jq = (lane >> 4) * 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = i0 + xi0 + 16 * m + (lane & 15);
    if (i >= p.M) continue;
    RowMap rm = map_row(i, p.c_rpb);
    const long crow = (long)g * p.c_gstride + (long)rm.b * p.c_bstride + (long)rm.t * p.ldc;
    bool zero_row = false;
    if (ep.row_len) zero_row = rm.t >= ep.row_len[rm.b];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int j = j0 + yj0 + 16 * n + jq;
      if (j >= p.N) continue;
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
      if (ep.preact) st4(reinterpret_cast<T*>(ep.preact) + coff, v);
      if (ep.act == FH_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if (ep.drop.p > 0.f) {
        const unsigned long long didx = ((unsigned long long)g * p.M + i) * (unsigned long long)p.N + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_scale(ep.drop, didx + e);
      }
      if (ep.gelu_aux) {
        float a[4];
        ld4(reinterpret_cast<const T*>(ep.gelu_aux) + coff, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(a[e]);
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

// ------------------------------------------------------------------ the kernel
template <typename T, int XL, int YL>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int BK = 8 * VEC;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
  unsigned char* ldsX = lds;
  unsigned char* ldsY = lds + TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN - 1) / BN;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
  const int i0 = ti * BM, j0 = tj * BN;

  const int tiles_per_seg = (p.Kseg + BK - 1) / BK;
  const int ntiles = tiles_per_seg * p.X.nseg;
  const int per_split = (ntiles + p.splitk - 1) / p.splitk;
  const int t0 = blockIdx.y * per_split;
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

  gemm_epilogue<T>(p, acc, i0, j0, xi0, yj0, lane, g);
}

// ------------------------------------------------------------------ bf16 main loop with direct-to-LDS loads (LDS-DMA)
// global_load_lds_dwordx4 moves 16 B per lane straight into LDS (no VGPR staging, no ds_write pass): one wave
// instruction fills 1 KiB of LINEAR LDS.  Tiles are therefore unpadded; bank conflicts are removed with an XOR swizzle
// applied to the per-lane SOURCE address and, identically, to the fragment reads (guide rule 21).  Double-buffered:
// tile t+1 streams in while tile t feeds the MFMAs; one barrier per K-tile.
__device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int strided_key(int kl) { return ((kl & 3) << 1) | (((kl >> 3) & 1) << 3); }

template <int LAYOUT> struct DmaLoader;

template <> struct DmaLoader<FH_R_CONTIG> {  // LDS tile [128 rows][128 B]; 16-B chunk c of row R stored at chunk c ^ ((R>>1)&7)
  const bf16* base;
  long off[4];
  long segdelta;
  unsigned vmask;
  int kch[4];  // logical k-chunk (8 elements) this lane fetches for its i-th wave instruction
  int wave;
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = tid >> 6;
    const int lane = tid & 63;
    vmask = 0;
    segdelta = (long)(o.tshift[1] - o.tshift[0]) * o.ld + (o.segoff[1] - o.segoff[0]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = wave * 32 + u * 8 + (lane >> 3);
      kch[u] = (lane & 7) ^ ((row >> 1) & 7);
      const int idx = idx0 + row;
      const bool rv = idx < idx_limit;
      RowMap rm = map_row(rv ? idx : 0, o.rpb);
      off[u] = (long)rm.b * o.bstride + (long)(rm.t + o.tshift[0]) * o.ld + o.segoff[0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ts = rm.t + o.tshift[s];
        const bool v = rv && (s < o.nseg) && (o.tvalid == 0 || (ts >= 0 && ts < o.tvalid));
        vmask |= (v ? 1u : 0u) << (2 * u + s);
      }
    }
  }
  __device__ __forceinline__ void issue(const GemmOperand& o, int seg, int k0, int Kseg, unsigned char* tile) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + kch[u] * 8;
      const bool v = (k < Kseg) && ((vmask >> (2 * u + seg)) & 1u);
      const void* src = v ? (const void*)(base + off[u] + (long)seg * segdelta + k) : (const void*)g_zero_chunk;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + (wave * 4 + u) * 1024), 16, 0, 0);
    }
  }
};

template <> struct DmaLoader<FH_R_STRIDED> {  // LDS tile [64 k][256 B]; chunk c of k-row kl stored at chunk c ^ strided_key(kl)
  const bf16* base;
  int col[4];
  int kl[4];
  unsigned cvalid;
  int wave;
  __device__ __forceinline__ void init(const GemmOperand& o, int g, int idx0, int idx_limit, int tid) {
    base = reinterpret_cast<const bf16*>(o.p) + (long)g * o.gstride;
    wave = tid >> 6;
    const int lane = tid & 63;
    cvalid = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kl[u] = (wave * 4 + u) * 4 + (lane >> 4);
      const int c = (lane & 15) ^ strided_key(kl[u]);
      col[u] = idx0 + c * 8;
      cvalid |= (col[u] < idx_limit ? 1u : 0u) << u;
    }
  }
  __device__ __forceinline__ void issue(const GemmOperand& o, int seg, int k0, int Kseg, unsigned char* tile) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = k0 + kl[u];
      const void* src = (const void*)g_zero_chunk;
      if (((cvalid >> u) & 1u) && r < Kseg) {
        RowMap rm = map_row(r, o.rpb);
        src = base + o.segoff[seg] + (long)rm.b * o.bstride + (long)rm.t * o.ld + col[u];
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(tile + (wave * 4 + u) * 1024), 16, 0, 0);
    }
  }
};

template <int LAYOUT> __device__ __forceinline__ bf16x8 frag_dma(const unsigned char* lds, int row0, int ks, int lane);
template <> __device__ __forceinline__ bf16x8 frag_dma<FH_R_CONTIG>(const unsigned char* lds, int row0, int ks, int lane) {
  const int R = row0 + (lane & 15);
  const int phys = (ks * 4 + (lane >> 4)) ^ ((R >> 1) & 7);
  return *reinterpret_cast<const bf16x8*>(lds + R * 128 + phys * 16);
}
template <> __device__ __forceinline__ bf16x8 frag_dma<FH_R_STRIDED>(const unsigned char* lds, int row0, int ks, int lane) {
  const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
  const int kl = ks * 32 + 8 * q + qq;
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

template <int XL, int YL>
__global__ void __launch_bounds__(256) gemm_dma_kernel(const GemmParams p) {
  constexpr int BK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 buffers x (X tile 16 KiB + Y tile 16 KiB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int tiles_j = (p.N + BN - 1) / BN;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
  const int i0 = ti * BM, j0 = tj * BN;
  const int tiles_per_seg = (p.Kseg + BK - 1) / BK;
  const int ntiles = tiles_per_seg * p.X.nseg;
  const int per_split = (ntiles + p.splitk - 1) / p.splitk;
  const int t0 = blockIdx.y * per_split;
  const int t1 = min(ntiles, t0 + per_split);
  if (t0 >= t1) return;

  DmaLoader<XL> lx;
  DmaLoader<YL> ly;
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
    lx.issue(p.X, seg, k0, p.Kseg, lds);
    ly.issue(p.Y, seg, k0, p.Kseg, lds + 16384);
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    unsigned char* bx = lds + cur * 32768;
    unsigned char* by = bx + 16384;
    if (t + 1 < t1) {
      const int seg = (t + 1) / tiles_per_seg, k0 = (t + 1 - seg * tiles_per_seg) * BK;
      lx.issue(p.X, seg, k0, p.Kseg, lds + (cur ^ 1) * 32768);
      ly.issue(p.Y, seg, k0, p.Kseg, lds + (cur ^ 1) * 32768 + 16384);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fx[4], fy[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) fx[m] = frag_dma<XL>(bx, xi0 + 16 * m, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n) fy[n] = frag_dma<YL>(by, yj0 + 16 * n, ks, lane);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[n], fx[m], acc[n][m], 0, 0, 0);
    }
    __syncthreads();
  }
  gemm_epilogue<bf16>(p, acc, i0, j0, xi0, yj0, lane, g);
}

template <typename T> int launch_t(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + BM - 1) / BM, tiles_j = (p.N + BN - 1) / BN;
  dim3 grid(tiles_i * tiles_j, p.splitk, p.groups), block(256);
  const int xl = p.X.layout, yl = p.Y.layout;
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

int launch_dma(const GemmParams& p, hipStream_t stream) {
  const int tiles_i = (p.M + BM - 1) / BM, tiles_j = (p.N + BN - 1) / BN;
  dim3 grid(tiles_i * tiles_j, p.splitk, p.groups), block(256);
  const size_t smem = 65536;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_dma_kernel<FH_R_CONTIG, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_dma_kernel<FH_R_CONTIG, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_dma_kernel<FH_R_STRIDED, FH_R_CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)gemm_dma_kernel<FH_R_STRIDED, FH_R_STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int xl = p.X.layout, yl = p.Y.layout;
  if (xl == FH_R_CONTIG && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_dma_kernel<FH_R_CONTIG, FH_R_CONTIG>), grid, block, smem, stream, p);
  else if (xl == FH_R_CONTIG && yl == FH_R_STRIDED)
    hipLaunchKernelGGL((gemm_dma_kernel<FH_R_CONTIG, FH_R_STRIDED>), grid, block, smem, stream, p);
  else if (xl == FH_R_STRIDED && yl == FH_R_CONTIG)
    hipLaunchKernelGGL((gemm_dma_kernel<FH_R_STRIDED, FH_R_CONTIG>), grid, block, smem, stream, p);
  else
    hipLaunchKernelGGL((gemm_dma_kernel<FH_R_STRIDED, FH_R_STRIDED>), grid, block, smem, stream, p);
  FH_CHECK_LAUNCH();
  return 0;
}

// FH_GEMM_PATH=dma selects the LDS-DMA main loop for bf16 (A/B comparisons; see tools/gemm_bench.py)
int g_gemm_path = -1;  // -1: from env FH_GEMM_PATH; 0: register staging (default); 1: LDS-DMA
bool use_dma() {
  if (g_gemm_path < 0) {
    const char* e = getenv("FH_GEMM_PATH");
    g_gemm_path = (e && e[0] == 'd') ? 1 : 0;  // default: register staging (measured faster on the student's shapes)
  }
  return g_gemm_path == 1;
}

}  // namespace

// ------------------------------------------------------------------ optional per-launch HIP-event timing (bench.py roofline)
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int variant; };
std::vector<ProfRec> g_prof;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
bool g_prof_on = false;
}  // namespace

extern "C" int fh_gemm_set_path(int path) {
  g_gemm_path = path;
  return 0;
}
extern "C" int fh_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
// sums per (dtype*4 + X layout*2 + Y layout) variant; returns the number of launches recorded and clears the log
extern "C" int fh_prof_collect(double* ms /*[8]*/, double* flops /*[8]*/, long* launches /*[8]*/) {
  for (int i = 0; i < 8; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  int n = 0;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms[r.variant] += t;
    flops[r.variant] += r.flops;
    launches[r.variant] += 1;
    g_pool.emplace_back(r.a, r.b);
    ++n;
  }
  g_prof.clear();
  return n;
}

int fh_gemm_launch(const GemmParams& p, int dtype, hipStream_t stream) {
  const int vec = dtype == 0 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0 || p.Kseg <= 0) return 0;
  // vector-access preconditions (all hold for the student's shapes; fail loudly otherwise)
  if (p.N % 4 != 0) return -101;
  if (p.X.layout == FH_R_CONTIG && (p.Kseg % vec || p.X.ld % vec || p.X.bstride % vec)) return -102;
  if (p.Y.layout == FH_R_CONTIG && (p.Kseg % vec || p.Y.ld % vec || p.Y.bstride % vec)) return -103;
  if (p.X.layout == FH_R_STRIDED && (p.M % vec || p.X.ld % vec || p.X.bstride % vec)) return -104;
  if (p.Y.layout == FH_R_STRIDED && (p.N % vec || p.Y.ld % vec || p.Y.bstride % vec)) return -105;
  if (p.X.nseg != p.Y.nseg || p.X.nseg < 1 || p.X.nseg > 2) return -106;
  if (p.splitk > 1 && !p.ep.atomic) return -107;
  if (!p.ep.atomic && (p.ldc % 4 || p.c_bstride % 4 || p.c_gstride % 4)) return -108;
  if (p.ep.row_len && p.c_rpb <= 0) return -109;
  if ((p.X.segoff[0] | p.X.segoff[1] | p.Y.segoff[0] | p.Y.segoff[1] | p.X.gstride | p.Y.gstride) % vec) return -110;
  if (!g_prof_on) return dtype == 0 ? launch_t<float>(p, stream) : (use_dma() ? launch_dma(p, stream) : launch_t<bf16>(p, stream));
  ProfRec r;
  if (!g_pool.empty()) {
    r.a = g_pool.back().first; r.b = g_pool.back().second;
    g_pool.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -120;
  }
  r.flops = 2.0 * p.M * p.N * (double)p.Kseg * p.X.nseg * p.groups;
  r.variant = (dtype ? 4 : 0) + p.X.layout * 2 + p.Y.layout;
  hipEventRecord(r.a, stream);
  const int rc = dtype == 0 ? launch_t<float>(p, stream) : (use_dma() ? launch_dma(p, stream) : launch_t<bf16>(p, stream));
  hipEventRecord(r.b, stream);
  g_prof.push_back(r);
  return rc;
}
