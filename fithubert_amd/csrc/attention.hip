// Multi-head self-attention (flash-style, online softmax) forward and backward on a packed q|k|v tensor.
//
//   reference: fairseq MultiheadAttention as called at modules/module.py:558-564 (self-attention, key_padding_mask,
//   dropout on the probabilities) -- q scaled by head_dim^-0.5, scores = q k^T per head, masked_fill(-inf), fp32
//   softmax, dropout, @ v.  (Third-party arithmetic; semantics pinned in oracle/student_ref.py::_mha.)
//
// Layout: qkv [B][T][3*D] (q | k | v, head h at columns h*dh..), out [B][T][D].  One workgroup = 128 .. 512 queries of one
// (batch, head); each wave owns 32 queries.  K/V tiles of 64 keys go through LDS (forward: streamed, double-buffered).  Scores are computed TRANSPOSED
// (S^T = K Q^T) so that a lane holds one query column: the softmax reductions are two row exchanges (v_permlane16/32_swap), the
// accumulator tiles of S^T are directly the B operand of the P.V product (no LDS round trip), and V^T / K^T come
// from transposing LDS reads (ds_read_b64_tr_b16).  Head dim 40 is zero-padded to the MFMA K granularity in LDS only.
// Backward recomputes P from the saved log-sum-exp.  bf16, T <= 256 (the 10 s student): ONE fused pass per (batch, head) -- the
// dK/dV decomposition (wave owns keys) plus a dQ contraction through an LDS scratch, attn_bwd_fused_kernel.  Otherwise: a dQ pass
// (the forward's decomposition) and a dK/dV pass.  These kernels are VALU-bound (exp2 + dropout hash per score), not MFMA-bound.
#include <cstdlib>
#include <type_traits>

#include "common.cuh"
#include "kernels.h"
#include "attn_frag.cuh"

extern int fh_g_reserve_cus;                       // gemm.hip: CUs persistent launches leave to other streams' kernels
int fh_stream_reserve_cus(hipStream_t st);

namespace {

// ------------------------------------------------------------------------------------------------ forward
// DROP is a template parameter and the key mask lives in a separate instance of the tile step: left as run-time tests both were
// if-converted into the one loop body -- the teacher (no dropout, full tiles) executed the dropout hash (16 quarter-rate
// v_mul_lo_u32 per tile and wave) and 32 compare / select pairs per tile for nothing: 249 VALU instructions per 64-key tile and
// wave where the softmax itself needs ~100 (s_memtime: 1630 cycles per wave and tile against 512 of MFMA).
// (batch, head, block) of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), which would
// put the query / key blocks of one (batch, head) -- readers of the same K / V (Q / dO) rows -- on as many L2s as there are blocks:
// XCD k takes a contiguous range of the logical order instead (as the GEMM loops' xcd_remap), so those rows come through ONE L2.
__device__ __forceinline__ void attn_block(int& b, int& h, int& blk) {
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * (int)gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int per = nwg >> 3, rem = nwg & 7, xcd = lin & 7;
  const int lg = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (lin >> 3);
  blk = lg % gx;
  const int bh = lg / gx;
  h = bh % gy;
  b = bh / gy;
}

template <typename T, int NDT, int NW, bool DROP>
__global__ void __launch_bounds__(NW * 64, NW >= 8 ? 4 : 2) attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                       float* __restrict__ lse, const int* __restrict__ key_len, int B,
                                                       int Tn, int H, int dh, float scale_log2e, DropCfg drop) {
  // K / V are STREAMED through two (K tile | V tile) buffers (36 KiB): every thread fetches its 16-byte pieces of tile k + 1 into
  // registers right after the barrier of tile k, computes tile k, and stores them when it comes round -- the loads' latency and the
  // HBM traffic sit under the softmax / MFMA work; one barrier per tile, any T.  (Round 1 staged ALL tiles of a (batch, head) up
  // front for T <= 512: no barrier in the key loop, but s_memtime stamps showed 25 K of a workgroup's 70 K cycles at T = 499 in
  // front of the loop -- 16 serialised load round trips, then, batched, an HBM-bound burst at ~11 B / cycle / CU with every CU
  // staging at once -- and above T = 512 the old tiled path loaded synchronously.  Streamed: 255 -> 202 us per teacher layer
  // together with the DROP / tail-mask instances below, and the long-utterance path no longer waits for every tile.)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PAIR = 2 * 64 * AT<T>::RS;
  constexpr int NK = AT<T>::NK;
  typedef typename AT<T>::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  int b, h, blk;
  attn_block(b, h, blk);
  const int q0 = blk * (32 * NW) + wave * 32;
  const int D = H * dh;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * Tn * ld + h * dh;
  const int klen = key_len ? min(key_len[b], Tn) : Tn;
  // dropout keys of this (batch, head)'s Tn x Tn index range (Tn <= 65536: at most two high words)
  const DropKeySpan dspan = drop_key_span(drop, ((unsigned long long)b * H + h) * Tn * drop_row_pitch(Tn));

  zero_lds<T>(lds, 2 * PAIR);  // (columns dh .. 63 of the padded rows are never written again)
  frag qf[2][NK];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = q0 + 16 * qt + r;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) qf[qt][ks] = gfrag<T>(base + (long)query * ld, query < Tn, ks, lane, dh);
  }
  __syncthreads();
  // this thread's pieces of one (K tile | V tile) pair -- piece i = row (i / cpr) % 64 of K (i < 64 cpr) or V
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int NPC = (2 * 64 * (64 / VEC) + NW * 64 - 1) / (NW * 64);  // pieces per thread at the widest head
  const int cpr = dh / VEC, npiece = 2 * 64 * cpr;
  V16 pre[NPC];
  int prow[NPC], pdst[NPC];  // row inside the tile (-1: no piece), LDS byte offset inside the buffer
  long psrc[NPC];            // element offset from `base` at tile 0
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = (int)threadIdx.x + j * NW * 64;
    const int isv = i >= 64 * cpr, rem = i - isv * 64 * cpr, row = rem / cpr, c = rem - row * cpr;
    prow[j] = i < npiece ? row : -1;
    pdst[j] = isv * 64 * AT<T>::RS + row * AT<T>::RS + c * 16;
    psrc[j] = (long)D + (long)isv * D + (long)row * ld + c * VEC;
  }
  auto fetch = [&](int kt0) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      pre[j] = v16_zero();
      if (prow[j] >= 0 && kt0 + prow[j] < Tn) pre[j] = ld16(base + psrc[j] + (long)kt0 * ld);
    }
  };
  auto deposit = [&](unsigned char* buf) {
#pragma unroll
    for (int j = 0; j < NPC; ++j)
      if (prow[j] >= 0) st16(buf + pdst[j], pre[j]);
  };
  fetch(0);
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x4 o[2][NDT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile_step = [&](int kt0, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;  // the last tile of a length that is no multiple of 64: keys past klen are masked
    unsigned char* Ks = lds + ((kt0 >> 6) & 1) * PAIR;
    const unsigned char* Vs = Ks + 64 * AT<T>::RS;
    // the buffer's previous readers (tile k - 2) finished before they arrived at the barrier of tile k - 1, which this wave has passed
    deposit(Ks);
    __syncthreads();
    if (kt0 + 64 < klen) fetch(kt0 + 64);
    f32x4 s[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      s[kb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const frag a = lfrag<T>(Ks, 16 * kb + r, ks, lane);
        s[kb][0] = mma<T>(a, qf[0][ks], s[kb][0]);
        s[kb][1] = mma<T>(a, qf[1][ks], s[kb][1]);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int query = q0 + 16 * qt + r;
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = s[kb][qt][e];
          if (TAIL && kt0 + 16 * kb + 4 * q + e >= klen) v = -INFINITY;
          s[kb][qt][e] = v;
          mx = fmaxf(mx, v);
        }
      mx = xor_max(mx) * scale_log2e;  // scale > 0: max commutes with it; the per-element scaling rides in the fma below
      const float mnew = fmaxf(m[qt], mx);
      const float alpha = fast_exp2(m[qt] - mnew);
      m[qt] = mnew;
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = fast_exp2(fmaf(s[kb][qt][e], scale_log2e, -mnew));
          rs += p;
          s[kb][qt][e] = p;
        }
      rs = xor_sum(rs);
      l[qt] = l[qt] * alpha + rs;
      if (__any(alpha != 1.0f)) {  // (a running maximum that did not move leaves alpha = 2^0 in every lane: skipping is exact)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[qt][dt][e] *= alpha;
      }
      if (DROP && drop.p > 0.f) {  // (the run-time test stays: without it the compiler keeps all 16 hashes in flight and spills under the 128-register cap)
        const unsigned long long rowbase = (((unsigned long long)b * H + h) * Tn + query) * drop_row_pitch(Tn) + kt0;
        const DropKey dk = drop_key_at(dspan, rowbase);
        const uint32_t lo = (uint32_t)rowbase;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            float d0, d1;
            drop_pair(dk, lo + 16 * kb + 4 * q + e, d0, d1);
            s[kb][qt][e] *= d0;
            s[kb][qt][e + 1] *= d1;
          }
      }
    }
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) mma_transposed<T, NDT>(Vs, m2, s[2 * m2][qt], s[2 * m2 + 1][qt], o[qt], lane);
  };
#ifdef FH_ATTN_DBG_NOLOOP   // timing aid: prologue + epilogue alone
  const int nfull = 0;
  l[0] = l[1] = 1.f;
#else
  const int nfull = klen >> 6;
#endif
  for (int kt = 0; kt < nfull; ++kt) tile_step(kt * 64, std::false_type{});
#ifndef FH_ATTN_DBG_NOLOOP
  if (klen & 63) tile_step(nfull * 64, std::true_type{});
#endif

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = q0 + 16 * qt + r;
    if (query >= Tn) continue;
    const float inv = 1.0f / l[qt];
    if (q == 0 && lse) lse[((long)b * H + h) * Tn + query] = m[qt] + log2f(l[qt]);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[qt][dt][e] * inv;
        st4(out + ((long)b * Tn + query) * D + h * dh + d, v);
      }
    }
  }
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
template <typename T> __device__ __forceinline__ uint32_t pack_h2(float a, float b) {  // one v_cvt_pk_{bf16,f16}_f32
  if constexpr (ElemTraits<T>::DT == 2) {
    return pack2<f16>(a, b);
  } else {
    union { bf16x2 v; uint32_t u; } c;
    c.v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return c.u;
  }
}
// acc[dt] += Tile^T[d-tile dt][rows] * P with P already packed: lo / hi = the 4 bf16 of the two C tiles covering tile rows
// [32 m2, 32 m2 + 16) and [32 m2 + 16, 32 m2 + 32)  (mma_transposed<bf16> without its conversions)
template <typename T, int NDT>
__device__ __forceinline__ void mma_transposed_packed(const unsigned char* tile, int m2, uint2 lo, uint2 hi, f32x4 (&acc)[NDT], int lane) {
  const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
  union { struct { uint2 lo, hi; } s; bf16x8 v; } pb;
  pb.s.lo = lo;
  pb.s.hi = hi;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    const unsigned char* a0 = tile + (32 * m2 + 4 * q + qq) * AT<bf16>::RS + (16 * dt + 4 * pp) * 2;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * AT<bf16>::RS));
    acc[dt] = mfma16<T>(u.v, pb.v, acc[dt]);
  }
}
// Per-lane state of a wave that owns 32 keys (two 16-key C tiles, lane <-> key) and walks 16-query blocks: the dK/dV side of
// both backward forms (bf16).  block() turns the S and dP tiles of one 16-query block into the packed bf16 P.drop and dS tiles.
//  * masked keys: exp2(-inf) = 0 through an additive bias instead of a select per score
//  * dropout: keys 2j and 2j + 1 share one hash word (drop_pair) and sit in neighbouring lanes, so a lane hashes the words of TWO
//    of its four queries per C tile (4q + 2 par + {0, 1}) and takes the other two from its neighbour (one DPP move per word).
//    Index = ((b H + h) Tn + query) * pitch + key, split as in the forward: (row base + 64-key tile start) picks the key of the
//    high word, the tile-relative offset is added to the low word only.
struct KeyScore {
  float kbias[2], scale, scale_log2e, inv_keep;
  uint32_t t16, koff, hshift, hi0, key0, key1;
  unsigned long long ib0, pitch;
  int par;
  __device__ __forceinline__ void init(const DropCfg& drop, int b, int h, int H, int Tn, int klen, int k0, int lane, float sc,
                                       float sl2) {
    const int r = lane & 15, q = lane >> 4;
    kbias[0] = k0 + r < klen ? 0.f : -INFINITY;
    kbias[1] = k0 + 16 + r < klen ? 0.f : -INFINITY;
    scale = sc;
    scale_log2e = sl2;
    par = r & 1;
    const DropKeySpan dspan = drop_key_span(drop, ((unsigned long long)b * H + h) * Tn * drop_row_pitch(Tn));
    t16 = dspan.k.thresh >> 8;
    inv_keep = dspan.k.inv_keep;
    hi0 = dspan.hi0;
    key0 = dspan.k.k;
    key1 = dspan.k_next;
    pitch = drop_row_pitch(Tn);
    const int kt0w = k0 & ~63;
    ib0 = (((unsigned long long)b * H + h) * Tn + 4 * q + 2 * par) * pitch + kt0w;
    koff = (uint32_t)(k0 - kt0w) + (uint32_t)(r & ~1);
    hshift = 16u * par;
  }
  // q16 = first query of the 16-query block; lse4 / dl4 = log-sum-exp (log2 domain, +inf for rows past Tn) and delta of this
  // lane's four queries q16 + 4q .. 4q + 3
  template <typename T, bool DROP>
  __device__ __forceinline__ void block(int q16, const f32x4 (&s)[2], const f32x4 (&dp)[2], f32x4 lse4, f32x4 dl4, uint2 (&pdp)[2],
                                        uint2 (&dsp)[2]) const {
    uint32_t hw[2][4];  // [kt][e] hash words
    if (DROP) {
      uint32_t mine[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const unsigned long long rb = ib0 + (unsigned long long)(q16 + jj) * pitch;
        const uint32_t key = (uint32_t)(rb >> 32) == hi0 ? key0 : key1;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) mine[kt][jj] = fh_mix32(((uint32_t)rb + koff + 16u * kt) ^ key);
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine[kt][jj], 0xB1, 0xf, 0xf, true);  // lane ^ 1
          hw[kt][jj] = par ? other : mine[kt][jj];
          hw[kt][2 + jj] = par ? mine[kt][jj] : other;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float pdv[4], dsv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = fast_exp2(fmaf(s[kt][e], scale_log2e, kbias[kt] - lse4[e]));
        float dsc = 1.f;
        if (DROP) dsc = __builtin_amdgcn_ubfe(hw[kt][e], hshift, 16u) >= t16 ? inv_keep : 0.f;
        pdv[e] = p * dsc;
        dsv[e] = (p * scale) * fmaf(dp[kt][e], dsc, -dl4[e]);
      }
      pdp[kt] = make_uint2(pack_h2<T>(pdv[0], pdv[1]), pack_h2<T>(pdv[2], pdv[3]));
      dsp[kt] = make_uint2(pack_h2<T>(dsv[0], dsv[1]), pack_h2<T>(dsv[2], dsv[3]));
    }
  }
};

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <typename T, int NDT, int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dqkv,
                                                          const int* __restrict__ key_len, int B, int Tn, int H, int dh,
                                                          float scale, float scale_log2e, DropCfg drop, int resident) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PAIR = 2 * 64 * AT<T>::RS;
  constexpr int NK = AT<T>::NK;
  constexpr int VEC = ElemTraits<T>::VEC;
  typedef typename AT<T>::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  int b, h, blk;
  attn_block(b, h, blk);
  const int q0 = blk * (32 * NW) + wave * 32;
  const int D = H * dh;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * Tn * ld + h * dh;
  const int klen = key_len ? min(key_len[b], Tn) : Tn;
  // dropout keys of this (batch, head)'s Tn x Tn index range (Tn <= 65536: at most two high words)
  const DropKeySpan dspan = drop_key_span(drop, ((unsigned long long)b * H + h) * Tn * drop_row_pitch(Tn));

  const int ntile = (klen + 63) / 64;
  zero_lds<T>(lds, (resident ? ntile : 1) * PAIR);
  if (resident) {
    __syncthreads();
    stage_tiles<T>(2 * ntile, Tn, dh, [&](int t, const T*& src, long& sld, unsigned char*& dst, int& row0) {
      const int kt = t >> 1, isv = t & 1;  // key tile, value tile of 64 keys
      src = base + D + isv * D;
      sld = ld;
      dst = lds + kt * PAIR + isv * 64 * AT<T>::RS;
      row0 = kt * 64;
    });
  }
  frag qf[2][NK], dof[2][NK];
  float L2[2], dl[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = q0 + 16 * qt + r;
    const bool valid = query < Tn;
    const T* orow = out + ((long)b * Tn + query) * D + h * dh;
    const T* drow = dout + ((long)b * Tn + query) * D + h * dh;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      qf[qt][ks] = gfrag<T>(base + (long)query * ld, valid, ks, lane, dh);
      dof[qt][ks] = gfrag<T>(drow, valid, ks, lane, dh);
      const int d0 = AT<T>::KD * ks + VEC * q;
      if (valid && d0 < dh) {
        float a[VEC], c[VEC];
        unpack16<T>(ld16(orow + d0), a);
        unpack16<T>(ld16(drow + d0), c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) part = fmaf(a[e], c[e], part);
      }
    }
    dl[qt] = xor_sum(part);
    L2[qt] = valid ? lse[((long)b * H + h) * Tn + query] : INFINITY;
    if (valid && q == 0) delta[((long)b * H + h) * Tn + query] = dl[qt];
  }
  f32x4 dq[2][NDT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (resident) __syncthreads();
  for (int kt0 = 0; kt0 < klen; kt0 += 64) {
    const unsigned char* Ks = lds + (resident ? (kt0 >> 6) * PAIR : 0);
    const unsigned char* Vs = Ks + 64 * AT<T>::RS;
    if (!resident) {
      __syncthreads();
      stage_tile<T>(lds, base + D, ld, kt0, Tn, dh);
      stage_tile<T>(lds + 64 * AT<T>::RS, base + 2 * D, ld, kt0, Tn, dh);
      __syncthreads();
    }
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
      f32x4 ds[2][2];  // [half][qt]
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kb = 2 * m2 + half;
        f32x4 s[2], dp[2];
        s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const frag ak = lfrag<T>(Ks, 16 * kb + r, ks, lane);
          const frag av = lfrag<T>(Vs, 16 * kb + r, ks, lane);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            s[qt] = mma<T>(ak, qf[qt][ks], s[qt]);
            dp[qt] = mma<T>(av, dof[qt][ks], dp[qt]);
          }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int query = q0 + 16 * qt + r;
          const unsigned long long rowbase = (((unsigned long long)b * H + h) * Tn + query) * drop_row_pitch(Tn) + kt0;
          const DropKey dk = drop_key_at(dspan, rowbase);
          const uint32_t lo = (uint32_t)rowbase;
          float dsc[4] = {1.f, 1.f, 1.f, 1.f};
          if (drop.p > 0.f) {
            drop_pair(dk, lo + 16 * kb + 4 * q, dsc[0], dsc[1]);
            drop_pair(dk, lo + 16 * kb + 4 * q + 2, dsc[2], dsc[3]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = kt0 + 16 * kb + 4 * q + e;
            float p = key < klen ? fast_exp2(fmaf(s[qt][e], scale_log2e, -L2[qt])) : 0.f;
            ds[half][qt][e] = p * (dp[qt][e] * dsc[e] - dl[qt]) * scale;
          }
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) mma_transposed<T, NDT>(Ks, m2, ds[0][qt], ds[1][qt], dq[qt], lane);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int query = q0 + 16 * qt + r;
    if (query >= Tn) continue;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dq[qt][dt][e];
        st4(dqkv + ((long)b * Tn + query) * ld + h * dh + d, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <typename T, int NDT, int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, const int* __restrict__ key_len, int B,
                                                           int Tn, int H, int dh, float scale, float scale_log2e,
                                                           DropCfg drop, int resident) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PAIR = 2 * 64 * AT<T>::RS + 512;  // Q tile | dO tile | 64 x LSE | 64 x delta
  constexpr int NK = AT<T>::NK;
  typedef typename AT<T>::frag frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  int b, h, blk;
  attn_block(b, h, blk);
  const int k0 = blk * (32 * NW) + wave * 32;
  const int D = H * dh;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * Tn * ld + h * dh;
  const T* dobase = dout + (long)b * Tn * D + h * dh;
  const int klen = key_len ? min(key_len[b], Tn) : Tn;
  // dropout keys of this (batch, head)'s Tn x Tn index range (Tn <= 65536: at most two high words)
  const DropKeySpan dspan = drop_key_span(drop, ((unsigned long long)b * H + h) * Tn * drop_row_pitch(Tn));

  KeyScore ksc;
  ksc.init(drop, b, h, H, Tn, klen, k0, lane, scale, scale_log2e);
  const int ntile = (Tn + 63) / 64;
  zero_lds<T>(lds, (resident ? ntile : 1) * PAIR);
  auto stage_q = [&](int slot, int qt0) {
    unsigned char* sb = lds + slot * PAIR;
    stage_tile<T>(sb, base, ld, qt0, Tn, dh);
    stage_tile<T>(sb + 64 * AT<T>::RS, dobase, D, qt0, Tn, dh);
    if (threadIdx.x < 64) {
      float* L2w = reinterpret_cast<float*>(sb + 2 * 64 * AT<T>::RS);
      const int query = qt0 + threadIdx.x;
      L2w[threadIdx.x] = query < Tn ? lse[((long)b * H + h) * Tn + query] : INFINITY;
      L2w[64 + threadIdx.x] = query < Tn ? delta[((long)b * H + h) * Tn + query] : 0.f;
    }
  };
  if (resident) {
    __syncthreads();
    stage_tiles<T>(2 * ntile, Tn, dh, [&](int t, const T*& src, long& sld, unsigned char*& dst, int& row0) {
      const int kt = t >> 1, isdo = t & 1;  // query tile, dO tile of 64 queries
      src = isdo ? dobase : base;
      sld = isdo ? (long)D : ld;
      dst = lds + kt * PAIR + isdo * 64 * AT<T>::RS;
      row0 = kt * 64;
    });
    for (int i = threadIdx.x; i < ntile * 64; i += blockDim.x) {  // log-sum-exp and delta of every query
      float* L2w = reinterpret_cast<float*>(lds + (i >> 6) * PAIR + 2 * 64 * AT<T>::RS);
      L2w[i & 63] = i < Tn ? lse[((long)b * H + h) * Tn + i] : INFINITY;
      L2w[64 + (i & 63)] = i < Tn ? delta[((long)b * H + h) * Tn + i] : 0.f;
    }
  }
  frag kf[2][NK], vf[2][NK];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      kf[kt][ks] = gfrag<T>(base + D + (long)key * ld, key < Tn, ks, lane, dh);
      vf[kt][ks] = gfrag<T>(base + 2 * D + (long)key * ld, key < Tn, ks, lane, dh);
    }
  }
  f32x4 dk[2][NDT], dv[2][NDT];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dk[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  if (resident) __syncthreads();
  for (int qt0 = 0; qt0 < Tn; qt0 += 64) {
    const unsigned char* Qs = lds + (resident ? (qt0 >> 6) * PAIR : 0);
    const unsigned char* Os = Qs + 64 * AT<T>::RS;
    const float* L2s = reinterpret_cast<const float*>(Qs + 2 * 64 * AT<T>::RS);
    const float* Dls = L2s + 64;
    if (!resident) {
      __syncthreads();
      stage_q(0, qt0);
      __syncthreads();
    }
    if constexpr (sizeof(T) == 2) {
      // bf16: the shared score step (KeyScore) -- branch-free, pair hashes shared between neighbouring lanes, one bf16 conversion
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        uint2 pdp[2][2], dsp[2][2];  // [half][kt]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int qb = 2 * m2 + half;
          f32x4 s[2], dp[2];
          s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const frag aq = lfrag<T>(Qs, 16 * qb + r, ks, lane);
            const frag ao = lfrag<T>(Os, 16 * qb + r, ks, lane);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              s[kt] = mma<T>(aq, kf[kt][ks], s[kt]);
              dp[kt] = mma<T>(ao, vf[kt][ks], dp[kt]);
            }
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(L2s + 16 * qb + 4 * q);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(Dls + 16 * qb + 4 * q);
          if (drop.p > 0.f) ksc.template block<T, true>(qt0 + 16 * qb, s, dp, l4, dl, pdp[half], dsp[half]);
          else ksc.template block<T, false>(qt0 + 16 * qb, s, dp, l4, dl, pdp[half], dsp[half]);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          mma_transposed_packed<T, NDT>(Os, m2, pdp[0][kt], pdp[1][kt], dv[kt], lane);
          mma_transposed_packed<T, NDT>(Qs, m2, dsp[0][kt], dsp[1][kt], dk[kt], lane);
        }
      }
    } else {
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
      f32x4 pd[2][2], ds[2][2];  // [half][kt]
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qb = 2 * m2 + half;
        f32x4 s[2], dp[2];
        s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const frag aq = lfrag<T>(Qs, 16 * qb + r, ks, lane);
          const frag ao = lfrag<T>(Os, 16 * qb + r, ks, lane);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            s[kt] = mma<T>(aq, kf[kt][ks], s[kt]);
            dp[kt] = mma<T>(ao, vf[kt][ks], dp[kt]);
          }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int key = k0 + 16 * kt + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ql = 16 * qb + 4 * q + e;
            const int query = qt0 + ql;
            float p = key < klen ? fast_exp2(s[kt][e] * scale_log2e - L2s[ql]) : 0.f;
            float dsc = 1.f;
            if (drop.p > 0.f) {
              // same (key-tile-relative) split of the index as the forward: hi word of (row base + tile start), lo + offset
              const int kt0 = key & ~63;
              const unsigned long long rowbase = (((unsigned long long)b * H + h) * Tn + query) * drop_row_pitch(Tn) + kt0;
              dsc = drop_pair_one(drop_key_at(dspan, rowbase), (uint32_t)rowbase + (key - kt0));
            }
            pd[half][kt][e] = p * dsc;
            ds[half][kt][e] = p * (dp[kt][e] * dsc - Dls[ql]) * scale;
          }
        }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        mma_transposed<T, NDT>(Os, m2, pd[0][kt], pd[1][kt], dv[kt], lane);
        mma_transposed<T, NDT>(Qs, m2, ds[0][kt], ds[1][kt], dk[kt], lane);
      }
    }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
    if (key >= Tn) continue;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dk[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + D + h * dh + d, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dv[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + 2 * D + h * dh + d, v);
      }
    }
  }
}

// q | k | v bias gradient out of the fused backward kernels: column sums of a workgroup's dQ / dK / dV contributions (fp32, before the
// rounding of the stored gradient), reduced over the 16 rows of a tile in the wave, over the waves through LDS (`red`: a region no
// wave reads any more), then one atomicAdd per column and workgroup into dbias [3][D].  Replaces a pass over the 3 D-wide gradient
// (2.2 GB per student step at B = 256).  Every thread of the workgroup calls it (one barrier).
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
template <int NDT>
__device__ __forceinline__ void qkv_bias_sums(float* red, float* __restrict__ dbias, const f32x4 (&dk)[2][NDT], const f32x4 (&dv)[2][NDT],
                                              const bool (&kvalid)[2], const f32x4& qsum, int h, int dh, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  constexpr int NC = 16 * NDT;
  float* redk = red;                   // [8 waves][NC]
  float* redv = red + 8 * NC;          // [8 waves][NC]
  float* redq = red + 16 * NC;         // [2 query halves][NC]
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        if (kvalid[kt]) {
          a += dk[kt][dt][e];
          c += dv[kt][dt][e];
        }
      a = row16_sum(a);
      c = row16_sum(c);
      if (r == 0) {
        redk[wave * NC + 16 * dt + 4 * q + e] = a;
        redv[wave * NC + 16 * dt + 4 * q + e] = c;
      }
    }
  if (wave < 2 * NDT) {
    const int qb2 = wave / NDT, dt = wave - qb2 * NDT;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = row16_sum(qsum[e]);
      if (r == 0) redq[qb2 * NC + 16 * dt + 4 * q + e] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 * NC) {
    const int which = threadIdx.x / NC, c = threadIdx.x - which * NC;
    if (c < dh) {
      float s = 0.f;
      if (which == 0) s = redq[c] + redq[NC + c];
      else
        for (int w = 0; w < 8; ++w) s += (which == 1 ? redk : redv)[w * NC + c];
      atomicAdd(dbias + (long)which * D + h * dh + c, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: fused dQ / dK / dV
// One workgroup = one (batch, head), Tn <= 256, bf16; 8 waves, wave w owns keys [32w, 32w + 32) exactly like the dK/dV pass above
// (its K / V fragments and dK / dV accumulators stay in registers while it walks the resident Q / dO tiles).  What the two-pass
// form computes twice -- P (an exp2 per score), the dropout mask (a hash per score) and dS -- is computed once: after each
// 32-query step the wave leaves its dS block, transposed ([key][query], bf16), in an LDS scratch; one barrier later six waves
// (one per 16-query x 16-d output tile) contract the scratch with the resident K tile over all 256 keys, dQ^T = K^T dS^T, both
// operands through transposing LDS reads, and store that dQ block (complete: every key has been seen).  The scratch is double
// buffered, so the step costs ONE barrier and the dQ MFMAs overlap the next step's exp / hash work.
// LDS: 4 x (Q | dO | lse | delta) tiles 75,776 B + K 36,864 B + 2 x dS^T scratch 36,864 B = 149,504 B.
constexpr int FUSED_SRS = 72;  // scratch row: 32 queries x 2 B + 8 B pad (conflict-free transposing reads)
template <typename T, int NDT, bool DROP>
__global__ void __launch_bounds__(512) attn_bwd_fused_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                             const T* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ delta, T* __restrict__ dqkv,
                                                             const int* __restrict__ key_len, int B, int Tn, int H, int dh,
                                                             float scale, float scale_log2e, DropCfg drop, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RS = AT<T>::RS;
  constexpr int PAIR = 2 * 64 * RS + 512;  // Q tile | dO tile | 64 x LSE | 64 x delta
  constexpr int NK = AT<T>::NK;
  typedef typename AT<T>::frag frag;
  unsigned char* Kt = lds + 4 * PAIR;
  unsigned char* Sc = Kt + 256 * RS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  // Which (batch, head) pairs.  The workgroup owns its CU (149.5 KB of LDS), so nothing of its own hides the load round trips of its
  // prologue or the drain of its last stores (measured, B = 256, T = 249: prologue 78 + key loop 240 + epilogue 46 us per launch).
  //  * grid (1, H, B): one pair per workgroup.  The heads of one utterance read and write neighbouring 80-byte (2 dh) pieces of the
  //    same memory rows: dealt round-robin over the XCDs in dispatch order they would pull every 128-byte line through up to eight
  //    L2s; XCD k takes a contiguous range of (b, h) instead (attn_block; 394 -> 364 us).
  //  * grid (G, 1, 1), round 6: PERSISTENT -- one workgroup per CU walks pairs; the stores of pair i drain while the loads of pair
  //    i + 1 are already in flight.
  const int npair = B * H;
  int pair, pair_end, pair_step = 1;
  if (gridDim.y == 1 && gridDim.z == 1 && (int)gridDim.x != npair) {
    // XCD x (workgroups w = x mod 8, dealt round-robin) takes the contiguous eighth [x0, x1) of the pairs, and ITS workgroups walk it
    // interleaved (j, j + G8, ...): at any time the XCD works on ~G8 consecutive pairs -- neighbouring heads of the same utterances,
    // whose shared lines are then in its L2 (a contiguous range PER WORKGROUP measured no better than no remap at all: head i + 1
    // came 30 us after head i, when the line had long left the L2)
    const int G = gridDim.x, w = blockIdx.x, x = w & 7, j = w >> 3;
    const int per = npair >> 3, rem = npair & 7;
    const int x0 = x * per + min(x, rem);
    pair_step = (G - x + 7) >> 3;                 // workgroups of this XCD
    pair = x0 + j;
    pair_end = x0 + per + (x < rem ? 1 : 0);
  } else {
    int b_, h_, blk_;
    attn_block(b_, h_, blk_);
    pair = b_ * H + h_;
    pair_end = pair + 1;
  }
  const int k0 = wave * 32;
  const int D = H * dh;
  const long ld = 3L * D;
  for (; pair < pair_end; pair += pair_step) {
  const int b = pair / H, h = pair - b * H;
  const T* base = qkv + (long)b * Tn * ld + h * dh;
  const T* dobase = dout + (long)b * Tn * D + h * dh;
  const T* obase = out + (long)b * Tn * D + h * dh;
  const int klen = key_len ? min(key_len[b], Tn) : Tn;
  const int ntile = (Tn + 63) / 64;

  zero_lds<T>(lds, 4 * PAIR + 256 * RS);  // the scratch is fully rewritten every step
  __syncthreads();
  stage_tiles<T>(3 * ntile, Tn, dh, [&](int t, const T*& src, long& sld, unsigned char*& dst, int& row0) {
    const int kt = t / 3, which = t - 3 * kt;  // query, dO, key tile of 64 rows
    src = which == 0 ? base : (which == 1 ? dobase : base + D);
    sld = which == 1 ? (long)D : ld;
    dst = which == 2 ? Kt + kt * 64 * RS : lds + kt * PAIR + which * 64 * RS;
    row0 = kt * 64;
  });
  {  // lse, and delta[query] = sum_d out * dout: two threads per query row
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    float part = 0.f;
    if (row < Tn)
      for (int c = half; c < dh / 8; c += 2) {
        float a[8], g[8];
        unpack16<T>(ld16(obase + (long)row * D + c * 8), a);
        unpack16<T>(ld16(dobase + (long)row * D + c * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf(a[e], g[e], part);
      }
    part += __shfl_xor(part, 1, 64);
    if (half == 0) {
      float* L2w = reinterpret_cast<float*>(lds + (row >> 6) * PAIR + 2 * 64 * RS);
      L2w[row & 63] = row < Tn ? lse[((long)b * H + h) * Tn + row] : INFINITY;
      L2w[64 + (row & 63)] = part;
      if (row < Tn) delta[((long)b * H + h) * Tn + row] = part;
    }
  }
  frag kf[2][NK], vf[2][NK];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      kf[kt][ks] = gfrag<T>(base + D + (long)key * ld, key < Tn, ks, lane, dh);
      vf[kt][ks] = gfrag<T>(base + 2 * D + (long)key * ld, key < Tn, ks, lane, dh);
    }
  }
  f32x4 dk[2][NDT], dv[2][NDT];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dk[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  KeyScore ks_;
  ks_.init(drop, b, h, H, Tn, klen, k0, lane, scale, scale_log2e);
  __syncthreads();

#ifdef FH_ATTN_DBG_NOLOOP   // timing aid (tools/scratch/mkvariant.py): prologue + epilogue of the fused backward alone
  const int nstep = 0;
#else
  const int nstep = (Tn + 31) / 32;  // 32-query steps
#endif
  f32x4 qsum = f32x4{0.f, 0.f, 0.f, 0.f};   // column sums of this wave's dQ tiles (q bias gradient)
  const int qq = r >> 2, pp = r & 3;
  for (int st = 0; st < nstep; ++st) {
    const int m2 = st & 1;
    const unsigned char* Qs = lds + (st >> 1) * PAIR;
    const unsigned char* Os = Qs + 64 * RS;
    const float* L2s = reinterpret_cast<const float*>(Qs + 2 * 64 * RS);
    const float* Dls = L2s + 64;
    unsigned char* Sb = Sc + (st & 1) * 256 * FUSED_SRS;
    uint2 pdp[2][2], dsp[2][2];  // packed bf16 P.drop and dS, [half][kt]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int qb = 2 * m2 + half;
      f32x4 s[2], dp[2];
      s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const frag aq = lfrag<T>(Qs, 16 * qb + r, ks, lane);
        const frag ao = lfrag<T>(Os, 16 * qb + r, ks, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          s[kt] = mma<T>(aq, kf[kt][ks], s[kt]);
          dp[kt] = mma<T>(ao, vf[kt][ks], dp[kt]);
        }
      }
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(L2s + 16 * qb + 4 * q);  // lse of queries 4q .. 4q + 3
      const f32x4 dl = *reinterpret_cast<const f32x4*>(Dls + 16 * qb + 4 * q);
      ks_.template block<T, DROP>(st * 32 + 16 * half, s, dp, l4, dl, pdp[half], dsp[half]);
      // dS^T block of this wave: row = key, 4 consecutive queries per lane
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        *reinterpret_cast<uint2*>(Sb + (k0 + 16 * kt + r) * FUSED_SRS + (16 * half + 4 * q) * 2) = dsp[half][kt];
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      mma_transposed_packed<T, NDT>(Os, m2, pdp[0][kt], pdp[1][kt], dv[kt], lane);
      mma_transposed_packed<T, NDT>(Qs, m2, dsp[0][kt], dsp[1][kt], dk[kt], lane);
    }
    __syncthreads();
    if (wave < 2 * NDT) {  // dQ^T tile (16 d x 16 queries) of this step: K^T [d][keys] . dS^T [keys][queries], all 256 key rows
      const int qb2 = wave / NDT, dt = wave - qb2 * NDT;
      const unsigned char* a0 = Kt + (4 * q + qq) * RS + (16 * dt + 4 * pp) * 2;
      const unsigned char* b0 = Sb + (4 * q + qq) * FUSED_SRS + (16 * qb2 + 4 * pp) * 2;
      union { struct { s16x4 lo, hi; } s; bf16x8 v; } ua[8], ub[8];
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        ua[kc].s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 32 * kc * RS));
        ua[kc].s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (32 * kc + 16) * RS));
        ub[kc].s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b0 + 32 * kc * FUSED_SRS));
        ub[kc].s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b0 + (32 * kc + 16) * FUSED_SRS));
      }
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};  // two chains: half the MFMA dependency latency
#pragma unroll
      for (int kc = 0; kc < 8; kc += 2) {
        acc0 = mfma16<T>(ua[kc].v, ub[kc].v, acc0);
        acc1 = mfma16<T>(ua[kc + 1].v, ub[kc + 1].v, acc1);
      }
      const int query = st * 32 + 16 * qb2 + r, d = 16 * dt + 4 * q;
      if (query < Tn && d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc0[e] + acc1[e];
          qsum[e] += v[e];
        }
        st4(dqkv + ((long)b * Tn + query) * ld + h * dh + d, v);
      }
    }
  }
#ifdef FH_ATTN_DBG_NOEPI
  if (scale == 12345.f)
#endif
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
    if (key >= Tn) continue;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dk[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + D + h * dh + d, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dv[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + 2 * D + h * dh + d, v);
      }
    }
  }
#ifdef FH_ATTN_DBG_NOEPI
  if (scale == 12345.f)
#endif
  if (dbias) {   // (uniform) -- the Q / dO tiles at the head of LDS are dead behind the loop's last barrier
    const bool kvalid[2] = {k0 + r < Tn, k0 + 16 + r < Tn};
    qkv_bias_sums<NDT>(reinterpret_cast<float*>(lds), dbias, dk, dv, kvalid, qsum, h, dh, D);
  }
  __syncthreads();   // (persistent walk: the next pair's prologue rewrites every LDS tile)
  }
}

// ------------------------------------------------------------------------------------------------ backward: fused, any length
// The fused form above keeps every Q / dO tile of a (batch, head) resident and owns at most 256 keys (8 waves x 32), so utterances
// past T2 = 256 -- train-clean-100's MEDIAN utterance has T2 = 349, its longest 613 -- fell back to the two-pass form, which computes
// P, the dropout mask and dS twice.  This kernel covers them: the keys are cut into nkb = ceil(Tn / 256) blocks of kpb keys (equal
// shares, a multiple of 32), one workgroup per (key block, head, batch).  A workgroup keeps ITS keys' K / V fragments in registers
// and its K block in LDS (for the dQ contraction) and STREAMS the 64-query (Q | dO | lse | delta) tiles through two buffers -- every
// thread fetches its 16-byte pieces of tile i + 1 into registers right behind the barrier of tile i, as the forward streams K / V --
// so LDS holds 2 tiles + K block + 2 dS^T scratch blocks = 111,616 B at any length.  dK / dV of a key block are complete; dQ of a
// query block is the sum over the key blocks: with more than one block each workgroup writes its fp32 partial to dq_part[kb] and
// dq_combine_kernel adds them in a fixed order (deterministic, no atomics) and rounds once.  delta = rowsum(out * dout) comes from
// attn_delta_kernel (a workgroup no longer sees a query's whole row of work before it needs delta).
template <typename T>
__global__ void __launch_bounds__(256) attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, float* __restrict__ delta,
                                                         int B, int Tn, int H, int dh) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, t, h), h fastest: neighbouring threads read neighbouring heads
  if (i >= (long)B * Tn * H) return;
  const int h = (int)(i % H);
  const long bt = i / H;
  const int t = (int)(bt % Tn), b = (int)(bt / Tn);
  const T* o = out + bt * H * dh + h * dh;
  const T* g = dout + bt * H * dh + h * dh;
  float part = 0.f;
  for (int c = 0; c < dh / 8; ++c) {
    float a[8], d[8];
    unpack16<T>(ld16(o + c * 8), a);
    unpack16<T>(ld16(g + c * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) part = fmaf(a[e], d[e], part);
  }
  delta[((long)b * H + h) * Tn + t] = part;
}
template <typename T>
__global__ void __launch_bounds__(256) dq_combine_kernel(const float* __restrict__ part, int nkb, T* __restrict__ dqkv, long rows, int D) {
  const int cpr = D / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cpr) return;
  const long row = i / cpr;
  const int c = (int)(i - row * cpr) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const float* p = part + ((long)kb * rows + row) * D + c;
    float v[8];
    ld4(p, v);
    ld4(p + 4, v + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
  st16(dqkv + row * 3 * D + c, pack16<T>(acc));
}

template <typename T, int NDT, bool DROP>
__global__ void __launch_bounds__(512) attn_bwd_fused_stream_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                                    T* __restrict__ dqkv, float* __restrict__ dq_part,
                                                                    const int* __restrict__ key_len, int B, int Tn, int H, int dh,
                                                                    int kpb, float scale, float scale_log2e, DropCfg drop,
                                                                    float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RS = AT<T>::RS;
  constexpr int PAIR = 2 * 64 * RS + 512;  // Q tile | dO tile | 64 x LSE | 64 x delta
  constexpr int NK = AT<T>::NK;
  typedef typename AT<T>::frag frag;
  unsigned char* Kt = lds + 2 * PAIR;
  unsigned char* Sc = Kt + 256 * RS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  // (b, h, key block) under the XCD-contiguous remap, as the resident form: neighbouring heads' workgroups share memory rows
  int b, h, kb;
  attn_block(b, h, kb);
  const int nkb = gridDim.x;
  const int kbase = kb * kpb;
  const int D = H * dh;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * Tn * ld + h * dh;
  const T* dobase = dout + (long)b * Tn * D + h * dh;
  const int klen = key_len ? min(key_len[b], Tn) : Tn;
  const int kend = min(Tn, kbase + kpb);       // keys [kbase, kend) are this workgroup's
  const int kvis = min(klen, kend);            // ... of which [kbase, kvis) are not padding
  const int k0 = kbase + wave * 32;            // this wave's 32 keys (a wave with k0 >= kvis only takes part in the barriers and the dQ contraction)
  const bool active = k0 < kvis;
  const int ntile = (Tn + 63) / 64;

  zero_lds<T>(lds, 2 * PAIR + 256 * RS + 2 * 256 * FUSED_SRS);   // (the dS^T rows of keys nobody owns stay zero for the whole launch)
  __syncthreads();
  stage_tiles<T>(4, kend, dh, [&](int t, const T*& src, long& sld, unsigned char*& dst, int& row0) {
    src = base + D;
    sld = ld;
    dst = Kt + t * 64 * RS;
    row0 = kbase + t * 64;
  });
  frag kf[2][NK], vf[2][NK];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      kf[kt][ks] = gfrag<T>(base + D + (long)key * ld, key < kend, ks, lane, dh);
      vf[kt][ks] = gfrag<T>(base + 2 * D + (long)key * ld, key < kend, ks, lane, dh);
    }
  }
  f32x4 dk[2][NDT], dv[2][NDT];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dk[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  KeyScore ks_;
  ks_.init(drop, b, h, H, Tn, kvis, k0, lane, scale, scale_log2e);

  // this thread's pieces of one (Q tile | dO tile) pair: piece i = row (i / cpr) % 64 of Q (i < 64 cpr) or dO
  constexpr int NPC = 2;   // 2 x 64 rows x (64 / 8) pieces at the widest head = 1024 = 2 x 512 threads
  const int cpr = dh / 8, npiece = 2 * 64 * cpr;
  V16 pre[NPC];
  int prow[NPC], pdst[NPC];
  const T* psrc[NPC];
  long pstep[NPC];   // elements per query row of the source
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int i = (int)threadIdx.x + j * 512;
    const int isdo = i >= 64 * cpr, rem = i - isdo * 64 * cpr, row = rem / cpr, c = rem - row * cpr;
    prow[j] = i < npiece ? row : -1;
    pdst[j] = isdo * 64 * RS + row * RS + c * 16;
    pstep[j] = isdo ? (long)D : ld;
    psrc[j] = (isdo ? dobase : base) + (long)row * pstep[j] + c * 8;
  }
  float pl = INFINITY, pd = 0.f;
  auto fetch = [&](int qt0) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      pre[j] = v16_zero();
      if (prow[j] >= 0 && qt0 + prow[j] < Tn) pre[j] = ld16(psrc[j] + (long)qt0 * pstep[j]);
    }
    if (threadIdx.x < 64) {
      const int query = qt0 + threadIdx.x;
      pl = query < Tn ? lse[((long)b * H + h) * Tn + query] : INFINITY;
      pd = query < Tn ? delta[((long)b * H + h) * Tn + query] : 0.f;
    }
  };
  auto deposit = [&](unsigned char* sb) {
#pragma unroll
    for (int j = 0; j < NPC; ++j)
      if (prow[j] >= 0) st16(sb + pdst[j], pre[j]);
    if (threadIdx.x < 64) {
      float* L2w = reinterpret_cast<float*>(sb + 2 * 64 * RS);
      L2w[threadIdx.x] = pl;
      L2w[64 + threadIdx.x] = pd;
    }
  };
  fetch(0);
  const int qq = r >> 2, pp = r & 3;
  f32x4 qsum = f32x4{0.f, 0.f, 0.f, 0.f};   // column sums of this wave's dQ tiles (q bias gradient)
  for (int tile = 0; tile < ntile; ++tile) {
    unsigned char* sb = lds + (tile & 1) * PAIR;
    // the buffer's readers (the score steps of tile - 2) finished before the barriers this wave has passed since
    deposit(sb);
    __syncthreads();   // (also orders the K block staged above in front of the first contraction)
    if (tile + 1 < ntile) fetch((tile + 1) * 64);
    const unsigned char* Qs = sb;
    const unsigned char* Os = Qs + 64 * RS;
    const float* L2s = reinterpret_cast<const float*>(Qs + 2 * 64 * RS);
    const float* Dls = L2s + 64;
#pragma unroll 1
    for (int m2 = 0; m2 < 2; ++m2) {
      const int st = 2 * tile + m2;
      if (st * 32 >= Tn) break;   // (uniform: the last tile of an odd number of 32-query steps)
      unsigned char* Sb = Sc + (st & 1) * 256 * FUSED_SRS;
      if (active) {
        uint2 pdp[2][2], dsp[2][2];  // packed P.drop and dS, [half][kt]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int qb = 2 * m2 + half;
          f32x4 s[2], dp[2];
          s[0] = s[1] = dp[0] = dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const frag aq = lfrag<T>(Qs, 16 * qb + r, ks, lane);
            const frag ao = lfrag<T>(Os, 16 * qb + r, ks, lane);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              s[kt] = mma<T>(aq, kf[kt][ks], s[kt]);
              dp[kt] = mma<T>(ao, vf[kt][ks], dp[kt]);
            }
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(L2s + 16 * qb + 4 * q);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(Dls + 16 * qb + 4 * q);
          ks_.template block<T, DROP>(st * 32 + 16 * half, s, dp, l4, dl, pdp[half], dsp[half]);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
            *reinterpret_cast<uint2*>(Sb + (wave * 32 + 16 * kt + r) * FUSED_SRS + (16 * half + 4 * q) * 2) = dsp[half][kt];
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          mma_transposed_packed<T, NDT>(Os, m2, pdp[0][kt], pdp[1][kt], dv[kt], lane);
          mma_transposed_packed<T, NDT>(Qs, m2, dsp[0][kt], dsp[1][kt], dk[kt], lane);
        }
      }
      __syncthreads();
      if (wave < 2 * NDT) {  // dQ^T tile (16 d x 16 queries) of this step over this block's keys: K^T [d][keys] . dS^T [keys][queries]
        const int qb2 = wave / NDT, dt = wave - qb2 * NDT;
        const unsigned char* a0 = Kt + (4 * q + qq) * RS + (16 * dt + 4 * pp) * 2;
        const unsigned char* b0 = Sb + (4 * q + qq) * FUSED_SRS + (16 * qb2 + 4 * pp) * 2;
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } ua[8], ub[8];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          ua[kc].s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 32 * kc * RS));
          ua[kc].s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + (32 * kc + 16) * RS));
          ub[kc].s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b0 + 32 * kc * FUSED_SRS));
          ub[kc].s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b0 + (32 * kc + 16) * FUSED_SRS));
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 8; kc += 2) {
          acc0 = mfma16<T>(ua[kc].v, ub[kc].v, acc0);
          acc1 = mfma16<T>(ua[kc + 1].v, ub[kc + 1].v, acc1);
        }
        const int query = st * 32 + 16 * qb2 + r, d = 16 * dt + 4 * q;
        if (query < Tn && d < dh) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc0[e] + acc1[e];
            qsum[e] += v[e];
          }
          if (nkb == 1) st4(dqkv + ((long)b * Tn + query) * ld + h * dh + d, v);
          else st4(dq_part + (((long)kb * B + b) * Tn + query) * D + h * dh + d, v);
        }
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = k0 + 16 * kt + r;
    if (key >= kend) continue;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dh) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dk[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + D + h * dh + d, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dv[kt][dt][e];
        st4(dqkv + ((long)b * Tn + key) * ld + 2 * D + h * dh + d, v);
      }
    }
  }
  if (dbias) {   // (uniform) -- both Q / dO buffers are dead behind the loop's last barrier
    const bool kvalid[2] = {k0 + r < kend, k0 + 16 + r < kend};
    qkv_bias_sums<NDT>(reinterpret_cast<float*>(lds), dbias, dk, dv, kvalid, qsum, h, dh, D);
  }
}

int check_dims(int dtype, int H, int dh, int dh_max = 64) {
  const int vec = dtype == 0 ? 4 : 8;
  if (dh % vec || dh % 4 || dh > dh_max || H <= 0) return -601;  // NDT = 3 builds cover dh <= 48 (student dh = 40), NDT = 4 dh <= 64
  return 0;
}

template <typename T> int resident_tiles(int n) {  // 0 = tiled path
  static int mode = -1;  // FH_ATTN_RESIDENT=0 forces the tiled path (A/B comparisons)
  if (mode < 0) {
    const char* e = getenv("FH_ATTN_RESIDENT");
    mode = (e && e[0] == '0') ? 0 : 1;
  }
  if (!mode) return 0;
  const int ntile = (n + 63) / 64;
  return (size_t)ntile * (2 * 64 * AT<T>::RS + 512) <= 150 * 1024 ? ntile : 0;
}
template <typename K> void set_lds(K kernel, size_t bytes) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Workgroup width.  Resident mode keeps every K/V (or Q/dO) tile of one (batch, head) in LDS, which leaves room for ONE
// workgroup per CU: with 4 waves that is one wave per SIMD and every LDS / shuffle / exp latency is exposed.  16 waves
// (512 queries) per workgroup share the same resident tiles -- 4 waves per SIMD at <= 128 VGPRs.
template <typename T, int NDT, int NW>
int fwd_launch_t(const T* qkv, T* out, float* lse, const int* key_len, int B, int Tn, int H, int dh, float sl2, fh_drop drop,
                 hipStream_t st) {
  dim3 grid((Tn + 32 * NW - 1) / (32 * NW), H, B), block(NW * 64);
  const size_t smem = (size_t)2 * 2 * 64 * AT<T>::RS;  // two (K | V) tile buffers
  if (drop.p > 0.f) {
    set_lds(attn_fwd_kernel<T, NDT, NW, true>, smem);
    hipLaunchKernelGGL((attn_fwd_kernel<T, NDT, NW, true>), grid, block, smem, st, qkv, out, lse, key_len, B, Tn, H, dh, sl2, drop);
  } else {
    set_lds(attn_fwd_kernel<T, NDT, NW, false>, smem);
    hipLaunchKernelGGL((attn_fwd_kernel<T, NDT, NW, false>), grid, block, smem, st, qkv, out, lse, key_len, B, Tn, H, dh, sl2, drop);
  }
  FH_CHECK_LAUNCH();
  return 0;
}
template <typename T, int NDT, int NW>
int bwd_launch_t(const T* qkv, const T* out, const T* dout, const float* lse, float* delta, T* dqkv, const int* key_len, int B,
                 int Tn, int H, int dh, float scale, float sl2, fh_drop drop, int res, hipStream_t st) {
  dim3 grid((Tn + 32 * NW - 1) / (32 * NW), H, B), block(NW * 64);
  const size_t s1 = (size_t)(res ? res : 1) * 2 * 64 * AT<T>::RS, s2 = (size_t)(res ? res : 1) * (2 * 64 * AT<T>::RS + 512);
  set_lds(attn_bwd_dq_kernel<T, NDT, NW>, s1);
  set_lds(attn_bwd_dkv_kernel<T, NDT, NW>, s2);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NDT, NW>), grid, block, s1, st, qkv, out, dout, lse, delta, dqkv, key_len, B, Tn, H, dh,
                     scale, sl2, drop, res);
  FH_CHECK_LAUNCH();
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NDT, NW>), grid, block, s2, st, qkv, dout, lse, delta, dqkv, key_len, B, Tn, H, dh, scale,
                     sl2, drop, res);
  FH_CHECK_LAUNCH();
  return 0;
}
template <typename T, int NDT>
int bwd_fused_launch(const T* qkv, const T* out, const T* dout, const float* lse, float* delta, T* dqkv,
                     const int* key_len, int B, int Tn, int H, int dh, float scale, float sl2, fh_drop drop, hipStream_t st, float* dbias) {
  const size_t smem = 4 * (2 * 64 * AT<bf16>::RS + 512) + 256 * AT<bf16>::RS + 2 * 256 * FUSED_SRS;
  // persistent walk (one workgroup per CU, each over a contiguous range of (batch, head) pairs) once there are at least two pairs per
  // CU; FH_ATTN_PERSIST=0 keeps one pair per workgroup (A/B comparisons)
  static int persist = -1, ncu = 0;
  if (persist < 0) {
    const char* e = getenv("FH_ATTN_PERSIST");
    persist = (e && e[0] == '0') ? 0 : 1;
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  // (data-parallel runs hold CUs back for the collectives' kernels -- fh_gemm_reserve_cus / fh_gemm_stream_reserve_cus: a persistent
  // workgroup owns its CU for the whole launch, so the walk leaves them free exactly as the streaming GEMM loop does)
  const int sres = fh_stream_reserve_cus(st), res = sres > fh_g_reserve_cus ? sres : fh_g_reserve_cus;
  const int pgrid = ncu - res > 8 ? ncu - res : 8;
  const dim3 grid = (persist && (long)B * H >= 2L * pgrid) ? dim3(pgrid, 1, 1) : dim3(1, H, B);
  if (drop.p > 0.f) {
    set_lds(attn_bwd_fused_kernel<T, NDT, true>, smem);
    hipLaunchKernelGGL((attn_bwd_fused_kernel<T, NDT, true>), grid, dim3(512), smem, st, qkv, out, dout, lse, delta, dqkv, key_len,
                       B, Tn, H, dh, scale, sl2, drop, dbias);
  } else {
    set_lds(attn_bwd_fused_kernel<T, NDT, false>, smem);
    hipLaunchKernelGGL((attn_bwd_fused_kernel<T, NDT, false>), grid, dim3(512), smem, st, qkv, out, dout, lse, delta, dqkv, key_len,
                       B, Tn, H, dh, scale, sl2, drop, dbias);
  }
  FH_CHECK_LAUNCH();
  return 0;
}
// scratch of the any-length fused backward: nkb x B x Tn x D floats (0 when one key block suffices)
int fused_stream_blocks(int Tn) { return (Tn + 255) / 256; }
template <typename T, int NDT>
int bwd_fused_stream_launch(const T* qkv, const T* out, const T* dout, const float* lse, float* delta, T* dqkv, float* dq_part,
                            const int* key_len, int B, int Tn, int H, int dh, float scale, float sl2, fh_drop drop, hipStream_t st,
                            float* dbias) {
  const int nkb = fused_stream_blocks(Tn);
  const int kpb = (((Tn + nkb - 1) / nkb) + 31) / 32 * 32;
  const size_t smem = 2 * (2 * 64 * AT<T>::RS + 512) + 256 * AT<T>::RS + 2 * 256 * FUSED_SRS;
  const long nd = (long)B * Tn * H;
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, out, dout, delta, B, Tn, H, dh);
  FH_CHECK_LAUNCH();
  if (drop.p > 0.f) {
    set_lds(attn_bwd_fused_stream_kernel<T, NDT, true>, smem);
    hipLaunchKernelGGL((attn_bwd_fused_stream_kernel<T, NDT, true>), dim3(nkb, H, B), dim3(512), smem, st, qkv, dout, lse, delta, dqkv, dq_part,
                       key_len, B, Tn, H, dh, kpb, scale, sl2, drop, dbias);
  } else {
    set_lds(attn_bwd_fused_stream_kernel<T, NDT, false>, smem);
    hipLaunchKernelGGL((attn_bwd_fused_stream_kernel<T, NDT, false>), dim3(nkb, H, B), dim3(512), smem, st, qkv, dout, lse, delta, dqkv, dq_part,
                       key_len, B, Tn, H, dh, kpb, scale, sl2, drop, dbias);
  }
  FH_CHECK_LAUNCH();
  if (nkb > 1) {
    const long rows = (long)B * Tn, n = rows * (H * dh / 8);
    hipLaunchKernelGGL(dq_combine_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dq_part, nkb, dqkv, rows, H * dh);
    FH_CHECK_LAUNCH();
  }
  return 0;
}
bool fused_bwd_enabled() {  // FH_ATTN_FUSED=0 selects the two-pass backward (A/B comparisons)
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("FH_ATTN_FUSED");
    mode = (e && e[0] == '0') ? 0 : 1;
  }
  return mode != 0;
}
int wide_waves(int res, int Tn, int max_nw = 16) {  // waves per workgroup: FH_ATTN_WAVES = 4 | 8 | 16 overrides
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("FH_ATTN_WAVES");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 4 || forced == 8 || forced == 16) return forced;
  if (max_nw == 0) return 0;   // (query of the override alone)
  if (!res) return 4;
  const int nw = Tn > 256 ? 16 : (Tn > 128 ? 8 : 4);
  return nw < max_nw ? nw : max_nw;
}
// Forward: the width that wastes the fewest query rows (a workgroup covers 32 rows per wave: T = 349 is 68 % of two 8-wave or one
// 16-wave workgroup but 91 % of three 4-wave ones -- 180 -> 148 us at B = 128, 40-wide heads; the median LibriSpeech utterance), 8 waves
// on a tie (K / V are streamed, so 16 waves share nothing more and measured 2 - 7 % slower at T = 499).  FH_ATTN_WAVES overrides.
int fwd_waves(int Tn) {
  const int forced = wide_waves(0, 0, 0);
  if (forced == 4 || forced == 8 || forced == 16) return forced;
  const int pad8 = (Tn + 255) / 256 * 256, pad4 = (Tn + 127) / 128 * 128;
  return pad4 < pad8 ? 4 : 8;
}
}  // namespace

// 16-bit launchers (T = bf16 | f16)
namespace {
template <typename T>
int fwd16(const void* qkv, void* out, float* lse, const int* key_len, int B, int Tn, int H, int dh, float sl2, fh_drop drop, hipStream_t st) {
  const T* q = (const T*)qkv;
  // Workgroup width: 16 waves share one stream of K / V tiles, but at small batches the grid must still cover the chip -- halve
  // the width while the launch has fewer workgroups than CUs (B = 4, T = 499: 48 workgroups of 16 waves -> 192 of 4)
  int nw = fwd_waves(Tn);
  while (nw > 4 && (long)B * H * ((Tn + 32 * nw - 1) / (32 * nw)) < 256) nw >>= 1;
  if (dh > 48) {  // 64-wide heads: the teacher (HuBERT Base 768 / 12, Large 1024 / 16), the ex.yaml student
    switch (nw) {
      case 16: return fwd_launch_t<T, 4, 16>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
      case 8: return fwd_launch_t<T, 4, 8>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
      default: return fwd_launch_t<T, 4, 4>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
    }
  }
  switch (nw) {
    case 16: return fwd_launch_t<T, 3, 16>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
    case 8: return fwd_launch_t<T, 3, 8>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
    default: return fwd_launch_t<T, 3, 4>(q, (T*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
  }
}
template <typename T>
int bwd16(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, const int* key_len, int B, int Tn,
          int H, int dh, float scale, float sl2, fh_drop drop, hipStream_t st, float* dq_scratch, float* dbias, bool* bias_done) {
  const int res = resident_tiles<T>(Tn);
  const int nw = wide_waves(res, Tn, 8);  // the 16-wave backward builds spill under the 128-VGPR cap
  const T *q = (const T*)qkv, *o = (const T*)out, *d = (const T*)dout;
  if (res && Tn <= 256 && Tn > 64 && fused_bwd_enabled()) {
    *bias_done = dbias != nullptr;
    if (dh > 48) return bwd_fused_launch<T, 4>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, st, dbias);
    return bwd_fused_launch<T, 3>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, st, dbias);
  }
  if (Tn > 256 && dq_scratch && (H * dh) % 8 == 0 && fused_bwd_enabled()) {   // longer utterances: the streamed fused form
    *bias_done = dbias != nullptr;
    if (dh > 48) return bwd_fused_stream_launch<T, 4>(q, o, d, lse, delta, (T*)dqkv, dq_scratch, key_len, B, Tn, H, dh, scale, sl2, drop, st, dbias);
    return bwd_fused_stream_launch<T, 3>(q, o, d, lse, delta, (T*)dqkv, dq_scratch, key_len, B, Tn, H, dh, scale, sl2, drop, st, dbias);
  }
  if (dh > 48) {
    if (nw == 8) return bwd_launch_t<T, 4, 8>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
    return bwd_launch_t<T, 4, 4>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
  }
  if (nw == 8) return bwd_launch_t<T, 3, 8>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
  return bwd_launch_t<T, 3, 4>(q, o, d, lse, delta, (T*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
}
}  // namespace

int fh_attention_fwd_launch(int dtype, const void* qkv, void* out, float* lse, const int* key_len, int B, int Tn, int H,
                            int dh, float scale, fh_drop drop, hipStream_t st) {
  if (int rc = check_dims(dtype, H, dh, 64)) return rc;
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == 0) {  // fp32 parity mode: 4-wave workgroups (the wider builds are tuned for the bf16 register budget)
    const float* q = (const float*)qkv;
    if (dh > 48) return fwd_launch_t<float, 4, 4>(q, (float*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
    return fwd_launch_t<float, 3, 4>(q, (float*)out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
  }
  return dtype == 2 ? fwd16<f16>(qkv, out, lse, key_len, B, Tn, H, dh, sl2, drop, st)
                    : fwd16<bf16>(qkv, out, lse, key_len, B, Tn, H, dh, sl2, drop, st);
}

long fh_attention_bwd_scratch_floats(int dtype, int B, int Tn, int H, int dh) {
  if (dtype == 0 || Tn <= 256) return 0;
  return (long)fused_stream_blocks(Tn) * B * Tn * H * dh;
}
int fh_attention_bwd_launch(int dtype, const void* qkv, const void* out, const void* dout, const float* lse,
                            float* delta, void* dqkv, const int* key_len, int B, int Tn, int H, int dh, float scale,
                            fh_drop drop, hipStream_t st, float* dq_scratch, float* dbias) {
  if (int rc = check_dims(dtype, H, dh, 64)) return rc;
  const float sl2 = scale * 1.4426950408889634f;
  bool bias_done = false;
  int rc;
  if (dtype == 0) {
    const int res = resident_tiles<float>(Tn);
    if (dh > 48)
      rc = bwd_launch_t<float, 4, 4>((const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
    else
      rc = bwd_launch_t<float, 3, 4>((const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, res, st);
  } else {
    rc = dtype == 2 ? bwd16<f16>(qkv, out, dout, lse, delta, dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, st, dq_scratch, dbias, &bias_done)
                    : bwd16<bf16>(qkv, out, dout, lse, delta, dqkv, key_len, B, Tn, H, dh, scale, sl2, drop, st, dq_scratch, dbias, &bias_done);
  }
  // dbias [3 H dh] += column sums of dqkv: out of the fused kernels' accumulators, else by a pass over the stored gradient
  if (!rc && dbias && !bias_done) rc = fh_colsum_launch(dtype, dqkv, dbias, B * Tn, 3 * H * dh, 3L * H * dh, 1, 0, 0, 0, 1.f, st);
  return rc;
}
