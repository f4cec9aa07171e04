// Attention-logit and value-relation transfer losses fused with their gradient (reference utils/utils.py:244-278 patched in at
// train.py:64-78; losses train.py:327-370).
//
// The reference's patched last layer returns  attn = (q d^-1/2) k^T  (padded keys filled with -inf) and
// v_rel = (v d^-1/2) v^T  as (B H) x T x T tensors for student and teacher, and the loss reduces them:
//   mse  : mean over the finite entries of (S_s - S_t)^2            (train.py:331-341: inf / nan entries scrubbed and discounted)
//   kldiv: mean over rows of  sum_k softmax(S_t) (log softmax(S_t) - log softmax(S_s))     (train.py:342-349, 358-367)
// Both relations are functions of the packed q|k|v projection only.  Here NOTHING of size T x T reaches HBM: a workgroup owns
// one (utterance, head, 64-query block), streams 64-key tiles of the student's and the teacher's operands through LDS, forms
// the two logit tiles, reduces the loss, and turns the tile of dL/dS_s straight into  dA += scale G B  and  dB += scale G^T A
// (A / B = the row / column operand: q / k for the logits, v / v for the value relation) -- fp32 throughout.  kl-div rows need
// their log-sum-exp first: one extra sweep over the key tiles (online max / sum), then the sweep above with normalised rows.
//
// Masked keys (index >= min(student valid keys, teacher valid keys) of the utterance): the mse form reproduces the reference's
// scrubbing exactly (such entries are inf or nan there, contribute 0, and their count leaves the denominator).  For kl-div the
// reference's arithmetic yields nan on a padded batch (0 * -inf inside F.kl_div); masked keys here simply carry zero
// probability in both distributions and contribute nothing.
#include "kernels.h"
#include "common.cuh"
#include "attn_frag.cuh"

namespace {

constexpr int RT = 64;        // query rows / key columns per tile
constexpr int RMAXD = 64;     // widest head
constexpr int RSTR = RMAXD + 1;

struct RelArgs {
  const void* s_qkv;
  const void* t_qkv;
  float* d_qkv;        // fp32 [B][T][3 Ds], accumulated (or null)
  float* partials;     // [gridDim] loss partial sums
  const int* s_len;
  const int* t_len;
  int B, T, H, Ds, Dt;
  int a_off_s, b_off_s, a_off_t, b_off_t;  // element offsets of the row / column operand inside a packed q|k|v row
  float scale_s, scale_t;
  int mode;            // 0 mse, 1 kl-div, 2 emit the student's relation tensor to `out`, 3 gradient of it from `gin`
  float* out;          // mode 2: fp32 [B H][T][T]
  const float* gin;    // mode 3: fp32 [B H][T][T] upstream gradient (entries of masked keys are ignored)
  int masked;          // key masks apply (attention logits); the value relation is unmasked
  float gscale;        // weight / normaliser for kl (rows); for mse the normaliser is computed from the lengths
  float weight;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i) { return to_f(p[i]); }

// stage rows [r0, r0 + RT) of one head's operand (dh wide) into LDS as fp32, zero beyond Tn
template <typename T>
__device__ __forceinline__ void stage(float* dst, const T* src, long row_stride, int r0, int Tn, int dh) {
  for (int i = threadIdx.x; i < RT * dh; i += blockDim.x) {
    const int r = i / dh, d = i - r * dh;
    const int t = r0 + r;
    dst[r * RSTR + d] = t < Tn ? to_f(src[(long)t * row_stride + d]) : 0.f;
  }
}

__device__ __forceinline__ void tile_dot(const float* A, const float* Bm, int dh, int ty, int tx, float scale, float (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int d = 0; d < dh; ++d) {
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = A[(4 * ty + i) * RSTR + d];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Bm[(4 * tx + j) * RSTR + d];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] *= scale;
}

// reductions over the 16 lanes (tx) that share a row group
__device__ __forceinline__ float row_max16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename TS, typename TT>
__global__ void __launch_bounds__(256) relation_kd_kernel(RelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // student row operand   [RT][RSTR]
  float* At = As + RT * RSTR;       // teacher row operand
  float* Bs = At + RT * RSTR;       // student column operand
  float* Bt = Bs + RT * RSTR;       // teacher column operand
  float* Gs = Bt + RT * RSTR;       // dL/dS_s tile
  __shared__ float red[256];
  __shared__ float s_den;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, dhs = a.Ds / a.H, dht = a.Dt / a.H;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int q0 = qb * RT;
  const TS* sq = reinterpret_cast<const TS*>(a.s_qkv) + (long)b * T * 3 * a.Ds + h * dhs;
  const TT* tq = reinterpret_cast<const TT*>(a.t_qkv) + (long)b * T * 3 * a.Dt + h * dht;
  int klen = T;
  if (a.masked) {
    if (a.s_len) klen = min(klen, a.s_len[b]);
    if (a.t_len) klen = min(klen, a.t_len[b]);
  }
  if (threadIdx.x == 0) {
    // mse normaliser: all entries minus those of masked key columns (train.py:336-341), over the whole batch
    double den = (double)a.B * a.H * T * (double)T;
    if (a.masked && a.mode == 0)
      for (int bb = 0; bb < a.B; ++bb) {
        int kl = T;
        if (a.s_len) kl = min(kl, a.s_len[bb]);
        if (a.t_len) kl = min(kl, a.t_len[bb]);
        den -= (double)a.H * (T - kl) * (double)T;
      }
    s_den = (float)den;
  }
  const bool has_t = a.t_qkv != nullptr;
  stage(As, sq + a.a_off_s, 3L * a.Ds, q0, T, dhs);
  if (has_t) stage(At, tq + a.a_off_t, 3L * a.Dt, q0, T, dht);
  __syncthreads();
  const float gsc = a.mode == 0 ? a.weight * 2.0f / s_den : a.weight * a.gscale;
  const int nkt = (T + RT - 1) / RT;
  float m_s[4], l_s[4], m_t[4], l_t[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m_s[i] = m_t[i] = -INFINITY; l_s[i] = l_t[i] = 0.f; }
  if (a.mode == 1) {
    // ---- sweep 1: row-wise log-sum-exp of both logit matrices (online)
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();
      stage(Bs, sq + a.b_off_s, 3L * a.Ds, kt * RT, T, dhs);
      stage(Bt, tq + a.b_off_t, 3L * a.Dt, kt * RT, T, dht);
      __syncthreads();
      float ss[4][4], st[4][4];
      tile_dot(As, Bs, dhs, ty, tx, a.scale_s, ss);
      tile_dot(At, Bt, dht, ty, tx, a.scale_t, st);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool v = kt * RT + 4 * tx + j < klen;
          if (!v) { ss[i][j] = -INFINITY; st[i][j] = -INFINITY; }
          ms = fmaxf(ms, ss[i][j]);
          mt = fmaxf(mt, st[i][j]);
        }
        ms = fmaxf(row_max16(ms), m_s[i]);
        mt = fmaxf(row_max16(mt), m_t[i]);
        float es = 0.f, et = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          es += ss[i][j] == -INFINITY ? 0.f : __expf(ss[i][j] - ms);
          et += st[i][j] == -INFINITY ? 0.f : __expf(st[i][j] - mt);
        }
        es = row_sum16(es);
        et = row_sum16(et);
        l_s[i] = l_s[i] * (m_s[i] == -INFINITY ? 0.f : __expf(m_s[i] - ms)) + es;
        l_t[i] = l_t[i] * (m_t[i] == -INFINITY ? 0.f : __expf(m_t[i] - mt)) + et;
        m_s[i] = ms;
        m_t[i] = mt;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // m <- log-sum-exp
      m_s[i] += __logf(l_s[i]);
      m_t[i] += __logf(l_t[i]);
    }
  }
  // ---- sweep 2: loss and gradient tiles
  float loss = 0.f;
  // dA of this query block: thread -> row (threadIdx / 4), a quarter of the head width
  const int gr = threadIdx.x >> 2, gq = threadIdx.x & 3;
  const int dper = (dhs + 3) / 4;
  float dA[RMAXD / 4];
#pragma unroll
  for (int e = 0; e < RMAXD / 4; ++e) dA[e] = 0.f;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage(Bs, sq + a.b_off_s, 3L * a.Ds, kt * RT, T, dhs);
    if (has_t) stage(Bt, tq + a.b_off_t, 3L * a.Dt, kt * RT, T, dht);
    __syncthreads();
    float ss[4][4], st[4][4];
    if (a.mode != 3) tile_dot(As, Bs, dhs, ty, tx, a.scale_s, ss);
    if (has_t) tile_dot(At, Bt, dht, ty, tx, a.scale_t, st);
    if (a.mode >= 2) {   // the relation tensor itself (API surface, utils/utils.py:263,265) / its backward
      float* o = a.out + ((long)(b * a.H + h) * T) * T;
      const float* gi = a.gin + ((long)(b * a.H + h) * T) * T;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qr = q0 + 4 * ty + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kcol = kt * RT + 4 * tx + j;
          float g = 0.f;
          if (qr < T && kcol < T) {
            if (a.mode == 2) o[(long)qr * T + kcol] = kcol < klen ? ss[i][j] : -INFINITY;
            else if (kcol < klen) g = gi[(long)qr * T + kcol];
          }
          Gs[(4 * ty + i) * RSTR + 4 * tx + j] = g;
        }
      }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool rv = q0 + 4 * ty + i < T;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kcol = kt * RT + 4 * tx + j;
        const bool v = rv && kcol < klen;     // (kcol < klen <= T)
        float g = 0.f;
        if (v) {
          if (a.mode == 0) {
            const float d = ss[i][j] - st[i][j];
            loss = fmaf(d, d, loss);
            g = d * gsc;
          } else {
            const float lps = ss[i][j] - m_s[i], lpt = st[i][j] - m_t[i];
            const float ps = __expf(lps), pt = __expf(lpt);
            loss = fmaf(pt, lpt - lps, loss);
            g = (ps - pt) * gsc;
          }
        }
        Gs[(4 * ty + i) * RSTR + 4 * tx + j] = g;
      }
    }
    if (a.d_qkv && a.mode != 2) {
      __syncthreads();
      // dA[r][d] += scale * sum_j G[r][j] B[j][d]
      for (int e = 0; e < dper; ++e) {
        const int d = gq * dper + e;
        if (d < dhs) {
          float s = 0.f;
          for (int j = 0; j < RT; ++j) s = fmaf(Gs[gr * RSTR + j], Bs[j * RSTR + d], s);
          dA[e] += s;
        }
      }
      // dB[j][d] += scale * sum_r G[r][j] A[r][d]     (one contribution per query block: fp32 atomics)
      const int kj = kt * RT + gr;
      if (kj < T) {
        float* dst = a.d_qkv + ((long)b * T + kj) * 3 * a.Ds + a.b_off_s + h * dhs;
        for (int e = 0; e < dper; ++e) {
          const int d = gq * dper + e;
          if (d < dhs) {
            float s = 0.f;
            for (int r = 0; r < RT; ++r) s = fmaf(Gs[r * RSTR + gr], As[r * RSTR + d], s);
            atomicAdd(dst + d, s * a.scale_s);
          }
        }
      }
    }
  }
  if (a.d_qkv && a.mode != 2 && q0 + gr < T) {
    float* dst = a.d_qkv + ((long)b * T + q0 + gr) * 3 * a.Ds + a.a_off_s + h * dhs;
    for (int e = 0; e < dper; ++e) {
      const int d = gq * dper + e;
      if (d < dhs) atomicAdd(dst + d, dA[e] * a.scale_s);
    }
  }
  red[threadIdx.x] = loss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && a.partials) a.partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------ 16-bit operands: the same losses on the matrix cores
// The fp32 kernel above forms its 64 x 64 logit tiles with VALU FMAs: 5 - 6 TFLOP/s, 23 - 28 ms per loss term at B = 128, T = 499 -- half
// a training step for an optional term.  With 16-bit q|k|v (the bf16 / fp16 modes) the tiles are MFMA products of exactly the same
// inputs (a product of two 16-bit values is exact in fp32; only the order of the fp32 additions differs), so the logits, the loss and
// G = dL/dS_s are as accurate as before.  Decomposition, as the attention kernels (attn_frag.cuh):
//   * a workgroup (4 waves) owns 64 rows of ONE operand (SWAP = false: the row operand -- queries; SWAP = true: the column operand --
//     keys) and sweeps 64-row tiles of the other through LDS; wave w computes the TRANSPOSED tiles  M^T[swept row][own row]  for its
//     16 own rows, student and teacher, so a lane owns one own row (column lane & 15);
//   * the tile of G^T is, in accumulator layout, directly the B operand of  dOwn^T[d][own] += Swept_s^T[d][swept] G^T[swept][own]
//     (mma_transposed: the swept student tile is read transposed from the same LDS image).  G is carried into that product as a
//     head + remainder PAIR of 16-bit values (two MFMAs), unscaled (O(1): the 1 / (B H T) normaliser would underflow fp16), and the
//     fp32 sums are scaled once at the end -- the gradient keeps fp32-grade accuracy (tests: 2e-4 of its norm);
//   * so the row operand's gradient comes from the SWAP = false launch and the column operand's from the SWAP = true launch, each
//     complete per workgroup: no atomics, bit-reproducible.  kl-div rows need log-sum-exp of both logit rows: the SWAP = false launch
//     computes them in a first sweep (its lanes own the queries) and leaves them in the scratch for the SWAP = true launch.
template <typename T> __device__ __forceinline__ void split16(float v, T& hi, T& lo) {
  hi = (T)v;
  lo = (T)(v - (float)hi);
}
template <typename TS, typename TT, int NDT, bool SWAP>
__global__ void __launch_bounds__(256) relation_mfma_kernel(RelArgs a, float* __restrict__ lse /* [2][B H Tp] */, int Tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int TILE = 64 * AT<TS>::RS;
  unsigned char* Ws = lds;               // swept student tile [64][RS]
  unsigned char* Wt = lds + TILE;        // swept teacher tile
  float* lrow = reinterpret_cast<float*>(lds + 2 * TILE);   // SWAP: lse_s | lse_t of the swept (query) rows of the tile
  __shared__ float red[256];
  const int ob = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int T = a.T, dhs = a.Ds / a.H, dht = a.Dt / a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const long lds_ = 3L * a.Ds, ldt_ = 3L * a.Dt;
  const TS* sq = reinterpret_cast<const TS*>(a.s_qkv) + (long)b * T * lds_ + h * dhs;
  const TT* tq = reinterpret_cast<const TT*>(a.t_qkv) + (long)b * T * ldt_ + h * dht;
  const int own_off_s = SWAP ? a.b_off_s : a.a_off_s, swp_off_s = SWAP ? a.a_off_s : a.b_off_s;
  const int own_off_t = SWAP ? a.b_off_t : a.a_off_t, swp_off_t = SWAP ? a.a_off_t : a.b_off_t;
  int klen = T;
  if (a.masked) {
    if (a.s_len) klen = min(klen, a.s_len[b]);
    if (a.t_len) klen = min(klen, a.t_len[b]);
  }
  // mse normaliser: all entries minus those of masked key columns (train.py:336-341), over the whole batch
  float den = 1.f;
  if (a.mode == 0) {
    double dd = (double)a.B * a.H * T * (double)T;
    if (a.masked)
      for (int bb = 0; bb < a.B; ++bb) {
        int kl = T;
        if (a.s_len) kl = min(kl, a.s_len[bb]);
        if (a.t_len) kl = min(kl, a.t_len[bb]);
        dd -= (double)a.H * (T - kl) * (double)T;
      }
    den = (float)dd;
  }
  const float gsc = a.mode == 0 ? a.weight * 2.0f / den : a.weight * a.gscale;
  const int own = ob * 64 + wave * 16 + r;          // this lane's own row (query, or key when SWAP)
  const bool own_in = own < T;
  // own-row fragments (global -> registers, once)
  typename AT<TS>::frag os[AT<TS>::NK];
  typename AT<TT>::frag ot[AT<TT>::NK];
#pragma unroll
  for (int ks = 0; ks < AT<TS>::NK; ++ks) os[ks] = gfrag<TS>(sq + own_off_s + (long)own * lds_, own_in, ks, lane, dhs);
#pragma unroll
  for (int ks = 0; ks < AT<TT>::NK; ++ks) ot[ks] = gfrag<TT>(tq + own_off_t + (long)own * ldt_, own_in, ks, lane, dht);
  zero_lds<TS>(lds, 2 * TILE);   // (columns dh .. 63 of the padded rows are never written again)
  __syncthreads();
  const int nt = (T + 63) / 64;
  const float L2E = 1.4426950408889634f;
  const long lbase = ((long)b * a.H + h) * Tp;
  float* lse_s = lse ? lse + lbase : nullptr;
  float* lse_t = lse ? lse + (long)a.B * a.H * Tp + lbase : nullptr;

  auto tiles = [&](int t0, f32x4 (&ss)[4], f32x4 (&st)[4]) {   // M^T tiles of the 64 swept rows from t0 (raw dot products x scale)
    __syncthreads();                                            // the previous tile's readers are done
    stage_tile<TS>(Ws, sq + swp_off_s, lds_, t0, T, dhs);
    stage_tile<TT>(Wt, tq + swp_off_t, ldt_, t0, T, dht);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      ss[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
      st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < AT<TS>::NK; ++ks) ss[kb] = mma<TS>(lfrag<TS>(Ws, 16 * kb + r, ks, lane), os[ks], ss[kb]);
#pragma unroll
      for (int ks = 0; ks < AT<TT>::NK; ++ks) st[kb] = mma<TT>(lfrag<TT>(Wt, 16 * kb + r, ks, lane), ot[ks], st[kb]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ss[kb][e] *= a.scale_s;
        st[kb][e] *= a.scale_t;
      }
    }
  };
  // entry (kb, e) of a tile at t0: swept row t0 + 16 kb + 4 q + e
  float m_s = 0.f, m_t = 0.f;   // kl-div, SWAP = false: log-sum-exp of this lane's query row
  if (a.mode == 1 && !SWAP) {
    float ms = -INFINITY, mt = -INFINITY, ls = 0.f, lt = 0.f;
    for (int kt = 0; kt < nt; ++kt) {
      f32x4 ss[4], st[4];
      tiles(kt * 64, ss, st);
      float xs = -INFINITY, xt = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool v = kt * 64 + 16 * kb + 4 * q + e < klen;
          if (!v) { ss[kb][e] = -INFINITY; st[kb][e] = -INFINITY; }
          xs = fmaxf(xs, ss[kb][e]);
          xt = fmaxf(xt, st[kb][e]);
        }
      xs = fmaxf(xor_max(xs), ms);
      xt = fmaxf(xor_max(xt), mt);
      float es = 0.f, et = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          es += ss[kb][e] == -INFINITY ? 0.f : fast_exp2((ss[kb][e] - xs) * L2E);
          et += st[kb][e] == -INFINITY ? 0.f : fast_exp2((st[kb][e] - xt) * L2E);
        }
      es = xor_sum(es);
      et = xor_sum(et);
      ls = ls * (ms == -INFINITY ? 0.f : fast_exp2((ms - xs) * L2E)) + es;
      lt = lt * (mt == -INFINITY ? 0.f : fast_exp2((mt - xt) * L2E)) + et;
      ms = xs;
      mt = xt;
    }
    m_s = ms + __logf(ls);
    m_t = mt + __logf(lt);
    if (lse && q == 0 && own_in) {
      lse_s[own] = m_s;
      lse_t[own] = m_t;
    }
  }
  // ---- loss and gradient sweep
  float loss = 0.f;
  f32x4 dO[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dO[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_grad = a.d_qkv != nullptr;
  for (int kt = 0; kt < nt; ++kt) {
    f32x4 ss[4], st[4];
    tiles(kt * 64, ss, st);
    if (SWAP && a.mode == 1) {   // log-sum-exp of the swept (query) rows of this tile
      if (threadIdx.x < 128) {
        const int row = kt * 64 + (threadIdx.x & 63);
        const float* src = threadIdx.x < 64 ? lse_s : lse_t;
        lrow[threadIdx.x] = row < T ? src[row] : 0.f;
      }
      __syncthreads();
    }
    f32x4 gh[4], gl[4];   // G^T tile as head / remainder (values representable in TS), unscaled
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sw = kt * 64 + 16 * kb + 4 * q + e;       // swept row
        const int qi = SWAP ? sw : own, ki = SWAP ? own : sw;
        const bool v = qi < T && ki < klen;
        float g = 0.f;
        if (v) {
          if (a.mode == 0) {
            const float d = ss[kb][e] - st[kb][e];
            if (!SWAP) loss = fmaf(d, d, loss);
            g = d;
          } else {
            const float rs = SWAP ? lrow[16 * kb + 4 * q + e] : m_s, rt = SWAP ? lrow[64 + 16 * kb + 4 * q + e] : m_t;
            const float lps = ss[kb][e] - rs, lpt = st[kb][e] - rt;
            const float ps = fast_exp2(lps * L2E), pt = fast_exp2(lpt * L2E);
            if (!SWAP) loss = fmaf(pt, lpt - lps, loss);
            g = ps - pt;
          }
        }
        TS hi, lo;
        split16<TS>(g, hi, lo);
        gh[kb][e] = (float)hi;
        gl[kb][e] = (float)lo;
      }
    if (want_grad) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        mma_transposed<TS, NDT>(Ws, m2, gh[2 * m2], gh[2 * m2 + 1], dO, lane);
        mma_transposed<TS, NDT>(Ws, m2, gl[2 * m2], gl[2 * m2 + 1], dO, lane);
      }
    }
  }
  if (want_grad && own_in) {   // dOwn^T in accumulator layout: row d = 16 dt + 4 q + e, column = own row
    float* dst = a.d_qkv + ((long)b * T + own) * lds_ + own_off_s + h * dhs;
    const float sc = gsc * a.scale_s;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 16 * dt + 4 * q;
      if (d < dhs) {   // (head widths are multiples of 8: a group of four is inside or outside)
        float4 cur = *reinterpret_cast<float4*>(dst + d);
        cur.x = fmaf(dO[dt][0], sc, cur.x);
        cur.y = fmaf(dO[dt][1], sc, cur.y);
        cur.z = fmaf(dO[dt][2], sc, cur.z);
        cur.w = fmaf(dO[dt][3], sc, cur.w);
        *reinterpret_cast<float4*>(dst + d) = cur;
      }
    }
  }
  if (!SWAP) {
    red[threadIdx.x] = loss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0 && a.partials) a.partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
  }
}

// out[0] = sum(partials) / normaliser  (fp64)
__global__ void relation_final_kernel(const float* partials, int n, double* out, RelArgs a) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double den = (double)a.B * a.H * a.T;
    if (a.mode == 0) {
      den *= (double)a.T;
      if (a.masked)
        for (int bb = 0; bb < a.B; ++bb) {
          int kl = a.T;
          if (a.s_len) kl = min(kl, a.s_len[bb]);
          if (a.t_len) kl = min(kl, a.t_len[bb]);
          den -= (double)a.H * (a.T - kl) * (double)a.T;
        }
    }
    out[0] = sh[0] / den;
  }
}

}  // namespace

// scratch floats of fh_relation_kd: one loss partial per (utterance, head, 64-row block), then -- for the 16-bit kernels' kl-div form --
// the log-sum-exp of every student and teacher logit row ([2][B H Tp], Tp = T rounded up to 64)
static int rel_nparts(int B, int T, int H) { return B * H * ((T + RT - 1) / RT); }
static int rel_tp(int T) { return (T + RT - 1) / RT * RT; }
extern "C" int fh_relation_kd_nblk(int B, int T, int H) { return rel_nparts(B, T, H) + 2 * B * H * rel_tp(T); }

static int relation_launch(int dtype, int teacher_dtype, const RelArgs& a, hipStream_t st);

// relation tensor itself: out[(b h)][q][k] = scale * A_q . B_k  (masked keys -inf), or its backward into d_qkv
extern "C" int fh_relation_tensor(int dtype, const void* qkv, float* out, const float* grad_out, float* d_qkv, const int* key_len,
                                  int B, int T, int H, int D, int which, fh_stream_t stream) {
  if (B <= 0 || T <= 0 || H <= 0 || D % H || D / H > RMAXD || (which != 0 && which != 1)) return -601;
  if ((out != nullptr) == (grad_out != nullptr) || (grad_out && !d_qkv)) return -604;
  RelArgs a;
  a.s_qkv = qkv; a.t_qkv = nullptr; a.d_qkv = d_qkv; a.partials = nullptr; a.s_len = key_len; a.t_len = nullptr;
  a.B = B; a.T = T; a.H = H; a.Ds = D; a.Dt = D;
  a.a_off_s = a.a_off_t = which == 0 ? 0 : 2 * D;
  a.b_off_s = a.b_off_t = which == 0 ? D : 2 * D;
  a.scale_s = a.scale_t = 1.0f / sqrtf((float)(D / H));
  a.mode = out ? 2 : 3;
  a.out = out; a.gin = grad_out;
  a.masked = which == 0;
  a.weight = 1.f; a.gscale = 1.f;
  return relation_launch(dtype, dtype, a, (hipStream_t)stream);
}

extern "C" int fh_relation_kd(int dtype, int teacher_dtype, const void* s_qkv, const void* t_qkv, float* d_qkv, float* partials,
                              double* loss_out, const int* s_len, const int* t_len, int B, int T, int H, int Ds, int Dt, int which,
                              int mode, float weight, fh_stream_t stream) {
  if (B <= 0 || T <= 0 || H <= 0 || Ds % H || Dt % H || Ds / H > RMAXD || Dt / H > RMAXD) return -601;
  if ((which != 0 && which != 1) || (mode != 0 && mode != 1)) return -602;
  if (which == 1 && mode == 0) return -603;  // the value relation is a kl-div loss only (train.py:358-367)
  RelArgs a;
  a.s_qkv = s_qkv; a.t_qkv = t_qkv; a.d_qkv = d_qkv; a.partials = partials; a.s_len = s_len; a.t_len = t_len;
  a.B = B; a.T = T; a.H = H; a.Ds = Ds; a.Dt = Dt;
  a.a_off_s = which == 0 ? 0 : 2 * Ds;       // q | v
  a.b_off_s = which == 0 ? Ds : 2 * Ds;      // k | v
  a.a_off_t = which == 0 ? 0 : 2 * Dt;
  a.b_off_t = which == 0 ? Dt : 2 * Dt;
  a.scale_s = 1.0f / sqrtf((float)(Ds / H));
  a.scale_t = 1.0f / sqrtf((float)(Dt / H));
  a.mode = mode;
  a.masked = which == 0;
  a.weight = weight;
  a.gscale = 1.0f / ((float)B * H * T);
  a.out = nullptr; a.gin = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = relation_launch(dtype, teacher_dtype, a, st);
  if (rc) return rc;
  hipLaunchKernelGGL(relation_final_kernel, dim3(1), dim3(256), 0, st, (const float*)partials, rel_nparts(B, T, H), loss_out, a);
  FH_CHECK_LAUNCH();
  return 0;
}

template <typename TS, typename TT, int NDT>
static int relation_mfma_launch(const RelArgs& a, hipStream_t st) {
  const dim3 grid((a.T + RT - 1) / RT, a.H, a.B), block(256);
  const size_t smem = (size_t)2 * 64 * AT<TS>::RS + 128 * sizeof(float);
  const int Tp = rel_tp(a.T);
  float* lse = a.mode == 1 ? a.partials + rel_nparts(a.B, a.T, a.H) : nullptr;
  hipLaunchKernelGGL((relation_mfma_kernel<TS, TT, NDT, false>), grid, block, smem, st, a, lse, Tp);   // loss, row operand's gradient, (log-sum-exp rows)
  FH_CHECK_LAUNCH();
  if (a.d_qkv) {
    hipLaunchKernelGGL((relation_mfma_kernel<TS, TT, NDT, true>), grid, block, smem, st, a, lse, Tp);  // column operand's gradient
    FH_CHECK_LAUNCH();
  }
  return 0;
}

static int relation_launch(int dtype, int teacher_dtype, const RelArgs& a, hipStream_t st) {
  const dim3 grid((a.T + RT - 1) / RT, a.H, a.B), block(256);
  // 16-bit student and teacher projections, loss modes: the MFMA kernels (the fp32 parity mode and the relation-tensor modes of the
  // module API stay on the fp32 kernel)
  if (dtype != 0 && teacher_dtype != 0 && a.mode <= 1 && a.t_qkv && a.partials && (a.Ds / a.H) % 8 == 0 && (a.Dt / a.H) % 8 == 0) {
    const bool wide = a.Ds / a.H > 48;
#define FH_RM(TS, TT) return wide ? relation_mfma_launch<TS, TT, 4>(a, st) : relation_mfma_launch<TS, TT, 3>(a, st)
    if (dtype == 2) { if (teacher_dtype == 2) FH_RM(f16, f16); else FH_RM(f16, bf16); }
    else { if (teacher_dtype == 2) FH_RM(bf16, f16); else FH_RM(bf16, bf16); }
#undef FH_RM
  }
  const size_t smem = (size_t)5 * RT * RSTR * sizeof(float);
#define FH_REL(TS, TT)                                                                                                       \
  do {                                                                                                                       \
    static bool attr = false;                                                                                                \
    if (!attr) {                                                                                                             \
      hipFuncSetAttribute((const void*)relation_kd_kernel<TS, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);    \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((relation_kd_kernel<TS, TT>), grid, block, smem, st, a);                                              \
  } while (0)
#define FH_REL_T(TS)                                  \
  do {                                                \
    if (teacher_dtype == 0) FH_REL(TS, float);        \
    else if (teacher_dtype == 2) FH_REL(TS, f16);     \
    else FH_REL(TS, bf16);                            \
  } while (0)
  if (dtype == 0) FH_REL_T(float);
  else if (dtype == 2) FH_REL_T(f16);
  else FH_REL_T(bf16);
#undef FH_REL_T
#undef FH_REL
  FH_CHECK_LAUNCH();
  return 0;
}
