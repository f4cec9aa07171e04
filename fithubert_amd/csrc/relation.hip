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

extern "C" int fh_relation_kd_nblk(int B, int T, int H) { return B * H * ((T + RT - 1) / RT); }

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
  hipLaunchKernelGGL(relation_final_kernel, dim3(1), dim3(256), 0, st, (const float*)partials, fh_relation_kd_nblk(B, T, H), loss_out, a);
  FH_CHECK_LAUNCH();
  return 0;
}

static int relation_launch(int dtype, int teacher_dtype, const RelArgs& a, hipStream_t st) {
  const dim3 grid((a.T + RT - 1) / RT, a.H, a.B), block(256);
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
