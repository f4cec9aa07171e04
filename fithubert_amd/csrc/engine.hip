// Student engine: CustomStudentModel.forward (reference modules/model.py:420-552) and its backward as one native plan.
//
// Data layout in HBM: every activation is channels-last [B][T][C] (so a conv window is a contiguous k*Cin run and
// conv1d == GEMM with overlapping rows), all in one caller-provided workspace; the 12 transformer layers and the
// 12 projection heads keep their activations in [L][...] arrays with a uniform stride so that the weight-gradient
// GEMMs of all layers run as ONE grouped launch (fills 256 CUs without split-K).  Parameters live in one flat fp32
// buffer (q|k|v adjacent => fused QKV GEMM); gradients in a flat fp32 buffer of the same layout (one all-reduce).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gemm.h"
#include "kernels.h"
#include "common.cuh"

namespace {
// FH_SAVE_GELU_GRAD=0: the conv stack saves pre-activations and the backward evaluates GELU' (the round-1 form; A/B comparisons)
int save_gelu_grad() {
  static const int on = [] { const char* e = getenv("FH_SAVE_GELU_GRAD"); return (e && e[0] == '0') ? 0 : 1; }();
  return on;
}

#define RC(x)                 \
  do {                        \
    int rc__ = (x);           \
    if (rc__ != 0) return rc__; \
  } while (0)

inline long rup(long x, long a) { return (x + a - 1) / a * a; }

// ---- merged projection heads (LayerWiseProjHead = ConvTranspose1d(k = s = f) then Linear, module.py:649-661): the two affine
// maps compose, pred[(t, j)] = (Wlin Wup_j) x_t + (Wlin b_up + b_lin), so the engine runs ONE GEMM per head against weights merged
// once per optimizer step -- 38 % fewer head FLOPs, no upsampled intermediate -- and maps the merged weights' gradient back onto
// the reference's parameters with small fp32 GEMMs (chain rule).
// bm[l][p] = b_lin[l][p] + sum_co Wlin[l][p][co] * b_up[l][co]
__global__ void head_bias_merge_kernel(const float* __restrict__ params, long head0, long head_stride, long o_lin_w, long o_lin_b,
                                       long o_up_b, float* __restrict__ bm, int L, int P, int D) {
  // one WAVE per output (l, p): the row of Wlin is read coalesced (one thread per output walked it with a D-float stride: 68 us
  // for 4.4 M multiply-adds, 1 % of a B = 4 step)
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= L * P) return;
  const int l = i / P, pp = i - l * P;
  const float* hp = params + head0 + (long)l * head_stride;
  const float* w = hp + o_lin_w + (long)pp * D;
  float s = 0.f;
  for (int co = lane; co < D; co += 64) s = fmaf(w[co], hp[o_up_b + co], s);
  s = wave_sum(s);
  if (lane == 0) bm[i] = s + hp[o_lin_b + pp];
}
// dbl[l][p] = this pass's column sums of d_pred.  db_lin += dbl;  db_up[co] += sum_p Wlin[p][co] dbl[p];
// dWlin[p][co] += dbl[p] * b_up[co]   (the merged bias depends on Wlin too)
__global__ void head_bias_bwd_kernel(const float* __restrict__ params, float* __restrict__ grads, long head_base, long head_stride,
                                     long o_lin_w, long o_lin_b, long o_up_b, const float* __restrict__ dbl, int P, int D) {
  // grid (heads, ceil(D / 16)); a workgroup owns 16 columns co of one head: 16 x 16 threads = (row lane, column)
  __shared__ float part[16][17];
  const int l = blockIdx.x;
  const float* hp = params + head_base + (long)l * head_stride;
  float* hg = grads + head_base + (long)l * head_stride;
  const float* d = dbl + (long)l * P;
  if (blockIdx.y == 0)
    for (int pp = threadIdx.x; pp < P; pp += blockDim.x) hg[o_lin_b + pp] += d[pp];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int co = blockIdx.y * 16 + cl;
  float s = 0.f;
  if (co < D) {
    const float bu = hp[o_up_b + co];
    for (int pp = rl; pp < P; pp += 16) {
      const long i = o_lin_w + (long)pp * D + co;
      const float dv = d[pp];
      s = fmaf(hp[i], dv, s);
      hg[i] += dv * bu;
    }
  }
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && co < D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += part[r][cl];
    hg[o_up_b + co] += t;
  }
}

// bf16 TRANSPOSED copies of the four Linear weights of every transformer layer, one launch for all of them (grid z = layer x
// matrix): the data gradients  dX = dY . W  then read W^T with the reduction index contiguous -- both GEMM operands R_CONTIG, which is
// what the fastest main loops (8-phase, small-M) take; the R_STRIDED form read W through transposing LDS loads in the ring loops only.
struct LayerTArgs {
  long src_off[4], dst_off[4];  // element offsets inside a layer's parameter block / transposed block
  int rows[4], cols[4];         // of the source matrix [rows][cols]
  long src_stride, dst_stride, src0;
};
template <typename H>
__global__ void __launch_bounds__(256) layer_transpose_kernel(const float* __restrict__ params, H* __restrict__ dst, LayerTArgs a) {
  __shared__ float tile[32][33];
  const int l = blockIdx.z >> 2, m = blockIdx.z & 3;
  const int R = a.rows[m], C = a.cols[m];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (c0 >= C || r0 >= R) return;
  const float* src = params + a.src0 + (long)l * a.src_stride + a.src_off[m];
  H* out = dst + (long)l * a.dst_stride + a.dst_off[m];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) out[(long)c * R + r] = from_f<H>(tile[tx][ty + 8 * i]);
  }
}

struct PEntry {
  std::string name;
  long off;
  int ndim;
  long dims[4];
  long numel() const {
    long n = 1;
    for (int i = 0; i < ndim; ++i) n *= dims[i];
    return n;
  }
};

struct ParamLayout {
  std::vector<PEntry> list;
  long total = 0;
  long conv_w[FH_MAX_CONV];
  long gn_w, gn_b, ln_w, ln_b, post_w, post_b, pos_b, pos_g, pos_v, tr_w, tr_b;
  long layer0, layer_stride;
  long l_qkv_w, l_qkv_b, l_o_w, l_o_b, l_ln1_w, l_ln1_b, l_fc1_w, l_fc1_b, l_fc2_w, l_fc2_b, l_ln2_w, l_ln2_b;
  long enc_ln_w, enc_ln_b, ups_w, ups_b;
  long head0, head_stride, h_up_w, h_up_b, h_lin_w, h_lin_b;
  long sp_w0 = -1, sp_b0 = -1, sp_w2 = -1, sp_b2 = -1;  // DistilHuBERT-style head (n_tasks >= 2)
  long cnn_w = -1, cnn_b = -1;                           // cnn_proj_head.1 (model.py:305-310)

  long add(const std::string& name, std::initializer_list<long> dims) {
    PEntry e;
    e.name = name;
    e.off = total;
    e.ndim = (int)dims.size();
    int i = 0;
    for (long d : dims) e.dims[i++] = d;
    for (; i < 4; ++i) e.dims[i] = 1;
    list.push_back(e);
    total += rup(e.numel(), 8);
    return e.off;
  }

  explicit ParamLayout(const fh_student_config& c) {
    const long D = c.embed_dim, Fd = c.ffn_dim, P = c.pred_dim;
    long cin = 1;
    for (int i = 0; i < c.n_conv; ++i) {
      conv_w[i] = add("feature_extractor.conv_layers." + std::to_string(i) + ".0.weight", {c.conv_dim[i], cin, c.conv_k[i]});
      cin = c.conv_dim[i];
    }
    gn_w = add("feature_extractor.conv_layers.0.2.weight", {c.conv_dim[0]});
    gn_b = add("feature_extractor.conv_layers.0.2.bias", {c.conv_dim[0]});
    ln_w = add("layer_norm.weight", {cin});
    ln_b = add("layer_norm.bias", {cin});
    post_w = add("post_extract_proj.weight", {D, cin});
    post_b = add("post_extract_proj.bias", {D});
    pos_b = add("encoder.pos_conv.0.bias", {D});
    pos_g = add("encoder.pos_conv.0.weight_g", {1, 1, c.conv_pos});
    pos_v = add("encoder.pos_conv.0.weight_v", {D, D / c.conv_pos_groups, c.conv_pos});
    tr_w = tr_b = -1;
    if (c.tr_factor > 0) {  // the TR conv is module tr_layer_index of encoder.layers (module.py:247-248)
      tr_w = add("encoder.layers." + std::to_string(c.tr_index) + ".weight", {D, D, c.tr_factor});
      tr_b = add("encoder.layers." + std::to_string(c.tr_index) + ".bias", {D});
    }
    layer0 = total;
    for (int l = 0; l < c.layers; ++l) {
      const std::string p = "encoder.layers." + std::to_string(l + (c.tr_factor > 0 && l >= c.tr_index ? 1 : 0)) + ".";
      const long base = total;
      long o;
      o = add(p + "self_attn.q_proj.weight", {D, D}); if (!l) l_qkv_w = o - base;
      add(p + "self_attn.k_proj.weight", {D, D});
      add(p + "self_attn.v_proj.weight", {D, D});
      o = add(p + "self_attn.q_proj.bias", {D}); if (!l) l_qkv_b = o - base;
      add(p + "self_attn.k_proj.bias", {D});
      add(p + "self_attn.v_proj.bias", {D});
      o = add(p + "self_attn.out_proj.weight", {D, D}); if (!l) l_o_w = o - base;
      o = add(p + "self_attn.out_proj.bias", {D}); if (!l) l_o_b = o - base;
      o = add(p + "self_attn_layer_norm.weight", {D}); if (!l) l_ln1_w = o - base;
      o = add(p + "self_attn_layer_norm.bias", {D}); if (!l) l_ln1_b = o - base;
      o = add(p + "fc1.weight", {Fd, D}); if (!l) l_fc1_w = o - base;
      o = add(p + "fc1.bias", {Fd}); if (!l) l_fc1_b = o - base;
      o = add(p + "fc2.weight", {D, Fd}); if (!l) l_fc2_w = o - base;
      o = add(p + "fc2.bias", {D}); if (!l) l_fc2_b = o - base;
      o = add(p + "final_layer_norm.weight", {D}); if (!l) l_ln2_w = o - base;
      o = add(p + "final_layer_norm.bias", {D}); if (!l) l_ln2_b = o - base;
      if (!l) layer_stride = total - base;
    }
    enc_ln_w = add("encoder.layer_norm.weight", {D});
    enc_ln_b = add("encoder.layer_norm.bias", {D});
    ups_w = ups_b = -1;
    const long f = c.tr_factor > 0 ? c.tr_factor : 1;
    if (c.tr_factor > 0) {
      ups_w = add("upsampler.weight", {D, D, f});  // model.py:341-348: used by the DistilHuBERT-style head only (model.py:503-506)
      ups_b = add("upsampler.bias", {D});
    }
    if (c.cnn_head) {
      cnn_w = add("cnn_proj_head.1.weight", {P, D});
      cnn_b = add("cnn_proj_head.1.bias", {P});
    }
    head0 = total;
    if (c.n_tasks > 0) {
      head_stride = 0; h_up_w = h_up_b = h_lin_w = h_lin_b = -1;
      const long I = c.inter_dim, Nt = c.n_tasks;
      sp_w0 = add("proj_head.0.weight", {I * Nt, D});
      sp_b0 = add("proj_head.0.bias", {I * Nt});
      if (Nt > 1) {
        sp_w2 = add("proj_head.2.weight", {Nt, I, P});   // SplitLinear: [task][Din][Dout]
        sp_b2 = add("proj_head.2.bias", {1, 1, Nt, P});
      } else {                                          // in_split == 1: a plain nn.Linear (module.py:604-605)
        sp_w2 = add("proj_head.2.layer.weight", {P, I});
        sp_b2 = add("proj_head.2.layer.bias", {P});
      }
      return;
    }
    for (int l = 0; l < c.layers; ++l) {
      const std::string p = "proj_head." + std::to_string(l) + ".";
      const long base = total;
      long o;
      if (c.tr_factor > 0) {
        o = add(p + "upsampler.weight", {D, D, f}); if (!l) h_up_w = o - base;
        o = add(p + "upsampler.bias", {D}); if (!l) h_up_b = o - base;
      } else if (!l) {
        h_up_w = h_up_b = -1;
      }
      o = add(p + "lin_proj.weight", {P, D}); if (!l) h_lin_w = o - base;
      o = add(p + "lin_proj.bias", {P}); if (!l) h_lin_b = o - base;
      if (!l) head_stride = total - base;
    }
  }
};

struct WBuf {
  std::string name;
  long off;  // bytes
  int ndim;
  long dims[4];
  int esize;
};

struct Shapes {
  int B, S, n;
  int T[FH_MAX_CONV];  // frames after conv layer i
  int Tp, T2, f;       // frames after the frontend, after time reduction, reduction factor (1 if none)
  long Mp, M2;
  // tr_layer_index > 0: transformer layers [0, idx) run on the Tp frames, [idx, L) on the T2 reduced ones.  Every per-layer array
  // keeps ONE row stride Ms (= Mp then; M2 in the shipped idx == 0 layout, which is unchanged), a layer uses the first M(l) rows of
  // its slot; `xo` gets one more slot: 0 = encoder LayerNorm output, l + 1 = output of layer l < idx, idx + 1 = TR output,
  // l + 2 = output of layer l >= idx.
  int idx, Ts;
  long Ms;
  int Tl(int l) const { return l < idx ? Tp : T2; }
  long Ml(int l) const { return l < idx ? Mp : M2; }
  int slot_in(int l) const { return l + (idx > 0 && l >= idx ? 1 : 0); }
  int slots(int L) const { return L + 1 + (idx > 0 ? 1 : 0); }
};

inline int make_shapes(const fh_student_config& c, int B, int S, Shapes& s) {
  s.B = B; s.S = S; s.n = c.n_conv;
  long t = S;
  for (int i = 0; i < c.n_conv; ++i) {
    if (t < c.conv_k[i]) return -801;
    t = (t - c.conv_k[i]) / c.conv_stride[i] + 1;
    s.T[i] = (int)t;
  }
  s.Tp = (int)t;
  s.f = c.tr_factor > 0 ? c.tr_factor : 1;
  s.T2 = c.tr_factor > 0 ? (s.Tp - c.tr_factor) / c.tr_factor + 1 : s.Tp;
  if (s.T2 < 1) return -801;
  s.Mp = (long)B * s.Tp;
  s.M2 = (long)B * s.T2;
  s.idx = c.tr_factor > 0 ? c.tr_index : 0;
  s.Ts = s.idx > 0 ? s.Tp : s.T2;
  s.Ms = (long)B * s.Ts;
  return 0;
}

// compute-dtype weight buffer
struct WcLayout {
  long cast_off = 0;      // bf16 copy of the whole flat parameter buffer (bf16 mode only)
  long conv_wf[FH_MAX_CONV];
  long conv_wt[FH_MAX_CONV];   // [k][Cin][Cout]: the data gradients' weight operand with the reduction index (Cout) contiguous (round 5)
  long tr_wf = -1, head_wup = -1, head_wup_stride = 0, pos_wf, pos_wb, pos_norm;
  long head_wm = -1, head_bm = -1, head_wupf = -1;  // merged head weights [L][f*P][D] (compute dtype), merged bias [L][P] fp32,
                                                    // fp32 copy of the permuted upsampler weights [L][f][D][D]
  long layT = -1, layT_stride = 0, t_qkv = 0, t_o = 0, t_fc1 = 0, t_fc2 = 0;  // bf16 transposed layer weights (bytes / elements)
  long total = 0;
  WcLayout(const fh_student_config& c, const ParamLayout& pl) {
    const int es = c.dtype == 0 ? 4 : 2;
    long o = 0;
    if (c.dtype != 0) { cast_off = 0; o = rup(pl.total * 2, 256); }
    long cin = 1;
    for (int i = 0; i < c.n_conv; ++i) {
      conv_wf[i] = o;
      if (i > 0) o += rup((long)c.conv_dim[i] * cin * c.conv_k[i] * es, 256);
      conv_wt[i] = o;
      if (i > 0) o += rup((long)c.conv_dim[i] * cin * c.conv_k[i] * es, 256);
      cin = c.conv_dim[i];
    }
    const long D = c.embed_dim;
    if (c.tr_factor > 0) {
      tr_wf = o; o += rup(D * D * c.tr_factor * es, 256);
      head_wup = o; head_wup_stride = D * D * c.tr_factor;  // elements
      o += rup(head_wup_stride * c.layers * es, 256);
      if (c.n_tasks == 0) {
        head_wm = o; o += rup((long)c.layers * c.tr_factor * c.pred_dim * D * es, 256);
        head_bm = o; o += rup((long)c.layers * c.pred_dim * 4, 256);
        head_wupf = head_wup;   // (round 1 kept an fp32 copy for the chain rule of the merged heads; it runs in the compute dtype now)
      }
    }
    const long cp = fh_posconv_cp(c.embed_dim / c.conv_pos_groups);
    const long pw = (long)c.conv_pos_groups * c.conv_pos * cp * cp;
    pos_wf = o; o += rup(pw * es, 256);
    pos_wb = o; o += rup(pw * es, 256);
    pos_norm = o; o += rup(3L * c.conv_pos * 4, 256);
    if (c.dtype != 0) {  // [L] x { Wqkv^T [D][3D] | Wo^T [D][D] | W1^T [D][Fd] | W2^T [Fd][D] }
      const long Fd = c.ffn_dim;
      t_qkv = 0; t_o = 3 * D * D; t_fc1 = t_o + D * D; t_fc2 = t_fc1 + D * Fd;
      layT_stride = t_fc2 + Fd * D;
      layT = o; o += rup(layT_stride * c.layers * 2, 256);
    }
    total = o;
  }
};

struct WsLayout {
  std::vector<WBuf> bufs;
  long total = 0;
  long add(const std::string& name, std::initializer_list<long> dims, int esize) {
    WBuf b;
    b.name = name; b.off = total; b.ndim = (int)dims.size(); b.esize = esize;
    int i = 0; long n = 1;
    for (long d : dims) { b.dims[i++] = d; n *= d; }
    for (; i < 4; ++i) b.dims[i] = 1;
    bufs.push_back(b);
    total += rup(n * esize, 256);
    return b.off;
  }
  long find(const char* name) const {
    for (auto& b : bufs) if (b.name == name) return b.off;
    return -1;
  }
};

// conv 0's backward inside layer 1's data-gradient GEMM (fh_gemm_epilogue.c0_*): k = 1 layer 1 over 128 conv-0 channels, 10 taps, 16-bit;
// FH_CONV0_FUSED_BWD=0 restores the two kernels (A/B)
inline bool conv0_bwd_fusable(const fh_student_config& c) {
  static const int on = [] { const char* e = getenv("FH_CONV0_FUSED_BWD"); return (e && e[0] == '0') ? 0 : 1; }();
  return on && c.dtype != 0 && c.n_conv >= 2 && c.conv_k[1] == 1 && c.conv_stride[1] == 1 && c.conv_dim[0] == 128 && c.conv_k[0] == 10 /* the epilogue's and the combine's sum layouts agree at 10 taps */ &&
         256 * c.conv_stride[0] + 16 <= 2048 && c.feature_grad_mult > 0.f;
}

struct Plan {
  fh_student_config c;
  fh_student_call k;
  ParamLayout pl;
  WcLayout wl;
  Shapes sh;
  WsLayout ws;
  int es;
  int ok;
  // workspace offsets
  long y[FH_MAX_CONV], z[FH_MAX_CONV], dz[FH_MAX_CONV];
  long c0_mean, c0_rstd, c0_scratch;
  bool c0_fused = false;
  long ln0, ln0_mean, ln0_rstd, features, x0, zc, psum, ln1_mean, ln1_rstd, x1;
  long xo, qkv, attn, lse, s1, st1_mean, st1_rstd, xa, z1, hbuf, f2, s2, st2_mean, st2_rstd;
  long u = -1, pred, hz = -1, hh = -1, dhz = -1, gfeat = -1, cnn_pred = -1, dgf = -1, dwm = -1, dwm16 = -1, dbl = -1;
  // backward
  long du, ghead, dxin, ds2, df2, dz1, dx1b, ds1, dao, da, dqkv, delta;
  long dx1tr, dpsum, dzc, dx0, dfeat, dln0, dylast, pos_dws, attn_scr = -1;

  Plan(const fh_student_config& cfg, const fh_student_call& call) : c(cfg), k(call), pl(cfg), wl(cfg, pl) {
    es = c.dtype == 0 ? 4 : 2;
    ok = make_shapes(c, k.B, k.S, sh);
    if (ok != 0) return;
    const long B = k.B, D = c.embed_dim, Fd = c.ffn_dim, P = c.pred_dim, L = c.layers, H = c.heads;
    const long Mp = sh.Mp, M2 = sh.M2, Ms = sh.Ms;
    const int Ts = sh.Ts;
    const bool tr = k.training != 0;
    for (int i = 0; i < c.n_conv; ++i) {
      y[i] = ws.add("y" + std::to_string(i), {B, sh.T[i], c.conv_dim[i]}, es);
      z[i] = dz[i] = -1;
      if (tr && i > 0) z[i] = ws.add("z" + std::to_string(i), {B, sh.T[i], c.conv_dim[i]}, es);
    }
    c0_mean = ws.add("c0_mean", {B, c.conv_dim[0]}, 4);
    c0_rstd = ws.add("c0_rstd", {B, c.conv_dim[0]}, 4);
    c0_fused = tr && conv0_bwd_fusable(c);   // evaluated ONCE: the scratch size, the zeroing and the data-gradient launch all read this flag
    c0_scratch = ws.add("c0_scratch", {c0_fused ? fh_conv0_fused_scratch_n((int)B, sh.T[0], c.conv_dim[0])
                                                : fh_conv0_scratch_n((int)B, sh.T[0], c.conv_dim[0])}, 4);
    const long E = c.conv_dim[c.n_conv - 1];
    ln0 = ws.add("ln0", {Mp, E}, es);
    ln0_mean = ws.add("ln0_mean", {Mp}, 4);
    ln0_rstd = ws.add("ln0_rstd", {Mp}, 4);
    features = ws.add("features", {B, sh.Tp, D}, es);
    x0 = ws.add("x0", {B, sh.Tp, D}, es);
    zc = ws.add("zc", {B, sh.Tp, D}, es);
    psum = ws.add("psum", {B, sh.Tp, D}, es);
    ln1_mean = ws.add("ln1_mean", {Mp}, 4);
    ln1_rstd = ws.add("ln1_rstd", {Mp}, 4);
    x1 = ws.add("x1", {B, sh.Tp, D}, es);
    // tr_layer_index == 0 (or no TR layer): slot 0 = input of layer 0 (time-reduced), slot l + 1 = output of layer l.
    // tr_layer_index > 0: slot stride = the T' frames; see Shapes
    xo = ws.add("xo", {sh.slots((int)L), B, Ts, D}, es);
    qkv = ws.add("qkv", {L, Ms, 3 * D}, es);
    attn = ws.add("attn", {L, Ms, D}, es);
    lse = ws.add("lse", {L, B * H, Ts}, 4);
    s1 = ws.add("s1", {L, Ms, D}, es);
    st1_mean = ws.add("st1_mean", {L, Ms}, 4);
    st1_rstd = ws.add("st1_rstd", {L, Ms}, 4);
    xa = ws.add("xa", {L, Ms, D}, es);
    z1 = ws.add("z1", {L, Ms, Fd}, es);
    hbuf = ws.add("h", {L, Ms, Fd}, es);
    f2 = ws.add("f2", {L, B, Ts, D}, es);
    s2 = ws.add("s2", {L, Ms, D}, es);
    st2_mean = ws.add("st2_mean", {L, Ms}, 4);
    st2_rstd = ws.add("st2_rstd", {L, Ms}, 4);
    const long Lh = c.n_tasks > 0 ? c.n_tasks : (k.heads_on ? L : 1);
    const long Mh = c.n_tasks > 0 ? M2 * sh.f : M2;   // rows the DistilHuBERT-style head runs on (behind the shared upsampler: model.py:503-506)
    if (c.n_tasks > 0) {
      const long IN = (long)c.inter_dim * c.n_tasks;
      if (c.tr_factor > 0) u = ws.add("u", {B, (long)sh.T2 * sh.f, D}, es);
      if (tr) hz = ws.add("hz", {Mh, IN}, es);
      hh = ws.add("hh", {Mh, IN}, es);
      if (tr) dhz = ws.add("dhz", {Mh, IN}, es);
    }
    // layer-wise heads with heads on and tr_layer_index > 0: the heads of the layers in front of the TR conv see T' frames, so a head's
    // slot holds Ts * f frames and a layer fills the first T(l) * f of them
    pred = ws.add("pred", {Lh, B, (c.n_tasks == 0 && k.heads_on ? (long)Ts : (long)sh.T2) * sh.f, P}, es);
    if (c.cnn_head) {
      gfeat = ws.add("gfeat", {B, sh.Tp, D}, es);
      cnn_pred = ws.add("cnn_pred", {B, sh.Tp, P}, es);
      if (tr) dgf = ws.add("dgf", {B, sh.Tp, D}, es);
    }
    if (tr) {
      du = -1;
      if (c.n_tasks > 0 && c.tr_factor > 0) du = ws.add("du", {B, (long)sh.T2 * sh.f, D}, es);
      if (c.n_tasks == 0 && c.tr_factor > 0) {  // gradient of the merged head weights / this pass's d_pred column sums
        dwm = ws.add("dwm", {Lh, (long)sh.f * P, D}, 4);
        if (c.dtype != 0) dwm16 = ws.add("dwm16", {Lh, (long)sh.f * P, D}, 2);
        dbl = ws.add("dbl", {Lh, P}, 4);
      }
      ghead = ws.add("ghead", {L, Ms, D}, es);
      dxin = ws.add("dxin", {2, Ms, D}, es);
      ds2 = ws.add("ds2", {Ms, D}, es);
      df2 = ws.add("df2", {L, Ms, D}, es);
      dz1 = ws.add("dz1", {L, Ms, Fd}, es);
      dx1b = ws.add("dx1b", {Ms, D}, es);
      ds1 = ws.add("ds1", {Ms, D}, es);
      dao = ws.add("dao", {L, Ms, D}, es);
      da = ws.add("da", {Ms, D}, es);
      dqkv = ws.add("dqkv", {L, Ms, 3 * D}, es);
      delta = ws.add("delta", {B * H, Ts}, 4);
      {  // fp32 dQ partials of the any-length fused attention backward (16-bit modes, T > 256)
        long n = fh_attention_bwd_scratch_floats(c.dtype, (int)B, sh.T2, (int)H, (int)(D / H));
        if (sh.idx > 0) { const long n2 = fh_attention_bwd_scratch_floats(c.dtype, (int)B, sh.Tp, (int)H, (int)(D / H)); if (n2 > n) n = n2; }
        if (n > 0) attn_scr = ws.add("attn_scr", {n}, 4);
      }
      dx1tr = ws.add("dx1tr", {B, sh.Tp, D}, es);
      dpsum = ws.add("dpsum", {B, sh.Tp, D}, es);
      dzc = ws.add("dzc", {B, sh.Tp, D}, es);
      dx0 = ws.add("dx0", {B, sh.Tp, D}, es);
      dfeat = ws.add("dfeat", {B, sh.Tp, D}, es);
      dln0 = ws.add("dln0", {Mp, E}, es);
      dylast = ws.add("dylast", {Mp, E}, es);
      const long cp = fh_posconv_cp(c.embed_dim / c.conv_pos_groups);
      pos_dws = ws.add("pos_dws", {(long)c.conv_pos_groups * c.conv_pos * cp * cp}, 4);
      for (int i = 0; i < c.n_conv; ++i) dz[i] = ws.add("dz" + std::to_string(i), {B, sh.T[i], c.conv_dim[i]}, es);
    }
  }
};

inline int validate(const fh_student_config& c) {
  if (c.dtype < 0 || c.dtype > 2) return -810;
  if (c.n_conv < 1 || c.n_conv > FH_MAX_CONV) return -811;
  if (c.conv_k[0] > 16) return -812;
  const int vec = c.dtype == 0 ? 4 : 8;
  for (int i = 0; i < c.n_conv; ++i)
    if (c.conv_dim[i] % 64) return -813;
  if (c.embed_dim % c.heads || c.embed_dim % c.conv_pos_groups || c.embed_dim % vec || c.ffn_dim % vec || c.pred_dim % 8)
    return -814;
  if (c.tr_factor != 0 && (c.tr_factor < 2 || c.tr_factor > 8)) return -815;
  if (c.tr_index < 0 || c.tr_index > c.layers || (c.tr_factor == 0 && c.tr_index != 0)) return -819;
  if (!fh_posconv_cp(c.embed_dim / c.conv_pos_groups) || (c.embed_dim / c.conv_pos_groups) % 2) return -818;
  if (c.n_tasks < 0 || (c.n_tasks > 0 && (c.inter_dim <= 0 || c.inter_dim % vec))) return -817;
  for (int i = 1; i < c.n_conv; ++i) {
    const int kk = c.conv_k[i], s = c.conv_stride[i];
    if (!((kk == s) || (kk == 1 && s == 1) || (kk == 3 && s == 2))) return -816;
  }
  return 0;
}

struct Ctx {
  const Plan& p;
  const float* params;
  const unsigned char* wc;
  unsigned char* ws;
  hipStream_t st;
  int dt;
  template <typename T = void> T* W(long off) const { return reinterpret_cast<T*>(ws + off); }
  // natural-layout weight in compute dtype
  const void* nat(long poff) const {
    return p.c.dtype == 0 ? (const void*)(params + poff) : (const void*)(wc + p.wl.cast_off + poff * 2);
  }
  const void* rel(long boff) const { return wc + boff; }
  // transposed bf16 weight `toff` of layer l (bf16 mode), or null
  const void* layT(int l, long toff) const {
    return p.wl.layT >= 0 ? (const void*)(wc + p.wl.layT + ((long)l * p.wl.layT_stride + toff) * 2) : nullptr;
  }
  fh_drop drop(float prob, uint32_t site) const {
    fh_drop d;
    d.p = p.k.training ? prob : 0.f;
    d.seed = p.k.seed;
    d.site = site;
    return d;
  }
  unsigned char* at(long off, long elems) const { return ws + off + elems * p.es; }
};

enum { SITE_INPUT = 1, SITE_ENC = 2, SITE_LAYER0 = 16 };

inline int splitk_for(long tiles, long red) {
  long sk = (512 + tiles - 1) / tiles;
  const long maxsk = red / 512 > 0 ? red / 512 : 1;
  if (sk > maxsk) sk = maxsk;
  if (sk < 1) sk = 1;
  if (sk > 256) sk = 256;
  return (int)sk;
}

// Side stream for weight / bias gradients whose only consumer is the optimizer: they read per-layer buffers that the rest of
// the backward never rewrites and accumulate into the flat gradient buffer, so they can run beside the data-gradient chain --
// whose short, single-wave GEMMs and latency-bound LayerNorm / attention kernels leave CUs idle -- instead of in front of it.
// fork(): the side stream waits for everything queued on the main stream so far; join(): the main stream waits for the side
// stream (end of backward: the workspace goes back to the allocator only after that).  Off while the GEMM tuner is timing
// launches, and with FH_WGRAD_STREAM=0.
struct SideStream {
  hipStream_t s = nullptr, own = nullptr;
  hipEvent_t ev[32];
  int next = 0, state = -1;  // -1 unknown, 0 off, 1 on
  bool used = false;
  // fh_student_side_stream: the caller's stream (e.g. one restricted to a share of the CUs) in place of the library's own
  void replace(hipStream_t caller) {
    on();
    if (!own) own = s;
    s = caller ? caller : own;
  }
  bool on() {
    if (state < 0) {
      const char* e = getenv("FH_WGRAD_STREAM");
      state = (e && e[0] == '0') ? 0 : 1;
      // lowest priority: the data-gradient chain (main stream) takes the CUs it can use, the weight gradients fill the rest
      int lo = 0, hi = 0;
      if (state && (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                    hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess))
        state = 0;
      for (int i = 0; state && i < 32; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) state = 0;
    }
    return state == 1 && !fh_gemm_tuning();
  }
  hipStream_t fork(hipStream_t main) {
    if (!on()) return main;
    hipEvent_t e = ev[next++ & 31];
    if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(s, e, 0) != hipSuccess) return main;
    used = true;
    return s;
  }
  int join(hipStream_t main) {
    if (!used) return 0;
    used = false;
    hipEvent_t e = ev[next++ & 31];
    if (hipEventRecord(e, s) != hipSuccess) return -830;
    return hipStreamWaitEvent(main, e, 0) == hipSuccess ? 0 : -830;
  }
};
SideStream g_wside;

// Gradient-segment notification (fh_student_backward_notify): when a contiguous range of the flat gradient buffer is final, the
// caller's communication stream is made to wait for the kernels that write it -- the main stream's position and, when the side
// stream carried some of them, the side stream's -- and the caller is told the range.
struct Notifier {
  const fh_grad_notify* n = nullptr;
  hipStream_t main = nullptr;
  int seg = 0;
  bool heads_done = false, layers_done = false;
  hipEvent_t ev[16];
  int next = 0, state = -1;
  bool on() {
    if (!n || !n->ready) return false;
    if (state < 0) {
      state = 1;
      for (int i = 0; i < 16; ++i)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) state = 0;
    }
    return state == 1;
  }
  int wait_on(hipStream_t producer) {
    hipStream_t comm = (hipStream_t)n->comm_stream;
    if (comm == producer) return 0;
    hipEvent_t e = ev[next++ & 15];
    if (hipEventRecord(e, producer) != hipSuccess || hipStreamWaitEvent(comm, e, 0) != hipSuccess) return -831;
    return 0;
  }
  int fire(long first, long count, bool with_side) {
    if (!on() || count <= 0) return 0;
    RC(wait_on(main));
    if (with_side && g_wside.used) RC(wait_on(g_wside.s));
    n->ready(n->user, seg, first, count);
    return 0;
  }
};
Notifier g_notify;

// weight-gradient GEMM: dW[n][kk] += sum_m dC[m][n] * A(m)[kk]   (atomic accumulate into the fp32 grad buffer)
inline int wgrad(const Ctx& x, const void* dC, long ld_dc, const GemmOperand& A, float* dW, int N, int Kout, long Mred,
                 int groups, long dc_g, long dw_g, int jdiv = 0, long sj1 = 0, long sj0 = 0, int idiv = 0, long si1 = 0,
                 long si0 = 0, long ldc = -1, hipStream_t st = nullptr) {
  GemmParams g = fh_gemm_params();
  g.X = fh_operand(dC, FH_R_STRIDED, ld_dc);
  g.X.gstride = dc_g;
  g.Y = A;
  g.C = dW;
  g.M = N; g.N = Kout; g.Kseg = (int)Mred;
  g.ldc = ldc >= 0 ? ldc : Kout;
  g.c_gstride = dw_g;
  g.groups = groups;
  g.ep.atomic = 1;
  g.ep.jdiv = jdiv; g.ep.sj1 = sj1; g.ep.sj0 = sj0;
  g.ep.idiv = idiv; g.ep.si1 = si1; g.ep.si0 = si0;
  const long tiles = (long)((N + 127) / 128) * ((Kout + 127) / 128) * groups;
  g.splitk = splitk_for(tiles, Mred);
  return fh_gemm_launch(g, x.dt, st ? st : x.st);
}

int prepare(const Plan& p, const float* params, unsigned char* wc, hipStream_t st) {
  const fh_student_config& c = p.c;
  const int dt = c.dtype;
  // every cast / permute of master weights into a kernel layout is ONE launch (weight_layout_kernel; round 5: they were ~22 tiny
  // launches that queued one by one behind the teacher's persistent GEMMs)
  fh_layout_jobs jobs;
  jobs.n = 0;
  if (dt != 0) jobs.add(0, p.wl.cast_off, 1, 1, (int)p.pl.total, 0, 1, 2);
  long cin = c.conv_dim[0];
  for (int i = 1; i < c.n_conv; ++i) {
    // [Cout][Cin][k] -> [Cout][k][Cin]
    jobs.add(p.pl.conv_w[i], p.wl.conv_wf[i], c.conv_dim[i], (int)cin, c.conv_k[i], 0, 2, 1);
    // [Cout][Cin][k] -> [k][Cin][Cout]: dX = dZ . W reads W with Cout contiguous -- both operands of the data-gradient GEMMs R_CONTIG,
    // which the 8-phase loops take (the R_STRIDED form went through transposing LDS reads in the ring loops: layer 2's data gradient
    // 1.76 -> 1.63 ms at B = 128, tools/scratch/dgrad_probe.py)
    jobs.add(p.pl.conv_w[i], p.wl.conv_wt[i], c.conv_dim[i], (int)cin, c.conv_k[i], 2, 1, 0);
    cin = c.conv_dim[i];
  }
  const int D = c.embed_dim;
  if (c.tr_factor > 0) {
    jobs.add(p.pl.tr_w, p.wl.tr_wf, D, D, c.tr_factor, 0, 2, 1);
    // ConvTranspose1d weight [Cin][Cout][k] -> [k][Cout][Cin]: all layer-wise heads as one grouped job (the bf16 mode's chain rule runs on
    // these bf16 copies too; the fp32 mode's copy IS the fp32 one), or the shared upsampler in front of the DistilHuBERT-style head
    // (model.py:341-348)
    if (c.n_tasks == 0)
      jobs.add(p.pl.head0 + p.pl.h_up_w, p.wl.head_wup, D, D, c.tr_factor, 2, 1, 0, c.layers, p.pl.head_stride, p.wl.head_wup_stride);
    else
      jobs.add(p.pl.ups_w, p.wl.head_wup, D, D, c.tr_factor, 2, 1, 0);
  }
  RC(fh_layout_jobs_launch(dt, params, wc, jobs, st));
  if (c.tr_factor > 0) {
    const int es = dt == 0 ? 4 : 2;
    if (c.n_tasks == 0) {
      // merged weights Wm[l][j][p][ci] = sum_co Wlin[l][p][co] * Wup[l][ci][co][j]
      const int P = c.pred_dim, f = c.tr_factor;
      const unsigned char* nat = dt == 0 ? (const unsigned char*)params : wc + p.wl.cast_off;
      for (int j = 0; j < f; ++j) {
        GemmParams g = fh_gemm_params();
        g.X = fh_operand(nat + (p.pl.head0 + p.pl.h_lin_w) * es, FH_R_CONTIG, D);
        g.X.gstride = p.pl.head_stride;
        g.Y = fh_operand(wc + p.wl.head_wup + (long)j * D * D * es, FH_R_STRIDED, D);
        g.Y.gstride = p.wl.head_wup_stride;
        g.C = wc + p.wl.head_wm + (long)j * P * D * es;
        g.M = P; g.N = D; g.Kseg = D; g.ldc = D;
        g.c_gstride = (long)f * P * D;
        g.groups = c.layers;
        RC(fh_gemm_launch(g, dt, st));
      }
      hipLaunchKernelGGL(head_bias_merge_kernel, dim3((c.layers * P + 3) / 4), dim3(256), 0, st, params, p.pl.head0,
                         p.pl.head_stride, p.pl.h_lin_w, p.pl.h_lin_b, p.pl.h_up_b, (float*)(wc + p.wl.head_bm), c.layers, P, D);
      FH_CHECK_LAUNCH();
    }
  }
  RC(fh_posconv_prep_launch(dt, params + p.pl.pos_g, params + p.pl.pos_v, wc + p.wl.pos_wf, wc + p.wl.pos_wb,
                            (float*)(wc + p.wl.pos_norm), D, c.conv_pos_groups, c.conv_pos, st));
  if (p.wl.layT >= 0) {
    const int Fd = c.ffn_dim;
    LayerTArgs a;
    a.src0 = p.pl.layer0; a.src_stride = p.pl.layer_stride; a.dst_stride = p.wl.layT_stride;
    a.src_off[0] = p.pl.l_qkv_w; a.dst_off[0] = p.wl.t_qkv; a.rows[0] = 3 * D; a.cols[0] = D;
    a.src_off[1] = p.pl.l_o_w;   a.dst_off[1] = p.wl.t_o;   a.rows[1] = D;     a.cols[1] = D;
    a.src_off[2] = p.pl.l_fc1_w; a.dst_off[2] = p.wl.t_fc1; a.rows[2] = Fd;    a.cols[2] = D;
    a.src_off[3] = p.pl.l_fc2_w; a.dst_off[3] = p.wl.t_fc2; a.rows[3] = D;     a.cols[3] = Fd;
    const int maxc = D > Fd ? D : Fd, maxr = 3 * D > Fd ? 3 * D : Fd;
    if (dt == 2)
      hipLaunchKernelGGL(layer_transpose_kernel<f16>, dim3((maxc + 31) / 32, (maxr + 31) / 32, c.layers * 4), dim3(256), 0, st, params,
                         (f16*)(wc + p.wl.layT), a);
    else
      hipLaunchKernelGGL(layer_transpose_kernel<bf16>, dim3((maxc + 31) / 32, (maxr + 31) / 32, c.layers * 4), dim3(256), 0, st, params,
                         (bf16*)(wc + p.wl.layT), a);
    FH_CHECK_LAUNCH();
  }
  return 0;
}

int forward(const Ctx& x, const float* wave, const int* lens, const fh_student_kd* kd = nullptr) {
  const Plan& p = x.p;
  const fh_student_config& c = p.c;
  const Shapes& s = p.sh;
  const int dt = x.dt, B = s.B, D = c.embed_dim, Fd = c.ffn_dim, P = c.pred_dim, L = c.layers, H = c.heads;
  const bool tr = p.k.training != 0;
  const int* len_p = lens;
  const int* len_2 = lens ? lens + B : nullptr;
  const float* prm = x.params;

  // ---- conv frontend (modules/module.py:94-102)
  RC(fh_conv0_fwd_launch(dt, wave, prm + p.pl.conv_w[0], prm + p.pl.gn_w, prm + p.pl.gn_b, x.W(p.y[0]), x.W<float>(p.c0_mean),
                         x.W<float>(p.c0_rstd), x.W<float2>(p.c0_scratch), B, s.S, c.conv_dim[0], c.conv_k[0], c.conv_stride[0],
                         1e-5f, x.st));
  for (int i = 1; i < c.n_conv; ++i) {
    const int cin = c.conv_dim[i - 1], kk = c.conv_k[i], sd = c.conv_stride[i];
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(x.W(p.y[i - 1]), FH_R_CONTIG, (long)sd * cin);
    g.X.rpb = s.T[i]; g.X.bstride = (long)s.T[i - 1] * cin;
    g.Y = fh_operand(x.rel(p.wl.conv_wf[i]), FH_R_CONTIG, (long)kk * cin);
    g.C = x.W(p.y[i]);
    g.M = B * s.T[i]; g.N = c.conv_dim[i]; g.Kseg = kk * cin; g.ldc = c.conv_dim[i];
    g.ep.act = FH_ACT_GELU;
    if (tr) {  // z[i] receives GELU'(pre-activation): what the backward multiplies with (fh_gemm_epilogue.aux_factor)
      g.ep.preact = x.W(p.z[i]);
      g.ep.aux_factor = save_gelu_grad();
    }
    RC(fh_gemm_launch(g, dt, x.st));
  }
  // ---- LayerNorm + post_extract_proj (+ dropout_input, zero padded frames)   model.py:446-489, module.py:273-274
  const int E = c.conv_dim[c.n_conv - 1];
  RC(fh_layernorm_fwd_launch(dt, x.W(p.y[c.n_conv - 1]), prm + p.pl.ln_w, prm + p.pl.ln_b, x.W(p.ln0), x.W<float>(p.ln0_mean),
                             x.W<float>(p.ln0_rstd), (int)s.Mp, E, 1e-5f, x.drop(0.f, 0), x.st));
  {
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(x.W(p.ln0), FH_R_CONTIG, E);
    g.Y = fh_operand(x.nat(p.pl.post_w), FH_R_CONTIG, E);
    g.M = (int)s.Mp; g.N = D; g.Kseg = E; g.ldc = D;
    g.c_rpb = s.Tp; g.c_bstride = (long)s.Tp * D;
    g.ep.bias = prm + p.pl.post_b;
    g.ep.row_len = len_p;
    const bool aliased = !tr || c.dropout_input == 0.f;  // reference aliasing of results["features"], see oracle
    if (aliased) {
      g.C = x.W(p.features);
      RC(fh_gemm_launch(g, dt, x.st));
      RC((int)hipMemcpyAsync(x.W(p.x0), x.W(p.features), s.Mp * D * p.es, hipMemcpyDeviceToDevice, x.st));
    } else {
      g.C = x.W(p.x0);
      g.ep.preact = x.W(p.features);
      g.ep.drop = x.drop(c.dropout_input, SITE_INPUT);
      RC(fh_gemm_launch(g, dt, x.st));
    }
  }
  // ---- cnn_proj_head on the pre-dropout projection (model.py:482-487): GELU -> Linear(D, P)
  if (c.cnn_head && p.k.heads_on) {
    RC(fh_unary_launch(dt, x.W(p.features), x.W(p.gfeat), s.Mp * D, 0, x.st));
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(x.W(p.gfeat), FH_R_CONTIG, D);
    g.Y = fh_operand(x.nat(p.pl.cnn_w), FH_R_CONTIG, D);
    g.C = x.W(p.cnn_pred);
    g.M = (int)s.Mp; g.N = P; g.Kseg = D; g.ldc = P;
    g.ep.bias = prm + p.pl.cnn_b;
    RC(fh_gemm_launch(g, dt, x.st));
  }
  // ---- positional conv + residual, LayerNorm, dropout   module.py:276-294
  RC(fh_posconv_fwd_launch(dt, x.W(p.x0), x.rel(p.wl.pos_wf), prm + p.pl.pos_b, x.W(p.zc), x.W(p.psum), B, s.Tp, D,
                           c.conv_pos_groups, c.conv_pos, x.st));
  const bool tr_first = c.tr_factor > 0 && s.idx == 0;   // the shipped layout: the encoder LayerNorm writes x1, the TR conv slot 0
  RC(fh_layernorm_fwd_launch(dt, x.W(p.psum), prm + p.pl.enc_ln_w, prm + p.pl.enc_ln_b, tr_first ? x.W(p.x1) : x.W(p.xo),
                             x.W<float>(p.ln1_mean), x.W<float>(p.ln1_rstd), (int)s.Mp, D, 1e-5f, x.drop(c.dropout, SITE_ENC), x.st));
  const long Ms = s.Ms;
  // ---- time-reduction conv (k = s = factor) in front of layer tr_layer_index   module.py:247-248,317-321
  auto tr_conv = [&]() -> int {
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(tr_first ? x.W(p.x1) : (void*)x.at(p.xo, (long)s.idx * Ms * D), FH_R_CONTIG, (long)s.f * D);
    g.X.rpb = s.T2; g.X.bstride = (long)s.Tp * D;
    g.Y = fh_operand(x.rel(p.wl.tr_wf), FH_R_CONTIG, (long)s.f * D);
    g.C = tr_first ? x.W(p.xo) : (void*)x.at(p.xo, (long)(s.idx + 1) * Ms * D);
    g.M = (int)s.M2; g.N = D; g.Kseg = s.f * D; g.ldc = D;
    g.ep.bias = prm + p.pl.tr_b;
    return fh_gemm_launch(g, dt, x.st);
  };
  // ---- transformer layers (post-LN)   module.py:557-582
  const float scale = 1.0f / sqrtf((float)(D / H));
  for (int l = 0; l < L; ++l) {
    if (c.tr_factor > 0 && l == s.idx) RC(tr_conv());
    const long M2 = s.Ml(l);             // rows of THIS layer (T' frames in front of the TR conv, reduced frames behind it)
    const int Tl = s.Tl(l);
    const int* len_l = l < s.idx ? len_p : len_2;
    const float* lp = prm + p.pl.layer0 + (long)l * p.pl.layer_stride;
    const long lpo = p.pl.layer0 + (long)l * p.pl.layer_stride;
    unsigned char* xin = x.at(p.xo, (long)s.slot_in(l) * Ms * D);
    unsigned char* xout = x.at(p.xo, (long)(s.slot_in(l) + 1) * Ms * D);
    const uint32_t site = SITE_LAYER0 + 4 * l;
    {  // fused q|k|v projection
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(xin, FH_R_CONTIG, D);
      g.Y = fh_operand(x.nat(lpo + p.pl.l_qkv_w), FH_R_CONTIG, D);
      g.C = x.at(p.qkv, (long)l * Ms * 3 * D);
      g.M = (int)M2; g.N = 3 * D; g.Kseg = D; g.ldc = 3 * D;
      g.ep.bias = lp + p.pl.l_qkv_b;
      RC(fh_gemm_launch(g, dt, x.st));
    }
    RC(fh_attention_fwd_launch(dt, x.at(p.qkv, (long)l * Ms * 3 * D), x.at(p.attn, (long)l * Ms * D),
                               x.W<float>(p.lse) + (long)l * B * H * s.Ts, len_l, B, Tl, H, D / H, scale,
                               x.drop(c.attention_dropout, site + 0), x.st));
    {  // out_proj + dropout1 + residual
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.at(p.attn, (long)l * Ms * D), FH_R_CONTIG, D);
      g.Y = fh_operand(x.nat(lpo + p.pl.l_o_w), FH_R_CONTIG, D);
      g.C = x.at(p.s1, (long)l * Ms * D);
      g.M = (int)M2; g.N = D; g.Kseg = D; g.ldc = D;
      g.ep.bias = lp + p.pl.l_o_b;
      g.ep.drop = x.drop(c.dropout, site + 1);
      g.ep.residual = xin;
      RC(fh_gemm_launch(g, dt, x.st));
    }
    RC(fh_layernorm_fwd_launch(dt, x.at(p.s1, (long)l * Ms * D), lp + p.pl.l_ln1_w, lp + p.pl.l_ln1_b, x.at(p.xa, (long)l * Ms * D),
                               x.W<float>(p.st1_mean) + l * Ms, x.W<float>(p.st1_rstd) + l * Ms, (int)M2, D, 1e-5f,
                               x.drop(0.f, 0), x.st));
    {  // fc1 + GELU + dropout2
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.at(p.xa, (long)l * Ms * D), FH_R_CONTIG, D);
      g.Y = fh_operand(x.nat(lpo + p.pl.l_fc1_w), FH_R_CONTIG, D);
      g.C = x.at(p.hbuf, (long)l * Ms * Fd);
      g.M = (int)M2; g.N = Fd; g.Kseg = D; g.ldc = Fd;
      g.ep.bias = lp + p.pl.l_fc1_b;
      g.ep.act = FH_ACT_GELU;
      if (tr) g.ep.preact = x.at(p.z1, (long)l * Ms * Fd);
      g.ep.drop = x.drop(c.activation_dropout, site + 2);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    {  // fc2 (layer_result) + dropout3 + residual
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.at(p.hbuf, (long)l * Ms * Fd), FH_R_CONTIG, Fd);
      g.Y = fh_operand(x.nat(lpo + p.pl.l_fc2_w), FH_R_CONTIG, Fd);
      g.C = x.at(p.s2, (long)l * Ms * D);
      g.M = (int)M2; g.N = D; g.Kseg = Fd; g.ldc = D;
      g.ep.bias = lp + p.pl.l_fc2_b;
      // fc2's output in front of dropout and residual = layer_results[i][2] of the reference's API; no loss and no backward reads it:
      // the distillation hot path (fh_student_call.training == 2) does not store the 12 copies (0.73 GB per B = 256 step)
      g.ep.preact = p.k.training == 2 ? nullptr : x.at(p.f2, (long)l * Ms * D);
      g.ep.drop = x.drop(c.dropout, site + 3);
      g.ep.residual = x.at(p.xa, (long)l * Ms * D);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    RC(fh_layernorm_fwd_launch(dt, x.at(p.s2, (long)l * Ms * D), lp + p.pl.l_ln2_w, lp + p.pl.l_ln2_b, xout,
                               x.W<float>(p.st2_mean) + l * Ms, x.W<float>(p.st2_rstd) + l * Ms, (int)M2, D, 1e-5f,
                               x.drop(0.f, 0), x.st));
  }
  if (c.tr_factor > 0 && s.idx == L) RC(tr_conv());   // tr_layer_index == encoder_layers: the TR conv is the encoder's last module
  const long M2 = s.M2;
  const unsigned char* enc_out = x.at(p.xo, (long)(s.slots(L) - 1) * Ms * D);   // the encoder's output x (reduced frames whenever a TR layer exists)
  // ---- DistilHuBERT-style head on the encoder output (model.py:503-516): [shared upsampler ->] Linear -> GELU -> SplitLinear (module.py:606-619)
  if (c.n_tasks > 0) {
    if (!p.k.heads_on) return 0;
    const int Nt = c.n_tasks, I = c.inter_dim;
    const void* head_in = enc_out;
    long Mh = M2;
    if (c.tr_factor > 0) {
      // model.py:401-407,504-505: x = upsampler(x), ConvTranspose1d(D, D, k = s = f) over time == one GEMM whose row m holds the f
      // up-sampled frames of x_m:  u[m][(j, co)] = sum_ci x[m][ci] W[ci][co][j] + b[co]
      GemmParams gu = fh_gemm_params();
      gu.X = fh_operand(enc_out, FH_R_CONTIG, D);
      gu.Y = fh_operand(x.rel(p.wl.head_wup), FH_R_CONTIG, D);   // [j][co][ci]
      gu.C = x.W(p.u);
      gu.M = (int)M2; gu.N = s.f * D; gu.Kseg = D; gu.ldc = (long)s.f * D;
      gu.ep.bias = prm + p.pl.ups_b;
      gu.ep.bias_mod = D;
      RC(fh_gemm_launch(gu, dt, x.st));
      head_in = x.W(p.u);
      Mh = M2 * s.f;
    }
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(head_in, FH_R_CONTIG, D);
    g.Y = fh_operand(x.nat(p.pl.sp_w0), FH_R_CONTIG, D);
    g.C = x.W(p.hh);
    g.M = (int)Mh; g.N = I * Nt; g.Kseg = D; g.ldc = (long)I * Nt;
    g.ep.bias = prm + p.pl.sp_b0;
    g.ep.act = FH_ACT_GELU;
    if (tr) g.ep.preact = x.W(p.hz);
    RC(fh_gemm_launch(g, dt, x.st));
    GemmParams g2 = fh_gemm_params();  // pred[n] = hh[:, n*I:(n+1)*I] . W2[n]  (+ b2[n])
    g2.X = fh_operand(x.W(p.hh), FH_R_CONTIG, (long)I * Nt);
    g2.X.gstride = I;
    if (Nt > 1) {
      g2.Y = fh_operand(x.nat(p.pl.sp_w2), FH_R_STRIDED, P);
      g2.Y.gstride = (long)I * P;
    } else {
      g2.Y = fh_operand(x.nat(p.pl.sp_w2), FH_R_CONTIG, I);  // nn.Linear weight [P][I]
    }
    g2.C = x.W(p.pred);
    g2.M = (int)Mh; g2.N = P; g2.Kseg = I; g2.ldc = P;
    g2.c_gstride = Mh * P;
    g2.groups = Nt;
    g2.ep.bias = prm + p.pl.sp_b2;
    g2.ep.bias_gstride = P;
    RC(fh_gemm_launch(g2, dt, x.st));
    return 0;
  }
  // ---- projection heads (LayerWiseProjHead, module.py:649-661): ConvTranspose1d(k = s = f) then Linear D -> P
  {
    const int Lh = p.k.heads_on ? L : 1;
    const int l0 = p.k.heads_on ? 0 : L - 1;
    const long hp0 = p.pl.head0 + (long)l0 * p.pl.head_stride;
    if (c.tr_factor > 0) {
      // merged head: pred[l][m][(j, p)] = x[l][m] . Wm[l][(j, p)] + bm[l][p]  -- row m of C holds the f up-sampled frames.
      // Heads on: one grouped launch per run of layers with equal frame counts (all of them when tr_layer_index == 0).  Heads off:
      // final_proj on the encoder output (model.py:500-502)
      const long pred_g = (p.k.heads_on ? Ms : M2) * s.f * P;
      auto head_group = [&](int la, int n, long rows, const void* in) -> int {
        GemmParams g = fh_gemm_params();
        g.X = fh_operand(in, FH_R_CONTIG, D);
        g.X.gstride = Ms * D;
        g.Y = fh_operand((const unsigned char*)x.rel(p.wl.head_wm) + (long)la * s.f * P * D * p.es, FH_R_CONTIG, D);
        g.Y.gstride = (long)s.f * P * D;
        g.C = x.at(p.pred, (long)(la - l0) * pred_g);
        g.M = (int)rows; g.N = s.f * P; g.Kseg = D; g.ldc = (long)s.f * P;
        g.c_gstride = pred_g;
        g.groups = n;
        g.ep.bias = (const float*)x.rel(p.wl.head_bm) + (long)la * P;
        g.ep.bias_gstride = P;
        g.ep.bias_mod = P;
        if (kd && kd->teacher) {
          // KD loss in this GEMM's epilogue (train.py:250-293): the workspace tensor "pred" receives d loss / d pred, `pred` itself is
          // never written; row m = (utterance, reduced frame), column = (sub-frame j, feature)
          if (!tr || !p.k.heads_on || s.idx > 0) return -823;
          g.c_rpb = s.T2;
          g.c_bstride = (long)s.T2 * s.f * P;
          g.ep.kd_teacher = kd->teacher;
          g.ep.kd_tdtype = kd->teacher_dtype;
          g.ep.kd_lstride = kd->teacher_lstride;
          g.ep.kd_layer = kd->teacher_layer;
          g.ep.kd_w = kd->layer_w;
          g.ep.kd_Tt = kd->Tt;
          g.ep.kd_P = P;
          g.ep.kd_type = kd->loss_type;
          g.ep.kd_grad_scale = kd->grad_scale;
          g.ep.kd_count = (float)((long)B * s.T2 * s.f * P);
          g.ep.kd_partials = kd->partials;
          if (kd->Tt < s.T2 * s.f) return -824;
        }
        return fh_gemm_launch(g, dt, x.st);
      };
      if (!p.k.heads_on) {
        RC(head_group(l0, 1, M2, enc_out));
      } else {
        if (s.idx > 0) RC(head_group(0, s.idx, s.Mp, x.at(p.xo, 1L * Ms * D)));
        if (s.idx < L) RC(head_group(s.idx, L - s.idx, M2, x.at(p.xo, (long)(s.slot_in(s.idx) + 1) * Ms * D)));
      }
      if (kd && kd->teacher) {
        const int nblk = (int)(((M2 + 63) / 64) * ((s.f * P + 63) / 64));
        RC(fh_kd_loss_final_launch(kd->partials, kd->layer_w, kd->out_losses, Lh, nblk, (double)B * s.T2 * s.f * P, x.st));
      }
      return 0;
    }
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(x.at(p.xo, (long)(l0 + 1) * M2 * D), FH_R_CONTIG, D);
    g.X.gstride = M2 * D;
    g.Y = fh_operand(x.nat(hp0 + p.pl.h_lin_w), FH_R_CONTIG, D);
    g.Y.gstride = p.pl.head_stride;
    g.C = x.W(p.pred);
    g.M = (int)(M2 * s.f); g.N = P; g.Kseg = D; g.ldc = P;
    g.c_gstride = M2 * s.f * P;
    g.groups = Lh;
    g.ep.bias = prm + hp0 + p.pl.h_lin_b;
    g.ep.bias_gstride = p.pl.head_stride;
    RC(fh_gemm_launch(g, dt, x.st));
  }
  return 0;
}

int backward_body(const Ctx& x, float* grads, const float* wave, const int* lens, const void* d_pred, const void* d_hidden,
                  const void* d_cnn, const void* d_qkv_last) {
  const Plan& p = x.p;
  const fh_student_config& c = p.c;
  const Shapes& s = p.sh;
  const int dt = x.dt, B = s.B, D = c.embed_dim, Fd = c.ffn_dim, P = c.pred_dim, L = c.layers, H = c.heads;
  if (!p.k.training) return -820;
  const int* len_p = lens;
  const int* len_2 = lens ? lens + B : nullptr;
  const long M2 = s.M2, Ms = s.Ms;
  const float scale = 1.0f / sqrtf((float)(D / H));
  const int Lh = p.k.heads_on ? L : 1;
  const int l0 = p.k.heads_on ? 0 : L - 1;
  const long hp0 = p.pl.head0 + (long)l0 * p.pl.head_stride;
  const bool tr_first = c.tr_factor > 0 && s.idx == 0;
  const bool tr_last = c.tr_factor > 0 && s.idx == L;      // the TR conv is the encoder's last module: its output is the encoder output
  const unsigned char* enc_out = x.at(p.xo, (long)(s.slots(L) - 1) * Ms * D);
  // gradient w.r.t. the ENCODER OUTPUT (the DistilHuBERT-style head, or final_proj with the heads off): slot L - 1 of ghead -- the
  // output of the last transformer layer, or of the TR conv when that comes last
  unsigned char* g_enc = x.at(p.ghead, (long)(L - 1) * Ms * D);

  // ---------------- heads: d_pred -> du -> ghead (+ d_hidden), and head weight gradients
  bool have_ghead = false;
  const bool split_head = c.n_tasks > 0;
  if (d_pred && split_head) {
    const int Nt = c.n_tasks, I = c.inter_dim;
    const long IN = (long)I * Nt;
    const bool ups = c.tr_factor > 0;                      // shared upsampler in front of the head (model.py:503-506)
    const long Mh = ups ? M2 * s.f : M2;
    const void* head_in = ups ? (const void*)x.W(p.u) : (const void*)enc_out;
    {  // dhz[:, n-block] = (d_pred[n] . W2[n]^T) * gelu'(hz)
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(d_pred, FH_R_CONTIG, P);
      g.X.gstride = Mh * P;
      if (Nt > 1) {
        g.Y = fh_operand(x.nat(p.pl.sp_w2), FH_R_CONTIG, P);
        g.Y.gstride = (long)I * P;
      } else {
        g.Y = fh_operand(x.nat(p.pl.sp_w2), FH_R_STRIDED, I);
      }
      g.C = x.W(p.dhz);
      g.M = (int)Mh; g.N = I; g.Kseg = P; g.ldc = IN;
      g.c_gstride = I;
      g.groups = Nt;
      g.ep.gelu_aux = x.W(p.hz);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    {  // dW2[n][r][j] += sum_m hh[m][n*I + r] * d_pred[n][m][j] ; db2[n] = colsum(d_pred[n])
      if (Nt > 1) {
        GemmOperand A = fh_operand(d_pred, FH_R_STRIDED, P);
        A.gstride = Mh * P;
        RC(wgrad(x, x.W(p.hh), IN, A, grads + p.pl.sp_w2, I, P, Mh, Nt, I, (long)I * P));
      } else {  // dW[j][r] += sum_m d_pred[m][j] * hh[m][r]
        GemmOperand A = fh_operand(x.W(p.hh), FH_R_STRIDED, I);
        RC(wgrad(x, d_pred, P, A, grads + p.pl.sp_w2, P, I, Mh, 1, 0, 0));
      }
      RC(fh_colsum_launch(dt, d_pred, grads + p.pl.sp_b2, (int)Mh, P, P, Nt, Mh * P, P, 0, 1.f, x.st));
    }
    {  // gradient w.r.t. the head's input = dhz . W0: the encoder output's gradient (+ d_hidden[L-1]), or du behind the shared upsampler
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.W(p.dhz), FH_R_CONTIG, IN);
      g.Y = fh_operand(x.nat(p.pl.sp_w0), FH_R_STRIDED, D);
      g.C = ups ? (void*)x.W(p.du) : (void*)g_enc;
      g.M = (int)Mh; g.N = D; g.Kseg = (int)IN; g.ldc = D;
      if (d_hidden && !ups && !tr_last) g.ep.residual = (const unsigned char*)d_hidden + (long)(L - 1) * Ms * D * p.es;
      RC(fh_gemm_launch(g, dt, x.st));
      GemmOperand A = fh_operand(head_in, FH_R_STRIDED, D);
      RC(wgrad(x, x.W(p.dhz), IN, A, grads + p.pl.sp_w0, (int)IN, D, Mh, 1, 0, 0));
      RC(fh_colsum_launch(dt, x.W(p.dhz), grads + p.pl.sp_b0, (int)Mh, (int)IN, IN, 1, 0, 0, 0, 1.f, x.st));
    }
    if (ups) {
      // shared upsampler backward (ConvTranspose1d k = s = f as one GEMM, see the forward): du as [M2][(j, co)]
      //   d enc_out[m][ci] = sum_(j,co) du[m][(j,co)] W[ci][co][j]   (+ d_hidden[L-1])
      //   dW[ci][co][j]   += sum_m enc_out[m][ci] du[m][(j,co)] ;  db[co] += sum_(m,j) du[m][(j,co)]
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.W(p.du), FH_R_CONTIG, (long)s.f * D);
      g.Y = fh_operand(x.rel(p.wl.head_wup), FH_R_STRIDED, D);   // [(j, co)][ci]: the reduction index is the row
      g.C = g_enc;
      g.M = (int)M2; g.N = D; g.Kseg = s.f * D; g.ldc = D;
      if (d_hidden && !tr_last) g.ep.residual = (const unsigned char*)d_hidden + (long)(L - 1) * Ms * D * p.es;
      RC(fh_gemm_launch(g, dt, x.st));
      GemmOperand A = fh_operand(enc_out, FH_R_STRIDED, D);
      RC(wgrad(x, x.W(p.du), (long)s.f * D, A, grads + p.pl.ups_w, s.f * D, D, M2, 1, 0, 0, /*jdiv*/ 1, /*sj1*/ (long)D * s.f, /*sj0*/ 0,
               /*idiv*/ D, /*si1*/ 1, /*si0*/ s.f));
      RC(fh_colsum_launch(dt, x.W(p.du), grads + p.pl.ups_b, (int)(M2 * s.f), D, D, 1, 0, 0, 0, 1.f, x.st));
    }
    have_ghead = true;
  } else if (d_pred) {
    if (c.tr_factor > 0) {
      const long fP = (long)s.f * P;
      const long pred_g = (p.k.heads_on ? Ms : M2) * fP;     // a head's slot in pred / d_pred (forward)
      // ghead[l] = d_pred[l] (as [rows][f*P]) . Wm[l]  (+ d_hidden): one grouped launch per run of layers with equal frame counts
      auto ghead_group = [&](int la, int n, long rows, unsigned char* out, bool add_hidden) -> int {
        GemmParams g = fh_gemm_params();
        g.X = fh_operand((const unsigned char*)d_pred + (long)(la - l0) * pred_g * p.es, FH_R_CONTIG, fP);
        g.X.gstride = pred_g;
        g.Y = fh_operand((const unsigned char*)x.rel(p.wl.head_wm) + (long)la * fP * D * p.es, FH_R_STRIDED, D);
        g.Y.gstride = fP * D;
        g.C = out;
        g.M = (int)rows; g.N = D; g.Kseg = (int)fP; g.ldc = D;
        g.c_gstride = Ms * D;
        g.groups = n;
        if (d_hidden && add_hidden) g.ep.residual = (const unsigned char*)d_hidden + (long)la * Ms * D * p.es;
        return fh_gemm_launch(g, dt, x.st);
      };
      if (!p.k.heads_on) {
        RC(ghead_group(l0, 1, M2, g_enc, !tr_last));
      } else {
        if (s.idx > 0) RC(ghead_group(0, s.idx, s.Mp, x.at(p.ghead, 0), true));
        if (s.idx < L) RC(ghead_group(s.idx, L - s.idx, M2, x.at(p.ghead, (long)s.idx * Ms * D), true));
      }
      // The heads' PARAMETER gradients are a branch of their own -- d_pred and the layer outputs in, the gradient buffer out, nothing
      // of the data-gradient chain reads them -- so the whole branch goes to the side stream (in order there).
      const hipStream_t hs = g_wside.fork(x.st);
      // dWm[l][(j, p)][ci] = sum_m d_pred[l][m][(j, p)] * x[l][m][ci]   (fp32 scratch), this pass's column sums of d_pred
      RC((int)hipMemsetAsync(x.W(p.dwm), 0, (long)Lh * fP * D * 4, hs));
      RC((int)hipMemsetAsync(x.W(p.dbl), 0, (long)Lh * P * 4, hs));
      auto head_wgrad_group = [&](int la, int n, long rows, const void* in) -> int {
        GemmOperand A = fh_operand(in, FH_R_STRIDED, D);
        A.gstride = Ms * D;
        const unsigned char* dp = (const unsigned char*)d_pred + (long)(la - l0) * pred_g * p.es;
        RC(wgrad(x, dp, fP, A, x.W<float>(p.dwm) + (long)(la - l0) * fP * D, (int)fP, D, rows, n, pred_g, fP * D, 0, 0, 0, 0, 0, 0, -1, hs));
        return fh_colsum_launch(dt, dp, x.W<float>(p.dbl) + (long)(la - l0) * P, (int)(rows * s.f), P, P, n, pred_g, P, 0, 1.f, hs);
      };
      if (!p.k.heads_on) {
        RC(head_wgrad_group(l0, 1, M2, enc_out));
      } else {
        if (s.idx > 0) RC(head_wgrad_group(0, s.idx, s.Mp, x.at(p.xo, 1L * Ms * D)));
        if (s.idx < L) RC(head_wgrad_group(s.idx, L - s.idx, M2, x.at(p.xo, (long)(s.slot_in(s.idx) + 1) * Ms * D)));
      }
      // chain rule onto the reference's parameters, accumulated into the gradient buffer: in the compute dtype (fp32 mode: exact fp32
      // GEMMs on the fp32 masters; bf16 mode: the fp32 merged-weight gradient is rounded to bf16 once and the four grouped products run
      // on the bf16 MFMA with fp32 accumulation -- the fp32 form cost 0.5 ms per optimizer step, 7 % of a step at the reference's batch)
      const unsigned char* dwm_c = x.W<unsigned char>(p.dwm);
      if (dt != 0) {
        RC(fh_cast_launch(dt, x.W<float>(p.dwm), x.W(p.dwm16), (long)Lh * fP * D, hs));
        dwm_c = x.W<unsigned char>(p.dwm16);
      }
      const unsigned char* wupc = (const unsigned char*)x.rel(p.wl.head_wupf) + (long)l0 * p.wl.head_wup_stride * p.es;
      for (int j = 0; j < s.f; ++j) {
        {  // dWlin[p][co] += sum_ci dWm[j][p][ci] * Wup[ci][co][j]
          GemmParams g = fh_gemm_params();
          g.X = fh_operand(dwm_c + (long)j * P * D * p.es, FH_R_CONTIG, D);
          g.X.gstride = fP * D;
          g.Y = fh_operand(wupc + (long)j * D * D * p.es, FH_R_CONTIG, D);
          g.Y.gstride = p.wl.head_wup_stride;
          g.C = grads + hp0 + p.pl.h_lin_w;
          g.M = P; g.N = D; g.Kseg = D; g.ldc = D;
          g.c_gstride = p.pl.head_stride;
          g.groups = Lh;
          g.ep.atomic = 1;
          RC(fh_gemm_launch(g, dt, hs));
        }
        {  // dWup[ci][co][j] += sum_p dWm[j][p][ci] * Wlin[p][co]
          GemmParams g = fh_gemm_params();
          g.X = fh_operand(dwm_c + (long)j * P * D * p.es, FH_R_STRIDED, D);
          g.X.gstride = fP * D;
          g.Y = fh_operand(x.nat(hp0 + p.pl.h_lin_w), FH_R_STRIDED, D);
          g.Y.gstride = p.pl.head_stride;
          g.C = grads + hp0 + p.pl.h_up_w + j;
          g.M = D; g.N = D; g.Kseg = P; g.ldc = (long)D * s.f;
          g.c_gstride = p.pl.head_stride;
          g.groups = Lh;
          g.ep.atomic = 1;
          g.ep.idiv = 1; g.ep.si1 = (long)D * s.f; g.ep.si0 = 0;
          g.ep.jdiv = 1; g.ep.sj1 = s.f; g.ep.sj0 = 0;
          RC(fh_gemm_launch(g, dt, hs));
        }
      }
      hipLaunchKernelGGL(head_bias_bwd_kernel, dim3(Lh, (D + 15) / 16), dim3(256), 0, hs, x.params, grads, hp0, p.pl.head_stride, p.pl.h_lin_w,
                         p.pl.h_lin_b, p.pl.h_up_b, (const float*)x.W<float>(p.dbl), P, D);
      FH_CHECK_LAUNCH();
    } else {
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(d_pred, FH_R_CONTIG, P);
      g.X.gstride = M2 * P;
      g.Y = fh_operand(x.nat(hp0 + p.pl.h_lin_w), FH_R_STRIDED, D);
      g.Y.gstride = p.pl.head_stride;
      g.C = x.at(p.ghead, (long)l0 * M2 * D);
      g.M = (int)M2; g.N = D; g.Kseg = P; g.ldc = D;
      g.c_gstride = M2 * D;
      g.groups = Lh;
      if (d_hidden) g.ep.residual = (const unsigned char*)d_hidden + (long)l0 * M2 * D * p.es;
      RC(fh_gemm_launch(g, dt, x.st));
    }
    have_ghead = true;
    // weight grads of the heads (Linear only; the time-reduced recipe handled them through the merged weights above)
    if (c.tr_factor == 0) {
      GemmOperand A = fh_operand(x.at(p.xo, (long)(l0 + 1) * M2 * D), FH_R_STRIDED, D);
      A.gstride = M2 * D;
      RC(wgrad(x, d_pred, P, A, grads + hp0 + p.pl.h_lin_w, P, D, M2, Lh, M2 * P, p.pl.head_stride));
      RC(fh_colsum_launch(dt, d_pred, grads + hp0 + p.pl.h_lin_b, (int)M2, P, P, Lh, M2 * P, p.pl.head_stride, 0, 1.f, x.st));
    }
  }

  // the projection heads' parameter gradients are complete (as queued work): first segment of the data-parallel exchange
  if (have_ghead && g_notify.on()) {
    RC(g_notify.fire(p.pl.head0, p.pl.total - p.pl.head0, true));
    g_notify.seg++;
    g_notify.heads_done = true;
  }
  // ---------------- grouped weight / bias gradients of layers [l0, l0 + nl): one launch per linear, on the side stream.  Issued
  // once for all layers behind the layer loop (chunks of 4 layers from inside the loop were measured: the smaller groups lose more
  // than the extra overlap gains).
  auto layer_wgrads = [&](int l0, int nl, long M2) -> int {   // M2 = rows of every layer of the run
    if (nl <= 0) return 0;
    const long lo = p.pl.layer0 + (long)l0 * p.pl.layer_stride, ls = p.pl.layer_stride;
    const hipStream_t ws_ = g_wside.fork(x.st);
    GemmOperand A = fh_operand(x.at(p.hbuf, (long)l0 * Ms * Fd), FH_R_STRIDED, Fd);
    A.gstride = Ms * Fd;
    RC(wgrad(x, x.at(p.df2, (long)l0 * Ms * D), D, A, grads + lo + p.pl.l_fc2_w, D, Fd, M2, nl, Ms * D, ls, 0, 0, 0, 0, 0, 0, -1, ws_));
    // (fc2's and out-proj's bias gradients come out of the LayerNorm backward kernels that WRITE df2 / dao: no pass over them here)
    A = fh_operand(x.at(p.xa, (long)l0 * Ms * D), FH_R_STRIDED, D);
    A.gstride = Ms * D;
    RC(wgrad(x, x.at(p.dz1, (long)l0 * Ms * Fd), Fd, A, grads + lo + p.pl.l_fc1_w, Fd, D, M2, nl, Ms * Fd, ls, 0, 0, 0, 0, 0, 0, -1, ws_));
    RC(fh_colsum_launch(dt, x.at(p.dz1, (long)l0 * Ms * Fd), grads + lo + p.pl.l_fc1_b, (int)M2, Fd, Fd, nl, Ms * Fd, ls, 0, 1.f, ws_));
    A = fh_operand(x.at(p.attn, (long)l0 * Ms * D), FH_R_STRIDED, D);
    A.gstride = Ms * D;
    RC(wgrad(x, x.at(p.dao, (long)l0 * Ms * D), D, A, grads + lo + p.pl.l_o_w, D, D, M2, nl, Ms * D, ls, 0, 0, 0, 0, 0, 0, -1, ws_));
    A = fh_operand(x.at(p.xo, (long)s.slot_in(l0) * Ms * D), FH_R_STRIDED, D);  // layer inputs = consecutive slots
    A.gstride = Ms * D;
    RC(wgrad(x, x.at(p.dqkv, (long)l0 * Ms * 3 * D), 3 * D, A, grads + lo + p.pl.l_qkv_w, 3 * D, D, M2, nl, Ms * 3 * D, ls, 0, 0, 0, 0, 0, 0, -1, ws_));
    // (the q|k|v bias gradients come with the attention backward; only a last layer that took transfer gradients into dqkv is summed here)
    if (d_qkv_last && l0 + nl == L)
      RC(fh_colsum_launch(dt, x.at(p.dqkv, (long)(L - 1) * Ms * 3 * D), grads + p.pl.layer0 + (long)(L - 1) * ls + p.pl.l_qkv_b, (int)M2, 3 * D,
                          3 * D, 1, 0, 0, 0, 1.f, ws_));
    return 0;
  };
  // ---------------- time-reduction conv backward: gradient w.r.t. its output (reduced frames) -> dx1tr (T' frames; the frames past
  // f x T2, which the conv does not reach, were zeroed at the head of the backward), weight / bias gradients
  auto tr_backward = [&](const void* dy_tr) -> int {
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(dy_tr, FH_R_CONTIG, D);
    g.Y = fh_operand(x.rel(p.wl.tr_wf), FH_R_STRIDED, (long)s.f * D);
    g.C = x.W(p.dx1tr);
    g.M = (int)M2; g.N = s.f * D; g.Kseg = D;
    g.ldc = (long)s.f * D; g.c_rpb = s.T2; g.c_bstride = (long)s.Tp * D;
    RC(fh_gemm_launch(g, dt, x.st));
    GemmOperand A = fh_operand(tr_first ? (const void*)x.W(p.x1) : (const void*)x.at(p.xo, (long)s.idx * Ms * D), FH_R_STRIDED, (long)s.f * D);
    A.rpb = s.T2; A.bstride = (long)s.Tp * D;
    RC(wgrad(x, dy_tr, D, A, grads + p.pl.tr_w, D, s.f * D, M2, 1, 0, 0, /*jdiv*/ D, /*sj1*/ 1, /*sj0*/ s.f, 0, 0, 0,
             /*ldc*/ (long)D * s.f));
    return fh_colsum_launch(dt, dy_tr, grads + p.pl.tr_b, (int)M2, D, D, 1, 0, 0, 0, 1.f, x.st);
  };
  // ---------------- transformer layers, last to first
  const void* dy_next = nullptr;  // gradient w.r.t. the output of layer l coming from what follows it
  if (tr_last && have_ghead && (split_head || !p.k.heads_on)) {
    // the encoder output is the TR conv's: its gradient (from the head) goes through the conv first
    RC(tr_backward(g_enc));
    dy_next = x.W(p.dx1tr);
  }
  for (int l = L - 1; l >= 0; --l) {
    if (c.tr_factor > 0 && !tr_last && l == s.idx - 1) {   // crossing the TR conv: dy_next is the gradient of ITS output
      if (!dy_next) return -821;
      RC(tr_backward(dy_next));
      dy_next = x.W(p.dx1tr);
    }
    const long M2 = s.Ml(l);
    const int Tl = s.Tl(l);
    const int* len_l = l < s.idx ? len_p : len_2;
    const long lpo = p.pl.layer0 + (long)l * p.pl.layer_stride;
    const float* lp = x.params + lpo;
    float* lg = grads + lpo;
    const uint32_t site = SITE_LAYER0 + 4 * l;
    const void* g_head = nullptr;
    if (have_ghead && ((p.k.heads_on && !split_head) || (l == L - 1 && !tr_last))) g_head = x.at(p.ghead, (long)l * Ms * D);
    else if (d_hidden) g_head = (const unsigned char*)d_hidden + (long)l * Ms * D * p.es;
    const void* dy = dy_next ? dy_next : g_head;
    const void* dy2 = dy_next ? g_head : nullptr;
    if (!dy) return -821;  // no gradient reaches this layer's output
    // LN2 backward -> ds2 ; df2 = ds2 * mask3
    const bool drop3 = c.dropout > 0.f;
    unsigned char* df2_l = x.at(p.df2, (long)l * Ms * D);
    RC(fh_layernorm_bwd_launch(dt, dy, dy2, x.at(p.s2, (long)l * Ms * D), x.W<float>(p.st2_mean) + l * Ms,
                               x.W<float>(p.st2_rstd) + l * Ms, lp + p.pl.l_ln2_w, x.W(p.ds2), df2_l, lg + p.pl.l_ln2_w,
                               lg + p.pl.l_ln2_b, (int)M2, D, x.drop(0.f, 0), x.drop(drop3 ? c.dropout : 0.f, site + 3), x.st,
                               /*fc2's bias gradient = column sums of df2*/ lg + p.pl.l_fc2_b));
    {  // dz1 = (df2 . W2) * mask2 * gelu'(z1)
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(df2_l, FH_R_CONTIG, D);
      if (const void* wt = x.layT(l, p.wl.t_fc2)) g.Y = fh_operand(wt, FH_R_CONTIG, D);   // W2^T [Fd][D]
      else g.Y = fh_operand(x.nat(lpo + p.pl.l_fc2_w), FH_R_STRIDED, Fd);
      g.C = x.at(p.dz1, (long)l * Ms * Fd);
      g.M = (int)M2; g.N = Fd; g.Kseg = D; g.ldc = Fd;
      g.ep.drop = x.drop(c.activation_dropout, site + 2);
      g.ep.gelu_aux = x.at(p.z1, (long)l * Ms * Fd);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    {  // dx1 = dz1 . W1 + ds2
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.at(p.dz1, (long)l * Ms * Fd), FH_R_CONTIG, Fd);
      if (const void* wt = x.layT(l, p.wl.t_fc1)) g.Y = fh_operand(wt, FH_R_CONTIG, Fd);  // W1^T [D][Fd]
      else g.Y = fh_operand(x.nat(lpo + p.pl.l_fc1_w), FH_R_STRIDED, D);
      g.C = x.W(p.dx1b);
      g.M = (int)M2; g.N = D; g.Kseg = Fd; g.ldc = D;
      g.ep.residual = x.W(p.ds2);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    unsigned char* dao_l = x.at(p.dao, (long)l * Ms * D);
    RC(fh_layernorm_bwd_launch(dt, x.W(p.dx1b), nullptr, x.at(p.s1, (long)l * Ms * D), x.W<float>(p.st1_mean) + l * Ms,
                               x.W<float>(p.st1_rstd) + l * Ms, lp + p.pl.l_ln1_w, x.W(p.ds1), dao_l, lg + p.pl.l_ln1_w,
                               lg + p.pl.l_ln1_b, (int)M2, D, x.drop(0.f, 0), x.drop(c.dropout, site + 1), x.st,
                               /*out-proj's bias gradient = column sums of dao*/ lg + p.pl.l_o_b));
    {  // da = dao . Wo
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(dao_l, FH_R_CONTIG, D);
      if (const void* wt = x.layT(l, p.wl.t_o)) g.Y = fh_operand(wt, FH_R_CONTIG, D);     // Wo^T [D][D]
      else g.Y = fh_operand(x.nat(lpo + p.pl.l_o_w), FH_R_STRIDED, D);
      g.C = x.W(p.da);
      g.M = (int)M2; g.N = D; g.Kseg = D; g.ldc = D;
      RC(fh_gemm_launch(g, dt, x.st));
    }
    RC(fh_attention_bwd_launch(dt, x.at(p.qkv, (long)l * Ms * 3 * D), x.at(p.attn, (long)l * Ms * D), x.W(p.da),
                               x.W<float>(p.lse) + (long)l * B * H * s.Ts, x.W<float>(p.delta),
                               x.at(p.dqkv, (long)l * Ms * 3 * D), len_l, B, Tl, H, D / H, scale,
                               x.drop(c.attention_dropout, site + 0), x.st, p.attn_scr >= 0 ? x.W<float>(p.attn_scr) : nullptr,
                               // the q|k|v bias gradient = column sums of dqkv, out of the fused kernels' accumulators -- except where the
                               // transfer gradients below still join dqkv (layer_wgrads sums that layer's stored gradient)
                               (l == L - 1 && d_qkv_last) ? nullptr : lg + p.pl.l_qkv_b));
    if (l == L - 1 && d_qkv_last)  // attention-logit / value-relation transfer gradients enter at the last layer's q|k|v
      RC(fh_unary_launch(dt, d_qkv_last, x.at(p.dqkv, (long)l * Ms * 3 * D), M2 * 3 * D, 1, x.st));
    {  // dxin = dqkv . Wqkv + ds1
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.at(p.dqkv, (long)l * Ms * 3 * D), FH_R_CONTIG, 3 * D);
      if (const void* wt = x.layT(l, p.wl.t_qkv)) g.Y = fh_operand(wt, FH_R_CONTIG, 3 * D);  // Wqkv^T [D][3D]
      else g.Y = fh_operand(x.nat(lpo + p.pl.l_qkv_w), FH_R_STRIDED, D);
      g.C = x.at(p.dxin, (long)(l & 1) * Ms * D);
      g.M = (int)M2; g.N = D; g.Kseg = 3 * D; g.ldc = D;
      g.ep.residual = x.W(p.ds1);
      RC(fh_gemm_launch(g, dt, x.st));
    }
    dy_next = x.at(p.dxin, (long)(l & 1) * Ms * D);
  }
  RC(layer_wgrads(0, s.idx, s.Mp));          // layers in front of the TR conv (none in the shipped layout)
  RC(layer_wgrads(s.idx, L - s.idx, s.M2));
  if (g_notify.on()) {  // every transformer layer's weights, biases and LayerNorm parameters
    RC(g_notify.fire(p.pl.layer0, (long)L * p.pl.layer_stride, true));
    g_notify.seg++;
    g_notify.layers_done = true;
  }
  const void* d_x1;  // gradient w.r.t. the encoder LayerNorm(+dropout) output
  if (tr_first) {
    RC(tr_backward(dy_next));
    d_x1 = x.W(p.dx1tr);
  } else {
    d_x1 = dy_next;
  }
  // ---------------- encoder LayerNorm (+dropout) backward, positional conv backward
  RC(fh_layernorm_bwd_launch(dt, d_x1, nullptr, x.W(p.psum), x.W<float>(p.ln1_mean), x.W<float>(p.ln1_rstd),
                             x.params + p.pl.enc_ln_w, x.W(p.dpsum), nullptr, grads + p.pl.enc_ln_w, grads + p.pl.enc_ln_b,
                             (int)s.Mp, D, x.drop(c.dropout, SITE_ENC), x.drop(0.f, 0), x.st));
  RC(fh_posconv_bwd_data_launch(dt, x.W(p.dpsum), x.W(p.zc), x.rel(p.wl.pos_wb), x.W(p.dzc), x.W(p.dx0), B, s.Tp, D,
                                c.conv_pos_groups, c.conv_pos, x.st));
  RC(fh_posconv_bwd_weight_launch(dt, x.W(p.dzc), x.W(p.x0), x.W<float>(p.pos_dws), grads + p.pl.pos_b, B, s.Tp, D,
                                  c.conv_pos_groups, c.conv_pos, x.st));
  RC(fh_posconv_wn_bwd_launch(x.params + p.pl.pos_g, x.params + p.pl.pos_v, (const float*)x.rel(p.wl.pos_norm),
                              x.W<float>(p.pos_dws), grads + p.pl.pos_g, grads + p.pl.pos_v, D, c.conv_pos_groups, c.conv_pos, x.st));
  // ---------------- post_extract_proj backward
  const int E = c.conv_dim[c.n_conv - 1];
  // results["features"] without a cnn_proj_head is the projection itself.  When it aliases the encoder input (dropout_input == 0:
  // F.dropout returns its argument and the encoder zeroes padded frames IN PLACE, see the oracle), its gradient shares that
  // zeroing: add it before the row mask.  (With a head and dropout_input == 0 the reference cannot run a padded batch at all:
  // the in-place zeroing invalidates the GELU's saved input.)
  const bool cnn_direct = d_cnn && !(c.cnn_head && p.k.heads_on);
  const bool cnn_aliased = cnn_direct && c.dropout_input == 0.f;
  if (cnn_aliased) RC(fh_unary_launch(dt, d_cnn, x.W(p.dx0), s.Mp * D, 1, x.st));
  RC(fh_mask_drop_launch(dt, x.W(p.dx0), x.W(p.dfeat), s.Mp, D, s.Tp, len_p, x.drop(c.dropout_input, SITE_INPUT), x.st));
  if (d_cnn && c.cnn_head && p.k.heads_on) {  // through cnn_proj_head: dfeat += (d_cnn . Wc) * gelu'(features); dWc, dbc
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(d_cnn, FH_R_CONTIG, P);
    g.Y = fh_operand(x.nat(p.pl.cnn_w), FH_R_STRIDED, D);
    g.C = x.W(p.dgf);
    g.M = (int)s.Mp; g.N = D; g.Kseg = P; g.ldc = D;
    g.ep.gelu_aux = x.W(p.features);
    RC(fh_gemm_launch(g, dt, x.st));
    RC(fh_unary_launch(dt, x.W(p.dgf), x.W(p.dfeat), s.Mp * D, 1, x.st));
    GemmOperand A = fh_operand(x.W(p.gfeat), FH_R_STRIDED, D);
    RC(wgrad(x, d_cnn, P, A, grads + p.pl.cnn_w, P, D, s.Mp, 1, 0, 0));
    RC(fh_colsum_launch(dt, d_cnn, grads + p.pl.cnn_b, (int)s.Mp, P, P, 1, 0, 0, 0, 1.f, x.st));
  } else if (cnn_direct && !cnn_aliased) {  // pre-dropout projection: its gradient joins dfeat behind the dropout / row mask
    RC(fh_unary_launch(dt, d_cnn, x.W(p.dfeat), s.Mp * D, 1, x.st));
  }
  {
    GemmParams g = fh_gemm_params();
    g.X = fh_operand(x.W(p.dfeat), FH_R_CONTIG, D);
    g.Y = fh_operand(x.nat(p.pl.post_w), FH_R_STRIDED, E);
    g.C = x.W(p.dln0);
    g.M = (int)s.Mp; g.N = E; g.Kseg = D; g.ldc = E;
    RC(fh_gemm_launch(g, dt, x.st));
    GemmOperand A = fh_operand(x.W(p.ln0), FH_R_STRIDED, E);
    RC(wgrad(x, x.W(p.dfeat), D, A, grads + p.pl.post_w, D, E, s.Mp, 1, 0, 0));
    RC(fh_colsum_launch(dt, x.W(p.dfeat), grads + p.pl.post_b, (int)s.Mp, D, D, 1, 0, 0, 0, 1.f, x.st));
  }
  // the LayerNorm behind the extractor sits OUTSIDE the reference's no_grad block (model.py:432-447): its weight / bias gradients
  // exist even when the extractor is frozen
  const int nl = c.n_conv - 1;
  RC(fh_layernorm_bwd_launch(dt, x.W(p.dln0), nullptr, x.W(p.y[nl]), x.W<float>(p.ln0_mean), x.W<float>(p.ln0_rstd),
                             x.params + p.pl.ln_w, x.W(p.dylast), nullptr, grads + p.pl.ln_w, grads + p.pl.ln_b, (int)s.Mp, E,
                             x.drop(0.f, 0), x.drop(0.f, 0), x.st));
  if (c.feature_grad_mult <= 0.f) return 0;  // model.py:432-434: extractor under no_grad
  // ---------------- conv frontend backward (GradMultiply: model.py:430-431 scales the gradient entering it)
  const void* dy0 = nullptr;
  if (nl == 0) {
    if (c.feature_grad_mult != 1.f) return -822;
    dy0 = x.W(p.dylast);
  } else {
    RC(fh_mul_gelu_grad_launch(dt, x.W(p.dylast), x.W(p.z[nl]), x.W(p.dz[nl]), s.Mp * E, c.feature_grad_mult, x.st, save_gelu_grad()));
  }
  for (int i = nl; i >= 1; --i) {
    const int cin = c.conv_dim[i - 1], cout = c.conv_dim[i], kk = c.conv_k[i], sd = c.conv_stride[i];
    const int Ti = s.T[i], Tin = s.T[i - 1];
    const long Mi = (long)B * Ti;
    // weight gradient: dW[co][ci][j] += sum_m dz[m][co] * window(m)[j*Cin + ci]
    GemmOperand A = fh_operand(x.W(p.y[i - 1]), FH_R_STRIDED, (long)sd * cin);
    A.rpb = Ti; A.bstride = (long)Tin * cin;
    RC(wgrad(x, x.W(p.dz[i]), cout, A, grads + p.pl.conv_w[i], cout, kk * cin, Mi, 1, 0, 0, /*jdiv*/ cin, /*sj1*/ 1, /*sj0*/ kk,
             0, 0, 0, /*ldc*/ (long)cin * kk, g_wside.fork(x.st)));  // dz[i] is final; y[i-1] is a forward activation
    // data gradient -> dz[i-1] = d(y[i-1]) * gelu'(z[i-1])   (for layer 0 the GELU' lives inside the conv0 backward)
    unsigned char* dst = x.W<unsigned char>(i - 1 >= 1 ? p.dz[i - 1] : p.dz[0]);
    const unsigned char* aux = i - 1 >= 1 ? x.W<unsigned char>(p.z[i - 1]) : nullptr;
    const void* wf = x.rel(p.wl.conv_wf[i]);
    const void* wt = x.rel(p.wl.conv_wt[i]);   // [k Cin][Cout]
    if (i == 1 && p.c0_fused) {
      // layer 1's data gradient IS d loss / d y0: its tiles are consumed in the GEMM's epilogue by conv 0's weight-gradient sums (one
      // utterance per group, so that a 256-frame tile shares its GroupNorm statistics and waveform window); dz[0] is never written
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.W(p.dz[1]), FH_R_CONTIG, cout);
      g.X.gstride = (long)Ti * cout;
      g.Y = fh_operand(wf, FH_R_STRIDED, (long)kk * cin);
      g.C = nullptr;
      g.M = Ti; g.N = cin; g.Kseg = cout; g.ldc = cin;
      g.groups = B;
      const long nchunk = fh_conv0_nchunk(s.T[0]);
      g.ep.c0_wave = wave;
      g.ep.c0_w = x.params + p.pl.conv_w[0];
      g.ep.c0_gamma = x.params + p.pl.gn_w;
      g.ep.c0_beta = x.params + p.pl.gn_b;
      g.ep.c0_mean = x.W<float>(p.c0_mean);
      g.ep.c0_rstd = x.W<float>(p.c0_rstd);
      g.ep.c0_wpart = reinterpret_cast<float*>(x.W<float2>(p.c0_scratch) + (long)B * nchunk * c.conv_dim[0] + (long)B * fh_conv0_ab_width(c.conv_dim[0]));
      g.ep.c0_S = s.S; g.ep.c0_K = c.conv_k[0]; g.ep.c0_stride = c.conv_stride[0];
      RC(fh_gemm_launch(g, dt, x.st));
      RC(fh_conv0_bwd_launch(dt, wave, x.params + p.pl.conv_w[0], x.params + p.pl.gn_w, x.params + p.pl.gn_b, nullptr,
                             x.W<float>(p.c0_mean), x.W<float>(p.c0_rstd), x.W<float2>(p.c0_scratch),
                             x.W<float2>(p.c0_scratch) + (long)B * nchunk * c.conv_dim[0], grads + p.pl.conv_w[0], grads + p.pl.gn_w,
                             grads + p.pl.gn_b, B, s.S, c.conv_dim[0], c.conv_k[0], c.conv_stride[0], 1.0f, x.st, 1, /*blocks of sums per utterance*/ 1));
      return 0;
    }
    if (kk == sd || (kk == 1 && sd == 1)) {
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.W(p.dz[i]), FH_R_CONTIG, cout);
      g.Y = fh_operand(wt, FH_R_CONTIG, cout);
      g.C = dst;
      g.M = (int)Mi; g.N = kk * cin; g.Kseg = cout;
      g.ldc = (long)sd * cin; g.c_rpb = Ti; g.c_bstride = (long)Tin * cin;
      g.ep.gelu_aux = aux;
      g.ep.aux_factor = save_gelu_grad();
      RC(fh_gemm_launch(g, dt, x.st));   // (the frames the conv does not reach were zeroed at the head of the backward: frontend_zero_unreached)
    } else if (getenv("FH_DGRAD_SPLIT")) {  // A/B aid: the two-product form (odd frames, then even frames with two segments)
      const int n_even = (2 * Ti <= Tin - 1) ? Ti + 1 : Ti;
      GemmParams g = fh_gemm_params();
      g.X = fh_operand(x.W(p.dz[i]), FH_R_CONTIG, cout);
      g.Y = fh_operand(wf, FH_R_STRIDED, (long)kk * cin);
      g.Y.segoff[0] = cin;
      g.C = dst + (long)cin * p.es;
      g.M = (int)Mi; g.N = cin; g.Kseg = cout;
      g.ldc = 2L * cin; g.c_rpb = Ti; g.c_bstride = (long)Tin * cin;
      g.ep.gelu_aux = aux ? aux + (long)cin * p.es : nullptr;
      g.ep.aux_factor = save_gelu_grad();
      RC(fh_gemm_launch(g, dt, x.st));
      GemmParams e = fh_gemm_params();
      e.X = fh_operand(x.W(p.dz[i]), FH_R_CONTIG, cout);
      e.X.rpb = n_even; e.X.bstride = (long)Ti * cout; e.X.nseg = 2; e.X.tshift[0] = 0; e.X.tshift[1] = -1; e.X.tvalid = Ti;
      e.Y = fh_operand(wf, FH_R_STRIDED, (long)kk * cin);
      e.Y.nseg = 2; e.Y.segoff[0] = 0; e.Y.segoff[1] = 2L * cin;
      e.C = dst;
      e.M = B * n_even; e.N = cin; e.Kseg = cout;
      e.ldc = 2L * cin; e.c_rpb = n_even; e.c_bstride = (long)Tin * cin;
      e.ep.gelu_aux = aux;
      e.ep.aux_factor = save_gelu_grad();
      RC(fh_gemm_launch(e, dt, x.st));
    } else {  // k = 3, s = 2: dx[2u] = dz[u] . W_0 + dz[u-1] . W_2,  dx[2u+1] = dz[u] . W_1
      // ONE product: output row u holds the frame pair (2u, 2u+1) -- 2*Cin contiguous columns --, reduction segment 0 is dz[u]
      // against [W_0 | W_1], segment 1 is dz[u-1] against [W_2 | 0]: the zero half is never multiplied (Y.seg1_limit: the column
      // tiles of the odd frames run one segment), dz is read once and every output row is written whole.  When the input has an
      // odd number of frames the last pair's odd frame does not exist (ep.tail_cols).
      const int n_even = (2 * Ti <= Tin - 1) ? Ti + 1 : Ti;
      GemmParams e = fh_gemm_params();
      e.X = fh_operand(x.W(p.dz[i]), FH_R_CONTIG, cout);
      e.X.rpb = n_even; e.X.bstride = (long)Ti * cout; e.X.nseg = 2; e.X.tshift[0] = 0; e.X.tshift[1] = -1; e.X.tvalid = Ti;
      e.Y = fh_operand(wt, FH_R_CONTIG, cout);   // rows (tap, ci): segment 1 = the rows of tap 2
      e.Y.nseg = 2; e.Y.segoff[0] = 0; e.Y.segoff[1] = 2L * cin * cout; e.Y.seg1_limit = cin;
      e.C = dst;
      e.M = B * n_even; e.N = 2 * cin; e.Kseg = cout;
      e.ldc = 2L * cin; e.c_rpb = n_even; e.c_bstride = (long)Tin * cin;
      e.ep.gelu_aux = aux;
      e.ep.aux_factor = save_gelu_grad();
      e.ep.tail_cols = 2 * n_even > Tin ? cin : 0;
      RC(fh_gemm_launch(e, dt, x.st));
    }
    if (i - 1 == 0) dy0 = dst;
  }
  RC(fh_conv0_bwd_launch(dt, wave, x.params + p.pl.conv_w[0], x.params + p.pl.gn_w, x.params + p.pl.gn_b, dy0,
                         x.W<float>(p.c0_mean), x.W<float>(p.c0_rstd), x.W<float2>(p.c0_scratch),
                         x.W<float2>(p.c0_scratch) + (long)B * fh_conv0_nchunk(s.T[0]) * c.conv_dim[0], grads + p.pl.conv_w[0],
                         grads + p.pl.gn_w, grads + p.pl.gn_b, B, s.S, c.conv_dim[0], c.conv_k[0], c.conv_stride[0], 1.0f, x.st, /*stats_ready: the forward's waveform sums are still in c0_scratch*/ 1));
  return 0;
}

// The input frames a conv layer's data gradient does not reach (behind the last window of an utterance) must read as zeros downstream.
// The dz buffers belong to the backward alone, so these B-row fills are issued at its HEAD (round 6): behind their GEMMs -- two launches
// of a few KB on the main queue -- each waited 0.14 - 0.28 ms for a CU beside the weight-gradient stream's long workgroups
// (tools/scratch/gap_report.py), on the critical path of the frontend backward.  Same rows as the branches of backward_body's layer loop.
int frontend_zero_unreached(const Ctx& x) {
  const Plan& p = x.p;
  const fh_student_config& c = p.c;
  const Shapes& s = p.sh;
  const int nl = c.n_conv - 1, B = (int)s.B;
  if (c.feature_grad_mult <= 0.f || nl < 1) return 0;
  for (int i = nl; i >= 1; --i) {
    if (i == 1 && p.c0_fused) break;   // dz[0] is never written: conv 0's sums come out of layer 1's data-gradient epilogue
    const int cin = c.conv_dim[i - 1], kk = c.conv_k[i], sd = c.conv_stride[i];
    const int Ti = s.T[i], Tin = s.T[i - 1];
    void* dst = x.W<unsigned char>(i - 1 >= 1 ? p.dz[i - 1] : p.dz[0]);
    int first;
    if (kk == sd || (kk == 1 && sd == 1)) first = sd * (Ti - 1) + kk;
    else {
      const int n_even = (2 * Ti <= Tin - 1) ? Ti + 1 : Ti;
      first = getenv("FH_DGRAD_SPLIT") ? (n_even == Ti + 1 ? 2 * Ti + 1 : 2 * Ti) : 2 * n_even;
    }
    if (first < Tin) RC(fh_zero_rows_launch(x.dt, dst, B, Tin, cin, first, x.st));
  }
  return 0;
}

// The parameter-gradient side stream is joined on EVERY exit of the backward -- the frozen-extractor return (feature_grad_mult == 0,
// model.py:432-434) and every error return included: the caller's next work on the main stream (gradient all-reduce, AdamW, the
// allocator handing the workspace out again) must be ordered behind the side stream's atomics into `grads` and its reads of `ws`.
int backward(const Ctx& x, float* grads, const float* wave, const int* lens, const void* d_pred, const void* d_hidden,
             const void* d_cnn = nullptr, const void* d_qkv_last = nullptr, const fh_grad_notify* nt = nullptr) {
  g_notify.n = nt;
  g_notify.main = x.st;
  g_notify.seg = 0;
  g_notify.heads_done = g_notify.layers_done = false;
  int rc = 0;
  {
    // small fills whose consumers sit deep in the backward, issued HERE: in front of their consumers each of them waited 0.2 - 0.7 ms
    // for a CU beside the weight-gradient stream's large launches (per-dispatch timeline, tools/timeline_list.py) -- the frames of the
    // time-reduction conv's data gradient that the conv does not reach, and the positional conv's weight-gradient accumulators
    const Plan& p = x.p;
    const fh_student_config& c = p.c;
    const Shapes& s = p.sh;
    const int D = c.embed_dim;
    if (c.tr_factor > 0) rc = fh_zero_rows_launch(x.dt, x.W(p.dx1tr), s.B, s.Tp, D, s.f * s.T2, x.st);
    const long cp = fh_posconv_cp(D / c.conv_pos_groups);
    if (!rc) rc = (int)hipMemsetAsync(x.W(p.pos_dws), 0, (long)c.conv_pos_groups * c.conv_pos * cp * cp * 4, x.st);
  }
  if (!rc && x.p.c0_fused) {
    // the utterances' blocks of conv-0 weight-gradient sums, which layer 1's data-gradient epilogue accumulates into: zeroed HERE, at
    // the head of the backward -- in front of that launch the 0.8 MB fill waited 0.74 ms for a CU beside the weight-gradient stream
    const Plan& p = x.p;
    const long B = p.sh.B, nchunk = fh_conv0_nchunk(p.sh.T[0]);
    float* wpart = reinterpret_cast<float*>(x.W<float2>(p.c0_scratch) + B * nchunk * p.c.conv_dim[0] + B * fh_conv0_ab_width(p.c.conv_dim[0]));
    rc = (int)hipMemsetAsync(wpart, 0, (size_t)fh_conv0_fused_wpart_floats((int)B, p.c.conv_dim[0]) * sizeof(float), x.st);
  }
  if (!rc) rc = frontend_zero_unreached(x);
  if (!rc) rc = backward_body(x, grads, wave, lens, d_pred, d_hidden, d_cnn, d_qkv_last);
  const int j = g_wside.join(x.st);
  if (!rc) rc = j;
  if (!rc && g_notify.on()) {
    // everything not reported yet, in buffer order (the main stream has joined the side stream: one wait covers all of it)
    const ParamLayout& pl = x.p.pl;
    const long lay_end = pl.layer0 + (long)x.p.c.layers * pl.layer_stride;
    long a = 0;
    auto flush_to = [&](long b) -> int {
      if (b > a) RC(g_notify.fire(a, b - a, false));
      return 0;
    };
    if (g_notify.layers_done) { rc = flush_to(pl.layer0); a = lay_end; }
    if (!rc && g_notify.heads_done) { rc = flush_to(pl.head0); a = pl.total; }
    if (!rc) rc = flush_to(pl.total);
  }
  g_notify.n = nullptr;
  return rc;
}

}  // namespace

extern "C" {

long fh_student_param_count(const fh_student_config* c) {
  if (!c || validate(*c)) return -1;
  return ParamLayout(*c).total;
}

int fh_student_param_info(const fh_student_config* c, int idx, char* name, int name_cap, long* offset, int* ndim, long* dims) {
  if (!c || validate(*c)) return -1;
  ParamLayout pl(*c);
  if (idx < 0 || idx >= (int)pl.list.size()) return 1;
  const PEntry& e = pl.list[idx];
  if (name && name_cap > 0) {
    std::strncpy(name, e.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (offset) *offset = e.off;
  if (ndim) *ndim = e.ndim;
  if (dims) for (int i = 0; i < 4; ++i) dims[i] = e.dims[i];
  return 0;
}

long fh_student_wc_bytes(const fh_student_config* c) {
  if (!c || validate(*c)) return -1;
  ParamLayout pl(*c);
  return WcLayout(*c, pl).total;
}

long fh_student_ws_bytes(const fh_student_config* c, const fh_student_call* k) {
  if (!c || !k || validate(*c)) return -1;
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  return p.ws.total;
}

int fh_student_ws_info(const fh_student_config* c, const fh_student_call* k, const char* name, long* byte_offset, int* ndim,
                       long* dims, int* elem_bytes) {
  if (!c || !k || validate(*c)) return -1;
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  for (auto& b : p.ws.bufs)
    if (b.name == name) {
      if (byte_offset) *byte_offset = b.off;
      if (ndim) *ndim = b.ndim;
      if (dims) for (int i = 0; i < 4; ++i) dims[i] = b.dims[i];
      if (elem_bytes) *elem_bytes = b.esize;
      return 0;
    }
  return 1;
}


int fh_student_frames(const fh_student_config* c, int S, int* T_conv, int* T_enc) {
  if (!c || validate(*c)) return -1;
  Shapes s;
  int rc = make_shapes(*c, 1, S, s);
  if (rc) return rc;
  if (T_conv) *T_conv = s.Tp;
  if (T_enc) *T_enc = s.T2;
  return 0;
}

int fh_student_side_stream(fh_stream_t stream) {
  g_wside.replace((hipStream_t)stream);
  return g_wside.state == 1 ? 0 : 1;
}

int fh_student_prepare(const fh_student_config* c, const float* params, void* wc, fh_stream_t stream) {
  if (!c) return -1;
  RC(validate(*c));
  fh_student_call k;
  k.B = 1; k.S = 400000; k.training = 0; k.heads_on = 1; k.seed = 0;
  Plan p(*c, k);
  return prepare(p, params, (unsigned char*)wc, (hipStream_t)stream);
}

int fh_student_forward(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                       const float* wave, const int* lens, fh_stream_t stream) {
  return fh_student_forward_kd(c, k, params, wc, ws, wave, lens, nullptr, stream);
}
long fh_student_kd_partials(const fh_student_config* c, const fh_student_call* k) {
  if (!c || !k || validate(*c) != 0) return -1;
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  if (c->n_tasks > 0 || c->tr_factor <= 0) return 0;   // the fused form exists for the merged layer-wise heads
  const long Lh = k->heads_on ? c->layers : 1;
  return Lh * ((p.sh.M2 + 63) / 64) * (((long)p.sh.f * c->pred_dim + 63) / 64);
}
int fh_student_forward_kd(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                          const float* wave, const int* lens, const fh_student_kd* kd, fh_stream_t stream) {
  if (!c || !k) return -1;
  RC(validate(*c));
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  if (kd && kd->teacher && (c->n_tasks > 0 || c->tr_factor <= 0 || !k->training || !k->heads_on)) return -823;
  Ctx x{p, params, (const unsigned char*)wc, (unsigned char*)ws, (hipStream_t)stream, c->dtype};
  return forward(x, wave, lens, kd);
}

int fh_student_backward(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                        float* grads, const float* wave, const int* lens, const void* d_pred, const void* d_hidden,
                        fh_stream_t stream) {
  if (!c || !k) return -1;
  RC(validate(*c));
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  Ctx x{p, params, (const unsigned char*)wc, (unsigned char*)ws, (hipStream_t)stream, c->dtype};
  return backward(x, grads, wave, lens, d_pred, d_hidden);
}

int fh_student_backward_cnn(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                            float* grads, const float* wave, const int* lens, const void* d_pred, const void* d_hidden,
                            const void* d_cnn, fh_stream_t stream) {
  if (!c || !k) return -1;
  RC(validate(*c));
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  Ctx x{p, params, (const unsigned char*)wc, (unsigned char*)ws, (hipStream_t)stream, c->dtype};
  return backward(x, grads, wave, lens, d_pred, d_hidden, d_cnn);
}

int fh_student_backward_ex(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                           float* grads, const float* wave, const int* lens, const fh_student_grads* g, fh_stream_t stream) {
  if (!c || !k || !g) return -1;
  RC(validate(*c));
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  Ctx x{p, params, (const unsigned char*)wc, (unsigned char*)ws, (hipStream_t)stream, c->dtype};
  return backward(x, grads, wave, lens, g->d_pred, g->d_hidden, g->d_cnn, g->d_qkv_last);
}

int fh_student_backward_notify(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                               float* grads, const float* wave, const int* lens, const fh_student_grads* g, const fh_grad_notify* n,
                               fh_stream_t stream) {
  if (!c || !k || !g) return -1;
  RC(validate(*c));
  Plan p(*c, *k);
  if (p.ok != 0) return p.ok;
  Ctx x{p, params, (const unsigned char*)wc, (unsigned char*)ws, (hipStream_t)stream, c->dtype};
  return backward(x, grads, wave, lens, g->d_pred, g->d_hidden, g->d_cnn, g->d_qkv_last, n);
}

}  // extern "C"
