// The frozen teacher's transformer layers as ONE library call (fh_teacher_encoder_forward, include/fithubert_hip.h): the host side
// no longer walks the layers from Python (8 - 9 ctypes calls and as many tensor allocations per layer: 1.4 ms of host time per
// forward, which bounds the step at the reference's own batch of 3 utterances -- DESIGN.md section 4, small batches).
// fairseq TransformerSentenceEncoderLayer as the reference loads it (utils/utils.py:102-149) and hooks it (utils/utils.py:61-99):
// post-LN (HuBERT / wav2vec 2.0 Base) and pre-LN (layer_norm_first, HuBERT Large) forms; the layer output is what the forward hook
// captures.  Same kernels, same order, same arithmetic as the per-layer entry points it replaces.
#include "gemm.h"
#include "kernels.h"
#include "../../include/fithubert_hip.h"

namespace {
inline long al256(long n) { return (n + 255) & ~255L; }
struct Scratch {
  long qkv, attn, o, x1, h, hn, total;
  Scratch(int dtype, long M, int D, int F) {
    const long es = dtype == 0 ? 4 : 2;
    long at = 0;
    auto add = [&](long n) { long a = at; at += al256(n * es); return a; };
    qkv = add(M * 3 * D); attn = add(M * D); o = add(M * D); x1 = add(M * D); h = add(M * F); hn = add(M * D);
    total = at;
  }
};
int linear(int dt, const void* x, const void* w, const float* b, const void* res, void* pre, void* y, long M, int K, int N, int gelu,
           hipStream_t st) {
  GemmParams g = fh_gemm_params();
  g.X = fh_operand(x, FH_R_CONTIG, K);
  g.Y = fh_operand(w, FH_R_CONTIG, K);
  g.C = y;
  g.M = (int)M; g.N = N; g.Kseg = K; g.ldc = N;
  g.ep.bias = b;
  g.ep.act = gelu ? FH_ACT_GELU : FH_ACT_NONE;
  g.ep.residual = res;
  g.ep.preact = pre;
  return fh_gemm_launch(g, dt, st);
}
}  // namespace

#define RC(e) do { int rc_ = (e); if (rc_) return rc_; } while (0)

extern "C" long fh_teacher_encoder_scratch_bytes(int dtype, int B, int T, int D, int F) {
  return Scratch(dtype, (long)B * T, D, F).total;
}

extern "C" int fh_teacher_encoder_forward(int dtype, const fh_teacher_layer* layers, int n_layers, int pre_ln, const void* x_in,
                                          void* hidden, void* fc2_out, void* qkv_last, const int* key_len, int B, int T, int D, int F,
                                          int heads, void* scratch, fh_stream_t stream) {
  if (!layers || !x_in || !hidden || !scratch || n_layers < 1 || heads < 1 || D % heads) return -301;
  hipStream_t st = (hipStream_t)stream;
  const long M = (long)B * T, es = dtype == 0 ? 4 : 2;
  const int dh = D / heads;
  const Scratch s(dtype, M, D, F);
  unsigned char* base = static_cast<unsigned char*>(scratch);
  void *attn = base + s.attn, *o = base + s.o, *x1 = base + s.x1, *h = base + s.h, *hn = base + s.hn;
  const fh_drop nodrop{0.f, 0u, 0u};
  const float scale = 1.0f / sqrtf((float)dh);
  const void* x = x_in;
  for (int i = 0; i < n_layers; ++i) {
    const fh_teacher_layer& w = layers[i];
    void* out = static_cast<unsigned char*>(hidden) + (long)i * M * D * es;
    void* lr = fc2_out ? static_cast<unsigned char*>(fc2_out) + (long)i * M * D * es : nullptr;
    void* qkv = (qkv_last && i == n_layers - 1) ? qkv_last : base + s.qkv;   // the last layer's projections: attention / value-relation transfer
    if (pre_ln) {   // x += attn(LN(x)); x += ffn(LN(x)); the residual adds ride in the GEMM epilogues
      RC(fh_layernorm_fwd_launch(dtype, x, w.g1, w.b1, hn, nullptr, nullptr, (int)M, D, 1e-5f, nodrop, st));
      RC(linear(dtype, hn, w.wqkv, w.bqkv, nullptr, nullptr, qkv, M, D, 3 * D, 0, st));
      RC(fh_attention_fwd_launch(dtype, qkv, attn, nullptr, key_len, B, T, heads, dh, scale, nodrop, st));
      RC(linear(dtype, attn, w.wo, w.bo, x, nullptr, x1, M, D, D, 0, st));
      RC(fh_layernorm_fwd_launch(dtype, x1, w.g2, w.b2, hn, nullptr, nullptr, (int)M, D, 1e-5f, nodrop, st));
      RC(linear(dtype, hn, w.w1, w.c1, nullptr, nullptr, h, M, D, F, 1, st));
      RC(linear(dtype, h, w.w2, w.c2, x1, lr, out, M, F, D, 0, st));
    } else {        // x = LN(x + attn(x)); x = LN(x + ffn(x))
      RC(linear(dtype, x, w.wqkv, w.bqkv, nullptr, nullptr, qkv, M, D, 3 * D, 0, st));
      RC(fh_attention_fwd_launch(dtype, qkv, attn, nullptr, key_len, B, T, heads, dh, scale, nodrop, st));
      RC(linear(dtype, attn, w.wo, w.bo, nullptr, nullptr, o, M, D, D, 0, st));
      RC(fh_add_layernorm_fwd_launch(dtype, o, x, w.g1, w.b1, x1, (int)M, D, 1e-5f, st));
      RC(linear(dtype, x1, w.w1, w.c1, nullptr, nullptr, h, M, D, F, 1, st));
      void* f2 = lr ? lr : o;
      RC(linear(dtype, h, w.w2, w.c2, nullptr, nullptr, f2, M, F, D, 0, st));
      RC(fh_add_layernorm_fwd_launch(dtype, f2, x1, w.g2, w.b2, out, (int)M, D, 1e-5f, st));
    }
    x = out;
  }
  return 0;
}
