/* fithubert_hip.h -- C ABI of the MI355X-native FitHuBERT distillation hot path.
 *
 * The reference (glory20h/FitHuBERT) is pure Python on torch: it has no native boundary of its
 * own (last row).  The boundary below is therefore defined by this
 * repository; each entry point cites the reference code whose arithmetic it replaces.
 * Plain pointers + sizes + a hipStream_t; no torch types.  All functions are stream-ordered,
 * allocate nothing, never synchronise, and return 0 on success (a hipError_t or a negative
 * argument-check code otherwise).
 *
 * dtype: 0 = float (exact fp32 MFMA, parity mode), 1 = bf16 storage / bf16 MFMA / fp32 accumulate,
 *        2 = IEEE half storage / f16 MFMA / fp32 accumulate -- the reference's own AMP precision (train.py:469,496,
 *            data/conf/fithubert.yaml:11 use_fp16), same MFMA rate as bf16, three more significand bits.
 */
#ifndef FITHUBERT_HIP_H
#define FITHUBERT_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fh_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------------------------------
 * Generic MFMA GEMM   C[i][j] = epilogue(alpha * sum_r X(i,r) * Y(j,r))
 * Replaces every nn.Linear / nn.Conv1d / nn.ConvTranspose1d matmul of the student and their
 * autograd backward (modules/module.py:36-102 conv blocks, :230-236 TR conv, :498-515 MHA+FFN
 * linears, :634-647 LayerWiseProjHead; modules/model.py:298-302 post_extract_proj).
 * ------------------------------------------------------------------------------------------ */
enum { FH_R_CONTIG = 0, FH_R_STRIDED = 1 };
enum { FH_ACT_NONE = 0, FH_ACT_GELU = 1 };

typedef struct {
  float p;       /* drop probability, 0 = off */
  uint32_t seed; /* per-step seed */
  uint32_t site; /* dropout-site id */
} fh_drop;

typedef struct {
  const void* p;
  int layout;     /* FH_R_CONTIG: element (idx, r) at base(idx) + r ; FH_R_STRIDED: at base(r) + idx */
  long ld;        /* elements between consecutive mapped rows */
  int rpb;        /* mapped rows per batch item (0 = no batch mapping) */
  long bstride;   /* elements between batch items */
  long gstride;   /* elements between groups */
  int nseg;       /* 1 or 2 reduction segments */
  int tshift[2];  /* R_CONTIG: per-segment shift of the mapped row; rows outside [0, tvalid) read as 0 */
  int tvalid;     /* 0 = no check */
  long segoff[2]; /* per-segment element offset */
  int seg1_limit; /* > 0: operand indices >= seg1_limit read as 0 in segment 1, and output tiles that lie entirely beyond it skip
                     the segment (Y operand: the k = 3, s = 2 conv data gradient as ONE product whose odd output rows have a
                     single tap); multiple of 8; 0 = no limit */
} fh_gemm_operand;

typedef struct {
  float alpha;
  const float* bias; /* fp32 [N] or NULL */
  long bias_gstride;
  int bias_mod;      /* bias index = j % bias_mod when > 0 */
  void* preact;      /* receives alpha*acc + bias (same addressing as C) or NULL */
  int act;           /* FH_ACT_* */
  fh_drop drop;      /* dropout after the activation, index = (g*M + i)*N + j */
  const void* gelu_aux; /* multiply by gelu'(aux) (same addressing as C) or NULL */
  const void* residual; /* added last (same addressing as C) or NULL */
  const int* row_len;   /* rows t >= row_len[b] are written as 0, or NULL */
  int out_f32;          /* C is float even when dtype is bf16 */
  int atomic;           /* C (fp32) [map(i,j)] += alpha*acc : weight-gradient accumulation / split-K */
  int idiv; long si1, si0; /* map(i,j) = (i/idiv)*si1 + (i%idiv)*si0 + (j/jdiv)*sj1 + (j%jdiv)*sj0 */
  int jdiv; long sj1, sj0; /* (idiv == 0: i*ldc ; jdiv == 0: j) */
  int tail_cols;           /* > 0: the LAST mapped row of every batch item (t == c_rpb - 1) is written in columns < tail_cols only
                              (an output row that packs two frames when the item has an odd number of them); multiple of 8 */
  int aux_factor;          /* 1: the saved tensor holds GELU'(pre-activation) instead of the pre-activation.  Forward (act = GELU,
                              preact set): `preact` receives GELU'(alpha X.Y + bias) -- the sigmoid of the activation is at hand there,
                              the derivative costs 5 more VALU issues; backward (gelu_aux set): C = (X.Y) * gelu_aux, no GELU'
                              evaluation (it cost ~17 issues per element in the conv data-gradient epilogues) */
  /* KD-loss epilogue (train.py:250-293 inside the merged projection heads' GEMM, round 3): with kd_teacher set, v = alpha X.Y + bias is
   * rounded to the output type -- exactly the `pred` element the unfused path stores -- and compared with the teacher frame it
   * predicts: row i = (utterance b = i / c_rpb, frame t = i % c_rpb), column c = (sub-frame j = c / kd_P, feature c % kd_P) against
   * teacher[kd_layer[g]][b][(N / kd_P) t + j][c % kd_P].  C receives d loss / d v = kd_grad_scale * kd_w[g] / kd_count * (2 (v - t) | sign(v - t));
   * the sum of (v - t)^2 | |v - t| of every 64-row x 64-column block goes to kd_partials[(g * ceil(M / 64) + i / 64) * ceil(N / 64) + c / 64]
   * (one wave each, fixed order: bit-reproducible; fh_kd_loss_finish turns them into the per-layer means).  `pred` itself is never
   * written.  Requires c_rpb > 0, rows of one utterance contiguous, no other per-row operand; runs on loop 7 (16-bit) / the fp32 kernel. */
  const void* kd_teacher;
  int kd_tdtype;           /* element type of the teacher tensor (dtype code) */
  long kd_lstride;         /* elements between teacher layers */
  const int* kd_layer;     /* [groups] teacher layer of each group */
  const float* kd_w;       /* [groups] loss weight of each group (0: the group's gradient is 0 and its loss is not accumulated) */
  int kd_Tt, kd_P, kd_type; /* teacher frames per utterance, feature width, 0 = mse / 1 = l1 */
  float kd_grad_scale, kd_count; /* see above; kd_count = elements of one group's loss mean (B * Tp * P) */
  float* kd_partials;
  /* conv-0 weight-gradient epilogue (modules/module.py:65-71 backward, round 3): with c0_wpart set the product is d loss / d y0 of ONE
   * utterance per group (rows = frames, N = C0 = 128 columns = channels: the k = 1 conv layer 1's data gradient) and is CONSUMED in
   * registers instead of stored: g = v * GELU'(GroupNorm(conv0(wave))) with the conv recomputed from the waveform, and the sums
   * A_j = sum_t g x_j[t] (j < c0_K), S1 = sum g, S2 = sum g xhat of the tile's 256 frames are ADDED (fp32 atomics; zero it first) to
   * c0_wpart[(group * C0 + channel) * (10 + 2) + j]  (one block of sums per utterance, the layout conv0_bwd_onepass_kernel leaves with
   * one block; fh_conv0_backward's combine step finishes).  d loss / d y0 (1.05 GB at B = 128) is never written or read back.  16-bit, c0_K <= 10, loop 4. */
  const float* c0_wave;    /* [groups][c0_S] */
  const float* c0_w;       /* [C0][c0_K] */
  const float* c0_gamma;   /* [C0] */
  const float* c0_beta;
  const float* c0_mean;    /* [groups][C0] GroupNorm statistics of the forward */
  const float* c0_rstd;
  float* c0_wpart;
  int c0_S, c0_K, c0_stride;
} fh_gemm_epilogue;

typedef struct {
  fh_gemm_operand X, Y;
  void* C;
  int M, N;   /* output rows i, columns j */
  int Kseg;   /* reduction length per segment */
  long ldc;   /* C rows are mapped like an R_CONTIG operand */
  int c_rpb;
  long c_bstride;
  long c_gstride;
  int groups;
  int splitk; /* low 16 bits: reduction splits (> 1 requires atomic); bit 16: co-locate the tiles of one split on one XCD (set by the
                 per-shape tuner, fh_gemm_tune) */
  fh_gemm_epilogue ep;
} fh_gemm_desc;

int fh_gemm(const fh_gemm_desc* d, int dtype, fh_stream_t stream);
/* Optional per-launch HIP-event timing of the GEMM kernel on its own stream (used by bench.py's roofline object).
 * Slots are indexed loop*8 + dtype*4 + X.layout*2 + Y.layout (loop: the main loop that ran, fh_gemm_set_path numbering).
 * collect() waits for the recorded events, sums into 64-entry arrays, and clears. */
/* bf16 main loop: 0 = 16x16x32 MFMA register staging (128x128 tile), 1 = small-M loop (64x64 tiles, the whole reduction extent in LDS
 * in one LDS-DMA trip; R_CONTIG x R_CONTIG), 2 = 32x32x16 MFMA register
 * staging (128x128), 3 / 4 / 5 = large-tile LDS-DMA rings (256x256 BK 64, 256x128 BK 32, 256x256 BK 32), 6 = 256x256 ping-pong
 * ring, 7 = 256x256 8-phase loop, 8 = the same as a STREAMING persistent launch (one workgroup per CU walks the tiles and issues the
 * next tile's first two K-tiles before the epilogue of the current one; launches with the bf16-staged epilogue only, others fall
 * back to loop 2), 9 = per-shape choice (default), -1 = re-read env FH_GEMM_PATH (reg | 1..8). */
int fh_gemm_set_path(int path);
/* Restrict the per-shape choice for launches on `stream` to the loops whose bit is set in `mask` (bit i = loop i, loop 2 always stays
 * available as the fallback; 0 = no restriction).  The 256 x 256 loops claim 128 - 160 KiB of a CU's LDS, i.e. the whole CU; loops 2
 * (40 KiB) and 4 (72 KiB) let kernels of OTHER streams share it -- the frozen teacher's stream beside the student's. */
int fh_gemm_stream_loops(fh_stream_t stream, unsigned mask);
/* The main loop the last bf16 launch ran (fh_gemm_set_path numbering): what a forced or tuned path resolved to. */
int fh_gemm_last_loop(void);
/* CUs the persistent loop 8 leaves free (rounded up to a multiple of 8, one per XCD; returns the previous value, n < 0 only reads).
 * A persistent workgroup holds its CU's registers and LDS for the whole launch, so kernels of other streams that must make
 * progress meanwhile -- RCCL's collectives in a data-parallel run (train.py:494) -- need CUs it does not occupy. */
int fh_gemm_reserve_cus(int n);
/* The same for the launches of ONE stream (n <= 0 removes the entry; the larger of the two values applies): a data-parallel run holds
 * CUs back on the frozen teacher's low-priority stream only -- its persistent launches are the long ones (up to 1.6 ms), the
 * student's leave every CU within ~0.1 - 0.4 ms. */
int fh_gemm_stream_reserve_cus(fh_stream_t stream, int n);
/* A HIP stream restricted to `count` of the device's CUs: bits [first, first + count) of the CU mask (hipExtStreamCreateWithCUMask;
 * consecutive bits are dealt round-robin over the 8 XCDs, so a contiguous range is an equal share of every XCD).  Partitions the chip
 * between the frozen teacher's forward (train.py:131-134) and the student's forward / backward instead of letting their kernels share
 * CUs; the persistent GEMM loop sizes its grid to the share.  The handle is a hipStream_t; fh_stream_destroy releases it. */
int fh_stream_create_cu_range(int first, int count, fh_stream_t* out);
int fh_stream_destroy(fh_stream_t stream);
/* Diagnostic: n single-wave workgroups on `stream`, each writing (XCC id << 16 | HW_REG_HW_ID[15:0]) -- which CUs a stream reaches. */
int fh_stream_cu_census(unsigned* out, int n, fh_stream_t stream);
/* Diagnostic stand-in for a collective's reduction kernel: `workgroups` x 512 threads, dst[i] += src[i] over n floats (what an
 * RCCL ring all-reduce of the student's gradients, train.py:494, runs beside the step; tools/rccl_standin.py times its completion
 * beside a persistent GEMM launch for different fh_gemm_reserve_cus settings). */
int fh_debug_copy_reduce(float* dst, const float* src, long n, int workgroups, fh_stream_t stream);
/* Per-shape main-loop table used by path 9.  mode 1: every bf16 launch of a shape class not seen yet times its candidate
 * loops (and, for atomic weight-gradient launches, split-K factors) on the stream, BLOCKING, writing / accumulating the
 * output several times -- only for throw-away passes (StudentEngine.autotune).  mode 0: stop measuring; unmeasured shapes
 * run path 2.  mode -1: forget the table.  Returns the number of measured shape classes. */
int fh_gemm_tune(int mode);
/* copy the table out (returns the number of entries; fills up to cap) / merge entries in (returns the table size) */
int fh_gemm_tune_export(unsigned long long* keys, int* paths, int* splitks, int cap);
int fh_gemm_tune_import(const unsigned long long* keys, const int* paths, const int* splitks, int n);
int fh_prof_enable(int on);
int fh_prof_collect(double* ms /*[64]*/, double* flops /*[64]*/, long* launches /*[64]*/);
/* The same log launch by launch (bench.py's per-class table): rows[i] = {M, N, K, groups, X layout, Y layout, loop, atomic, tag},
 * ms[i], flops[i]; returns the number of rows written (<= cap) and clears the log.  fh_prof_tag sets the tag of the launches recorded
 * from then on (returns the previous one). */
int fh_prof_collect_launches(double* ms, double* flops, int* rows, int cap);
int fh_prof_tag(int tag);

/* ------------------------------------------------------------------------------------------
 * Waveform frontend layer 0: Conv1d(1, C0, K, stride, bias=False) + GroupNorm(C0, C0) + GELU.
 * Replaces modules/module.py:65-71 applied at :99-100.  wave [B][S] fp32, y [B][T0][C0] (dtype),
 * T0 = (S-K)/stride + 1.  scratch: fh_conv0_scratch_floats() floats.  mean/rstd: [B][C0] fp32 (saved
 * for backward).  Backward accumulates (+=) into dw [C0][K], dgamma, dbeta (fp32), scaled by grad_scale
 * (feature_grad_mult, modules/model.py:430-431).
 * ------------------------------------------------------------------------------------------ */
long fh_conv0_scratch_floats(int B, int S, int C0, int K, int stride);
int fh_conv0_forward(int dtype, const float* wave, const float* w, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, float* scratch, int B, int S, int C0, int K, int stride, float eps,
                     fh_stream_t stream);
int fh_conv0_backward(int dtype, const float* wave, const float* w, const float* gamma, const float* beta,
                      const void* dy, const float* mean, const float* rstd, float* scratch, float* dw, float* dgamma,
                      float* dbeta, int B, int S, int C0, int K, int stride, float grad_scale, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps as given), optional dropout on the output; saves mean / rstd.
 * Replaces fairseq LayerNorm at modules/model.py:447, modules/module.py:281,569,580.
 * Backward: dy_total = (dy [+ dy2]) [* dropmask(out_drop)]; dx = d/dx; optional dx2 = dx * dropmask(in_drop);
 * dgamma / dbeta accumulate (+=, fp32 atomics).
 * ------------------------------------------------------------------------------------------ */
int fh_layernorm_forward(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                         float* rstd, int M, int C, float eps, fh_drop drop, fh_stream_t stream);
/* y = LayerNorm(T(x + res)): the post-LN residual add + LayerNorm of the frozen teacher's transformer layers (fairseq
 * TransformerSentenceEncoderLayer, layer_norm_first = False; reference teacher via utils/utils.py:51-99).  Forward only. */
int fh_add_layernorm_forward(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                             int M, int C, float eps, fh_stream_t stream);
/* y = gelu(LayerNorm(x)): the conv -> LayerNorm -> GELU block of a layer_norm-mode feature extractor (HuBERT-Large teacher;
 * fairseq ConvFeatureExtractionModel, mode 'layer_norm').  Forward only. */
int fh_layernorm_gelu_forward(int dtype, const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps,
                              fh_stream_t stream);
/* y = gelu(x), exact-erf form (fairseq get_activation_fn('gelu') -> F.gelu; the teacher's fc1 activation, reference teacher via
 * utils/utils.py:51-99).  y may alias x.  n must be a multiple of the 16-byte vector (4 fp32 / 8 bf16).  Forward only. */
int fh_gelu_forward(int dtype, const void* x, void* y, long n, fh_stream_t stream);
int fh_layernorm_backward(int dtype, const void* dy, const void* dy2, const void* x, const float* mean,
                          const float* rstd, const float* gamma, void* dx, void* dx2, float* dgamma, float* dbeta,
                          int M, int C, fh_drop out_drop, fh_drop in_drop, fh_stream_t stream);
/* out[n % mod] += scale * sum_m X[m][n]  (bias gradients) */
int fh_colsum(int dtype, const void* X, float* out, int M, int N, long ldx, int groups, long x_gstride,
              long out_gstride, int mod, float scale, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolutional positional embedding (modules/module.py:186-208, applied :276-278):
 *   w = g * v / ||v||  (weight_norm dim=2);  zc = conv1d(x, w, bias, pad=K/2, groups)[..., :-1];  y = x + gelu(zc)
 * prepare: builds the kernel-layout weights w_fwd / w_bwd ([G][K][32][32] dtype each) and norm (3*K floats scratch).
 * backward_data: dzc = dsum * gelu'(zc) (written), dx = dsum + conv_transpose(dzc).
 * backward_weight: dws [G][K][32][32] fp32 += sum dzc * x ; dbias += colsum(dzc).
 * weightnorm_backward: dg += , dv += from dws.
 * ------------------------------------------------------------------------------------------ */
int fh_posconv_prepare(int dtype, const float* g, const float* v, void* w_fwd, void* w_bwd, float* norm, int D,
                       int groups, int K, fh_stream_t stream);
int fh_posconv_forward(int dtype, const void* x, const void* w_fwd, const float* bias, void* zc, void* y, int B, int T,
                       int D, int groups, int K, fh_stream_t stream);
int fh_posconv_backward_data(int dtype, const void* dsum, const void* zc, const void* w_bwd, void* dzc, void* dx, int B,
                             int T, int D, int groups, int K, fh_stream_t stream);
int fh_posconv_backward_weight(int dtype, const void* dzc, const void* x, float* dws, float* dbias, int B, int T, int D,
                               int groups, int K, fh_stream_t stream);
int fh_posconv_weightnorm_backward(const float* g, const float* v, const float* norm, const float* dws, float* dg,
                                   float* dv, int D, int groups, int K, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention on packed qkv [B][T][3*H*dh] -> out [B][T][H*dh]; lse [B][H][T] (log2 domain).
 * Replaces fairseq MultiheadAttention called at modules/module.py:558-564 (q scaling, key_padding_mask as
 * per-batch valid key count key_len, fp32 softmax, dropout on probabilities).  Backward writes dqkv.
 * Head dim dh <= 64 (multiple of 8): 40 (fithubert.yaml), 64 (ex.yaml student, HuBERT Base / Large teachers).
 * ------------------------------------------------------------------------------------------ */
int fh_attention_forward(int dtype, const void* qkv, void* out, float* lse, const int* key_len, int B, int T, int H,
                         int dh, float scale, fh_drop drop, fh_stream_t stream);
int fh_attention_backward(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                          void* dqkv, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop,
                          fh_stream_t stream);
/* The same with caller-provided scratch (fh_attention_backward_scratch_floats floats; 0 for fp32 or T <= 256): 16-bit launches
 * longer than 256 frames then run the ONE-pass fused backward too -- key blocks of <= 256 keys per workgroup, Q / dO tiles streamed,
 * fp32 dQ partials of the key blocks in the scratch, summed in a fixed order -- instead of the two-pass form (which computes P, the
 * dropout mask and dS twice). */
long fh_attention_backward_scratch_floats(int dtype, int B, int T, int H, int dh);
int fh_attention_backward_ws(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                             void* dqkv, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop, float* scratch,
                             fh_stream_t stream);
/* ... and the gradient of the q | k | v projection bias (modules/module.py:558-564: fairseq's in-projection biases) with it:
 * dbias [3 * H * dh] fp32 += column sums of dqkv over the B * T rows.  The 16-bit fused kernels take them from their fp32 dQ / dK / dV
 * accumulators (no pass over the stored gradient); the other paths add a column-sum pass.  dbias must hold the running sum (zeros for
 * a fresh gradient); the order of the atomic adds is not fixed. */
int fh_attention_backward_bias(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                               void* dqkv, float* dbias, const int* key_len, int B, int T, int H, int dh, float scale, fh_drop drop,
                               float* scratch, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layer-wise KD loss + gradient (train.py:250-293,372-378):  out[l] = mean_{b,t,d} f(pred_l - teacher_{tl[l]}),
 * out[L] = sum_l w[l] * out[l];  dpred_l = grad_scale * w[l] * f'(.) / (B*Tp*P).  loss_type 0 = mse, 1 = l1.
 * pred [L][B][Tp][P] (dtype); teacher [..][B][Tt][P] (teacher_dtype), narrowed to Tp frames (train.py:282).
 * partials: L * fh_kd_loss_nblk() floats; out_losses: L+1 doubles.
 * ------------------------------------------------------------------------------------------ */
int fh_kd_loss_nblk(int dtype, int B, int Tp, int P);
int fh_kd_loss(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
               const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
               long teacher_lstride, int loss_type, float grad_scale, fh_stream_t stream);
/* Same with the cosine term of the DistilHuBERT-style loss (train.py:302-314,372-378; ex.yaml: l1 + cosine):
 *   sim_l = mean_{b,t} -logsigmoid(cos(pred_l[b,t,:], teacher[b,t,:])),
 *   out[l] = mean f + sim_l (what the reference logs per layer), out[L] = rec_weight * sum_l w[l] mean f + sim_weight * sum_l w[l] sim_l,
 *   dpred = d out[L] / d pred.   partials: 2 * L * fh_kd_loss_rows_nblk(B, Tp) floats. */
int fh_kd_loss_rows_nblk(int B, int Tp);
int fh_kd_loss_rows(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
                    const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
                    long teacher_lstride, int loss_type, float rec_weight, float sim_weight, fh_stream_t stream);
/* the same with dpred multiplied by grad_scale (the fp16 mode's loss scale; the reported losses are not scaled) */
int fh_kd_loss_rows_scaled(int dtype, int teacher_dtype, const void* pred, const void* teacher, void* dpred, const float* layer_w,
                           const int* teacher_layer, float* partials, double* out_losses, int L, int B, int Tp, int Tt, int P,
                           long teacher_lstride, int loss_type, float rec_weight, float sim_weight, float grad_scale,
                           fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight preparation and optimizer (train.py:407-421; AMP casts implicit at train.py:469,496).
 * ------------------------------------------------------------------------------------------ */
int fh_cast(int dtype, const float* src, void* dst, long n, fh_stream_t stream);
int fh_permute3_cast(int dtype, const float* src, void* dst, int d0, int d1, int d2, int p0, int p1, int p2,
                     fh_stream_t stream);
int fh_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
             float wd, float bc1, float bc2, float grad_scale, const float* decay_mask, fh_stream_t stream);
/* The optimizer step of the fp16 mode -- torch.cuda.amp.GradScaler.step/update as Lightning's precision=16 drives it
 * (train.py:469,496) -- without a host sync: the flat gradient (which carries the loss scale; grad_scale un-folds it together with
 * 1/world and 1/accumulation) is checked for non-finite values, a step that has one is skipped, and the applied-step count that
 * the schedule (linear warm-up to lr_peak over warmup_steps, linear decay to 0 at total_steps; total_steps = 0: constant) and the
 * Adam bias corrections follow lives in `state` (device memory, zero-initialised by the caller, 32 bytes).  The caller reads
 * `skipped` back whenever convenient and lowers its loss scale. */
typedef struct {
  int found_inf;   /* scratch of the current step */
  int good_steps;  /* optimizer steps applied so far */
  int skipped;     /* steps skipped because of a non-finite gradient */
  int reserved;
  float lr, bc1, bc2, reserved2; /* scalars of the step in flight (written by the library) */
} fh_amp_state;
int fh_adamw_amp(float* p, const float* g, float* m, float* v, long n, float lr_peak, int warmup_steps, int total_steps, float beta1,
                 float beta2, float eps, float wd, float grad_scale, fh_amp_state* state, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Student engine: the whole CustomStudentModel forward (modules/model.py:420-552) and its backward as ONE
 * native plan over a flat fp32 parameter buffer and a caller-provided activation workspace.
 * Parameter names reported by fh_student_param_info are the reference's state-dict keys,
 * so checkpoints are interchangeable.  Stateless: every call recomputes the (cheap) layout.
 * ------------------------------------------------------------------------------------------ */
#define FH_MAX_CONV 12
typedef struct {
  int dtype;                       /* 0 = float, 1 = bf16 activations / MFMA */
  int n_conv;                      /* conv_feature_layers (modules/model.py:267) */
  int conv_dim[FH_MAX_CONV], conv_k[FH_MAX_CONV], conv_stride[FH_MAX_CONV];
  int embed_dim, ffn_dim, heads, layers;  /* encoder_embed_dim, encoder_ffn_embed_dim, encoder_attention_heads, encoder_layers */
  int conv_pos, conv_pos_groups;
  int tr_factor;                   /* time-reduction Conv1d(k = s = tr_reduce_factor), modules/module.py:230-236 (0 = no TR layer) */
  int pred_dim;                    /* pred_head_final_dim */
  float dropout, attention_dropout, activation_dropout, dropout_input;
  float feature_grad_mult;
  /* heads: n_tasks == 0: one LayerWiseProjHead per layer (layerwise_proj, modules/module.py:625-661).
   * n_tasks >= 1: DistilHuBERT-style head on the LAST layer only (modules/model.py:364-368,508-516): Linear(D, inter*n_tasks)
   * -> GELU -> SplitLinear(inter, n_tasks, pred_dim) (modules/module.py:585-619; a plain Linear when n_tasks == 1); "pred" is then
   * [n_tasks][B][T][pred_dim]. */
  int n_tasks, inter_dim;
  /* CNN distillation (train.py:241-246): 1 = cnn_proj_head GELU -> Linear(embed_dim, pred_dim) on the post_extract_proj output
   * (model.py:305-310, present when pred_dim != embed_dim and cnn_loss_weight > 0); the workspace tensor "cnn_pred" holds it */
  int cnn_head;
  /* tr_layer_index (modules/module.py:247-248): the TR conv sits in front of transformer layer tr_index (0 .. layers; == layers:
   * behind the last one).  Layers in front of it run on the T' frames of the conv frontend, the rest on the reduced frames.  With
   * n_tasks >= 1 and a TR layer the DistilHuBERT-style head reads the encoder output through the shared `upsampler`
   * ConvTranspose1d (modules/model.py:341-348,401-407,503-506). */
  int tr_index;
} fh_student_config;

/* per-call shape info */
typedef struct {
  int B, S;        /* batch, samples */
  int training;    /* 0: eval; 1: dropout on, activations saved for the backward; 2: as 1 without the per-layer copies of fc2's raw output
                    * (`f2` = layer_results[i][2] of modules/model.py: API surface that no loss and no backward reads -- the distillation hot path) */
  int heads_on;    /* 1: all layer-wise projection heads (distillation); 0: final_proj only (deployment) */
  uint32_t seed;   /* dropout seed of this step */
} fh_student_call;

long fh_student_param_count(const fh_student_config* c);
/* enumerate parameters: returns 0 and fills the outputs while idx is valid, 1 past the end */
int fh_student_param_info(const fh_student_config* c, int idx, char* name, int name_cap, long* offset, int* ndim,
                          long* dims /*[4]*/);
long fh_student_wc_bytes(const fh_student_config* c);
long fh_student_ws_bytes(const fh_student_config* c, const fh_student_call* k);
/* locate a named activation in the workspace; returns 0 on success */
int fh_student_ws_info(const fh_student_config* c, const fh_student_call* k, const char* name, long* byte_offset,
                       int* ndim, long* dims /*[4]*/, int* elem_bytes);
/* frames after the conv frontend / after time reduction for S samples */
int fh_student_frames(const fh_student_config* c, int S, int* T_conv, int* T_enc);

/* The stream the student's backward issues its parameter gradients on (weight / bias gradients whose only consumer is the optimizer,
 * train.py:407-421): by default a lowest-priority stream of the library; a caller that partitions the chip's CUs between streams
 * (fh_stream_create_cu_range) hands its own here.  NULL restores the library's.  Returns 0, or 1 when side-stream issue is off. */
int fh_student_side_stream(fh_stream_t stream);
/* fp32 master params -> compute-dtype / kernel-layout weights (once per optimizer step) */
int fh_student_prepare(const fh_student_config* c, const float* params, void* wc, fh_stream_t stream);
/* lens: device int [2][B] = valid frames after the frontend, valid frames after time reduction; NULL = no padding */
int fh_student_forward(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc,
                       void* ws, const float* wave, const int* lens, fh_stream_t stream);
/* The training forward with the layer-wise KD loss (train.py:250-293, mse / l1 with per-layer weights) computed INSIDE the merged
 * projection heads' GEMM epilogue: the workspace tensor "pred" receives d loss / d pred (what fh_kd_loss would have written to dpred),
 * out_losses the L per-layer means and the weighted total, and `pred` itself is never written or read back (1.17 GB each way at
 * B = 128).  Exists for the layer-wise heads with time reduction (fithubert.yaml); teacher [Lt][B][Tt][P] as fh_kd_loss takes it, ready on
 * `stream` when the call is made.  partials: fh_student_kd_partials(c, k) floats.  kd == NULL: fh_student_forward. */
typedef struct {
  const void* teacher;
  int teacher_dtype;
  long teacher_lstride;      /* elements between teacher layers (B * Tt * P) */
  const int* teacher_layer;  /* [L] teacher layer of each student layer */
  const float* layer_w;      /* [L] */
  int Tt;                    /* teacher frames per utterance (>= frames of "pred") */
  int loss_type;             /* 0 = mse, 1 = l1 */
  float grad_scale;          /* rec_loss_weight (x the loss scale of the fp16 mode) */
  float* partials;
  double* out_losses;        /* [L + 1] */
} fh_student_kd;
long fh_student_kd_partials(const fh_student_config* c, const fh_student_call* k);
int fh_student_forward_kd(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc, void* ws,
                          const float* wave, const int* lens, const fh_student_kd* kd, fh_stream_t stream);
/* grads += d loss / d params.  d_pred: gradient w.r.t. the workspace tensor "pred" (same shape/dtype) or NULL;
 * d_hidden: gradient w.r.t. "xo" slots 1..L (layer outputs, [L][B*T_enc][D], dtype) or NULL. */
int fh_student_backward(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc,
                        void* ws, float* grads, const float* wave, const int* lens, const void* d_pred,
                        const void* d_hidden, fh_stream_t stream);
/* same, plus d_cnn: gradient w.r.t. results["features"] as the loss sees it -- "cnn_pred" [B][T_conv][pred_dim] with a
 * cnn_proj_head, the pre-dropout "features" [B][T_conv][embed_dim] without -- or NULL */
int fh_student_backward_cnn(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc,
                            void* ws, float* grads, const float* wave, const int* lens, const void* d_pred,
                            const void* d_hidden, const void* d_cnn, fh_stream_t stream);
/* all upstream gradients in one struct; d_qkv_last: gradient w.r.t. the LAST layer's packed q|k|v projection (workspace "qkv"
 * slot L-1, [B*T_enc][3*embed_dim]) -- the entry point of the attention-logit / value-relation transfer losses
 * (utils/utils.py:244-278, train.py:327-370), whose logits are functions of that projection only */
typedef struct {
  const void* d_pred;
  const void* d_hidden;
  const void* d_cnn;
  const void* d_qkv_last;
} fh_student_grads;
int fh_student_backward_ex(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc,
                           void* ws, float* grads, const float* wave, const int* lens, const fh_student_grads* g,
                           fh_stream_t stream);
/* Backward that REPORTS gradient segments as they become final, so that the data-parallel exchange (the reference gets it from
 * DDP's bucketed all-reduce hooks under Lightning strategy='ddp', train.py:494) starts while earlier layers are still in backward.
 * Each time a contiguous range [first, first + count) of the flat gradient buffer has received its last contribution, the library
 * (1) makes `comm_stream` wait -- hipStreamWaitEvent, device side only -- for every kernel that writes the range, on whichever of
 * its streams they were queued, and (2) calls `ready(user, segment, first, count)` synchronously on the calling host thread; the
 * callee enqueues its collective for that range on `comm_stream` (or on a stream ordered behind it).  Segments arrive in completion
 * order: projection heads, transformer layers, then the encoder front (time reduction, positional conv, projection, conv
 * frontend); together the ranges cover the whole buffer exactly once.  n == NULL: same as fh_student_backward_ex. */
typedef void (*fh_grad_ready_fn)(void* user, int segment, long first, long count);
typedef struct {
  fh_stream_t comm_stream;
  fh_grad_ready_fn ready;
  void* user;
} fh_grad_notify;
int fh_student_backward_notify(const fh_student_config* c, const fh_student_call* k, const float* params, const void* wc,
                               void* ws, float* grads, const float* wave, const int* lens, const fh_student_grads* g,
                               const fh_grad_notify* n, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Attention-logit and value-relation transfer, loss fused with its gradient.  Replaces the relation tensors of the reference's
 * patched last layer (utils/utils.py:244-278: attn = (q d^-1/2) k^T with padded keys at -inf, v_rel = (v d^-1/2) v^T, both
 * (B H) x T x T) and their losses (train.py:327-352 attention mse with inf / nan scrubbing or kl-div; train.py:358-367 value-
 * relation kl-div) -- without materialising anything of size T x T.
 *   s_qkv [B][T][3*Ds] (dtype), t_qkv [B][T][3*Dt] (teacher_dtype): the packed q|k|v projections of the two last layers
 *   which : 0 = attention logits (q, k; key masks apply), 1 = value relation (v, v; unmasked, kl-div only)
 *   mode  : 0 = mse over the entries that are finite in the reference, 1 = kl_div(log_softmax(student), softmax(teacher)) summed
 *           over keys, mean over rows.  (Padded batch + kl-div: the reference's arithmetic gives nan; masked keys contribute 0 here.)
 *   s_len / t_len : int32 [B] valid keys of student / teacher (a key is masked when either masks it), or NULL
 *   d_qkv : fp32 [B][T][3*Ds], ACCUMULATES weight * dLoss/d s_qkv (zero it first; several calls may add into it), or NULL
 *   partials : fp32 scratch, fh_relation_kd_nblk(B, T, H) floats (loss partials, then the kl-div form's row log-sum-exps);
 *   loss_out : double[1] <- the (unweighted) loss.  16-bit student AND teacher projections: the logit tiles, and the gradient
 *   products with G as a 16-bit head + remainder pair, run on the matrix cores (two launches: row operand, column operand; no
 *   atomics); an fp32 side selects the fp32 VALU kernel
 * ------------------------------------------------------------------------------------------ */
/* The relation tensor itself, for the module API (the reference's patched layer RETURNS attn / v_rel, utils/utils.py:263-265,278):
 * out != NULL: out[(b h)][q][k] = (A_q . B_k) d^-1/2 fp32, masked keys (k >= key_len[b], which = 0 only) = -inf;
 * grad_out != NULL: its backward, d_qkv (fp32 [B][T][3*D]) += d out / d qkv contracted with grad_out.  Exactly one of the two. */
int fh_relation_tensor(int dtype, const void* qkv, float* out, const float* grad_out, float* d_qkv, const int* key_len, int B,
                       int T, int H, int D, int which, fh_stream_t stream);
int fh_relation_kd_nblk(int B, int T, int H);
int fh_relation_kd(int dtype, int teacher_dtype, const void* s_qkv, const void* t_qkv, float* d_qkv, float* partials,
                   double* loss_out, const int* s_len, const int* t_len, int B, int T, int H, int Ds, int Dt, int which,
                   int mode, float weight, fh_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Frozen-teacher conv frontend on the student's kernels.  The teacher (fairseq HubertModel /
 * Wav2Vec2Model, loaded at utils/utils.py:102-149) is third party; since round 3 its transformer layers run on these kernels too.
 *   fh_conv1d_forward : channels-last Conv1d(Cin, Cout, k, stride) [+ bias] [+ GELU] as ONE MFMA GEMM over
 *                       overlapping windows.  x [B][T][Cin], wf [Cout][k][Cin] (dtype), y [B][Tout][Cout].
 *   fh_posconv_wide_* : weight-normed grouped positional conv with up to 64 channels per group (768 / 16 = 48):
 *                       y = x + gelu(conv1d(x, g*v/||v||, bias, pad = K/2)[..., :-1]).  w_fwd: [G][K][64][64] dtype.
 * ------------------------------------------------------------------------------------------ */
int fh_conv1d_forward(int dtype, const void* x, const void* wf, const float* bias, void* y, int B, int T, int Cin, int Cout,
                      int k, int stride, int gelu, fh_stream_t stream);
/* y = [gelu](x W^T + bias) [+ residual]; `preact` (optional) receives x W^T + bias.  x [M][K], W [N][K] (dtype), bias fp32 [N],
 * residual / preact / y [M][N] (dtype): every nn.Linear of the frozen teacher's transformer layers (fairseq
 * TransformerSentenceEncoderLayer as loaded at utils/utils.py:102-149), the pre-LN form's residual adds folded into the epilogue. */
int fh_linear_forward(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* preact, void* y, int M,
                      int K, int N, int gelu, fh_stream_t stream);
/* First block of a layer_norm-mode conv frontend (modules/module.py:54-64 form; the HuBERT-Large teacher): Conv1d(1, C0, K, stride)
 * + bias -> LayerNorm over the C0 channels -> GELU in one pass, y [B][T0][C0] (dtype); C0 in {256, 512}. */
int fh_conv0_layernorm_gelu_forward(int dtype, const float* wave, const float* w, const float* bias, const float* gamma,
                                    const float* beta, void* y, int B, int S, int C0, int K, int stride, float eps, fh_stream_t stream);
int fh_posconv_wide_prepare(int dtype, const float* g, const float* v, void* w_fwd, float* norm2, int D, int groups, int K,
                            fh_stream_t stream);
int fh_posconv_wide_forward(int dtype, const void* x, const void* w_fwd, const float* bias, void* y, int B, int T, int D,
                            int groups, int K, fh_stream_t stream);

/* The teacher's transformer layers in ONE call (fairseq TransformerSentenceEncoderLayer x n_layers as loaded at
 * utils/utils.py:102-149; the per-layer outputs are what the reference's forward hooks capture, utils/utils.py:61-99).
 *   layers   : host array of n_layers weight tables (device pointers; W* in `dtype`, biases / LayerNorm parameters fp32);
 *              wqkv = rows of q_proj | k_proj | v_proj stacked ([3D][D]), g1 / b1 = self_attn_layer_norm, g2 / b2 = final_layer_norm
 *   pre_ln   : 0 = post-LN (layer_norm_first False: HuBERT / wav2vec 2.0 Base), 1 = pre-LN (HuBERT Large)
 *   x        : [B][T][D] input of layer 0 (after the positional conv and, post-LN, the encoder LayerNorm)
 *   hidden   : [n_layers][B][T][D] <- every layer's output            fc2_out : same shape, every layer's fc2 output (the hook's
 *              second value), or NULL       qkv_last : [B][T][3D] <- the LAST layer's q|k|v projections (utils/utils.py:244-263), or NULL
 *   key_len  : int32 [B] valid frames of a right-padded batch, or NULL     scratch : fh_teacher_encoder_scratch_bytes(...) bytes */
typedef struct fh_teacher_layer {
  const void* wqkv; const float* bqkv;
  const void* wo;   const float* bo;
  const float* g1;  const float* b1;
  const void* w1;   const float* c1;
  const void* w2;   const float* c2;
  const float* g2;  const float* b2;
} fh_teacher_layer;
long fh_teacher_encoder_scratch_bytes(int dtype, int B, int T, int D, int F);
int fh_teacher_encoder_forward(int dtype, const fh_teacher_layer* layers, int n_layers, int pre_ln, const void* x, void* hidden,
                               void* fc2_out, void* qkv_last, const int* key_len, int B, int T, int D, int F, int heads,
                               void* scratch, fh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FITHUBERT_HIP_H */
