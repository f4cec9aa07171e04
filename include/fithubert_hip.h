/* fithubert_hip.h -- C ABI of the MI355X-native FitHuBERT distillation hot path.
 *
 * The reference (glory20h/FitHuBERT) is pure Python on torch: it has no native boundary of its
 * own (last row).  The boundary below is therefore defined by this
 * repository; each entry point cites the reference code whose arithmetic it replaces.
 * Plain pointers + sizes + a hipStream_t; no torch types.  All functions are stream-ordered,
 * allocate nothing, never synchronise, and return 0 on success (a hipError_t or a negative
 * argument-check code otherwise).
 *
 * dtype: 0 = float (exact fp32 MFMA, parity mode), 1 = bf16 storage / bf16 MFMA / fp32 accumulate.
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
  int splitk; /* > 1 requires atomic */
  fh_gemm_epilogue ep;
} fh_gemm_desc;

int fh_gemm(const fh_gemm_desc* d, int dtype, fh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FITHUBERT_HIP_H */
