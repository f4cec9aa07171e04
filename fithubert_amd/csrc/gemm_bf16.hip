// The 16-bit GEMM main loops of gemm_loops.cuh instantiated for bf16 (v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, fp32 accumulate).
#include "gemm_loops.cuh"

int fh_gemm_path_bf16(const GemmParams& p, int path, hipStream_t stream) { return launch_path_h<bf16>(p, path, stream); }
