// The 16-bit GEMM main loops of gemm_loops.cuh instantiated for f16 (v_mfma_f32_16x16x32_f16 / 32x32x16_f16, fp32 accumulate).
#include "gemm_loops.cuh"

int fh_gemm_path_f16(const GemmParams& p, int path, hipStream_t stream) { return launch_path_h<f16>(p, path, stream); }
