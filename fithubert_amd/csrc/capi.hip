// extern "C" entry points (include/fithubert_hip.h)
#include "gemm.h"

extern "C" int fh_gemm(const fh_gemm_desc* d, int dtype, fh_stream_t stream) {
  if (!d) return -1;
  return fh_gemm_launch(*d, dtype, (hipStream_t)stream);
}
