// Throw-away microbenchmark: HBM write throughput of a GEMM epilogue's store shape (no compute).
//   mode 0: per wave instruction 8 rows x 128 B (the staged epilogue's shape; wave owns 128 rows x 64 columns of a 256 x 256 tile)
//   mode 1: per wave instruction 2 rows x 512 B (a wave writes whole tile rows: 32 rows per wave)
//   mode 2: linear fill (64 lanes x 16 B contiguous)
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void __launch_bounds__(512) store_kernel(uint16_t* C, int M, int N, int mode) {
  const int tiles_j = (N + 255) / 256;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 v = make_uint4(lane, wave, ti, tj);
  if (mode == 0) {
    const int xi0 = (wave >> 2) * 128, yj0 = (wave & 3) * 64;
    for (int it = 0; it < 16; ++it) {
      const int i = ti * 256 + xi0 + it * 8 + (lane >> 3), j = tj * 256 + yj0 + 8 * (lane & 7);
      if (i < M && j < N) *reinterpret_cast<uint4*>(C + (long)i * N + j) = v;
    }
  } else if (mode == 1) {
    for (int it = 0; it < 16; ++it) {
      const int i = ti * 256 + wave * 32 + it * 2 + (lane >> 5), j = tj * 256 + 8 * (lane & 31);
      if (i < M && j < N) *reinterpret_cast<uint4*>(C + (long)i * N + j) = v;
    }
  } else {
    const long base = ((long)blockIdx.x * 512 + threadIdx.x) * 8;
    const long total = (long)M * N, stride = (long)gridDim.x * 512 * 8;
    for (long o = base; o + 8 <= total; o += stride) *reinterpret_cast<uint4*>(C + o) = v;
  }
}
extern "C" int run_store(void* C, int M, int N, int mode, void* stream) {
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  hipLaunchKernelGGL(store_kernel, dim3(tiles), dim3(512), 0, (hipStream_t)stream, (uint16_t*)C, M, N, mode);
  return (int)hipGetLastError();
}
