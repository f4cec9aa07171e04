// Tap loop of the 16-bit positional-conv kernels (posconv.hip: 32-channel groups; posconv_wide.hip: 48-channel groups as flattened tap
// pairs): Toeplitz order over K = 16 NI taps, weight fragments requested through inline asm with counted waits.
#pragma once
#include <type_traits>
#include "common.cuh"

namespace {

// The compiler sinks an ordinary "next tap" weight load down to its first use (the load of tap j + 1 at the end of iteration j ends
// up at the top of iteration j + 1, `s_waitcnt vmcnt` right behind it), so the weight fragments are requested through inline asm and
// every use sits behind a COUNTED wait the compiler cannot move: vmcnt is one in-order queue, a fragment is there once at most N
// younger requests are outstanding.
template <int N, typename F, int M = 0> __device__ __forceinline__ void static_steps(F&& f) {
  if constexpr (M < N) {
    f(std::integral_constant<int, M>{});
    static_steps<N, F, M + 1>(static_cast<F&&>(f));
  }
}
// The hand-counted protocol trusts the compiler in one respect it is not told about: it treats a fragment as defined at the REQUEST, so
// a copy, rematerialisation or spill it inserted between a request and its wait would read registers the load has not written yet
// (tests/test_abi.py checks the spill side: no scratch in these kernels).  -DFH_POSCONV_PLAIN_LOADS builds the same tap loop on
// ordinary loads the compiler tracks itself (slower: it sinks them to their first use) -- the reference build of
// tests/test_ops_gpu.py::test_positional_conv_asm_requests_against_the_plain_load_build, which compares the two bit for bit
// over many launches under load (fithubert_amd/build.py links it as libfithubert_hip_plainloads.so).
#ifdef FH_POSCONV_PLAIN_LOADS
template <int OFF> __device__ __forceinline__ void wfrag_request(bf16x8& d, const void* sbase, uint32_t voff) {
  d = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(sbase) + voff + OFF);
}
template <int N> __device__ __forceinline__ void wfrag_wait(bf16x8 (&)[2]) {}
template <int N> __device__ __forceinline__ void wfrag_wait(bf16x8 (&)[3]) {}
#else
template <int OFF> __device__ __forceinline__ void wfrag_request(bf16x8& d, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wfrag_wait(bf16x8 (&w)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N));
}
template <int N> __device__ __forceinline__ void wfrag_wait(bf16x8 (&w)[3]) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : "n"(N));
}
#endif
// Toeplitz order.  Output tile tt of a wave multiplies tap j with the window rows [16 tt + j, 16 tt + j + 16): taps j and j + 16 of
// neighbouring tiles read the SAME rows.  So the taps are walked in 16 classes rho = j mod 16; inside a class, step m reads the
// fragment at row 16 m + rho ONCE and feeds it to tap i = m - tt (j = rho + 16 i) of every tile tt -- NT * CB MFMAs per 1 KB LDS read
// where the tap-major order has CB (the tap-major loop is LDS-port bound: per workgroup and tap 16 KB of fragment reads = 128 port
// cycles, two-way bank conflicted, against 128 cycles of MFMA per SIMD).  Tap i is live for steps i .. i + NT - 1; behind its last
// step its registers are refilled with the next class's tap i, i.e. every fragment is requested NI - NT + ... steps (>= 0.5 K cycles)
// before its first use.
// ncls classes; xof(c): the lane's window fragment of class c at step 0 (step m: + m * mstride bytes); wof(c): tap 0 of class c (tap i:
// + i * istride elements; its CB fragments 1 KB apart, fragment order).
template <typename T, int CB, int NT, int NI, typename XOf, typename WOf>
__device__ __forceinline__ void toeplitz_taps(int ncls, XOf&& xof, WOf&& wof, long istride, int mstride, int lane,
                                              f32x4 (&acc)[CB][NT]) {
  const uint32_t voff = lane * 16;
  bf16x8 W[NI][CB];
  // Compiler fences at both ends of the tap loop: a vector-memory access the compiler moved in between the requests and their waits
  // (an epilogue load hoisted above the loop) would be younger than a request without being counted.  Inside, the only memory
  // accesses are the LDS fragment reads (lgkmcnt) and the asm requests.
  asm volatile("" ::: "memory");
  {
    const T* w0 = wof(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const T* wp = w0 + i * istride;
      wfrag_request<0>(W[i][0], wp, voff);
      wfrag_request<1024>(W[i][1], wp, voff);
      if constexpr (CB == 3) wfrag_request<2048>(W[i][2], wp, voff);
    }
  }
  // one tap class; REFILL: request the next class's tap behind every tap's last step.  The last class requests nothing: a request whose
  // result is never used would get registers the compiler hands to something else while the load is still in flight.
  auto tap_class = [&](auto refill, const unsigned char* xr, const T* wnext) {
    constexpr bool REFILL = decltype(refill)::value;
    static_steps<NI + NT - 1>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(xr + m * mstride);
      // requests younger than tap m's: the taps behind it of its own class + this class's refills so far
      if constexpr (m < NI) wfrag_wait<CB * (NI - 1 - m + (REFILL && m - NT + 1 > 0 ? m - NT + 1 : 0))>(W[m]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int i = m - tt;
        if (i < 0 || i >= NI) continue;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb][tt] = mfma16<T>(W[i][cb], x, acc[cb][tt]);
      }
      constexpr int d = m - (NT - 1);
      if constexpr (REFILL && d >= 0 && d < NI) {
        const T* wp = wnext + d * istride;
        wfrag_request<0>(W[d][0], wp, voff);
        wfrag_request<1024>(W[d][1], wp, voff);
        if constexpr (CB == 3) wfrag_request<2048>(W[d][2], wp, voff);
      }
    });
  };
  for (int c = 0; c + 1 < ncls; ++c) tap_class(std::true_type{}, xof(c), wof(c + 1));
  tap_class(std::false_type{}, xof(ncls - 1), (const T*)nullptr);
  asm volatile("" ::: "memory");
}

}  // namespace
