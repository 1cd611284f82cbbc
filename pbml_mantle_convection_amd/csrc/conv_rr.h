// Shapes and filter-bank packing of the row-reuse bf16 convolution kernel (conv_rr_bf16.hip); shared with the batched
// packing kernel in conv_api.hip.
#pragma once
#include "conv_common.h"

constexpr int RR_R = 16, RR_TW = 64, RR_STRIPS = 4;

// fragment types of a row pair.  Slot g of type T is pair q = 4 T + g of the 4K pairs of two rows:
// row offset rho = q / 2K, kx = (q % 2K) / 2, cb = q % 2.  A fragment of type T multiplied by the filter fragment (T, kappa)
// accumulates into output row (pair base row) - kappa; slot g carries filter row ky = kappa + rho (zero outside [0, K)).
template <int K> struct RR {
  static constexpr int PPR = 2 * K;
  static constexpr int NTYPES = (2 * PPR) / 4;
  static_assert((2 * PPR) % 4 == 0, "pairs of two rows fill whole fragments");
  static constexpr int rho(int T, int g) { return (4 * T + g) / PPR; }
  static constexpr int kx(int T, int g) { return ((4 * T + g) % PPR) / 2; }
  static constexpr int cb(int T, int g) { return (4 * T + g) % 2; }
  static constexpr int kmin(int T) { return -rho(T, 3); }
  static constexpr int kmax(int T) { return K - 1 - rho(T, 0); }
  static constexpr int fbase(int T) { return T == 0 ? 0 : fbase(T - 1) + (kmax(T - 1) - kmin(T - 1) + 1); }
  static constexpr int NFRAG = fbase(NTYPES - 1) + (kmax(NTYPES - 1) - kmin(NTYPES - 1) + 1);
  static constexpr int fidx(int T, int kappa) { return fbase(T) + kappa - kmin(T); }
  static constexpr int TIH = RR_R + K - 1, TIW = RR_TW + K - 1;
  static constexpr int PLANE = (TIH * TIW + 15) / 16 * 16;      // 16-byte slots per channel-block plane (== 0 mod 16: see below)
};
static_assert(RR<5>::NFRAG == 26 && RR<3>::NFRAG == 10, "filter fragments per chunk");

// element i of the packed bank [chunk][ntile][frag][lane][8]
template <typename G>
__device__ __forceinline__ float rr_pack_value(const G& g, const float* __restrict__ wu, size_t i, int dgrad, int ntiles) {
  const int K = g.K;
  const int nfrag = K == 5 ? RR<5>::NFRAG : RR<3>::NFRAG;
  const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
  size_t r = i >> 9;
  const int f = (int)(r % nfrag); r /= nfrag;
  const int nt = (int)(r % ntiles);
  const int ck = (int)(r / ntiles);
  // frag index -> (type, kappa)
  int T = 0, kappa = 0;
  if (K == 5) {
#pragma unroll
    for (int t = 0; t < RR<5>::NTYPES; ++t) if (f >= RR<5>::fbase(t)) { T = t; kappa = f - RR<5>::fbase(t) + RR<5>::kmin(t); }
  } else {
#pragma unroll
    for (int t = 0; t < RR<3>::NTYPES; ++t) if (f >= RR<3>::fbase(t)) { T = t; kappa = f - RR<3>::fbase(t) + RR<3>::kmin(t); }
  }
  const int n = lane & 15, gq = lane >> 4, q = 4 * T + gq, ppr = 2 * K;
  const int rho = q / ppr, kx = (q % ppr) / 2, cb = q % 2, ky = kappa + rho;
  if (ky < 0 || ky >= K) return 0.f;
  const int kin = ck * 16 + cb * 8 + e, kout = nt * 16 + n;
  if (!dgrad) return bank_source(g, wu, kout, kin, ky, kx);
  return bank_source(g, wu, kin, kout, K - 1 - ky, K - 1 - kx);
}

