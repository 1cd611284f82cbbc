// f32 convolution path (the parity-gate precision mode): exact-f32 FMA chains on the vector ALU.
// Forward / input-gradient share one LDS-tiled direct kernel (the input gradient is the same
// kernel on the padded domain with the rotated, transposed bank); the filter gradient is a
// tile-local reduction with a deterministic two-stage combine.  The bf16 MFMA path lives in
// conv_bf16.hip and implements the same entry points for dtype == MC_BF16.
#include "conv_common.h"
#include <type_traits>

namespace {

constexpr int TS = 16;  // output tile edge

// ------------------------------------------------------------------------------------------------
// y[n][co][oy][ox] = b[co] + sum_{ci,ky,kx} W[co][ci][ky][kx] * xpad[n][ci][oy+ky][ox+kx]
// block = 16x16 output pixels x 16 output channels; loops over 8-channel input blocks.
// Bank layout (f32): [cbin][tap][ci8][CoutP]; indices are wave-uniform -> scalar loads.
// ------------------------------------------------------------------------------------------------
// FUSE: 0 plain; 1 = sources are raw conv outputs, normalised + activated while staged (exact erf / exp forms);
// 2 = input-gradient epilogue (dz = dA * act'(z) + GroupNorm-backward partial sums), see ConvFuse.
template <int K, int FUSE>
__global__ __launch_bounds__(256) void k_conv_direct_f32(ConvGeom g, const float* __restrict__ x0,
                                                         const float* __restrict__ x1, const float* __restrict__ bank,
                                                         const float* __restrict__ bias, float* __restrict__ y0,
                                                         float* __restrict__ y1, float* __restrict__ part, ConvFuse fz) {
  constexpr int TI = TS + K - 1;
  __shared__ float xs[TI * TI][8];
  __shared__ float red[4][32];
  const int tile = blockIdx.x, cog = blockIdx.y, n = blockIdx.z;
  const int ty0 = (tile / g.tiles_x) * TS, tx0 = (tile % g.tiles_x) * TS;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int oy = ty0 + ty, ox = tx0 + tx;
  const int co0 = cog * 16;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int cb = 0; cb < g.CBin; ++cb) {
    __syncthreads();
    const float* src = cb < g.CB0 ? x0 : x1;
    const int scb = cb < g.CB0 ? cb : cb - g.CB0;
    const int sC8 = cb < g.CB0 ? g.CB0 : g.CB1;
    float psc[8], psh[8];
    int pact = -1;
    if (FUSE == 1) {
      const float* ct = cb < g.CB0 ? fz.coef0 : fz.coef1;
      const int a = cb < g.CB0 ? fz.act0 : fz.act1;
      if (ct != nullptr || a != MC_ACT_NONE) { pact = a; load_coef8(ct, n, sC8 * 8, scb, psc, psh); }
    }
    for (int i = threadIdx.x; i < TI * TI; i += 256) {
      int r = i / TI, c = i % TI;
      int sy = pad_map(ty0 + r - g.pad, g.H, g.pad_mode), sx = pad_map(tx0 + c - g.pad, g.W, g.pad_mode);
      float4 a = make_float4(0, 0, 0, 0), b = a;
      if (sy >= 0 && sx >= 0) {
        const float* p = src + cb8_index(n, scb, sy, sx, sC8, g.H, g.W);
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
        if (FUSE == 1 && pact >= 0) {
          float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          act_fwd8<false>(v, psc, psh, pact, v);
          a = make_float4(v[0], v[1], v[2], v[3]);
          b = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
      *reinterpret_cast<float4*>(&xs[i][0]) = a;
      *reinterpret_cast<float4*>(&xs[i][4]) = b;
    }
    __syncthreads();
    const float* wb = bank + (size_t)cb * K * K * 8 * g.CoutP + co0;
    // The 16 filter values of a (tap, input channel) are consecutive: with constant offsets the compiler merges their scalar
    // loads (s_load_dwordx16).  Only the last channel group of a bank whose CoutP is not a multiple of 16 has 8 columns: that
    // block takes the clamped form, whose unused upper accumulators never read past the end of the bank.  (Round 2 first
    // clamped unconditionally: sixteen separate scalar loads per channel, the f32 path ran at 15 instead of 44 TFLOP/s.)
    const int comax = g.CoutP - 1 - co0;
    auto taps = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float* xv = xs[(ty + ky) * TI + tx + kx];
          float4 a = *reinterpret_cast<const float4*>(xv);
          float4 b = *reinterpret_cast<const float4*>(xv + 4);
          float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          const float* wt = wb + (size_t)(ky * K + kx) * 8 * g.CoutP;
#pragma unroll
          for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = fmaf(v[ci], wt[ci * g.CoutP + (FULL ? co : min(co, comax))], acc[co]);
        }
    };
    if (comax >= 15) taps(std::true_type{});
    else taps(std::false_type{});
  }

  const bool valid = oy < g.Ho && ox < g.Wo;
  float q1[16], q2[16];                                  // per-channel contributions to the tile partials
#pragma unroll
  for (int i = 0; i < 16; ++i) { q1[i] = 0.f; q2[i] = 0.f; }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int cob = cog * 2 + half;
    if (cob >= g.CBout) break;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int co = cob * 8 + j;
      o[j] = (co < g.Cout) ? acc[half * 8 + j] + (bias ? bias[co] : 0.f) : 0.f;
      acc[half * 8 + j] = o[j];
    }
    if (FUSE == 2) {
      // o = dA on the padded domain; final pixels become dz = dA * act'(z) and enter (sum dz, sum dz * yhat)
      const int p = fz.epad, fr = fz.ezero ? 0 : p + 1;
      const int iy = oy - p, ix = ox - p;
      const bool fin = valid && iy >= fr && iy < fz.ehs - fr && ix >= fr && ix < fz.ews - fr;
      if (fin) {
        float yv[8];
        V8<float>::ld(reinterpret_cast<const float*>(fz.ey) + cb8_index(n, cob, iy, ix, g.CBout, fz.ehs, fz.ews), yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float sc = 1.f, sh = 0.f, me = 0.f, rs = 0.f;
          if (fz.ecoef) {
            const float4 c4 = reinterpret_cast<const float4*>(fz.ecoef)[(size_t)n * g.CoutP + cob * 8 + j];
            sc = c4.x; sh = c4.y; me = c4.z; rs = c4.w;
          }
          const float dz = o[j] * act_bwd(yv[j] * sc + sh, fz.eact);
          q1[half * 8 + j] = dz;
          q2[half * 8 + j] = dz * (yv[j] - me) * rs;
          o[j] = dz;
        }
      }
    } else if (valid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { q1[half * 8 + j] = o[j]; q2[half * 8 + j] = o[j] * o[j]; }
    }
    if (valid) {
      if (g.split8 > 0 && cob >= g.split8)
        V8<float>::st(y1 + cb8_index(n, cob - g.split8, oy, ox, g.CBout - g.split8, g.Ho, g.Wo), o);
      else
        V8<float>::st(y0 + cb8_index(n, cob, oy, ox, g.split8 > 0 ? g.split8 : g.CBout, g.Ho, g.Wo), o);
    }
  }
  if (FUSE == 2) part = fz.epart;
  if (part) {
    // per-tile partials per output channel, from the f32 results: (sum, sumsq) of y, or (sum dz, sum dz * yhat)
#pragma unroll
    for (int co = 0; co < 16; ++co) {
      float s = wave_sum(q1[co]), ss = wave_sum(q2[co]);
      if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][co * 2] = s; red[threadIdx.x >> 6][co * 2 + 1] = ss; }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      int co = co0 + (threadIdx.x >> 1);
      if (co < g.CoutP) {
        float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        part[(((size_t)n * (FUSE == 2 ? fz.estride : g.tiles) + tile) * g.CoutP + co) * 2 + (threadIdx.x & 1)] = r;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// filter gradient: block (gx, cbin, cogroup) accumulates over a grid-strided set of (n, tile):
//   dW[co][ci][ky][kx] = sum_{n,oy,ox} dy[n][co][oy][ox] * xpad[n][ci][oy+ky][ox+kx]
// thread t < 8*K*K owns (tap = t/8, ci = t%8) and 16 output channels; threads 8*K*K..+15 own the
// bias gradient of one output channel each (cbin == 0 blocks only).
// partial layout: [G][CoutP][CinP*K*K + 1]  (last column = bias)
// ------------------------------------------------------------------------------------------------
template <int K, bool PRO>
__global__ __launch_bounds__(256) void k_wgrad_direct_f32(ConvGeom g, const float* __restrict__ x0,
                                                          const float* __restrict__ x1, const float* __restrict__ dy,
                                                          float* __restrict__ part, ConvFuse fz) {
  constexpr int TI = TS + K - 1;
  __shared__ float xs[TI * TI][8];
  __shared__ float dys[TS * TS][16];
  const int cb = blockIdx.y, cog = blockIdx.z, co0 = cog * 16;
  const int t = threadIdx.x;
  const int tap = t >> 3, ci = t & 7;
  const bool wthread = t < 8 * K * K;
  const bool bthread = (cb == 0) && t >= 8 * K * K && t < 8 * K * K + 16;
  const int ky = tap / K, kx = tap % K;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bacc = 0.f;
  const float* src = cb < g.CB0 ? x0 : x1;
  const int scb = cb < g.CB0 ? cb : cb - g.CB0;
  const int sC8 = cb < g.CB0 ? g.CB0 : g.CB1;
  const int work = g.N * g.tiles;
  for (int wi = blockIdx.x; wi < work; wi += gridDim.x) {
    const int n = wi / g.tiles, tile = wi % g.tiles;
    const int ty0 = (tile / g.tiles_x) * TS, tx0 = (tile % g.tiles_x) * TS;
    __syncthreads();
    float psc[8], psh[8];
    int pact = -1;
    if (PRO) {
      const float* ct = cb < g.CB0 ? fz.coef0 : fz.coef1;
      const int a = cb < g.CB0 ? fz.act0 : fz.act1;
      if (ct != nullptr || a != MC_ACT_NONE) { pact = a; load_coef8(ct, n, sC8 * 8, scb, psc, psh); }
    }
    for (int i = t; i < TI * TI; i += 256) {
      int r = i / TI, c = i % TI;
      int sy = pad_map(ty0 + r - g.pad, g.H, g.pad_mode), sx = pad_map(tx0 + c - g.pad, g.W, g.pad_mode);
      float4 a = make_float4(0, 0, 0, 0), b = a;
      if (sy >= 0 && sx >= 0) {
        const float* p = src + cb8_index(n, scb, sy, sx, sC8, g.H, g.W);
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
        if (PRO && pact >= 0) {
          float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          act_fwd8<false>(v, psc, psh, pact, v);
          a = make_float4(v[0], v[1], v[2], v[3]);
          b = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
      *reinterpret_cast<float4*>(&xs[i][0]) = a;
      *reinterpret_cast<float4*>(&xs[i][4]) = b;
    }
    {
      int r = t >> 4, c = t & 15;
      int oy = ty0 + r, ox = tx0 + c;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        int cob = cog * 2 + half;
        float4 a = make_float4(0, 0, 0, 0), b = a;
        if (oy < g.Ho && ox < g.Wo && cob < g.CBout) {
          const float* p = dy + cb8_index(n, cob, oy, ox, g.CBout, g.Ho, g.Wo);
          a = *reinterpret_cast<const float4*>(p);
          b = *reinterpret_cast<const float4*>(p + 4);
        }
        *reinterpret_cast<float4*>(&dys[t][half * 8]) = a;
        *reinterpret_cast<float4*>(&dys[t][half * 8 + 4]) = b;
      }
    }
    __syncthreads();
    if (wthread) {
      for (int py = 0; py < TS; ++py)
#pragma unroll 4
        for (int px = 0; px < TS; ++px) {
          float xv = xs[(py + ky) * TI + px + kx][ci];
          const float4* d4 = reinterpret_cast<const float4*>(dys[py * TS + px]);
          float4 d0 = d4[0], d1 = d4[1], d2 = d4[2], d3 = d4[3];
          acc[0] = fmaf(xv, d0.x, acc[0]); acc[1] = fmaf(xv, d0.y, acc[1]); acc[2] = fmaf(xv, d0.z, acc[2]); acc[3] = fmaf(xv, d0.w, acc[3]);
          acc[4] = fmaf(xv, d1.x, acc[4]); acc[5] = fmaf(xv, d1.y, acc[5]); acc[6] = fmaf(xv, d1.z, acc[6]); acc[7] = fmaf(xv, d1.w, acc[7]);
          acc[8] = fmaf(xv, d2.x, acc[8]); acc[9] = fmaf(xv, d2.y, acc[9]); acc[10] = fmaf(xv, d2.z, acc[10]); acc[11] = fmaf(xv, d2.w, acc[11]);
          acc[12] = fmaf(xv, d3.x, acc[12]); acc[13] = fmaf(xv, d3.y, acc[13]); acc[14] = fmaf(xv, d3.z, acc[14]); acc[15] = fmaf(xv, d3.w, acc[15]);
        }
    } else if (bthread) {
      int co = t - 8 * K * K;
      for (int p = 0; p < TS * TS; ++p) bacc += dys[p][co];
    }
  }
  const int nch = wg_chunks(g.CinP);
  float* pb = part + (size_t)blockIdx.x * wg_slab_floats(g.CoutP, g.CinP, K * K);
  if (wthread) {
    int cig = cb * 8 + ci;
#pragma unroll
    for (int co = 0; co < 16; ++co)
      if (co0 + co < g.CoutP) pb[wg_index(tap, cig, co0 + co, g.CoutP, nch)] = acc[co];
  } else if (bthread) {
    int co = co0 + t - 8 * K * K;
    if (co < g.CoutP) pb[(size_t)K * K * nch * g.CoutP * 16 + co] = bacc;
  }
}

}  // namespace

int mc_conv2d_f32(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                  void* y1, float* part, const ConvFuse& fz, int fuse, hipStream_t s) {
  dim3 grid(g.tiles, cdiv(g.CoutP, 16), g.N);
#define CL(K, FU) hipLaunchKernelGGL((k_conv_direct_f32<K, FU>), grid, dim3(256), 0, s, g, (const float*)x0, (const float*)x1, (const float*)bank, bias, (float*)y0, (float*)y1, part, fz)
#define CLF(K) do { if (fuse == 0) CL(K, 0); else if (fuse == 1) CL(K, 1); else CL(K, 2); } while (0)
  if (g.K == 5) CLF(5);
  else if (g.K == 3) CLF(3);
  else return MC_EUNSUPPORTED;
#undef CLF
#undef CL
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_wgrad_f32(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, const ConvFuse& fz, int fuse,
                 hipStream_t s) {
  dim3 grid(g.wgrad_G, g.CBin, cdiv(g.CoutP, 16));
#define WL(K, PRO) hipLaunchKernelGGL((k_wgrad_direct_f32<K, PRO>), grid, dim3(256), 0, s, g, (const float*)x0, (const float*)x1, (const float*)dy, (float*)part, fz)
  if (g.K == 5) { if (fuse) WL(5, true); else WL(5, false); }
  else if (g.K == 3) { if (fuse) WL(3, true); else WL(3, false); }
  else return MC_EUNSUPPORTED;
#undef WL
  MC_CHECK_LAUNCH();
  return MC_OK;
}
