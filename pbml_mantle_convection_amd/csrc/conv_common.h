// Geometry shared by the f32 and bf16 convolution paths.
#pragma once
#include <stdlib.h>
#include "common.h"

struct ConvGeom {
  int N, H, W, Ho, Wo, K, pad, pad_mode;
  int Cin0, Cin1, Cin;      // real channels
  int CB0, CB1, CBin, CinP; // 8-channel blocks of source 0 / 1 / both; padded channel count
  int Cout, CBout, CoutP;
  int split8;               // dgrad: first split8 output blocks go to y0 (0 = no split)
  int tiles_x, tiles_y, tiles;
  int wgrad_G;              // number of partial slabs of the filter-gradient reduction
  int sym_h, U;             // x-mirrored filters / unique filters
  int nh, nv, nq;           // pairs mirrored in x, in y, quadruples mirrored about both axes (symmetry h/2, v/2, hv/4)
  int dtype;
  int out_f32;
};

// Device-side view of mc_conv_prologue / mc_conv_epilogue (include/mantle_hip.h): what a conv launch fuses on its input
// side (GroupNorm affine + activation of the producer, applied while the tile is staged) and, for an input-gradient
// launch, on its output side (dz = dA * act'(z) + the GroupNorm-backward partial sums).
struct ConvFuse {
  const float* coef0; const float* coef1;   // (scale, shift, mean, rstd) tables of source 0 / 1, or NULL
  int act0, act1;                           // MC_ACT_* per source
  const void* ey;                           // epilogue: raw conv output of the producer, CB8 [N][CBout][ehs][ews]
  const float* ecoef;                       // its table or NULL
  float* epart;                             // [N][tiles][CoutP][2]
  int eact, epad, ezero, ehs, ews;          // ezero: forward padding was zeros (every interior pixel is final)
  int estride;                              // slots per sample of epart
  int ey16;                                 // epilogue: ey is f16 (MC_MIX16 forward tensors), else the launch's element type
};
static inline ConvFuse conv_fuse_none() {
  ConvFuse f;
  f.coef0 = f.coef1 = nullptr; f.act0 = f.act1 = MC_ACT_NONE;
  f.ey = nullptr; f.ecoef = nullptr; f.epart = nullptr; f.eact = MC_ACT_NONE; f.epad = 0; f.ezero = 1; f.ehs = f.ews = 0; f.estride = 0;
  f.ey16 = 0;
  return f;
}

// filter-gradient partial slab (one per reduction workgroup): P[tap][16-channel input chunk][co][16] f32 followed by
// the bias gradient [CoutP].  The 16 input channels of a chunk are the fastest axis so that the 16-lane groups of
// the MFMA accumulator layout store 64 contiguous bytes (a [co][ci][tap] layout scattered 4-byte stores 100 B
// apart and cost ~100 us per deep-level layer).
static __host__ __device__ inline int wg_chunks(int CinP) { return (CinP + 15) / 16; }
static __host__ __device__ inline size_t wg_slab_floats(int CoutP, int CinP, int KK) {
  return (size_t)KK * wg_chunks(CinP) * CoutP * 16 + CoutP;
}
static __host__ __device__ inline size_t wg_index(int tap, int cip, int co, int CoutP, int nch) {
  return ((size_t)(tap * nch + (cip >> 4)) * CoutP + co) * 16 + (cip & 15);
}
static inline long wgrad_target_blocks2() {
  static long v = 0;
  if (!v) { const char* e = getenv("MC_WGRAD_BLOCKS2"); v = e ? atol(e) : 512; if (v < 16) v = 16; }
  return v;
}
static inline long wgrad_min_slabs() {
  static long v = 0;
  if (!v) { const char* e = getenv("MC_WGRAD_MIN_G"); v = e ? atol(e) : 16; if (v < 1) v = 1; }
  return v;
}
static inline long wgrad_target_blocks() {
  static long v = 0;
  if (!v) { const char* e = getenv("MC_WGRAD_BLOCKS"); v = e ? atol(e) : 768; if (v < 16) v = 16; }
  return v;
}

// fills g from d; returns MC_OK or an error code.  tile_h/tile_w = output tile of the kernel family.
static inline int conv_geom(const mc_conv_desc* d, int tile_h, int tile_w, ConvGeom& g) {
  if (!d) return MC_EINVAL;
  if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c_in0 <= 0 || d->c_in1 < 0 || d->c_out <= 0) return MC_EINVAL;
  if (d->k != 3 && d->k != 5) return MC_EUNSUPPORTED;
  if (d->pad < 0 || d->pad > d->k - 1) return MC_EINVAL;
  if (d->pad_mode < MC_PAD_ZEROS || d->pad_mode > MC_PAD_REFLECT) return MC_EINVAL;
  if (d->pad_mode == MC_PAD_REFLECT && (d->pad >= d->h || d->pad >= d->w)) return MC_EINVAL;
  if (d->c_in1 > 0 && (d->c_in0 % 8) != 0) return MC_EUNSUPPORTED;
  if (d->sym_h < 0 || (d->sym_h & 1) || d->sym_v < 0 || (d->sym_v & 1) || d->sym_hv < 0 || (d->sym_hv & 3) ||
      d->sym_h + d->sym_v + d->sym_hv > d->c_out)
    return MC_EINVAL;
  if (d->c_out_split != 0 && (d->c_out_split < 0 || d->c_out_split >= d->c_out || (d->c_out_split % 8) != 0))
    return MC_EINVAL;
  g.N = d->n; g.H = d->h; g.W = d->w; g.K = d->k; g.pad = d->pad; g.pad_mode = d->pad_mode;
  g.Ho = d->h + 2 * d->pad - d->k + 1;
  g.Wo = d->w + 2 * d->pad - d->k + 1;
  if (g.Ho <= 0 || g.Wo <= 0) return MC_EINVAL;
  g.Cin0 = d->c_in0; g.Cin1 = d->c_in1; g.Cin = d->c_in0 + d->c_in1;
  g.CB0 = (d->c_in0 + 7) / 8; g.CB1 = (d->c_in1 + 7) / 8; g.CBin = g.CB0 + g.CB1; g.CinP = g.CBin * 8;
  g.Cout = d->c_out; g.CBout = (d->c_out + 7) / 8; g.CoutP = g.CBout * 8;
  g.split8 = d->c_out_split / 8;
  g.tiles_x = (g.Wo + tile_w - 1) / tile_w;
  g.tiles_y = (g.Ho + tile_h - 1) / tile_h;
  g.tiles = g.tiles_x * g.tiles_y;
  g.sym_h = d->sym_h; g.nh = d->sym_h / 2; g.nv = d->sym_v / 2; g.nq = d->sym_hv / 4;
  g.U = d->c_out - g.nh - g.nv - 3 * g.nq;
  g.dtype = d->dtype;
  g.out_f32 = mc_is16(d->dtype) ? d->out_f32 : 0;
  if (g.out_f32 < 0 || g.out_f32 > 1) return MC_EINVAL;
  if (g.out_f32 && (d->c_out > 16 || d->c_out_split != 0)) return MC_EUNSUPPORTED;
  // number of partial slabs of the filter-gradient reduction: enough workgroups to fill the chip
  // (~1024 with the other grid dimensions), bounded by 64 MiB of partials and by the work available
  long slab = (long)wg_slab_floats(g.CoutP, g.CinP, g.K * g.K) * 4;
  long cap = (64L << 20) / slab;
  if (cap < 32) cap = 32;
  long G, work;
  if (mc_is16(g.dtype)) {
    int ntiles = (g.Cout + 15) / 16;
    int ntw = (ntiles % 2 == 0) ? 2 : 1;
    long other = (long)((g.CBin + 1) / 2) * ((ntiles + ntw - 1) / ntw);
    // resident workgroups: 3 per CU with one co-tile per block (39 KB LDS), 2 per CU with two (55 KB); a grid of exactly
    // one resident wave avoids a half-empty second wave
    const long target = ntw == 1 ? wgrad_target_blocks() : wgrad_target_blocks2();
    G = ntw == 1 ? (target + other - 1) / other : target / other;      // two co-tiles: never spill into a second resident wave
    if (G < wgrad_min_slabs()) G = wgrad_min_slabs();
    work = (long)g.N * ((g.Ho + 15) / 16) * ((g.Wo + 31) / 32);     // 16 x 32 pixel work items
  } else {
    G = 1024;
    work = (long)g.N * g.tiles;
  }
  if (G > cap) G = cap;
  if (G > work) G = work;
  g.wgrad_G = (int)G;
  return MC_OK;
}

// padded input-channel index of global input channel ci (concat: source 0 then source 1)
static __host__ __device__ inline int cin_padded_index(int ci, int Cin0, int CB0) {
  return ci < Cin0 ? ci : CB0 * 8 + (ci - Cin0);
}

// ------------------------------------------------------------------------------------------------
// filter-bank element generators shared by the single-layer and the batched pack kernels.
// G is any struct with the fields K, Cout, CBin, CB0, Cin0, Cin1, Cin, U, nh, nv, nq, CBout, CinP, CoutP.
// Output channel co >= U is a mirrored copy of a unique filter (the reference's torch.cat order: x-flips of unique [0, nh),
// y-flips of [nh, nh + nv), then the x-, y- and xy-flips of [nh + nv, nh + nv + nq); symmetric_layers_torch.py:118-136).
template <typename G>
__host__ __device__ __forceinline__ int mirror_source(const G& g, int co, bool& fx, bool& fy) {
  fx = fy = false;
  if (co < g.U) return co;
  int j = co - g.U;
  if (j < g.nh) { fx = true; return j; }
  j -= g.nh;
  if (j < g.nv) { fy = true; return g.nh + j; }
  j -= g.nv;
  const int blk = g.nq > 0 ? j / g.nq : 0;
  fx = blk == 0 || blk == 2; fy = blk == 1 || blk == 2;
  return g.nh + g.nv + (g.nq > 0 ? j % g.nq : 0);
}
// ------------------------------------------------------------------------------------------------
template <typename G>
__device__ __forceinline__ float bank_source(const G& g, const float* __restrict__ wu, int co, int cip, int ky, int kx) {
  // forward filter W_full[co][ci(cip)][ky][kx] read from the unique bank; cip = padded concat channel index
  int ob = cip / 8, oj = cip % 8;
  bool ok = co < g.Cout && ob < g.CBin && (ob < g.CB0 ? (ob * 8 + oj < g.Cin0) : ((ob - g.CB0) * 8 + oj < g.Cin1));
  if (!ok) return 0.f;
  int ci = ob < g.CB0 ? ob * 8 + oj : g.Cin0 + (ob - g.CB0) * 8 + oj;
  const int cw = g.Cin;
  bool fx, fy;
  const int u = mirror_source(g, co, fx, fy);
  const int kxs = fx ? g.K - 1 - kx : kx, kys = fy ? g.K - 1 - ky : ky;
  return wu[(((size_t)u * cw + ci) * g.K + kys) * g.K + kxs];
}

// f32 bank [cbin][tap][ci8][CoutP] (dgrad: [cb over C_out][tap][j][CinP], rotated taps)
template <typename G>
__device__ __forceinline__ float pack_value_f32(const G& g, const float* __restrict__ wu, size_t i, int dgrad) {
  const int K = g.K, KK = K * K;
  const int cop = dgrad ? g.CinP : g.CoutP;
  int o = (int)(i % cop);
  size_t r = i / cop;
  int j = (int)(r % 8); r /= 8;
  int tap = (int)(r % KK);
  int cb = (int)(r / KK);
  if (!dgrad) return bank_source(g, wu, o, cb * 8 + j, tap / K, tap % K);
  return bank_source(g, wu, cb * 8 + j, o, K - 1 - tap / K, K - 1 - tap % K);
}

// bf16 bank [chunk][step][ntile][lane][8]; pair j = 4 step + (lane >> 4): tap = j / 2, cb = j % 2
template <typename G>
__device__ __forceinline__ float pack_value_bf16(const G& g, const float* __restrict__ wu, size_t i, int dgrad, int steps,
                                                 int ntiles) {
  const int K = g.K, KK = K * K;
  int e = (int)(i & 7);
  int lane = (int)((i >> 3) & 63);
  size_t r = i >> 9;
  int nt = (int)(r % ntiles); r /= ntiles;
  int s = (int)(r % steps);
  int ck = (int)(r / steps);
  int n = lane & 15, gq = lane >> 4;
  int j = 4 * s + gq;
  int tap = j / 2, cb = j % 2;
  if (tap >= KK) return 0.f;
  int kin = ck * 16 + cb * 8 + e, kout = nt * 16 + n;
  int ky = tap / K, kx = tap % K;
  if (!dgrad) return bank_source(g, wu, kout, kin, ky, kx);
  return bank_source(g, wu, kin, kout, K - 1 - ky, K - 1 - kx);
}
