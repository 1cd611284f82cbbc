// Layout conversion, GroupNorm(+activation) forward/backward, pooling, bicubic resampling,
// curl head.  All HBM-bound streaming kernels over the CB8 layout: one thread moves one
// 8-channel vector (32 B f32 / 16 B bf16), consecutive lanes take consecutive pixels.
#include "common.h"

namespace {

// =================================================================================================
// NCHW f32 <-> CB8
// =================================================================================================
// flat index -> (x, y, channel block, sample); 32-bit divisions when the tensor allows it (the three 64-bit div/mod
// pairs were most of the instructions of the boundary-conversion kernels)
__device__ __forceinline__ void split_index(size_t i, bool small, int Wd, int H, int C8, int& xo, int& y, int& cb, int& n) {
  if (small) {
    unsigned r = (unsigned)i;
    xo = (int)(r % (unsigned)Wd); r /= (unsigned)Wd;
    y = (int)(r % (unsigned)H); r /= (unsigned)H;
    cb = (int)(r % (unsigned)C8);
    n = (int)(r / (unsigned)C8);
  } else {
    xo = (int)(i % Wd);
    size_t r = i / Wd;
    y = (int)(r % H); r /= H;
    cb = (int)(r % C8);
    n = (int)(r / C8);
  }
}

template <typename T>
__global__ void k_pack_nchw(const float* __restrict__ x, int N, int C, int SC, int H, int W, int pad_w, int mode,
                            const float* __restrict__ cs, T* __restrict__ out) {
  const int Wp = W + 2 * pad_w, C8 = (C + 7) / 8;
  size_t total = (size_t)N * C8 * H * Wp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int xo, y, cb, n;
    split_index(i, total <= 0xffffffffull, Wp, H, C8, xo, y, cb, n);
    int xs = pad_map(xo - pad_w, W, mode);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = cb * 8 + j;
      v[j] = (c < C && xs >= 0) ? x[(((size_t)n * SC + c) * H + y) * W + xs] * (cs ? cs[c] : 1.f) : 0.f;
    }
    V8<T>::st(out + i * 8, v);
  }
}

template <typename T>
__global__ void k_unpack_nchw(const T* __restrict__ x, int N, int C, int H, int W, int crop,
                              const float* __restrict__ mean_nc, float* __restrict__ out) {
  const int Wo = W - 2 * crop, C8 = (C + 7) / 8;
  size_t total = (size_t)N * C8 * H * Wo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int xo, y, cb, n;
    split_index(i, total <= 0xffffffffull, Wo, H, C8, xo, y, cb, n);
    float v[8];
    V8<T>::ld(x + cb8_index(n, cb, y, xo + crop, C8, H, W), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = cb * 8 + j;
      if (c < C) {
        float m = mean_nc ? mean_nc[n * C + c] : 0.f;
        out[(((size_t)n * C + c) * H + y) * Wo + xo] = v[j] - m;
      }
    }
  }
}

template <typename T>
__global__ void k_pack_grad_nchw(const float* __restrict__ g, int N, int C, int H, int W, int crop,
                                 const float* __restrict__ mean_nc, T* __restrict__ out) {
  const int Wi = W - 2 * crop, C8 = (C + 7) / 8;
  size_t total = (size_t)N * C8 * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int xo, y, cb, n;
    split_index(i, total <= 0xffffffffull, W, H, C8, xo, y, cb, n);
    int xi = xo - crop;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = cb * 8 + j;
      float val = 0.f;
      if (c < C) {
        if (xi >= 0 && xi < Wi) val = g[(((size_t)n * C + c) * H + y) * Wi + xi];
        if (mean_nc) val -= mean_nc[n * C + c];
      }
      v[j] = val;
    }
    V8<T>::st(out + i * 8, v);
  }
}

__global__ void k_zero_f32(float* __restrict__ p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ __launch_bounds__(1024) void k_sum_hw(const float* __restrict__ x, int hw, float scale, float* __restrict__ out) {
  // one block per plane (deterministic: fixed per-thread strides, ordered combine; round 1 added 64 block partials per
  // plane with float atomics)
  const float* p = x + (size_t)blockIdx.x * hw;
  float s = 0.f;
  if ((hw & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {        // planes are 16-byte aligned: 4 pixels per load
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = threadIdx.x; i < hw / 4; i += blockDim.x) {
      const float4 v = p4[i];
      s += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    s = (s + s1) + (s2 + s3);
  } else {
    for (int i = threadIdx.x; i < hw; i += blockDim.x) s += p[i];
  }
  s = wave_sum(s);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    out[blockIdx.x] = t * scale;
  }
}

// =================================================================================================
// GroupNorm statistics from the conv epilogue's per-tile (sum, sumsq) partials
// =================================================================================================
// Per-channel totals of a [count][CP][2] partial array for the cpg channels of one group, one wave per (n, group).
// Lane l owns channel l % cpg and every (64 / cpg)-th partial; the lanes of a channel are then combined with a butterfly
// over the upper lane bits, so all channels of the group are summed at once (the former per-channel loop with two full
// double-precision wave reductions per channel made these tiny kernels ~15 us each, 50 launches per step).
// Requires cpg to be a power of two <= 64; afterwards EVERY lane holds the totals of its channel l % cpg.
__device__ __forceinline__ void group_channel_sums(const float* __restrict__ part, size_t base_n, int count, int CP, int c0,
                                                   int cpg, double& a1, double& a2) {
  const int lane = threadIdx.x & 63, cl = lane % cpg, per = 64 / cpg;
  a1 = 0.0; a2 = 0.0;
  int t = lane / cpg;
  // eight loads in flight (the loop was bound by one load latency per iteration: 30 us for the 1024 slots of a level-0 layer)
  for (; t + 7 * per < count; t += 8 * per) {
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(part + ((base_n + t + k * per) * CP + c0 + cl) * 2);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1 += (double)v[k].x; a2 += (double)v[k].y; }
  }
  for (; t < count; t += per) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((base_n + t) * CP + c0 + cl) * 2);
    a1 += (double)v.x;
    a2 += (double)v.y;
  }
  for (int o = cpg; o < 64; o <<= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
}
__device__ __forceinline__ bool pow2_le64(int v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; }

// (sum, sumsq) partials of a CB8 tensor: block (tile t, channel block cb, image n) sums the rows t, t + tiles, ...
template <typename T>
__global__ __launch_bounds__(256) void k_gn_partials(const T* __restrict__ y, int C8, int H, int W, int tiles,
                                                     float* __restrict__ part) {
  const int t = blockIdx.x, cb = blockIdx.y, n = blockIdx.z;
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  const int nrows = (H - t + tiles - 1) / tiles;
  for (int i = threadIdx.x; i < nrows * W; i += blockDim.x) {
    const int ry = i / W, xx = i - ry * W, yy = t + ry * tiles;
    float v[8];
    V8<T>::ld(y + cb8_index(n, cb, yy, xx, C8, H, W), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[2 * j] += v[j]; s[2 * j + 1] += v[j] * v[j]; }
  }
  __shared__ float red[4][16];
  int idx;
  float r = wave_sum16(s, threadIdx.x & 63, idx);
  if ((threadIdx.x & 3) == 0) red[threadIdx.x >> 6][idx] = r;
  __syncthreads();
  if (threadIdx.x < 16) {
    float tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    part[(((size_t)n * tiles + t) * (C8 * 8) + cb * 8 + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] = tot;
  }
}

// one wave per (n, g); also optional per-(n,c) means (block g handles its own channels) and the (scale, shift, mean,
// rstd) table [n][CP][4] read by consumers that normalise the raw conv output on load (same f32 expressions as gn_coef)
__global__ void k_gn_finalize(const float* __restrict__ part, int tiles, int C, int CP, int groups, int hw,
                              float eps, float* __restrict__ stats, float* __restrict__ chan_mean,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ coef4) {
  const int n = blockIdx.y, g = blockIdx.x;
  const int cpg = C / groups;
  double s = 0.0, ss = 0.0;
  const bool p2 = pow2_le64(cpg);
  if (p2) {
    double cs, css;
    group_channel_sums(part, (size_t)n * tiles, tiles, CP, g * cpg, cpg, cs, css);
    const int lane = threadIdx.x & 63;
    if (chan_mean && lane < cpg) chan_mean[n * C + g * cpg + lane] = (float)(cs / (double)hw);
    s = cs; ss = css;
    for (int o = 1; o < cpg; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  } else {
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      double cs = 0.0, css = 0.0;
      for (int t = threadIdx.x; t < tiles; t += blockDim.x) {
        const float* p = part + (((size_t)n * tiles + t) * CP + c) * 2;
        cs += (double)p[0];
        css += (double)p[1];
      }
      cs = wave_sum_d(cs);
      css = wave_sum_d(css);
      if (chan_mean && threadIdx.x == 0) chan_mean[n * C + c] = (float)(cs / (double)hw);
      s += cs;
      ss += css;
    }
  }
  // every lane holds the group totals
  const double m = (double)cpg * (double)hw;
  const double mean = s / m;
  double var = ss / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, rstdf = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0 && stats) {
    stats[((size_t)n * groups + g) * 2 + 0] = meanf;
    stats[((size_t)n * groups + g) * 2 + 1] = rstdf;
  }
  if (coef4) {
    for (int cl = threadIdx.x; cl < cpg; cl += blockDim.x) {
      const int c = g * cpg + cl;
      const float ga = gamma[c], be = beta[c];
      reinterpret_cast<float4*>(coef4)[(size_t)n * CP + c] = make_float4(rstdf * ga, be - meanf * rstdf * ga, meanf, rstdf);
    }
  }
}

// =================================================================================================
// a = act(GN(y)) [+ AvgPool(POOL)(a)]
// =================================================================================================
constexpr int GN_ROWS = 8;
// launch-shape knobs (environment overrides are for tuning runs only)
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int gn_fwd_vpt() { static int v = env_int("MC_GN_FWD_VPT", 8); return v; }
// rows per block of the backward-apply kernel (tools/bench_gn.py sweep on MI355X): 8 rows for wide images, more for narrow
// ones so that a block still streams >= 16 vectors per thread
// threads per block of the one-launch GroupNorm kernels (one block per (sample, channel block)): at 73-88 registers a
// 1024-thread block fills a CU alone, so a grid of more blocks than CUs (level 4: 16 channel blocks x 32 samples) ran in two
// rounds; 512-thread blocks sit two to a CU and finish in one (each thread then walks two pixels instead of one)
static int gn_small_threads(int blocks) {
  static int v = env_int("MC_GN_SMALL_THREADS", 0);
  if (v >= 512) return v;                       // (k_gn_act_small needs one wave per group of a channel block: >= 8 waves)
  return blocks > 256 ? 512 : 1024;             // in-step A/B, 1024 / auto / 512 / 256: 9.71 / 9.67 / 9.67 / 9.76 ms
}
static int gn_apply_rows(int h, int w) { static int v = env_int("MC_GN_ROWS", 0); if (v > 0) return v; return w > 256 ? GN_ROWS : (w > 32 ? 16 : 32); }

struct GnArgs {
  int N, C, C8, H, W, groups, cpg, post, act;
  const float* stats;
  const float* gamma;
  const float* beta;
};

__device__ __forceinline__ void gn_coef(const GnArgs& a, int n, int cb, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int c = cb * 8 + j;
    if (a.post == MC_POST_GN_ACT && c < a.C) {
      int g = c / a.cpg;
      float mean = a.stats[((size_t)n * a.groups + g) * 2], rstd = a.stats[((size_t)n * a.groups + g) * 2 + 1];
      float ga = a.gamma[c], be = a.beta[c];
      sc[j] = rstd * ga;
      sh[j] = be - mean * rstd * ga;
    } else {
      sc[j] = (c < a.C) ? 1.f : 0.f;
      sh[j] = 0.f;
    }
  }
}

template <typename T, int POOL>
__global__ void k_gn_act_fwd(GnArgs a, const T* __restrict__ y, T* __restrict__ out, T* __restrict__ pooled) {
  // one thread per POOLxPOOL pixel block of one channel block
  const int Hb = (a.H + POOL - 1) / POOL, Wb = (a.W + POOL - 1) / POOL;
  const int Hp = a.H / POOL, Wp = a.W / POOL;
  const int n = (int)blockIdx.z, cb = blockIdx.y;
  float sc[8], sh[8];
  gn_coef(a, n, cb, sc, sh);
  const int act = a.post == MC_POST_NONE ? MC_ACT_NONE : a.act;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hb * Wb; i += gridDim.x * blockDim.x) {
    int by = i / Wb, bx = i % Wb;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
      for (int dx = 0; dx < POOL; ++dx) {
        int yy = by * POOL + dy, xx = bx * POOL + dx;
        if (yy < a.H && xx < a.W) {
          float v[8];
          size_t idx = cb8_index(n, cb, yy, xx, a.C8, a.H, a.W);
          V8<T>::ld(y + idx, v);
          act_fwd8<FastMath<T>::value>(v, sc, sh, act, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[j];
          if (out) V8<T>::st(out + idx, v);
        }
      }
    if (POOL > 1 && by < Hp && bx < Wp) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= 1.0f / (POOL * POOL);
      V8<T>::st(pooled + cb8_index(n, cb, by, bx, a.C8, Hp, Wp), acc);
    }
  }
}

// a = act(GN(y)) and AvgPool2d(2)(a) with COALESCED rows: a thread handles the pixels (2r, x) and (2r + 1, x), so consecutive
// lanes read and write consecutive 16-byte vectors, and takes the horizontal neighbour's sums from the adjacent lane (needs an
// even width; the one-thread-per-2x2-block form above touches every second vector of a row per instruction: 182 us instead
// of ~125 us at 506 x 512).  Pooling uses the f32 activation values, like the block form.
template <typename T>
__global__ void k_gn_act_fwd_pool2_rows(GnArgs a, const T* __restrict__ y, T* __restrict__ out, T* __restrict__ pooled) {
  const int n = (int)blockIdx.z, cb = blockIdx.y;
  float sc[8], sh[8];
  gn_coef(a, n, cb, sc, sh);
  const int act = a.post == MC_POST_NONE ? MC_ACT_NONE : a.act;
  const size_t base = ((size_t)n * a.C8 + cb) * a.H * a.W * 8;
  const int Hh = (a.H + 1) / 2, Hp = a.H / 2, Wp = a.W / 2;
  const int total = Hh * a.W, span = gridDim.x * blockDim.x;             // (span and W are even: lanes 2k, 2k + 1 share a row)
  for (int i0 = blockIdx.x * blockDim.x; i0 < total; i0 += span) {
    const int i = i0 + threadIdx.x;
    const bool live = i < total;
    const int r = live ? i / a.W : 0, x = live ? i - r * a.W : 0;
    float s8[8], v0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      const size_t i0e = base + ((size_t)(2 * r) * a.W + x) * 8;
      V8<T>::ld(y + i0e, v0);
      act_fwd8<FastMath<T>::value>(v0, sc, sh, act, v0);
      if (out) V8<T>::st(out + i0e, v0);
      if (2 * r + 1 < a.H) {
        V8<T>::ld(y + i0e + (size_t)a.W * 8, v1);
        act_fwd8<FastMath<T>::value>(v1, sc, sh, act, v1);
        if (out) V8<T>::st(out + i0e + (size_t)a.W * 8, v1);
      }
    }
    // ((a00 + a01) + a10) + a11: the summation order of the block form, of k_gn_act_small and of k_avgpool (bit-identical
    // pooled tensors whichever kernel a layer takes)
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] = 0.25f * (((v0[j] + __shfl_xor(v0[j], 1, 64)) + v1[j]) + __shfl_xor(v1[j], 1, 64));
    if (live && (x & 1) == 0 && r < Hp && (x >> 1) < Wp)
      V8<T>::st(pooled + (((size_t)n * a.C8 + cb) * Hp * Wp + (size_t)r * Wp + (x >> 1)) * 8, s8);
  }
}

// Small layers: GroupNorm statistics from the conv's partial sums AND a = act(GN(y)) [+ AvgPool] in one launch, one block per
// (sample, channel block); needs every group inside one channel block (channels per group 1, 2, 4 or 8).  The (mean, rstd)
// pairs are also written out for the backward pass.  Same f64 sums and f32 expressions as k_gn_finalize / gn_coef.
template <typename T, int POOL>
__global__ __launch_bounds__(1024) void k_gn_act_small(GnArgs a, const T* __restrict__ y, const float* __restrict__ part, int tiles,
                                                       float eps, float* __restrict__ stats_out, T* __restrict__ out,
                                                       T* __restrict__ pooled) {
  const int n = (int)blockIdx.y, cb = blockIdx.x, CP = a.C8 * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float s_mean[8], s_rstd[8];
  const int ng = 8 / a.cpg;                                   // groups of this channel block (some may lie beyond C)
  if (wave < ng) {
    const int c0 = cb * 8 + wave * a.cpg;
    if (c0 < a.C) {
      double cs, css;
      group_channel_sums(part, (size_t)n * tiles, tiles, CP, c0, a.cpg, cs, css);
      for (int o = 1; o < a.cpg; o <<= 1) { cs += __shfl_xor(cs, o, 64); css += __shfl_xor(css, o, 64); }
      const double m = (double)a.cpg * (double)a.H * (double)a.W;
      const double mean = cs / m;
      double var = css / m - mean * mean;
      if (var < 0.0) var = 0.0;
      const float meanf = (float)mean, rstdf = (float)(1.0 / sqrt(var + (double)eps));
      if (lane == 0) {
        s_mean[wave] = meanf; s_rstd[wave] = rstdf;
        const int g = c0 / a.cpg;
        stats_out[((size_t)n * a.groups + g) * 2] = meanf;
        stats_out[((size_t)n * a.groups + g) * 2 + 1] = rstdf;
      }
    }
  }
  __syncthreads();
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j;
    sc[j] = 0.f; sh[j] = 0.f;
    if (c < a.C) {
      const float mean = s_mean[j / a.cpg], rstd = s_rstd[j / a.cpg], ga = a.gamma[c], be = a.beta[c];
      sc[j] = rstd * ga;
      sh[j] = be - mean * rstd * ga;
    }
  }
  const int Hb = (a.H + POOL - 1) / POOL, Wb = (a.W + POOL - 1) / POOL;
  const int Hp = a.H / POOL, Wp = a.W / POOL;
  for (int i = threadIdx.x; i < Hb * Wb; i += blockDim.x) {
    const int by = i / Wb, bx = i - by * Wb;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
      for (int dx = 0; dx < POOL; ++dx) {
        const int yy = by * POOL + dy, xx = bx * POOL + dx;
        if (yy < a.H && xx < a.W) {
          float v[8];
          const size_t idx = cb8_index(n, cb, yy, xx, a.C8, a.H, a.W);
          V8<T>::ld(y + idx, v);
          act_fwd8<FastMath<T>::value>(v, sc, sh, a.act, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[j];
          V8<T>::st(out + idx, v);
        }
      }
    if (POOL > 1 && by < Hp && bx < Wp) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= 1.0f / (POOL * POOL);
      V8<T>::st(pooled + cb8_index(n, cb, by, bx, a.C8, Hp, Wp), acc);
    }
  }
}

template <typename T>
__global__ void k_avgpool(const T* __restrict__ x, int C8, int H, int W, int f, T* __restrict__ out) {
  const int Hp = H / f, Wp = W / f;
  const int n = blockIdx.z, cb = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hp * Wp; i += gridDim.x * blockDim.x) {
    int by = i / Wp, bx = i % Wp;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        float v[8];
        V8<T>::ld(x + cb8_index(n, cb, by * f + dy, bx * f + dx, C8, H, W), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= 1.0f / (float)(f * f);
    V8<T>::st(out + cb8_index(n, cb, by, bx, C8, Hp, Wp), acc);
  }
}

// =================================================================================================
// backward of act(GN(y))
// =================================================================================================
// phase 1: partial[n][blk][c][2] = (sum dz, sum dz * yhat) over this block's pixels
// GK: compile-time (source 0, source 1) kinds: 0 = decided at run time, 1 = (PADFOLD, NONE), 2 = (PLAIN, NONE),
// 3 = (PADFOLD, PADFOLD_POOL)
template <int GK> struct GKinds {
  static constexpr int k0 = GK == 1 ? MC_GSRC_PADFOLD : (GK == 2 ? MC_GSRC_PLAIN : (GK == 3 ? MC_GSRC_PADFOLD : -1));
  static constexpr int k1 = (GK == 1 || GK == 2) ? MC_GSRC_NONE : (GK == 3 ? MC_GSRC_PADFOLD_POOL : -1);
};
static int gkind_of(const mc_grad_src& g0, const mc_grad_src& g1) {
  if (g0.c8_total > 0 || g1.c8_total > 0) return 0;        // slices of a concatenated tensor: generic path
  if (g0.kind == MC_GSRC_PADFOLD && g1.kind == MC_GSRC_NONE) return 1;
  if (g0.kind == MC_GSRC_PLAIN && g1.kind == MC_GSRC_NONE) return 2;
  if (g0.kind == MC_GSRC_PADFOLD && g1.kind == MC_GSRC_PADFOLD_POOL) return 3;
  return 0;
}

// TY: storage type of y (MC_MIX16: y is f16 while the gradient tensors T are bf16)
template <typename T, int GK, typename TY = T>
__global__ __launch_bounds__(256, 6) void k_gn_bwd_reduce(GnArgs a, const TY* __restrict__ y, mc_grad_src g0,
                                                       mc_grad_src g1, float* __restrict__ part, int CP) {
  const int n = (int)blockIdx.z, cb = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  float sc[8], sh[8], mean[8], rstd[8];
  gn_coef(a, n, cb, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int c = cb * 8 + j;
    if (c < a.C) {
      int g = c / a.cpg;
      mean[j] = a.stats[((size_t)n * a.groups + g) * 2];
      rstd[j] = a.stats[((size_t)n * a.groups + g) * 2 + 1];
    } else { mean[j] = 0.f; rstd[j] = 0.f; }
  }
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // rows blk, blk + nblk, ... of this block, flattened with the columns so that narrow images (W < 256) still use every
  // thread (a per-row loop left 7/8 of the threads idle at the deep levels: ~25 us floor per launch)
  const int nrows = (a.H - blk + nblk - 1) / nblk;
#ifndef MC_GN_RED_UNROLL
#define MC_GN_RED_UNROLL 1   /* A/B on MI355X: 2 = no gain (4.1 TB/s either way), 4 spills (3x slower) */
#endif
  // MC_GN_RED_UNROLL positions per iteration, every load issued before the first use: the loop is bound by the bytes it
  // keeps in flight (two 16-byte loads per position), not by arithmetic
  const int total = nrows * a.W;
  for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * MC_GN_RED_UNROLL) {
    float v[MC_GN_RED_UNROLL][8], da[MC_GN_RED_UNROLL][8];
#pragma unroll
    for (int u = 0; u < MC_GN_RED_UNROLL; ++u) {
      const int i = min(i0 + u * (int)blockDim.x, total - 1);
      const int ry = i / a.W, xx = i - ry * a.W, yy = blk + ry * nblk;
#pragma unroll
      for (int j = 0; j < 8; ++j) da[u][j] = 0.f;
      V8<TY>::ld(y + cb8_index(n, cb, yy, xx, a.C8, a.H, a.W), v[u]);
      grad_fetch_add<T, GKinds<GK>::k0>(g0, n, cb, yy, xx, a.C8, da[u]);
      grad_fetch_add<T, GKinds<GK>::k1>(g1, n, cb, yy, xx, a.C8, da[u]);
    }
#pragma unroll
    for (int u = 0; u < MC_GN_RED_UNROLL; ++u) {
      const float live = (i0 + u * (int)blockDim.x < total) ? 1.f : 0.f;     // the clamped tail position contributes nothing
      float ga[8];
      act_bwd8<FastMath<T>::value>(v[u], sc, sh, a.act, ga);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dz = live * da[u][j] * ga[j];
        s1[j] += dz;
        s2[j] += dz * (v[u][j] - mean[j]) * rstd[j];
      }
    }
  }
  __shared__ float red[4][16];
  {
    float s[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[2 * j] = s1[j]; s[2 * j + 1] = s2[j]; }
    int idx;
    float r = wave_sum16(s, threadIdx.x & 63, idx);
    if ((threadIdx.x & 3) == 0) red[threadIdx.x >> 6][idx] = r;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    int c = cb * 8 + (threadIdx.x >> 1);
    part[(((size_t)n * nblk + blk) * CP + c) * 2 + (threadIdx.x & 1)] = r;
  }
}

// phase 2: one block per group g: m12[n][g] = (sum_c gamma_c s1, sum_c gamma_c s2) / M for every sample, and
// dgamma[c] += sum_n s2[n][c], dbeta[c] += sum_n s1[n][c].  DETERMINISTIC: the 16 waves of the block take the samples
// round-robin, park their per-(sample, channel) sums in LDS, and one thread per (channel, kind) adds them in sample order
// (round 1 used N float atomics per channel: the training step was not reproducible from run to run).
constexpr int GNF_CHUNK = 256;          // samples per LDS round
__global__ __launch_bounds__(1024) void k_gn_bwd_finalize(const float* __restrict__ part, int N, int blocks, int C, int CP,
                                                          int groups, int hw, const float* __restrict__ gamma,
                                                          float* __restrict__ m12, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  const int g = blockIdx.x, cpg = C / groups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  __shared__ float sm[GNF_CHUNK][16];                      // [sample][2 * channel-in-group + (0: s1, 1: s2)], cpg <= 8
  const double M = (double)cpg * (double)hw;
  if (pow2_le64(cpg) && cpg <= 8) {
    float tot = 0.f;                                        // thread t < 2 cpg: running sum of value t over the samples
    for (int n0 = 0; n0 < N; n0 += GNF_CHUNK) {
      const int nn = min(GNF_CHUNK, N - n0);
      for (int k = wave; k < nn; k += nwaves) {
        const int n = n0 + k;
        double a1, a2;
        group_channel_sums(part, (size_t)n * blocks, blocks, CP, g * cpg, cpg, a1, a2);
        const int c = g * cpg + (lane % cpg);
        if (lane < cpg) { sm[k][2 * lane] = (float)a1; sm[k][2 * lane + 1] = (float)a2; }
        const float ga = gamma ? gamma[c] : 1.f;
        double m1 = ga * a1, m2 = ga * a2;
        for (int o = 1; o < cpg; o <<= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
        if (lane == 0 && m12) {
          m12[((size_t)n * groups + g) * 2 + 0] = (float)(m1 / M);
          m12[((size_t)n * groups + g) * 2 + 1] = (float)(m2 / M);
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < 2 * cpg)
        for (int k = 0; k < nn; ++k) tot += sm[k][threadIdx.x];
      __syncthreads();
    }
    if ((int)threadIdx.x < 2 * cpg) {
      const int c = g * cpg + (threadIdx.x >> 1);
      if (threadIdx.x & 1) { if (dgamma) dgamma[c] += tot; }
      else if (dbeta) dbeta[c] += tot;
    }
    return;
  }
  // general group width: one wave walks the samples and the channels in order
  if (wave != 0) return;
  for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
    float tg = 0.f, tb = 0.f;
    for (int n = 0; n < N; ++n) {
      double a1 = 0.0, a2 = 0.0;
      for (int t = lane; t < blocks; t += 64) {
        const float* p = part + (((size_t)n * blocks + t) * CP + c) * 2;
        a1 += (double)p[0];
        a2 += (double)p[1];
      }
      a1 = wave_sum_d(a1);
      a2 = wave_sum_d(a2);
      tb += (float)a1;
      tg += (float)a2;
    }
    if (lane == 0) {
      if (dgamma) dgamma[c] += tg;
      if (dbeta) dbeta[c] += tb;
    }
  }
  if (m12) {
    for (int n = 0; n < N; ++n) {
      double m1 = 0.0, m2 = 0.0;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        double a1 = 0.0, a2 = 0.0;
        for (int t = lane; t < blocks; t += 64) {
          const float* p = part + (((size_t)n * blocks + t) * CP + c) * 2;
          a1 += (double)p[0];
          a2 += (double)p[1];
        }
        a1 = wave_sum_d(a1);
        a2 = wave_sum_d(a2);
        const float ga = gamma ? gamma[c] : 1.f;
        m1 += ga * a1;
        m2 += ga * a2;
      }
      if (lane == 0) {
        m12[((size_t)n * groups + g) * 2 + 0] = (float)(m1 / M);
        m12[((size_t)n * groups + g) * 2 + 1] = (float)(m2 / M);
      }
    }
  }
}

// phase 2 for partial tables with MANY slots per sample (the input-gradient epilogue writes one per (tile, strip): 1100+ at
// 506 x 512): one block per (group, sample) -- N x groups blocks instead of `groups` (the one-block form above walks the
// samples with 16 waves: 62 us for a level-0 layer, latency-bound) -- writes m12[n][g] and the per-(sample, channel) sums
// pc[n][CP][2] = (sum dz, sum dz yhat); dgamma / dbeta are accumulated from pc in sample order by k_gn_param_grads.
// The four waves take a quarter of the slots each and are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_gn_bwd_finalize_n(const float* __restrict__ part, int blocks, int C, int CP, int groups, int hw,
                                                           const float* __restrict__ gamma, float* __restrict__ m12,
                                                           float* __restrict__ pc) {
  const int g = blockIdx.x, n = blockIdx.y, cpg = C / groups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double sm[4][16];
  const int per = (blocks + 3) / 4, b0 = wave * per, cnt = max(0, min(per, blocks - b0));
  double a1, a2;
  group_channel_sums(part, (size_t)n * blocks + b0, cnt, CP, g * cpg, cpg, a1, a2);      // cpg is a power of two <= 8 (host check)
  if (lane < cpg) { sm[wave][2 * lane] = a1; sm[wave][2 * lane + 1] = a2; }
  __syncthreads();
  if (threadIdx.x < 2 * cpg) {
    const double t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    pc[((size_t)n * CP + g * cpg + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] = (float)t;
    sm[0][threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < 2 && m12) {
    double m = 0.0;
    for (int k = 0; k < cpg; ++k) m += (double)(gamma ? gamma[g * cpg + k] : 1.f) * sm[0][2 * k + threadIdx.x];
    m12[((size_t)n * groups + g) * 2 + threadIdx.x] = (float)(m / ((double)cpg * (double)hw));
  }
}

// phase 3: dy = rstd (dz gamma - m1 - yhat m2)   (GN)   |   dy = da act'(y)   (act only)
template <typename T, int GK = 0, typename TY = T>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(GnArgs a, const TY* __restrict__ y, const float* __restrict__ m12,
                                                      mc_grad_src g0, mc_grad_src g1, T* __restrict__ dy, int rows_pb) {
  const int n = (int)blockIdx.z, cb = blockIdx.y;
  float sc[8], sh[8], mean[8], rstd[8], ga[8], m1[8], m2[8];
  gn_coef(a, n, cb, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int c = cb * 8 + j;
    mean[j] = 0.f; rstd[j] = 0.f; ga[j] = 0.f; m1[j] = 0.f; m2[j] = 0.f;
    if (c < a.C) {
      if (a.post == MC_POST_GN_ACT) {
        int g = c / a.cpg;
        mean[j] = a.stats[((size_t)n * a.groups + g) * 2];
        rstd[j] = a.stats[((size_t)n * a.groups + g) * 2 + 1];
        ga[j] = a.gamma[c];
        m1[j] = m12[((size_t)n * a.groups + g) * 2];
        m2[j] = m12[((size_t)n * a.groups + g) * 2 + 1];
      } else { ga[j] = 1.f; rstd[j] = 1.f; }
    }
  }
  float cA[8], cB[8], cC[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cA[j] = rstd[j] * ga[j];                          // (act-only layers: rstd = ga = 1, m1 = m2 = 0)
    cB[j] = -rstd[j] * rstd[j] * m2[j];
    cC[j] = -rstd[j] * m1[j];
  }
  // each block streams GN_ROWS rows: the per-(n, channel-block) coefficient prologue is amortised over several
  // vectors per thread (one vector per thread left these kernels latency-bound at ~1.3 TB/s).  (A 4-way manual batching
  // of the loads was tried and was SLOWER: 131 VGPRs cut the occupancy of this streaming kernel.)
  const int y0 = blockIdx.x * rows_pb, nrows = min(rows_pb, a.H - y0);
  for (int i = threadIdx.x; i < nrows * a.W; i += blockDim.x) {      // rows x columns flattened (narrow images)
    const int ry = i / a.W, xx = i - ry * a.W, yy = y0 + ry;
    float v[8], da[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
    size_t idx = cb8_index(n, cb, yy, xx, a.C8, a.H, a.W);
    V8<TY>::ld(y + idx, v);
    grad_fetch_add<T, GKinds<GK>::k0>(g0, n, cb, yy, xx, a.C8, da);
    grad_fetch_add<T, GKinds<GK>::k1>(g1, n, cb, yy, xx, a.C8, da);
    float gz[8];
    act_bwd8<FastMath<T>::value>(v, sc, sh, a.act, gz);
    // dy = rstd (gamma dz - m1 - yhat m2) = cA dz + cB (y - mean) + cC  (three packed FMAs per channel pair)
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 dz = (f32x2){da[j], da[j + 1]} * (f32x2){gz[j], gz[j + 1]};
      const f32x2 t = (f32x2){v[j], v[j + 1]} - (f32x2){mean[j], mean[j + 1]};
      const f32x2 r = pk_fma((f32x2){cA[j], cA[j + 1]}, dz, pk_fma((f32x2){cB[j], cB[j + 1]}, t, (f32x2){cC[j], cC[j + 1]}));
      o[j] = r.x; o[j + 1] = r.y;
    }
    V8<T>::st(dy + idx, o);
  }
}

// Small layers (levels >= 2 of the U-Net: <= 128 x 128 pixels): the three phases above in ONE launch.  One block per (sample,
// channel block) walks its 8 channels x H x W twice -- sums, block reduction in a fixed order, coefficients, dy -- the second
// walk is served by the caches.  The separate reduce / finalize / apply launches of such a layer take 40-50 us for < 1 MB of
// data: they are bound by launch latency, not by bytes.  Needs every group inside one channel block (channels per group
// 1, 2, 4 or 8).  The per-(sample, channel) sums go to `pc` [N][CP][2]; k_gn_param_grads adds them to dgamma / dbeta in sample
// order for all such layers at the end of the backward pass.
template <typename T, int GK, typename TY = T>
__global__ __launch_bounds__(1024) void k_gn_bwd_small(GnArgs a, const TY* __restrict__ y, mc_grad_src g0, mc_grad_src g1,
                                                       T* __restrict__ dy, float* __restrict__ pc, int CP) {
  const int n = (int)blockIdx.y, cb = blockIdx.x;
  float sc[8], sh[8], mean[8], rstd[8], ga[8];
  gn_coef(a, n, cb, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j;
    mean[j] = 0.f; rstd[j] = 0.f; ga[j] = 0.f;
    if (c < a.C) {
      const int g = c / a.cpg;
      mean[j] = a.stats[((size_t)n * a.groups + g) * 2];
      rstd[j] = a.stats[((size_t)n * a.groups + g) * 2 + 1];
      ga[j] = a.gamma[c];
    }
  }
  const int total = a.H * a.W;
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int yy = i / a.W, xx = i - yy * a.W;
    float v[8], da[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gz[8];
    V8<TY>::ld(y + cb8_index(n, cb, yy, xx, a.C8, a.H, a.W), v);
    grad_fetch_add<T, GKinds<GK>::k0>(g0, n, cb, yy, xx, a.C8, da);
    grad_fetch_add<T, GKinds<GK>::k1>(g1, n, cb, yy, xx, a.C8, da);
    act_bwd8<FastMath<T>::value>(v, sc, sh, a.act, gz);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = da[j] * gz[j];
      s[2 * j] += dz;
      s[2 * j + 1] += dz * (v[j] - mean[j]) * rstd[j];
    }
  }
  __shared__ float red[16][16];
  __shared__ float tot[16];
  {
    int idx;
    const float r = wave_sum16(s, threadIdx.x & 63, idx);
    if ((threadIdx.x & 3) == 0) red[threadIdx.x >> 6][idx] = r;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float r = 0.f;
    const int nw = (int)blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) r += red[w][threadIdx.x];          // fixed order: deterministic
    tot[threadIdx.x] = r;
    const int c = cb * 8 + (threadIdx.x >> 1);
    if (c < CP) pc[((size_t)n * CP + c) * 2 + (threadIdx.x & 1)] = r;
  }
  __syncthreads();
  // m1 = sum_c gamma_c s1_c / M, m2 = sum_c gamma_c s2_c / M over the channels of the group (all inside this channel block)
  float cA[8], cB[8], cC[8];
  const float invM = 1.0f / ((float)a.cpg * (float)total);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j;
    float m1 = 0.f, m2 = 0.f;
    if (c < a.C) {
      const int j0 = (j / a.cpg) * a.cpg;
      for (int k = 0; k < a.cpg; ++k) {
        const float gk = a.gamma[cb * 8 + j0 + k];
        m1 += gk * tot[2 * (j0 + k)];
        m2 += gk * tot[2 * (j0 + k) + 1];
      }
      m1 *= invM; m2 *= invM;
    }
    cA[j] = rstd[j] * ga[j];
    cB[j] = -rstd[j] * rstd[j] * m2;
    cC[j] = -rstd[j] * m1;
  }
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int yy = i / a.W, xx = i - yy * a.W;
    float v[8], da[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gz[8], o[8];
    const size_t idx = cb8_index(n, cb, yy, xx, a.C8, a.H, a.W);
    V8<TY>::ld(y + idx, v);
    grad_fetch_add<T, GKinds<GK>::k0>(g0, n, cb, yy, xx, a.C8, da);
    grad_fetch_add<T, GKinds<GK>::k1>(g1, n, cb, yy, xx, a.C8, da);
    act_bwd8<FastMath<T>::value>(v, sc, sh, a.act, gz);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 dz = (f32x2){da[j], da[j + 1]} * (f32x2){gz[j], gz[j + 1]};
      const f32x2 t = (f32x2){v[j], v[j + 1]} - (f32x2){mean[j], mean[j + 1]};
      const f32x2 r = pk_fma((f32x2){cA[j], cA[j + 1]}, dz, pk_fma((f32x2){cB[j], cB[j + 1]}, t, (f32x2){cC[j], cC[j + 1]}));
      o[j] = r.x; o[j + 1] = r.y;
    }
    V8<T>::st(dy + idx, o);
  }
}

// dgamma[c] += sum_n pc[n][c][1], dbeta[c] += sum_n pc[n][c][0], samples in order, for up to GP_MAX layers per launch
constexpr int GP_MAX = 32;
struct GpJob { const float* pc; float* dgamma; float* dbeta; int N, C, CP, first; };
struct GpTable { int n; GpJob j[GP_MAX]; };
__global__ void k_gn_param_grads(GpTable t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int ji = 0;
#pragma unroll 1
  for (int k = 1; k < t.n; ++k) if (i >= t.j[k].first) ji = k;
  const GpJob& j = t.j[ji];
  const int c = i - j.first;
  if (c >= j.C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int n = 0; n < j.N; ++n) { s1 += j.pc[((size_t)n * j.CP + c) * 2]; s2 += j.pc[((size_t)n * j.CP + c) * 2 + 1]; }
  if (j.dgamma) j.dgamma[c] += s2;
  if (j.dbeta) j.dbeta[c] += s1;
}

// in-place adjoint of F.pad on a padded-domain gradient: one thread per border TARGET pixel (a pixel of the
// interior frame of thickness p+1) gathers its halo sources; sources are halo positions only, so no hazards.
// (buf1 / C8a: a second buffer of the same H x W -- the two input-gradient outputs of a convolution over concatenated
// sources -- folded by the same launch: channel blocks >= C8a belong to buf1, which has C8 - C8a of them)
template <typename T>
__global__ void k_fold_padded(T* __restrict__ buf, int C8, int H, int W, int p, int mode, int all, T* __restrict__ buf1, int C8a) {
  const int n = blockIdx.z;
  int cb = blockIdx.y;
  if (buf1 && cb >= C8a) { buf = buf1; cb -= C8a; C8 -= C8a; } else if (buf1) { C8 = C8a; }
  const int t = p + 1;
  const int band = all ? H * W : 2 * t * W;         // top + bottom bands (all columns); tiny images: every pixel
  const int side = all ? 0 : (H - 2 * t) * 2 * t;   // left + right bands of the remaining rows
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < band + side; i += gridDim.x * blockDim.x) {
    int yy, xx;
    if (all) {
      yy = i / W;
      xx = i - yy * W;
    } else if (i < band) {
      int r = i / W;
      xx = i - r * W;
      yy = r < t ? r : H - 2 * t + r;
    } else {
      int k = i - band;
      int r = k / (2 * t), c = k - r * 2 * t;
      yy = t + r;
      xx = c < t ? c : W - 2 * t + c;
    }
    int cy[6], cx[6];
    int ny = fold_candidates(yy, H, p, mode, cy), nx = fold_candidates(xx, W, p, mode, cx);
    if (ny == 1 && nx == 1) continue;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        float v[8];
        V8<T>::ld(buf + cb8_index(n, cb, cy[a], cx[b], C8, Hp, Wp), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    V8<T>::st(buf + cb8_index(n, cb, yy + p, xx + p, C8, Hp, Wp), acc);
  }
}

// The same padding adjoint for a gradient buffer whose non-frame pixels already hold dz = dA * act'(z) (written by the
// input-gradient kernel's epilogue): EVERY frame pixel (within p + 1 of the border; `all`: every pixel) gets its halo
// sources added, is turned into dz in place and contributes to this block's (sum dz, sum dz * yhat) partials
// part[n][stride][CP][2] at slot first + blockIdx.x.  With zero padding there is nothing to fold and no frame.
template <typename T, typename TY = T>
__global__ __launch_bounds__(256) void k_fold_padded_dz(T* __restrict__ buf, int C8, int H, int W, int p, int mode, int all,
                                                        const TY* __restrict__ y, const float* __restrict__ coef4, int act,
                                                        float* __restrict__ part, int stride, int first) {
  const int n = blockIdx.z, cb = blockIdx.y;
  const int t = p + 1;
  const int band = all ? H * W : 2 * t * W;
  const int side = all ? 0 : (H - 2 * t) * 2 * t;
  const int Hp = H + 2 * p, Wp = W + 2 * p, CP = C8 * 8;
  float sc[8], sh[8], me[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (coef4) {
      const float4 c4 = reinterpret_cast<const float4*>(coef4)[(size_t)n * CP + cb * 8 + j];
      sc[j] = c4.x; sh[j] = c4.y; me[j] = c4.z; rs[j] = c4.w;
    } else { sc[j] = 1.f; sh[j] = 0.f; me[j] = 0.f; rs[j] = 0.f; }
  }
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one pixel per thread (the launch covers band + side)
  if (i < band + side) {
    int yy, xx;
    if (all) {
      yy = i / W;
      xx = i - yy * W;
    } else if (i < band) {
      int r = i / W;
      xx = i - r * W;
      yy = r < t ? r : H - 2 * t + r;
    } else {
      int k = i - band;
      int r = k / (2 * t), c = k - r * 2 * t;
      yy = t + r;
      xx = c < t ? c : W - 2 * t + c;
    }
    int cy[6], cx[6];
    const int ny = fold_candidates(yy, H, p, mode, cy), nx = fold_candidates(xx, W, p, mode, cx);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        float v[8];
        V8<T>::ld(buf + cb8_index(n, cb, cy[a], cx[b], C8, Hp, Wp), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    float yv[8], gz[8];
    V8<TY>::ld(y + cb8_index(n, cb, yy, xx, C8, H, W), yv);
    act_bwd8<FastMath<T>::value>(yv, sc, sh, act, gz);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = acc[j] * gz[j];
      acc[j] = dz;
      s[2 * j] = dz;
      s[2 * j + 1] = dz * (yv[j] - me[j]) * rs[j];
    }
    V8<T>::st(buf + cb8_index(n, cb, yy + p, xx + p, C8, Hp, Wp), acc);
  }
  __shared__ float red[4][16];
  int idx;
  float r = wave_sum16(s, threadIdx.x & 63, idx);
  if ((threadIdx.x & 3) == 0) red[threadIdx.x >> 6][idx] = r;
  __syncthreads();
  if (threadIdx.x < 16) {
    float tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    part[(((size_t)n * stride + first + blockIdx.x) * CP + cb * 8 + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] = tot;
  }
}

// GroupNorm backward, last phase, from dz (see mc_gn_bwd_apply_dz): dy = scale dz - rstd (m1 + yhat m2); coef4 == NULL
// (activation-only layer): dy = dz.  Rows x columns flattened like k_gn_bwd_apply.
template <typename T, int GK, typename TY = T>
__global__ __launch_bounds__(256) void k_gn_bwd_apply_dz(mc_grad_src gs, const TY* __restrict__ y, int C, int C8, int H, int W,
                                                         int groups, const float* __restrict__ coef4,
                                                         const float* __restrict__ m12, T* __restrict__ dy, int rows_pb) {
  const int n = blockIdx.z, cb = blockIdx.y;
  float cA[8], cB[8], cC[8], me[8];
  const int cpg = C / groups;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j;
    cA[j] = (c < C) ? 1.f : 0.f; cB[j] = 0.f; cC[j] = 0.f; me[j] = 0.f;
    if (coef4 && c < C) {
      const float4 c4 = reinterpret_cast<const float4*>(coef4)[(size_t)n * C8 * 8 + c];
      const int g = c / cpg;
      const float m1 = m12[((size_t)n * groups + g) * 2], m2 = m12[((size_t)n * groups + g) * 2 + 1];
      cA[j] = c4.x;                     // rstd * gamma
      cB[j] = -c4.w * c4.w * m2;
      cC[j] = -c4.w * m1;
      me[j] = c4.z;
    }
  }
  const int y0 = blockIdx.x * rows_pb, nrows = min(rows_pb, H - y0);
  for (int i = threadIdx.x; i < nrows * W; i += blockDim.x) {
    const int ry = i / W, xx = i - ry * W, yy = y0 + ry;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dz[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
    const size_t idx = cb8_index(n, cb, yy, xx, C8, H, W);
    if (coef4) V8<TY>::ld(y + idx, v);
    grad_fetch_add<T, GK>(gs, n, cb, yy, xx, C8, dz);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 t = (f32x2){v[j], v[j + 1]} - (f32x2){me[j], me[j + 1]};
      const f32x2 r = pk_fma((f32x2){cA[j], cA[j + 1]}, (f32x2){dz[j], dz[j + 1]},
                             pk_fma((f32x2){cB[j], cB[j + 1]}, t, (f32x2){cC[j], cC[j + 1]}));
      o[j] = r.x; o[j + 1] = r.y;
    }
    V8<T>::st(dy + idx, o);
  }
}

// =================================================================================================
// bicubic resampling with host-built tap tables
// =================================================================================================
// forward: separable.  One block = a 16 x 64 tile of output pixels of one (image, channel block): the input window
// the tile's taps reference (<= 12 x 36 pixels at scale 2) is staged in LDS, filtered along x into a planar f32
// intermediate, then along y.  6.75 vector FMAs and ~0.4 global loads per output instead of 16 and 16.  Outputs whose
// taps leave the window (tables that are not the clamped, monotone bicubic ones) take the direct 16-tap gather.
constexpr int FOH = 16, FOW = 64, FWH = 12, FWW = 36;
template <typename T>
__device__ __forceinline__ void bicubic_direct(const T* __restrict__ x, int n, int cb, int C8, int Hi, int Wi, int yo, int xo,
                                               const int* __restrict__ iy, const float* __restrict__ wy,
                                               const int* __restrict__ ix, const float* __restrict__ wx, float (&acc)[8],
                                               const float* __restrict__ coef4, int act) {
  float sc[8], sh[8];
  if (act >= 0) load_coef8(coef4, n, C8 * 8, cb, sc, sh);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int ys = iy[yo * 4 + a];
    float wa = wy[yo * 4 + a];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v[8];
      V8<T>::ld(x + cb8_index(n, cb, ys, ix[xo * 4 + b], C8, Hi, Wi), v);
      if (act >= 0) {
        act_fwd8<FastMath<T>::value>(v, sc, sh, act, v);
        // the staged window holds the activation in the storage type
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = round_storage<T>(v[j]);
      }
      float w = wa * wx[xo * 4 + b];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
    }
  }
}

// one CB8 vector held as raw registers (Q = 1: 8 bf16, Q = 2: 8 f32) -> act(scale * v + shift) in the storage type
template <typename T> struct XformRaw;
template <> struct XformRaw<bf16_t> {
  static __device__ __forceinline__ void apply(uint4 (&r)[1], const float (&sc)[8], const float (&sh)[8], int act) {
    r[0] = xform_bf16x8(r[0], sc, sh, act);
  }
};
template <> struct XformRaw<f16_t> {
  static __device__ __forceinline__ void apply(uint4 (&r)[1], const float (&sc)[8], const float (&sh)[8], int act) {
    r[0] = xform_f16x8(r[0], sc, sh, act);
  }
};
template <> struct XformRaw<float> {
  static __device__ __forceinline__ void apply(uint4 (&r)[2], const float (&sc)[8], const float (&sh)[8], int act) {
    float v[8] = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), __uint_as_float(r[0].w),
                  __uint_as_float(r[1].x), __uint_as_float(r[1].y), __uint_as_float(r[1].z), __uint_as_float(r[1].w)};
    act_fwd8<false>(v, sc, sh, act, v);
    r[0] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    r[1] = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
  }
};

// coef4 / act: x is a raw conv output; the producer's GroupNorm affine + activation are applied while the window is staged
// (act < 0: x is used as it is)
template <typename T>
__global__ __launch_bounds__(256) void k_bicubic_fwd(const T* __restrict__ x, int C8, int Hi, int Wi, int Ho, int Wo,
                                                     const int* __restrict__ iy, const float* __restrict__ wy,
                                                     const int* __restrict__ ix, const float* __restrict__ wx,
                                                     T* __restrict__ out, int tiles_x, const float* __restrict__ coef4,
                                                     int act) {
  __shared__ __attribute__((aligned(16))) T win[FWH * FWW * 8];
  __shared__ float4 tmp[2][FWH][FOW];
  const int n = blockIdx.z, cb = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * FOH, tx0 = (blockIdx.x % tiles_x) * FOW;
  const int ylo = iy[ty0 * 4], xlo = ix[tx0 * 4];            // clamped bicubic tables are non-decreasing
  {
    constexpr int VPT = (FWH * FWW + 255) / 256, Q = sizeof(T) == 4 ? 2 : 1;
    uint4 rv[VPT][Q];
#pragma unroll
    for (int m = 0; m < VPT; ++m) {
      int i = min((int)threadIdx.x + m * 256, FWH * FWW - 1);
      int r = i / FWW, c = i - r * FWW;
      const T* p = x + cb8_index(n, cb, min(ylo + r, Hi - 1), min(xlo + c, Wi - 1), C8, Hi, Wi);
#pragma unroll
      for (int q = 0; q < Q; ++q) rv[m][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p) + 16 * q);
    }
    if (act >= 0) {
      float sc[8], sh[8];
      load_coef8(coef4, n, C8 * 8, cb, sc, sh);
#pragma unroll
      for (int m = 0; m < VPT; ++m) XformRaw<T>::apply(rv[m], sc, sh, act);
    }
#pragma unroll
    for (int m = 0; m < VPT; ++m) {
      int i = threadIdx.x + m * 256;
      if (i < FWH * FWW)
#pragma unroll
        for (int q = 0; q < Q; ++q) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(&win[i * 8]) + 16 * q) = rv[m][q];
    }
  }
  const int lx = threadIdx.x & 63, xo = min(tx0 + lx, Wo - 1);
  int jx[4];
  float fx[4];
  bool xin = true;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    jx[b] = ix[xo * 4 + b] - xlo;
    fx[b] = wx[xo * 4 + b];
    xin = xin && jx[b] >= 0 && jx[b] < FWW;
    jx[b] = min(max(jx[b], 0), FWW - 1);
  }
  __syncthreads();
  // x pass
#pragma unroll
  for (int m = 0; m < FWH / 4; ++m) {
    int r = (threadIdx.x >> 6) + 4 * m;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v[8];
      V8<T>::ld(&win[(r * FWW + jx[b]) * 8], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += fx[b] * v[j];
    }
    tmp[0][r][lx] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    tmp[1][r][lx] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  // y pass
#pragma unroll
  for (int k = 0; k < FOH / 4; ++k) {
    int yo = ty0 + (threadIdx.x >> 6) + 4 * k;
    if (yo >= Ho || tx0 + lx >= Wo) continue;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool fast = xin;
    int jy[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      jy[a] = iy[yo * 4 + a] - ylo;
      fast = fast && jy[a] >= 0 && jy[a] < FWH;
    }
    if (fast) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float wa = wy[yo * 4 + a];
        float4 p0 = tmp[0][jy[a]][lx], p1 = tmp[1][jy[a]][lx];
        acc[0] += wa * p0.x; acc[1] += wa * p0.y; acc[2] += wa * p0.z; acc[3] += wa * p0.w;
        acc[4] += wa * p1.x; acc[5] += wa * p1.y; acc[6] += wa * p1.z; acc[7] += wa * p1.w;
      }
    } else {
      bicubic_direct<T>(x, n, cb, C8, Hi, Wi, yo, xo, iy, wy, ix, wx, acc, coef4, act);
    }
    V8<T>::st(out + cb8_index(n, cb, yo, xo, C8, Ho, Wo), acc);
  }
}

// The upsample as a walk down the output rows (the forward twin of k_bicubic_bwd_walk below): thread = output column of a
// strip, block = strip x chunk of output rows of one (sample, channel block).  An input row crosses LDS once (storage type,
// activated on the way if asked), every thread interpolates it along x into registers (f32), and a register window of the
// four x-interpolated rows under the current output row slides with the first tap; an output row is 4 x 8 FMAs and one
// coalesced 16-byte store per thread.  Same summation order as the tiled kernel (x taps, then y taps) on unclamped rows.
template <typename T, int SW>
__global__ __launch_bounds__(SW) void k_bicubic_fwd_walk(const T* __restrict__ x, int C8, int Hi, int Wi, int Ho, int Wo,
                                                         const int* __restrict__ iy, const float* __restrict__ wy,
                                                         const int* __restrict__ ix, const float* __restrict__ wx,
                                                         T* __restrict__ out, int rpc, int fw, int strips,
                                                         const float* __restrict__ coef4, int act) {
  constexpr int Q = sizeof(T) == 4 ? 2 : 1;
  __shared__ uint4 crow[2][Q][SW];
  const int n = blockIdx.z, cb = blockIdx.y, c = threadIdx.x;
  const int strip = blockIdx.x % strips, chunk = blockIdx.x / strips;
  const int r0 = chunk * rpc, r1 = min(Ho, r0 + rpc);
  const int F0 = strip * fw, F1 = min(Wo, F0 + fw);
  if (r0 >= r1 || F0 >= F1) return;                                     // (block-uniform)
  const int C0 = ix[F0 * 4];                                            // clamped tables are non-decreasing
  const int ncw = min(ix[(F1 - 1) * 4 + 3] - C0 + 1, SW);
  const bool outok = F0 + c < F1;
  const int xo = min(F0 + c, F1 - 1);
  int jx[4];
  float fx[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    jx[b] = min(max(ix[xo * 4 + b] - C0, 0), SW - 1);
    fx[b] = wx[xo * 4 + b];
  }
  float sc[8], sh[8];
  if (act >= 0) load_coef8(coef4, n, C8 * 8, cb, sc, sh);
  const T* src = x + cb8_index(n, cb, 0, min(C0 + c, Wi - 1), C8, Hi, Wi);
  const size_t rstride = (size_t)Wi * 8;
  auto ldrow = [&](int yi, uint4 (&r)[Q]) {
    if (c < ncw) {
      const char* p = reinterpret_cast<const char*>(src + (size_t)min(yi, Hi - 1) * rstride);
#pragma unroll
      for (int q = 0; q < Q; ++q) r[q] = *reinterpret_cast<const uint4*>(p + 16 * q);
    }
  };
  float win[4][8];
  int buf = 0;
  // input row (registers) -> LDS -> this thread's x interpolation, into the top of the window
  auto push = [&](uint4 (&r)[Q]) {
    if (c < ncw) {
      if (act >= 0) XformRaw<T>::apply(r, sc, sh, act);
#pragma unroll
      for (int q = 0; q < Q; ++q) crow[buf][q][c] = r[q];
    }
    __syncthreads();                       // (one barrier per input row: the buffer written two rows on was read before the next one)
#pragma unroll
    for (int j = 0; j < 8; ++j) { win[0][j] = win[1][j]; win[1][j] = win[2][j]; win[2][j] = win[3][j]; win[3][j] = 0.f; }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v[8];
      if constexpr (sizeof(T) == 4) {
        const uint4 lo = crow[buf][0][jx[b]], hi = crow[buf][Q - 1][jx[b]];
        v[0] = __uint_as_float(lo.x); v[1] = __uint_as_float(lo.y); v[2] = __uint_as_float(lo.z); v[3] = __uint_as_float(lo.w);
        v[4] = __uint_as_float(hi.x); v[5] = __uint_as_float(hi.y); v[6] = __uint_as_float(hi.z); v[7] = __uint_as_float(hi.w);
      } else {
        V8<T>::ld(reinterpret_cast<const T*>(&crow[buf][0][jx[b]]), v);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) win[3][j] += fx[b] * v[j];
    }
    buf ^= 1;
  };
  int basey = __builtin_amdgcn_readfirstlane(iy[r0 * 4]);               // window = input rows basey .. basey + 3 (clamped)
  uint4 nx[Q];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ldrow(basey + k, nx);
    push(nx);
  }
  ldrow(basey + 4, nx);                                                  // the next row: in flight until the window slides
  for (int yo = r0; yo < r1; ++yo) {
    const int t0 = __builtin_amdgcn_readfirstlane(iy[yo * 4]), t1 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 1]),
              t2 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 2]), t3 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 3]);
    const float w0 = wy[yo * 4], w1 = wy[yo * 4 + 1], w2 = wy[yo * 4 + 2], w3 = wy[yo * 4 + 3];
    while (basey < t0) {
      push(nx);
      ++basey;
      ldrow(basey + 4, nx);
    }
    float o[8];
    if (t1 == basey + 1 && t2 == basey + 2 && t3 == basey + 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = w0 * win[0][j];
        a += w1 * win[1][j]; a += w2 * win[2][j]; a += w3 * win[3][j];
        o[j] = a;
      }
    } else {                                                             // clamped taps: several read one row
      const int q1 = t1 - basey, q2 = t2 - basey, q3 = t3 - basey;
      float wk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wk[k] = (k == 0 ? w0 : 0.f) + (q1 == k ? w1 : 0.f) + (q2 == k ? w2 : 0.f) + (q3 == k ? w3 : 0.f);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = wk[0] * win[0][j] + wk[1] * win[1][j] + wk[2] * win[2][j] + wk[3] * win[3][j];
    }
    if (outok) V8<T>::st(out + cb8_index(n, cb, yo, xo, C8, Ho, Wo), o);
  }
}

// adjoint of the bicubic upsample, separable.  One block = a 16 x 16 tile of input (low-res) pixels of one channel block:
// the (folded) output-gradient window the tile touches (<= 40 x 40) is staged in LDS in the storage type, reduced along
// y with the transposed tap lists (lanes run over window columns: conflict-free), then along x from a planar f32
// intermediate.  Pixels whose lists leave the window (image borders of large scale factors) gather from global memory.
constexpr int BT = 16, BWIN = 40;
// BYT / BXT: tap-list lengths held on chip (8 covers every list of an even x2 upsample, 12 the odd sizes; longer lists take
// the per-pixel gather path).  With 12 slots for 8-entry lists a third of both passes multiplied zeros (331 us at level 0).
template <typename T, int BYT, int BXT>
__global__ __launch_bounds__(256) void k_bicubic_bwd(mc_grad_src g, int C8, int Hi, int Wi, const int* __restrict__ tys,
                                                     const int* __restrict__ tyj, const float* __restrict__ tyw,
                                                     const int* __restrict__ txs, const int* __restrict__ txj,
                                                     const float* __restrict__ txw, T* __restrict__ dx, int tiles_x, int tiles) {
  __shared__ __attribute__((aligned(16))) T win[BWIN * BWIN * 8];
  __shared__ float4 tmp[2][BT][BWIN];
  __shared__ int s_yj[BT][BYT];
  __shared__ float s_yw[BT][BYT];
  const int n = blockIdx.z, cb = blockIdx.y;
  constexpr int VPT = (BWIN * BWIN + 255) / 256, Q = sizeof(T) == 4 ? 2 : 1;
  const int pad = g.kind == MC_GSRC_PLAIN ? 0 : g.pad;
  const int hs = g.hs + 2 * pad, ws = g.ws + 2 * pad;
  const T* base = reinterpret_cast<const T*>(g.ptr);
  // Round 3: blocks are PERSISTENT over the tiles of their (sample, channel block) and the raw window of tile t + 1 is in
  // flight (registers) while tile t runs its two passes: one block per tile exposed the load latency once per tile with only
  // three blocks resident per CU (45 KB of LDS each).
  uint4 rv[VPT][Q];
  auto window_origin = [&](int tile, int& ty0, int& tx0, int& ylo, int& xlo) {
    ty0 = (tile / tiles_x) * BT; tx0 = (tile % tiles_x) * BT;
    // output ranges referenced by this tile: the transposed tap lists are sorted by output index and monotone in the
    // input index, so the range starts at the first entry of the first row's list (uniform scalar loads)
    ylo = tyj[tys[ty0]]; xlo = txj[txs[tx0]];
  };
  auto load_window = [&](int ylo, int xlo) {
#pragma unroll
    for (int m = 0; m < VPT; ++m) {
      int i = threadIdx.x + m * 256;
      int r = i / BWIN, c = i - r * BWIN;
      bool ok = i < BWIN * BWIN && ylo + r < g.hs && xlo + c < g.ws;
      const T* p = base + cb8_index(n, cb + (g.c8_total > 0 ? g.cb_off : 0), min(ylo + r, g.hs - 1) + pad, min(xlo + c, g.ws - 1) + pad,
                                  g.c8_total > 0 ? g.c8_total : C8, hs, ws);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p) + 16 * q);
        rv[m][q] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile >= tiles) return;
  int ty0, tx0, ylo, xlo;
  window_origin(tile, ty0, tx0, ylo, xlo);
  load_window(ylo, xlo);
  for (; tile < tiles; tile += gridDim.x) {
    const int ty1 = min(ty0 + BT, Hi), tx1 = min(tx0 + BT, Wi);
    // ---- the y tap lists, this thread's x taps, the staged window -> LDS
    if (threadIdx.x < BT * BYT) {
      int ly = threadIdx.x / BYT, k = threadIdx.x - ly * BYT;
      int yi = min(ty0 + ly, Hi - 1);
      int a0 = tys[yi], cnt = tys[yi + 1] - a0;
      int r = k < cnt ? tyj[a0 + k] - ylo : -1;
      bool live = r >= 0 && r < BWIN;                       // dead / out-of-window entries: weight 0 (branch-free passes);
      s_yj[ly][k] = live ? r : 0;                           // pixels with out-of-window entries take the slow path
      s_yw[ly][k] = live ? tyw[a0 + k] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < VPT; ++m) {
      int i = threadIdx.x + m * 256;
      if (i < BWIN * BWIN)
#pragma unroll
        for (int q = 0; q < Q; ++q) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(&win[i * 8]) + 16 * q) = rv[m][q];
    }
    const int ly = threadIdx.x / BT, lx = threadIdx.x % BT;                            // x pass: one thread per pixel
    const int yi = min(ty0 + ly, Hi - 1), xi = min(tx0 + lx, Wi - 1);
    const int a0 = tys[yi], a1 = tys[yi + 1], b0 = txs[xi], nb = txs[xi + 1] - b0;
    int xj[BXT];
    float xw[BXT];
#pragma unroll
    for (int k = 0; k < BXT; ++k) {
      xj[k] = k < nb ? txj[b0 + k] - xlo : 0;
      xw[k] = k < nb ? txw[b0 + k] : 0.f;
    }
    bool fast = a1 - a0 <= BYT && nb <= BXT && (a1 == a0 || (tyj[a0] - ylo >= 0 && tyj[a1 - 1] - ylo < BWIN));
#pragma unroll
    for (int k = 0; k < BXT; ++k) {
      fast = fast && xj[k] >= 0 && xj[k] < BWIN;
      xj[k] = min(max(xj[k], 0), BWIN - 1);
    }
    __syncthreads();
    // the next tile's window: in flight during both passes
    const int cty0 = ty0, ctx0 = tx0;
    if (tile + (int)gridDim.x < tiles) {
      window_origin(tile + gridDim.x, ty0, tx0, ylo, xlo);
      load_window(ylo, xlo);
    }
    // y pass: tmp[ly][c] = sum_a w_a win[yo_a - ylo][c]
    for (int i = threadIdx.x; i < BT * BWIN; i += 256) {
      int py = i / BWIN, c = i - py * BWIN;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < BYT; ++k) {
        int r = s_yj[py][k];
        float wa = s_yw[py][k];
        float v[8];
        V8<T>::ld(&win[(r * BWIN + c) * 8], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wa * v[j];
      }
      tmp[0][py][c] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      tmp[1][py][c] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    if (cty0 + ly < ty1 && ctx0 + lx < tx1) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (fast) {
#pragma unroll
        for (int k = 0; k < BXT; ++k) {                         // dead entries: index 0, weight 0
          float w = xw[k];
          float4 p0 = tmp[0][ly][xj[k]], p1 = tmp[1][ly][xj[k]];
          acc[0] += w * p0.x; acc[1] += w * p0.y; acc[2] += w * p0.z; acc[3] += w * p0.w;
          acc[4] += w * p1.x; acc[5] += w * p1.y; acc[6] += w * p1.z; acc[7] += w * p1.w;
        }
      } else {
        for (int a = a0; a < a1; ++a) {
          int yo = tyj[a];
          float wa = tyw[a];
          for (int b = b0; b < b0 + nb; ++b) {
            float w = wa * txw[b];
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            grad_fetch_add<T>(g, n, cb, yo, txj[b], C8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
          }
        }
      }
      V8<T>::st(dx + cb8_index(n, cb, yi, xi, C8, Hi, Wi), acc);
    }
    __syncthreads();                                       // tmp / win / the tap lists are rewritten by the next tile
  }
}

// The same adjoint as a WALK down the output (high-res) rows: a block owns a strip of SW output columns of one (sample,
// channel block) and a chunk of input rows; thread = output column.  Each output row is read ONCE, straight from global
// memory into registers (two rows in flight), and scattered with the forward taps (4 per row, block-uniform) into a register
// window of the four input rows that are still open; the window slides when the first tap moves on, and the row that
// leaves it is complete in y: it crosses LDS once (f32, even / odd columns apart: conflict-free both ways) and every thread
// pair gathers one input pixel's x taps from it.  Against the tiled kernel above this reads the gradient 1.1 x instead of
// 1.6 x (6 extra rows per chunk, no column halo at <= 512 columns), moves 320 B instead of 760 B through LDS per input
// pixel, and needs about half the instructions.  Any upsample ratio >= 1 in y (the window follows the table); x tap lists
// of <= BXT entries.  Rows whose taps are clamped (image border) take a select form of the same scatter.
template <typename T, int BXT, int SW>
__global__ __launch_bounds__(SW) void k_bicubic_bwd_walk(mc_grad_src g, int C8, int Hi, int Wi, const int* __restrict__ iy,
                                                         const float* __restrict__ wy, const int* __restrict__ tys,
                                                         const int* __restrict__ tyj, const int* __restrict__ txs,
                                                         const int* __restrict__ txj, const float* __restrict__ txw,
                                                         T* __restrict__ dx, int rpc, int cw, int strips) {
  constexpr int PL = SW + 8, ODD = SW / 2 + 8, Q = sizeof(T) == 4 ? 2 : 1;
  __shared__ float4 trow[2][2][PL];
  const int n = blockIdx.z, cb = blockIdx.y, c = threadIdx.x;
  const int strip = blockIdx.x % strips, chunk = blockIdx.x / strips;
  const int m0 = chunk * rpc, m1 = min(Hi, m0 + rpc);
  const int X0 = strip * cw, X1 = min(Wi, X0 + cw);
  if (m0 >= m1 || X0 >= X1) return;                                    // (block-uniform)
  const int F0 = txj[txs[X0]];                                         // first output column this strip's pixels gather from
  const int pad = g.kind == MC_GSRC_PLAIN ? 0 : g.pad;
  const int hs = g.hs + 2 * pad, ws = g.ws + 2 * pad;
  const bool colok = F0 + c < g.ws;
  const T* col = reinterpret_cast<const T*>(g.ptr) + cb8_index(n, cb + (g.c8_total > 0 ? g.cb_off : 0), pad, min(F0 + c, g.ws - 1) + pad,
                                                               g.c8_total > 0 ? g.c8_total : C8, hs, ws);
  const size_t rstride = (size_t)ws * 8;
  // this thread's half pixel of the x pass: input column X0 + c / 2, channels 4 (c & 1) .. + 3
  const int Xc = X0 + (c >> 1), half = c & 1;
  const bool act = Xc < X1;
  int xs[BXT];
  float xw[BXT];
  {
    const int Xq = min(Xc, Wi - 1), b0 = txs[Xq], nb = txs[Xq + 1] - b0;
#pragma unroll
    for (int k = 0; k < BXT; ++k) {
      int j = k < nb ? txj[b0 + k] - F0 : 0;
      j = min(max(j, 0), SW - 1);
      xs[k] = (j >> 1) + (j & 1) * ODD;
      xw[k] = k < nb ? txw[b0 + k] : 0.f;
    }
  }
  const int myslot = (c >> 1) + (c & 1) * ODD;
  const int fs = __builtin_amdgcn_readfirstlane(tyj[tys[m0]]), fe = __builtin_amdgcn_readfirstlane(tyj[tys[m1] - 1]);
  int basey = __builtin_amdgcn_readfirstlane(iy[fs * 4]);
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  int buf = 0;
  auto ldrow = [&](int yo, uint4 (&r)[Q]) {
    const char* p = reinterpret_cast<const char*>(col + (size_t)yo * rstride);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      uint4 v = *reinterpret_cast<const uint4*>(p + 16 * q);
      r[q] = colok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  // the oldest open input row is complete: x pass (if it is one of this chunk's rows), then the window slides
  auto emit = [&]() {
    if (basey >= m0 && basey < m1) {
      trow[buf][0][myslot] = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
      trow[buf][1][myslot] = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
      __syncthreads();                     // (one barrier per row: the buffer written two rows on was read before the next one)
      if (act) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < BXT; ++k) {                                 // dead entries: slot 0, weight 0
          const float4 p = trow[buf][half][xs[k]];
          o.x += xw[k] * p.x; o.y += xw[k] * p.y; o.z += xw[k] * p.z; o.w += xw[k] * p.w;
        }
        T* d = dx + cb8_index(n, cb, basey, Xc, C8, Hi, Wi) + 4 * half;
        if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(d) = o;
        else *reinterpret_cast<uint2*>(d) = make_uint2(pk_bf16(o.x, o.y), pk_bf16(o.z, o.w));
      }
      buf ^= 1;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = acc[3][j]; acc[3][j] = 0.f; }
    ++basey;
  };
  uint4 p0[Q], p1[Q];
  ldrow(fs, p0);
  ldrow(min(fs + 1, fe), p1);
  for (int yo = fs; yo <= fe; ++yo) {
    uint4 cur[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { cur[q] = p0[q]; p0[q] = p1[q]; }
    if (yo + 2 <= fe) ldrow(yo + 2, p1);
    const int t0 = __builtin_amdgcn_readfirstlane(iy[yo * 4]), t1 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 1]),
              t2 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 2]), t3 = __builtin_amdgcn_readfirstlane(iy[yo * 4 + 3]);
    const float w0 = wy[yo * 4], w1 = wy[yo * 4 + 1], w2 = wy[yo * 4 + 2], w3 = wy[yo * 4 + 3];
    while (basey < t0) emit();
    float v[8];
    if constexpr (sizeof(T) == 4) {
      v[0] = __uint_as_float(cur[0].x); v[1] = __uint_as_float(cur[0].y); v[2] = __uint_as_float(cur[0].z); v[3] = __uint_as_float(cur[0].w);
      v[4] = __uint_as_float(cur[Q - 1].x); v[5] = __uint_as_float(cur[Q - 1].y); v[6] = __uint_as_float(cur[Q - 1].z); v[7] = __uint_as_float(cur[Q - 1].w);
    } else {
      v[0] = __uint_as_float(cur[0].x << 16); v[1] = __uint_as_float(cur[0].x & 0xffff0000u);
      v[2] = __uint_as_float(cur[0].y << 16); v[3] = __uint_as_float(cur[0].y & 0xffff0000u);
      v[4] = __uint_as_float(cur[0].z << 16); v[5] = __uint_as_float(cur[0].z & 0xffff0000u);
      v[6] = __uint_as_float(cur[0].w << 16); v[7] = __uint_as_float(cur[0].w & 0xffff0000u);
    }
    if (t1 == basey + 1 && t2 == basey + 2 && t3 == basey + 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[0][j] += w0 * v[j]; acc[1][j] += w1 * v[j]; acc[2][j] += w2 * v[j]; acc[3][j] += w3 * v[j]; }
    } else {                                                             // clamped taps: several land on one row
      const int r1 = t1 - basey, r2 = t2 - basey, r3 = t3 - basey;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = (k == 0 ? w0 : 0.f) + (r1 == k ? w1 : 0.f) + (r2 == k ? w2 : 0.f) + (r3 == k ? w3 : 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] += wk * v[j];
      }
    }
  }
#pragma unroll 1
  for (int k = 0; k < 4; ++k) emit();
}

// (Round 3 tried a streaming form -- wave = input row, lanes over the output columns, the row's transposed tap list summed
// straight from global memory, then an x pass from one f32 row in LDS: +0.12 ms per step.  Every output row is read by the
// four input rows it feeds, and those re-reads come from L2, not from the 16 KB L1: 4 x the bytes cross the L2 -> CU path.
// The LDS window above is the right structure; what it costs is reading the window 8 + 8 times through LDS.)
// adjoint of the bicubic upsample for LARGE scale factors (NewFluidNet: x4, x8, x16), where the tile window of the kernel
// above does not fit and its per-pixel fallback gathers taps_y x taps_x (up to 64 x 64) vectors: two 1-D passes through an
// f32 intermediate [n][c8][hi][wo][8] instead (taps_y + taps_x per pixel, coalesced along x).
template <typename T>
__global__ void k_bicubic_bwd_ypass(mc_grad_src g, int C8, int Hi, int Wo, const int* __restrict__ tys,
                                    const int* __restrict__ tyj, const float* __restrict__ tyw, float* __restrict__ tmp) {
  const int n = blockIdx.z, cb = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hi * Wo; i += gridDim.x * blockDim.x) {
    const int yi = i / Wo, xo = i - yi * Wo;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = tys[yi]; a < tys[yi + 1]; ++a) {
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      grad_fetch_add<T>(g, n, cb, tyj[a], xo, C8, v);
      const float w = tyw[a];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
    }
    V8<float>::st(tmp + cb8_index(n, cb, yi, xo, C8, Hi, Wo), acc);
  }
}
template <typename T>
__global__ void k_bicubic_bwd_xpass(const float* __restrict__ tmp, int C8, int Hi, int Wi, int Wo, const int* __restrict__ txs,
                                    const int* __restrict__ txj, const float* __restrict__ txw, T* __restrict__ dx) {
  const int n = blockIdx.z, cb = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hi * Wi; i += gridDim.x * blockDim.x) {
    const int yi = i / Wi, xi = i - yi * Wi;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = txs[xi]; b < txs[xi + 1]; ++b) {
      float v[8];
      V8<float>::ld(tmp + cb8_index(n, cb, yi, txj[b], C8, Hi, Wo), v);
      const float w = txw[b];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
    }
    V8<T>::st(dx + cb8_index(n, cb, yi, xi, C8, Hi, Wi), acc);
  }
}

template <typename T>
__global__ void k_rect_copy(const T* __restrict__ src, int hs, int ws, int sy, int sx, T* __restrict__ dst, int hd, int wd,
                            int dy, int dx, int rh, int rw, int C8, int accumulate) {
  const int n = blockIdx.z, cb = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rh * rw; i += gridDim.x * blockDim.x) {
    const int r = i / rw, c = i - r * rw;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (src) V8<T>::ld(src + cb8_index(n, cb, sy + r, sx + c, C8, hs, ws), v);
    T* d = dst + cb8_index(n, cb, dy + r, dx + c, C8, hd, wd);
    if (accumulate) {
      float o[8];
      V8<T>::ld(d, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    V8<T>::st(d, v);
  }
}

// =================================================================================================
// torch.cat along channels of more than two operands; sum of two gradient sources
// =================================================================================================
constexpr int CAT_MAX = 12;
struct CatTable { int n; const void* src[CAT_MAX]; int c8[CAT_MAX]; int first[CAT_MAX + 1]; };
template <typename T>
__global__ void k_concat_cb8(CatTable t, int C8, int HW, T* __restrict__ out) {
  const int n = blockIdx.z, cb = blockIdx.y;
  int k = 0;
#pragma unroll 1
  for (int j = 1; j < t.n; ++j) if (cb >= t.first[j]) k = j;
  const uint4* s = reinterpret_cast<const uint4*>(t.src[k]) + ((size_t)n * t.c8[k] + (cb - t.first[k])) * HW * (sizeof(T) / 2);
  uint4* d = reinterpret_cast<uint4*>(out) + ((size_t)n * C8 + cb) * HW * (sizeof(T) / 2);
  const int total = HW * (int)(sizeof(T) / 2);                    // 16-byte pieces per plane
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) d[i] = s[i];
}
template <typename T>
__global__ void k_gsrc_sum(mc_grad_src g0, mc_grad_src g1, int C8, int H, int W, T* __restrict__ out) {
  const int n = blockIdx.z, cb = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    const int y = i / W, x = i - y * W;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    grad_fetch_add<T>(g0, n, cb, y, x, C8, acc);
    grad_fetch_add<T>(g1, n, cb, y, x, C8, acc);
    V8<T>::st(out + cb8_index(n, cb, y, x, C8, H, W), acc);
  }
}

// =================================================================================================
// curl head (Unet :2038-2068)
// =================================================================================================
// raw stencil values (before wall/corner fix-up) at interior-clamped positions
__device__ __forceinline__ float curl_u_raw(const float* a, int y, int x, int H, int W) {
  // u_in[y'][x'] = 0.5 (a[y'+2][x'+1] - a[y'][x'+1]), y' in [0,H-3], x' in [0,W-3]; replicate pad by 1
  int yc = min(max(y - 1, 0), H - 3), xc = min(max(x - 1, 0), W - 3);
  return 0.5f * (a[(size_t)(yc + 2) * W + xc + 1] - a[(size_t)yc * W + xc + 1]);
}
__device__ __forceinline__ float curl_v_raw(const float* a, int y, int x, int H, int W) {
  int yc = min(max(y - 1, 0), H - 3), xc = min(max(x - 1, 0), W - 3);
  return -0.5f * (a[(size_t)(yc + 1) * W + xc + 2] - a[(size_t)(yc + 1) * W + xc]);
}

__global__ void k_curl_fwd(const float* __restrict__ a_, int H, int W, int64_t abs_, float ab, float* __restrict__ u,
                           float* __restrict__ v) {
  const int n = blockIdx.y;
  const float* a = a_ + (size_t)n * abs_;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    int y = i / W, x = i % W;
    bool corner = (y == 0 || y == H - 1) && (x == 0 || x == W - 1);
    float uu, vv;
    if (x == 0) uu = -curl_u_raw(a, y, 1, H, W);
    else if (x == W - 1) uu = -curl_u_raw(a, y, W - 2, H, W);
    else uu = curl_u_raw(a, y, x, H, W);
    if (y == 0) vv = -curl_v_raw(a, 1, x, H, W);
    else if (y == H - 1) vv = -curl_v_raw(a, H - 2, x, H, W);
    else vv = curl_v_raw(a, y, x, H, W);
    u[(size_t)n * H * W + i] = corner ? 0.f : ab * uu;
    v[(size_t)n * H * W + i] = corner ? 0.f : ab * vv;
  }
}

// adjoint: fold the wall / replicate structure of (gu, gv) onto the interior stencil outputs,
// then apply the transposed stencils.
__global__ void k_curl_bwd_fold(const float* __restrict__ gu, const float* __restrict__ gv, int H, int W,
                                float* __restrict__ eu, float* __restrict__ ev) {
  // eu/ev: [n][H-2][W-2] effective gradients w.r.t. the stencil outputs u_in, v_in
  const int n = blockIdx.y;
  const float* gun = gu + (size_t)n * H * W;
  const float* gvn = gv + (size_t)n * H * W;
  const int Hi = H - 2, Wi = W - 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hi * Wi; i += gridDim.x * blockDim.x) {
    int yp = i / Wi, xp = i % Wi;
    // u: rows y with clamp(y-1)==yp : y = yp+1, plus y=0 if yp==0, y=H-1 if yp==H-3
    float su = 0.f, sv = 0.f;
    int ys[3], ny = 0;
    ys[ny++] = yp + 1;
    if (yp == 0) ys[ny++] = 0;
    if (yp == Hi - 1) ys[ny++] = H - 1;
    int xs[3], nx = 0;
    xs[nx++] = xp + 1;
    if (xp == 0) xs[nx++] = 0;
    if (xp == Wi - 1) xs[nx++] = W - 1;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        int y = ys[a], x = xs[b];
        bool corner = (y == 0 || y == H - 1) && (x == 0 || x == W - 1);
        if (corner) continue;
        // u: wall columns x==0 / W-1 take -u[:, 1] / -u[:, W-2], which are u_in[.., 0] / u_in[.., W-3]
        su += ((x == 0 || x == W - 1) ? -1.f : 1.f) * gun[(size_t)y * W + x];
        sv += ((y == 0 || y == H - 1) ? -1.f : 1.f) * gvn[(size_t)y * W + x];
      }
    eu[(size_t)n * Hi * Wi + i] = su;
    ev[(size_t)n * Hi * Wi + i] = sv;
  }
}

__global__ void k_curl_bwd_stencil(const float* __restrict__ eu, const float* __restrict__ ev, int H, int W, float ab,
                                   float* __restrict__ ga_, int64_t gabs) {
  const int n = blockIdx.y;
  const int Hi = H - 2, Wi = W - 2;
  const float* eun = eu + (size_t)n * Hi * Wi;
  const float* evn = ev + (size_t)n * Hi * Wi;
  float* ga = ga_ + (size_t)n * gabs;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    int y = i / W, x = i % W;
    float acc = 0.f;
    // u_in[yp][xp] = 0.5 (a[yp+2][xp+1] - a[yp][xp+1])
    int xp = x - 1;
    if (xp >= 0 && xp < Wi) {
      if (y - 2 >= 0 && y - 2 < Hi) acc += 0.5f * eun[(size_t)(y - 2) * Wi + xp];
      if (y < Hi) acc -= 0.5f * eun[(size_t)y * Wi + xp];
    }
    // v_in[yp][xp] = -0.5 (a[yp+1][xp+2] - a[yp+1][xp])
    int yp = y - 1;
    if (yp >= 0 && yp < Hi) {
      if (x - 2 >= 0 && x - 2 < Wi) acc -= 0.5f * evn[(size_t)yp * Wi + x - 2];
      if (x < Wi) acc += 0.5f * evn[(size_t)yp * Wi + x];
    }
    ga[i] = ab * acc;
  }
}

__global__ void k_clip_fwd(const float* __restrict__ t, int hw, int64_t bs, float lo, float hi, float* __restrict__ o) {
  const int n = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x)
    o[(size_t)n * hw + i] = fminf(fmaxf(t[(size_t)n * bs + i], lo), hi);
}
// torch.clip passes the gradient where lo <= x <= hi
__global__ void k_clip_bwd(const float* __restrict__ go, const float* __restrict__ t, int hw, int64_t bs, int64_t gbs, float lo,
                           float hi, float* __restrict__ gi) {
  const int n = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
    float x = t[(size_t)n * bs + i];
    gi[(size_t)n * gbs + i] = (x >= lo && x <= hi) ? go[(size_t)n * hw + i] : 0.f;
  }
}

inline dim3 grid1(size_t total, int block = 256, int cap = 8192) {
  size_t g = (total + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}
inline dim3 grid3(int per_plane, int c8, int n, int block = 256, int capx = 64) {
  int gx = (per_plane + block - 1) / block;
  if (gx > capx) gx = capx;
  if (gx < 1) gx = 1;
  return dim3(gx, c8, n);
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================

extern "C" {



int mc_pack_nchw(const float* x, int32_t n, int32_t c, int32_t src_c, int32_t h, int32_t w, int32_t pad_w,
                 int32_t pad_mode, const float* chan_scale, int32_t dtype, void* out, void* stream) {
  if (!x || !out || n <= 0 || c <= 0 || src_c < c || h <= 0 || w <= 0 || pad_w < 0) return MC_EINVAL;
  if (pad_mode == MC_PAD_REFLECT && pad_w >= w) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  size_t total = (size_t)n * ((c + 7) / 8) * h * (w + 2 * pad_w);
  if (dtype == MC_F32) hipLaunchKernelGGL(k_pack_nchw<float>, grid1(total), dim3(256), 0, s, x, n, c, src_c, h, w, pad_w, pad_mode, chan_scale, (float*)out);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_pack_nchw<bf16_t>, grid1(total), dim3(256), 0, s, x, n, c, src_c, h, w, pad_w, pad_mode, chan_scale, (bf16_t*)out);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_pack_nchw<f16_t>, grid1(total), dim3(256), 0, s, x, n, c, src_c, h, w, pad_w, pad_mode, chan_scale, (f16_t*)out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_unpack_nchw(const void* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_w, const float* mean_nc,
                   int32_t dtype, float* out, void* stream) {
  if (!x || !out || n <= 0 || c <= 0 || h <= 0 || w <= 2 * crop_w || crop_w < 0) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  size_t total = (size_t)n * ((c + 7) / 8) * h * (w - 2 * crop_w);
  if (dtype == MC_F32) hipLaunchKernelGGL(k_unpack_nchw<float>, grid1(total), dim3(256), 0, s, (const float*)x, n, c, h, w, crop_w, mean_nc, out);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_unpack_nchw<bf16_t>, grid1(total), dim3(256), 0, s, (const bf16_t*)x, n, c, h, w, crop_w, mean_nc, out);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_unpack_nchw<f16_t>, grid1(total), dim3(256), 0, s, (const f16_t*)x, n, c, h, w, crop_w, mean_nc, out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_pack_grad_nchw(const float* g, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_w, const float* mean_nc,
                      int32_t dtype, void* out, void* stream) {
  if (!g || !out || n <= 0 || c <= 0 || h <= 0 || w <= 2 * crop_w || crop_w < 0) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  size_t total = (size_t)n * ((c + 7) / 8) * h * w;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_pack_grad_nchw<float>, grid1(total), dim3(256), 0, s, g, n, c, h, w, crop_w, mean_nc, (float*)out);
  else if (mc_is16(dtype)) hipLaunchKernelGGL(k_pack_grad_nchw<bf16_t>, grid1(total), dim3(256), 0, s, g, n, c, h, w, crop_w, mean_nc, (bf16_t*)out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_sum_hw(const float* x, int32_t nc, int32_t hw, float scale, float* out, void* stream) {
  if (!x || !out || nc <= 0 || hw <= 0) return MC_EINVAL;
  hipLaunchKernelGGL(k_sum_hw, dim3(nc), dim3(1024), 0, (hipStream_t)stream, x, hw, scale, out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_partials(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype, int32_t tiles, float* part,
                   void* stream) {
  if (!y || !part || n <= 0 || c <= 0 || h <= 0 || w <= 0 || tiles <= 0 || tiles > h) return MC_EINVAL;
  const int C8 = (c + 7) / 8;
  dim3 g(tiles, C8, n);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_gn_partials<float>, g, dim3(256), 0, s, (const float*)y, C8, h, w, tiles, part);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_gn_partials<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)y, C8, h, w, tiles, part);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_gn_partials<f16_t>, g, dim3(256), 0, s, (const f16_t*)y, C8, h, w, tiles, part);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_finalize_coef(const float* part, int32_t n, int32_t tiles, int32_t c, int32_t groups, int32_t hw, float eps,
                        const float* gamma, const float* beta, float* stats, float* coef4, void* stream) {
  if (!part || n <= 0 || tiles <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || hw <= 0) return MC_EINVAL;
  if (!stats && !coef4) return MC_EINVAL;
  if (coef4 && (!gamma || !beta)) return MC_EINVAL;
  int CP = ((c + 7) / 8) * 8;
  hipLaunchKernelGGL(k_gn_finalize, dim3(groups, n), dim3(64), 0, (hipStream_t)stream, part, tiles, c, CP, groups, hw,
                     eps, stats, nullptr, gamma, beta, coef4);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_finalize(const float* part, int32_t n, int32_t tiles, int32_t c, int32_t groups, int32_t hw, float eps,
                   float* stats, float* chan_mean, void* stream) {
  if (!part || n <= 0 || tiles <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || hw <= 0) return MC_EINVAL;
  if (!stats && !chan_mean) return MC_EINVAL;
  int CP = ((c + 7) / 8) * 8;
  hipLaunchKernelGGL(k_gn_finalize, dim3(groups, n), dim3(64), 0, (hipStream_t)stream, part, tiles, c, CP, groups, hw,
                     eps, stats, chan_mean, nullptr, nullptr, nullptr);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

static int fill_gn_args(GnArgs& a, int n, int c, int h, int w, int groups, const float* stats, const float* gamma,
                        const float* beta, int post, int act) {
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MC_EINVAL;
  if (post == MC_POST_GN_ACT && (groups <= 0 || c % groups != 0 || !stats || !gamma || !beta)) return MC_EINVAL;
  if (post < MC_POST_NONE || post > MC_POST_GN_ACT || act < MC_ACT_NONE || act > MC_ACT_ELU) return MC_EINVAL;
  a.N = n; a.C = c; a.C8 = (c + 7) / 8; a.H = h; a.W = w;
  a.groups = groups > 0 ? groups : 1;
  a.cpg = c / a.groups; a.post = post; a.act = act;
  a.stats = stats; a.gamma = gamma; a.beta = beta;
  return MC_OK;
}

int mc_gn_act_fwd(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups, const float* stats,
                  const float* gamma, const float* beta, int32_t post, int32_t act, int32_t pool, int32_t dtype,
                  void* a_out, void* pooled, void* stream) {
  GnArgs a;
  int rc = fill_gn_args(a, n, c, h, w, groups, stats, gamma, beta, post, act);
  if (rc) return rc;
  if (!y || (!a_out && pool == 1) || (pool > 1 && !pooled)) return MC_EINVAL;
  if (pool != 1 && pool != 2 && pool != 4) return MC_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int per = cdiv(h, pool) * cdiv(w, pool);
  dim3 g(max(1, min(cdiv(per, 256 * gn_fwd_vpt()), 4096)), a.C8, n);
  if (pool == 2 && (w & 1) == 0 && h >= 2 && dtype != MC_F32) {     // row-pair form (16-bit types: one 16-byte vector per pixel)
    dim3 gp(max(1, min(cdiv(cdiv(h, 2) * w, 256 * 2), 4096)), a.C8, n);
    if (dtype == MC_BF16) hipLaunchKernelGGL(k_gn_act_fwd_pool2_rows<bf16_t>, gp, dim3(256), 0, s, a, (const bf16_t*)y, (bf16_t*)a_out, (bf16_t*)pooled);
    else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_gn_act_fwd_pool2_rows<f16_t>, gp, dim3(256), 0, s, a, (const f16_t*)y, (f16_t*)a_out, (f16_t*)pooled);
    else return MC_EUNSUPPORTED;
    MC_CHECK_LAUNCH();
    return MC_OK;
  }
#define GN_LAUNCH(T, P) hipLaunchKernelGGL((k_gn_act_fwd<T, P>), g, dim3(256), 0, s, a, (const T*)y, (T*)a_out, (T*)pooled)
  if (dtype == MC_F32) { if (pool == 1) GN_LAUNCH(float, 1); else if (pool == 2) GN_LAUNCH(float, 2); else GN_LAUNCH(float, 4); }
  else if (dtype == MC_BF16) { if (pool == 1) GN_LAUNCH(bf16_t, 1); else if (pool == 2) GN_LAUNCH(bf16_t, 2); else GN_LAUNCH(bf16_t, 4); }
  else if (dtype == MC_MIX16) { if (pool == 1) GN_LAUNCH(f16_t, 1); else if (pool == 2) GN_LAUNCH(f16_t, 2); else GN_LAUNCH(f16_t, 4); }
  else return MC_EUNSUPPORTED;
#undef GN_LAUNCH
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_avgpool_fwd(const void* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t f, int32_t dtype, void* out,
                   void* stream) {
  if (!x || !out || n <= 0 || c <= 0 || f < 1 || h < f || w < f) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int C8 = (c + 7) / 8;
  dim3 g = grid3((h / f) * (w / f), C8, n, 256, 4096);
  if (dtype == MC_F32) hipLaunchKernelGGL(k_avgpool<float>, g, dim3(256), 0, s, (const float*)x, C8, h, w, f, (float*)out);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_avgpool<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)x, C8, h, w, f, (bf16_t*)out);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_avgpool<f16_t>, g, dim3(256), 0, s, (const f16_t*)x, C8, h, w, f, (f16_t*)out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int32_t mc_gn_bwd_blocks(int32_t h, int32_t w) {
  static int env_vpt = env_int("MC_GN_RED_VPT", 0);
  // vectors per thread of the phase-1 reduction.  Round 2 used 32 on large images (measured on the kernel alone); inside the
  // round-3 step 8 is better everywhere (whole step at 32 x 506 x 512, 8 / 16 / 32 / 64: 10.15 / 10.19 / 10.33 / 10.57 ms
  // mixed, 10.10 / 10.10 / 10.31 plain bf16: the two-source reduction of the encoder's last layers wants many short blocks)
  const int vpt = env_vpt > 0 ? env_vpt : 8;
  int b = cdiv(h * w, 256 * vpt);
  if (b > 128) b = 128;
  if (b < 1) b = 1;
  return b;
}

static int check_gsrc(const mc_grad_src* g) {
  if (!g) return MC_OK;
  if (g->kind == MC_GSRC_NONE) return MC_OK;
  if (!g->ptr || g->hs <= 0 || g->ws <= 0) return MC_EINVAL;
  if (g->kind < MC_GSRC_NONE || g->kind > MC_GSRC_PLAIN_POOL) return MC_EINVAL;
  if ((g->kind == MC_GSRC_PADFOLD_POOL || g->kind == MC_GSRC_PLAIN_POOL) && g->pool < 1) return MC_EINVAL;
  if ((g->kind == MC_GSRC_PADFOLD || g->kind == MC_GSRC_PADFOLD_POOL) && (g->pad < 0 || g->pad > 2)) return MC_EUNSUPPORTED;
  if (g->c8_total < 0 || g->cb_off < 0 || (g->c8_total > 0 && g->cb_off >= g->c8_total)) return MC_EINVAL;
  return MC_OK;
}
static mc_grad_src gsrc_or_none(const mc_grad_src* g) {
  mc_grad_src z;
  z.ptr = nullptr; z.kind = MC_GSRC_NONE; z.pad = 0; z.pad_mode = 0; z.pool = 1; z.hs = 0; z.ws = 0; z.c8_total = 0; z.cb_off = 0;
  return g ? *g : z;
}

int mc_gn_act_bwd_reduce(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups, const float* stats,
                         const float* gamma, const float* beta, int32_t post, int32_t act, int32_t dtype,
                         const mc_grad_src* g0, const mc_grad_src* g1, float* partials, void* stream) {
  GnArgs a;
  int rc = fill_gn_args(a, n, c, h, w, groups, stats, gamma, beta, post, act);
  if (rc) return rc;
  if (post != MC_POST_GN_ACT || !y || !partials || !g0) return MC_EINVAL;
  if ((rc = check_gsrc(g0)) || (rc = check_gsrc(g1))) return rc;
  dim3 g(mc_gn_bwd_blocks(h, w), a.C8, n);
  hipStream_t s = (hipStream_t)stream;
  const mc_grad_src s0 = gsrc_or_none(g0), s1 = gsrc_or_none(g1);
#define RED(T, GK) hipLaunchKernelGGL((k_gn_bwd_reduce<T, GK>), g, dim3(256), 0, s, a, (const T*)y, s0, s1, partials, a.C8 * 8)
#define REDH(GK) hipLaunchKernelGGL((k_gn_bwd_reduce<bf16_t, GK, f16_t>), g, dim3(256), 0, s, a, (const f16_t*)y, s0, s1, partials, a.C8 * 8)
  if (dtype == MC_F32) RED(float, 0);
  else if (dtype == MC_BF16) {
    switch (gkind_of(s0, s1)) { case 1: RED(bf16_t, 1); break; case 2: RED(bf16_t, 2); break; case 3: RED(bf16_t, 3); break; default: RED(bf16_t, 0); }
  } else if (dtype == MC_MIX16) {
    switch (gkind_of(s0, s1)) { case 1: REDH(1); break; case 2: REDH(2); break; case 3: REDH(3); break; default: REDH(0); }
  } else return MC_EUNSUPPORTED;
#undef REDH
#undef RED
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_act_bwd_finalize(const float* partials, int32_t n, int32_t blocks, int32_t c, int32_t groups, int32_t hw,
                           const float* gamma, float* m12, float* dgamma, float* dbeta, void* stream) {
  if (!partials || n <= 0 || blocks <= 0 || c <= 0 || groups <= 0 || c % groups || hw <= 0) return MC_EINVAL;
  hipLaunchKernelGGL(k_gn_bwd_finalize, dim3(groups), dim3(1024), 0, (hipStream_t)stream, partials, n, blocks, c,
                     ((c + 7) / 8) * 8, groups, hw, gamma, m12, dgamma, dbeta);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_act_bwd_finalize_n(const float* partials, int32_t n, int32_t blocks, int32_t c, int32_t groups, int32_t hw,
                             const float* gamma, float* m12, float* chan_sums, void* stream) {
  if (!partials || !chan_sums || n <= 0 || blocks <= 0 || c <= 0 || groups <= 0 || c % groups || hw <= 0) return MC_EINVAL;
  const int cpg = c / groups;
  if (cpg != 1 && cpg != 2 && cpg != 4 && cpg != 8) return MC_EUNSUPPORTED;
  hipLaunchKernelGGL(k_gn_bwd_finalize_n, dim3(groups, n), dim3(256), 0, (hipStream_t)stream, partials, blocks, c,
                     ((c + 7) / 8) * 8, groups, hw, gamma, m12, chan_sums);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_act_bwd_apply(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups, const float* stats,
                        const float* m12, const float* gamma, const float* beta, int32_t post, int32_t act,
                        int32_t dtype, const mc_grad_src* g0, const mc_grad_src* g1, void* dy, void* stream) {
  GnArgs a;
  int rc = fill_gn_args(a, n, c, h, w, groups, stats, gamma, beta, post, act);
  if (rc) return rc;
  if (!y || !dy || !g0 || (post == MC_POST_GN_ACT && !m12)) return MC_EINVAL;
  if ((rc = check_gsrc(g0)) || (rc = check_gsrc(g1))) return rc;
  if (post == MC_POST_NONE) a.act = MC_ACT_NONE;
  const int rows = gn_apply_rows(h, w);
  dim3 g(cdiv(h, rows), a.C8, n);
  hipStream_t s = (hipStream_t)stream;
  const mc_grad_src s0 = gsrc_or_none(g0), s1 = gsrc_or_none(g1);
#define APP(T, GK) hipLaunchKernelGGL((k_gn_bwd_apply<T, GK>), g, dim3(256), 0, s, a, (const T*)y, m12, s0, s1, (T*)dy, rows)
#define APPH(GK) hipLaunchKernelGGL((k_gn_bwd_apply<bf16_t, GK, f16_t>), g, dim3(256), 0, s, a, (const f16_t*)y, m12, s0, s1, (bf16_t*)dy, rows)
  if (dtype == MC_F32) APP(float, 0);
  else if (dtype == MC_BF16) {
    switch (gkind_of(s0, s1)) { case 1: APP(bf16_t, 1); break; case 2: APP(bf16_t, 2); break; case 3: APP(bf16_t, 3); break; default: APP(bf16_t, 0); }
  } else if (dtype == MC_MIX16) {
    switch (gkind_of(s0, s1)) { case 1: APPH(1); break; case 2: APPH(2); break; case 3: APPH(3); break; default: APPH(0); }
  } else return MC_EUNSUPPORTED;
#undef APPH
#undef APP
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_act_fwd_small(const void* y, const float* stat_partials, int32_t tiles, int32_t n, int32_t c, int32_t h, int32_t w,
                        int32_t groups, float eps, const float* gamma, const float* beta, int32_t act, int32_t pool,
                        int32_t dtype, float* stats_out, void* a_out, void* pooled, void* stream) {
  GnArgs a;
  int rc = fill_gn_args(a, n, c, h, w, groups, stats_out, gamma, beta, MC_POST_GN_ACT, act);
  if (rc) return rc;
  if (!y || !stat_partials || tiles <= 0 || !gamma || !beta || !stats_out || !a_out || (pool > 1 && !pooled)) return MC_EINVAL;
  if (pool != 1 && pool != 2) return MC_EUNSUPPORTED;
  if (a.cpg != 1 && a.cpg != 2 && a.cpg != 4 && a.cpg != 8) return MC_EUNSUPPORTED;
  dim3 g(a.C8, n);
  hipStream_t s = (hipStream_t)stream;
#define FS(T, P) hipLaunchKernelGGL((k_gn_act_small<T, P>), g, dim3(gn_small_threads(a.C8 * n)), 0, s, a, (const T*)y, stat_partials, tiles, eps, stats_out, (T*)a_out, (T*)pooled)
  if (dtype == MC_F32) { if (pool == 1) FS(float, 1); else FS(float, 2); }
  else if (dtype == MC_BF16) { if (pool == 1) FS(bf16_t, 1); else FS(bf16_t, 2); }
  else if (dtype == MC_MIX16) { if (pool == 1) FS(f16_t, 1); else FS(f16_t, 2); }
  else return MC_EUNSUPPORTED;
#undef FS
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_act_bwd_small(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups, const float* stats,
                        const float* gamma, const float* beta, int32_t act, int32_t dtype, const mc_grad_src* g0,
                        const mc_grad_src* g1, void* dy, float* chan_sums, void* stream) {
  GnArgs a;
  int rc = fill_gn_args(a, n, c, h, w, groups, stats, gamma, beta, MC_POST_GN_ACT, act);
  if (rc) return rc;
  if (!y || !dy || !g0 || !stats || !gamma || !beta || !chan_sums) return MC_EINVAL;
  if (a.cpg != 1 && a.cpg != 2 && a.cpg != 4 && a.cpg != 8) return MC_EUNSUPPORTED;     // a group must lie inside one channel block
  if ((rc = check_gsrc(g0)) || (rc = check_gsrc(g1))) return rc;
  dim3 g(a.C8, n);
  hipStream_t s = (hipStream_t)stream;
  const mc_grad_src s0 = gsrc_or_none(g0), s1 = gsrc_or_none(g1);
  const int CP = a.C8 * 8;
#define SM(T, GK) hipLaunchKernelGGL((k_gn_bwd_small<T, GK>), g, dim3(gn_small_threads(a.C8 * n)), 0, s, a, (const T*)y, s0, s1, (T*)dy, chan_sums, CP)
#define SMH(GK) hipLaunchKernelGGL((k_gn_bwd_small<bf16_t, GK, f16_t>), g, dim3(gn_small_threads(a.C8 * n)), 0, s, a, (const f16_t*)y, s0, s1, (bf16_t*)dy, chan_sums, CP)
  if (dtype == MC_F32) SM(float, 0);
  else if (dtype == MC_BF16) {
    switch (gkind_of(s0, s1)) { case 1: SM(bf16_t, 1); break; case 2: SM(bf16_t, 2); break; case 3: SM(bf16_t, 3); break; default: SM(bf16_t, 0); }
  } else if (dtype == MC_MIX16) {
    switch (gkind_of(s0, s1)) { case 1: SMH(1); break; case 2: SMH(2); break; case 3: SMH(3); break; default: SMH(0); }
  } else return MC_EUNSUPPORTED;
#undef SMH
#undef SM
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_param_grads_batched(const float* const* chan_sums, const int32_t* n, const int32_t* c, float* const* dgamma,
                              float* const* dbeta, int32_t jobs, void* stream) {
  if (!chan_sums || !n || !c || !dgamma || !dbeta || jobs <= 0) return MC_EINVAL;
  for (int base = 0; base < jobs; base += GP_MAX) {
    GpTable t;
    t.n = jobs - base < GP_MAX ? jobs - base : GP_MAX;
    int first = 0;
    for (int k = 0; k < t.n; ++k) {
      const int i = base + k;
      if (!chan_sums[i] || n[i] <= 0 || c[i] <= 0) return MC_EINVAL;
      t.j[k].pc = chan_sums[i]; t.j[k].dgamma = dgamma[i]; t.j[k].dbeta = dbeta[i];
      t.j[k].N = n[i]; t.j[k].C = c[i]; t.j[k].CP = ((c[i] + 7) / 8) * 8; t.j[k].first = first;
      first += ((c[i] + 63) / 64) * 64;
    }
    hipLaunchKernelGGL(k_gn_param_grads, dim3(first / 64), dim3(64), 0, (hipStream_t)stream, t);
  }
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_fold_padded2(void* buf0, int32_t c0, void* buf1, int32_t c1, int32_t n, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode,
                    int32_t dtype, void* stream) {
  if (!buf0 || n <= 0 || c0 <= 0 || hs <= 0 || ws <= 0 || pad < 0 || pad > 2 || (buf1 && c1 <= 0)) return MC_EINVAL;
  if (pad == 0 || pad_mode == MC_PAD_ZEROS) return MC_OK;
  const int C8a = (c0 + 7) / 8, C8 = C8a + (buf1 ? (c1 + 7) / 8 : 0), t = pad + 1;
  int all = (hs < 2 * t || ws < 2 * t) ? 1 : 0;
  int total = all ? hs * ws : 2 * t * ws + (hs - 2 * t) * 2 * t;
  dim3 g(cdiv(total, 256), C8, n);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_fold_padded<float>, g, dim3(256), 0, s, (float*)buf0, C8, hs, ws, pad, pad_mode, all, (float*)buf1, C8a);
  else if (mc_is16(dtype)) hipLaunchKernelGGL(k_fold_padded<bf16_t>, g, dim3(256), 0, s, (bf16_t*)buf0, C8, hs, ws, pad, pad_mode, all, (bf16_t*)buf1, C8a);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_fold_padded(void* buf, int32_t n, int32_t c, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode, int32_t dtype,
                   void* stream) {
  if (!buf || c <= 0) return MC_EINVAL;
  return mc_fold_padded2(buf, c, nullptr, 0, n, hs, ws, pad, pad_mode, dtype, stream);
}

static void fold_shape(int hs, int ws, int pad, int& all, int& total) {
  const int t = pad + 1;
  all = (hs < 2 * t || ws < 2 * t) ? 1 : 0;
  total = all ? hs * ws : 2 * t * ws + (hs - 2 * t) * 2 * t;
}

int32_t mc_fold_blocks(int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode) {
  if (hs <= 0 || ws <= 0 || pad <= 0 || pad_mode == MC_PAD_ZEROS) return 0;
  int all, total;
  fold_shape(hs, ws, pad, all, total);
  return cdiv(total, 256);
}

int mc_fold_padded_dz(void* buf, int32_t n, int32_t c, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode, int32_t dtype,
                      const void* y, const float* coef, int32_t act, float* partials, int32_t part_stride_blocks,
                      int32_t part_first_block, void* stream) {
  if (!buf || !y || !partials || n <= 0 || c <= 0 || hs <= 0 || ws <= 0 || pad < 0 || pad > 2) return MC_EINVAL;
  if (act < MC_ACT_NONE || act > MC_ACT_ELU) return MC_EINVAL;
  const int blocks = mc_fold_blocks(hs, ws, pad, pad_mode);
  if (blocks == 0) return MC_OK;                              // zero padding: every interior pixel was final already
  if (part_first_block < 0 || part_first_block + blocks > part_stride_blocks) return MC_EINVAL;
  int all, total;
  fold_shape(hs, ws, pad, all, total);
  const int C8 = (c + 7) / 8;
  dim3 g(blocks, C8, n);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_fold_padded_dz<float>, g, dim3(256), 0, s, (float*)buf, C8, hs, ws, pad, pad_mode, all, (const float*)y, coef, act, partials, part_stride_blocks, part_first_block);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_fold_padded_dz<bf16_t>, g, dim3(256), 0, s, (bf16_t*)buf, C8, hs, ws, pad, pad_mode, all, (const bf16_t*)y, coef, act, partials, part_stride_blocks, part_first_block);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL((k_fold_padded_dz<bf16_t, f16_t>), g, dim3(256), 0, s, (bf16_t*)buf, C8, hs, ws, pad, pad_mode, all, (const f16_t*)y, coef, act, partials, part_stride_blocks, part_first_block);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gn_bwd_apply_dz(const mc_grad_src* dz, const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups,
                       const float* coef, const float* m12, int32_t dtype, void* dy, void* stream) {
  if (!dz || !dy || n <= 0 || c <= 0 || h <= 0 || w <= 0) return MC_EINVAL;
  if (coef && (!y || !m12 || groups <= 0 || c % groups != 0)) return MC_EINVAL;
  int rc = check_gsrc(dz);
  if (rc) return rc;
  if ((dz->kind != MC_GSRC_PADFOLD && dz->kind != MC_GSRC_PLAIN) || dz->c8_total != 0 || dz->hs != h || dz->ws != w) return MC_EUNSUPPORTED;
  const int rows = gn_apply_rows(h, w), C8 = (c + 7) / 8;
  dim3 g(cdiv(h, rows), C8, n);
  hipStream_t s = (hipStream_t)stream;
  const int gr = groups > 0 ? groups : 1;
#define APZ(T, GK) hipLaunchKernelGGL((k_gn_bwd_apply_dz<T, GK>), g, dim3(256), 0, s, *dz, (const T*)y, c, C8, h, w, gr, coef, m12, (T*)dy, rows)
  if (dtype == MC_F32) { if (dz->kind == MC_GSRC_PADFOLD) APZ(float, MC_GSRC_PADFOLD); else APZ(float, MC_GSRC_PLAIN); }
  else if (dtype == MC_BF16) { if (dz->kind == MC_GSRC_PADFOLD) APZ(bf16_t, MC_GSRC_PADFOLD); else APZ(bf16_t, MC_GSRC_PLAIN); }
  else if (dtype == MC_MIX16) {
#define APZH(GK) hipLaunchKernelGGL((k_gn_bwd_apply_dz<bf16_t, GK, f16_t>), g, dim3(256), 0, s, *dz, (const f16_t*)y, c, C8, h, w, gr, coef, m12, (bf16_t*)dy, rows)
    if (dz->kind == MC_GSRC_PADFOLD) APZH(MC_GSRC_PADFOLD); else APZH(MC_GSRC_PLAIN);
#undef APZH
  }
  else return MC_EUNSUPPORTED;
#undef APZ
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_rect_copy(const void* src, int32_t hs, int32_t ws, int32_t sy, int32_t sx, void* dst, int32_t hd, int32_t wd,
                 int32_t dy, int32_t dx, int32_t rh, int32_t rw, int32_t n, int32_t c, int32_t accumulate, int32_t dtype,
                 void* stream) {
  if (!dst || n <= 0 || c <= 0 || rh <= 0 || rw <= 0) return MC_EINVAL;
  if (dy < 0 || dx < 0 || dy + rh > hd || dx + rw > wd) return MC_EINVAL;
  if (src && (sy < 0 || sx < 0 || sy + rh > hs || sx + rw > ws)) return MC_EINVAL;
  const int C8 = (c + 7) / 8;
  dim3 g(max(1, min(cdiv(rh * rw, 256), 512)), C8, n);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_rect_copy<float>, g, dim3(256), 0, s, (const float*)src, hs, ws, sy, sx, (float*)dst, hd, wd, dy, dx, rh, rw, C8, accumulate);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_rect_copy<f16_t>, g, dim3(256), 0, s, (const f16_t*)src, hs, ws, sy, sx, (f16_t*)dst, hd, wd, dy, dx, rh, rw, C8, accumulate);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_rect_copy<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)src, hs, ws, sy, sx, (bf16_t*)dst, hd, wd, dy, dx, rh, rw, C8, accumulate);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_concat_cb8(const void* const* srcs, const int32_t* src_c, int32_t n_src, int32_t n, int32_t h, int32_t w,
                  int32_t dtype, void* out, void* stream) {
  if (!srcs || !src_c || !out || n_src < 1 || n_src > CAT_MAX || n <= 0 || h <= 0 || w <= 0) return MC_EINVAL;
  CatTable t;
  t.n = n_src;
  int C8 = 0;
  for (int k = 0; k < n_src; ++k) {
    if (!srcs[k] || src_c[k] <= 0) return MC_EINVAL;
    if (k < n_src - 1 && (src_c[k] % 8) != 0) return MC_EUNSUPPORTED;      // only the last operand may end in a partial block
    t.src[k] = srcs[k]; t.c8[k] = (src_c[k] + 7) / 8; t.first[k] = C8;
    C8 += t.c8[k];
  }
  t.first[n_src] = C8;
  hipStream_t s = (hipStream_t)stream;
  const int per = h * w * (dtype == MC_F32 ? 2 : 1);          // (16-bit types: a plain copy of 16-byte pieces)
  dim3 g(max(1, min(cdiv(per, 256 * 4), 1024)), C8, n);
  if (dtype == MC_F32) hipLaunchKernelGGL(k_concat_cb8<float>, g, dim3(256), 0, s, t, C8, h * w, (float*)out);
  else if (mc_is16(dtype)) hipLaunchKernelGGL(k_concat_cb8<bf16_t>, g, dim3(256), 0, s, t, C8, h * w, (bf16_t*)out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_gsrc_sum(const mc_grad_src* g0, const mc_grad_src* g1, int32_t n, int32_t c, int32_t h, int32_t w,
                int32_t dtype, void* out, void* stream) {
  if (!g0 || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0) return MC_EINVAL;
  int rc;
  if ((rc = check_gsrc(g0)) || (rc = check_gsrc(g1))) return rc;
  const int C8 = (c + 7) / 8;
  dim3 g(max(1, min(cdiv(h * w, 256 * 4), 1024)), C8, n);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MC_F32) hipLaunchKernelGGL(k_gsrc_sum<float>, g, dim3(256), 0, s, gsrc_or_none(g0), gsrc_or_none(g1), C8, h, w, (float*)out);
  else if (mc_is16(dtype)) hipLaunchKernelGGL(k_gsrc_sum<bf16_t>, g, dim3(256), 0, s, gsrc_or_none(g0), gsrc_or_none(g1), C8, h, w, (bf16_t*)out);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_bicubic_fwd_act(const void* x, const float* coef, int32_t act, int32_t n, int32_t c, int32_t hi, int32_t wi,
                       int32_t ho, int32_t wo, const int32_t* idx_y, const float* wgt_y, const int32_t* idx_x,
                       const float* wgt_x, int32_t dtype, void* out, void* stream) {
  if (!x || !out || !idx_y || !wgt_y || !idx_x || !wgt_x || n <= 0 || c <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0)
    return MC_EINVAL;
  if (act < MC_ACT_NONE || act > MC_ACT_ELU) return MC_EINVAL;
  const int a = (coef == nullptr && act == MC_ACT_NONE) ? -1 : act;      // -1: x is used as it is
  int C8 = (c + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  static const int walk = env_int("MC_BICUBIC_FWD_WALK", 1);
  if (walk && ho >= hi && wo >= wi) {                                    // upsampling: the row walk (taps span <= 4 rows / 4 columns)
    const int SW = wo <= 128 ? 128 : (wo <= 256 ? 256 : 512);
    // one strip when the output row fits the block; else strips of SW - 8 output columns (their input span <= SW - 4)
    const int fw = wo <= SW ? wo : SW - 8, strips = cdiv(wo, fw);
    static const int target = env_int("MC_BICUBIC_FWD_WALK_BLOCKS", 1024);
    const int chunks = max(1, min(cdiv(target, C8 * n * strips), cdiv(ho, 8)));
    const int rpc = cdiv(ho, chunks);
    dim3 g(cdiv(ho, rpc) * strips, C8, n);
#define FWW_(T, S) hipLaunchKernelGGL((k_bicubic_fwd_walk<T, S>), g, dim3(S), 0, s, (const T*)x, C8, hi, wi, ho, wo, idx_y, wgt_y, idx_x, wgt_x, (T*)out, rpc, fw, strips, coef, a)
#define FWS_(T) do { if (SW == 128) FWW_(T, 128); else if (SW == 256) FWW_(T, 256); else FWW_(T, 512); } while (0)
    if (dtype == MC_F32) FWS_(float);
    else if (dtype == MC_BF16) FWS_(bf16_t);
    else if (dtype == MC_MIX16) FWS_(f16_t);
    else return MC_EUNSUPPORTED;
#undef FWS_
#undef FWW_
    MC_CHECK_LAUNCH();
    return MC_OK;
  }
  int tiles_x = cdiv(wo, FOW), tiles_y = cdiv(ho, FOH);
  dim3 g(tiles_x * tiles_y, C8, n);
  if (dtype == MC_F32) hipLaunchKernelGGL(k_bicubic_fwd<float>, g, dim3(256), 0, s, (const float*)x, C8, hi, wi, ho, wo, idx_y, wgt_y, idx_x, wgt_x, (float*)out, tiles_x, coef, a);
  else if (dtype == MC_BF16) hipLaunchKernelGGL(k_bicubic_fwd<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)x, C8, hi, wi, ho, wo, idx_y, wgt_y, idx_x, wgt_x, (bf16_t*)out, tiles_x, coef, a);
  else if (dtype == MC_MIX16) hipLaunchKernelGGL(k_bicubic_fwd<f16_t>, g, dim3(256), 0, s, (const f16_t*)x, C8, hi, wi, ho, wo, idx_y, wgt_y, idx_x, wgt_x, (f16_t*)out, tiles_x, coef, a);
  else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_bicubic_fwd(const void* x, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                   const int32_t* idx_y, const float* wgt_y, const int32_t* idx_x, const float* wgt_x, int32_t dtype,
                   void* out, void* stream) {
  return mc_bicubic_fwd_act(x, nullptr, MC_ACT_NONE, n, c, hi, wi, ho, wo, idx_y, wgt_y, idx_x, wgt_x, dtype, out, stream);
}

int mc_bicubic_bwd_taps(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                        const int32_t* tys, const int32_t* tyj, const float* tyw, const int32_t* txs, const int32_t* txj,
                        const float* txw, int32_t max_taps_y, int32_t max_taps_x, int32_t dtype, void* dx, void* stream) {
  if (!gs || !dx || !tys || !tyj || !tyw || !txs || !txj || !txw || n <= 0 || c <= 0) return MC_EINVAL;
  int rc = check_gsrc(gs);
  if (rc) return rc;
  if (gs->hs != ho || gs->ws != wo) return MC_EINVAL;
  if (gs->kind != MC_GSRC_PLAIN && gs->kind != MC_GSRC_PADFOLD) return MC_EUNSUPPORTED;   // the window is staged raw
  int C8 = (c + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  int tiles_x = cdiv(wi, BT), tiles_y = cdiv(hi, BT);
  // persistent blocks: ~6 resident per CU over the (tile, channel block, sample) grid, several tiles each
  static const int per_cu = env_int("MC_BICUBIC_BLOCKS_PER_CU", 6);
  const int want = max(1, (256 * per_cu) / max(1, C8 * n));
  dim3 g(min(tiles_x * tiles_y, want), C8, n);
  const bool y8 = max_taps_y > 0 && max_taps_y <= 8, x8 = max_taps_x > 0 && max_taps_x <= 8;
#define BW(T, A, B) hipLaunchKernelGGL((k_bicubic_bwd<T, A, B>), g, dim3(256), 0, s, *gs, C8, hi, wi, tys, tyj, tyw, txs, txj, txw, (T*)dx, tiles_x, tiles_x * tiles_y)
#define BWT(T) do { if (y8 && x8) BW(T, 8, 8); else if (y8) BW(T, 8, 12); else if (x8) BW(T, 12, 8); else BW(T, 12, 12); } while (0)
  if (dtype == MC_F32) BWT(float);
  else if (mc_is16(dtype)) BWT(bf16_t);
  else return MC_EUNSUPPORTED;
#undef BWT
#undef BW
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_bicubic_bwd_walk(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                        const int32_t* idx_y, const float* wgt_y, const int32_t* tys, const int32_t* tyj, const int32_t* txs,
                        const int32_t* txj, const float* txw, int32_t max_taps_x, int32_t dtype, void* dx, void* stream) {
  if (!gs || !dx || !idx_y || !wgt_y || !tys || !tyj || !txs || !txj || !txw || n <= 0 || c <= 0) return MC_EINVAL;
  int rc = check_gsrc(gs);
  if (rc) return rc;
  if (gs->hs != ho || gs->ws != wo || hi <= 0 || wi <= 0 || ho < hi || wo < wi) return MC_EINVAL;
  if (gs->kind != MC_GSRC_PLAIN && gs->kind != MC_GSRC_PADFOLD) return MC_EUNSUPPORTED;   // rows are read raw
  if (max_taps_x <= 0 || max_taps_x > 12) return MC_EUNSUPPORTED;
  const int C8 = (c + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  const int SW = wo <= 128 ? 128 : (wo <= 256 ? 256 : 512);
  // one strip when the row fits the block; else strips of cw input columns whose taps span <= SW output columns:
  // input column X gathers from output columns [s (X - 1.5) - 0.5, s (X + 2.5) - 0.5), s = wo / wi
  int cw = wi, strips = 1;
  if (wo > SW || wi > SW / 2) {
    cw = min(SW / 2, (int)((int64_t)(SW - 1) * wi / wo) - 3);
    if (cw < 1) return MC_EUNSUPPORTED;
    strips = cdiv(wi, cw);
  }
  static const int target = env_int("MC_BICUBIC_WALK_BLOCKS", 1024);   // (in-step A/B, 256 / 512 / 1024 / 2048 / 4096: +0.08 / 0 / -0.04 / -0.02 / +0.03 ms)
  const int chunks = max(1, min(cdiv(target, C8 * n * strips), cdiv(hi, 4)));
  const int rpc = cdiv(hi, chunks);
  dim3 g(cdiv(hi, rpc) * strips, C8, n);
#define BWW(T, B, S) hipLaunchKernelGGL((k_bicubic_bwd_walk<T, B, S>), g, dim3(S), 0, s, *gs, C8, hi, wi, idx_y, wgt_y, tys, tyj, txs, txj, txw, (T*)dx, rpc, cw, strips)
#define BWS(T, B) do { if (SW == 128) BWW(T, B, 128); else if (SW == 256) BWW(T, B, 256); else BWW(T, B, 512); } while (0)
#define BWT(T) do { if (max_taps_x <= 8) BWS(T, 8); else BWS(T, 12); } while (0)
  if (dtype == MC_F32) BWT(float);
  else if (mc_is16(dtype)) BWT(bf16_t);
  else return MC_EUNSUPPORTED;
#undef BWT
#undef BWS
#undef BWW
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_bicubic_bwd(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                   const int32_t* tys, const int32_t* tyj, const float* tyw, const int32_t* txs, const int32_t* txj,
                   const float* txw, int32_t dtype, void* dx, void* stream) {
  return mc_bicubic_bwd_taps(gs, n, c, hi, wi, ho, wo, tys, tyj, tyw, txs, txj, txw, 0, 0, dtype, dx, stream);
}

int mc_bicubic_bwd_separable(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                             const int32_t* tys, const int32_t* tyj, const float* tyw, const int32_t* txs, const int32_t* txj,
                             const float* txw, int32_t dtype, float* ws, void* dx, void* stream) {
  if (!gs || !dx || !ws || !tys || !tyj || !tyw || !txs || !txj || !txw || n <= 0 || c <= 0) return MC_EINVAL;
  int rc = check_gsrc(gs);
  if (rc) return rc;
  if (gs->hs != ho || gs->ws != wo || gs->kind == MC_GSRC_NONE) return MC_EINVAL;
  const int C8 = (c + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  dim3 g1(max(1, min(cdiv(hi * wo, 256), 2048)), C8, n), g2(max(1, min(cdiv(hi * wi, 256), 2048)), C8, n);
  if (dtype == MC_F32) {
    hipLaunchKernelGGL(k_bicubic_bwd_ypass<float>, g1, dim3(256), 0, s, *gs, C8, hi, wo, tys, tyj, tyw, ws);
    hipLaunchKernelGGL(k_bicubic_bwd_xpass<float>, g2, dim3(256), 0, s, ws, C8, hi, wi, wo, txs, txj, txw, (float*)dx);
  } else if (mc_is16(dtype)) {
    hipLaunchKernelGGL(k_bicubic_bwd_ypass<bf16_t>, g1, dim3(256), 0, s, *gs, C8, hi, wo, tys, tyj, tyw, ws);
    hipLaunchKernelGGL(k_bicubic_bwd_xpass<bf16_t>, g2, dim3(256), 0, s, ws, C8, hi, wi, wo, txs, txj, txw, (bf16_t*)dx);
  } else return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_curl_head_fwd(const float* a, const float* t_in, int32_t n, int32_t h, int32_t w, int64_t in_batch_stride,
                     float a_bound, float t_lo, float t_hi, float* u, float* v, float* t_out, void* stream) {
  if (!a || !u || !v || n <= 0 || h < 3 || w < 3 || ((t_in == nullptr) != (t_out == nullptr))) return MC_EINVAL;
  dim3 g(min(cdiv(h * w, 256), 4096), n);
  hipLaunchKernelGGL(k_curl_fwd, g, dim3(256), 0, (hipStream_t)stream, a, h, w, in_batch_stride, a_bound, u, v);
  if (t_in) hipLaunchKernelGGL(k_clip_fwd, g, dim3(256), 0, (hipStream_t)stream, t_in, h * w, in_batch_stride, t_lo, t_hi, t_out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_curl_head_bwd(const float* gu, const float* gv, const float* gt_out, const float* t_in, int32_t n, int32_t h,
                     int32_t w, float a_bound, float t_lo, float t_hi, float* ga, float* gt_in, int64_t g_batch_stride,
                     int64_t in_batch_stride, float* ws, void* stream) {
  if (!gu || !gv || !ga || !ws || n <= 0 || h < 3 || w < 3) return MC_EINVAL;
  if (gt_in && (!gt_out || !t_in)) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* eu = ws;
  float* ev = ws + (size_t)n * (h - 2) * (w - 2);
  dim3 g1(min(cdiv((h - 2) * (w - 2), 256), 4096), n);
  hipLaunchKernelGGL(k_curl_bwd_fold, g1, dim3(256), 0, s, gu, gv, h, w, eu, ev);
  dim3 g2(min(cdiv(h * w, 256), 4096), n);
  hipLaunchKernelGGL(k_curl_bwd_stencil, g2, dim3(256), 0, s, eu, ev, h, w, a_bound, ga, g_batch_stride);
  if (gt_in) hipLaunchKernelGGL(k_clip_bwd, g2, dim3(256), 0, s, gt_out, t_in, h * w, in_batch_stride, g_batch_stride, t_lo, t_hi, gt_in);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

}  // extern "C"
