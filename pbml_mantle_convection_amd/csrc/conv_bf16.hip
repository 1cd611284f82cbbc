// bf16 convolution path for gfx950: implicit GEMM on v_mfma_f32_16x16x32_bf16, f32 accumulate.
//
//   M = 16 consecutive output pixels of one row (one M-tile), N = 16 output channels (one N-tile),
//   K = 32 = 4 (tap, 8-channel block) pairs.  The k x k convolution IS a dense contraction with
//   K = C_in * k^2 (400 ... 4800 for the U-Net), so every conv goes through the matrix cores.
//
// Data layout: activations are CB8 ([N][C/8][H][W][8] bf16): one (pixel, channel-block) is a 16-byte
// vector = exactly one lane's MFMA A-fragment for one k-group, and a tile row is contiguous in HBM.
//
// One workgroup = TH x TW output pixels x (NT*16) output channels of one image.  Per 16-channel chunk
// of the (concatenated) input it stages the (TH+k-1) x (TW+k-1) input window (padding resolved by
// index mirroring at staging time, never in the MFMA loop) and the chunk's slice of the pre-packed
// filter bank into LDS, then every wave runs MT x NT accumulator tiles over 13 (k=5) / 5 (k=3) K-steps:
// one ds_read_b128 per A fragment, NT ds_read_b128 B fragments per step.  The epilogue adds the bias,
// takes the GroupNorm (sum, sum^2) partials from the f32 accumulators, transposes through LDS and
// writes 16-byte CB8 vectors.  The input gradient reuses the kernel on the zero-padded (k-1) domain
// with the rotated / transposed bank.
#include "conv_common.h"

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int CHUNK_CB = 2;     // 8-channel blocks per K-chunk (16 input channels)

template <int K> struct KSteps { static constexpr int pairs = K * K * CHUNK_CB; static constexpr int steps = (pairs + 3) / 4; };

struct Bf16Cfg { int th, tw, nt, mt; };

// output-tile configuration by number of N-tiles handled per workgroup
__host__ __device__ inline int pick_nt(int n_tiles) { return (n_tiles % 4 == 0) ? 4 : ((n_tiles % 2 == 0) ? 2 : 1); }

// ------------------------------------------------------------------------------------------------
// bank packing: bank[chunk][step][ntile][lane][8]  (bf16), lane = 16 g + n:
//   element e of lane (n, g) at step s = W[co = ntile*16 + n][ci = chunk*16 + cb*8 + e][tap]
//   with pair j = 4 s + g, tap = j / 2, cb = j % 2.
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_bf16(ConvGeom g, const float* __restrict__ wu, int dgrad, bf16_t* __restrict__ bank,
                            int chunks, int steps, int ntiles) {
  const int K = g.K, KK = K * K;
  const size_t total = (size_t)chunks * steps * ntiles * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int e = (int)(i & 7);
    int lane = (int)((i >> 3) & 63);
    size_t r = i >> 9;
    int nt = (int)(r % ntiles); r /= ntiles;
    int s = (int)(r % steps);
    int ck = (int)(r / steps);
    int n = lane & 15, gq = lane >> 4;
    int j = 4 * s + gq;
    int tap = j / CHUNK_CB, cb = j % CHUNK_CB;
    float v = 0.f;
    if (tap < KK) {
      int kin = ck * 16 + cb * 8 + e;     // padded input-channel index of this conv
      int kout = nt * 16 + n;             // padded output-channel index of this conv
      int co, cip;                        // forward (co, padded ci)
      int ky = tap / K, kx = tap % K;
      if (!dgrad) { co = kout; cip = kin; }
      else { co = kin; cip = kout; ky = K - 1 - ky; kx = K - 1 - kx; }
      int ob = cip / 8, oj = cip % 8;
      bool ok = co < g.Cout && ob < g.CBin &&
                (ob < g.CB0 ? (ob * 8 + oj < g.Cin0) : ((ob - g.CB0) * 8 + oj < g.Cin1));
      if (ok) {
        int ci = ob < g.CB0 ? ob * 8 + oj : g.Cin0 + (ob - g.CB0) * 8 + oj;
        int u = co, kxs = kx;
        if (co >= g.U) { u = co - g.U; kxs = K - 1 - kx; }
        v = wu[(((size_t)u * g.Cin + ci) * K + ky) * K + kxs];
      }
    }
    bank[i] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------
// forward / input-gradient kernel
// ------------------------------------------------------------------------------------------------
template <int K, int TH, int TW, int NT, int MT, bool OUT_F32 = false>
__global__ __launch_bounds__(256) void k_conv_mfma_bf16(ConvGeom g, const bf16_t* __restrict__ x0,
                                                        const bf16_t* __restrict__ x1, const bf16_t* __restrict__ bank,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y0,
                                                        bf16_t* __restrict__ y1, float* __restrict__ part, int n_groups) {
  constexpr int TIH = TH + K - 1, TIW = TW + K - 1;
  constexpr int PLANE = (TIH * TIW + 15) / 16 * 16;           // 16-byte slots per channel-block plane
  constexpr int STEPS = KSteps<K>::steps;
  constexpr int MTILES_X = TW / 16;
  static_assert(TH * MTILES_X == 4 * MT, "tile / wave decomposition mismatch");
  constexpr int IN_SLOTS = CHUNK_CB * PLANE;
  constexpr int W_SLOTS = STEPS * NT * 64;
  static_assert(!OUT_F32 || NT == 1, "f32 output is for the single-N-tile configuration");
  constexpr int OUT_SLOTS = TH * TW * NT * (OUT_F32 ? 4 : 2);  // [pixel][NT*16 couts] bf16 (f32) = NT*2 (4) slots per pixel
  constexpr int LDS_SLOTS = (IN_SLOTS + W_SLOTS) > OUT_SLOTS ? (IN_SLOTS + W_SLOTS) : OUT_SLOTS;
  __shared__ uint4 lds[LDS_SLOTS];
  __shared__ float red[4][NT * 16 * 2];
  uint4* in_s = lds;
  uint4* w_s = lds + IN_SLOTS;

  const int tile = blockIdx.x, grp = blockIdx.y, n = blockIdx.z;
  const int ty0 = (tile / g.tiles_x) * TH, tx0 = (tile % g.tiles_x) * TW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, gq = lane >> 4;
  const int ntile0 = grp * NT;                                  // first global N-tile of this block
  const int ntiles_total = gridDim.y * NT;
  const int chunks = (g.CBin + CHUNK_CB - 1) / CHUNK_CB;

  // per-lane A offsets (16-byte slots) for every K-step
  int aoff[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    int j = 4 * s + gq;
    int tap = j / CHUNK_CB, cb = j % CHUNK_CB;
    if (tap >= K * K) { tap = 0; }                              // dummy pair: weights are zero
    aoff[s] = cb * PLANE + (tap / K) * TIW + (tap % K);
  }
  // M-tile bases of this wave: tile t = wave*MT + i -> (row, col0)
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int ck = 0; ck < chunks; ++ck) {
    __syncthreads();
    // ---- stage the input window of this chunk
    for (int i = threadIdx.x; i < CHUNK_CB * TIH * TIW; i += 256) {
      int cb = i / (TIH * TIW);
      int rem = i - cb * (TIH * TIW);
      int r = rem / TIW, c = rem - r * TIW;
      int gcb = ck * CHUNK_CB + cb;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gcb < g.CBin) {
        int sy = pad_map(ty0 + r - g.pad, g.H, g.pad_mode), sx = pad_map(tx0 + c - g.pad, g.W, g.pad_mode);
        if (sy >= 0 && sx >= 0) {
          const bf16_t* src = gcb < g.CB0 ? x0 : x1;
          int scb = gcb < g.CB0 ? gcb : gcb - g.CB0;
          int sC8 = gcb < g.CB0 ? g.CB0 : g.CB1;
          v = *reinterpret_cast<const uint4*>(src + cb8_index(n, scb, sy, sx, sC8, g.H, g.W));
        }
      }
      in_s[cb * PLANE + rem] = v;
    }
    // ---- stage this chunk's slice of the bank: [step][nt][lane]
    {
      const uint4* bsrc = reinterpret_cast<const uint4*>(bank) + (size_t)ck * STEPS * ntiles_total * 64;
      for (int i = threadIdx.x; i < W_SLOTS; i += 256) {
        int ln = i & 63;
        int r = i >> 6;
        int t = r % NT, s = r / NT;
        w_s[i] = bsrc[((size_t)s * ntiles_total + ntile0 + t) * 64 + ln];
      }
    }
    __syncthreads();
    // ---- MFMA loop
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      bf16x8 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const bf16x8*>(&w_s[(s * NT + t) * 64 + lane]);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int mt = wave * MT + i;
        int row = mt / MTILES_X, col0 = (mt % MTILES_X) * 16;
        bf16x8 a = *reinterpret_cast<const bf16x8*>(&in_s[row * TIW + col0 + m + aoff[s]]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[t], acc[i][t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, statistics, transpose through LDS, 16-byte stores
  __syncthreads();
  bf16_t* out_s = reinterpret_cast<bf16_t*>(lds);               // [TH*TW][NT*16]
  float* out_f = reinterpret_cast<float*>(lds);
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int co = (ntile0 + t) * 16 + m;                             // C/D: col = lane & 15
    float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int mt = wave * MT + i;
      int row = mt / MTILES_X, col0 = (mt % MTILES_X) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int px = col0 + gq * 4 + r;                             // C/D: row = (lane >> 4) * 4 + reg
        float v = acc[i][t][r] + bv;
        bool valid = (ty0 + row) < g.Ho && (tx0 + px) < g.Wo && co < g.Cout;
        if (valid) { s1 += v; s2 += v * v; }
        if (OUT_F32) out_f[(row * TW + px) * 16 + m] = co < g.Cout ? v : 0.f;
        else out_s[(row * TW + px) * (NT * 16) + t * 16 + m] = f2bf(co < g.Cout ? v : 0.f);
      }
    }
    ssum[t] = s1; ssq[t] = s2;
  }
  if (part) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s1 = ssum[t], s2 = ssq[t];
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lane < 16) { red[wave][(t * 16 + lane) * 2] = s1; red[wave][(t * 16 + lane) * 2 + 1] = s2; }
    }
  }
  __syncthreads();
  if (part && threadIdx.x < NT * 32) {
    int co = ntile0 * 16 + (threadIdx.x >> 1);
    if (co < g.CoutP) {
      float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      part[(((size_t)n * g.tiles + tile) * g.CoutP + co) * 2 + (threadIdx.x & 1)] = r;
    }
  }
  if (OUT_F32) {
    float* yf = reinterpret_cast<float*>(y0);
    for (int i = threadIdx.x; i < TH * TW * 4; i += 256) {      // 16-byte pieces: 4 per pixel (16 f32 channels)
      int q = i & 3, pix = i >> 2;
      int row = pix / TW, px = pix % TW;
      int oy = ty0 + row, ox = tx0 + px;
      int cob = q >> 1;
      if (oy < g.Ho && ox < g.Wo && cob < g.CBout)
        *reinterpret_cast<uint4*>(yf + cb8_index(n, cob, oy, ox, g.CBout, g.Ho, g.Wo) + (q & 1) * 4) =
            reinterpret_cast<const uint4*>(out_f)[pix * 4 + q];
    }
    return;
  }
  for (int i = threadIdx.x; i < TH * TW * NT * 2; i += 256) {
    int cbl = i % (NT * 2);
    int pix = i / (NT * 2);
    int row = pix / TW, px = pix % TW;
    int oy = ty0 + row, ox = tx0 + px;
    int cob = ntile0 * 2 + cbl;
    if (oy < g.Ho && ox < g.Wo && cob < g.CBout) {
      uint4 v = reinterpret_cast<const uint4*>(out_s)[pix * (NT * 2) + cbl];
      if (g.split8 > 0 && cob >= g.split8)
        *reinterpret_cast<uint4*>(y1 + cb8_index(n, cob - g.split8, oy, ox, g.CBout - g.split8, g.Ho, g.Wo)) = v;
      else
        *reinterpret_cast<uint4*>(y0 + cb8_index(n, cob, oy, ox, g.split8 > 0 ? g.split8 : g.CBout, g.Ho, g.Wo)) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// filter gradient on the matrix cores:
//   dW[co][ci][ky][kx] = sum_{n,oy,ox} dy[n][co][oy][ox] * xpad[n][ci][oy+ky][ox+kx]
// as D[i = co][j = ci] += A[i][k] B[k][j] with k = 32 consecutive output pixels of one row, one MFMA per
// (tap, co-tile) and k-group.  Both operands need the PIXEL index on the MFMA k axis while the CB8 tiles in
// LDS hold 8 CHANNELS per 16-byte vector: the transposition is done by ds_read_b64_tr_b16 (4 pixels x 16
// channels per 16-lane group, delivered channel-per-lane) — two reads per fragment, any tap shift stays
// 16-byte aligned because a pixel step is a whole vector.  Plane strides are == 4 (mod 16) slots so the
// two channel-block planes of a read land on disjoint banks.
//
// Block (G, chunk, co-group): loops over (image, 16x32-pixel tile) work items; wave w owns taps w, w+4, ...
// (7/6/6/6 of 25) for NTW co-tiles; wave 3 also accumulates the bias gradient through an all-ones B fragment.
// Partials: [G][CoutP][CinP*K*K + 1] f32, combined deterministically by k_wgrad_finalize.
// ------------------------------------------------------------------------------------------------
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

__device__ __forceinline__ bf16x8 tr_frag(const short* base_lo) {
  // base_lo: this lane's address for pixels 8g..8g+3; pixels 8g+4..8g+7 are 4 slots (64 bytes) further
  v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)base_lo);
  v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(base_lo + 32));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int WTH = 16, WTW = 32;      // work-item tile (output pixels)

template <int K, int NTW>
__global__ __launch_bounds__(256) void k_wgrad_mfma_bf16(ConvGeom g, const bf16_t* __restrict__ x0,
                                                         const bf16_t* __restrict__ x1, const bf16_t* __restrict__ dy,
                                                         float* __restrict__ part, int tiles_x, int tiles) {
  constexpr int KK = K * K;
  constexpr int TIH = WTH + K - 1, TIW = WTW + K - 1;
  constexpr int XPS = ((TIH * TIW + 15) / 16) * 16 + 4;       // x plane stride (slots), == 4 mod 16
  constexpr int DPS = WTH * WTW + 4;                          // dy plane stride (slots)
  constexpr int NTAP = (KK + 3) / 4;                          // taps per wave (upper bound)
  __shared__ uint4 xs[2 * XPS];
  __shared__ uint4 ds[NTW * 2 * DPS];
  const int chunk = blockIdx.y, cog = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = (lane & 15) >> 2, p = lane & 3, gq = lane >> 4;
  // per-lane short offsets (2-byte units) inside a plane pair for pixel 8g + q of a row start
  const int lane_x = ((p >> 1) * XPS + 8 * gq + q) * 8 + (p & 1) * 4;
  const int lane_d = ((p >> 1) * DPS + 8 * gq + q) * 8 + (p & 1) * 4;
  const short* xs_s = reinterpret_cast<const short*>(xs);
  const short* ds_s = reinterpret_cast<const short*>(ds);

  f32x4 acc[NTAP][NTW];
#pragma unroll
  for (int a = 0; a < NTAP; ++a)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  static_assert(3 + 4 * (NTAP - 1) >= KK, "wave 3 needs a free accumulator slot for the bias gradient");
  const bool do_bias = (wave == 3) && (chunk == 0);
  const bf16x8 ones = (bf16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  const int work = g.N * tiles;
  for (int wi = blockIdx.x; wi < work; wi += gridDim.x) {
    const int n = wi / tiles, tile = wi % tiles;
    const int ty0 = (tile / tiles_x) * WTH, tx0 = (tile % tiles_x) * WTW;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * TIH * TIW; i += 256) {
      int cb = i / (TIH * TIW);
      int rem = i - cb * (TIH * TIW);
      int r = rem / TIW, c = rem - r * TIW;
      int gcb = chunk * 2 + cb;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gcb < g.CBin) {
        int sy = pad_map(ty0 + r - g.pad, g.H, g.pad_mode), sx = pad_map(tx0 + c - g.pad, g.W, g.pad_mode);
        if (sy >= 0 && sx >= 0) {
          const bf16_t* src = gcb < g.CB0 ? x0 : x1;
          int scb = gcb < g.CB0 ? gcb : gcb - g.CB0;
          int sC8 = gcb < g.CB0 ? g.CB0 : g.CB1;
          v = *reinterpret_cast<const uint4*>(src + cb8_index(n, scb, sy, sx, sC8, g.H, g.W));
        }
      }
      xs[cb * XPS + rem] = v;
    }
    for (int i = threadIdx.x; i < NTW * 2 * WTH * WTW; i += 256) {
      int pl = i / (WTH * WTW);                     // plane = co-tile * 2 + half
      int rem = i - pl * (WTH * WTW);
      int r = rem / WTW, c = rem - r * WTW;
      int cob = (cog * NTW) * 2 + pl;
      int oy = ty0 + r, ox = tx0 + c;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (cob < g.CBout && oy < g.Ho && ox < g.Wo)
        v = *reinterpret_cast<const uint4*>(dy + cb8_index(n, cob, oy, ox, g.CBout, g.Ho, g.Wo));
      ds[pl * DPS + rem] = v;
    }
    __syncthreads();
    for (int row = 0; row < WTH; ++row) {
      bf16x8 a[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) a[t] = tr_frag(ds_s + (t * 2 * DPS + row * WTW) * 8 + lane_d);
#pragma unroll
      for (int ti = 0; ti < NTAP; ++ti) {
        int tap = wave + 4 * ti;
        if (tap < KK) {
          int ky = tap / K, kx = tap % K;
          bf16x8 b = tr_frag(xs_s + ((row + ky) * TIW + kx) * 8 + lane_x);
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[ti][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[ti][t], 0, 0, 0);
        }
      }
      if (do_bias) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          acc[NTAP - 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], ones, acc[NTAP - 1][t], 0, 0, 0);
      }
    }
  }
  // ---- write this block's partial slab
  const int cols = g.CinP * KK + 1;
  float* pb = part + (size_t)blockIdx.x * g.CoutP * cols;
  const int cip = chunk * 16 + (lane & 15);
#pragma unroll
  for (int ti = 0; ti < NTAP; ++ti) {
    int tap = wave + 4 * ti;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int co = (cog * NTW + t) * 16 + gq * 4 + r;
        if (co >= g.CoutP) continue;
        if (tap < KK) {
          if (cip < g.CinP) pb[(size_t)co * cols + (size_t)cip * KK + tap] = acc[ti][t][r];
        } else if (do_bias && ti == NTAP - 1 && (lane & 15) == 0) {
          pb[(size_t)co * cols + (size_t)g.CinP * KK] = acc[ti][t][r];
        }
      }
  }
}

inline Bf16Cfg cfg_for(int c_out) {
  int ntiles = (c_out + 15) / 16;
  int nt = pick_nt(ntiles);
  if (nt == 1) return {32, 32, 1, 16};
  if (nt == 2) return {16, 32, 2, 8};
  return {16, 16, 4, 4};
}

}  // namespace

int mc_bf16_tile(const mc_conv_desc* d, int* th, int* tw) {
  Bf16Cfg c = cfg_for(d->c_out);
  *th = c.th; *tw = c.tw;
  return MC_OK;
}

static void bank_dims(const ConvGeom& g, int dgrad, int& chunks, int& steps, int& ntiles) {
  // forward conv: K over padded C_in, N over C_out; dgrad conv: K over C_out, N over padded C_in
  int cb_in = dgrad ? g.CBout : g.CBin;
  int c_out = dgrad ? g.CinP : g.Cout;
  chunks = (cb_in + CHUNK_CB - 1) / CHUNK_CB;
  steps = g.K == 5 ? KSteps<5>::steps : KSteps<3>::steps;
  int nt_total = (c_out + 15) / 16;
  int nt = pick_nt(nt_total);
  ntiles = (nt_total + nt - 1) / nt * nt;
}

size_t mc_bf16_bank_bytes(const ConvGeom& g, int dgrad) {
  int chunks, steps, ntiles;
  bank_dims(g, dgrad, chunks, steps, ntiles);
  return (size_t)chunks * steps * ntiles * 64 * 16;
}

int mc_bf16_pack(const ConvGeom& g, const float* w, int dgrad, void* packed, hipStream_t s) {
  int chunks, steps, ntiles;
  bank_dims(g, dgrad, chunks, steps, ntiles);
  size_t total = (size_t)chunks * steps * ntiles * 64 * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_bf16, dim3(blocks), dim3(256), 0, s, g, w, dgrad, (bf16_t*)packed, chunks, steps, ntiles);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

const char* mc_bf16_kernel_name(const ConvGeom& g) {
  Bf16Cfg c = cfg_for(g.Cout);
  if (g.out_f32) return g.K == 5 ? "k_conv_mfma_bf16<5,32,32,1,16,true>" : "k_conv_mfma_bf16<3,32,32,1,16,true>";
  if (g.K == 5) return c.nt == 1 ? "k_conv_mfma_bf16<5,32,32,1,16>" : (c.nt == 2 ? "k_conv_mfma_bf16<5,16,32,2,8>" : "k_conv_mfma_bf16<5,16,16,4,4>");
  return c.nt == 1 ? "k_conv_mfma_bf16<3,32,32,1,16>" : (c.nt == 2 ? "k_conv_mfma_bf16<3,16,32,2,8>" : "k_conv_mfma_bf16<3,16,16,4,4>");
}

int mc_conv2d_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                   void* y1, float* part, hipStream_t s) {
  Bf16Cfg c = cfg_for(g.Cout);
  int nt_total = (g.Cout + 15) / 16;
  int groups = (nt_total + c.nt - 1) / c.nt;
  dim3 grid(g.tiles, groups, g.N);
#define LAUNCH(K, TH, TW, NT, MT)                                                                                    \
  hipLaunchKernelGGL((k_conv_mfma_bf16<K, TH, TW, NT, MT>), grid, dim3(256), 0, s, g, (const bf16_t*)x0,             \
                     (const bf16_t*)x1, (const bf16_t*)bank, bias, (bf16_t*)y0, (bf16_t*)y1, part, groups)
  if (g.out_f32) {
    if (g.K == 5) hipLaunchKernelGGL((k_conv_mfma_bf16<5, 32, 32, 1, 16, true>), grid, dim3(256), 0, s, g, (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)bank, bias, (bf16_t*)y0, (bf16_t*)y1, part, groups);
    else hipLaunchKernelGGL((k_conv_mfma_bf16<3, 32, 32, 1, 16, true>), grid, dim3(256), 0, s, g, (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)bank, bias, (bf16_t*)y0, (bf16_t*)y1, part, groups);
  } else if (g.K == 5) {
    if (c.nt == 1) LAUNCH(5, 32, 32, 1, 16); else if (c.nt == 2) LAUNCH(5, 16, 32, 2, 8); else LAUNCH(5, 16, 16, 4, 4);
  } else if (g.K == 3) {
    if (c.nt == 1) LAUNCH(3, 32, 32, 1, 16); else if (c.nt == 2) LAUNCH(3, 16, 32, 2, 8); else LAUNCH(3, 16, 16, 4, 4);
  } else {
    return MC_EUNSUPPORTED;
  }
#undef LAUNCH
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_wgrad_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, hipStream_t s) {
  const int tiles_x = (g.Wo + WTW - 1) / WTW, tiles_y = (g.Ho + WTH - 1) / WTH;
  const int tiles = tiles_x * tiles_y;
  const int ntiles = (g.Cout + 15) / 16;
  const int ntw = pick_nt(ntiles) >= 2 ? 2 : 1;      // two co-tiles per block keep LDS at 55 KB (2-3 blocks per CU)
  dim3 grid(g.wgrad_G, (g.CBin + 1) / 2, (ntiles + ntw - 1) / ntw);
#define WLAUNCH(K, NTW)                                                                                              \
  hipLaunchKernelGGL((k_wgrad_mfma_bf16<K, NTW>), grid, dim3(256), 0, s, g, (const bf16_t*)x0, (const bf16_t*)x1,     \
                     (const bf16_t*)dy, (float*)part, tiles_x, tiles)
  if (g.K == 5) { if (ntw == 1) WLAUNCH(5, 1); else WLAUNCH(5, 2); }
  else if (g.K == 3) { if (ntw == 1) WLAUNCH(3, 1); else WLAUNCH(3, 2); }
  else return MC_EUNSUPPORTED;
#undef WLAUNCH
  MC_CHECK_LAUNCH();
  return MC_OK;
}
