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
// filter gradient (first version): tile-local f32 reduction on the vector ALU from bf16 tiles.
// Same decomposition and partial layout as the f32 path (conv_f32.hip).  [MFMA version: see k_wgrad_mfma]
// ------------------------------------------------------------------------------------------------
constexpr int TS = 16;
template <int K>
__global__ __launch_bounds__(256) void k_wgrad_direct_bf16(ConvGeom g, const bf16_t* __restrict__ x0,
                                                           const bf16_t* __restrict__ x1, const bf16_t* __restrict__ dy,
                                                           float* __restrict__ part) {
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
  const bf16_t* src = cb < g.CB0 ? x0 : x1;
  const int scb = cb < g.CB0 ? cb : cb - g.CB0;
  const int sC8 = cb < g.CB0 ? g.CB0 : g.CB1;
  const int work = g.N * g.tiles;
  for (int wi = blockIdx.x; wi < work; wi += gridDim.x) {
    const int n = wi / g.tiles, tile = wi % g.tiles;
    const int ty0 = (tile / g.tiles_x) * TS, tx0 = (tile % g.tiles_x) * TS;
    __syncthreads();
    for (int i = t; i < TI * TI; i += 256) {
      int r = i / TI, c = i % TI;
      int sy = pad_map(ty0 + r - g.pad, g.H, g.pad_mode), sx = pad_map(tx0 + c - g.pad, g.W, g.pad_mode);
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (sy >= 0 && sx >= 0) V8<bf16_t>::ld(src + cb8_index(n, scb, sy, sx, sC8, g.H, g.W), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[i][j] = v[j];
    }
    {
      int r = t >> 4, c = t & 15;
      int oy = ty0 + r, ox = tx0 + c;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        int cob = cog * 2 + half;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (oy < g.Ho && ox < g.Wo && cob < g.CBout) V8<bf16_t>::ld(dy + cb8_index(n, cob, oy, ox, g.CBout, g.Ho, g.Wo), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) dys[t][half * 8 + j] = v[j];
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
  const int cols = g.CinP * K * K + 1;
  float* pb = part + (size_t)blockIdx.x * g.CoutP * cols;
  if (wthread) {
    int cig = cb * 8 + ci;
#pragma unroll
    for (int co = 0; co < 16; ++co)
      if (co0 + co < g.CoutP) pb[(size_t)(co0 + co) * cols + (size_t)cig * K * K + tap] = acc[co];
  } else if (bthread) {
    int co = co0 + t - 8 * K * K;
    if (co < g.CoutP) pb[(size_t)co * cols + (size_t)g.CinP * K * K] = bacc;
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
  // NOTE: the filter-gradient partial layout assumes 16x16 tiles; recompute the tiling for that
  ConvGeom w = g;
  w.tiles_x = (g.Wo + TS - 1) / TS; w.tiles_y = (g.Ho + TS - 1) / TS; w.tiles = w.tiles_x * w.tiles_y;
  long work = (long)w.N * w.tiles;
  if (w.wgrad_G > work) w.wgrad_G = (int)work;
  dim3 grid(w.wgrad_G, g.CBin, cdiv(g.CoutP, 16));
  if (g.K == 5)
    hipLaunchKernelGGL(k_wgrad_direct_bf16<5>, grid, dim3(256), 0, s, w, (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)dy, (float*)part);
  else if (g.K == 3)
    hipLaunchKernelGGL(k_wgrad_direct_bf16<3>, grid, dim3(256), 0, s, w, (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)dy, (float*)part);
  else
    return MC_EUNSUPPORTED;
  MC_CHECK_LAUNCH();
  return MC_OK;
}
