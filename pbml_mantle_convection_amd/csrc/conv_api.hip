// C-ABI dispatch for the convolution entry points + filter-bank packing + filter-gradient combine.
#include "conv_rr.h"

int mc_conv2d_f32(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                  void* y1, float* part, const ConvFuse& fz, int fuse, hipStream_t s);
int mc_wgrad_f32(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, const ConvFuse& fz, int fuse,
                 hipStream_t s);
// bf16 MFMA path (conv_bf16.hip)
int mc_bf16_tile(const mc_conv_desc* d, int* th, int* tw);
size_t mc_bf16_bank_bytes(const ConvGeom& g, int dgrad);
int mc_bf16_pack(const ConvGeom& g, const float* w_unique, int dgrad, void* packed, hipStream_t s);
int mc_conv2d_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                   void* y1, float* part, const ConvFuse& fz, int fuse, hipStream_t s);
int mc_wgrad_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, const ConvFuse& fz, int fuse,
                  hipStream_t s);
const char* mc_bf16_kernel_name(const ConvGeom& g);
void mc_bf16_bank_dims(const ConvGeom& g, int dgrad, int& chunks, int& steps, int& ntiles);
// row-reuse bf16 path for single-output-tile layers (conv_rr_bf16.hip)
bool mc_rr_applies(int dtype, int cout, int wo, bool full_pad);
size_t mc_rr_bank_bytes(const ConvGeom& g, int dgrad);
int mc_rr_pack(const ConvGeom& g, const float* w_unique, int dgrad, void* packed, hipStream_t s);
int mc_conv2d_rr(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0, void* y1,
                 float* part, const ConvFuse& fz, int fuse, hipStream_t s);
const char* mc_rr_kernel_name(const ConvGeom& g, int fuse);

namespace {

// forward / input-gradient kernel family of a descriptor: the row-reuse kernel takes the layers with one 16-channel
// output tile per work-group (an odd number of output tiles), the wide-tile kernel the rest
bool rr_desc(const mc_conv_desc* d) {
  return mc_rr_applies(d->dtype, d->c_out, d->w + 2 * d->pad - d->k + 1, d->pad == d->k - 1);
}

int geom_for(const mc_conv_desc* d, ConvGeom& g) {
  if (!d) return MC_EINVAL;
  int th = 16, tw = 16;
  if (mc_is16(d->dtype)) {
    if (rr_desc(d)) { th = RR_R; tw = RR_TW; }
    else {
      int rc = mc_bf16_tile(d, &th, &tw);
      if (rc) return rc;
    }
  } else if (d->dtype != MC_F32) {        // (MC_BF16 and MC_MIX16 share every tile shape and bank size)
    return MC_EUNSUPPORTED;
  }
  return conv_geom(d, th, tw, g);
}
// partial-sum slots per sample the forward / input-gradient kernel writes (one per tile; per (tile, strip) for row reuse)
int stat_slots(const mc_conv_desc* d, const ConvGeom& g) { return rr_desc(d) ? g.tiles * RR_STRIPS : g.tiles; }

// f32 bank: [cbin][tap][ci8][CoutP]
__global__ void k_pack_f32(ConvGeom g, const float* __restrict__ wu, int dgrad, float* __restrict__ bank, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    bank[i] = pack_value_f32(g, wu, i, dgrad);
}

// ---- batched packing: the job table travels by value in the kernel arguments (<= PK_MAX jobs per launch)
struct PkJob {
  int K, Cout, CBin, CB0, Cin0, Cin1, Cin, U, nh, nv, nq, CBout, CinP, CoutP;
  int dgrad, steps, ntiles, bf16, f16, rr, first_block;
  unsigned total;
  const float* w;
  void* out;
};
constexpr int PK_MAX = 24;
struct PkTable { int n; PkJob j[PK_MAX]; };

__global__ void k_pack_batched(PkTable t) {
  int ji = 0;
#pragma unroll 1
  for (int k = 1; k < t.n; ++k) if ((int)blockIdx.x >= t.j[k].first_block) ji = k;
  const PkJob& job = t.j[ji];
  const int nblk = (ji + 1 < t.n ? t.j[ji + 1].first_block : (int)gridDim.x) - job.first_block;
  for (size_t i = (size_t)(blockIdx.x - job.first_block) * blockDim.x + threadIdx.x; i < job.total; i += (size_t)nblk * blockDim.x) {
    if (job.rr || job.bf16) {
      const float v = job.rr ? rr_pack_value(job, job.w, i, job.dgrad, job.ntiles)
                             : pack_value_bf16(job, job.w, i, job.dgrad, job.steps, job.ntiles);
      reinterpret_cast<bf16_t*>(job.out)[i] = job.f16 ? __builtin_bit_cast(bf16_t, (_Float16)v) : f2bf(v);
    } else {
      reinterpret_cast<float*>(job.out)[i] = pack_value_f32(job, job.w, i, job.dgrad);
    }
  }
}

// dW_unique[u][ci][ky][kx] += sum_G part[G](tap, ci, u) + the same sum over every mirrored copy of u at its mirrored tap
// 64 outputs per block; the four waves split the slab range and combine through LDS (deterministic order)
struct WfJob {
  int K, U, nh, nv, nq, Cin, Cin0, CB0, CinP, CoutP, Cout, G, first_block;
  const float* part;
  float* dw;
  float* db;
};
constexpr int WF_MAX = 32;
struct WfTable { int n; WfJob j[WF_MAX]; };

constexpr int WF_WAVES = 16;      // waves per block: each sums every 16th slab (the loop is bound by the latency of its
                                  // strided loads: 4 waves per block took 129 us on a level-0 layer's 768 slabs)
__device__ __forceinline__ void wgrad_finalize_block(const WfJob& g, int lblock) {
  // outputs are enumerated in slab order [tap][chunk][u][16] (coalesced reads of every slab), then the bias entries
  const int K = g.K, KK = K * K;
  const int nch = wg_chunks(g.CinP);
  const size_t slab = wg_slab_floats(g.CoutP, g.CinP, KK);
  const size_t nW = (size_t)KK * nch * g.U * 16;
  const size_t total = nW + g.Cout;
  const int e = threadIdx.x & 63, w = threadIdx.x >> 6, NWV = (int)blockDim.x >> 6;   // 16 or 4 waves per block
  const size_t i = (size_t)lblock * 64 + e;
  __shared__ float red[WF_WAVES][64];
  float s = 0.f;
  long dst = -1;                  // index into dw (>= 0), or -2 - co for the bias
  if (i < total) {
    size_t o1, ox[3] = {0, 0, 0};
    int nx = 0;                      // mirrored copies of this unique filter (0, 1 or 3)
    bool live = true;
    if (i < nW) {
      int cl = (int)(i & 15);
      size_t r = i >> 4;
      int u = (int)(r % g.U); r /= g.U;
      int chunk = (int)(r % nch);
      int tap = (int)(r / nch);
      int cip = chunk * 16 + cl;
      int ci = cip < g.CB0 * 8 ? cip : g.Cin0 + (cip - g.CB0 * 8);
      live = cip < g.CB0 * 8 ? cip < g.Cin0 : ci < g.Cin;
      int ky = tap / K, kx = tap - ky * K;
      o1 = wg_index(tap, cip, u, g.CoutP, nch);
      const int tfx = ky * K + (K - 1 - kx), tfy = (K - 1 - ky) * K + kx, tfxy = (K - 1 - ky) * K + (K - 1 - kx);
      if (u < g.nh) { nx = 1; ox[0] = wg_index(tfx, cip, g.U + u, g.CoutP, nch); }
      else if (u < g.nh + g.nv) { nx = 1; ox[0] = wg_index(tfy, cip, g.U + u, g.CoutP, nch); }
      else if (u < g.nh + g.nv + g.nq) {
        const int c0 = g.U + g.nh + g.nv + (u - g.nh - g.nv);
        nx = 3;
        ox[0] = wg_index(tfx, cip, c0, g.CoutP, nch); ox[1] = wg_index(tfy, cip, c0 + g.nq, g.CoutP, nch);
        ox[2] = wg_index(tfxy, cip, c0 + 2 * g.nq, g.CoutP, nch);
      }
      dst = ((long)u * g.Cin + ci) * KK + tap;
    } else {
      o1 = (size_t)KK * nch * g.CoutP * 16 + (i - nW);
      dst = -2 - (long)(i - nW);
    }
    if (live) {
      const float* part = g.part;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int G = w;
      for (; G + 3 * NWV < g.G; G += 4 * NWV) {
        a0 += part[(size_t)G * slab + o1]; a1 += part[(size_t)(G + NWV) * slab + o1];
        a2 += part[(size_t)(G + 2 * NWV) * slab + o1]; a3 += part[(size_t)(G + 3 * NWV) * slab + o1];
        for (int m = 0; m < nx; ++m) {
          a0 += part[(size_t)G * slab + ox[m]]; a1 += part[(size_t)(G + NWV) * slab + ox[m]];
          a2 += part[(size_t)(G + 2 * NWV) * slab + ox[m]]; a3 += part[(size_t)(G + 3 * NWV) * slab + ox[m]];
        }
      }
      for (; G < g.G; G += NWV) {
        a0 += part[(size_t)G * slab + o1];
        for (int m = 0; m < nx; ++m) a0 += part[(size_t)G * slab + ox[m]];
      }
      s = (a0 + a1) + (a2 + a3);
    } else {
      dst = -1;
    }
  }
  red[w][e] = s;
  __syncthreads();
  if (w == 0 && dst != -1) {
    float r = 0.f;
    for (int k = 0; k < NWV; k += 4) r += (red[k][e] + red[k + 1][e]) + (red[k + 2][e] + red[k + 3][e]);
    if (dst >= 0) { if (g.dw) g.dw[dst] += r; }
    else if (g.db) g.db[-2 - dst] += r;
  }
}

__global__ __launch_bounds__(64 * WF_WAVES) void k_wgrad_finalize(WfTable t) {
  int ji = 0;
#pragma unroll 1
  for (int k = 1; k < t.n; ++k) if ((int)blockIdx.x >= t.j[k].first_block) ji = k;
  wgrad_finalize_block(t.j[ji], blockIdx.x - t.j[ji].first_block);
}

int wf_fill(const mc_conv_desc* d, const void* partials, float* dw, float* db, WfJob& j, int& blocks) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!partials || (!dw && !db)) return MC_EINVAL;
  j.K = g.K; j.U = g.U; j.Cin = g.Cin; j.Cin0 = g.Cin0; j.CB0 = g.CB0; j.CinP = g.CinP; j.CoutP = g.CoutP; j.Cout = g.Cout;
  j.nh = g.nh; j.nv = g.nv; j.nq = g.nq; j.G = g.wgrad_G; j.part = (const float*)partials; j.dw = dw; j.db = db;
  size_t total = (size_t)g.K * g.K * wg_chunks(g.CinP) * g.U * 16 + g.Cout;
  blocks = (int)((total + 63) / 64);
  return MC_OK;
}

}  // namespace

// The input-gradient bank of a layer is consumed by the kernel family of the INPUT-GRADIENT convolution (whose output
// channels are this layer's input channels), the forward bank by this layer's own family.
static bool bank_is_rr(const ConvGeom& g, int dgrad) {
  // (the input-gradient convolution runs on the padded domain of this layer's input: width W + 2 pad)
  return dgrad ? mc_rr_applies(g.dtype, g.Cin, g.W + 2 * g.pad, true) : mc_rr_applies(g.dtype, g.Cout, g.Wo, g.pad == g.K - 1);
}

extern "C" {

int mc_version(void) { return 200; }

const char* mc_strerror(int code) {
  switch (code) {
    case MC_OK: return "ok";
    case MC_EINVAL: return "invalid argument or shape";
    case MC_EUNSUPPORTED: return "unsupported configuration";
    case MC_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

size_t mc_packed_weight_bytes(const mc_conv_desc* d, int32_t dgrad) {
  ConvGeom g;
  if (geom_for(d, g)) return 0;
  if (bank_is_rr(g, dgrad)) return mc_rr_bank_bytes(g, dgrad);
  if (mc_is16(g.dtype)) return mc_bf16_bank_bytes(g, dgrad);
  int cbin = dgrad ? g.CBout : g.CBin, cop = dgrad ? g.CinP : g.CoutP;
  return (size_t)cbin * g.K * g.K * 8 * cop * sizeof(float);
}

int mc_pack_weights(const mc_conv_desc* d, const float* w_unique, int32_t dgrad, void* packed, void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!w_unique || !packed) return MC_EINVAL;
  if (bank_is_rr(g, dgrad)) return mc_rr_pack(g, w_unique, dgrad, packed, (hipStream_t)stream);
  if (mc_is16(g.dtype)) return mc_bf16_pack(g, w_unique, dgrad, packed, (hipStream_t)stream);
  size_t total = mc_packed_weight_bytes(d, dgrad) / sizeof(float);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, w_unique, dgrad, (float*)packed, total);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

const char* mc_conv_kernel_name(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return "unsupported";
  if (rr_desc(d)) return mc_rr_kernel_name(g, 0);
  if (mc_is16(g.dtype)) return mc_bf16_kernel_name(g);
  return g.K == 5 ? "k_conv_direct_f32<5>" : "k_conv_direct_f32<3>";
}

int32_t mc_conv_tiles(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return -1;
  return stat_slots(d, g);
}

static bool act_ok(int a) { return a >= MC_ACT_NONE && a <= MC_ACT_ELU; }

// fills the prologue half of fz; returns 1 when a source really needs the transform
static int fill_prologue(const mc_conv_prologue* pro, ConvFuse& fz, int& rc) {
  rc = MC_OK;
  if (!pro) return 0;
  if (!act_ok(pro->act0) || !act_ok(pro->act1)) { rc = MC_EINVAL; return 0; }
  fz.coef0 = pro->coef0; fz.coef1 = pro->coef1; fz.act0 = pro->act0; fz.act1 = pro->act1;
  return (pro->coef0 || pro->coef1 || pro->act0 != MC_ACT_NONE || pro->act1 != MC_ACT_NONE) ? 1 : 0;
}

int mc_conv2d_fused(const mc_conv_desc* d, const void* x0, const void* x1, const mc_conv_prologue* pro,
                    const void* packed_w, const float* bias, void* y0, void* y1, float* stat_partials,
                    const mc_conv_epilogue* epi, void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!x0 || !packed_w || !y0 || (g.Cin1 > 0 && !x1) || (g.split8 > 0 && !y1)) return MC_EINVAL;
  ConvFuse fz = conv_fuse_none();
  int fuse = fill_prologue(pro, fz, rc);
  if (rc) return rc;
  if (epi) {
    // input-gradient launch on the padded domain of a (hs x ws) tensor with a single, unsplit output
    if (fuse) return MC_EUNSUPPORTED;                       // (the gradient tensor dY is never a raw conv output)
    if (!epi->y || !epi->partials || !act_ok(epi->act) || epi->pad < 0 || epi->hs <= 0 || epi->ws <= 0) return MC_EINVAL;
    if (epi->pad_mode < MC_PAD_ZEROS || epi->pad_mode > MC_PAD_REFLECT) return MC_EINVAL;
    if (g.split8 > 0 || stat_partials || g.out_f32) return MC_EUNSUPPORTED;
    if (g.Ho != epi->hs + 2 * epi->pad || g.Wo != epi->ws + 2 * epi->pad) return MC_EINVAL;
    fz.ey = epi->y; fz.ecoef = epi->coef; fz.epart = epi->partials; fz.eact = epi->act; fz.epad = epi->pad;
    fz.ezero = epi->pad_mode == MC_PAD_ZEROS || epi->pad == 0; fz.ehs = epi->hs; fz.ews = epi->ws;
    if (epi->y_f16 && g.dtype != MC_BF16) return MC_EINVAL;   // (an f16 y belongs to the bf16 input-gradient launch of an MC_MIX16 layer)
    fz.ey16 = epi->y_f16 ? 1 : 0;
    if (epi->part_stride < stat_slots(d, g)) return MC_EINVAL;
    fz.estride = epi->part_stride;
    fuse = 2;
  }
  if (rr_desc(d)) return mc_conv2d_rr(g, x0, x1, packed_w, bias, y0, y1, stat_partials, fz, fuse, (hipStream_t)stream);
  if (mc_is16(g.dtype)) return mc_conv2d_bf16(g, x0, x1, packed_w, bias, y0, y1, stat_partials, fz, fuse, (hipStream_t)stream);
  return mc_conv2d_f32(g, x0, x1, packed_w, bias, y0, y1, stat_partials, fz, fuse, (hipStream_t)stream);
}

int mc_conv2d(const mc_conv_desc* d, const void* x0, const void* x1, const void* packed_w, const float* bias, void* y0,
              void* y1, float* stat_partials, void* stream) {
  return mc_conv2d_fused(d, x0, x1, nullptr, packed_w, bias, y0, y1, stat_partials, nullptr, stream);
}

// One byte past the highest address a forward launch of `d` reads through `packed_w`, restated on the host from the indexing
// of the kernel family that takes the launch (round 2's memory fault: the f32 kernel read 32 B past a bank whose padded
// output-channel count is not a multiple of 16; nothing compared the kernels' reach with mc_packed_weight_bytes).
size_t mc_conv_bank_read_extent(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return 0;
  const int KK = g.K * g.K;
  if (rr_desc(d)) {
    // bank [chunk][output tile][fragment][lane][8]: a work-group column (grid.y) reads the NFRAG * 64 16-byte slots of its
    // (chunk, output tile); chunks = ceil(CBin / 2), grid.y = ceil(Cout / 16)
    const size_t nfrag = g.K == 5 ? RR<5>::NFRAG : RR<3>::NFRAG;
    const size_t chunks = (g.CBin + 1) / 2, groups = (g.Cout + 15) / 16;
    return (((chunks - 1) * groups + (groups - 1)) * nfrag * 64 + (nfrag * 64 - 1)) * 16 + 16;
  }
  if (mc_is16(g.dtype)) {
    // bank [chunk][K-step][N-tile][lane][8]; grid.y = ceil(n_tiles / NT) work-group columns of NT tiles each
    int chunks, steps, ntiles;
    mc_bf16_bank_dims(g, 0, chunks, steps, ntiles);            // (ntiles = grid.y * NT: the launch's ntiles_total)
    const size_t last = ((size_t)(chunks - 1) * steps + (steps - 1)) * ntiles + (ntiles - 1);
    return (last * 64 + 63) * 16 + 16;
  }
  // f32: bank [cbin][tap][ci8][CoutP]; group cog reads columns co0 .. co0 + min(15, CoutP - 1 - co0)
  const int ncog = (g.CoutP + 15) / 16, co0 = 16 * (ncog - 1);
  const int comax = g.CoutP - 1 - co0;
  const size_t last = (((size_t)(g.CBin - 1) * KK + (KK - 1)) * 8 + 7) * g.CoutP + co0 + (comax >= 15 ? 15 : comax);
  return (last + 1) * sizeof(float);
}

size_t mc_wgrad_partial_bytes(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return 0;
  return (size_t)g.wgrad_G * wg_slab_floats(g.CoutP, g.CinP, g.K * g.K) * sizeof(float);
}

int mc_conv2d_wgrad_fused(const mc_conv_desc* d, const void* x0, const void* x1, const mc_conv_prologue* pro,
                          const void* dy, void* partials, void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!x0 || !dy || !partials || (g.Cin1 > 0 && !x1)) return MC_EINVAL;
  ConvFuse fz = conv_fuse_none();
  const int fuse = fill_prologue(pro, fz, rc);
  if (rc) return rc;
  if (mc_is16(g.dtype)) return mc_wgrad_bf16(g, x0, x1, dy, partials, fz, fuse, (hipStream_t)stream);
  return mc_wgrad_f32(g, x0, x1, dy, partials, fz, fuse, (hipStream_t)stream);
}

int mc_conv2d_wgrad(const mc_conv_desc* d, const void* x0, const void* x1, const void* dy, void* partials,
                    void* stream) {
  return mc_conv2d_wgrad_fused(d, x0, x1, nullptr, dy, partials, stream);
}

int mc_conv2d_wgrad_finalize_batched(const mc_conv_desc* descs, const void* const* partials, float* const* dw,
                                     float* const* db, int32_t n, void* stream) {
  if (!descs || !partials || !dw || !db || n <= 0) return MC_EINVAL;
  // two passes: layers with many small slabs (levels 0-1: 768 x 25 KB) want 16 waves per 64 outputs (each sums every 16th
  // slab: latency-bound otherwise), layers with few large slabs (deep levels: 16 x 1.6 MB) 4 waves (1024-thread blocks with
  // one or two loads per thread cost 338 us for the whole network; the order of the sums is fixed either way)
  for (int pass = 0; pass < 2; ++pass) {
    WfTable t;
    t.n = 0;
    int blocks = 0;
    auto flush = [&]() {
      if (t.n == 0) return;
      hipLaunchKernelGGL(k_wgrad_finalize, dim3(blocks), dim3(pass == 0 ? 64 * WF_WAVES : 256), 0, (hipStream_t)stream, t);
      t.n = 0; blocks = 0;
    };
    for (int k = 0; k < n; ++k) {
      WfJob j;
      int b = 0;
      int rc = wf_fill(&descs[k], partials[k], dw[k], db[k], j, b);
      if (rc) return rc;
      if ((j.G >= 128) != (pass == 0)) continue;
      j.first_block = blocks;
      t.j[t.n++] = j;
      blocks += b;
      if (t.n == WF_MAX) flush();
    }
    flush();
    MC_CHECK_LAUNCH();
  }
  return MC_OK;
}

int mc_conv2d_wgrad_finalize(const mc_conv_desc* d, const void* partials, float* dw_unique, float* dbias,
                             void* stream) {
  return mc_conv2d_wgrad_finalize_batched(d, &partials, &dw_unique, &dbias, 1, stream);
}

int mc_pack_weights_batched(const mc_conv_desc* descs, const float* const* w_unique, const int32_t* dgrad,
                            void* const* packed, int32_t n, void* stream) {
  if (!descs || !w_unique || !dgrad || !packed || n <= 0) return MC_EINVAL;
  for (int base = 0; base < n; base += PK_MAX) {
    PkTable t;
    t.n = n - base < PK_MAX ? n - base : PK_MAX;
    int blocks = 0;
    for (int k = 0; k < t.n; ++k) {
      ConvGeom g;
      int rc = geom_for(&descs[base + k], g);
      if (rc) return rc;
      if (!w_unique[base + k] || !packed[base + k]) return MC_EINVAL;
      PkJob& j = t.j[k];
      j.K = g.K; j.Cout = g.Cout; j.CBin = g.CBin; j.CB0 = g.CB0; j.Cin0 = g.Cin0; j.Cin1 = g.Cin1; j.Cin = g.Cin; j.U = g.U;
      j.nh = g.nh; j.nv = g.nv; j.nq = g.nq;
      j.CBout = g.CBout; j.CinP = g.CinP; j.CoutP = g.CoutP; j.dgrad = dgrad[base + k]; j.bf16 = mc_is16(g.dtype);
      j.f16 = (g.dtype == MC_MIX16 && !j.dgrad) ? 1 : 0;       // forward banks of MC_MIX16 are f16, input-gradient banks bf16
      j.rr = bank_is_rr(g, j.dgrad);
      size_t total;
      if (j.rr) {
        j.steps = 0;
        j.ntiles = ((j.dgrad ? g.CinP : g.Cout) + 15) / 16;
        total = mc_rr_bank_bytes(g, j.dgrad) / 2;
      } else if (j.bf16) {
        int chunks;
        mc_bf16_bank_dims(g, j.dgrad, chunks, j.steps, j.ntiles);
        total = (size_t)chunks * j.steps * j.ntiles * 64 * 8;
      } else {
        j.steps = 0; j.ntiles = 0;
        total = (size_t)(j.dgrad ? g.CBout : g.CBin) * g.K * g.K * 8 * (j.dgrad ? g.CinP : g.CoutP);
      }
      j.total = (unsigned)total;
      j.w = w_unique[base + k];
      j.out = packed[base + k];
      j.first_block = blocks;
      int b = (int)((total + 256 * 8 - 1) / (256 * 8));
      blocks += b < 1 ? 1 : b;
    }
    hipLaunchKernelGGL(k_pack_batched, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    MC_CHECK_LAUNCH();
  }
  return MC_OK;
}

}  // extern "C"
