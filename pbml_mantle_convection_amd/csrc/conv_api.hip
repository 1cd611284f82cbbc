// C-ABI dispatch for the convolution entry points + filter-bank packing + filter-gradient combine.
#include "conv_common.h"

int mc_conv2d_f32(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                  void* y1, float* part, hipStream_t s);
int mc_wgrad_f32(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, hipStream_t s);
// bf16 MFMA path (conv_bf16.hip)
int mc_bf16_tile(const mc_conv_desc* d, int* th, int* tw);
size_t mc_bf16_bank_bytes(const ConvGeom& g, int dgrad);
int mc_bf16_pack(const ConvGeom& g, const float* w_unique, int dgrad, void* packed, hipStream_t s);
int mc_conv2d_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                   void* y1, float* part, hipStream_t s);
int mc_wgrad_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, hipStream_t s);
const char* mc_bf16_kernel_name(const ConvGeom& g);

namespace {

int geom_for(const mc_conv_desc* d, ConvGeom& g) {
  if (!d) return MC_EINVAL;
  int th = 16, tw = 16;
  if (d->dtype == MC_BF16) {
    int rc = mc_bf16_tile(d, &th, &tw);
    if (rc) return rc;
  } else if (d->dtype != MC_F32) {
    return MC_EUNSUPPORTED;
  }
  return conv_geom(d, th, tw, g);
}

// f32 bank: [cbin][tap][ci8][CoutP]
__global__ void k_pack_f32(ConvGeom g, const float* __restrict__ wu, int dgrad, float* __restrict__ bank) {
  const int K = g.K, KK = K * K;
  if (!dgrad) {
    size_t total = (size_t)g.CBin * KK * 8 * g.CoutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      int co = (int)(i % g.CoutP);
      size_t r = i / g.CoutP;
      int j = (int)(r % 8); r /= 8;
      int tap = (int)(r % KK);
      int cb = (int)(r / KK);
      int ci = cb < g.CB0 ? cb * 8 + j : g.Cin0 + (cb - g.CB0) * 8 + j;
      bool ok = (cb < g.CB0 ? (cb * 8 + j < g.Cin0) : ((cb - g.CB0) * 8 + j < g.Cin1)) && co < g.Cout;
      float v = 0.f;
      if (ok) {
        int ky = tap / K, kx = tap % K;
        int u = co, kxs = kx;
        if (co >= g.U) { u = co - g.U; kxs = K - 1 - kx; }   // x-mirrored copy (symmetric_layers_torch.py:121-123)
        v = wu[(((size_t)u * g.Cin + ci) * K + ky) * K + kxs];
      }
      bank[i] = v;
    }
  } else {
    // dgrad conv: in-channels = forward c_out, out-channels = forward (padded) c_in, rotated taps
    const int CBd = g.CBout, CoP = g.CinP;
    size_t total = (size_t)CBd * KK * 8 * CoP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      int o = (int)(i % CoP);
      size_t r = i / CoP;
      int j = (int)(r % 8); r /= 8;
      int tap = (int)(r % KK);
      int cb = (int)(r / KK);
      int co = cb * 8 + j;                 // forward output channel
      int ob = o / 8, oj = o % 8;          // forward input channel (padded index)
      int ci = ob < g.CB0 ? ob * 8 + oj : g.Cin0 + (ob - g.CB0) * 8 + oj;
      bool ok = (ob < g.CB0 ? (ob * 8 + oj < g.Cin0) : ((ob - g.CB0) * 8 + oj < g.Cin1)) && co < g.Cout;
      float v = 0.f;
      if (ok) {
        int ky = K - 1 - tap / K, kx = K - 1 - tap % K;
        int u = co, kxs = kx;
        if (co >= g.U) { u = co - g.U; kxs = K - 1 - kx; }
        v = wu[(((size_t)u * g.Cin + ci) * K + ky) * K + kxs];
      }
      bank[i] = v;
    }
  }
}

// dW_unique[u][ci][ky][kx] += sum_G part[G][u][...] + (u < h/2) sum_G part[G][U+u][..][ky][K-1-kx]
// 64 outputs per block; the four waves split the slab range and combine through LDS (deterministic order)
__global__ __launch_bounds__(256) void k_wgrad_finalize(ConvGeom g, const float* __restrict__ part,
                                                        float* __restrict__ dw, float* __restrict__ db) {
  const int K = g.K, KK = K * K;
  const int cols = g.CinP * KK + 1;
  const size_t slab = (size_t)g.CoutP * cols;
  const size_t nW = (size_t)g.U * g.Cin * KK;
  const size_t total = nW + g.Cout;
  const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + e;
  __shared__ float red[4][64];
  float s = 0.f;
  if (i < total) {
    size_t o1, o2 = 0;
    bool two = false;
    if (i < nW) {
      int kx = (int)(i % K);
      size_t r = i / K;
      int ky = (int)(r % K); r /= K;
      int ci = (int)(r % g.Cin);
      int u = (int)(r / g.Cin);
      int cip = cin_padded_index(ci, g.Cin0, g.CB0);
      o1 = (size_t)u * cols + (size_t)cip * KK + ky * K + kx;
      if (u < g.sym_h / 2) { two = true; o2 = (size_t)(g.U + u) * cols + (size_t)cip * KK + ky * K + (K - 1 - kx); }
    } else {
      o1 = (size_t)(i - nW) * cols + (size_t)g.CinP * KK;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int G = w;
    for (; G + 12 < g.wgrad_G; G += 16) {
      a0 += part[(size_t)G * slab + o1]; a1 += part[(size_t)(G + 4) * slab + o1];
      a2 += part[(size_t)(G + 8) * slab + o1]; a3 += part[(size_t)(G + 12) * slab + o1];
      if (two) {
        a0 += part[(size_t)G * slab + o2]; a1 += part[(size_t)(G + 4) * slab + o2];
        a2 += part[(size_t)(G + 8) * slab + o2]; a3 += part[(size_t)(G + 12) * slab + o2];
      }
    }
    for (; G < g.wgrad_G; G += 4) {
      a0 += part[(size_t)G * slab + o1];
      if (two) a0 += part[(size_t)G * slab + o2];
    }
    s = (a0 + a1) + (a2 + a3);
  }
  red[w][e] = s;
  __syncthreads();
  if (w == 0 && i < total) {
    float r = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (i < nW) { if (dw) dw[i] += r; }
    else if (db) db[i - nW] += r;
  }
}

}  // namespace

extern "C" {

int mc_version(void) { return 100; }

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
  if (g.dtype == MC_BF16) return mc_bf16_bank_bytes(g, dgrad);
  int cbin = dgrad ? g.CBout : g.CBin, cop = dgrad ? g.CinP : g.CoutP;
  return (size_t)cbin * g.K * g.K * 8 * cop * sizeof(float);
}

int mc_pack_weights(const mc_conv_desc* d, const float* w_unique, int32_t dgrad, void* packed, void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!w_unique || !packed) return MC_EINVAL;
  if (g.dtype == MC_BF16) return mc_bf16_pack(g, w_unique, dgrad, packed, (hipStream_t)stream);
  size_t total = mc_packed_weight_bytes(d, dgrad) / sizeof(float);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, w_unique, dgrad, (float*)packed);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

const char* mc_conv_kernel_name(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return "unsupported";
  if (g.dtype == MC_BF16) return mc_bf16_kernel_name(g);
  return g.K == 5 ? "k_conv_direct_f32<5>" : "k_conv_direct_f32<3>";
}

int32_t mc_conv_tiles(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return -1;
  return g.tiles;
}

int mc_conv2d(const mc_conv_desc* d, const void* x0, const void* x1, const void* packed_w, const float* bias, void* y0,
              void* y1, float* stat_partials, void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!x0 || !packed_w || !y0 || (g.Cin1 > 0 && !x1) || (g.split8 > 0 && !y1)) return MC_EINVAL;
  if (g.dtype == MC_BF16) return mc_conv2d_bf16(g, x0, x1, packed_w, bias, y0, y1, stat_partials, (hipStream_t)stream);
  return mc_conv2d_f32(g, x0, x1, packed_w, bias, y0, y1, stat_partials, (hipStream_t)stream);
}

size_t mc_wgrad_partial_bytes(const mc_conv_desc* d) {
  ConvGeom g;
  if (geom_for(d, g)) return 0;
  return (size_t)g.wgrad_G * g.CoutP * ((size_t)g.CinP * g.K * g.K + 1) * sizeof(float);
}

int mc_conv2d_wgrad(const mc_conv_desc* d, const void* x0, const void* x1, const void* dy, void* partials,
                    void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!x0 || !dy || !partials || (g.Cin1 > 0 && !x1)) return MC_EINVAL;
  if (g.dtype == MC_BF16) return mc_wgrad_bf16(g, x0, x1, dy, partials, (hipStream_t)stream);
  return mc_wgrad_f32(g, x0, x1, dy, partials, (hipStream_t)stream);
}

int mc_conv2d_wgrad_finalize(const mc_conv_desc* d, const void* partials, float* dw_unique, float* dbias,
                             void* stream) {
  ConvGeom g;
  int rc = geom_for(d, g);
  if (rc) return rc;
  if (!partials || (!dw_unique && !dbias)) return MC_EINVAL;
  size_t total = (size_t)g.U * g.Cin * g.K * g.K + g.Cout;
  int blocks = (int)((total + 63) / 64);
  hipLaunchKernelGGL(k_wgrad_finalize, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (const float*)partials,
                     dw_unique, dbias);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

}  // extern "C"
