// Shared device helpers for libmantle_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mantle_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits

#define MC_CHECK_LAUNCH()                                 \
  do {                                                    \
    hipError_t e__ = hipGetLastError();                   \
    if (e__ != hipSuccess) return (int)e__;               \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950: RNE, NaN-preserving
  return __builtin_bit_cast(bf16_t, b);
}
// two values -> one packed dword (lo = a, hi = b) with ONE v_cvt_pk_bf16_f32 (the scalar casts above cost a convert,
// a mask, a shift and an or per pair)
typedef float mc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mc_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((mc_f32x2){a, b}, mc_bf16x2));
}

// IEEE half as a storage type of its own (MC_MIX16: every FORWARD tensor is f16 -- 11 significant bits, the momentum
// residual's second differences need them -- while gradient tensors stay bf16 for their range)
struct f16_t { uint16_t v; };
typedef _Float16 mc_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {       // v_cvt_pk_f16_f32 (RNE)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((mc_f32x2){a, b}, mc_f16x2));
}
__device__ __forceinline__ mc_f32x2 unpk_f16(uint32_t w) {
  return __builtin_convertvector(__builtin_bit_cast(mc_f16x2, w), mc_f32x2);
}
// the same value after a round trip through the 16-bit storage type T (f32: unchanged)
template <typename T> __device__ __forceinline__ float round_storage(float v) { return v; }

// ---- 8-channel vector access in the CB8 layout ------------------------------------------------
template <typename T> struct V8;
template <> struct V8<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&o)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
};
template <> struct V8<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&o)[8]) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
    o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
    o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[8]) {
    uint4 a;
    a.x = pk_bf16(o[0], o[1]);
    a.y = pk_bf16(o[2], o[3]);
    a.z = pk_bf16(o[4], o[5]);
    a.w = pk_bf16(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = a;
  }
};

template <> struct V8<f16_t> {
  static __device__ __forceinline__ void ld(const f16_t* p, float (&o)[8]) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    const mc_f32x2 x = unpk_f16(a.x), y = unpk_f16(a.y), z = unpk_f16(a.z), w = unpk_f16(a.w);
    o[0] = x.x; o[1] = x.y; o[2] = y.x; o[3] = y.y; o[4] = z.x; o[5] = z.y; o[6] = w.x; o[7] = w.y;
  }
  static __device__ __forceinline__ void st(f16_t* p, const float (&o)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pk_f16(o[0], o[1]), pk_f16(o[2], o[3]), pk_f16(o[4], o[5]), pk_f16(o[6], o[7]));
  }
};
template <> __device__ __forceinline__ float round_storage<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float round_storage<f16_t>(float v) { return (float)(_Float16)v; }

__device__ __forceinline__ size_t cb8_index(int n, int cb, int y, int x, int C8, int H, int W) {
  return ((((size_t)n * C8 + cb) * H + y) * (size_t)W + x) * 8;
}

// ---- padding index maps (F.pad modes; reflect excludes the edge pixel) -------------------------
// returns the source index for padded coordinate i (may be <0 or >=n), or -1 for "zero".
__device__ __forceinline__ int pad_map(int i, int n, int mode) {
  if (i >= 0 && i < n) return i;
  if (mode == MC_PAD_ZEROS) return -1;
  if (mode == MC_PAD_REPLICATE) return i < 0 ? 0 : n - 1;
  // reflect
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return (i >= 0 && i < n) ? i : -1;
}

// branch-free variant for batched staging loads: returns a clamped (always valid) index and sets ok
__device__ __forceinline__ int pad_map_sel(int i, int n, int mode, bool& ok) {
  int refl = i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i);
  int clmp = min(max(i, 0), n - 1);
  int j = mode == MC_PAD_REFLECT ? refl : clmp;
  ok = (mode != MC_PAD_ZEROS) || (i >= 0 && i < n);
  return min(max(j, 0), n - 1);
}

// ---- activations and derivatives (nn.GELU() exact erf form etc.) -------------------------------
__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case MC_ACT_GELU: return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
    case MC_ACT_RELU: return z > 0.f ? z : 0.f;
    case MC_ACT_SILU: return z / (1.0f + expf(-z));
    case MC_ACT_TANH: return tanhf(z);
    case MC_ACT_SELU: {
      const float al = 1.6732632423543772848170429916717f, sc = 1.0507009873554804934193349852946f;
      return sc * (z > 0.f ? z : al * (expf(z) - 1.0f));
    }
    case MC_ACT_ELU: return z > 0.f ? z : (expf(z) - 1.0f);
    default: return z;
  }
}
__device__ __forceinline__ float act_bwd(float z, int act) {
  switch (act) {
    case MC_ACT_GELU: {
      float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
      return cdf + z * pdf;
    }
    case MC_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case MC_ACT_SILU: {
      float s = 1.0f / (1.0f + expf(-z));
      return s * (1.0f + z * (1.0f - s));
    }
    case MC_ACT_TANH: {
      float t = tanhf(z);
      return 1.0f - t * t;
    }
    case MC_ACT_SELU: {
      const float al = 1.6732632423543772848170429916717f, sc = 1.0507009873554804934193349852946f;
      return z > 0.f ? sc : sc * al * expf(z);
    }
    case MC_ACT_ELU: return z > 0.f ? 1.f : expf(z);
    default: return 1.f;
  }
}

// bf16-mode GELU / GELU': odd minimax polynomials in z on |z| <= 4 (clamped outside), FMAs only - no erf / exp / rcp.
// The GroupNorm kernels were VALU-bound on the erff/expf forms (about 160 VALU slots per 32 bytes of traffic).
//   Phi(z)   = 0.5 + z P7(z^2):  |error| <= 4.3e-5 (GELU = z Phi(z): <= 1.7e-4 absolute), Phi(+-4) = 1 / 0 to rounding (fit
//              constraint; the value is NOT clamped to [0, 1]: it may leave the interval by the fit error)
//   GELU'(z) = 0.5 + z Q8(z^2):  |error| <= 5e-4   (the saturation: GELU'(4) = 1.0005 is mapped to 1)
// both far below the bf16 rounding (2^-9 relative) of the tensors they produce; the fp32 path uses erff/expf.
#define GELU_CDF_COEF 3.989080743e-01f, -6.630133159e-02f, 9.743199580e-03f, -1.069904774e-03f, 8.376238344e-05f, -4.337137479e-06f, 1.308749780e-07f, -1.722451212e-09f
#define GELU_GRAD_COEF 7.976932610e-01f, -2.647867851e-01f, 5.822368255e-02f, -8.560219625e-03f, 8.634741876e-04f, -5.865809327e-05f, 2.541382519e-06f, -6.288852653e-08f, 6.715542261e-10f
__device__ __forceinline__ float gelu_cdf_poly(float z) {
  const float zc = fminf(fmaxf(z, -4.0f), 4.0f), w = zc * zc;
  const float p = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(-1.722451212e-09f, w, 1.308749780e-07f), w, -4.337137479e-06f), w, 8.376238344e-05f), w, -1.069904774e-03f), w, 9.743199580e-03f), w, -6.630133159e-02f), w, 3.989080743e-01f);
  return fmaf(zc, p, 0.5f);      // within 4.3e-5 of [0, 1] by the fit: no clamp (8 VALU per 8 channels in the conv prologue)
}
__device__ __forceinline__ float gelu_grad_poly(float z) {
  const float c[9] = {GELU_GRAD_COEF};
  const float zc = fminf(fmaxf(z, -4.0f), 4.0f), w = zc * zc;
  float q = c[8];
#pragma unroll
  for (int k = 7; k >= 0; --k) q = fmaf(q, w, c[k]);
  return fmaf(zc, q, 0.5f);
}
template <bool FAST> __device__ __forceinline__ float act_fwd_t(float z, int act) {
  if (FAST && act == MC_ACT_GELU) return z * gelu_cdf_poly(z);
  return act_fwd(z, act);
}
template <bool FAST> __device__ __forceinline__ float act_bwd_t(float z, int act) {
  if (FAST && act == MC_ACT_GELU) return gelu_grad_poly(z);
  return act_bwd(z, act);
}
// ---- 8 channels at once: z = v * sc + sh, then act / act'.  The bf16 mode evaluates the GELU polynomials on channel
// pairs with packed-f32 FMAs (v_pk_fma_f32: two FMAs per lane per issue), which halves the VALU cost again.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_clamp(f32x2 a, float lo, float hi) {
  return (f32x2){fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi)};
}
template <int N> __device__ __forceinline__ f32x2 pk_horner(f32x2 w, const float (&c)[N]) {
  f32x2 p = (f32x2){c[N - 1], c[N - 1]};
#pragma unroll
  for (int k = N - 2; k >= 0; --k) p = pk_fma(p, w, (f32x2){c[k], c[k]});
  return p;
}
__device__ __forceinline__ f32x2 gelu_cdf_poly2(f32x2 z) {
  const float c[8] = {GELU_CDF_COEF};
  const f32x2 zc = pk_clamp(z, -4.0f, 4.0f);
  return pk_fma(zc, pk_horner(zc * zc, c), (f32x2){0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 gelu_grad_poly2(f32x2 z) {
#ifndef MC_GELU_GRAD_EXP
  // default: all-FMA fit, |error| <= 5e-4.  A/B on MI355X: the exp form below costs +0.2 ms/step and changes none of the
  // gradient diagnostics (tools/diag_convae.py: the bf16 storage rounding dominates by two orders of magnitude)
  const float c[9] = {GELU_GRAD_COEF};
  const f32x2 zc = pk_clamp(z, -4.0f, 4.0f);
  return pk_fma(zc, pk_horner(zc * zc, c), (f32x2){0.5f, 0.5f});
#else
  // GELU'(z) = Phi(z) + z pdf(z): the cdf polynomial plus ONE v_exp_f32 per channel (|error| <= 4.5e-5)
  const f32x2 t = (z * z) * (f32x2){-0.72134752044448170368f, -0.72134752044448170368f};      // -z^2 / 2 * log2(e)
  const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  return pk_fma(z * (f32x2){0.39894228040143267794f, 0.39894228040143267794f}, e, gelu_cdf_poly2(z));
#endif
}
template <bool FAST>
__device__ __forceinline__ void act_fwd8(const float (&v)[8], const float (&sc)[8], const float (&sh)[8], int act, float (&o)[8]) {
  if (FAST && act == MC_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x2 z = pk_fma((f32x2){v[j], v[j + 1]}, (f32x2){sc[j], sc[j + 1]}, (f32x2){sh[j], sh[j + 1]});
      f32x2 r = z * gelu_cdf_poly2(z);
      o[j] = r.x; o[j + 1] = r.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = act_fwd(v[j] * sc[j] + sh[j], act);
  }
}
template <bool FAST>
__device__ __forceinline__ void act_bwd8(const float (&v)[8], const float (&sc)[8], const float (&sh)[8], int act, float (&o)[8]) {
  if (FAST && act == MC_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x2 z = pk_fma((f32x2){v[j], v[j + 1]}, (f32x2){sc[j], sc[j + 1]}, (f32x2){sh[j], sh[j + 1]});
      f32x2 r = gelu_grad_poly2(z);
      o[j] = r.x; o[j + 1] = r.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = act_bwd(v[j] * sc[j] + sh[j], act);
  }
}
template <typename T> struct FastMath { static constexpr bool value = false; };
template <> struct FastMath<bf16_t> { static constexpr bool value = true; };
template <> struct FastMath<f16_t> { static constexpr bool value = true; };
// MC_BF16 / MC_MIX16: 16-bit storage on the MFMA path
__host__ __device__ inline bool mc_is16(int dtype) { return dtype == MC_BF16 || dtype == MC_MIX16; }

// ---- GroupNorm + activation applied by the CONSUMER of a raw conv output ("normalise on load") -----------------
// coef4: [n][CP][4] f32 = (scale, shift, mean, rstd) per (sample, channel), written by mc_gn_finalize_coef with the
// same f32 expressions gn_coef() uses (scale = rstd * gamma, shift = beta - mean * rstd * gamma), so a consumer that
// evaluates act(scale * y + shift) obtains bit-identical values to the stand-alone mc_gn_act_fwd pass.  Padded channels
// hold zeros (act(0) = 0 for every supported activation).  A NULL table means "activation only" (scale 1, shift 0).
// The (n, channel block) of a staged tile is workgroup-uniform, so these loads go through the scalar cache.
__device__ __forceinline__ void load_coef8(const float* __restrict__ coef4, int n, int CP, int cb, float (&sc)[8], float (&sh)[8]) {
  if (coef4) {
    const float4* p = reinterpret_cast<const float4*>(coef4) + (size_t)n * CP + cb * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float4 v = p[j]; sc[j] = v.x; sh[j] = v.y; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
  }
}
// one CB8 vector (8 channels of one pixel): raw conv output -> activated value, in the storage type
__device__ __forceinline__ uint4 xform_bf16x8(uint4 a, const float (&sc)[8], const float (&sh)[8], int act) {
  float v[8];
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  act_fwd8<true>(v, sc, sh, act, v);
  uint4 o;
  o.x = pk_bf16(v[0], v[1]);
  o.y = pk_bf16(v[2], v[3]);
  o.z = pk_bf16(v[4], v[5]);
  o.w = pk_bf16(v[6], v[7]);
  return o;
}

__device__ __forceinline__ uint4 xform_f16x8(uint4 a, const float (&sc)[8], const float (&sh)[8], int act) {
  float v[8];
  V8<f16_t>::ld(reinterpret_cast<const f16_t*>(&a), v);
  act_fwd8<true>(v, sc, sh, act, v);
  return make_uint4(pk_f16(v[0], v[1]), pk_f16(v[2], v[3]), pk_f16(v[4], v[5]), pk_f16(v[6], v[7]));
}
// f16 CB8 vector -> bf16 CB8 vector (the filter-gradient kernel reads the f16 activations of the forward pass)
__device__ __forceinline__ uint4 f16x8_to_bf16x8(uint4 a) {
  const mc_f32x2 x = unpk_f16(a.x), y = unpk_f16(a.y), z = unpk_f16(a.z), w = unpk_f16(a.w);
  return make_uint4(pk_bf16(x.x, x.y), pk_bf16(y.x, y.y), pk_bf16(z.x, z.y), pk_bf16(w.x, w.y));
}

// ---- wave / block reductions (wave = 64) --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16 per-lane values -> 16 wave totals with 17 shuffles instead of 96: at every step a lane keeps the half of its
// values selected by one lane-index bit and adds the partner's copy of that half.  On return the lanes with
// (lane & 3) == 0 hold the wave total of value index idx (each of the 16 indices on exactly one such lane).
__device__ __forceinline__ float wave_sum16(const float (&s)[16], int lane, int& idx) {
  float t[8], u[4], v[2], w;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = (b5 ? s[8 + k] : s[k]) + __shfl_xor(b5 ? s[k] : s[8 + k], 32, 64);
#pragma unroll
  for (int k = 0; k < 4; ++k) u[k] = (b4 ? t[4 + k] : t[k]) + __shfl_xor(b4 ? t[k] : t[4 + k], 16, 64);
#pragma unroll
  for (int k = 0; k < 2; ++k) v[k] = (b3 ? u[2 + k] : u[k]) + __shfl_xor(b3 ? u[k] : u[2 + k], 8, 64);
  w = (b2 ? v[1] : v[0]) + __shfl_xor(b2 ? v[0] : v[1], 4, 64);
  w += __shfl_xor(w, 2, 64);
  w += __shfl_xor(w, 1, 64);
  idx = (b5 ? 8 : 0) + (b4 ? 4 : 0) + (b3 ? 2 : 0) + (b2 ? 1 : 0);
  return w;
}

// ---- gradient sources of an activated tensor (see mc_grad_src) ---------------------------------
// candidates of padded coordinates that fold onto interior coordinate i: writes up to 3 padded
// indices (already offset by +p, i.e. indices into the padded buffer) and returns the count.
__device__ __forceinline__ int fold_candidates(int i, int n, int p, int mode, int (&out)[6]) {
  int cnt = 0;
  out[cnt++] = i + p;
  if (mode == MC_PAD_REFLECT) {
    if (i >= 1 && i <= p) out[cnt++] = p - i;                       // padded coord -i
    int j = 2 * (n - 1) - i;                                        // mirror about the last pixel
    if (j >= n && j < n + p) out[cnt++] = j + p;
  } else if (mode == MC_PAD_REPLICATE) {
    if (i == 0) for (int q = 0; q < p && cnt < 6; ++q) out[cnt++] = q;
    if (i == n - 1) for (int q = 0; q < p && cnt < 6; ++q) out[cnt++] = n + p + q;
  }
  return cnt;
}

// KIND >= 0 fixes the source kind at compile time (the GroupNorm-backward kernels are instantiated for the common
// (source 0, source 1) pairs: their streaming loops are register- and SGPR-bound, and the generic dispatch costs both)
template <typename T, int KIND = -1>
__device__ __forceinline__ void grad_fetch_add(const mc_grad_src& g, int n, int cb, int y, int x, int C8,
                                               float (&acc)[8]) {
  // PADDED sources have had the padding adjoint folded onto their interior by mc_fold_padded: read at offset pad.
  const int kind = KIND >= 0 ? KIND : g.kind;
  if (kind == MC_GSRC_NONE || (KIND < 0 && g.ptr == nullptr)) return;
  const T* base = reinterpret_cast<const T*>(g.ptr);
  if (KIND < 0 && g.c8_total > 0) { C8 = g.c8_total; cb += g.cb_off; }     // slice of a concatenated tensor (generic path only)
  float v[8];
  if (kind == MC_GSRC_PLAIN) {
    V8<T>::ld(base + cb8_index(n, cb, y, x, C8, g.hs, g.ws), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
    return;
  }
  float scale = 1.0f;
  int yy = y, xx = x;
  if (kind == MC_GSRC_PADFOLD_POOL || kind == MC_GSRC_PLAIN_POOL) {
    if (g.pool == 2) { yy = y >> 1; xx = x >> 1; } else { yy = y / g.pool; xx = x / g.pool; }
    if (yy >= g.hs || xx >= g.ws) return;   // floor mode: trailing rows/cols are not pooled
    scale = 1.0f / (float)(g.pool * g.pool);
  }
  const int pad = kind == MC_GSRC_PLAIN_POOL ? 0 : g.pad;
  V8<T>::ld(base + cb8_index(n, cb, yy + pad, xx + pad, C8, g.hs + 2 * pad, g.ws + 2 * pad), v);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += scale * v[j];
}
