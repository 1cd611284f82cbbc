// bf16 convolution for layers with one 16-channel output tile per work-group (C_out <= 16, or 48 = 3 groups): the
// level-0 / level-1 layers of the U-Net, where two thirds of the FLOPs are and where the step is HBM-bound.
//
// "Row reuse": the implicit GEMM keeps v_mfma_f32_16x16x32_bf16 (A = filter fragment: M = 16 output channels; B = input
// fragment: N = 16 pixels of a row; D: lane = pixel, 4 registers = 4 consecutive output channels), but the K = 32 slots of a
// fragment are four (kx, channel block) pairs OF ONE INPUT ROW (pair order p = 2 kx + cb, 2K pairs per row, the pairs of two
// consecutive rows packed into 2K/2 fragment "types").  One input fragment then feeds the K output rows it contributes to
// (output row r = input row - ky) — K MFMAs per ds_read_b128 instead of one — and the filter fragments, 26 (k = 5) or
// 10 (k = 3) per 16-channel input chunk, are loop invariants held in registers.  LDS reads per MFMA drop from 1.1 to 0.19
// and the per-fragment address arithmetic disappears (every read is base + immediate): the old kernel was bound by
// exactly those (LDS 39 %, VALU issue 45 %, MFMA 35 % busy, none saturated).
//
// Work-group = 8 waves, specialised:
//   waves 0-3  MFMA: each owns a strip of 16 output rows x 16 columns of the 16 x 64 tile (16 accumulators), runs the
//              K loop on the LDS tile of the current stage and the epilogue (bias, GroupNorm partial sums, store);
//   waves 4-7  loaders: stage the (16+k-1) x (64+k-1) x 16-channel input window of the NEXT stage: raw buffer loads two
//              stages ahead into registers, then — "normalise on load" — the producer's GroupNorm affine + activation on
//              the vector ALU (it co-issues with the other waves' MFMAs), then ds_write into the other LDS buffer.
// One s_barrier per stage.  A stage = one 16-channel input chunk of one (image, tile) work item; work-groups are persistent.
// (Tried and dropped, round 2: handing the finished bf16 tile to the loader waves through the consumed window buffer so
// that the MFMA waves skip their store-issue-bound epilogue -- 1700-2300 of ~5600 cycles per tile.  Level-0 16->16 forward
// 117 -> 148 us: the ~80 KB a work-group moves per tile take 2000-3900 cycles of vector-memory issue whichever waves issue
// them -- the layer runs at 5.3 TB/s of HBM + halo traffic -- and the second barrier per tile puts that time on the MFMA
// waves' critical path instead of beside it.)
// In the input-gradient form (FUSE == 2) the loader waves — which have no activation to apply to dY — also run the
// epilogue: the MFMA waves hand their f32 accumulators over through LDS and continue with the next stage, the loaders
// (holding the producer's raw output y for the tile's pixels in registers) form dz = dA act'(z), the GroupNorm-backward
// partial sums and the stores beside the next stage's MFMAs (see ConvFuse / mc_conv_epilogue).
#include "conv_rr.h"
#include <type_traits>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// timing-only diagnostic build (tools/build_variant.sh stamps -DMC_RR_STAMPS): per-phase cycle totals of one work-group
#ifdef MC_RR_STAMPS
#define RR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); long long now_ = clock64(); st_acc[k] += now_ - st_prev; st_prev = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RR_STAMP(k) do {} while (0)
#endif

namespace {

__device__ __forceinline__ int rr_xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__global__ void k_pack_rr(ConvGeom g, const float* __restrict__ wu, int dgrad, bf16_t* __restrict__ bank, int ntiles, size_t total, int f16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = rr_pack_value(g, wu, i, dgrad, ntiles);
    bank[i] = f16 ? __builtin_bit_cast(bf16_t, (_Float16)v) : f2bf(v);
  }
}

// 16-bit element type of a launch.  H16 = false: bf16 everywhere.  H16 = true (MC_MIX16): FUSE 0 / 1 = a forward
// convolution whose sources, filter bank and (16-bit) output are f16; FUSE 2 = an input-gradient convolution (bf16 operands
// and output) whose epilogue reads the producer's raw output y as f16.
template <bool H16> __device__ __forceinline__ f32x4 rr_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool H16> __device__ __forceinline__ uint32_t rr_pk(float a, float b) {
  if constexpr (H16) return pk_f16(a, b);
  else return pk_bf16(a, b);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS map (16-byte slots):  inbuf [2][2][PLANE] | FUSE != 2: wbuf [2][NFRAG][64]  | FUSE == 2: exch [16 x 64 px][4]
//   wbuf: layers with several input chunks get the next stage's filter fragments delivered with its window; one chunk:
//   staged once, then loop invariants in registers.  FUSE == 2: the accumulator exchange area (f32, 64 KB; the 16-byte
//   quad index of a pixel is XOR-swizzled with bits 2-3 of the pixel index so that both sides are conflict-free); the
//   filter fragments of a one-chunk layer are staged through it once, before its first use; several chunks (a rare
//   shape): the MFMA waves fetch their fragments from global memory per stage.
// GELU: every fused activation is GELU (inline polynomial); otherwise the generic activation switch is compiled in.
template <int K, int FUSE, bool OUT_F32, bool GELU, bool H16 = false, bool DZM = false>
__global__ __launch_bounds__(512, 2) void k_conv_rr_bf16(ConvGeom g, const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1,
                                                         const bf16_t* __restrict__ bank, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ y0, bf16_t* __restrict__ y1,
                                                         float* __restrict__ part, ConvFuse fz) {
  using S = RR<K>;
  // FUSE == 2 comes in two forms.  DZM (MC_MIX16 layers with GELU: y is f16): the MFMA waves keep their accumulators and
  // form dz and the two sums themselves in PACKED f16 (8.5 vector instructions per element); the loader waves run the plain
  // schedule.  LEX (any other type / activation): the accumulators go to the loader waves through LDS (`exch`), which
  // evaluate act' in f32 beside the next stage's MFMAs.  Round 3 measured the packed-f16 math on the LEX schedule too
  // (level-0 16 -> 16: 293 us against 118 us for the plain input gradient): with the y loads and the dz stores on top of
  // the window loads, FOUR loader waves issue every vector-memory instruction of the work-group (~200 cycles each while the
  // memory pipeline is backed up) and the MFMA waves wait at the barrier three quarters of the time.
  // DZM is instantiated for one-chunk launches only (<= 16 input channels of the launch: the host checks)
  static_assert(!DZM || (FUSE == 2 && H16 && GELU), "the MFMA-side dz epilogue is the packed-f16 GELU form");
  constexpr bool LEX = FUSE == 2 && !DZM;
  constexpr int TIH = S::TIH, TIW = S::TIW, PLANE = S::PLANE, NFRAG = S::NFRAG, NTY = S::NTYPES;
  constexpr int PER = (TIH * TIW + 255) / 256;            // staging slots per loader thread and plane
  constexpr int YSLOTS = RR_R * RR_TW, YPER = YSLOTS / 256;
  constexpr int NW = LEX ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* const inbuf = reinterpret_cast<uint4*>(smem_raw);
  uint4* const wbuf = inbuf + 2 * 2 * PLANE;
  uint4* const exch = wbuf;                                                     // (FUSE == 2) [YSLOTS][4]

  const int grp = blockIdx.y;                                                    // 16-channel output tile of this block
  const int ntiles_total = gridDim.y;
  const int bid = rr_xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunks = (g.CBin + 1) / 2;
  const int items = g.N * g.tiles;
  const int my_items = (items - bid + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_items * chunks;
  if (total <= 0) return;                                   // (uniform: every wave of the work-group leaves)
  const bool w_global = LEX && chunks > 1;                 // filter fragments straight from global memory (rare shape)

  auto stage_coords = [&](int t, int& n, int& ty0, int& tx0, int& ck, int& tile) {
    const int jitem = t / chunks;
    ck = t - jitem * chunks;
    const int wi = bid + jitem * (int)gridDim.x;
    n = wi / g.tiles;
    tile = wi - n * g.tiles;
    ty0 = (tile / g.tiles_x) * RR_R;
    tx0 = (tile % g.tiles_x) * RR_TW;
  };
  if (wave >= RR_STRIPS) {
    // ================================================= loader waves =================================================
    const int tl = threadIdx.x - 64 * RR_STRIPS;          // 0 .. 255
    int s_rc[PER];
    unsigned s_off[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      int i = tl + it * 256;
      const bool live = i < TIH * TIW;
      if (!live) i = TIH * TIW - 1;
      const int r = i / TIW, c = i - r * TIW;
      s_rc[it] = (live ? 0 : (1 << 31)) | (r << 15) | c;
      s_off[it] = (unsigned)(r * g.W + c) * 16u;
    }
    // FUSE == 2 keeps ONE stage of loads in flight (issued before the long epilogue of the previous stage, committed after
    // it); the other forms two (the loads of stage t + 2 are issued while stage t + 1 is delivered)
    constexpr int NSET = LEX ? 1 : 2;
    v4u rin[NSET][2][PER];                                 // [register set][plane][slot]
    unsigned okm[2] = {0xffffffffu, 0xffffffffu};
    v4u yv[2][YPER];                                       // FUSE == 2: the producer's raw output at this thread's pixels
    float ecv[4] = {1.f, 0.f, 0.f, 0.f};                   // FUSE == 2: (scale, shift, mean, rstd) of channel lane & 15 (four
                                                           // scalars: v_readlane of an ext-vector element returned element 0)
    constexpr int WPER = (NFRAG * 64 + 255) / 256;         // filter-fragment slots per loader thread
    v4u wv[LEX ? 1 : 2][LEX ? 1 : WPER];       // the stage's filter fragments (staged through LDS for the MFMA waves)
    const bool w_every = !w_global && chunks > 1;          // several chunks: a fresh set per stage; one chunk: stage 0 only

    auto issue = [&](int t, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      int n, ty0, tx0, ck, tile;
      stage_coords(t, n, ty0, tx0, ck, tile);
      const bool interior = (ty0 - g.pad >= 0) && (ty0 - g.pad + TIH <= g.H) && (tx0 - g.pad >= 0) && (tx0 - g.pad + TIW <= g.W);
      const unsigned org16 = (unsigned)((ty0 - g.pad) * g.W + (tx0 - g.pad)) * 16u;
      okm[SET] = 0xffffffffu;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int gcb = ck * 2 + cb;
        const int gcc = min(gcb, g.CBin - 1);
        const bool second = gcc >= g.CB0;
        int scb = second ? gcc - g.CB0 : gcc;
        int sC8 = second ? g.CB1 : g.CB0;
        const void* sp = second ? (const void*)x1 : (const void*)x0;
        const size_t plane_bytes = (size_t)g.H * g.W * 16;
        const char* pbase = reinterpret_cast<const char*>(sp) + ((size_t)n * sC8 + scb) * plane_bytes;
        // one descriptor per (image, channel-block plane): out-of-range offsets return zeros = zero padding / missing block
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, gcb < g.CBin ? (int)plane_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < PER; ++it) {
          unsigned off;
          if (interior) {
            off = org16 + s_off[it];
          } else {
            const int r = (s_rc[it] >> 15) & 0x7fff, c = s_rc[it] & 0x7fff;
            bool oky, okx;
            const int sy = pad_map_sel(ty0 + r - g.pad, g.H, g.pad_mode, oky);
            const int sx = pad_map_sel(tx0 + c - g.pad, g.W, g.pad_mode, okx);
            off = (oky && okx) ? (unsigned)(sy * g.W + sx) * 16u : 0xFFFFFFF0u;
            if (FUSE == 1 && cb == 0 && !(oky && okx)) okm[SET] &= ~(1u << it);
          }
          rin[SET][cb][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
      }
      if constexpr (!LEX) {
        if (!w_global && (w_every || t == 0)) {
          const v4u* wsrc = reinterpret_cast<const v4u*>(bank) + (size_t)(ck * ntiles_total + grp) * NFRAG * 64;
#pragma unroll
          for (int it = 0; it < WPER; ++it) wv[SET][it] = wsrc[min(tl + it * 256, NFRAG * 64 - 1)];
        }
      }
    };
    // FUSE == 2: y and the coefficients of stage t's work item, for the epilogue one iteration later
    auto load_y = [&](int t) {
      if constexpr (LEX) {
        if ((t % chunks) != chunks - 1) return;
        int n, ty0, tx0, ck, tile;
        stage_coords(t, n, ty0, tx0, ck, tile);
        // y at the interior coordinates of the tile's (padded-domain) output pixels; clamped: only finalised pixels use it
        const bf16_t* ey = reinterpret_cast<const bf16_t*>(fz.ey);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int cbc = min(grp * 2 + cb, g.CBout - 1);
#pragma unroll
          for (int it = 0; it < YPER; ++it) {
            const int i = tl + it * 256, r = i / RR_TW, c = i - r * RR_TW;
            const int cy = min(max(ty0 + r - fz.epad, 0), fz.ehs - 1), cx = min(max(tx0 + c - fz.epad, 0), fz.ews - 1);
            yv[cb][it] = *reinterpret_cast<const v4u*>(ey + cb8_index(n, cbc, cy, cx, g.CBout, fz.ehs, fz.ews));
          }
        }
        ecv[0] = 1.f; ecv[1] = 0.f; ecv[2] = 0.f; ecv[3] = 0.f;
        if (fz.ecoef) {
          const float4 c4 = reinterpret_cast<const float4*>(fz.ecoef)[(size_t)n * g.CoutP + min(grp * 16 + (lane & 15), g.CoutP - 1)];
          ecv[0] = c4.x; ecv[1] = c4.y; ecv[2] = c4.z; ecv[3] = c4.w;
        }
      }
    };
    // FUSE == 2: epilogue of stage t from the accumulators the MFMA waves left in `exch`: dz = dA act'(z) for the pixels
    // whose value is final (raw dA otherwise), the per-wave (sum dz, sum dz yhat) partials, the stores
    auto epilogue_dz = [&](int t) {
      if constexpr (LEX) {
        if ((t % chunks) != chunks - 1) return;
        int n, ty0, tx0, ck, tile;
        stage_coords(t, n, ty0, tx0, ck, tile);
        const int p = fz.epad, fr = fz.ezero ? 0 : p + 1;
        float s[2][16];                                      // [channel block][2 * channel + (0: sum dz, 1: sum dz yhat)]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int j = 0; j < 16; ++j) s[cb][j] = 0.f;
        bf16_t* yout = y0;
        bool inb[YPER], fin[YPER];
        int pix[YPER];
        size_t oidx[YPER];
#pragma unroll
        for (int it = 0; it < YPER; ++it) {
          const int i = tl + it * 256, r = i / RR_TW, c = i - r * RR_TW;
          const int oy = ty0 + r, ox = tx0 + c, iy = oy - p, ix = ox - p;
          pix[it] = i;
          inb[it] = oy < g.Ho && ox < g.Wo;
          fin[it] = inb[it] && iy >= fr && iy < fz.ehs - fr && ix >= fr && ix < fz.ews - fr;
          oidx[it] = cb8_index(n, grp * 2, min(oy, g.Ho - 1), min(ox, g.Wo - 1), g.CBout, g.Ho, g.Wo);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          float da[YPER][8], o[YPER][8];
#pragma unroll
          for (int it = 0; it < YPER; ++it) {
            const int sw = (pix[it] >> 2) & 3;
            const f32x4 a0 = reinterpret_cast<const f32x4*>(exch)[pix[it] * 4 + ((2 * cb) ^ sw)];
            const f32x4 a1 = reinterpret_cast<const f32x4*>(exch)[pix[it] * 4 + ((2 * cb + 1) ^ sw)];
            da[it][0] = a0[0]; da[it][1] = a0[1]; da[it][2] = a0[2]; da[it][3] = a0[3];
            da[it][4] = a1[0]; da[it][5] = a1[1]; da[it][6] = a1[2]; da[it][7] = a1[3];
          }
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            // (scale, shift, mean, rstd) of channels 8 cb + 2h (+1): lanes 8 cb + 2h (+1) of every wave hold them; broadcast
            // and pinned to vector registers (two different SGPR pairs as v_pk_fma_f32 operands were mis-compiled)
            float c0[4], c1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              c0[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ecv[q]), 8 * cb + 2 * h));
              c1[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ecv[q]), 8 * cb + 2 * h + 1));
              asm volatile("" : "+v"(c0[q]), "+v"(c1[q]));
            }
#pragma unroll
            for (int it = 0; it < YPER; ++it) {
              const unsigned yw = yv[cb][it][h];
              const f32x2 yy = H16 ? unpk_f16(yw) : (f32x2){__uint_as_float(yw << 16), __uint_as_float(yw & 0xffff0000u)};
              const f32x2 z = pk_fma(yy, (f32x2){c0[0], c1[0]}, (f32x2){c0[1], c1[1]});
              f32x2 gp;
              if (GELU || fz.eact == MC_ACT_GELU) gp = gelu_grad_poly2(z);
              else gp = (f32x2){act_bwd(z.x, fz.eact), act_bwd(z.y, fz.eact)};
              const f32x2 dA = (f32x2){da[it][2 * h], da[it][2 * h + 1]};
              const f32x2 dz = dA * gp;
              const f32x2 yh = (yy - (f32x2){c0[2], c1[2]}) * (f32x2){c0[3], c1[3]};
              if (fin[it]) {
                s[cb][4 * h] += dz.x; s[cb][4 * h + 1] += dz.x * yh.x;
                s[cb][4 * h + 2] += dz.y; s[cb][4 * h + 3] += dz.y * yh.y;
              }
              o[it][2 * h] = fin[it] ? dz.x : dA.x; o[it][2 * h + 1] = fin[it] ? dz.y : dA.y;
            }
          }
          if (grp * 2 + cb < g.CBout) {
#pragma unroll
            for (int it = 0; it < YPER; ++it)
              if (inb[it])
                *reinterpret_cast<uint4*>(yout + oidx[it] + (size_t)cb * g.Ho * g.Wo * 8) =
                    make_uint4(pk_bf16(o[it][0], o[it][1]), pk_bf16(o[it][2], o[it][3]), pk_bf16(o[it][4], o[it][5]), pk_bf16(o[it][6], o[it][7]));
          }
        }
        // wave totals: one partial-sum slot per (tile, loader wave)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          int idx;
          const float tot = wave_sum16(s[cb], lane, idx);
          const int co = grp * 16 + cb * 8 + (idx >> 1);
          if ((lane & 3) == 0 && co < g.CoutP)
            fz.epart[(((size_t)n * fz.estride + (size_t)tile * RR_STRIPS + (wave - RR_STRIPS)) * g.CoutP + co) * 2 + (idx & 1)] = tot;
        }
      }
    };
    auto commit = [&](int t, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      int n, ty0, tx0, ck, tile;
      stage_coords(t, n, ty0, tx0, ck, tile);
      uint4* dst = inbuf + (t & 1) * 2 * PLANE;
      // slots of this thread inside a plane (dead slots: their duplicate is transformed too and dropped at the store,
      // so that the transform below is straight-line code whose dependent FMA chains the scheduler can interleave)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        uint4 v[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) v[it] = make_uint4(rin[SET][cb][it][0], rin[SET][cb][it][1], rin[SET][cb][it][2], rin[SET][cb][it][3]);
        if constexpr (FUSE == 1) {
          const int gcb = ck * 2 + cb;
          const bool second = gcb >= g.CB0;
          const float* ct = second ? fz.coef1 : fz.coef0;
          const int pact = second ? fz.act1 : fz.act0;
          if (gcb < g.CBin && (ct != nullptr || pact != MC_ACT_NONE)) {
            float psc[8], psh[8];
            load_coef8(ct, n, (second ? g.CB1 : g.CB0) * 8, second ? gcb - g.CB0 : gcb, psc, psh);
#pragma unroll
            // normalise on load.  (A/B on MI355X: the packed-f32 FMA form, 207 us at level 0, beats a single-issue-FMA form
            // with twice the instructions, 275-290 us: the loader waves are bound by instruction issue beside the MFMAs.)
            for (int it = 0; it < PER; ++it)
              v[it] = H16 ? xform_f16x8(v[it], psc, psh, GELU ? (int)MC_ACT_GELU : pact) : xform_bf16x8(v[it], psc, psh, GELU ? (int)MC_ACT_GELU : pact);
            if (okm[SET] != 0xffffffffu) {                  // zero padding stays zero (border tiles of zero-padded layers only)
#pragma unroll
              for (int it = 0; it < PER; ++it) if (!((okm[SET] >> it) & 1u)) v[it] = make_uint4(0, 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int it = 0; it < PER; ++it)
          if (s_rc[it] >= 0) dst[cb * PLANE + ((s_rc[it] >> 15) & 0x7fff) * TIW + (s_rc[it] & 0x7fff)] = v[it];
      }
      if constexpr (!LEX) {
        if (!w_global && (w_every || t == 0)) {
          v4u* wd = reinterpret_cast<v4u*>(wbuf) + (w_every ? (t & 1) : 0) * NFRAG * 64;
#pragma unroll
          for (int it = 0; it < WPER; ++it) if (tl + it * 256 < NFRAG * 64) wd[tl + it * 256] = wv[SET][it];
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
#ifdef MC_RR_STAMPS
    long long st_acc[4] = {0, 0, 0, 0}, st_prev = clock64();
#endif
    if constexpr (LEX) {
      // one-chunk layers: the filter fragments pass through `exch` once (registers local to this block)
      if (!w_global) {
        const v4u* wsrc = reinterpret_cast<const v4u*>(bank) + (size_t)grp * NFRAG * 64;
        v4u* wd = reinterpret_cast<v4u*>(wbuf);
#pragma unroll
        for (int it = 0; it < WPER; ++it) {
          const v4u w = wsrc[min(tl + it * 256, NFRAG * 64 - 1)];
          if (tl + it * 256 < NFRAG * 64) wd[tl + it * 256] = w;
        }
      }
      issue(0, S0{});
      commit(0, S0{});
      __syncthreads();
      // iteration t (the MFMA waves compute stage t): loads of stage t + 1 -> epilogue of stage t - 1 from `exch` (the long
      // vector-ALU phase, beside the MFMAs) -> y of stage t -> delivery of stage t + 1.  Two barriers per stage: the MFMA
      // waves write their accumulators to `exch` between them.
      // (one extra trip for the last epilogue: a single call site keeps the unrolled epilogue once in the instruction stream)
      for (int t = 0; t <= total; ++t) {
        if (t + 1 < total) issue(t + 1, S0{});
        RR_STAMP(3);
        if (t > 0) epilogue_dz(t - 1);
        if (t == total) break;
        load_y(t);
        RR_STAMP(0);
        if (t + 1 < total) commit(t + 1, S0{});
        RR_STAMP(1);
        __syncthreads();
        __syncthreads();
        RR_STAMP(2);
      }
#ifdef MC_RR_STAMPS
      if (blockIdx.x == 77 && blockIdx.y == 0 && threadIdx.x == 256)
        printf("loader (dz) stages %d: epilogue + y loads %lld commit %lld barriers %lld issue %lld\n", total, st_acc[0], st_acc[1], st_acc[2], st_acc[3]);
#endif
      return;
    } else {
    issue(0, S0{});
    if (total > 1) issue(1, std::integral_constant<int, NSET - 1>{});
    commit(0, S0{});
    __syncthreads();
    RR_STAMP(3);
    // iteration t (the MFMA waves compute stage t): loads of stage t + 2, delivery of stage t + 1
    for (int t = 0; t < total; t += 2) {
      if (t + 2 < total) issue(t + 2, S0{});                // set 0 held stage t (already in LDS)
      RR_STAMP(0);
      if (t + 1 < total) commit(t + 1, S1{});
      RR_STAMP(1);
      __syncthreads();
      RR_STAMP(2);
      if (t + 1 >= total) break;
      if (t + 3 < total) issue(t + 3, S1{});
      RR_STAMP(0);
      if (t + 2 < total) commit(t + 2, S0{});
      RR_STAMP(1);
      __syncthreads();
      RR_STAMP(2);
    }
    }
#ifdef MC_RR_STAMPS
    if (blockIdx.x == 77 && blockIdx.y == 0 && threadIdx.x == 256)
      printf("loader stages %d: issue %lld commit %lld barrier %lld prologue %lld\n", total, st_acc[0], st_acc[1], st_acc[2], st_acc[3]);
#endif
    return;
  }

  // =================================================== MFMA waves ===================================================
  const int strip = wave;                                  // 16-column strip of the tile
  const int m = lane & 15, gq = lane >> 4;
  // this lane's slot offset per fragment type inside a [2][PLANE] buffer (row pair base 0)
  int aoff[NTY];
#pragma unroll
  for (int T = 0; T < NTY; ++T) {
    int rho = S::rho(T, 0), kx = S::kx(T, 0), cb = S::cb(T, 0);
#pragma unroll
    for (int gg = 1; gg < 4; ++gg) if (gq == gg) { rho = S::rho(T, gg); kx = S::kx(T, gg); cb = S::cb(T, gg); }
    aoff[T] = cb * PLANE + rho * TIW + kx + strip * 16 + m;
  }
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = grp * 16 + gq * 4 + r;
    bv[r] = (bias && co < g.Cout) ? bias[co] : 0.f;
  }
  f32x4 acc[RR_R];
  bf16x8 wf[NFRAG];
  const int cob = grp * 2 + (gq >> 1);                      // channel block of this lane's four output channels
  const bool cobok = cob < g.CBout;
  const int cbc = min(cob, g.CBout - 1);
  const int tiles4 = g.tiles * RR_STRIPS;                   // partial-sum slots per sample: one per (tile, strip)

  __syncthreads();                                          // stage 0 and its filter fragments are in LDS
#ifdef MC_RR_STAMPS
  long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = clock64();
#endif

  for (int t = 0; t < total; ++t) {
    int n, ty0, tx0, ck, tile;
    stage_coords(t, n, ty0, tx0, ck, tile);
    if (ck == 0) {
#pragma unroll
      for (int r = 0; r < RR_R; ++r) acc[r] = (f32x4){bv[0], bv[1], bv[2], bv[3]};
    }
    // filter fragments of this chunk -> registers (loop invariants of the K loop; a single-chunk layer keeps them for
    // the whole kernel)
    // (DZM: re-read from LDS every stage -- 104 registers held across an epilogue that needs 32 more for y spill.  Moving
    // these reads and the y loads behind the previous stage's epilogue, under the barrier, was tried: 85 spilled registers --
    // the K loop then holds 104 + 64 + 32 + 16 live registers across the loop's back edge.)
    if (!DZM && chunks == 1 && t > 0) {
    } else if (w_global) {
      const uint4* wsrc = reinterpret_cast<const uint4*>(bank) + ((size_t)(ck * ntiles_total + grp) * NFRAG) * 64 + lane;
#pragma unroll
      for (int f = 0; f < NFRAG; ++f) wf[f] = __builtin_bit_cast(bf16x8, wsrc[(size_t)f * 64]);
    } else {
      const uint4* ws = wbuf + ((NW == 2 && chunks > 1) ? (t & 1) : 0) * NFRAG * 64 + lane;
#pragma unroll
      for (int f = 0; f < NFRAG; ++f) wf[f] = __builtin_bit_cast(bf16x8, ws[f * 64]);
    }
    // DZM: the producer's raw output y at this lane's pixels, in flight during the K loop.  Lanes of the even 16-lane rows
    // load the 16-byte CB8 vector of output row r, the odd rows that of row r + 1 (r even); v_permlane16_swap hands every
    // lane the four channels it holds accumulators for -- the store path of the epilogue run backwards.
    // DZM: the producer's raw output y at this lane's pixels, in flight during the K loop.  Lanes of the even 16-lane rows
    // load the 16-byte CB8 vector of output row r, the odd rows that of row r + 1 (r even); v_permlane16_swap hands every
    // lane the four channels it holds accumulators for -- the store path of the epilogue run backwards.
    v4u yl[DZM ? RR_R / 2 : 1];
    if constexpr (DZM) {
      const bf16_t* ey = reinterpret_cast<const bf16_t*>(fz.ey);
      const int cx = min(max(tx0 + strip * 16 + m - fz.epad, 0), fz.ews - 1);
#pragma unroll
      for (int r = 0; r < RR_R; r += 2) {
        const int cy = min(max(ty0 + r + (gq & 1) - fz.epad, 0), fz.ehs - 1);
        yl[r / 2] = *reinterpret_cast<const v4u*>(ey + cb8_index(n, cbc, cy, cx, g.CBout, fz.ehs, fz.ews));
      }
    }
    RR_STAMP(0);
    // ---- K loop: every input fragment of a row pair feeds up to K (+1) output rows.  Fragments are read three ahead
    // of their MFMAs into a ring of four registers (an LDS read takes ~130 cycles, the 5-6 MFMAs of a fragment 80-96);
    // the scheduling barriers keep the compiler from regrouping reads and MFMAs (it otherwise waits for every read
    // right after issuing it).
    const uint4* buf = inbuf + (t & 1) * 2 * PLANE;
    constexpr int NF = (TIH / 2) * NTY, AHEAD = 3;
    bf16x8 fr[4];
    auto ldf = [&](int k) { return *reinterpret_cast<const bf16x8*>(buf + aoff[k % NTY] + 2 * (k / NTY) * TIW); };
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) fr[k] = ldf(k);
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (k + AHEAD < NF) fr[(k + AHEAD) & 3] = ldf(k + AHEAD);
      const int T = k % NTY, i = 2 * (k / NTY);
#pragma unroll
      for (int kappa = S::kmin(T); kappa <= S::kmax(T); ++kappa) {
        const int r = i - kappa;
        if (r >= 0 && r < RR_R) acc[r] = rr_mfma<(H16 && FUSE != 2)>(wf[S::fidx(T, kappa)], fr[k & 3], acc[r]);   // (FUSE == 2: bf16 gradients)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    RR_STAMP(1);
    if constexpr (LEX) {
      // hand the accumulators to the loader waves: exch [pixel][4 quads of 4 channels], quad index swizzled
      __syncthreads();                                      // every wave is done with the window (and the loaders with `exch`)
      if (ck == chunks - 1) {
#pragma unroll
        for (int r = 0; r < RR_R; ++r) {
          const int pix = r * RR_TW + strip * 16 + m;
          reinterpret_cast<f32x4*>(exch)[pix * 4 + (gq ^ ((pix >> 2) & 3))] = acc[r];
        }
      }
    }
    RR_STAMP(4);
    if (!LEX && ck == chunks - 1) {
      // ---- epilogue: this lane holds, for pixel column ox and rows ty0 .. ty0 + 15, four consecutive output channels
      const int ox = tx0 + strip * 16 + m;
      f32x2 s1[2] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}}, s2[2] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}};
      if constexpr (DZM) {
        // ---- input gradient with the GroupNorm-backward reduction (MC_MIX16, GELU): dz = dA GELU'(scale y + shift) for the
        // pixels whose value is final (raw dA for the frame that still awaits the padding adjoint), (sum dz, sum dz yhat)
        // per channel.  The affine map, the polynomial GELU'(z) = 1/2 + z Q4(z^2) on |z| <= 3 (clamped: GELU'(3) = 1.012;
        // evaluated in f16: rms error 1.6e-3 over z ~ N(0, 1), max 1.2e-2 in the tails -- the level of the bf16 rounding of
        // dz itself) and g * y run on channel PAIRS (v_pk_*_f16); dA stays f32 and meets g through v_fma_mix_f32.  The
        // second sum is taken as sum dz y and turned into sum dz yhat = rstd (sum dz y - mean sum dz) per lane.
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const int p = fz.epad, fr_ = fz.ezero ? 0 : p + 1;
        const int ix = ox - p;
        const bool colok = ox < g.Wo && cobok;
        const bool colfin = colok && ix >= fr_ && ix < fz.ews - fr_;
        float csc[4] = {1.f, 1.f, 1.f, 1.f}, csh[4] = {0.f, 0.f, 0.f, 0.f}, cme[4] = {0.f, 0.f, 0.f, 0.f}, crs[4] = {0.f, 0.f, 0.f, 0.f};
        if (fz.ecoef) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float4 c4 = reinterpret_cast<const float4*>(fz.ecoef)[(size_t)n * g.CoutP + cbc * 8 + (gq & 1) * 4 + r];
            csc[r] = c4.x; csh[r] = c4.y; cme[r] = c4.z; crs[r] = c4.w;
          }
        }
        const h2 sc01 = __builtin_bit_cast(h2, pk_f16(csc[0], csc[1])), sc23 = __builtin_bit_cast(h2, pk_f16(csc[2], csc[3]));
        const h2 sh01 = __builtin_bit_cast(h2, pk_f16(csh[0], csh[1])), sh23 = __builtin_bit_cast(h2, pk_f16(csh[2], csh[3]));
        float zero = 0.f;
        asm volatile("" : "+v"(zero));                           // keeps dA * g a v_fma_mix_f32
        float sd[4] = {0.f, 0.f, 0.f, 0.f}, sy[4] = {0.f, 0.f, 0.f, 0.f};     // sum dz, sum dz y of this lane's four channels
        char* dst16 = reinterpret_cast<char*>(y0) + (cb8_index(n, cbc, ty0, ox, g.CBout, g.Ho, g.Wo)) * 2 + (size_t)(gq & 1) * g.Wo * 16;
        const size_t row_bytes = (size_t)g.Wo * 16;
        const int iy0 = ty0 - p;
        auto rows = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;       // every pixel of the tile is inside and final: no masks
#pragma unroll
          for (int r = 0; r < RR_R; r += 2) {
            const unsigned l0 = yl[r / 2][0], l1 = yl[r / 2][1], l2 = yl[r / 2][2], l3 = yl[r / 2][3];
            const auto u01 = __builtin_amdgcn_permlane16_swap(l0, l2, false, false);      // [0]: row r, [1]: row r + 1
            const auto u23 = __builtin_amdgcn_permlane16_swap(l1, l3, false, false);
            // four chains in lock-step: (row r | r + 1) x (channels 01 | 23)
            h2 yy[4] = {__builtin_bit_cast(h2, (unsigned)u01[0]), __builtin_bit_cast(h2, (unsigned)u23[0]),
                        __builtin_bit_cast(h2, (unsigned)u01[1]), __builtin_bit_cast(h2, (unsigned)u23[1])};
            h2 zc[4], w[4], q[4], gg[4];
            const h2 R = {(_Float16)3.0f, (_Float16)3.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) zc[c] = __builtin_elementwise_fma(yy[c], (c & 1) ? sc23 : sc01, (c & 1) ? sh23 : sh01);
#pragma unroll
            for (int c = 0; c < 4; ++c) zc[c] = __builtin_elementwise_max(zc[c], -R);
#pragma unroll
            for (int c = 0; c < 4; ++c) zc[c] = __builtin_elementwise_min(zc[c], R);
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = zc[c] * zc[c];
#pragma unroll
            for (int c = 0; c < 4; ++c)
              q[c] = __builtin_elementwise_fma((h2){(_Float16)1.3422040e-04f, (_Float16)1.3422040e-04f}, w[c],
                                               (h2){(_Float16)-3.8080227e-03f, (_Float16)-3.8080227e-03f});
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = __builtin_elementwise_fma(q[c], w[c], (h2){(_Float16)4.2797559e-02f, (_Float16)4.2797559e-02f});
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = __builtin_elementwise_fma(q[c], w[c], (h2){(_Float16)-2.4320666e-01f, (_Float16)-2.4320666e-01f});
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = __builtin_elementwise_fma(q[c], w[c], (h2){(_Float16)7.8907706e-01f, (_Float16)7.8907706e-01f});
#pragma unroll
            for (int c = 0; c < 4; ++c) gg[c] = __builtin_elementwise_fma(zc[c], q[c], (h2){(_Float16)0.5f, (_Float16)0.5f});
            h2 gm[4], ge[4], hy[4];                              // multiplier of the sums / of the stored value; g y
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              gm[c] = gg[c]; ge[c] = gg[c];
              if constexpr (!FULL) {
                const int iy = iy0 + r + (c >> 1);
                const bool fin = colfin && ty0 + r + (c >> 1) < g.Ho && iy >= fr_ && iy < fz.ehs - fr_;
                gm[c] = fin ? gg[c] : (h2){(_Float16)0.f, (_Float16)0.f};
                ge[c] = fin ? gg[c] : (h2){(_Float16)1.f, (_Float16)1.f};
              }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hy[c] = gm[c] * yy[c];
            unsigned pk[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float dA0 = acc[r + (c >> 1)][2 * (c & 1)], dA1 = acc[r + (c >> 1)][2 * (c & 1) + 1];
              const float dz0 = __builtin_fmaf(dA0, (float)ge[c].x, zero), dz1 = __builtin_fmaf(dA1, (float)ge[c].y, zero);
              sd[2 * (c & 1)] = __builtin_fmaf(dA0, (float)gm[c].x, sd[2 * (c & 1)]);
              sd[2 * (c & 1) + 1] = __builtin_fmaf(dA1, (float)gm[c].y, sd[2 * (c & 1) + 1]);
              sy[2 * (c & 1)] = __builtin_fmaf(dA0, (float)hy[c].x, sy[2 * (c & 1)]);
              sy[2 * (c & 1) + 1] = __builtin_fmaf(dA1, (float)hy[c].y, sy[2 * (c & 1) + 1]);
              pk[c] = pk_bf16(dz0, dz1);
            }
            const auto sx = __builtin_amdgcn_permlane16_swap(pk[0], pk[2], false, false);
            const auto sz = __builtin_amdgcn_permlane16_swap(pk[1], pk[3], false, false);
            if (FULL || (colok && ty0 + r + (gq & 1) < g.Ho))
              *reinterpret_cast<uint4*>(dst16 + (size_t)r * row_bytes) = make_uint4(sx[0], sz[0], sx[1], sz[1]);
          }
        };
        if (ty0 + RR_R <= g.Ho && tx0 + RR_TW <= g.Wo && grp * 2 + 1 < g.CBout && iy0 >= fr_ && iy0 + RR_R <= fz.ehs - fr_ &&
            tx0 - p >= fr_ && tx0 - p + RR_TW <= fz.ews - fr_)
          rows(std::true_type{});
        else
          rows(std::false_type{});
#pragma unroll
        for (int c = 0; c < 4; ++c) sy[c] = crs[c] * (sy[c] - cme[c] * sd[c]);      // sum dz yhat
        s1[0] = (f32x2){sd[0], sd[1]}; s1[1] = (f32x2){sd[2], sd[3]};
        s2[0] = (f32x2){sy[0], sy[1]}; s2[1] = (f32x2){sy[2], sy[3]};
      } else {
      constexpr int esz = OUT_F32 ? 4 : 2;
      char* dst;
      if (g.split8 > 0 && cbc >= g.split8)
        dst = reinterpret_cast<char*>(y1) + (cb8_index(n, cbc - g.split8, ty0, ox, g.CBout - g.split8, g.Ho, g.Wo) + (gq & 1) * 4) * esz;
      else
        dst = reinterpret_cast<char*>(y0) + (cb8_index(n, cbc, ty0, ox, g.split8 > 0 ? g.split8 : g.CBout, g.Ho, g.Wo) + (gq & 1) * 4) * esz;
      const size_t row_bytes = (size_t)g.Wo * 8 * esz;
      const bool colok = ox < g.Wo && cobok;
      if constexpr (OUT_F32) {
        {
#pragma unroll
        for (int r = 0; r < RR_R; ++r) {
          const f32x2 v01 = (f32x2){acc[r][0], acc[r][1]}, v23 = (f32x2){acc[r][2], acc[r][3]};
          if (colok && ty0 + r < g.Ho) {
            s1[0] += v01; s1[1] += v23;
            s2[0] = pk_fma(v01, v01, s2[0]); s2[1] = pk_fma(v23, v23, s2[1]);
            *reinterpret_cast<float4*>(dst + (size_t)r * row_bytes) = make_float4(v01.x, v01.y, v23.x, v23.y);
          }
        }
        }
      } else {
        // Two output rows at a time: after v_permlane16_swap of the packed halves the lanes of the even 16-lane rows hold
        // the whole 16-byte CB8 vector (8 channels) of output row r, the odd rows that of row r + 1 — 8 dwordx4 stores
        // per wave instead of 16 dwordx2 (the epilogue was store-issue bound: 2300 of 6800 cycles per stage).
        char* dst16 = dst - (gq & 1) * 4 * esz + (size_t)(gq & 1) * row_bytes;
        auto rows = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;      // tile inside the image, both channel blocks exist: no masks
#pragma unroll
          for (int r = 0; r < RR_R; r += 2) {
            const f32x2 a01 = (f32x2){acc[r][0], acc[r][1]}, a23 = (f32x2){acc[r][2], acc[r][3]};
            const f32x2 b01 = (f32x2){acc[r + 1][0], acc[r + 1][1]}, b23 = (f32x2){acc[r + 1][2], acc[r + 1][3]};
            if (FULL || (colok && ty0 + r < g.Ho)) {
              s1[0] += a01; s1[1] += a23;
              s2[0] = pk_fma(a01, a01, s2[0]); s2[1] = pk_fma(a23, a23, s2[1]);
            }
            if (FULL || (colok && ty0 + r + 1 < g.Ho)) {
              s1[0] += b01; s1[1] += b23;
              s2[0] = pk_fma(b01, b01, s2[0]); s2[1] = pk_fma(b23, b23, s2[1]);
            }
            const auto sx = __builtin_amdgcn_permlane16_swap(rr_pk<H16>(a01.x, a01.y), rr_pk<H16>(b01.x, b01.y), false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(rr_pk<H16>(a23.x, a23.y), rr_pk<H16>(b23.x, b23.y), false, false);
#ifdef MC_RR_NOSTORE   /* timing-only ablation: wrong results */
            if ((FULL || (colok && ty0 + r + (gq & 1) < g.Ho)) && sx[0] == 0x12345678u)
#else
            if (FULL || (colok && ty0 + r + (gq & 1) < g.Ho))
#endif
              *reinterpret_cast<uint4*>(dst16 + (size_t)r * row_bytes) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        };
        if (ty0 + RR_R <= g.Ho && tx0 + RR_TW <= g.Wo && grp * 2 + 1 < g.CBout) rows(std::true_type{});
        else rows(std::false_type{});
      }
      }
      RR_STAMP(5);
      float* pp = DZM ? fz.epart : part;
      const int slots = DZM ? fz.estride : tiles4;
      if (pp) {
        // sum over the 16 pixel lanes of each 16-lane row with DPP row rotations (no LDS traffic: the ds_bpermute
        // butterfly cost 660 cycles per stage); every lane ends with the totals, lane m stores value index m >> 1
        float q8[8] = {s1[0].x, s2[0].x, s1[0].y, s2[0].y, s1[1].x, s2[1].x, s1[1].y, s2[1].y};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          q8[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q8[k]), 0x128, 0xf, 0xf, false));   // row_ror:8
          q8[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q8[k]), 0x124, 0xf, 0xf, false));   // row_ror:4
          q8[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q8[k]), 0x122, 0xf, 0xf, false));   // row_ror:2
          q8[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q8[k]), 0x121, 0xf, 0xf, false));   // row_ror:1
        }
        float q1 = q8[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) q1 = (m >> 1) == k ? q8[k] : q1;
        const int co = grp * 16 + gq * 4 + (m >> 2);
        if ((m & 1) == 0 && co < g.CoutP) {
          pp[(((size_t)n * slots + (size_t)tile * RR_STRIPS + strip) * g.CoutP + co) * 2 + ((m >> 1) & 1)] = q1;
        }
      }
    }
    RR_STAMP(2);
    __syncthreads();                                        // stage t consumed; stage t + 1 delivered
    RR_STAMP(3);
  }
#ifdef MC_RR_STAMPS
  if (blockIdx.x == 77 && blockIdx.y == 0 && threadIdx.x == 0)
    printf("mfma stages %d: wload %lld kloop %lld exch %lld rows %lld stats %lld barrier %lld\n", total, st_acc[0], st_acc[1], st_acc[4], st_acc[5], st_acc[2], st_acc[3]);
#endif
}

}  // namespace

// ----------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------
// Which layers take the row-reuse kernel (Cout, output width wo): every layer with an odd number of 16-channel output
// tiles (one tile per work-group: C_out <= 16, 48); layers with 2 .. 8 output tiles too when the image is at least
// MC_RR_MINW pixels wide — each output tile is then a separate work-group column (grid.y) that stages the window again,
// which the idle loader waves absorb, and the MFMA work per stage is the same.  MC_CONV_RR=0 turns the kernel off, 1
// restricts it to the odd case.
bool mc_rr_applies(int dtype, int cout, int wo, bool full_pad) {
  const int ntiles = (cout + 15) / 16;
  static const int on = [] { const char* e = getenv("MC_CONV_RR"); return e ? atoi(e) : 2; }();
  // (48 until the wide-tile kernel's K loop and wave grid were fixed in round 3; with them the 63/64-wide level takes the
  // wide-tile kernel: 100 vs 48 = -0.015 ... -0.03 ms in three same-box A/Bs, 200 = +-0)
  static const int minw = [] { const char* e = getenv("MC_RR_MINW"); return e ? atoi(e) : 100; }();
  // launches with full padding (pad = k - 1: the input-gradient convolutions) run on a domain 2 (k - 1) pixels wider than
  // the layer: the 64-wide tiles then waste up to a whole tile column, which the wide-tile kernel's 16 / 32-wide tiles do not
  // (A/B on MI355X, CFG-3 step, mixed / bf16: 0: 12.16 / 10.74 ms, 140: 12.12 / 10.70, 270: 12.32 / 10.86, 520: 12.62 / 11.18)
  static const int minw_full = [] { const char* e = getenv("MC_RR_MINW_DGRAD"); return e ? atoi(e) : 140; }();
  if (!on || !mc_is16(dtype)) return false;
  if (full_pad && minw_full > 0 && wo < minw_full) return false;
  if (ntiles % 2 == 1) return true;
  return on >= 2 && ntiles <= 8 && wo >= minw;
}

void mc_rr_tile(int* th, int* tw) { *th = RR_R; *tw = RR_TW; }
int mc_rr_stat_slots(const ConvGeom& g) { return g.tiles * RR_STRIPS; }

static void rr_bank_dims(const ConvGeom& g, int dgrad, int& chunks, int& ntiles, int& nfrag) {
  const int cb_in = dgrad ? g.CBout : g.CBin;
  const int c_out = dgrad ? g.CinP : g.Cout;
  chunks = (cb_in + 1) / 2;
  ntiles = (c_out + 15) / 16;
  nfrag = g.K == 5 ? RR<5>::NFRAG : RR<3>::NFRAG;
}
size_t mc_rr_bank_bytes(const ConvGeom& g, int dgrad) {
  int chunks, ntiles, nfrag;
  rr_bank_dims(g, dgrad, chunks, ntiles, nfrag);
  return (size_t)chunks * ntiles * nfrag * 64 * 16;
}
int mc_rr_pack(const ConvGeom& g, const float* w, int dgrad, void* packed, hipStream_t s) {
  int chunks, ntiles, nfrag;
  rr_bank_dims(g, dgrad, chunks, ntiles, nfrag);
  const size_t total = (size_t)chunks * ntiles * nfrag * 64 * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_rr, dim3(blocks), dim3(256), 0, s, g, w, dgrad, (bf16_t*)packed, ntiles, total,
                     (g.dtype == MC_MIX16 && !dgrad) ? 1 : 0);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

const char* mc_rr_kernel_name(const ConvGeom& g, int fuse) {
  if (g.K == 5) return g.out_f32 ? "k_conv_rr_bf16<5,f32out>" : (fuse == 2 ? "k_conv_rr_bf16<5,dz>" : (fuse == 1 ? "k_conv_rr_bf16<5,norm>" : "k_conv_rr_bf16<5>"));
  return g.out_f32 ? "k_conv_rr_bf16<3,f32out>" : (fuse == 2 ? "k_conv_rr_bf16<3,dz>" : (fuse == 1 ? "k_conv_rr_bf16<3,norm>" : "k_conv_rr_bf16<3>"));
}

template <int K, int FUSE, bool F32, bool GELU, bool H16 = false, bool DZM = false>
static int rr_launch(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0, void* y1,
                     float* part, const ConvFuse& fz, hipStream_t s) {
  using S = RR<K>;
  constexpr bool LEX = FUSE == 2 && !DZM;                    // (the accumulator exchange area; see the kernel)
  const size_t lds = (size_t)2 * 2 * S::PLANE * 16 + (LEX ? (size_t)RR_R * RR_TW * 64 : (size_t)2 * S::NFRAG * 1024);
  // (the attribute belongs to the CURRENT DEVICE's function object: one flag per instantiation and device, so that a process
  // driving several GPUs opts every one of them in)
  static bool attr_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MC_EINVAL;
  bool& attr_set = attr_dev[dev];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rr_bf16<K, FUSE, F32, GELU, H16, DZM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int groups = (g.Cout + 15) / 16;
  const int items = g.tiles * g.N;
  static const int cap_total = [] { const char* e = getenv("MC_RR_CAP"); return e ? atoi(e) : 256; }();   // one work-group per CU
  int cap = cap_total / groups > 0 ? cap_total / groups : 1;
  const int bx = items < cap ? items : cap;
  hipLaunchKernelGGL((k_conv_rr_bf16<K, FUSE, F32, GELU, H16, DZM>), dim3(bx, groups, 1), dim3(512), lds, s, g, (const bf16_t*)x0,
                     (const bf16_t*)x1, (const bf16_t*)bank, bias, (bf16_t*)y0, (bf16_t*)y1, part, fz);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_conv2d_rr(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0, void* y1,
                 float* part, const ConvFuse& fz, int fuse, hipStream_t s) {
  if (g.K != 5 && g.K != 3) return MC_EUNSUPPORTED;
  if (fuse == 2 && g.out_f32) return MC_EUNSUPPORTED;
  // GELU-only instantiations carry the inline polynomial; any other activation takes the generic-switch build
  const bool gelu = fuse == 2 ? fz.eact == MC_ACT_GELU
                              : ((fz.act0 == MC_ACT_GELU || (fz.act0 == MC_ACT_NONE && !fz.coef0)) &&
                                 (fz.act1 == MC_ACT_GELU || (fz.act1 == MC_ACT_NONE && !fz.coef1)));
  // h16: MC_MIX16 forward launch (f16 operands), or an input-gradient launch whose epilogue reads an f16 y (fz.ey16)
  const bool h16 = fuse == 2 ? fz.ey16 != 0 : g.dtype == MC_MIX16;
#define RRL(K, FU, F32, GE, H) rr_launch<K, FU, F32, GE, H>(g, x0, x1, bank, bias, y0, y1, part, fz, s)
#define RRK(K, H)                                                                                               \
  do {                                                                                                          \
    if (fuse == 0) return g.out_f32 ? RRL(K, 0, true, true, H) : RRL(K, 0, false, true, H);                     \
    if (fuse == 2) return gelu ? RRL(K, 2, false, true, H) : RRL(K, 2, false, false, H);                        \
    if (g.out_f32) return gelu ? RRL(K, 1, true, true, H) : RRL(K, 1, true, false, H);                          \
    return gelu ? RRL(K, 1, false, true, H) : RRL(K, 1, false, false, H);                                       \
  } while (0)
  // MC_MIX16 + GELU + at most 16 input channels of the launch: the MFMA-side packed-f16 epilogue
  if (h16 && fuse == 2 && gelu && g.CBin <= 2) {
    if (g.K == 5) return rr_launch<5, 2, false, true, true, true>(g, x0, x1, bank, bias, y0, y1, part, fz, s);
    return rr_launch<3, 2, false, true, true, true>(g, x0, x1, bank, bias, y0, y1, part, fz, s);
  }
  if (h16) { if (g.K == 5) RRK(5, true); RRK(3, true); }
  if (g.K == 5) RRK(5, false);
  RRK(3, false);
#undef RRK
#undef RRL
}
