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
#include <type_traits>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

namespace {

// XCD-aware work-group id: hardware deals consecutive blockIdx round-robin over the 8 XCDs (each with a private L2);
// remap so that every XCD owns a CONTIGUOUS range of ids -> spatially adjacent tiles (which share their input halo)
// run on the same XCD and the halo re-reads hit that XCD's L2.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

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
                            int chunks, int steps, int ntiles, int f16) {
  const size_t total = (size_t)chunks * steps * ntiles * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = pack_value_bf16(g, wu, i, dgrad, steps, ntiles);
    bank[i] = f16 ? __builtin_bit_cast(bf16_t, (_Float16)v) : f2bf(v);
  }
}

// H16 (MC_MIX16): FUSE 0 / 1 = forward convolution on f16 operands with an f16 output; FUSE 2 = input-gradient convolution
// (bf16) whose epilogue reads the producer's raw output y as f16 (see conv_rr_bf16.hip)
template <bool H16> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// forward / input-gradient kernel (persistent over work items, register-prefetched staging)
//
// A workgroup walks a list of stages (work item = (image, tile); stage = one 16-channel chunk of it).
// The global loads of stage t+1 are issued into registers right after stage t has been written to LDS
// and retire under stage t's MFMA loop (issue-early / write-late split staging); padding is resolved
// by clamped addressing + select (or skipped entirely for interior tiles), so the loads batch.
//
// MFMA orientation: A = filter fragment (M = 16 output channels), B = input fragment (N = 16 pixels),
// so the accumulator holds, per lane, FOUR CONSECUTIVE OUTPUT CHANNELS of ONE pixel (C/D: col = lane&15 =
// pixel, row = 4*(lane>>4)+reg = channel): exactly 8 contiguous bytes of the CB8 output vector.  The
// epilogue therefore stores straight from registers (no LDS transpose, no extra barriers).
// ------------------------------------------------------------------------------------------------
#ifdef MC_EXP_NOSYNC   /* timing experiment only: racy */
#define MC_SYNC() do {} while (0)
#else
#define MC_SYNC() __syncthreads()
#endif
// FUSE: 0 = plain; 1 = prologue (the sources are raw conv outputs: GroupNorm affine + activation applied while the tile
// is staged); 2 = input-gradient epilogue (dz = dA * act'(z) and the GroupNorm-backward partial sums instead of dA).
// WN (wave columns): 1 = every wave owns MT M-tiles x all NT N-tiles of the block; 2 = the waves form a (WAVES / 2) x 2 grid,
// a wave owns MT M-tiles x NT / 2 N-tiles.  With WN = 1 all four waves of a work-group fetch the SAME NT filter fragments
// per K-step through the vector L1 (one 1 KB wave-load = 16 cycles of the CU's only address / L1 path: 8 resident waves x 4
// fragments = 512 cycles per 12-MFMA step whose matrix-pipe share is 384; cycle stamps: 548); with WN = 2 a wave fetches half
// of them for twice the pixels (2 fragments through the L1, 6 through LDS per 12 MFMAs).
template <int K, int TH, int TW, int NT, int MT, bool OUT_F32 = false, int FUSE = 0, bool H16 = false, int WN = 1>
#ifndef MC_CONV_WAVES
#define MC_CONV_WAVES 2
#endif
__global__ __launch_bounds__(64 * (TH * (TW / 16) / MT) * WN, (MT * NT / WN <= 8 ? MC_CONV_WAVES : 2)) void k_conv_mfma_bf16(
    ConvGeom g, const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, const bf16_t* __restrict__ bank,
    const float* __restrict__ bias, bf16_t* __restrict__ y0, bf16_t* __restrict__ y1, float* __restrict__ part,
    int n_groups, ConvFuse fz) {
  constexpr int TIH = TH + K - 1, TIW = TW + K - 1;
  constexpr int PLANE = (TIH * TIW + 15) / 16 * 16;           // 16-byte slots per channel-block plane
  constexpr int STEPS = KSteps<K>::steps;
  constexpr int MTILES_X = TW / 16;
  constexpr int WAVES = TH * MTILES_X / MT * WN, NTHR = 64 * WAVES;   // each wave owns MT 16-pixel M-tiles of the output tile
  constexpr int NTW = NT / WN;                                         // ... and NTW of the block's NT N-tiles
  static_assert(TH * MTILES_X * WN == WAVES * MT && NTW * WN == NT && (WAVES == 4 || WAVES == 8), "tile / wave decomposition mismatch");
  constexpr int IN_SLOTS = CHUNK_CB * PLANE;
  // NT == 1: the bank slice (13 KiB) is staged in LDS once per chunk.  NT > 1 (deep, channel-heavy layers): the slice
  // would be 27-53 KiB per chunk and re-staging it dominated the per-workgroup critical path, so B fragments are read
  // straight from the L2-resident bank (each lane's fragment is one contiguous 16-byte load) a K-step ahead.
#ifndef MC_WGLOBAL_ALL
#define MC_WGLOBAL_ALL 0
#endif
  constexpr bool WGLOBAL = NT > 1 || MC_WGLOBAL_ALL;
  constexpr int W_SLOTS = WGLOBAL ? 0 : STEPS * NT * 64;
  static_assert(!OUT_F32 || NT == 1, "f32 output is for the single-N-tile configuration");
  constexpr int IN_ELEMS = CHUNK_CB * TIH * TIW;
  constexpr int IN_ITERS = (IN_ELEMS + NTHR - 1) / NTHR;
  constexpr int W_ITERS = WGLOBAL ? 1 : (W_SLOTS + NTHR - 1) / NTHR;
  __shared__ uint4 lds[IN_SLOTS + (WGLOBAL ? 1 : W_SLOTS)];
  __shared__ float red[WAVES][NT * 16 * 2];
  uint4* in_s = lds;
  uint4* w_s = lds + IN_SLOTS;

  const int grp = blockIdx.y;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar
  const int m = lane & 15, gq = lane >> 4;
  const int ntile0 = grp * NT;                                  // first global N-tile of this block
  const int wm = wave / WN, nt_w0 = (wave % WN) * NTW;          // wave row; first in-block N-tile of this wave (scalars)
  const int ntiles_total = gridDim.y * NT;
  const int chunks = (g.CBin + CHUNK_CB - 1) / CHUNK_CB;
  const int items = g.N * g.tiles;
  const int my_items = (items - bid + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total_stages = my_items * chunks;

  // Staging loads are raw buffer loads: one descriptor per (image, source tensor); an out-of-range offset
  // returns zeros in hardware, which IS the zero padding / missing channel block — no branches, no selects,
  // so the loads of a stage issue back to back.  Iterations are split per channel block (descriptor is uniform).
  constexpr int PER_CB = (TIH * TIW + NTHR - 1) / NTHR;
  static_assert(IN_ITERS <= CHUNK_CB * PER_CB, "");
  v4u rin[CHUNK_CB][PER_CB];
  // FUSE == 1: per staged channel block the producer's (scale, shift) of its 8 channels (workgroup-uniform: scalar
  // loads), its activation (-1: the block is used as it is) and the slots that are zero padding (bit it of okm clear)
  float psc[CHUNK_CB][8], psh[CHUNK_CB][8];
  int pact[CHUNK_CB] = {-1, -1};
  unsigned okm = 0xffffffffu;
  int s_rc[PER_CB];                                             // window (row, col) of this thread's slot; bit 31: dead
  unsigned s_off[PER_CB];                                       // byte offset of the slot inside an interior tile's window
  int s_lds[PER_CB];                                            // LDS slot (negative: dead)
#pragma unroll
  for (int it = 0; it < PER_CB; ++it) {
    int i = threadIdx.x + it * NTHR;
    bool live = i < TIH * TIW;
    if (!live) i = TIH * TIW - 1;
    int r = i / TIW, c = i - r * TIW;
    s_rc[it] = (live ? 0 : (1 << 31)) | (r << 15) | c;
    s_off[it] = (unsigned)(r * g.W + c) * 16u;
    s_lds[it] = live ? r * TIW + c : -1;
  }
  auto prefetch = [&](int t) {
    const int jitem = t / chunks, ck = t - jitem * chunks;
    const int wi = bid + jitem * (int)gridDim.x;
    const int n = wi / g.tiles, tile = wi - n * g.tiles;
    const int ty0 = (tile / g.tiles_x) * TH, tx0 = (tile % g.tiles_x) * TW;
    const bool interior = (ty0 - g.pad >= 0) && (ty0 - g.pad + TIH <= g.H) && (tx0 - g.pad >= 0) && (tx0 - g.pad + TIW <= g.W);
    const unsigned org16 = (unsigned)((ty0 - g.pad) * g.W + (tx0 - g.pad)) * 16u;   // wraps for border tiles (unused there)
#pragma unroll
    for (int cb = 0; cb < CHUNK_CB; ++cb) {
      const int gcb = ck * CHUNK_CB + cb;
      const int gcc = min(gcb, g.CBin - 1);
      const bool second = gcc >= g.CB0;
      const int scb = second ? gcc - g.CB0 : gcc;
      const int sC8 = second ? g.CB1 : g.CB0;
      const size_t plane_bytes = (size_t)g.H * g.W * 16;
      const char* pbase = reinterpret_cast<const char*>(second ? x1 : x0) + ((size_t)n * sC8 + scb) * plane_bytes;
      // records = one channel-block plane; a missing block (gcb >= CBin) gets an empty descriptor -> all zeros
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pbase, 0, gcb < g.CBin ? (int)plane_bytes : 0, 0x00020000);
      if constexpr (FUSE == 1) {
        const float* ct = second ? fz.coef1 : fz.coef0;
        const int a = second ? fz.act1 : fz.act0;
        pact[cb] = (gcb < g.CBin && (ct != nullptr || a != MC_ACT_NONE)) ? a : -1;
        if (pact[cb] >= 0) load_coef8(ct, n, sC8 * 8, scb, psc[cb], psh[cb]);
        if (cb == 0) okm = 0xffffffffu;
      }
#pragma unroll
      for (int it = 0; it < PER_CB; ++it) {
        unsigned off;
        if (interior) {
          off = org16 + s_off[it];
        } else {
          int r = (s_rc[it] >> 15) & 0x7fff, c = s_rc[it] & 0x7fff;
          bool oky, okx;
          int sy = pad_map_sel(ty0 + r - g.pad, g.H, g.pad_mode, oky);
          int sx = pad_map_sel(tx0 + c - g.pad, g.W, g.pad_mode, okx);
          off = (oky && okx) ? (unsigned)(sy * g.W + sx) * 16u : 0xFFFFFFF0u;
          if (FUSE == 1 && cb == 0 && !(oky && okx)) okm &= ~(1u << it);
        }
#ifdef MC_EXP_NOLOAD   /* timing experiment only: wrong results */
        rin[cb][it] = (v4u){off, off, off, off};
#else
        rin[cb][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
#endif
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int cb = 0; cb < CHUNK_CB; ++cb) {
      if (FUSE == 1 && pact[cb] >= 0) {
        // normalise on load: a = act(scale * y + shift) in the storage type; zero padding stays zero
#pragma unroll
        for (int it = 0; it < PER_CB; ++it)
          if (s_lds[it] >= 0) {
            const uint4 raw = make_uint4(rin[cb][it][0], rin[cb][it][1], rin[cb][it][2], rin[cb][it][3]);
            uint4 v = H16 ? xform_f16x8(raw, psc[cb], psh[cb], pact[cb]) : xform_bf16x8(raw, psc[cb], psh[cb], pact[cb]);
            if (!((okm >> it) & 1u)) v = make_uint4(0, 0, 0, 0);
            in_s[cb * PLANE + s_lds[it]] = v;
          }
        continue;
      }
#pragma unroll
      for (int it = 0; it < PER_CB; ++it)
        if (s_lds[it] >= 0)
          in_s[cb * PLANE + s_lds[it]] = make_uint4(rin[cb][it][0], rin[cb][it][1], rin[cb][it][2], rin[cb][it][3]);
    }
  };
  // the bank slice changes only with the chunk: single-chunk layers (65 % of the FLOPs) stage it once
  auto stage_weights = [&](int ck) {
    const uint4* bsrc = reinterpret_cast<const uint4*>(bank) + (size_t)ck * STEPS * ntiles_total * 64;
    uint4 rw[W_ITERS];
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
      int i = min((int)threadIdx.x + it * NTHR, W_SLOTS - 1);
      int ln = i & 63;
      int r = i >> 6;
      int tt = r % NT, ss = r / NT;
      rw[it] = bsrc[((size_t)ss * ntiles_total + ntile0 + tt) * 64 + ln];
    }
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
      int i = threadIdx.x + it * NTHR;
      if (i < W_SLOTS) w_s[i] = rw[it];
    }
  };

  // per-lane bias of its four output channels per N-tile
  float bv[NTW][4];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int co = (ntile0 + nt_w0 + tt) * 16 + gq * 4 + r;
      bv[tt][r] = (bias && co < g.Cout) ? bias[co] : 0.f;
    }

  // cross-stage register prefetch only where the register budget allows it (single N-tile configuration);
  // the wider configurations load and commit a stage back to back (their K loop is MT*NT MFMAs per fragment pair)
#ifndef MC_PREFETCH_ALL
#define MC_PREFETCH_ALL 0
#endif
  constexpr bool PREFETCH = (NT == 1) || MC_PREFETCH_ALL;
  f32x4 acc[MT][NTW];
  if (PREFETCH && total_stages > 0) prefetch(0);
#ifdef MC_EXP_STAMPS   /* timing experiment only */
  long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP(k) do { long long now_ = clock64(); st_acc[k] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
  for (int t = 0; t < total_stages; ++t) {
#ifdef MC_EXP_STAMPS
    long long st_prev = clock64();
#endif
    const int jitem = t / chunks, ck = t - jitem * chunks;
    if (ck == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) acc[i][tt] = (f32x4){bv[tt][0], bv[tt][1], bv[tt][2], bv[tt][3]};   // bias folded in
    }
    if (!PREFETCH) prefetch(t);
    MC_SYNC();                                            // LDS free: previous MFMA loop done
    STAMP(0);
    commit();
    if (!WGLOBAL && (chunks > 1 || t == 0)) stage_weights(ck);
    const uint4* wglob = reinterpret_cast<const uint4*>(bank) + ((size_t)ck * STEPS * ntiles_total + ntile0 + nt_w0) * 64 + lane;
    STAMP(1);
    MC_SYNC();
    STAMP(2);
    if (PREFETCH && t + 1 < total_stages) prefetch(t + 1);      // in flight during the MFMA loop
    STAMP(3);
    // ---- MFMA loop (rolled over the K-steps: the per-lane operand offset is recomputed per step)
    const int wbase = (wm * MT / MTILES_X) * TIW + ((wm * MT) % MTILES_X) * 16 + m;
    // K loop, software pipelined: the fragments of step s+1 are read from LDS (MT + NT ds_read_b128 into their own
    // registers) while the MT*NT MFMAs of step s issue, so an MFMA never waits on the LDS read issued just before it.
    // This lane's (tap, channel-block) pair advances by 4 pairs = 2 taps per step: (ky, kx) is walked incrementally.
    int kx = gq >> 1, ky = 0;                                   // pair jp = 4 s + gq -> tap = jp / 2, cb = jp % 2
    const int cbk_off = (gq & 1) * PLANE + wbase;
    auto load_frags = [&](int sidx, bf16x8 (&xf)[MT], bf16x8 (&wf)[NTW]) {
      const bool dummy = ky >= K;                               // pairs past k*k: weights are zero, any valid address
      const uint4* ap = in_s + (cbk_off + (dummy ? 0 : ky * TIW + kx));
      kx += 2;
      if (kx >= K) { kx -= K; ky += 1; }
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        if (WGLOBAL) {
#ifdef MC_EXP_NOWLOAD   /* timing experiment only: wrong results */
          uint4 wv = make_uint4(sidx, tt, lane, 0);
#else
          uint4 wv = wglob[((size_t)sidx * ntiles_total + tt) * 64];
#endif
          wf[tt] = __builtin_bit_cast(bf16x8, wv);
        } else {
          wf[tt] = *reinterpret_cast<const bf16x8*>(&w_s[(sidx * NT + tt) * 64 + lane]);
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
        xf[i] = *reinterpret_cast<const bf16x8*>(ap + (i / MTILES_X) * TIW + (i % MTILES_X) * 16);
    };
    auto do_mfma = [&](const bf16x8 (&xf)[MT], const bf16x8 (&wf)[NTW]) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) acc[i][tt] = mfma16<(H16 && FUSE != 2)>(wf[tt], xf[i], acc[i][tt]);
    };
#ifndef MC_KPIPE
#define MC_KPIPE 0
#endif
#ifndef MC_KUNROLL
#define MC_KUNROLL 1    /* 1: the K loop of the L2-fed configurations fully unrolled; 0: rolled by two steps (A/B) */
#endif
    if constexpr (WGLOBAL && MC_KUNROLL) {
    // Fully unrolled: in the rolled form the fragment loads of the next step sit behind a loop-carried condition
    // (s + 1 < STEPS), so the compiler cannot know how many loads are outstanding at the MFMAs and waits for ALL of them
    // (s_waitcnt vmcnt(3..0) lgkmcnt(2..0) in front of the first MFMAs of every step: the loads issued one step ahead were
    // waited for at once).  With static step indices every wait counts exactly the older set (vmcnt(7) lgkmcnt(5));
    // sched_barrier keeps the sets in program order (the scheduler otherwise hoists every step's loads).  -0.08 ms per step.
#ifndef MC_KDEPTH
#define MC_KDEPTH 2     /* fragment sets in the ring: the loads of step s + MC_KDEPTH - 1 are issued before the MFMAs of step s (3, 4: +-0) */
#endif
    constexpr int KD = MC_KDEPTH;
    bf16x8 xr[KD][MT], wr[KD][NTW];
#pragma unroll
    for (int s = 0; s < KD - 1 && s < STEPS; ++s) load_frags(s, xr[s], wr[s]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + KD - 1 < STEPS) load_frags(s + KD - 1, xr[(s + KD - 1) % KD], wr[(s + KD - 1) % KD]);
      __builtin_amdgcn_sched_barrier(0);
      do_mfma(xr[s % KD], wr[s % KD]);
      __builtin_amdgcn_sched_barrier(0);
    }
    } else if constexpr (WGLOBAL || MC_KPIPE == 2) {
    bf16x8 xa[MT], xb[MT], wa[NTW], wb[NTW];
    load_frags(0, xa, wa);
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 2) {
      if (s + 1 < STEPS) load_frags(s + 1, xb, wb);
      do_mfma(xa, wa);
      if (s + 2 < STEPS) load_frags(s + 2, xa, wa);
      if (s + 1 < STEPS) do_mfma(xb, wb);
    }
    } else {
    bf16x8 xa[MT], wa[NTW];
#pragma unroll 1
#ifdef MC_EXP_NOK   /* timing experiment only */
    for (int s = 0; s < 1; ++s) {
#else
    for (int s = 0; s < STEPS; ++s) {
#endif
      load_frags(s, xa, wa);
#if MC_KPIPE == 1
      // all fragment reads of the step first, each into its own registers, then the MFMAs: an MFMA waits only for ITS
      // fragment (the default schedule reused one register quad and exposed the LDS latency before every MFMA)
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NTW, 0);
#endif
      do_mfma(xa, wa);
#if MC_KPIPE == 1
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTW, 0);
#endif
    }
    }
    STAMP(4);
    if (ck != chunks - 1) continue;

    // ---- epilogue of this work item, straight from the accumulators
    const int wi = bid + jitem * (int)gridDim.x;
    const int n = wi / g.tiles, tile = wi - n * g.tiles;
    const int ty0 = (tile / g.tiles_x) * TH, tx0 = (tile % g.tiles_x) * TW;
    f32x2 s1[NTW][2], s2[NTW][2];                                // per-lane (sum, sum of squares) of channel pairs
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int h = 0; h < 2; ++h) { s1[tt][h] = (f32x2){0.f, 0.f}; s2[tt][h] = (f32x2){0.f, 0.f}; }
    // output pointers of this lane's first M-tile per N-tile (bytes); later M-tiles are constant strides away
    const int oy0 = ty0 + (wm * MT) / MTILES_X, ox0 = tx0 + ((wm * MT) % MTILES_X) * 16 + m;
    char* dst0[NTW];
    bool cobok[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int cob = (ntile0 + nt_w0 + tt) * 2 + (gq >> 1);    // C/D: row = 4*(lane>>4)+reg = output channel
      cobok[tt] = cob < g.CBout;
      const int cbc = min(cob, g.CBout - 1);
      const int esz = OUT_F32 ? 4 : 2;
      if (g.split8 > 0 && cbc >= g.split8)
        dst0[tt] = reinterpret_cast<char*>(y1) + (cb8_index(n, cbc - g.split8, oy0, ox0, g.CBout - g.split8, g.Ho, g.Wo) + (gq & 1) * 4) * esz;
      else
        dst0[tt] = reinterpret_cast<char*>(y0) + (cb8_index(n, cbc, oy0, ox0, g.split8 > 0 ? g.split8 : g.CBout, g.Ho, g.Wo) + (gq & 1) * 4) * esz;
    }
    const size_t row_bytes = (size_t)g.Wo * 8 * (OUT_F32 ? 4 : 2);
    // tiles that lie completely inside the output skip every bounds test (wave-uniform branch); the VALU work of this
    // epilogue, not the MFMA loop or the memory system, bounded the single-N-tile kernel (PMC: 590 VALU / 104 MFMA per wave-tile)
    auto emit = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int oy = oy0 + i / MTILES_X, ox = ox0 + (i % MTILES_X) * 16;         // C/D: col = lane & 15 = pixel
        const bool inb = FULL || (oy < g.Ho && ox < g.Wo);
        const size_t off = (size_t)(i / MTILES_X) * row_bytes + (size_t)(i % MTILES_X) * 16 * 8 * (OUT_F32 ? 4 : 2);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const f32x2 v01 = (f32x2){acc[i][tt][0], acc[i][tt][1]}, v23 = (f32x2){acc[i][tt][2], acc[i][tt][3]};
          if (inb) {
            s1[tt][0] += v01; s1[tt][1] += v23;
            s2[tt][0] = pk_fma(v01, v01, s2[tt][0]); s2[tt][1] = pk_fma(v23, v23, s2[tt][1]);
          }
#ifdef MC_EXP_NOSTORE   /* timing experiment only */
          if (inb && cobok[tt] && v01.x == 123.456f) {
#else
          if (inb && cobok[tt]) {
#endif
            if (OUT_F32) {
              *reinterpret_cast<float4*>(dst0[tt] + off) = make_float4(v01.x, v01.y, v23.x, v23.y);
            } else {
              uint2 pk;
              if constexpr (H16) { pk.x = pk_f16(v01.x, v01.y); pk.y = pk_f16(v23.x, v23.y); }
              else { pk.x = pk_bf16(v01.x, v01.y); pk.y = pk_bf16(v23.x, v23.y); }
              *reinterpret_cast<uint2*>(dst0[tt] + off) = pk;
            }
          }
        }
      }
    };
    // FUSE == 2 (input gradient with the GroupNorm-backward reduction fused in): the accumulators hold dA, the gradient
    // w.r.t. the ACTIVATED tensor a = act(z), z = scale * y + shift, on the padded domain.  Pixels whose value is final
    // (interior; with reflect / replicate padding only those farther than pad from the border — the frame still awaits
    // the padding adjoint, mc_fold_padded_dz finishes it) are stored as dz = dA * act'(z) and enter the per-tile
    // (sum dz, sum dz * yhat) partials; every other pixel is stored as raw dA.
    auto emit_dz = [&]() {
      const int p = fz.epad, fr = fz.ezero ? 0 : p + 1;
      const int CBe = g.CBout;                                    // channel blocks of the producer's output = ours
      const bf16_t* ey = reinterpret_cast<const bf16_t*>(fz.ey);
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        const int cob = (ntile0 + nt_w0 + tt) * 2 + (gq >> 1);
        const int cbc = min(cob, g.CBout - 1);
        // (scale, shift, mean, rstd) of this lane's four channels
        float csc[4], csh[4], cme[4], crs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (fz.ecoef) {
            const float4 c4 = reinterpret_cast<const float4*>(fz.ecoef)[(size_t)n * g.CoutP + cbc * 8 + (gq & 1) * 4 + r];
            csc[r] = c4.x; csh[r] = c4.y; cme[r] = c4.z; crs[r] = c4.w;
          } else { csc[r] = 1.f; csh[r] = 0.f; cme[r] = 0.f; crs[r] = 0.f; }
        }
        uint2 yv[MT];
        bool fin[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int iy = oy0 + i / MTILES_X - p, ix = ox0 + (i % MTILES_X) * 16 - p;
          fin[i] = cobok[tt] && iy >= fr && iy < fz.ehs - fr && ix >= fr && ix < fz.ews - fr;
          const int cy = min(max(iy, 0), fz.ehs - 1), cx = min(max(ix, 0), fz.ews - 1);
          yv[i] = *reinterpret_cast<const uint2*>(ey + cb8_index(n, cbc, cy, cx, CBe, fz.ehs, fz.ews) + (gq & 1) * 4);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int oy = oy0 + i / MTILES_X, ox = ox0 + (i % MTILES_X) * 16;
          const bool inb = oy < g.Ho && ox < g.Wo;
          const size_t off = (size_t)(i / MTILES_X) * row_bytes + (size_t)(i % MTILES_X) * 16 * 8 * 2;
          float yf[4];
          if constexpr (H16) {
            const f32x2 ya = unpk_f16(yv[i].x), yb = unpk_f16(yv[i].y);
            yf[0] = ya.x; yf[1] = ya.y; yf[2] = yb.x; yf[3] = yb.y;
          } else {
            yf[0] = __uint_as_float(yv[i].x << 16); yf[1] = __uint_as_float(yv[i].x & 0xffff0000u);
            yf[2] = __uint_as_float(yv[i].y << 16); yf[3] = __uint_as_float(yv[i].y & 0xffff0000u);
          }
          float o[4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 yy = (f32x2){yf[2 * h], yf[2 * h + 1]};
            const f32x2 z = pk_fma(yy, (f32x2){csc[2 * h], csc[2 * h + 1]}, (f32x2){csh[2 * h], csh[2 * h + 1]});
            f32x2 gp;
            if (fz.eact == MC_ACT_GELU) gp = gelu_grad_poly2(z);
            else gp = (f32x2){act_bwd(z.x, fz.eact), act_bwd(z.y, fz.eact)};
            const f32x2 da = (f32x2){acc[i][tt][2 * h], acc[i][tt][2 * h + 1]};
            const f32x2 dz = da * gp;
            const f32x2 yh = (yy - (f32x2){cme[2 * h], cme[2 * h + 1]}) * (f32x2){crs[2 * h], crs[2 * h + 1]};
            if (fin[i]) { s1[tt][h] += dz; s2[tt][h] = pk_fma(dz, yh, s2[tt][h]); }
            o[2 * h] = fin[i] ? dz.x : da.x; o[2 * h + 1] = fin[i] ? dz.y : da.y;
          }
          if (inb && cobok[tt]) {
            uint2 pk;
            pk.x = (uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16);
            pk.y = (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16);
            *reinterpret_cast<uint2*>(dst0[tt] + off) = pk;
          }
        }
      }
    };
    if constexpr (FUSE == 2) emit_dz();
    else { if (ty0 + TH <= g.Ho && tx0 + TW <= g.Wo) emit(std::true_type{}); else emit(std::false_type{}); }
    STAMP(5);
    if (FUSE == 2) part = fz.epart;
    if (part) {
      // sum over the 16 pixel lanes of each 16-lane group with a halving butterfly (8 shuffles for the 8 values of an
      // N-tile instead of 32): afterwards lane m of a group holds the group total of value index m >> 1 (m even)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        float q8[8] = {s1[tt][0].x, s2[tt][0].x, s1[tt][0].y, s2[tt][0].y, s1[tt][1].x, s2[tt][1].x, s1[tt][1].y, s2[tt][1].y};
        const bool b3 = m & 8, b2 = m & 4, b1 = m & 2;
        float q4[4], q2[2], q1;
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k] = (b3 ? q8[4 + k] : q8[k]) + __shfl_xor(b3 ? q8[k] : q8[4 + k], 8, 64);
#pragma unroll
        for (int k = 0; k < 2; ++k) q2[k] = (b2 ? q4[2 + k] : q4[k]) + __shfl_xor(b2 ? q4[k] : q4[2 + k], 4, 64);
        q1 = (b1 ? q2[1] : q2[0]) + __shfl_xor(b1 ? q2[0] : q2[1], 2, 64);
        q1 += __shfl_xor(q1, 1, 64);
        // value index = 4 b3 + 2 b2 + b1 = (channel r = idx >> 1, idx & 1 = sum / sum of squares)
        if ((m & 1) == 0) red[wave][((nt_w0 + tt) * 16 + gq * 4) * 2 + (m >> 1)] = q1;
      }
      MC_SYNC();
      if (threadIdx.x < NT * 32) {
        int co = ntile0 * 16 + (threadIdx.x >> 1);
        if (co < g.CoutP) {
          float r = 0.f;
#pragma unroll
          for (int wv = 0; wv < WAVES / WN; ++wv) r += red[wv * WN + (int)(threadIdx.x >> 5) / NTW][threadIdx.x];   // the waves that own this N-tile
          part[(((size_t)n * (FUSE == 2 ? fz.estride : g.tiles) + tile) * g.CoutP + co) * 2 + (threadIdx.x & 1)] = r;
        }
      }
    }
    STAMP(6);
  }
#ifdef MC_EXP_STAMPS
  if (blockIdx.x == 777 && blockIdx.y == 0 && threadIdx.x == 0)
    printf("stamps tiles %d: wait1 %lld commit %lld wait2 %lld pfetch %lld kloop %lld epi %lld stats %lld\n", my_items, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6]);
#endif
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
// Partials: [G] slabs (layout: wg_index in conv_common.h), combined deterministically by k_wgrad_finalize.
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
#ifndef MC_WGRAD_BATCH_NTW
#define MC_WGRAD_BATCH_NTW 1          // batch the tap-fragment reads of a row for NTW >= this (3 = never, 1 = always)
#endif

// RS ("register shift", RS != 0): the K dimension of an MFMA is 4 rows x 8 pixels instead of 32 pixels of a row, so
// the B fragments of the K taps of a filter row are windows [kx, kx+8) of the SAME 12 pixels of a lane: three
// transposing reads + 8 v_alignbit feed 5 MFMAs (0.7-0.85 LDS reads per MFMA instead of 2.3; the kernel was LDS-issue
// bound).  RS = number of tap groups the 25 taps are split into: wave = (tap group tg = wave % RS, column group
// wave / RS of RS 8-column blocks).  RS 1: all taps + bias, one column block (26 NTW accumulators); RS 2: 13/12 taps,
// two column blocks; RS 4: 7/6/6/6 taps, the whole tile (no cross-wave sum).  The bias gradient rides in the spare
// slot of the last tap group.  Column groups are summed through LDS once per workgroup.
// PRO: x0 / x1 are raw conv outputs; the producer's GroupNorm affine + activation are applied while the tile is staged.
// XH (MC_MIX16): x0 / x1 are f16 tensors of the forward pass, converted to bf16 while the tile is staged (dy is bf16).
template <int K, int NTW, int RS, bool PRO = false, bool XH = false>
__global__ __launch_bounds__(256, (NTW == 1 && RS >= 2) ? 3 : 2) void k_wgrad_mfma_bf16(ConvGeom g, const bf16_t* __restrict__ x0,
                                                            const bf16_t* __restrict__ x1, const bf16_t* __restrict__ dy,
                                                            float* __restrict__ part, int tiles_x, int tiles, ConvFuse fz) {
  static_assert(RS == 0 || RS == 1 || RS == 2 || RS == 4, "tap groups");
  constexpr int NS = RS ? RS : 1;                               // tap groups
  constexpr int NG = 4 / NS;                                    // column groups (waves that share a tap group)
  constexpr int TPW = (K * K + NS) / NS;                        // accumulator slots of a wave (last group: + bias)
  constexpr int KK = K * K;
  constexpr int TIH = WTH + K - 1, TIW = WTW + K - 1;
  // RS: row strides == 4 and plane strides == 8 (mod 16 slots) make the 4 row groups x 2 planes of a transposing read
  // hit 8 distinct bank quads
  constexpr int XRS = RS ? (TIW + 11) / 16 * 16 + 4 : TIW;    // x row stride (slots)
  constexpr int DRS = RS ? WTW + 4 : WTW;                     // dy row stride (slots)
  constexpr int XPS = RS ? (TIH * XRS + 15) / 16 * 16 + 8 : ((TIH * TIW + 15) / 16) * 16 + 4;   // x plane stride (slots)
  constexpr int DPS = RS ? WTH * DRS + 8 : WTH * WTW + 4;     // dy plane stride (slots)
  static_assert(!RS || (XRS >= TIW && XRS >= 3 * 8 + 12 && XRS % 16 == 4 && DRS % 16 == 4 && XPS % 16 == 8 && DPS % 16 == 8),
                "register-shift LDS layout");
  constexpr int NTAP = (KK + 3) / 4;                          // taps per wave (upper bound)
  constexpr int NACC = RS ? TPW * NTW : NTAP * NTW;            // [tap slot (+ bias)][co tile]
  constexpr int X_ITERS = (TIH * TIW + 255) / 256;            // staging slots per thread and channel-block plane
  constexpr int D_ELEMS = NTW * 2 * WTH * WTW, D_ITERS = D_ELEMS / 256;
  static_assert(D_ELEMS % 256 == 0, "dy tile must divide evenly over the threads");
  __shared__ uint4 smem[2 * XPS + NTW * 2 * DPS];
  static_assert(RS == 0 || RS == 4 || sizeof(smem) >= NS * NACC * 256 * sizeof(float), "cross-wave reduction reuses the tile buffers");
  uint4* const xs = smem;
  uint4* const ds = smem + 2 * XPS;
  const int chunk = blockIdx.y, cog = blockIdx.z;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: tap offsets live in SGPRs
  const int q = (lane & 15) >> 2, p = lane & 3, gq = lane >> 4;
  // per-lane short offsets (2-byte units) inside a plane pair for pixel 8g + q of a row start
  // (RS: pixel q of row gq of the wave's first 8-column block)
  const int tg = wave % NS, cg = wave / NS;
  const int col0 = 8 * NS * cg;
  const int lane_x = RS ? ((p >> 1) * XPS + gq * XRS + col0 + q) * 8 + (p & 1) * 4 : ((p >> 1) * XPS + 8 * gq + q) * 8 + (p & 1) * 4;
  const int lane_d = RS ? ((p >> 1) * DPS + gq * DRS + col0 + q) * 8 + (p & 1) * 4 : ((p >> 1) * DPS + 8 * gq + q) * 8 + (p & 1) * 4;
  const short* xs_s = reinterpret_cast<const short*>(xs);
  const short* ds_s = reinterpret_cast<const short*>(ds);
  static_assert(3 + 4 * (NTAP - 1) >= KK, "wave 3 needs a free accumulator slot for the bias gradient");
  const bool do_bias = (RS == 0 ? wave == 3 : tg == NS - 1) && (chunk == 0);
  const bf16x8 ones = (bf16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const ptrdiff_t x1_delta = x1 ? reinterpret_cast<const char*>(x1) - reinterpret_cast<const char*>(x0) : (ptrdiff_t)0;

  // this wave's taps -> short offsets into the x tile (static over the whole kernel)
  int toff[NTAP];
#pragma unroll
  for (int ti = 0; ti < NTAP; ++ti) {
    int tap = wave + 4 * ti;
    toff[ti] = tap < KK ? ((tap / K) * TIW + (tap % K)) * 8 : -1;
  }
  const short* xtap[NTAP];                                      // this lane's x-tile address per tap (row 0)
#pragma unroll
  for (int ti = 0; ti < NTAP; ++ti) xtap[ti] = xs_s + lane_x + max(toff[ti], 0);
  // static staging slots of this thread (decoded once: the staging address arithmetic was a third of the kernel's VALU work);
  // the two channel-block planes of the chunk use the same slots (the plane is uniform per load: scalar base, and the
  // producer's normalisation coefficients of a plane are uniform too)
  int x_rc[X_ITERS];
  unsigned x_off[X_ITERS];
#pragma unroll
  for (int it = 0; it < X_ITERS; ++it) {
    int i = threadIdx.x + it * 256;
    bool live = i < TIH * TIW;
    if (!live) i = TIH * TIW - 1;
    int r = i / TIW, c = i - r * TIW;
    x_rc[it] = (live ? 0 : (1 << 31)) | (r << 15) | c;
    x_off[it] = (unsigned)(r * g.W + c) * 16u;
  }
  // dy staging: static element offset of this thread's slots inside a (image, co-group) tile that lies fully inside
  // (256 threads = 8 rows x 32 columns; slot `it` of a thread is plane it / 2, row + 8 (it & 1): one VGPR offset, the
  // rest is uniform or an immediate)
  static_assert(WTW == 32 && WTH == 16, "dy staging decode");
  const int d_r0 = threadIdx.x >> 5, d_c0 = threadIdx.x & 31;
  const unsigned d_off0 = (unsigned)(d_r0 * g.Wo + d_c0) * 16u;
  const int d_lds0 = d_r0 * DRS + d_c0;
  const bool co_full = (cog * NTW + NTW) * 2 <= g.CBout;        // every co plane of this block exists
  // channel blocks of this (chunk): uniform
  const char* xbase[2];
  bool xok[2];
  int xC8[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    int gcb = chunk * 2 + cb;
    xok[cb] = gcb < g.CBin;
    int gcc = min(gcb, g.CBin - 1);
    bool second = gcc >= g.CB0;
    xC8[cb] = second ? g.CB1 : g.CB0;
    xbase[cb] = reinterpret_cast<const char*>(x0) + (second ? x1_delta : (ptrdiff_t)0) +
                cb8_index(0, second ? gcc - g.CB0 : gcc, 0, 0, xC8[cb], g.H, g.W) * sizeof(bf16_t);
  }

  uint4 rx[2][X_ITERS], rd[D_ITERS];
  // PRO: (scale, shift) of the 8 channels of each plane for the sample being staged (uniform: scalar loads), the plane's
  // activation (-1: used as it is), and the slots that are zero padding (bit clear)
  float psc[2][8], psh[2][8];
  int pact[2] = {-1, -1};
  unsigned okm = 0xffffffffu;
  auto prefetch = [&](int wi_fwd) __attribute__((always_inline)) {
    const int wi = wi_fwd;
    const int n = wi / tiles, tile = wi - n * tiles;
    const int ty0 = (tile / tiles_x) * WTH, tx0 = (tile % tiles_x) * WTW;
    const bool interior = (ty0 - g.pad >= 0) && (ty0 - g.pad + TIH <= g.H) && (tx0 - g.pad >= 0) && (tx0 - g.pad + TIW <= g.W);
    const char* bpn[2] = {xbase[0] + (size_t)n * xC8[0] * g.H * g.W * 16, xbase[1] + (size_t)n * xC8[1] * g.H * g.W * 16};
    if constexpr (PRO) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int gcb = min(chunk * 2 + cb, g.CBin - 1);
        const bool second = gcb >= g.CB0;
        const float* ct = second ? fz.coef1 : fz.coef0;
        const int a = second ? fz.act1 : fz.act0;
        pact[cb] = (xok[cb] && (ct != nullptr || a != MC_ACT_NONE)) ? a : -1;
        if (pact[cb] >= 0) load_coef8(ct, n, xC8[cb] * 8, second ? gcb - g.CB0 : gcb, psc[cb], psh[cb]);
      }
      okm = 0xffffffffu;
    }
    if (interior) {
      const size_t org = (size_t)((ty0 - g.pad) * g.W + (tx0 - g.pad)) * 16;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int it = 0; it < X_ITERS; ++it) {
          uint4 v = *reinterpret_cast<const uint4*>(bpn[cb] + org + x_off[it]);
          rx[cb][it] = xok[cb] ? v : make_uint4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int it = 0; it < X_ITERS; ++it) {
        int r = (x_rc[it] >> 15) & 0x7fff, c = x_rc[it] & 0x7fff;
        bool oky, okx;
        int sy = pad_map_sel(ty0 + r - g.pad, g.H, g.pad_mode, oky);
        int sx = pad_map_sel(tx0 + c - g.pad, g.W, g.pad_mode, okx);
        const bool ok = oky && okx;
        if (PRO && !ok) okm &= ~(1u << it);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          uint4 v = *reinterpret_cast<const uint4*>(bpn[cb] + (size_t)(sy * g.W + sx) * 16);
          rx[cb][it] = (ok && xok[cb]) ? v : make_uint4(0, 0, 0, 0);
        }
      }
    }
    if (co_full && ty0 + WTH <= g.Ho && tx0 + WTW <= g.Wo) {
      const char* db = reinterpret_cast<const char*>(dy) + cb8_index(n, cog * NTW * 2, ty0, tx0, g.CBout, g.Ho, g.Wo) * sizeof(bf16_t);
#pragma unroll
      for (int it = 0; it < D_ITERS; ++it)
        rd[it] = *reinterpret_cast<const uint4*>(db + (size_t)((it >> 1) * g.Ho + 8 * (it & 1)) * g.Wo * 16 + d_off0);
      return;
    }
#pragma unroll
    for (int it = 0; it < D_ITERS; ++it) {
      int i = threadIdx.x + it * 256;
      int pl = i / (WTH * WTW);                     // plane = co-tile * 2 + half
      int rem = i - pl * (WTH * WTW);
      int r = rem / WTW, c = rem - r * WTW;
      int cob = (cog * NTW) * 2 + pl;
      int oy = ty0 + r, ox = tx0 + c;
      bool ok = cob < g.CBout && oy < g.Ho && ox < g.Wo;
      uint4 v = *reinterpret_cast<const uint4*>(dy + cb8_index(n, min(cob, g.CBout - 1), min(oy, g.Ho - 1), min(ox, g.Wo - 1),
                                                               g.CBout, g.Ho, g.Wo));
      rd[it] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int it = 0; it < X_ITERS; ++it)
      {
        int v = x_rc[it];
        asm volatile("" : "+v"(v));               // opaque: keeps the slot decode out of loop-invariant registers
        if (v >= 0) {
          uint4 q = rx[cb][it];
          if (PRO && pact[cb] >= 0) {
            q = XH ? f16x8_to_bf16x8(xform_f16x8(q, psc[cb], psh[cb], pact[cb])) : xform_bf16x8(q, psc[cb], psh[cb], pact[cb]);
            if (!((okm >> it) & 1u)) q = make_uint4(0, 0, 0, 0);
          } else if (XH) {
            q = f16x8_to_bf16x8(q);
          }
          xs[cb * XPS + ((v >> 15) & 0x7fff) * XRS + (v & 0x7fff)] = q;
        }
      }
#pragma unroll
    for (int it = 0; it < D_ITERS; ++it) ds[d_lds0 + (it >> 1) * DPS + 8 * (it & 1) * DRS] = rd[it];
  };

  f32x4 acc[NACC];                          // !RS: [tap slot][co tile]
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int work = g.N * tiles;
  if (bid < work) prefetch(bid);
  // The tap group of a wave is wave-uniform; the whole tile loop is instantiated per group (tg_c) and the branch sits
  // OUTSIDE it: with the branch inside, the two arms' accumulators were allocated to disjoint registers (+52 VGPRs).
  // Every arm executes the same sequence of workgroup barriers.
  auto tile_loop = [&](auto tg_c) __attribute__((always_inline)) {
  for (int wi = bid; wi < work; wi += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
#ifndef MC_WEXP_NOLOAD   /* timing-only ablations (results wrong): MC_WEXP_NOLOAD, MC_WEXP_NOK */
    if (wi + (int)gridDim.x < work) prefetch(wi + gridDim.x);   // next work item's loads retire under the MFMA loop
#endif
#ifdef MC_WEXP_NOK
    if (g.N > 1000000)
#endif
#ifndef MC_WGRAD_ROW_UNROLL
#define MC_WGRAD_ROW_UNROLL 2   /* A/B on MI355X: 2 and 4 equal within noise, 16 thrashes the instruction cache (70x slower) */
#endif
    if constexpr (RS != 0) {
      typedef unsigned int u2 __attribute__((ext_vector_type(2)));
      // TG is a compile-time copy of the wave-uniform tap group so that the accumulators stay statically indexed
      auto body = [&](auto tg_c) __attribute__((always_inline)) {
        constexpr int TG = decltype(tg_c)::value;
        constexpr int T0 = TG == 0 ? 0 : TPW + (TG - 1) * (TPW - 1);
        constexpr int T1 = (TPW + TG * (TPW - 1)) < KK ? (TPW + TG * (TPW - 1)) : KK;
        static_assert(TG < NS - 1 || (T1 == KK && T1 - T0 < TPW), "the last tap group ends the filter and has a spare slot");
#pragma unroll
        for (int rg = 0; rg < WTH / 4; ++rg) {                  // K block = rows 4rg..4rg+3 x an 8-column block
#pragma unroll
          for (int cb = 0; cb < NS; ++cb) {
            const short* dp = ds_s + lane_d + (rg * 4 * DRS + cb * 8) * 8;
            bf16x8 a[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) a[t] = tr_frag(dp + t * 2 * DPS * 8);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
              if (ky * K + K <= T0 || ky * K >= T1) continue;   // no tap of this filter row belongs to the wave
              const short* xp = xs_s + lane_x + ((rg * 4 + ky) * XRS + cb * 8) * 8;
              // 12 pixels of this lane's (row, channel) as 6 packed dwords
              const u2 b0 = __builtin_bit_cast(u2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)xp));
              const u2 b1 = __builtin_bit_cast(u2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(xp + 32)));
              const u2 b2 = __builtin_bit_cast(u2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(xp + 64)));
              const unsigned d[6] = {b0[0], b0[1], b1[0], b1[1], b2[0], b2[1]};
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                const int tap = ky * K + kx;
                if (tap < T0 || tap >= T1) continue;
                v4u w;
                if (kx % 2 == 0) {
                  w = (v4u){d[kx / 2], d[kx / 2 + 1], d[kx / 2 + 2], d[kx / 2 + 3]};
                } else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_alignbit(d[kx / 2 + j + 1], d[kx / 2 + j], 16);
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                  acc[(tap - T0) * NTW + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], __builtin_bit_cast(bf16x8, w), acc[(tap - T0) * NTW + t], 0, 0, 0);
              }
            }
            if (TG == NS - 1 && do_bias) {
#pragma unroll
              for (int t = 0; t < NTW; ++t)
                acc[(TPW - 1) * NTW + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], ones, acc[(TPW - 1) * NTW + t], 0, 0, 0);
            }
#ifndef MC_WGRAD_NOFENCE
            if constexpr (NS > 1) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every block's reads (spills)
#endif
          }
        }
      };
      body(tg_c);
    } else {
    // rows unrolled so that the row offsets become instruction immediates (the loop was VALU-issue bound on address adds)
#pragma unroll MC_WGRAD_ROW_UNROLL
    for (int row = 0; row < WTH; ++row) {
      bf16x8 a[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) a[t] = tr_frag(ds_s + lane_d + (t * 2 * DPS + row * WTW) * 8);
      if constexpr (NTW >= MC_WGRAD_BATCH_NTW) {
        // read all of the row's tap fragments first so the MFMAs run back to back: -11 % on the 64-channel
        // layers (NTW == 2, already at 2 waves/SIMD).  On NTW == 1 it costs a wave of occupancy and the level-0
        // kernel alone gets 20 % slower, but inside the step (wgrad overlapped on the side stream) the whole step
        // was still 0.05-0.1 ms faster in 3-way A/B, so it is on for both.
        bf16x8 bq[NTAP];
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti) bq[ti] = tr_frag(xtap[ti] + row * TIW * 8);
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti) {
          if (toff[ti] >= 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[ti * NTW + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bq[ti], acc[ti * NTW + t], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti) {
          if (toff[ti] >= 0) {
            bf16x8 b = tr_frag(xtap[ti] + row * TIW * 8);
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[ti * NTW + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[ti * NTW + t], 0, 0, 0);
          }
        }
      }
      if (do_bias) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          acc[(NTAP - 1) * NTW + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], ones, acc[(NTAP - 1) * NTW + t], 0, 0, 0);
      }
    }
    }
  }
  };
  if constexpr (NS == 1) tile_loop(std::integral_constant<int, 0>{});
  else if constexpr (NS == 2) { if (tg == 0) tile_loop(std::integral_constant<int, 0>{}); else tile_loop(std::integral_constant<int, 1>{}); }
  else {
    if (tg == 0) tile_loop(std::integral_constant<int, 0>{}); else if (tg == 1) tile_loop(std::integral_constant<int, 1>{});
    else if (tg == 2) tile_loop(std::integral_constant<int, 2>{}); else tile_loop(std::integral_constant<int, 3>{});
  }
  // ---- write this block's partial slab: P[tap][chunk][co][16] (+ bias); a 16-lane group stores 64 contiguous bytes
  const int nch = wg_chunks(g.CinP);
  float* pb = part + (size_t)bid * wg_slab_floats(g.CoutP, g.CinP, KK);
  if constexpr (RS != 0) {
    // column groups 1.. hand their accumulators to group 0 through the (now idle) tile buffers, one group per round
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    for (int src = 1; src < NG; ++src) {
      __syncthreads();
      if (cg == src) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) red[(tg * NACC + a) * 64 + lane] = acc[a];
      }
      __syncthreads();
      if (cg == 0) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] += red[(tg * NACC + a) * 64 + lane];
      }
    }
    if (cg != 0) return;
    const int t0 = tg == 0 ? 0 : TPW + (tg - 1) * (TPW - 1);
    const int nt = min(KK, TPW + tg * (TPW - 1)) - t0;
#pragma unroll
    for (int sl = 0; sl < TPW; ++sl)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int co = (cog * NTW + t) * 16 + gq * 4 + r;
          if (sl < nt && co < g.CoutP) pb[((size_t)((t0 + sl) * nch + chunk) * g.CoutP + co) * 16 + (lane & 15)] = acc[sl * NTW + t][r];
        }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int co = (cog * NTW + t) * 16 + gq * 4 + r;
          if (co < g.CoutP) pb[(size_t)KK * nch * g.CoutP * 16 + co] = acc[(TPW - 1) * NTW + t][r];
        }
    }
    return;
  }
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
          pb[((size_t)(tap * nch + chunk) * g.CoutP + co) * 16 + (lane & 15)] = acc[ti * NTW + t][r];
        } else if (do_bias && ti == NTAP - 1 && (lane & 15) == 0) {
          pb[(size_t)KK * nch * g.CoutP * 16 + co] = acc[ti * NTW + t][r];
        }
      }
  }
}

#ifndef MC_NT1_MT
#define MC_NT1_MT 8     /* M-tiles per wave of the single-N-tile configuration: 8 -> 4 waves, 4 -> 8 waves per workgroup */
#endif
// Tile height 12 (3 M-tiles per wave) where it covers the output rows with fewer padded rows than 16: the input gradients of
// the deep levels run on the padded domain (35 rows at level 4: 3 x 12 instead of 3 x 16, 67 at level 3: 72 instead of 80,
// 130 at level 2: 132 instead of 144).  K = 5, several N-tiles per block only.
inline Bf16Cfg cfg_for(int c_out, int k, int ho) {
  static const int th12 = getenv("MC_CONV_TH12") ? atoi(getenv("MC_CONV_TH12")) : 1;
  int ntiles = (c_out + 15) / 16;
  int nt = pick_nt(ntiles);
  if (nt == 1) return {16, 32, 1, MC_NT1_MT};
  const bool t12 = th12 && k == 5 && (ho + 11) / 12 * 12 < (ho + 15) / 16 * 16;
  // (24-row tiles, 6 M-tiles per wave -- each filter fragment feeds 6 MFMAs instead of 3 or 4 -- measured +0.04 ms where they
  // cover the rows as tightly as 12 / 16 and +0.2 ms everywhere: fragment reuse is not what bounds these launches)
  // 8-row tiles (2 M-tiles per wave, twice the work-groups) for images of <= 40 output rows (level 4: 31 forward, 35 on the
  // padded domain): -0.02 ... -0.05 ms per step; at <= 70 rows +-0, at <= 140 rows +0.15 ms; 4-row tiles +0.2 ms
  static const int th8 = getenv("MC_CONV_TH8") ? atoi(getenv("MC_CONV_TH8")) : 40;
  if (th8 && k == 5 && ho <= th8) return nt == 2 ? Bf16Cfg{8, 16, 2, 2} : Bf16Cfg{8, 16, 4, 2};
  if (nt == 2) return t12 ? Bf16Cfg{12, 16, 2, 3} : Bf16Cfg{16, 16, 2, 4};
  // 24-row tiles, 6 x 4 accumulator tiles per wave (16 / 6 = 2.7 L1 cycles and 8 / 4 = 2 LDS cycles per MFMA against the 4
  // cycles of CU time an MFMA has: the one configuration the matrix pipe bounds), where 24 rows cover the image as tightly as
  // the 12 / 16-row choice (level 3 input gradients: 67 -> 72 rows): -0.03 ms per step.  MC_CONV_TH24 = largest row count.
  // (The same tile for two-N-tile layers, 6 x 2 accumulators on 130 -> 144 rows: +0.02 ms.  The (24-row tiles ...) note below
  // was measured before the K loop's waits were exact.)
  static const int th24 = getenv("MC_CONV_TH24") ? atoi(getenv("MC_CONV_TH24")) : 80;
  if (th24 && k == 5 && ho <= th24 && (ho + 23) / 24 * 24 <= (t12 ? (ho + 11) / 12 * 12 : (ho + 15) / 16 * 16)) return Bf16Cfg{24, 16, 4, 6};
  return t12 ? Bf16Cfg{12, 16, 4, 3} : Bf16Cfg{16, 16, 4, 4};
}

}  // namespace

int mc_bf16_tile(const mc_conv_desc* d, int* th, int* tw) {
  Bf16Cfg c = cfg_for(d->c_out, d->k, d->h + 2 * d->pad - d->k + 1);
  *th = c.th; *tw = c.tw;
  return MC_OK;
}

void mc_bf16_bank_dims(const ConvGeom& g, int dgrad, int& chunks, int& steps, int& ntiles) {
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
  mc_bf16_bank_dims(g, dgrad, chunks, steps, ntiles);
  return (size_t)chunks * steps * ntiles * 64 * 16;
}

int mc_bf16_pack(const ConvGeom& g, const float* w, int dgrad, void* packed, hipStream_t s) {
  int chunks, steps, ntiles;
  mc_bf16_bank_dims(g, dgrad, chunks, steps, ntiles);
  size_t total = (size_t)chunks * steps * ntiles * 64 * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_bf16, dim3(blocks), dim3(256), 0, s, g, w, dgrad, (bf16_t*)packed, chunks, steps, ntiles,
                     (g.dtype == MC_MIX16 && !dgrad) ? 1 : 0);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

const char* mc_bf16_kernel_name(const ConvGeom& g) {
  Bf16Cfg c = cfg_for(g.Cout, g.K, g.Ho);
  if (g.out_f32) return g.K == 5 ? "k_conv_mfma_bf16<5,16,32,1,8,true>" : "k_conv_mfma_bf16<3,16,32,1,8,true>";
  if (g.K == 5 && c.th == 24) return "k_conv_mfma_bf16<5,24,16,4,6,false>";
  if (g.K == 5 && c.th == 12) return c.nt == 2 ? "k_conv_mfma_bf16<5,12,16,2,3,false>" : "k_conv_mfma_bf16<5,12,16,4,6,false,2x2>";
  if (g.K == 5 && c.th == 8) return c.nt == 2 ? "k_conv_mfma_bf16<5,8,16,2,2,false>" : "k_conv_mfma_bf16<5,8,16,4,4,false,2x2>";
  if (g.K == 5) return c.nt == 1 ? "k_conv_mfma_bf16<5,16,32,1,8,false>" : (c.nt == 2 ? "k_conv_mfma_bf16<5,16,16,2,4,false>" : "k_conv_mfma_bf16<5,16,16,4,8,false,2x2>");
  return c.nt == 1 ? "k_conv_mfma_bf16<3,16,32,1,8,false>" : (c.nt == 2 ? "k_conv_mfma_bf16<3,16,16,2,4,false>" : "k_conv_mfma_bf16<3,16,16,4,8,false,2x2>");
}

int mc_conv2d_bf16(const ConvGeom& g_in, const void* x0, const void* x1, const void* bank, const float* bias, void* y0,
                   void* y1, float* part, const ConvFuse& fz, int fuse, hipStream_t s) {
  const ConvGeom& g = g_in;
  Bf16Cfg c = cfg_for(g.Cout, g.K, g.Ho);
  int nt_total = (g.Cout + 15) / 16;
  int groups = (nt_total + c.nt - 1) / c.nt;
  int items = g.tiles * g.N;
  static int cap_total = getenv("MC_CONV_CAP") ? atoi(getenv("MC_CONV_CAP")) : 4096;   // (environment override: tuning runs only)
  int cap = cap_total / groups > 0 ? cap_total / groups : 1;     // persistent: a few resident workgroups per CU, several items each
  int bx = items < cap ? items : cap;
  dim3 grid(bx, groups, 1);
  if (fuse == 2 && g.out_f32) return MC_EUNSUPPORTED;
  const bool h16 = fuse == 2 ? fz.ey16 != 0 : g.dtype == MC_MIX16;
#define LAUNCH_H(K, TH, TW, NT, MT, F32, FU, H, WN)                                                                    \
  hipLaunchKernelGGL((k_conv_mfma_bf16<K, TH, TW, NT, MT, F32, FU, H, WN>), grid, dim3(64 * (TH * (TW / 16) / (MT)) * (WN)), 0, s, g, \
                     (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)bank, bias, (bf16_t*)y0, (bf16_t*)y1, part,  \
                     groups, fz)
#define LAUNCH_F(K, TH, TW, NT, MT, F32, FU, WN)                                                                       \
  do { if (h16) LAUNCH_H(K, TH, TW, NT, MT, F32, FU, true, WN); else LAUNCH_H(K, TH, TW, NT, MT, F32, FU, false, WN); } while (0)
#define LAUNCH_W(K, TH, TW, NT, MT, F32, WN)                                                                           \
  do { if (fuse == 0) LAUNCH_F(K, TH, TW, NT, MT, F32, 0, WN); else if (fuse == 1) LAUNCH_F(K, TH, TW, NT, MT, F32, 1, WN);   \
       else LAUNCH_F(K, TH, TW, NT, MT, false, 2, WN); } while (0)
#define LAUNCH(K, TH, TW, NT, MT, F32) LAUNCH_W(K, TH, TW, NT, MT, F32, 1)
  // four N-tiles per block: the waves as a 2 x 2 grid (a wave owns twice the M-tiles and half of the N-tiles; see the kernel)
  static const int wn2 = getenv("MC_CONV_WN") ? atoi(getenv("MC_CONV_WN")) : 2;
#define LAUNCH4(K, TH, TW, MT) do { if (wn2 == 2) LAUNCH_W(K, TH, TW, 4, 2 * (MT), false, 2); else LAUNCH_W(K, TH, TW, 4, MT, false, 1); } while (0)
  if (g.out_f32) {
    if (g.K == 5) LAUNCH(5, 16, 32, 1, MC_NT1_MT, true); else if (g.K == 3) LAUNCH(3, 16, 32, 1, MC_NT1_MT, true); else return MC_EUNSUPPORTED;
  } else if (g.K == 5) {
    if (c.nt == 1) LAUNCH(5, 16, 32, 1, MC_NT1_MT, false);
    else if (c.th == 24) LAUNCH_W(5, 24, 16, 4, 6, false, 1);
    else if (c.th == 12) { if (c.nt == 2) LAUNCH(5, 12, 16, 2, 3, false); else LAUNCH4(5, 12, 16, 3); }
    else if (c.th == 8) { if (c.nt == 2) LAUNCH(5, 8, 16, 2, 2, false); else LAUNCH4(5, 8, 16, 2); }
    else if (c.nt == 2) LAUNCH(5, 16, 16, 2, 4, false); else LAUNCH4(5, 16, 16, 4);
  } else if (g.K == 3) {
    if (c.nt == 1) LAUNCH(3, 16, 32, 1, MC_NT1_MT, false); else if (c.nt == 2) LAUNCH(3, 16, 16, 2, 4, false); else LAUNCH4(3, 16, 16, 4);
  } else {
    return MC_EUNSUPPORTED;
  }
#undef LAUNCH4
#undef LAUNCH_W
#undef LAUNCH
#undef LAUNCH_F
#undef LAUNCH_H
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_wgrad_bf16(const ConvGeom& g_in, const void* x0, const void* x1, const void* dy, void* part, const ConvFuse& fz,
                  int fuse, hipStream_t s) {
  const ConvGeom& g = g_in;
  const int tiles_x = (g.Wo + WTW - 1) / WTW, tiles_y = (g.Ho + WTH - 1) / WTH;
  const int tiles = tiles_x * tiles_y;
  const int ntiles = (g.Cout + 15) / 16;
  const int ntw = pick_nt(ntiles) >= 2 ? 2 : 1;      // two co-tiles per block keep LDS at 55 KB (2-3 blocks per CU)
  dim3 grid(g.wgrad_G, (g.CBin + 1) / 2, (ntiles + ntw - 1) / ntw);
  const bool xh = g.dtype == MC_MIX16;
#define WLAUNCH_X(K, NTW, RS, PRO, XH)                                                                                 \
  hipLaunchKernelGGL((k_wgrad_mfma_bf16<K, NTW, RS, PRO, XH>), grid, dim3(256), 0, s, g, (const bf16_t*)x0, (const bf16_t*)x1, \
                     (const bf16_t*)dy, (float*)part, tiles_x, tiles, fz)
#define WLAUNCH_P(K, NTW, RS, PRO) do { if (xh) WLAUNCH_X(K, NTW, RS, PRO, true); else WLAUNCH_X(K, NTW, RS, PRO, false); } while (0)
#define WLAUNCH(K, NTW, RS) do { if (fuse) WLAUNCH_P(K, NTW, RS, true); else WLAUNCH_P(K, NTW, RS, false); } while (0)
  // A/B knob: two decimal digits = tap groups for (one co-tile, two co-tiles); 0 = the row-at-a-time kernel.  Measured
  // alone at level 0 (16->16, 32x506x512): 0: 266 us, 1: 171, 2: 203, 4: 155; 64->64 at 63x64: 0: 44 us, 4: 37.  Inside
  // the training step 10 / 14 / 44 are equal (the filter gradients run on the side stream and are off the critical path).
  static const int rs = [] { const char* e = getenv("MC_WGRAD_RS"); return e ? atoi(e) : 44; }();
  const int v = ntw == 1 ? rs / 10 : rs % 10;
#define WPICK(K, NTW) do { if (v == 4) WLAUNCH(K, NTW, 4); else if (v == 2) WLAUNCH(K, NTW, 2); else if (v == 1 && NTW == 1) WLAUNCH(K, 1, 1); else WLAUNCH(K, NTW, 0); } while (0)
  if (g.K == 5) { if (ntw == 1) WPICK(5, 1); else WPICK(5, 2); }
  else if (g.K == 3) { if (ntw == 1) WPICK(3, 1); else WPICK(3, 2); }
  else return MC_EUNSUPPORTED;
#undef WPICK
#undef WLAUNCH
#undef WLAUNCH_P
#undef WLAUNCH_X
  MC_CHECK_LAUNCH();
  return MC_OK;
}
