// Fused loss forward + backward (Trainer.loss_fn / get_loss, reference multigpu.py:122-134, 250-305)
// and the build-defined Stokes momentum residual.  5-point stencil work on
// [N][H][W] f32 fields: tiny next to the network, so neighbours are simply re-read through L1/L2.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.f) - (x < 0.f)); }

// block-level reduction of up to 16 running sums, then ONE f64 atomic per slot per block (a few thousand
// atomics per launch; per-wave atomics on a handful of addresses serialise at ~11 ns each and cost milliseconds)
struct BlockSums {
  double v[16];
  __device__ __forceinline__ BlockSums() {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0;
  }
  template <int NSLOT>
  __device__ __forceinline__ void flush(double* dst, const int (&slot)[NSLOT]) {
    __shared__ double red[4][16];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      double r = wave_sum_d(v[i]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < NSLOT) {
      double r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      if (r != 0.0) atomicAdd(dst + slot[threadIdx.x], r);
    }
  }
};

__global__ void k_minmax(const float* __restrict__ uvp, int ct, int hw, float* __restrict__ mm) {
  const int n = blockIdx.x, f = blockIdx.y;  // f = 0 (u), 1 (v) or 2 (p: FluidNet mode scales the pressure term too)
  const float* p = uvp + ((size_t)n * ct + f) * hw;
  float lo = 3.4e38f, hi = -3.4e38f;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) { float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
  __shared__ float rl[4], rh[4];
  if ((threadIdx.x & 63) == 0) { rl[threadIdx.x >> 6] = lo; rh[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mm[(n * 3 + f) * 2 + 0] = fminf(fminf(rl[0], rl[1]), fminf(rl[2], rl[3]));
    mm[(n * 3 + f) * 2 + 1] = fmaxf(fmaxf(rh[0], rh[1]), fmaxf(rh[2], rh[3]));
  }
}

struct LossGeom {
  mc_loss_desc d;
  int ct;           // channels of uvp
  int64_t pbs;      // batch stride of the u, v, T prediction planes
  int64_t ppbs;     // batch stride of the p plane
};

// tile geometry of the stencil kernels of this file: rows x 64 pixels per workgroup pass.  A/B on MI355X at
// 32 x 506 x 506 (us, rows 8 / 16 / 32): k_loss 153 / 144 / 135, k_mom_residual 99 / 92 / 113, k_mom_adjoint 87 / 92 / 111
constexpr int LT_TH = 32, LT_TW = 64;                    // k_loss
constexpr int LS_LW = LT_TW + 4, LS_LH = LT_TH + 4;      // k_loss: halo 2 (divergence sign of the four neighbours)

__global__ __launch_bounds__(256) void k_loss(LossGeom g, const float* __restrict__ u_, const float* __restrict__ v_,
                                              const float* __restrict__ p_, const float* __restrict__ T_,
                                              const float* __restrict__ uvp, const float* __restrict__ mm,
                                              double* __restrict__ sums, float* __restrict__ gu_,
                                              float* __restrict__ gv_, float* __restrict__ gp_,
                                              float* __restrict__ gT_, int tiles_x, int tiles) {
  const int H = g.d.h, W = g.d.w, HW = H * W, n = blockIdx.y;
  const float* u = u_ + (size_t)n * g.pbs;
  const float* v = v_ + (size_t)n * g.pbs;
  const float* p = p_ ? p_ + (size_t)n * g.ppbs : nullptr;
  const bool hasT = g.d.t_grad >= 0;                              // FluidNet family: no temperature output
  const float* T = hasT ? T_ + (size_t)n * g.pbs : nullptr;
  const float* ut = uvp + ((size_t)n * g.ct + 0) * HW;
  const float* vt = uvp + ((size_t)n * g.ct + 1) * HW;
  const float* pt = g.d.p_pred ? uvp + ((size_t)n * g.ct + 2) * HW : nullptr;
  const float* Tt = hasT ? uvp + ((size_t)n * g.ct + (g.d.p_pred ? 3 : 2)) * HW : nullptr;
  const float k = (g.d.p_pred ? 3.f : 2.f) + (hasT ? 1.f : 0.f);
  const float NHW = (float)g.d.n * (float)HW;
  const float cdat = 1.0f / (k * NHW);
  float su = 1.f, sv = 1.f;
  if (g.d.loss_scale) {
    su = fminf(fmaxf(1.0f / (mm[(n * 3 + 0) * 2 + 1] - mm[(n * 3 + 0) * 2 + 0]), 1.0f), 10.0f);
    sv = fminf(fmaxf(1.0f / (mm[(n * 3 + 1) * 2 + 1] - mm[(n * 3 + 1) * 2 + 0]), 1.0f), 10.0f);
  }
  // FluidNet branch of get_loss keeps the SCALED pressure loss (`loss_p, _ = self.loss_fn(p_true, p)`, multigpu.py:146-148);
  // the Unet branch keeps the plain one (:262-266)
  const bool p_scaled = !hasT && g.d.loss_scale && g.d.p_pred;
  const float sp = p_scaled ? fminf(fmaxf(1.0f / (mm[(n * 3 + 2) * 2 + 1] - mm[(n * 3 + 2) * 2 + 0]), 1.0f), 10.0f) : 1.f;
  // (126 = the reference's literal factor of the derivative loss, multigpu.py:163-166; mc_loss_desc.inv_h belongs to the
  // momentum residual only)
  const float cdu = 126.0f / (k * (float)g.d.n * (float)(H - 2) * (float)W);
  const float cdv = 126.0f / (k * (float)g.d.n * (float)H * (float)(W - 2));
  // per-thread partial sums in f32 (a thread adds <= a few dozen pixels: relative rounding < 1e-5 of ITS share; the
  // combination across threads, blocks and samples is f64): f64 adds were a third of this kernel's issue cycles
  float a_us = 0, a_up = 0, a_vs = 0, a_vp = 0, a_pp = 0, a_tp = 0, a_du = 0, a_dv = 0;
  float a_m = 0, a_mx0 = 0, a_mx1 = 0, a_my0 = 0, a_my1 = 0;

  // u, v and their targets of the tile + halo live in LDS (the per-pixel form issued ~45 global loads per pixel)
  __shared__ float us[LS_LH * LS_LW], vs[LS_LH * LS_LW], uts[LS_LH * LS_LW], vts[LS_LH * LS_LW];
  const float wm = 1.0f / ((float)g.d.n * (float)(H - 2) * (float)(W - 2));
  const float wc = 1.0f / ((float)g.d.n * (float)(H - 2)), wr = 1.0f / ((float)g.d.n * (float)(W - 2));
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int y0 = (tile / tiles_x) * LT_TH, x0 = (tile % tiles_x) * LT_TW;
    __syncthreads();                                    // the previous tile has been consumed
    for (int t = threadIdx.x; t < LS_LH * LS_LW; t += 256) {
      const int ly = t / LS_LW, lx = t - ly * LS_LW, yy = y0 + ly - 2, xx = x0 + lx - 2;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const size_t o = in ? (size_t)yy * W + xx : 0;
      us[t] = in ? u[o] : 0.f;
      vs[t] = in ? v[o] : 0.f;
      uts[t] = in ? ut[o] : 0.f;
      vts[t] = in ? vt[o] : 0.f;
    }
    __syncthreads();
#pragma unroll 2
  for (int r = 0; r < LT_TH / 4; ++r) {
    const int ly = (threadIdx.x >> 6) + 4 * r, lx = threadIdx.x & 63, y = y0 + ly, x = x0 + lx;
    if (y >= H || x >= W) continue;
    const int i = y * W + x;
    const int cc = (ly + 2) * LS_LW + lx + 2;
    auto U = [&](int dy, int dx) { return us[cc + dy * LS_LW + dx]; };
    auto V = [&](int dy, int dx) { return vs[cc + dy * LS_LW + dx]; };
    auto UT = [&](int dy, int dx) { return uts[cc + dy * LS_LW + dx]; };
    auto VT = [&](int dy, int dx) { return vts[cc + dy * LS_LW + dx]; };
    // weight * sign(D) of the divergence term at (y + dy, x + dx), 0 outside the interior
    auto mass_wsgn = [&](int dy, int dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 1 || yy > H - 2 || xx < 1 || xx > W - 2) return 0.f;
      const float D = 0.5f * (U(dy, dx + 1) - U(dy, dx - 1)) + 0.5f * (V(dy + 1, dx) - V(dy - 1, dx));
      const float w = g.d.loss_type == 1 ? wm
                    : (xx == 1 ? wc : 0.f) + (xx == W - 2 ? wc : 0.f) + (yy == 1 ? wr : 0.f) + (yy == H - 2 ? wr : 0.f);
      return w * sgn(D);
    };
    const float bw = (g.d.loss_scale && (y < 2 || y >= H - 2 || x < 2 || x >= W - 2)) ? 11.f : 1.f;
    float gu = 0.f, gv = 0.f, gp = 0.f, gT = 0.f;
    {  // data terms
      float du = UT(0, 0) - U(0, 0), dv = VT(0, 0) - V(0, 0), dT = hasT ? Tt[i] - T[i] : 0.f;
      float wu = su * bw, wv = sv * bw;
      if (!g.d.l2) {
        a_us += fabsf(du * wu); a_up += fabsf(du);
        a_vs += fabsf(dv * wv); a_vp += fabsf(dv);
        a_tp += fabsf(dT);
        gu -= sgn(du) * wu * cdat; gv -= sgn(dv) * wv * cdat; gT -= sgn(dT) * cdat;
      } else {
        a_us += (du * wu) * (du * wu); a_up += du * du;
        a_vs += (dv * wv) * (dv * wv); a_vp += dv * dv;
        a_tp += dT * dT;
        gu -= 2.f * du * wu * wu * cdat; gv -= 2.f * dv * wv * wv * cdat; gT -= 2.f * dT * cdat;
      }
      if (p) {
        float dp = pt[i] - p[i];
        const float wp = p_scaled ? sp * bw : 1.f;
        if (!g.d.l2) { a_pp += fabsf(dp * wp); gp -= sgn(dp) * wp * cdat; }
        else { a_pp += (dp * wp) * (dp * wp); gp -= 2.f * dp * wp * wp * cdat; }
      }
    }
    if (g.d.loss_derivative) {
      // e_u[y'] = 126 ((ut[y'+1]-ut[y']) - (u[y'+1]-u[y'])), y' in [0,H-3]  (dy_top, multigpu.py:277-284)
      if (y <= H - 3) {
        float e = (UT(1, 0) - UT(0, 0)) - (U(1, 0) - U(0, 0));
        a_du += fabsf(126.0f * e);
        gu += cdu * sgn(e);
      }
      if (y >= 1 && y <= H - 2) {
        float e = (UT(0, 0) - UT(-1, 0)) - (U(0, 0) - U(-1, 0));
        gu -= cdu * sgn(e);
      }
      if (x <= W - 3) {
        float e = (VT(0, 1) - VT(0, 0)) - (V(0, 1) - V(0, 0));
        a_dv += fabsf(126.0f * e);
        gv += cdv * sgn(e);
      }
      if (x >= 1 && x <= W - 2) {
        float e = (VT(0, 0) - VT(0, -1)) - (V(0, 0) - V(0, -1));
        gv -= cdv * sgn(e);
      }
    }
    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
      float D = 0.5f * (U(0, 1) - U(0, -1)) + 0.5f * (V(1, 0) - V(-1, 0));
      float m = fabsf(D);
      a_m += m;
      if (x == 1) a_mx0 += m;
      if (x == W - 2) a_mx1 += m;
      if (y == 1) a_my0 += m;
      if (y == H - 2) a_my1 += m;
    }
    if (g.d.loss_type != 0) {
      gu += 0.5f * (mass_wsgn(0, -1) - mass_wsgn(0, 1));
      gv += 0.5f * (mass_wsgn(-1, 0) - mass_wsgn(1, 0));
    }
    gu_[(size_t)n * g.pbs + i] = gu;
    gv_[(size_t)n * g.pbs + i] = gv;
    if (gp_) gp_[(size_t)n * g.ppbs + i] = gp;
    if (gT_) gT_[(size_t)n * g.pbs + i] = g.d.t_grad ? gT : 0.f;
  }
  }
  BlockSums bs;
  bs.v[0] = a_us; bs.v[1] = a_up; bs.v[2] = a_vs; bs.v[3] = a_vp; bs.v[4] = a_pp; bs.v[5] = a_tp; bs.v[6] = a_du;
  bs.v[7] = a_dv; bs.v[8] = a_m; bs.v[9] = a_mx0; bs.v[10] = a_mx1; bs.v[11] = a_my0; bs.v[12] = a_my1;
  const int slots[13] = {MC_S_U_SCALED, MC_S_U_PLAIN, MC_S_V_SCALED, MC_S_V_PLAIN, MC_S_P_PLAIN, MC_S_T_PLAIN, MC_S_DU,
                         MC_S_DV, MC_S_MASS, MC_S_MASS_X0, MC_S_MASS_X1, MC_S_MASS_Y0, MC_S_MASS_Y1};
  bs.flush<13>(sums, slots);
}

// ------------------------------------------------------------------------------------------------
// momentum residual
// ------------------------------------------------------------------------------------------------
struct MomGeom {
  int N, H, W;
  int64_t pbs, ppbs;
  float ih, ra, lam;
};

// Residual, LDS-tiled like the adjoint below: a block walks 8 x 64 pixel tiles of its sample; s*u, s*v, p and the
// viscosity eta = clip(exp(-ln(FKT) T + ln(FKP) (1 - y)), 1e-8, 1) of the tile + a one-pixel halo are staged once (eta is
// evaluated here, 1.3 exp per pixel, and its centre values are kept in eta_out for the adjoint), then every face is formed
// from LDS.  The per-pixel form issued ~35 global loads per pixel.
constexpr int MA_TW = LT_TW, MA_LW = MA_TW + 2;         // momentum kernels: halo 1
constexpr int MR_TH = 16, MR_LH = MR_TH + 2;            // residual
constexpr int MA_TH = 8, MA_LH = MA_TH + 2;             // adjoint
__global__ __launch_bounds__(256) void k_mom_residual(MomGeom g, const float* __restrict__ u_, const float* __restrict__ v_,
                                                      const float* __restrict__ p_, const float* __restrict__ T_,
                                                      const float* __restrict__ yc, const float* __restrict__ paras,
                                                      const float* __restrict__ scaler, double* __restrict__ sums,
                                                      float* __restrict__ sx_, float* __restrict__ sy_,
                                                      float* __restrict__ eta_out, int tiles_x, int tiles) {
  const int H = g.H, W = g.W, HW = H * W, n = blockIdx.y;
  const float* u = u_ + (size_t)n * g.pbs;
  const float* v = v_ + (size_t)n * g.pbs;
  const float* p = p_ ? p_ + (size_t)n * g.ppbs : nullptr;
  const float* T = T_ + (size_t)n * g.pbs;
  const float lnfkt = logf(paras[n * 3 + 1]), lnfkp = logf(paras[n * 3 + 2]);
  const float s = scaler[n], ih = g.ih;
  const float c = g.lam / ((float)g.N * (float)(H - 2) * (float)(W - 2));
  __shared__ float us[MR_LH * MA_LW], vs[MR_LH * MA_LW], ps[MR_LH * MA_LW], es[MR_LH * MA_LW];
  float ax = 0, ay = 0;      // per-thread f32 partials (see k_loss)
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int i0 = (tile / tiles_x) * MR_TH, j0 = (tile % tiles_x) * MA_TW;
    __syncthreads();                                    // the previous tile has been consumed
    for (int t = threadIdx.x; t < MR_LH * MA_LW; t += 256) {
      const int li = t / MA_LW, lj = t - li * MA_LW, i = i0 + li - 1, j = j0 + lj - 1;
      const bool in = i >= 0 && i < H && j >= 0 && j < W;
      const size_t o = in ? (size_t)i * W + j : 0;
      float e = 0.f;
      if (in) {
        e = fminf(fmaxf(expf(-lnfkt * T[o] + lnfkp * (1.0f - yc[o])), 1e-8f), 1.0f);
        if (li >= 1 && li <= MR_TH && lj >= 1 && lj <= MA_TW) eta_out[(size_t)n * HW + o] = e;
      }
      us[t] = in ? s * u[o] : 0.f;
      vs[t] = in ? s * v[o] : 0.f;
      ps[t] = (in && p) ? p[o] : 0.f;
      es[t] = e;
    }
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < MR_TH / 4; ++r) {
      const int li = (threadIdx.x >> 6) + 4 * r, lj = threadIdx.x & 63, i = i0 + li, j = j0 + lj;
      if (i >= H || j >= W) continue;
      const size_t idx = (size_t)i * W + j;
      float ox = 0.f, oy = 0.f;
      if (i >= 1 && i <= H - 2 && j >= 1 && j <= W - 2) {     // every face below lies inside the domain
        const int cc = (li + 1) * MA_LW + lj + 1;
        auto U = [&](int di, int dj) { return us[cc + di * MA_LW + dj]; };
        auto V = [&](int di, int dj) { return vs[cc + di * MA_LW + dj]; };
        auto P = [&](int di, int dj) { return ps[cc + di * MA_LW + dj]; };
        auto ET = [&](int di, int dj) { return es[cc + di * MA_LW + dj]; };
        auto exf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di, dj + 1)); };   // x-face (i+di, j+dj+1/2)
        auto eyf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di + 1, dj)); };   // y-face (i+di+1/2, j+dj)
        auto dVdx = [&](int di, int dj) { return 0.5f * (V(di, dj + 1) - V(di, dj - 1)) * ih; };
        auto dUdy = [&](int di, int dj) { return 0.5f * (U(di + 1, dj) - U(di - 1, dj)) * ih; };
        float Fx1 = 2.f * exf(0, 0) * (U(0, 1) - U(0, 0)) * ih;
        float Fx0 = 2.f * exf(0, -1) * (U(0, 0) - U(0, -1)) * ih;
        float Ty1 = eyf(0, 0) * ((U(1, 0) - U(0, 0)) * ih + 0.5f * (dVdx(0, 0) + dVdx(1, 0)));
        float Ty0 = eyf(-1, 0) * ((U(0, 0) - U(-1, 0)) * ih + 0.5f * (dVdx(-1, 0) + dVdx(0, 0)));
        float Rx = -0.5f * (P(0, 1) - P(0, -1)) * ih + (Fx1 - Fx0) * ih + (Ty1 - Ty0) * ih;
        float Fy1 = 2.f * eyf(0, 0) * (V(1, 0) - V(0, 0)) * ih;
        float Fy0 = 2.f * eyf(-1, 0) * (V(0, 0) - V(-1, 0)) * ih;
        float Tx1 = exf(0, 0) * ((V(0, 1) - V(0, 0)) * ih + 0.5f * (dUdy(0, 0) + dUdy(0, 1)));
        float Tx0 = exf(0, -1) * ((V(0, 0) - V(0, -1)) * ih + 0.5f * (dUdy(0, -1) + dUdy(0, 0)));
        float Ry = -0.5f * (P(1, 0) - P(-1, 0)) * ih + (Fy1 - Fy0) * ih + (Tx1 - Tx0) * ih + g.ra * T[idx];
        ax += fabsf(Rx); ay += fabsf(Ry);
        ox = c * sgn(Rx); oy = c * sgn(Ry);
      }
      sx_[(size_t)n * HW + idx] = ox;
      sy_[(size_t)n * HW + idx] = oy;
    }
  }
  BlockSums bs;
  bs.v[0] = ax; bs.v[1] = ay;
  const int slots[2] = {MC_S_MOMX, MC_S_MOMY};
  bs.flush<2>(sums, slots);
}

// Adjoint of the residual stencils, LDS-tiled: an 8 x 64 pixel tile with a one-pixel halo of S_x, S_y and eta is
// staged once (zero outside the domain) and every face quantity is formed from LDS.  No domain masks are needed:
// S is zero on the outermost ring and beyond (k_mom_residual writes 0 there), so every difference of S that an
// out-of-domain face would multiply is itself zero, and the zero-filled eta keeps those faces finite.
// (The straight per-pixel form issued ~40 bounds-checked global loads per pixel: 235 us at 32 x 506 x 506.)
__global__ __launch_bounds__(256) void k_mom_adjoint(MomGeom g, const float* __restrict__ eta_,
                                                     const float* __restrict__ scaler,
                                                     const float* __restrict__ sx_, const float* __restrict__ sy_,
                                                     float* __restrict__ gu_, float* __restrict__ gv_,
                                                     float* __restrict__ gp_, float* __restrict__ gT_, int t_grad, int tiles_x) {
  const int H = g.H, W = g.W, HW = H * W, n = blockIdx.y;
  const int i0 = (blockIdx.x / tiles_x) * MA_TH, j0 = (blockIdx.x % tiles_x) * MA_TW;
  const float* sx = sx_ + (size_t)n * HW;
  const float* sy = sy_ + (size_t)n * HW;
  const float* et = eta_ + (size_t)n * HW;
  __shared__ float sxs[MA_LH * MA_LW], sys[MA_LH * MA_LW], ets[MA_LH * MA_LW];
  for (int t = threadIdx.x; t < MA_LH * MA_LW; t += 256) {
    const int li = t / MA_LW, lj = t - li * MA_LW, i = i0 + li - 1, j = j0 + lj - 1;
    const bool in = i >= 0 && i < H && j >= 0 && j < W;
    const size_t o = in ? (size_t)i * W + j : 0;
    sxs[t] = in ? sx[o] : 0.f;
    sys[t] = in ? sy[o] : 0.f;
    ets[t] = in ? et[o] : 0.f;
  }
  __syncthreads();
  const float s = scaler[n], ih = g.ih;
#pragma unroll 2
  for (int r = 0; r < MA_TH / 4; ++r) {
    const int li = (threadIdx.x >> 6) + 4 * r, lj = threadIdx.x & 63, i = i0 + li, j = j0 + lj;
    if (i >= H || j >= W) continue;
    const int c = (li + 1) * MA_LW + lj + 1;           // this pixel in the tile
    auto SX = [&](int di, int dj) { return sxs[c + di * MA_LW + dj]; };
    auto SY = [&](int di, int dj) { return sys[c + di * MA_LW + dj]; };
    auto ET = [&](int di, int dj) { return ets[c + di * MA_LW + dj]; };
    auto exf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di, dj + 1)); };     // x-face (i+di, j+dj+1/2)
    auto eyf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di + 1, dj)); };     // y-face (i+di+1/2, j+dj)
    auto dFx = [&](int di, int dj) { return ih * (SX(di, dj) - SX(di, dj + 1)); };
    auto dTy = [&](int di, int dj) { return ih * (SX(di, dj) - SX(di + 1, dj)); };
    auto dFy = [&](int di, int dj) { return ih * (SY(di, dj) - SY(di + 1, dj)); };
    auto dTx = [&](int di, int dj) { return ih * (SY(di, dj) - SY(di, dj + 1)); };
    auto E = [&](int di, int dj) { return eyf(di, dj) * dTy(di, dj); };
    auto Fh = [&](int di, int dj) { return exf(di, dj) * dTx(di, dj); };
    const float dU = 2.f * ih * (exf(0, -1) * dFx(0, -1) - exf(0, 0) * dFx(0, 0))
                   + ih * (E(-1, 0) - E(0, 0))
                   + 0.25f * ih * (Fh(-1, 0) - Fh(1, 0) + Fh(-1, -1) - Fh(1, -1));
    const float dV = 2.f * ih * (eyf(-1, 0) * dFy(-1, 0) - eyf(0, 0) * dFy(0, 0))
                   + ih * (Fh(0, -1) - Fh(0, 0))
                   + 0.25f * ih * (E(0, -1) - E(0, 1) + E(-1, -1) - E(-1, 1));
    const float dP = -0.5f * ih * (SX(0, -1) - SX(0, 1)) - 0.5f * ih * (SY(-1, 0) - SY(1, 0));
    const size_t idx = (size_t)i * W + j;
    gu_[(size_t)n * g.pbs + idx] += s * dU;
    gv_[(size_t)n * g.pbs + idx] += s * dV;
    if (gp_) gp_[(size_t)n * g.ppbs + idx] += dP;
    if (gT_ && t_grad) gT_[(size_t)n * g.pbs + idx] += g.ra * SY(0, 0);
  }
}

// out[n * c + j] = scale * sum_b part[n][b][j] (b in order: deterministic), j < c <= 4: the per-block sums of k_loss_fused
__global__ void k_partial_sums_finalize(const float* __restrict__ part, int n, int blocks, int c, float scale, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  const int s = i / c, j = i - s * c;
  float a = 0.f;
  for (int b = 0; b < blocks; ++b) a += part[((size_t)s * blocks + b) * 4 + j];
  out[i] = a * scale;
}

// ------------------------------------------------------------------------------------------------
// k_loss + k_mom_residual + k_mom_adjoint in ONE pass over the fields (Unet branch without the curl head: the predictions are
// channels of the network output).  A tile of 16 x 64 pixels with a halo of 2: u, v, their targets, p, T and the viscosity are
// staged once; the weighted residual signs S_x, S_y of the tile + 1 ring are formed in LDS (never written to memory: the
// separate kernels moved S_x, S_y and eta through HBM, 0.75 GB per step); then every pixel takes its data, derivative,
// divergence and momentum-adjoint gradient from LDS and is written once.  Same per-pixel expressions, in the same order, as
// the three kernels.  CB8IN: the predictions are read straight from the last convolution's f32 output [n][c8][h][cw][8]
// (columns crop .. crop + w, minus the per-(sample, channel) spatial mean; channels u, v, T, p = 0, 1, 2, 3), which saves the
// NCHW copy of the network output; else from planes with a batch stride like k_loss.
// (tile rows, measured at 32 x 506 x 506: 16 -> 252 us, 8 -> 219 us: the kernel is bound by the latency of its dependent LDS
// reads, not by bytes or arithmetic -- 28 KB of LDS and 123 VGPRs give 4 resident blocks per CU; forcing 5 with a register cap
// spilled (230 us); forming every face value once in LDS (four more arrays, three more barriers per tile) cost 314 us)
constexpr int LF_TH = 8, LF_TW = 64, LF_LH = LF_TH + 4, LF_LW = LF_TW + 4, LF_SH = LF_TH + 2, LF_SW = LF_TW + 2;
struct FusedIn {
  const float* u; const float* v; const float* p; const float* T;     // planes (CB8IN = false)
  const float* y8; const float* mean; int cw, crop, c, c8;           // CB8 f32 network output (CB8IN = true)
};
template <bool CB8IN, bool MOM>
__global__ __launch_bounds__(256) void k_loss_fused(LossGeom g, FusedIn in, const float* __restrict__ uvp,
                                                    const float* __restrict__ mm, const float* __restrict__ yc,
                                                    const float* __restrict__ paras, const float* __restrict__ scaler,
                                                    double* __restrict__ sums, float* __restrict__ gu_, float* __restrict__ gv_,
                                                    float* __restrict__ gp_, float* __restrict__ gT_, int tiles_x, int tiles,
                                                    float* __restrict__ gsum_part) {
  const int H = g.d.h, W = g.d.w, HW = H * W, n = blockIdx.y;
  const bool hasP = g.d.p_pred != 0;
  float q_u = 0.f, q_v = 0.f, q_T = 0.f, q_p = 0.f;             // this thread's share of the spatial sums of the gradient planes
  const float* ut = uvp + ((size_t)n * g.ct + 0) * HW;
  const float* vt = uvp + ((size_t)n * g.ct + 1) * HW;
  const float* pt = hasP ? uvp + ((size_t)n * g.ct + 2) * HW : nullptr;
  const float* Tt = uvp + ((size_t)n * g.ct + (hasP ? 3 : 2)) * HW;
  float mu = 0.f, mv = 0.f, mT = 0.f, mp = 0.f;
  if (CB8IN && in.mean) { mu = in.mean[n * in.c]; mv = in.mean[n * in.c + 1]; mT = in.mean[n * in.c + 2]; mp = hasP ? in.mean[n * in.c + 3] : 0.f; }
  // (u, v, T, p) of pixel (yy, xx) of this sample
  auto fetch = [&](int yy, int xx, float& fu, float& fv, float& fT, float& fp) {
    if constexpr (CB8IN) {
      const float4 q = *reinterpret_cast<const float4*>(in.y8 + ((((size_t)n * in.c8) * H + yy) * in.cw + xx + in.crop) * 8);
      fu = q.x - mu; fv = q.y - mv; fT = q.z - mT; fp = hasP ? q.w - mp : 0.f;
    } else {
      const size_t o = (size_t)yy * W + xx;
      fu = in.u[(size_t)n * g.pbs + o]; fv = in.v[(size_t)n * g.pbs + o]; fT = in.T[(size_t)n * g.pbs + o];
      fp = hasP ? in.p[(size_t)n * g.ppbs + o] : 0.f;
    }
  };
  const float k = hasP ? 4.f : 3.f;
  const float NHW = (float)g.d.n * (float)HW;
  const float cdat = 1.0f / (k * NHW);
  float su = 1.f, sv = 1.f;
  if (g.d.loss_scale) {
    su = fminf(fmaxf(1.0f / (mm[(n * 3 + 0) * 2 + 1] - mm[(n * 3 + 0) * 2 + 0]), 1.0f), 10.0f);
    sv = fminf(fmaxf(1.0f / (mm[(n * 3 + 1) * 2 + 1] - mm[(n * 3 + 1) * 2 + 0]), 1.0f), 10.0f);
  }
  const float cdu = 126.0f / (k * (float)g.d.n * (float)(H - 2) * (float)W);
  const float cdv = 126.0f / (k * (float)g.d.n * (float)H * (float)(W - 2));
  const float wm = 1.0f / ((float)g.d.n * (float)(H - 2) * (float)(W - 2));
  const float wc = 1.0f / ((float)g.d.n * (float)(H - 2)), wr = 1.0f / ((float)g.d.n * (float)(W - 2));
  float lnfkt = 0.f, lnfkp = 0.f, s = 1.f;
  if (MOM) { lnfkt = logf(paras[n * 3 + 1]); lnfkp = logf(paras[n * 3 + 2]); s = scaler[n]; }
  const float ih = g.d.inv_h;
  const float cmom = g.d.lambda_mom / ((float)g.d.n * (float)(H - 2) * (float)(W - 2));
  float a_us = 0, a_up = 0, a_vs = 0, a_vp = 0, a_pp = 0, a_tp = 0, a_du = 0, a_dv = 0;
  float a_m = 0, a_mx0 = 0, a_mx1 = 0, a_my0 = 0, a_my1 = 0, a_rx = 0, a_ry = 0;
  __shared__ float us[LF_LH * LF_LW], vs[LF_LH * LF_LW], uts[LF_LH * LF_LW], vts[LF_LH * LF_LW];
  __shared__ float ps[MOM ? LF_LH * LF_LW : 1], es[MOM ? LF_LH * LF_LW : 1], Ts[LF_LH * LF_LW];
  __shared__ float sxs[MOM ? LF_SH * LF_SW : 1], sys[MOM ? LF_SH * LF_SW : 1];
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int y0 = (tile / tiles_x) * LF_TH, x0 = (tile % tiles_x) * LF_TW;
    __syncthreads();                                    // the previous tile has been consumed
    for (int t = threadIdx.x; t < LF_LH * LF_LW; t += 256) {
      const int ly = t / LF_LW, lx = t - ly * LF_LW, yy = y0 + ly - 2, xx = x0 + lx - 2;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      float fu = 0.f, fv = 0.f, fT = 0.f, fp = 0.f, e = 0.f, tu = 0.f, tv = 0.f;
      if (ok) {
        const size_t o = (size_t)yy * W + xx;
        fetch(yy, xx, fu, fv, fT, fp);
        tu = ut[o]; tv = vt[o];
        if (MOM) e = fminf(fmaxf(expf(-lnfkt * fT + lnfkp * (1.0f - yc[o])), 1e-8f), 1.0f);
      }
      us[t] = fu; vs[t] = fv; uts[t] = tu; vts[t] = tv; Ts[t] = fT;
      if (MOM) { ps[t] = fp; es[t] = e; }
    }
    __syncthreads();
    if (MOM) {
      // weighted residual signs of the tile + 1 ring (0 outside the interior of the domain); |R| is summed over the tile only
      for (int t = threadIdx.x; t < LF_SH * LF_SW; t += 256) {
        const int li = t / LF_SW, lj = t - li * LF_SW, i = y0 + li - 1, j = x0 + lj - 1;
        float ox = 0.f, oy = 0.f;
        if (i >= 1 && i <= H - 2 && j >= 1 && j <= W - 2) {
          const int cc = (li + 1) * LF_LW + lj + 1;
          auto U = [&](int di, int dj) { return s * us[cc + di * LF_LW + dj]; };
          auto V = [&](int di, int dj) { return s * vs[cc + di * LF_LW + dj]; };
          auto P = [&](int di, int dj) { return ps[cc + di * LF_LW + dj]; };
          auto ET = [&](int di, int dj) { return es[cc + di * LF_LW + dj]; };
          auto exf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di, dj + 1)); };
          auto eyf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di + 1, dj)); };
          auto dVdx = [&](int di, int dj) { return 0.5f * (V(di, dj + 1) - V(di, dj - 1)) * ih; };
          auto dUdy = [&](int di, int dj) { return 0.5f * (U(di + 1, dj) - U(di - 1, dj)) * ih; };
          float Fx1 = 2.f * exf(0, 0) * (U(0, 1) - U(0, 0)) * ih;
          float Fx0 = 2.f * exf(0, -1) * (U(0, 0) - U(0, -1)) * ih;
          float Ty1 = eyf(0, 0) * ((U(1, 0) - U(0, 0)) * ih + 0.5f * (dVdx(0, 0) + dVdx(1, 0)));
          float Ty0 = eyf(-1, 0) * ((U(0, 0) - U(-1, 0)) * ih + 0.5f * (dVdx(-1, 0) + dVdx(0, 0)));
          float Rx = -0.5f * (P(0, 1) - P(0, -1)) * ih + (Fx1 - Fx0) * ih + (Ty1 - Ty0) * ih;
          float Fy1 = 2.f * eyf(0, 0) * (V(1, 0) - V(0, 0)) * ih;
          float Fy0 = 2.f * eyf(-1, 0) * (V(0, 0) - V(-1, 0)) * ih;
          float Tx1 = exf(0, 0) * ((V(0, 1) - V(0, 0)) * ih + 0.5f * (dUdy(0, 0) + dUdy(0, 1)));
          float Tx0 = exf(0, -1) * ((V(0, 0) - V(0, -1)) * ih + 0.5f * (dUdy(0, -1) + dUdy(0, 0)));
          float Ry = -0.5f * (P(1, 0) - P(-1, 0)) * ih + (Fy1 - Fy0) * ih + (Tx1 - Tx0) * ih + g.d.ra * Ts[cc];
          if (li >= 1 && li <= LF_TH && lj >= 1 && lj <= LF_TW) { a_rx += fabsf(Rx); a_ry += fabsf(Ry); }
          ox = cmom * sgn(Rx); oy = cmom * sgn(Ry);
        }
        sxs[t] = ox; sys[t] = oy;
      }
      __syncthreads();
    }
#pragma unroll 2
    for (int r = 0; r < LF_TH / 4; ++r) {
      const int ly = (threadIdx.x >> 6) + 4 * r, lx = threadIdx.x & 63, y = y0 + ly, x = x0 + lx;
      if (y >= H || x >= W) continue;
      const int i = y * W + x;
      const int cc = (ly + 2) * LF_LW + lx + 2;
      auto U = [&](int dy, int dx) { return us[cc + dy * LF_LW + dx]; };
      auto V = [&](int dy, int dx) { return vs[cc + dy * LF_LW + dx]; };
      auto UT = [&](int dy, int dx) { return uts[cc + dy * LF_LW + dx]; };
      auto VT = [&](int dy, int dx) { return vts[cc + dy * LF_LW + dx]; };
      auto mass_wsgn = [&](int dy, int dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 1 || yy > H - 2 || xx < 1 || xx > W - 2) return 0.f;
        const float D = 0.5f * (U(dy, dx + 1) - U(dy, dx - 1)) + 0.5f * (V(dy + 1, dx) - V(dy - 1, dx));
        const float w = g.d.loss_type == 1 ? wm
                      : (xx == 1 ? wc : 0.f) + (xx == W - 2 ? wc : 0.f) + (yy == 1 ? wr : 0.f) + (yy == H - 2 ? wr : 0.f);
        return w * sgn(D);
      };
      const float bw = (g.d.loss_scale && (y < 2 || y >= H - 2 || x < 2 || x >= W - 2)) ? 11.f : 1.f;
      float gu = 0.f, gv = 0.f, gp = 0.f, gT = 0.f;
      {  // data terms
        float du = UT(0, 0) - U(0, 0), dv = VT(0, 0) - V(0, 0), dT = Tt[i] - Ts[cc];
        float wu = su * bw, wv = sv * bw;
        if (!g.d.l2) {
          a_us += fabsf(du * wu); a_up += fabsf(du);
          a_vs += fabsf(dv * wv); a_vp += fabsf(dv);
          a_tp += fabsf(dT);
          gu -= sgn(du) * wu * cdat; gv -= sgn(dv) * wv * cdat; gT -= sgn(dT) * cdat;
        } else {
          a_us += (du * wu) * (du * wu); a_up += du * du;
          a_vs += (dv * wv) * (dv * wv); a_vp += dv * dv;
          a_tp += dT * dT;
          gu -= 2.f * du * wu * wu * cdat; gv -= 2.f * dv * wv * wv * cdat; gT -= 2.f * dT * cdat;
        }
        if (hasP) {
          float pv;
          if (MOM) pv = ps[cc]; else { float a_, b_, c_; fetch(y, x, a_, b_, c_, pv); }
          float dp = pt[i] - pv;
          if (!g.d.l2) { a_pp += fabsf(dp); gp -= sgn(dp) * cdat; }
          else { a_pp += dp * dp; gp -= 2.f * dp * cdat; }
        }
      }
      if (g.d.loss_derivative) {
        if (y <= H - 3) {
          float e = (UT(1, 0) - UT(0, 0)) - (U(1, 0) - U(0, 0));
          a_du += fabsf(126.0f * e);
          gu += cdu * sgn(e);
        }
        if (y >= 1 && y <= H - 2) {
          float e = (UT(0, 0) - UT(-1, 0)) - (U(0, 0) - U(-1, 0));
          gu -= cdu * sgn(e);
        }
        if (x <= W - 3) {
          float e = (VT(0, 1) - VT(0, 0)) - (V(0, 1) - V(0, 0));
          a_dv += fabsf(126.0f * e);
          gv += cdv * sgn(e);
        }
        if (x >= 1 && x <= W - 2) {
          float e = (VT(0, 0) - VT(0, -1)) - (V(0, 0) - V(0, -1));
          gv -= cdv * sgn(e);
        }
      }
      if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
        float D = 0.5f * (U(0, 1) - U(0, -1)) + 0.5f * (V(1, 0) - V(-1, 0));
        float m = fabsf(D);
        a_m += m;
        if (x == 1) a_mx0 += m;
        if (x == W - 2) a_mx1 += m;
        if (y == 1) a_my0 += m;
        if (y == H - 2) a_my1 += m;
      }
      if (g.d.loss_type != 0) {
        gu += 0.5f * (mass_wsgn(0, -1) - mass_wsgn(0, 1));
        gv += 0.5f * (mass_wsgn(-1, 0) - mass_wsgn(1, 0));
      }
      if (g.d.t_grad == 0) gT = 0.f;
      if (MOM) {
        const int c = (ly + 1) * LF_SW + lx + 1;           // this pixel in the S tile; eta through cc
        auto SX = [&](int di, int dj) { return sxs[c + di * LF_SW + dj]; };
        auto SY = [&](int di, int dj) { return sys[c + di * LF_SW + dj]; };
        auto ET = [&](int di, int dj) { return es[cc + di * LF_LW + dj]; };
        auto exf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di, dj + 1)); };
        auto eyf = [&](int di, int dj) { return 0.5f * (ET(di, dj) + ET(di + 1, dj)); };
        auto dFx = [&](int di, int dj) { return ih * (SX(di, dj) - SX(di, dj + 1)); };
        auto dTy = [&](int di, int dj) { return ih * (SX(di, dj) - SX(di + 1, dj)); };
        auto dFy = [&](int di, int dj) { return ih * (SY(di, dj) - SY(di + 1, dj)); };
        auto dTx = [&](int di, int dj) { return ih * (SY(di, dj) - SY(di, dj + 1)); };
        auto E = [&](int di, int dj) { return eyf(di, dj) * dTy(di, dj); };
        auto Fh = [&](int di, int dj) { return exf(di, dj) * dTx(di, dj); };
        const float dU = 2.f * ih * (exf(0, -1) * dFx(0, -1) - exf(0, 0) * dFx(0, 0))
                       + ih * (E(-1, 0) - E(0, 0))
                       + 0.25f * ih * (Fh(-1, 0) - Fh(1, 0) + Fh(-1, -1) - Fh(1, -1));
        const float dV = 2.f * ih * (eyf(-1, 0) * dFy(-1, 0) - eyf(0, 0) * dFy(0, 0))
                       + ih * (Fh(0, -1) - Fh(0, 0))
                       + 0.25f * ih * (E(0, -1) - E(0, 1) + E(-1, -1) - E(-1, 1));
        const float dP = -0.5f * ih * (SX(0, -1) - SX(0, 1)) - 0.5f * ih * (SY(-1, 0) - SY(1, 0));
        gu += s * dU;
        gv += s * dV;
        gp += dP;
        if (g.d.t_grad) gT += g.d.ra * SY(0, 0);
      }
      gu_[(size_t)n * g.pbs + i] = gu;
      gv_[(size_t)n * g.pbs + i] = gv;
      if (gp_) gp_[(size_t)n * g.ppbs + i] = gp;
      gT_[(size_t)n * g.pbs + i] = gT;
      q_u += gu; q_v += gv; q_T += gT; q_p += gp;
    }
  }
  if (gsum_part) {
    // per-block sums of the four gradient planes in a fixed order (thread -> wave -> block): the adjoint of the network's
    // spatial-mean subtraction needs their means, which saves a pass over the gradient tensor (mc_partial_sums_finalize)
    __shared__ float qred[4][4];
    const float r0 = wave_sum(q_u), r1 = wave_sum(q_v), r2 = wave_sum(q_T), r3 = wave_sum(q_p);
    if ((threadIdx.x & 63) == 0) { float* q = qred[threadIdx.x >> 6]; q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; }
    __syncthreads();
    if (threadIdx.x < 4)
      gsum_part[((size_t)n * gridDim.x + blockIdx.x) * 4 + threadIdx.x] =
          ((qred[0][threadIdx.x] + qred[1][threadIdx.x]) + qred[2][threadIdx.x]) + qred[3][threadIdx.x];
    __syncthreads();
  }
  BlockSums bs;
  bs.v[0] = a_us; bs.v[1] = a_up; bs.v[2] = a_vs; bs.v[3] = a_vp; bs.v[4] = a_pp; bs.v[5] = a_tp; bs.v[6] = a_du;
  bs.v[7] = a_dv; bs.v[8] = a_m; bs.v[9] = a_mx0; bs.v[10] = a_mx1; bs.v[11] = a_my0; bs.v[12] = a_my1; bs.v[13] = a_rx;
  bs.v[14] = a_ry;
  const int slots[15] = {MC_S_U_SCALED, MC_S_U_PLAIN, MC_S_V_SCALED, MC_S_V_PLAIN, MC_S_P_PLAIN, MC_S_T_PLAIN, MC_S_DU,
                         MC_S_DV, MC_S_MASS, MC_S_MASS_X0, MC_S_MASS_X1, MC_S_MASS_Y0, MC_S_MASS_Y1, MC_S_MOMX, MC_S_MOMY};
  bs.flush<15>(sums, slots);
}

// on-device batch assembly (ADTimeDataset.__getitem__, datasetio.py:229-280)
__global__ void k_assemble_adtime(const float* __restrict__ T, const float* __restrict__ uv, const float* __restrict__ t,
                                  const float* __restrict__ paras, const float* __restrict__ paras_nd,
                                  const float* __restrict__ xc, const float* __restrict__ yc, const int* __restrict__ pairs,
                                  int cy, int HW, float* __restrict__ x, float* __restrict__ y, float* __restrict__ scaler,
                                  float* __restrict__ paras_out) {
  const int b = blockIdx.y, i0 = pairs[2 * b], i1 = pairs[2 * b + 1];
  const float raq = paras[i0 * 3], fkt = paras[i0 * 3 + 1], fkp = paras[i0 * 3 + 2];
  const float lnfkt = logf(fkt), lnfkp = logf(fkp);
  const float s = 5.0f * expf(raq * 0.1f * 1.80167667f + lnfkt * 0.4330392f + lnfkp * -0.46052953f);
  const float inv_s = 1.0f / s, dt = t[i1] - t[i0];
  const float n0 = paras_nd[i0 * 3], n1 = paras_nd[i0 * 3 + 1], n2 = paras_nd[i0 * 3 + 2];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scaler[b] = s;
    paras_out[b * 3] = raq; paras_out[b * 3 + 1] = fkt; paras_out[b * 3 + 2] = fkp;
  }
  const float* T0 = T + (size_t)i0 * HW;
  const float* T1 = T + (size_t)i1 * HW;
  const float* u0 = uv + ((size_t)i0 * cy + 0) * HW;
  const float* v0 = uv + ((size_t)i0 * cy + 1) * HW;
  const float* u1 = uv + ((size_t)i1 * cy + 0) * HW;
  const float* v1 = uv + ((size_t)i1 * cy + 1) * HW;
  float* xb = x + (size_t)b * 10 * HW;
  float* yb = y + (size_t)b * 3 * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float Tp = T0[i], ycv = yc[i];
    const float eta = expf(-lnfkt * Tp + lnfkp * (1.0f - ycv));
    xb[i] = xc[i];
    xb[HW + i] = ycv;
    xb[2 * (size_t)HW + i] = dt;
    xb[3 * (size_t)HW + i] = n0;
    xb[4 * (size_t)HW + i] = n1;
    xb[5 * (size_t)HW + i] = n2;
    xb[6 * (size_t)HW + i] = log10f(fminf(fmaxf(eta, 1e-8f), 1.0f)) * 0.125f;
    xb[7 * (size_t)HW + i] = Tp;
    xb[8 * (size_t)HW + i] = u0[i] * inv_s;
    xb[9 * (size_t)HW + i] = v0[i] * inv_s;
    yb[i] = u1[i] * inv_s;
    yb[HW + i] = v1[i] * inv_s;
    yb[2 * (size_t)HW + i] = T1[i];
  }
}

// TS 'unet' branch (pytorch_networks_convae.py:411-446): 10-channel input and the wall / side conditions of the predicted T
__global__ void k_ts_build_input_unet(const float* __restrict__ T, const float* __restrict__ xc, const float* __restrict__ yc,
                                      const float* __restrict__ ycc, const float* __restrict__ paras, const float* __restrict__ nd,
                                      const float* __restrict__ dt, const float* __restrict__ up, const float* __restrict__ vp,
                                      int HW, float* __restrict__ out) {
  const int n = blockIdx.y;
  const float lnfkt = logf(paras[n * 3 + 1]), lnfkp = logf(paras[n * 3 + 2]);
  const float n0 = nd[n * 3], n1 = nd[n * 3 + 1], n2 = nd[n * 3 + 2];
  const float* Tn = T + (size_t)n * HW;
  float* o = out + (size_t)n * 10 * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float t = Tn[i];
    const float eta = fminf(fmaxf(expf(-lnfkt * t + lnfkp * (1.0f - ycc[i])), 1e-8f), 1.0f);
    o[i] = xc[i] * 0.25f;
    o[HW + i] = yc[i] * 0.25f;
    o[2 * (size_t)HW + i] = dt[(size_t)n * HW + i];
    o[3 * (size_t)HW + i] = n0;
    o[4 * (size_t)HW + i] = n1;
    o[5 * (size_t)HW + i] = n2;
    o[6 * (size_t)HW + i] = log10f(eta) * 0.125f;
    o[7 * (size_t)HW + i] = t;
    o[8 * (size_t)HW + i] = up[(size_t)n * HW + i];
    o[9 * (size_t)HW + i] = vp[(size_t)n * HW + i];
  }
}
// Trainer.get_loss with roll_forward > 1 (multigpu.py:207-248): the next input keeps channels 0..5 of the batch and takes
// T, u, v from the evaluation just done; the viscosity channel follows T after a pre-step only (update_v), with the depth of the
// UNSCALED yc channel (x holds the batch's raw channels; the input-pack kernel applies xc / 4, yc / 4, dt / R)
__global__ void k_roll_forward_update(float* __restrict__ x, int C, const float* __restrict__ u, const float* __restrict__ v,
                                      const float* __restrict__ T, size_t uvt_stride, const float* __restrict__ paras,
                                      int update_v, int HW) {
  const int n = blockIdx.y;
  const float lnfkt = logf(paras[n * 3 + 1]), lnfkp = logf(paras[n * 3 + 2]);
  float* o = x + (size_t)n * C * HW;
  const float *un = u + n * uvt_stride, *vn = v + n * uvt_stride, *Tn = T + n * uvt_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float t = Tn[i];
    if (update_v) {
      const float eta = fminf(fmaxf(expf(-lnfkt * t + lnfkp * (1.0f - o[HW + i])), 1e-8f), 1.0f);
      o[6 * (size_t)HW + i] = log10f(eta) * 0.125f;
    }
    o[7 * (size_t)HW + i] = t;
    o[8 * (size_t)HW + i] = un[i];
    o[9 * (size_t)HW + i] = vn[i];
  }
}
// T[:, 0, :] = 1, T[:, -1, :] = 0, then the side columns copy their inner neighbours (:441-444); one block per sample
__global__ void k_ts_wall_bc(const float* __restrict__ src, int H, int W, float* __restrict__ dst) {
  const float* s = src + (size_t)blockIdx.x * H * W;
  float* d = dst + (size_t)blockIdx.x * H * W;
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
    const int y = i / W, x = i - y * W;
    const int xs = x == 0 ? 1 : (x == W - 1 ? W - 2 : x);
    d[i] = y == 0 ? 1.f : (y == H - 1 ? 0.f : s[(size_t)y * W + xs]);
  }
}

// on-device batch assembly for the FluidNet family (NewADDataset.__getitem__, datasetio.py:595-654)
__global__ void k_assemble_newad(const float* __restrict__ T, const float* __restrict__ uvp, const float* __restrict__ t,
                                 const float* __restrict__ paras, const float* __restrict__ paras_nd,
                                 const float* __restrict__ xc, const float* __restrict__ yc, const int* __restrict__ idx,
                                 int cy, int HW, float* __restrict__ x, float* __restrict__ y, float* __restrict__ tw,
                                 float* __restrict__ scaler) {
  const int b = blockIdx.y, i0 = idx[b];
  const float raq = paras[i0 * 3], fkt = paras[i0 * 3 + 1], fkp = paras[i0 * 3 + 2];
  const float lnfkt = logf(fkt), lnfkp = logf(fkp);
  const float s = 5.0f * expf(raq * 0.1f * 1.80167667f + lnfkt * 0.4330392f + lnfkp * -0.46052953f);
  const float inv_s = 1.0f / s;
  const float n0 = paras_nd[i0 * 3], n1 = paras_nd[i0 * 3 + 1], n2 = paras_nd[i0 * 3 + 2];
  if (blockIdx.x == 0 && threadIdx.x == 0) { scaler[b] = s; tw[b] = t[i0]; }
  const float* T0 = T + (size_t)i0 * HW;
  const float* yi = uvp + (size_t)i0 * cy * HW;
  float* xb = x + (size_t)b * 7 * HW;
  float* yb = y + (size_t)b * cy * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float Tp = T0[i], ycv = yc[i];
    const float eta = expf(-lnfkt * Tp + lnfkp * (1.0f - ycv));
    xb[i] = xc[i] * 0.25f;
    xb[HW + i] = ycv * 0.25f;
    xb[2 * (size_t)HW + i] = log10f(fminf(fmaxf(eta, 1e-8f), 1.0f)) * 0.125f;
    xb[3 * (size_t)HW + i] = n0;
    xb[4 * (size_t)HW + i] = n1;
    xb[5 * (size_t)HW + i] = n2;
    xb[6 * (size_t)HW + i] = Tp;
    yb[i] = yi[i] * inv_s;
    yb[HW + i] = yi[HW + i] * inv_s;
    if (cy > 2) yb[2 * (size_t)HW + i] = yi[2 * (size_t)HW + i];
  }
}

// ------------------------------------------------------------------------------------------------
// inference rollout: TS input builder and the ADNet step (pytorch_networks_convae.py:372-395, 522-568)
// ------------------------------------------------------------------------------------------------
__global__ void k_ts_build_input(const float* __restrict__ T, const float* __restrict__ xc, const float* __restrict__ yc,
                                 const float* __restrict__ ycc, const float* __restrict__ paras,
                                 const float* __restrict__ nd, int HW, float* __restrict__ out) {
  const int n = blockIdx.y;
  const float lnfkt = logf(paras[n * 3 + 1]), lnfkp = logf(paras[n * 3 + 2]);
  const float n0 = nd[n * 3], n1 = nd[n * 3 + 1], n2 = nd[n * 3 + 2];
  const float* Tn = T + (size_t)n * HW;
  float* o = out + (size_t)n * 7 * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float t = Tn[i];
    const float eta = fminf(fmaxf(expf(-lnfkt * t + lnfkp * (1.0f - ycc[i])), 1e-8f), 1.0f);
    o[i] = xc[i] * 0.25f;
    o[HW + i] = yc[i] * 0.25f;
    o[2 * (size_t)HW + i] = log10f(eta) * 0.125f;
    o[3 * (size_t)HW + i] = n0;
    o[4 * (size_t)HW + i] = n1;
    o[5 * (size_t)HW + i] = n2;
    o[6 * (size_t)HW + i] = t;
  }
}

struct AdGeom { int N, H, W; int64_t uvs; float cn; };
__device__ __forceinline__ float ad_x(const float* xc, int i, int j, int W) { return j == 0 ? 0.f : (j == W - 1 ? 4.f : xc[(size_t)i * W + j]); }
__device__ __forceinline__ float ad_y(const float* yc, int i, int j, int H, int W) { return i == 0 ? 0.f : (i == H - 1 ? 1.f : yc[(size_t)i * W + j]); }

// ws[0] = bits of max |u|,|v| over the interior (uint order == float order for non-negative floats), ws[1] = bits of min dx_l
__global__ void k_adnet_reduce(AdGeom g, const float* __restrict__ u_, const float* __restrict__ v_,
                               const float* __restrict__ vs, const float* __restrict__ xc, uint32_t* __restrict__ ws) {
  const int n = blockIdx.y, H = g.H, W = g.W;
  const float s = vs ? vs[n] : 1.f;
  const float* u = u_ + (size_t)n * g.uvs;
  const float* v = v_ + (size_t)n * g.uvs;
  float mx = 0.f, mn = 3.4e38f;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (H - 2) * (W - 2); idx += gridDim.x * blockDim.x) {
    const int i = idx / (W - 2) + 1, j = idx % (W - 2) + 1;
    mx = fmaxf(mx, fmaxf(fabsf(s * u[(size_t)i * W + j]), fabsf(s * v[(size_t)i * W + j])));
    mn = fminf(mn, ad_x(xc, i, j, W) - ad_x(xc, i, j - 1, W));
  }
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(ws, __float_as_uint(mx));
    atomicMin(ws + 1, __float_as_uint(mn));
  }
}
__global__ void k_adnet_dt(float cn, const uint32_t* __restrict__ ws, float* __restrict__ dt) {
  const float uv = __uint_as_float(ws[0]), dx = __uint_as_float(ws[1]);
  const float dx2 = dx * dx;
  dt[0] = fminf(0.5f * cn * dx / uv, 0.5f * (dx2 * dx2) / (dx2 + dx2));
}
__global__ void k_adnet_ws_init(uint32_t* ws) { ws[0] = 0u; ws[1] = 0x7f7fffffu; }

__global__ __launch_bounds__(256) void k_adnet_step(AdGeom g, const float* __restrict__ u_, const float* __restrict__ v_,
                                                    const float* __restrict__ vs, const float* __restrict__ T_,
                                                    const float* __restrict__ rq_, const float* __restrict__ rqs,
                                                    const float* __restrict__ xc, const float* __restrict__ yc,
                                                    const float* __restrict__ dtp, float* __restrict__ out_) {
  const int n = blockIdx.y, H = g.H, W = g.W;
  const float s = vs ? vs[n] : 1.f, dt = dtp[0];
  const float* u = u_ + (size_t)n * g.uvs;
  const float* v = v_ + (size_t)n * g.uvs;
  const float* T = T_ + (size_t)n * H * W;
  const float* rq = rq_ ? rq_ + (size_t)n * H * W : nullptr;
  const float rqc = rqs ? rqs[n] : 0.f;
  float* out = out_ + (size_t)n * H * W;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * W; idx += gridDim.x * blockDim.x) {
    const int i = idx / W, j = idx - i * W;
    float r;
    if (i == 0) r = 1.0f;
    else if (i == H - 1) r = 0.0f;
    else {
      const int jj = min(max(j, 1), W - 2);                   // side columns replicate their neighbour
      const size_t c = (size_t)i * W + jj;
      const float Tc = T[c], uu = s * u[c], vv = s * v[c];
      const float xm = ad_x(xc, i, jj - 1, W), x0 = ad_x(xc, i, jj, W), xp = ad_x(xc, i, jj + 1, W);
      const float ym = ad_y(yc, i - 1, jj, H, W), y0 = ad_y(yc, i, jj, H, W), yp = ad_y(yc, i + 1, jj, H, W);
      const float dxl = x0 - xm, dxr = xp - x0, dyt = y0 - ym, dyb = yp - y0;
      const float gl = (Tc - T[c - 1]) / dxl, gr = (T[c + 1] - Tc) / dxr;
      const float gt = (Tc - T[c - W]) / dyt, gb = (T[c + W] - Tc) / dyb;
      const float dTdx = (uu > 0.f ? gl : 0.f) + (uu < 0.f ? gr : 0.f);
      const float dTdy = (vv > 0.f ? gt : 0.f) + (vv < 0.f ? gb : 0.f);
      const float lap = (gr - gl) / (0.5f * dxr + 0.5f * dxl) + (gb - gt) / (0.5f * dyb + 0.5f * dyt);
      r = Tc + dt * (-uu * dTdx - vv * dTdy + lap + (rq ? rq[c] : rqc));
    }
    out[idx] = r;
  }
}

__global__ void k_loss_finalize(mc_loss_desc d, const double* __restrict__ s, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double N = d.n, H = d.h, W = d.w, NHW = N * H * W;
  double lu = s[MC_S_U_SCALED] / NHW, lv = s[MC_S_V_SCALED] / NHW;
  double tu = s[MC_S_U_PLAIN] / NHW, tv = s[MC_S_V_PLAIN] / NHW;
  if (!d.loss_scale) { lu = tu; lv = tv; }
  double lp = d.p_pred ? s[MC_S_P_PLAIN] / NHW : 0.0, lT = s[MC_S_T_PLAIN] / NHW;
  if (d.loss_derivative) {
    lu += s[MC_S_DU] / (N * (H - 2) * W);
    lv += s[MC_S_DV] / (N * H * (W - 2));
    if (!d.loss_scale) { tu = lu; tv = lv; }   // reference aliasing quirk (multigpu.py:283-284)
  }
  if (d.t_grad < 0) lT = 0.0;                                     // no temperature output: averaged over 3 / 2 terms (multigpu.py:173-177)
  double loss = (lu + lv + lp + lT) / ((d.p_pred ? 3.0 : 2.0) + (d.t_grad < 0 ? 0.0 : 1.0));
  double mass = s[MC_S_MASS] / (N * (H - 2) * (W - 2));
  if (d.loss_type == 1) loss += mass;
  else if (d.loss_type == 2)
    loss += (s[MC_S_MASS_X0] + s[MC_S_MASS_X1]) / (N * (H - 2)) + (s[MC_S_MASS_Y0] + s[MC_S_MASS_Y1]) / (N * (W - 2));
  double mom = (s[MC_S_MOMX] + s[MC_S_MOMY]) / (N * (H - 2) * (W - 2));
  if (d.lambda_mom != 0.f) loss += (double)d.lambda_mom * mom;
  out[0] = (float)loss; out[1] = (float)tu; out[2] = (float)tv; out[3] = (float)lp; out[4] = (float)lT;
  out[5] = (float)mass; out[6] = (float)mom; out[7] = 0.f;
}

// workgroups per sample for the reducing loss kernels: every block ends in one f64 atomic per slot on a handful of
// addresses (serialised at ~10 ns each), so keep the total near 2-4 blocks per CU and grid-stride the pixels
int loss_blocks(int hw, int n) {
  static const int total = [] { const char* e = getenv("MC_LOSS_BLOCKS"); return e ? atoi(e) : 1024; }();   // A/B knob
  int b = cdiv(total, n);
  int most = cdiv(hw, 256);
  if (b > most) b = most;
  return b < 1 ? 1 : b;
}

int check_loss_desc(const mc_loss_desc* d) {
  if (!d || d->n <= 0 || d->h < 5 || d->w < 5) return MC_EINVAL;
  if (d->loss_type < 0 || d->loss_type > 2) return MC_EINVAL;
  return MC_OK;
}

}  // namespace

extern "C" {

int mc_loss_minmax(const float* uvp, int32_t n, int32_t ct, int32_t h, int32_t w, float* mm, void* stream) {
  if (!uvp || !mm || n <= 0 || ct < 2 || h <= 0 || w <= 0) return MC_EINVAL;
  hipLaunchKernelGGL(k_minmax, dim3(n, ct >= 3 ? 3 : 2), dim3(256), 0, (hipStream_t)stream, uvp, ct, h * w, mm);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_loss_fwd_bwd(const mc_loss_desc* d, const float* u, const float* v, const float* p, const float* T,
                    int64_t pbs, int64_t ppbs, const float* uvp, const float* mm, double* sums, float* gu, float* gv,
                    float* gp, float* gT, void* stream) {
  int rc = check_loss_desc(d);
  if (rc) return rc;
  const bool hasT = d->t_grad >= 0;
  if (!u || !v || !uvp || !sums || !gu || !gv || (hasT && (!T || !gT))) return MC_EINVAL;
  if (d->p_pred && (!p || !gp)) return MC_EINVAL;
  if (d->loss_scale && !mm) return MC_EINVAL;
  LossGeom g;
  g.d = *d; g.ct = (d->p_pred ? 3 : 2) + (hasT ? 1 : 0); g.pbs = pbs; g.ppbs = ppbs;
  const int tiles_x = cdiv(d->w, LT_TW), tiles = tiles_x * cdiv(d->h, LT_TH);
  dim3 grid(min(loss_blocks(d->h * d->w, d->n), tiles), d->n);
  hipLaunchKernelGGL(k_loss, grid, dim3(256), 0, (hipStream_t)stream, g, u, v, d->p_pred ? p : nullptr, T, uvp, mm, sums,
                     gu, gv, d->p_pred ? gp : nullptr, hasT ? gT : nullptr, tiles_x, tiles);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int32_t mc_loss_fused_blocks(int32_t n, int32_t h, int32_t w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  static const int total = [] { const char* e = getenv("MC_LOSS_FUSED_BLOCKS"); return e ? atoi(e) : 2048; }();
  const int tiles = cdiv(w, LF_TW) * cdiv(h, LF_TH);
  return max(1, min(cdiv(total, n), tiles));
}

int mc_partial_sums_finalize(const float* part, int32_t n, int32_t blocks, int32_t c, float scale, float* out, void* stream) {
  if (!part || !out || n <= 0 || blocks <= 0 || c <= 0 || c > 4) return MC_EINVAL;
  hipLaunchKernelGGL(k_partial_sums_finalize, dim3(cdiv(n * c, 128)), dim3(128), 0, (hipStream_t)stream, part, n, blocks, c, scale, out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_loss_fused(const mc_loss_desc* d, const float* u, const float* v, const float* p, const float* T, int64_t pbs, int64_t ppbs,
                  const float* y_cb8, int32_t cb8_w, int32_t cb8_crop, const float* cb8_mean, int32_t cb8_c, const float* uvp,
                  const float* mm, const float* yc, const float* paras, const float* scaler, double* sums, float* gu, float* gv,
                  float* gp, float* gT, int64_t g_pbs, int64_t g_ppbs, float* gsum_part, void* stream) {
  int rc = check_loss_desc(d);
  if (rc) return rc;
  if (d->t_grad < 0 || d->loss_type == 2) return MC_EUNSUPPORTED;       // needs the temperature channel; no curl head
  if (!uvp || !sums || !gu || !gv || !gT || (d->p_pred && !gp)) return MC_EINVAL;
  if (d->loss_scale && !mm) return MC_EINVAL;
  const bool mom = d->lambda_mom != 0.f;
  if (mom && (!yc || !paras || !scaler)) return MC_EINVAL;
  FusedIn in{};
  if (y_cb8) {
    const int need = d->p_pred ? 4 : 3;
    if (cb8_c < need || cb8_crop < 0 || cb8_w < d->w + 2 * cb8_crop) return MC_EINVAL;
    in.y8 = y_cb8; in.mean = cb8_mean; in.cw = cb8_w; in.crop = cb8_crop; in.c = cb8_c; in.c8 = (cb8_c + 7) / 8;
  } else {
    if (!u || !v || !T || (d->p_pred && !p)) return MC_EINVAL;
    in.u = u; in.v = v; in.p = d->p_pred ? p : nullptr; in.T = T;
  }
  LossGeom g;
  // (pbs / ppbs of LossGeom: strides of the plane INPUTS and of the gradient planes; they are the same tensor layout in every
  // caller of the plane form, and only the gradient strides matter for the CB8 form)
  if (!y_cb8 && (g_pbs != pbs || g_ppbs != ppbs)) return MC_EINVAL;
  g.d = *d; g.ct = (d->p_pred ? 3 : 2) + 1; g.pbs = g_pbs; g.ppbs = g_ppbs;
  dim3 grid(mc_loss_fused_blocks(d->n, d->h, d->w), d->n);
  const int tiles_x = cdiv(d->w, LF_TW), tiles = tiles_x * cdiv(d->h, LF_TH);
  hipStream_t s = (hipStream_t)stream;
  float* gpp = d->p_pred ? gp : nullptr;
#define LFK(CB, MO) hipLaunchKernelGGL((k_loss_fused<CB, MO>), grid, dim3(256), 0, s, g, in, uvp, mm, yc, paras, scaler, sums, gu, gv, gpp, gT, tiles_x, tiles, gsum_part)
  if (y_cb8) { if (mom) LFK(true, true); else LFK(true, false); }
  else { if (mom) LFK(false, true); else LFK(false, false); }
#undef LFK
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_momentum_residual(const mc_loss_desc* d, const float* u, const float* v, const float* p, const float* T,
                         int64_t pbs, int64_t ppbs, const float* yc, const float* paras, const float* scaler,
                         double* sums, float* sx, float* sy, float* eta_ws, void* stream) {
  int rc = check_loss_desc(d);
  if (rc) return rc;
  if (!u || !v || !T || !yc || !paras || !scaler || !sums || !sx || !sy || !eta_ws || d->t_grad < 0) return MC_EINVAL;
  MomGeom g{d->n, d->h, d->w, pbs, ppbs, d->inv_h, d->ra, d->lambda_mom};
  const int tiles_x = cdiv(d->w, MA_TW), tiles = tiles_x * cdiv(d->h, MR_TH);
  dim3 grid(min(2 * loss_blocks(d->h * d->w, d->n), tiles), d->n);      // two sums only: 2048 blocks measured best
  hipLaunchKernelGGL(k_mom_residual, grid, dim3(256), 0, (hipStream_t)stream, g, u, v, p, T, yc, paras, scaler, sums, sx, sy, eta_ws,
                     tiles_x, tiles);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_momentum_adjoint(const mc_loss_desc* d, const float* T, int64_t pbs, int64_t ppbs, const float* eta_ws, const float* paras,
                        const float* scaler, const float* sx, const float* sy, float* gu, float* gv, float* gp,
                        float* gT, void* stream) {
  int rc = check_loss_desc(d);
  if (rc) return rc;
  if (!T || !eta_ws || !paras || !scaler || !sx || !sy || !gu || !gv) return MC_EINVAL;
  MomGeom g{d->n, d->h, d->w, pbs, ppbs, d->inv_h, d->ra, d->lambda_mom};
  const int tiles_x = cdiv(d->w, MA_TW);
  dim3 grid(tiles_x * cdiv(d->h, MA_TH), d->n);
  hipLaunchKernelGGL(k_mom_adjoint, grid, dim3(256), 0, (hipStream_t)stream, g, eta_ws, scaler, sx, sy, gu, gv, gp, gT,
                     d->t_grad, tiles_x);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_assemble_adtime_batch(const float* T, const float* uv, const float* t, const float* paras, const float* paras_nd,
                             const float* xc, const float* yc, const int32_t* pairs, int32_t b, int32_t m, int32_t cy,
                             int32_t h, int32_t w, float* x, float* y, float* scaler, float* paras_out, void* stream) {
  if (!T || !uv || !t || !paras || !paras_nd || !xc || !yc || !pairs || !x || !y || !scaler || !paras_out) return MC_EINVAL;
  if (b <= 0 || m <= 0 || cy < 2 || h <= 0 || w <= 0) return MC_EINVAL;
  dim3 grid(max(1, min(cdiv(h * w, 256 * 4), 256)), b);
  hipLaunchKernelGGL(k_assemble_adtime, grid, dim3(256), 0, (hipStream_t)stream, T, uv, t, paras, paras_nd, xc, yc, pairs, cy,
                     h * w, x, y, scaler, paras_out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_assemble_newad_batch(const float* T, const float* uvp, const float* t, const float* paras, const float* paras_nd,
                            const float* xc, const float* yc, const int32_t* idx, int32_t b, int32_t m, int32_t cy, int32_t h,
                            int32_t w, float* x, float* y, float* t_weight, float* scaler, void* stream) {
  if (!T || !uvp || !t || !paras || !paras_nd || !xc || !yc || !idx || !x || !y || !t_weight || !scaler) return MC_EINVAL;
  if (b <= 0 || m <= 0 || cy < 2 || cy > 3 || h <= 0 || w <= 0) return MC_EINVAL;
  dim3 grid(max(1, min(cdiv(h * w, 256 * 4), 256)), b);
  hipLaunchKernelGGL(k_assemble_newad, grid, dim3(256), 0, (hipStream_t)stream, T, uvp, t, paras, paras_nd, xc, yc, idx, cy, h * w,
                     x, y, t_weight, scaler);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_ts_build_input(const float* T, const float* xc, const float* yc, const float* ycc, const float* paras,
                      const float* paras_nd, int32_t n, int32_t h, int32_t w, float* out, void* stream) {
  if (!T || !xc || !yc || !ycc || !paras || !paras_nd || !out || n <= 0 || h <= 0 || w <= 0) return MC_EINVAL;
  dim3 grid(max(1, min(cdiv(h * w, 256), 1024)), n);
  hipLaunchKernelGGL(k_ts_build_input, grid, dim3(256), 0, (hipStream_t)stream, T, xc, yc, ycc, paras, paras_nd, h * w, out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_ts_build_input_unet(const float* T, const float* xc, const float* yc, const float* ycc, const float* paras,
                           const float* paras_nd, const float* dt, const float* u_prev, const float* v_prev, int32_t n, int32_t h,
                           int32_t w, float* out, void* stream) {
  if (!T || !xc || !yc || !ycc || !paras || !paras_nd || !dt || !u_prev || !v_prev || !out || n <= 0 || h <= 0 || w <= 0)
    return MC_EINVAL;
  dim3 grid(max(1, min(cdiv(h * w, 256), 1024)), n);
  hipLaunchKernelGGL(k_ts_build_input_unet, grid, dim3(256), 0, (hipStream_t)stream, T, xc, yc, ycc, paras, paras_nd, dt, u_prev,
                     v_prev, h * w, out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_roll_forward_update(float* x, int32_t c, const float* u, const float* v, const float* t, int64_t uvt_batch_stride,
                           const float* paras, int32_t update_v, int32_t n, int32_t h, int32_t w, void* stream) {
  if (!x || !u || !v || !t || !paras || c < 10 || n <= 0 || h <= 0 || w <= 0 || uvt_batch_stride < (int64_t)h * w) return MC_EINVAL;
  dim3 grid(max(1, min(cdiv(h * w, 256), 1024)), n);
  hipLaunchKernelGGL(k_roll_forward_update, grid, dim3(256), 0, (hipStream_t)stream, x, c, u, v, t, (size_t)uvt_batch_stride, paras,
                     update_v, h * w);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_ts_wall_bc(const float* t_in, int32_t n, int32_t h, int32_t w, float* t_out, void* stream) {
  if (!t_in || !t_out || n <= 0 || h < 2 || w < 3 || t_in == t_out) return MC_EINVAL;
  hipLaunchKernelGGL(k_ts_wall_bc, dim3(n), dim3(1024), 0, (hipStream_t)stream, t_in, h, w, t_out);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_adnet_step(const float* u, const float* v, int64_t uv_stride, const float* vel_scale, const float* T_prev,
                  const float* raq_field, const float* raq_scalar, const float* xc, const float* yc, int32_t n, int32_t h,
                  int32_t w, float cn_max, int32_t compute_dt, float* dt_io, uint32_t* ws, float* T_next, void* stream) {
  if (!u || !v || !T_prev || !xc || !yc || !dt_io || !T_next || n <= 0 || h < 3 || w < 3) return MC_EINVAL;
  if (!raq_field && !raq_scalar) return MC_EINVAL;
  if (compute_dt && !ws) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  AdGeom g{n, h, w, uv_stride, cn_max};
  dim3 grid(max(1, min(cdiv(h * w, 256), 1024)), n);
  if (compute_dt) {
    hipLaunchKernelGGL(k_adnet_ws_init, dim3(1), dim3(1), 0, s, ws);
    hipLaunchKernelGGL(k_adnet_reduce, grid, dim3(256), 0, s, g, u, v, vel_scale, xc, ws);
    hipLaunchKernelGGL(k_adnet_dt, dim3(1), dim3(1), 0, s, cn_max, ws, dt_io);
  }
  hipLaunchKernelGGL(k_adnet_step, grid, dim3(256), 0, s, g, u, v, vel_scale, T_prev, raq_field, raq_scalar, xc, yc, dt_io, T_next);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

int mc_loss_finalize(const mc_loss_desc* d, const double* sums, float* out8, void* stream) {
  int rc = check_loss_desc(d);
  if (rc) return rc;
  if (!sums || !out8) return MC_EINVAL;
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, *d, sums, out8);
  MC_CHECK_LAUNCH();
  return MC_OK;
}

}  // extern "C"
