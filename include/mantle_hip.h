/*
 * mantle_hip.h — C ABI of libmantle_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the Stokes-surrogate training hot path of
 * agsiddhant/PBML_Mantle_Convection.  The reference has no FFI / operator registry: its
 * boundary is the Python surface, whose numeric back end is PyTorch ATen.
 * Every entry point below replaces the ATen call sequence of one reference call site
 * (cited per function, paths relative to the reference tree).  Host code (Python, ctypes)
 * mirrors the reference's module/trainer API on top of these.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  `stream` is a hipStream_t
 *    passed as void*.  All calls are asynchronous on `stream`, stateless and re-entrant.
 *  - The caller owns every buffer (including workspaces); the library never allocates,
 *    frees or retains device memory.
 *  - Return 0 on success; MC_E* (negative) for argument/shape/unsupported errors;
 *    positive values are hipError_t codes from the launch.  Nothing throws or aborts.
 *  - Activations use the CB8 layout: [N][C8][H][W][8] with C8 = ceil(C/8) channel blocks,
 *    element type f32 (MC_F32), bf16 (MC_BF16) or f16 forward / bf16 gradients (MC_MIX16).  Padded channels hold zeros.
 *    Network inputs/outputs and loss fields are plain NCHW / NHW f32.
 *  - Parameters are read in the REFERENCE's layout (unique mirrored-filter bank
 *    [U][C_in][k][k] f32, bias [C_out] f32; symmetric_layers_torch.py:96-107).
 */
#ifndef MANTLE_HIP_H
#define MANTLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_OK 0
#define MC_EINVAL (-1)
#define MC_EUNSUPPORTED (-2)
#define MC_EWORKSPACE (-3)

/* MC_MIX16 (the trainer's "mixed" precision): every tensor of the FORWARD pass (packed input, filter banks of the forward
 * convolutions, raw conv outputs, activations, pooled / upsampled tensors) is IEEE f16, every GRADIENT tensor (dY, dA,
 * padded-domain input gradients, their filter banks) is bf16; MFMA arithmetic with f32 accumulation either way.  An entry
 * point that takes `dtype` reads / writes each of its operands in the type of that operand's role: forward-only calls
 * (mc_pack_nchw, mc_gn_act_fwd, mc_bicubic_fwd ...) move f16, gradient-only calls (mc_fold_padded, mc_bicubic_bwd ...) move
 * bf16, the GroupNorm-backward and filter-gradient calls read y / x as f16 and gradients as bf16. */
enum { MC_F32 = 0, MC_BF16 = 1, MC_MIX16 = 2 };
enum { MC_PAD_ZEROS = 0, MC_PAD_REPLICATE = 1, MC_PAD_REFLECT = 2 };
enum { MC_ACT_NONE = 0, MC_ACT_GELU = 1, MC_ACT_RELU = 2, MC_ACT_SILU = 3, MC_ACT_TANH = 4,
       MC_ACT_SELU = 5, MC_ACT_ELU = 6 };
/* what follows the convolution inside one reference layer */
enum { MC_POST_NONE = 0,   /* plain nn.Conv2d (Unet conv[-1], ConvAE final conv)          */
       MC_POST_ACT = 1,    /* conv -> act (Unet conv[-2])                                  */
       MC_POST_GN_ACT = 2  /* conv -> GroupNorm -> act (FluidLayer; Unet conv[-3] + gn[0]) */ };
/* where the gradient w.r.t. an activated tensor comes from (backward) */
enum { MC_GSRC_NONE = 0,
       MC_GSRC_PLAIN = 1,      /* CB8 tensor of the same H x W                                         */
       MC_GSRC_PADFOLD = 2,    /* dgrad output on the padded domain (H+2p)x(W+2p) whose halo has been      */
                               /* folded onto the interior by mc_fold_padded: read at offset (p, p)    */
       MC_GSRC_PADFOLD_POOL = 3, /* same, through the adjoint of AvgPool(f): value/f^2 at (y/f, x/f)  */
       MC_GSRC_PLAIN_POOL = 4  /* CB8 tensor of size hs x ws through the adjoint of AvgPool(f)        */ };

typedef struct {
  int32_t n;          /* batch                                                       */
  int32_t h, w;       /* INPUT spatial size                                          */
  int32_t c_in0;      /* channels of source 0                                        */
  int32_t c_in1;      /* channels of source 1 (torch.cat on dim 1) or 0              */
  int32_t c_out;
  int32_t k;          /* square kernel, 3 or 5                                       */
  int32_t pad;        /* per side; output is (h + 2 pad - k + 1) x (w + 2 pad - k + 1) */
  int32_t pad_mode;   /* MC_PAD_*                                                    */
  int32_t dtype;      /* MC_F32 | MC_BF16 | MC_MIX16 : element type of x0, x1, y (MC_MIX16: f16; the input-gradient
                         convolution of such a layer is described with MC_BF16)       */
  int32_t sym_h;      /* number of x-mirrored filters (SymmetricConv2d symmetry['h']), 0 = plain Conv2d */
  int32_t c_out_split;/* dgrad only: first c_out_split output channels go to y0, the rest to y1 (0 = all to y0) */
  int32_t out_f32;    /* 16-bit dtypes only: 1 = write y0 as f32 CB8 (the network's last conv: u, v, p, T are not
                         quantised); requires c_out <= 16 and no c_out_split */
  int32_t sym_v;      /* number of y-mirrored filters (symmetry['v']) and of filters mirrored about both axes (symmetry['hv'],  */
  int32_t sym_hv;     /* quadruples: x-, y- and xy-flips): output channels [U, c_out) in the reference's torch.cat order, */
                      /* U = c_out - sym_h/2 - sym_v/2 - 3 sym_hv/4 unique filters (symmetric_layers_torch.py:118-136)   */
} mc_conv_desc;

typedef struct {
  const void* ptr;    /* CB8 tensor, element type = desc dtype                        */
  int32_t kind;       /* MC_GSRC_*                                                    */
  int32_t pad;        /* p of the conv that produced the padded-domain gradient       */
  int32_t pad_mode;   /* MC_PAD_* of that conv                                        */
  int32_t pool;       /* f for MC_GSRC_PADFOLD_POOL                                   */
  int32_t hs, ws;     /* unpadded spatial size of the tensor `ptr` is the gradient of */
  int32_t c8_total;   /* > 0: `ptr` holds c8_total channel blocks per sample and this source is   */
  int32_t cb_off;     /* the slice starting at block cb_off (gradient of one torch.cat operand)   */
} mc_grad_src;

/* "Normalise on load": a consumer of a raw conv output y applies act(scale * y + shift) — GroupNorm's affine map folded
 * with its statistics, then the activation (FluidLayer.forward, pytorch_networks_convae.py:790-799) — while it stages its
 * input tile, so the activated tensor never crosses HBM.  coef tables: [n][ceil(c/8)*8][4] f32 = (scale, shift, mean,
 * rstd) per (sample, channel) from mc_gn_finalize_coef; NULL = no affine map.  act = MC_ACT_NONE with a NULL table
 * means the source is used as it is (network input, pooled / upsampled tensors). */
typedef struct {
  const float* coef0; /* source 0 */
  const float* coef1; /* source 1 (second torch.cat operand) */
  int32_t act0, act1; /* MC_ACT_* */
} mc_conv_prologue;

/* Input-gradient epilogue: the launch computes dA (gradient w.r.t. the ACTIVATED tensor a = act(GN(y))) on the padded
 * domain; with an epilogue it stores dz = dA * act'(z), z = scale*y + shift, instead and emits the per-tile partial
 * sums (sum dz, sum dz * yhat) per channel that GroupNorm's backward needs — the separate reduction pass over (dA, y)
 * disappears.  With reflect / replicate padding the pixels within pad+1 of the border still await the padding adjoint:
 * they are stored as raw dA and finished (fold + dz + their partial sums) by mc_fold_padded_dz. */
typedef struct {
  const void* y;      /* raw conv output of the layer that produced the tensor: CB8 [n][c8][hs][ws], dtype of the desc */
  const float* coef;  /* its (scale, shift, mean, rstd) table, or NULL for an activation-only layer */
  int32_t act;        /* MC_ACT_* */
  int32_t pad;        /* p of the forward conv: the interior starts at (p, p) of the padded-domain output */
  int32_t pad_mode;   /* MC_PAD_* of the forward conv */
  int32_t hs, ws;     /* interior (unpadded) size */
  float* partials;    /* [n][part_stride][c8*8][2] f32; this launch fills slots 0 .. mc_conv_tiles(desc) - 1 of a sample */
  int32_t part_stride;/* slots per sample (>= tiles; the slots behind the tiles are mc_fold_padded_dz's) */
  int32_t y_f16;      /* 1: y is f16 -- the layer belongs to an MC_MIX16 network (the launch itself is MC_BF16); else 0 */
} mc_conv_epilogue;

int mc_version(void);
const char* mc_strerror(int code);

/* ---- boundary layout conversion ---------------------------------------------------------- */
/* NCHW f32 -> CB8 with optional W padding (Unet.forward's F.pad(inputs,(3,3,0,0),mode=r_p),
 * pytorch_networks_convae.py:1991).  out is [n][ceil(c/8)][h][w + 2 pad_w][8]. */
/* x holds src_c >= c channels per sample (only the first c are taken: get_loss feeds the net the
 * first ten of gVTp's channels, multigpu.py:234-248); chan_scale (nullable, [c]) multiplies each
 * channel on the way in (xc/4, yc/4, dt/roll_forward). */
int mc_pack_nchw(const float* x, int32_t n, int32_t c, int32_t src_c, int32_t h, int32_t w, int32_t pad_w,
                 int32_t pad_mode, const float* chan_scale, int32_t dtype, void* out, void* stream);
/* CB8 -> NCHW f32, optionally subtracting a per-(n,c) mean and cropping crop_w columns on
 * both sides ((y - mean(y))[..., 3:-3], pytorch_networks_convae.py:2024).  mean may be NULL. */
int mc_unpack_nchw(const void* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_w,
                   const float* mean_nc, int32_t dtype, float* out, void* stream);
/* adjoint of mc_unpack_nchw: g NCHW f32 [n][c][h][w - 2 crop_w] -> CB8 [n][c8][h][w][8],
 * zero in the cropped columns, minus mean_nc (the adjoint of the mean subtraction) if given. */
int mc_pack_grad_nchw(const float* g, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_w,
                      const float* mean_nc, int32_t dtype, void* out, void* stream);
/* per-(n,c) sums of an NCHW f32 tensor, scaled: out[n*c] = scale * sum_hw x. */
int mc_sum_hw(const float* x, int32_t nc, int32_t hw, float scale, float* out, void* stream);

/* ---- convolution (SymmetricConv2d.forward symmetric_layers_torch.py:113-138 +
 *      nn.Conv2d._conv_forward: F.pad(mode) + F.conv2d; FluidLayer :764-786, Unet heads :1933-1979) */
/* Bytes of the packed filter bank mc_pack_weights writes for this descriptor / direction. */
size_t mc_packed_weight_bytes(const mc_conv_desc* d, int32_t dgrad);
/* Expand the unique mirrored-filter bank (never materialised in the reference layout: the
 * x-flipped copies are generated while packing) into the kernel-side bank.  dgrad != 0 packs
 * the transposed, 180-degree-rotated bank used by mc_conv2d for the input gradient. */
int mc_pack_weights(const mc_conv_desc* d, const float* w_unique, int32_t dgrad, void* packed,
                    void* stream);
/* Symbol-style name of the kernel instantiation mc_conv2d launches for this descriptor (static string;
 * lets a profiler trace be matched to a descriptor). */
const char* mc_conv_kernel_name(const mc_conv_desc* d);
/* Number of spatial tiles per image the conv kernel uses for this descriptor (sizes stat partials). */
int32_t mc_conv_tiles(const mc_conv_desc* d);
/* y = conv(pad(cat(x0,x1))) + bias.  y0 (and y1 when c_out_split) are CB8 outputs.
 * stat_partials (nullable): [n][tiles][c_out8*8][2] f32 per-tile (sum, sum of squares) of y,
 * taken from the f32 accumulators — the first half of GroupNorm (FluidLayer :788). */
int mc_conv2d(const mc_conv_desc* d, const void* x0, const void* x1, const void* packed_w,
              const float* bias, void* y0, void* y1, float* stat_partials, void* stream);
/* The same convolution with the producer's GroupNorm + activation applied to its sources on load (pro, nullable) and,
 * for an input-gradient launch, the GroupNorm-backward reduction fused into the epilogue (epi, nullable; needs a
 * single-source, unsplit output).  FluidLayer.forward :790-799 and its autograd backward. */
int mc_conv2d_fused(const mc_conv_desc* d, const void* x0, const void* x1, const mc_conv_prologue* pro,
                    const void* packed_w, const float* bias, void* y0, void* y1, float* stat_partials,
                    const mc_conv_epilogue* epi, void* stream);
/* Filter/bias gradient.  partials: workspace of mc_wgrad_partial_bytes(d); the second call
 * reduces it deterministically, folds mirrored filters back onto the unique bank
 * (dW_unique[i] += flip_x(dW_full[U+i])) and ACCUMULATES into dw_unique / dbias. */
size_t mc_wgrad_partial_bytes(const mc_conv_desc* d);
/* One byte past the highest address a forward launch described by `d` reads through packed_w (the kernels' own indexing,
 * restated on the host).  For every layer: mc_conv_bank_read_extent(d) <= mc_packed_weight_bytes(d, 0), and for its
 * input-gradient launch dd: mc_conv_bank_read_extent(dd) <= mc_packed_weight_bytes(d, 1) (tests/test_abi_and_host.py). */
size_t mc_conv_bank_read_extent(const mc_conv_desc* d);
int mc_conv2d_wgrad(const mc_conv_desc* d, const void* x0, const void* x1, const void* dy,
                    void* partials, void* stream);
int mc_conv2d_wgrad_finalize(const mc_conv_desc* d, const void* partials, float* dw_unique,
                             float* dbias, void* stream);
/* mc_conv2d_wgrad with x0 / x1 given as RAW conv outputs: the activation is re-applied on load (pro as in mc_conv2d_fused). */
int mc_conv2d_wgrad_fused(const mc_conv_desc* d, const void* x0, const void* x1, const mc_conv_prologue* pro,
                          const void* dy, void* partials, void* stream);
/* Batched forms (one launch for many layers; per-layer tables are plain host arrays of length n):
 * the per-step bank packing and the filter-gradient combine are tiny per layer and latency-bound when
 * launched one layer at a time. */
int mc_pack_weights_batched(const mc_conv_desc* descs, const float* const* w_unique, const int32_t* dgrad,
                            void* const* packed, int32_t n, void* stream);
int mc_conv2d_wgrad_finalize_batched(const mc_conv_desc* descs, const void* const* partials, float* const* dw_unique,
                                     float* const* dbias, int32_t n, void* stream);

/* Adjoint of F.pad(mode) on a padded-domain gradient (output of mc_conv2d in input-gradient mode):
 * adds the halo of buf [n][c8][hs+2p][ws+2p][8] onto the interior positions it was padded from
 * (reflect: mirror about the edge pixel; replicate: onto the edge pixel; zeros: nothing). In place;
 * touches only the O(p (H+W)) border pixels. */
int mc_fold_padded(void* buf, int32_t n, int32_t c, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode,
                   int32_t dtype, void* stream);
/* The same for the two input-gradient outputs of a convolution over concatenated sources (same hs x ws, c0 and c1 channels)
 * in one launch; buf1 NULL = one buffer. */
int mc_fold_padded2(void* buf0, int32_t c0, void* buf1, int32_t c1, int32_t n, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode,
                    int32_t dtype, void* stream);
/* Companion of mc_conv2d_fused's epilogue for reflect / replicate padding: folds the halo onto the frame pixels (within
 * pad+1 of the border) like mc_fold_padded, then turns them into dz = dA * act'(z) in place and writes their partial sums
 * into `partials` [n][mc_fold_blocks()][c8*8][2] (the caller passes the slots behind the conv tiles' of the same table). */
int32_t mc_fold_blocks(int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode);
int mc_fold_padded_dz(void* buf, int32_t n, int32_t c, int32_t hs, int32_t ws, int32_t pad, int32_t pad_mode,
                      int32_t dtype, const void* y, const float* coef, int32_t act, float* partials,
                      int32_t part_stride_blocks, int32_t part_first_block, void* stream);

/* ---- GroupNorm + activation (FluidLayer :788-799; Unet :2016-2021) ------------------------ */
/* Per-tile (sum, sum of squares) partials [n][tiles][ceil(c/8)*8][2] of an existing CB8 tensor, for layers whose output is
 * assembled from several convolutions (BoundaryLearnedConvolution2D) and cannot take them from one conv epilogue. */
int mc_gn_partials(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t dtype, int32_t tiles, float* part,
                   void* stream);
/* (mean, rstd) per (n, group) from conv stat partials; eps 1e-5, biased variance.  Also emits
 * per-(n,c) means of y when chan_mean != NULL (Unet's spatial zero-mean, :2024). */
int mc_gn_finalize(const float* stat_partials, int32_t n, int32_t tiles, int32_t c, int32_t groups,
                   int32_t hw, float eps, float* stats_ng2, float* chan_mean, void* stream);
/* mc_gn_finalize + the (scale, shift, mean, rstd) table [n][ceil(c/8)*8][4] that consumers of the raw conv output read
 * ("normalise on load"): scale = rstd * gamma, shift = beta - mean * rstd * gamma. */
int mc_gn_finalize_coef(const float* stat_partials, int32_t n, int32_t tiles, int32_t c, int32_t groups, int32_t hw,
                        float eps, const float* gamma, const float* beta, float* stats_ng2, float* coef4, void* stream);

/* a = act(GN(y));  post = MC_POST_*.  pool > 1 additionally writes AvgPool2d(pool)(a) into
 * pooled (Unet :2002, ConvAE :1051).  y, a, pooled are CB8 of `dtype`.  a may be NULL when pool > 1: only the pooled
 * tensor is materialised (the full-resolution consumers normalise y on load). */
int mc_gn_act_fwd(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups,
                  const float* stats_ng2, const float* gamma, const float* beta, int32_t post,
                  int32_t act, int32_t pool, int32_t dtype, void* a, void* pooled, void* stream);
/* Small layers: mc_gn_finalize + mc_gn_act_fwd in one launch (one block per (sample, channel block); c / groups in
 * {1, 2, 4, 8}; pool 1 or 2): statistics from the conv's stat_partials [n][tiles][c8*8][2], written to stats_ng2 for the
 * backward pass, and a = act(GN(y)) [+ AvgPool2d(pool)]. */
int mc_gn_act_fwd_small(const void* y, const float* stat_partials, int32_t tiles, int32_t n, int32_t c, int32_t h, int32_t w,
                        int32_t groups, float eps, const float* gamma, const float* beta, int32_t act, int32_t pool,
                        int32_t dtype, float* stats_ng2, void* a, void* pooled, void* stream);
/* Backward of act(GN(y)) given the gradient sources of a.  Phase 1 reduces
 * (sum dz, sum dz*yhat) per (n,c) into partials [n][blocks][c8*8][2]; phase 2 (finalize)
 * turns them into per-(n,g) means and accumulates dgamma/dbeta (in sample order: deterministic); phase 3 writes dy. */
int32_t mc_gn_bwd_blocks(int32_t h, int32_t w);
int mc_gn_act_bwd_reduce(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups,
                         const float* stats_ng2, const float* gamma, const float* beta, int32_t post,
                         int32_t act, int32_t dtype, const mc_grad_src* g0, const mc_grad_src* g1,
                         float* partials, void* stream);
int mc_gn_act_bwd_finalize(const float* partials, int32_t n, int32_t blocks, int32_t c, int32_t groups,
                           int32_t hw, const float* gamma, float* m12_ng2, float* dgamma, float* dbeta,
                           void* stream);
/* The same for partial tables with many slots per sample (the input-gradient epilogue's): one block per (group, sample);
 * writes m12 [n][groups][2] and the per-(sample, channel) sums chan_sums [n][c8*8][2]; dgamma / dbeta follow from
 * mc_gn_param_grads_batched (samples in order: deterministic).  Channels per group: 1, 2, 4 or 8. */
int mc_gn_act_bwd_finalize_n(const float* partials, int32_t n, int32_t blocks, int32_t c, int32_t groups, int32_t hw,
                             const float* gamma, float* m12_ng2, float* chan_sums, void* stream);
int mc_gn_act_bwd_apply(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups,
                        const float* stats_ng2, const float* m12_ng2, const float* gamma,
                        const float* beta, int32_t post, int32_t act, int32_t dtype,
                        const mc_grad_src* g0, const mc_grad_src* g1, void* dy, void* stream);
/* Small layers: the three phases in one launch (one block per (sample, channel block); needs c / groups in {1, 2, 4, 8}).
 * Writes dy and the per-(sample, channel) sums chan_sums [n][ceil(c/8)*8][2] = (sum dz, sum dz * yhat); dgamma / dbeta are
 * accumulated from those tables, samples in order, by mc_gn_param_grads_batched (one launch for many layers: per-layer
 * tables are plain host arrays of length `jobs`). */
int mc_gn_act_bwd_small(const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups, const float* stats_ng2,
                        const float* gamma, const float* beta, int32_t act, int32_t dtype, const mc_grad_src* g0,
                        const mc_grad_src* g1, void* dy, float* chan_sums, void* stream);
int mc_gn_param_grads_batched(const float* const* chan_sums, const int32_t* n, const int32_t* c, float* const* dgamma,
                              float* const* dbeta, int32_t jobs, void* stream);
/* Phase 3 for a tensor whose dz = dA * act'(z) was already written by mc_conv2d_fused's epilogue (+ mc_fold_padded_dz):
 * dy = scale * dz - rstd (m1 + yhat m2)  (coef = the layer's table, m12 from mc_gn_act_bwd_finalize); with coef == NULL
 * (activation-only layer) dy = dz.  dz is read through a gradient source (MC_GSRC_PADFOLD or MC_GSRC_PLAIN). */
int mc_gn_bwd_apply_dz(const mc_grad_src* dz, const void* y, int32_t n, int32_t c, int32_t h, int32_t w, int32_t groups,
                       const float* coef, const float* m12_ng2, int32_t dtype, void* dy, void* stream);
/* ---- torch.cat of more than two operands along channels (NewFluidNet.forward, pytorch_networks_convae.py:1327-1332):
 * out [n][sum of blocks][h][w][8]; every operand but the last must have a multiple of 8 channels. */
int mc_concat_cb8(const void* const* srcs, const int32_t* src_c, int32_t n_src, int32_t n, int32_t h, int32_t w,
                  int32_t dtype, void* out, void* stream);
/* out (plain CB8 [n][c/8][h][w][8]) = g0 + g1: materialises the gradient of a tensor with two consumers
 * (the pooled feature maps of NewFluidNet feed a conv AND the next pooling level). */
int mc_gsrc_sum(const mc_grad_src* g0, const mc_grad_src* g1, int32_t n, int32_t c, int32_t h, int32_t w,
                int32_t dtype, void* out, void* stream);

/* Rectangle copy between CB8 tensors of the same channel count: dst[n][cb][dy+r][dx+c] (=|+=) src[n][cb][sy+r][sx+c] for
 * r < rh, c < rw; src == NULL writes zeros.  Used to cut the border strips of BoundaryLearnedConvolution2D out of its input
 * and to frame its output (pytorch_networks_convae.py:1022-1065). */
int mc_rect_copy(const void* src, int32_t hs, int32_t ws, int32_t sy, int32_t sx, void* dst, int32_t hd, int32_t wd,
                 int32_t dy, int32_t dx, int32_t rh, int32_t rw, int32_t n, int32_t c, int32_t accumulate, int32_t dtype,
                 void* stream);

/* ---- resampling (nn.AvgPool2d, nn.Upsample(mode='bicubic'); Unet :2002,2009,2014; ConvAE :1051,1079) */
int mc_avgpool_fwd(const void* x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t f, int32_t dtype,
                   void* out, void* stream);
/* taps: idx [out][4] int32 (clamped), wgt [out][4] f32, per axis (host-built in f64, A = -0.75). */
int mc_bicubic_fwd(const void* x, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                   const int32_t* idx_y, const float* wgt_y, const int32_t* idx_x, const float* wgt_x,
                   int32_t dtype, void* out, void* stream);
/* mc_bicubic_fwd of act(scale * x + shift): x is a raw conv output, activated while the input window is staged
 * (coef nullable = activation only). */
int mc_bicubic_fwd_act(const void* x, const float* coef, int32_t act, int32_t n, int32_t c, int32_t hi, int32_t wi,
                       int32_t ho, int32_t wo, const int32_t* idx_y, const float* wgt_y, const int32_t* idx_x,
                       const float* wgt_x, int32_t dtype, void* out, void* stream);
/* adjoint: transposed tap tables in CSR form per axis (start [in+1], j [], w []). */
int mc_bicubic_bwd(const mc_grad_src* g, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                   const int32_t* ty_start, const int32_t* ty_j, const float* ty_w,
                   const int32_t* tx_start, const int32_t* tx_j, const float* tx_w,
                   int32_t dtype, void* dx, void* stream);
/* The same with the longest tap list of each table given (host knowledge of the tables; 0 = unknown): lists of <= 8 entries
 * (every even x2 upsample) run a kernel instantiation that holds 8 instead of 12 taps per pixel on chip. */
int mc_bicubic_bwd_taps(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                        const int32_t* tys, const int32_t* tyj, const float* tyw, const int32_t* txs, const int32_t* txj,
                        const float* txw, int32_t max_taps_y, int32_t max_taps_x, int32_t dtype, void* dx, void* stream);
/* The same adjoint as a walk down the output rows (each read once into registers; a sliding register window of the four
 * open input rows; one LDS crossing per finished row for the x taps): takes the FORWARD y tables (idx_y, wgt_y: [ho][4], as
 * mc_bicubic_fwd) plus the transposed lists (ty_start / ty_j for the chunk extents, the x lists for the gather).  Any ratio
 * ho / hi >= 1; x tap lists of <= 12 entries (max_taps_x: the longest, host knowledge of the table), else MC_EUNSUPPORTED. */
int mc_bicubic_bwd_walk(const mc_grad_src* gs, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                        const int32_t* idx_y, const float* wgt_y, const int32_t* tys, const int32_t* tyj, const int32_t* txs,
                        const int32_t* txj, const float* txw, int32_t max_taps_x, int32_t dtype, void* dx, void* stream);
/* The same adjoint as two 1-D passes through an f32 workspace [n][ceil(c/8)][hi][wo][8]: for large scale factors
 * (NewFluidNet upsamples x4 ... x16 to the full grid, pytorch_networks_convae.py:1239-1244), where a pixel's tap lists
 * are too long for the tiled kernel. */
int mc_bicubic_bwd_separable(const mc_grad_src* g, int32_t n, int32_t c, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                             const int32_t* ty_start, const int32_t* ty_j, const float* ty_w,
                             const int32_t* tx_start, const int32_t* tx_j, const float* tx_w,
                             int32_t dtype, float* ws, void* dx, void* stream);

/* ---- curl head (Unet :2038-2070): a*a_bound -> u = da/dy, v = -da/dx, antisymmetric walls;
 *      optional T = clip(t_in, t_lo, t_hi) (:2040).  Planes are [h][w] f32 with a batch stride. --- */
int mc_curl_head_fwd(const float* a, const float* t_in, int32_t n, int32_t h, int32_t w, int64_t in_batch_stride,
                     float a_bound, float t_lo, float t_hi, float* u, float* v, float* t_out, void* stream);
/* ws: workspace of 2*n*(h-2)*(w-2) floats.  ga / gt_in: gradients w.r.t. a / t_in (same batch stride). */
int mc_curl_head_bwd(const float* gu, const float* gv, const float* gt_out, const float* t_in, int32_t n, int32_t h,
                     int32_t w, float a_bound, float t_lo, float t_hi, float* ga, float* gt_in,
                     int64_t g_batch_stride, int64_t in_batch_stride, float* ws, void* stream);

/* ---- loss (Trainer.loss_fn multigpu.py:122-134; get_loss :250-305) + build-defined momentum -- */
#define MC_LOSS_SLOTS 16
enum { MC_S_U_SCALED = 0, MC_S_U_PLAIN, MC_S_V_SCALED, MC_S_V_PLAIN, MC_S_P_PLAIN, MC_S_T_PLAIN,
       MC_S_DU, MC_S_DV, MC_S_MASS, MC_S_MASS_X0, MC_S_MASS_X1, MC_S_MASS_Y0, MC_S_MASS_Y1,
       MC_S_MOMX, MC_S_MOMY, MC_S_SPARE };
typedef struct {
  int32_t n, h, w;
  int32_t p_pred;          /* uvp = (u,v,p,T) if p_pred else (u,v,T)                      */
  int32_t loss_type;       /* 0 mae, 1 mass, 2 curl                                        */
  int32_t loss_scale;      /* Trainer.loss_scale                                           */
  int32_t loss_derivative; /* Trainer.loss_derivative                                      */
  int32_t l2;              /* 0: L1 (reference); 1: squared error for the data terms       */
  float lambda_mom;        /* weight of the momentum residual term (0 = off)               */
  float inv_h;             /* 1/h of the momentum residual (126).  NOT used by the derivative */
                           /* terms: their x126 is the reference's literal (multigpu.py:163-166) */
  float ra;                /* Rayleigh number in the buoyancy term (1)                     */
  int32_t t_grad;          /* 1: T is a network output (gets a gradient); 0: T is given;   */
                           /* -1: the network has no temperature output (FluidNet family,  */
                           /* multigpu.py:138-195): T / gT may be NULL, uvp is (u, v[, p]) */
} mc_loss_desc;
/* per-sample (min, max) over (H,W) of the truth channels u, v and (when ct >= 3) channel 2: mm [n][3][2] */
int mc_loss_minmax(const float* uvp, int32_t n, int32_t ct, int32_t h, int32_t w, float* mm, void* stream);
/* Fused forward + backward of the data / derivative / divergence terms.  u,v,p,T: predictions
 * [n][h][w] f32 with the given batch strides (p may be NULL); sums: MC_LOSS_SLOTS doubles
 * (zeroed by the caller); gu..gT: d(loss)/d(pred), same strides as the predictions (p and gp
 * use p_batch_stride: in curl mode p is a channel of the network output while u, v, T come
 * from the curl head) (overwritten).  The momentum term is added by mc_momentum_* below. */
int mc_loss_fwd_bwd(const mc_loss_desc* d, const float* u, const float* v, const float* p, const float* T,
                    int64_t pred_batch_stride, int64_t p_batch_stride, const float* uvp, const float* mm,
                    double* sums, float* gu, float* gv, float* gp, float* gT, void* stream);
/* mc_loss_fwd_bwd + mc_momentum_residual + mc_momentum_adjoint in ONE launch (the residual signs and the viscosity stay in
 * LDS), for the Unet branch without the curl head (loss_type 0 / 1, t_grad >= 0; else MC_EUNSUPPORTED).  The predictions come
 * either as planes (u, v, p, T with batch strides, as mc_loss_fwd_bwd; y_cb8 NULL) or straight from the last convolution's
 * f32 output y_cb8 [n][ceil(cb8_c / 8)][h][cb8_w][8]: columns cb8_crop .. cb8_crop + w, minus cb8_mean [n][cb8_c] (nullable),
 * channels u, v, T, p = 0, 1, 2, 3 (Unet.forward :2026-2036) -- which saves the NCHW copy of the network output.  yc [h][w],
 * paras [n][3], scaler [n] are read when lambda_mom != 0.  Gradients: planes with batch strides g_pbs / g_ppbs
 * (overwritten). */
int mc_loss_fused(const mc_loss_desc* d, const float* u, const float* v, const float* p, const float* T, int64_t pbs, int64_t ppbs,
                  const float* y_cb8, int32_t cb8_w, int32_t cb8_crop, const float* cb8_mean, int32_t cb8_c, const float* uvp,
                  const float* mm, const float* yc, const float* paras, const float* scaler, double* sums, float* gu, float* gv,
                  float* gp, float* gT, int64_t g_pbs, int64_t g_ppbs, float* gsum_part, void* stream);
/* gsum_part (nullable): [n][mc_loss_fused_blocks(n, h, w)][4] per-block sums of the gradient planes (u, v, T, p; deterministic
 * order); mc_partial_sums_finalize turns them into out[n * c + j] = scale * (sum over the blocks), j < c <= 4 -- the spatial
 * means the adjoint of the network's mean subtraction needs (mc_pack_grad_nchw), without a pass over the gradient tensor. */
int32_t mc_loss_fused_blocks(int32_t n, int32_t h, int32_t w);
int mc_partial_sums_finalize(const float* part, int32_t n, int32_t blocks, int32_t c, float scale, float* out, void* stream);
/* Stokes momentum residual (build-defined).  yc [h][w], paras [n][3] =
 * (RaQ, FKT, FKP), scaler [n].  sx, sy, eta_ws: workspaces [n][h][w] f32 (eta_ws receives the viscosity
 * field computed by mc_momentum_residual and is read again by mc_momentum_adjoint). */
int mc_momentum_residual(const mc_loss_desc* d, const float* u, const float* v, const float* p,
                         const float* T, int64_t pred_batch_stride, int64_t p_batch_stride, const float* yc, const float* paras,
                         const float* scaler, double* sums, float* sx, float* sy, float* eta_ws, void* stream);
int mc_momentum_adjoint(const mc_loss_desc* d, const float* T, int64_t pred_batch_stride, int64_t p_batch_stride, const float* eta_ws,
                        const float* paras, const float* scaler, const float* sx, const float* sy,
                        float* gu, float* gv, float* gp, float* gT, void* stream);
/* loss6 (+momentum) from the sums, exactly as get_loss combines them: out[8] f32 =
 * (loss, loss_true_u, loss_true_v, loss_p, loss_T, mean mass, mom, 0). */
int mc_loss_finalize(const mc_loss_desc* d, const double* sums, float* out8, void* stream);

/* ---- on-device batch assembly (ADTimeDataset.__getitem__, datasetio.py:229-280) -----------------
 * The whole dataset stays resident in HBM: T [m][h][w], uv [m][cy][h][w] (cy >= 2: u, v[, p]), t [m], paras [m][3] =
 * (RaQ, FKT, FKP), paras_nd [m][3], xc / yc [h][w], all f32.  For every (i0, i1) of pairs [b][2] writes
 *   x [b][10][h][w] = (xc, yc, t[i1]-t[i0], paras_nd[i0] x3, log10(clip(eta,1e-8,1))/8, T[i0], u[i0]/s, v[i0]/s)
 *   y [b][3][h][w]  = (u[i1]/s, v[i1]/s, T[i1]),  scaler [b] = s,  paras_out [b][3] = paras[i0]
 * with s = 5 exp(1.80167667 RaQ/10 + 0.4330392 ln FKT - 0.46052953 ln FKP) (scaler.py:6-13) and
 * eta = exp(-ln(FKT) T + ln(FKP) (1 - yc)) (pytorch_networks_convae.py:86-102). */
int mc_assemble_adtime_batch(const float* T, const float* uv, const float* t, const float* paras, const float* paras_nd,
                             const float* xc, const float* yc, const int32_t* pairs, int32_t b, int32_t m, int32_t cy,
                             int32_t h, int32_t w, float* x, float* y, float* scaler, float* paras_out, void* stream);

/* The FluidNet family's dataset (NewADDataset.__getitem__, datasetio.py:595-654), same residency: T [m][h][w], uvp
 * [m][cy][h][w] (cy = 2 or 3: u, v[, p]), t [m] (the item's time weight).  For every item idx[b] writes
 *   x [b][7][h][w] = (xc/4, yc/4, log10(clip(eta,1e-8,1))/8, paras_nd x3, T),  y [b][cy][h][w] = (u/s, v/s[, p]),
 *   t_weight [b] = t[idx], scaler [b] = s.  (The reference's optional 1e-5 input noise is host-side and not reproduced.) */
int mc_assemble_newad_batch(const float* T, const float* uvp, const float* t, const float* paras, const float* paras_nd,
                            const float* xc, const float* yc, const int32_t* idx, int32_t b, int32_t m, int32_t cy,
                            int32_t h, int32_t w, float* x, float* y, float* t_weight, float* scaler, void* stream);

/* ---- Trainer.get_loss with roll_forward = R > 1 (multigpu.py:207-248): the network is applied R * R times in a chain, each
 * input rebuilt from channels 0..5 of the batch and the previous evaluation's T, u, v.  x [n][c][h][w] f32 (c >= 10) holds the
 * batch's RAW channels (the input-pack kernel applies xc / 4, yc / 4, dt / R); this call overwrites channels 7, 8, 9 with t, u,
 * v ([h][w] planes with a batch stride: the curl head's outputs or channels of the network output) and, when update_v != 0
 * (after a pre-step; the reference leaves V alone after a round's last step, :246-247), channel 6 with
 * log10(clip(exp(-ln(FKT) t + ln(FKP) (1 - yc)), 1e-8, 1)) / 8, yc = channel 1 of x; paras [n][3] = (RaQ, FKT, FKP). */
int mc_roll_forward_update(float* x, int32_t c, const float* u, const float* v, const float* t, int64_t uvt_batch_stride,
                           const float* paras, int32_t update_v, int32_t n, int32_t h, int32_t w, void* stream);

/* ---- inference rollout (TS.forward / ADNet.forward, pytorch_networks_convae.py:266-568) ----------
 * Input builder of the 'newfluidnet' branch (:372-395): out [n][7][h][w] = (xc/4, yc/4, log10(clip(eta,1e-8,1))/8, nd0, nd1,
 * nd2, T) with eta = exp(-ln(FKT) T + ln(FKP) (1 - ycc)); T [n][h][w], xc/yc/ycc [h][w], paras [n][3] = (RaQ, FKT, FKP),
 * paras_nd [n][3]. */
int mc_ts_build_input(const float* T, const float* xc, const float* yc, const float* ycc, const float* paras,
                      const float* paras_nd, int32_t n, int32_t h, int32_t w, float* out, void* stream);
/* Input builder of the 'unet' branch (:411-436): out [n][10][h][w] = (xc/4, yc/4, dt, nd0, nd1, nd2,
 * log10(clip(eta,1e-8,1))/8, T, u_prev, v_prev); dt, u_prev, v_prev [n][h][w].  The network predicts the next T itself;
 * mc_ts_wall_bc applies the wall rows (bottom 1, top 0) and copies the side columns from their inner neighbours
 * (:441-444) out of place. */
int mc_ts_build_input_unet(const float* T, const float* xc, const float* yc, const float* ycc, const float* paras,
                           const float* paras_nd, const float* dt, const float* u_prev, const float* v_prev, int32_t n,
                           int32_t h, int32_t w, float* out, void* stream);
int mc_ts_wall_bc(const float* t_in, int32_t n, int32_t h, int32_t w, float* t_out, void* stream);
/* One explicit upwind advection-diffusion step (ADNet.forward :522-568) on the non-uniform grid xc / yc [h][w] (wall
 * coordinates 0 / 4 and 0 / 1 are substituted on the fly, as the reference writes them into its inputs): u, v [n][h][w]
 * with batch stride uv_stride, multiplied by vel_scale[n] (NULL = 1: TS un-scales the network's velocities, :397-398);
 * raq_field [n][h][w] or NULL with raq_scalar[n]; dt_io: device scalar; compute_dt != 0 -> the CFL step
 * min(0.5 CN_max dx_min / max|u,v|, dx_min^2 / 4) is computed into it first (ws: 2 x uint32 scratch), else it is read.
 * T_next [n][h][w]: interior updated, replicate-padded, first row 1, last row 0 (side columns = their neighbours, which
 * is also TS's side condition :466-469). */
int mc_adnet_step(const float* u, const float* v, int64_t uv_stride, const float* vel_scale, const float* T_prev,
                  const float* raq_field, const float* raq_scalar, const float* xc, const float* yc, int32_t n, int32_t h,
                  int32_t w, float cn_max, int32_t compute_dt, float* dt_io, uint32_t* ws, float* T_next, void* stream);

/* ---- optimizer (torch.optim.Adam, multigpu.py:761-763) --------------------------------------- */
/* One fused multi-tensor step over flat f32 buffers; grad_scale folds the 1/world_size of the
 * data-parallel average; lr is read from device memory so a captured graph can be replayed
 * while MultiStepLR changes it. step_count (device int32) is incremented by the kernel. */
int mc_adam_step_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                      const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      float grad_scale, int32_t* step_count_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MANTLE_HIP_H */
