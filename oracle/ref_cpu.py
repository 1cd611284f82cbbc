"""CPU oracle for the Stokes-surrogate training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pbml_mantle_convection_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / the timed CPU baseline.

It restates, as plain functions over a ``state_dict`` (reference key names), the
algorithm of these reference files (paths relative to the reference tree):

  symmetric_layers_torch.py:113-138   mirrored-filter bank expansion
  pytorch_networks_convae.py:27-52    get_mass
  pytorch_networks_convae.py:55-83    pad_grad
  pytorch_networks_convae.py:86-102   eta_torch
  pytorch_networks_convae.py:145-178  pad_uvp
  pytorch_networks_convae.py:183-260  fixed finite-difference kernels
  pytorch_networks_convae.py:702-799  FluidLayer
  pytorch_networks_convae.py:1765-2070 Unet
  .ipynb_checkpoints/pycold-checkpoint.py:989-1115 ConvAE
  multigpu.py:122-134                 Trainer.loss_fn
  multigpu.py:196-305                 Trainer.get_loss (unet branch)
  scaler.py:4-71                      scale_var / unscale_var

The arithmetic itself lives in PyTorch ATen (un-pinned by the reference; torch
2.10.0 CPU here), which is present on every box this runs on, so the oracle calls
the same ATen CPU ops the reference calls.  Parity is PINNED: tests/test_oracle_golden.py
checks every function below against fixtures in tests/golden/ that
tools/make_golden.py captured from the imported reference in the build container.

The Stokes momentum residual (``momentum_residual``) has NO reference
implementation; it is build-defined, composed from the
reference's face-difference stencils and viscosity law, and is pinned by
manufactured solutions / the constant-viscosity identity in tests (parity
unpinned w.r.t. the reference, by construction).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# activations (FluidLayer / Unet ctor: pytorch_networks_convae.py:737-751, 1806-1820)
# --------------------------------------------------------------------------------------
_ACTS = {
    "gelu": lambda t: F.gelu(t),  # exact erf form (nn.GELU() default)
    "relu": F.relu,
    "selu": F.selu,
    "tanh": torch.tanh,
    "elu": F.elu,
    "silu": F.silu,
}


def activation(name: str, t: Tensor) -> Tensor:
    return _ACTS[name](t)


# The device's bf16 mode evaluates GELU / GELU' with odd minimax polynomials on |z| <= 4 (csrc/common.h
# gelu_cdf_poly / gelu_grad_poly; the fp32 mode uses erff/expf).  Restated here so that the storage-rounding
# emulation below can reproduce the device's rounding decisions, and so that a CPU test pins their accuracy.
GELU_CDF_POLY = (3.989080743e-01, -6.630133159e-02, 9.743199580e-03, -1.069904774e-03, 8.376238344e-05, -4.337137479e-06, 1.308749780e-07, -1.722451212e-09)
GELU_GRAD_POLY = (7.976932610e-01, -2.647867851e-01, 5.822368255e-02, -8.560219625e-03, 8.634741876e-04,
                  -5.865809327e-05, 2.541382519e-06, -6.288852653e-08, 6.715542261e-10)


def _odd_poly(t: Tensor, coef) -> Tensor:
    zc = t.clamp(-4.0, 4.0)
    w = zc * zc
    p = torch.full_like(t, coef[-1])
    for ck in coef[-2::-1]:
        p = p * w + ck
    return zc * p + 0.5


def gelu_bf16_mode(t: Tensor) -> Tensor:
    return t * _odd_poly(t, GELU_CDF_POLY)      # (the device does not clamp Phi to [0, 1]: it stays within the fit error)


def gelu_grad_bf16_mode(t: Tensor) -> Tensor:
    return _odd_poly(t, GELU_GRAD_POLY)


# "mixed" mode, input-gradient epilogue of the wide layers (csrc/conv_rr_bf16.hip, epilogue_dz, packed-f16 form): GELU'(z) =
# 1/2 + z Q4(z^2) on |z| <= 3 (clamped), every operation rounded to f16 (v_pk_fma_f16 rounds once per FMA).
GELU_GRAD_POLY_F16 = (7.8907706e-01, -2.4320666e-01, 4.2797559e-02, -3.8080227e-03, 1.3422040e-04)


def gelu_grad_f16_mode(t: Tensor) -> Tensor:
    """The device's packed-f16 evaluation, emulated: inputs, coefficients and every FMA result rounded to f16."""
    h = torch.float16
    r16 = lambda v: v.to(torch.float32).to(h).to(torch.float64)  # noqa: E731
    zc = r16(t.to(torch.float64)).clamp(-3.0, 3.0)
    w = r16(zc * zc)
    c = [float(torch.tensor(ck, dtype=h)) for ck in GELU_GRAD_POLY_F16]
    q = torch.full_like(zc, c[-1])
    for ck in c[-2::-1]:
        q = r16(q * w + ck)
    return r16(zc * q + 0.5).to(t.dtype)


def torch_pad_mode(r_p: str) -> str:
    """'zeros' is spelled 'constant' for F.pad (pytorch_networks_convae.py:732-735)."""
    return "constant" if r_p == "zeros" else r_p


# --------------------------------------------------------------------------------------
# symmetric filter bank (symmetric_layers_torch.py:113-138)
# --------------------------------------------------------------------------------------
def symmetry_counts(c_o: int) -> Dict[str, int]:
    """FluidLayer's choice (pytorch_networks_convae.py:755-757): only 'h' is non-zero."""
    return {"h": int(c_o / 4) if c_o > 4 else int(c_o / 2), "v": 0, "hv": 0}


def unique_filters(c_o: int, sym: Dict[str, int]) -> int:
    """symmetric_layers_torch.py:88-93"""
    return c_o - sym.get("h", 0) // 2 - sym.get("v", 0) // 2 - 3 * sym.get("hv", 0) // 4


def expand_symmetric_weight(w_unique: Tensor, sym: Dict[str, int]) -> Tensor:
    """[U, C_in, k, k] unique filters -> [C_out, C_in, k, k] full bank.

    Order: unique | x-flip of the first h/2 | y-flip of the next v/2 | (x-flip, y-flip,
    xy-flip) of the next hv/4 (symmetric_layers_torch.py:118-136).
    """
    h, v, hv = sym.get("h", 0), sym.get("v", 0), sym.get("hv", 0)
    parts = [w_unique]
    at = 0
    if h > 0:
        parts.append(w_unique[at:at + h // 2].flip(3))
        at += h // 2
    if v > 0:
        parts.append(w_unique[at:at + v // 2].flip(2))
        at += v // 2
    if hv > 0:
        n = hv // 4
        blk = w_unique[at:at + n]
        parts += [blk.flip(3), blk.flip(2), blk.flip(2, 3)]
    return torch.cat(parts, dim=0)


def conv2d_same(x: Tensor, w: Tensor, b: Optional[Tensor], r_p: str, padding=None) -> Tensor:
    """nn.Conv2d(padding='same' or explicit, padding_mode=r_p)._conv_forward: explicit
    F.pad in the padding mode, then a valid cross-correlation."""
    kh, kw = w.shape[-2:]
    if padding is None:
        ph, pw = kh // 2, kw // 2
    else:
        ph, pw = padding
    mode = torch_pad_mode(r_p)
    if mode == "constant":
        xp = F.pad(x, (pw, pw, ph, ph))
    else:
        xp = F.pad(x, (pw, pw, ph, ph), mode=mode)
    return F.conv2d(xp, w, b)


# --------------------------------------------------------------------------------------
# BoundaryLearnedConvolution2D ("learned" padding, pytorch_networks_convae.py:802-1065) — SURVEY.md §8(f) row N4
# --------------------------------------------------------------------------------------
LEARNED_BANKS = ("conv", "conv_top_left", "conv_top_right", "conv_bottom_left", "conv_bottom_right", "conv_top",
                 "conv_bottom", "conv_left", "conv_right")


def boundary_learned_conv(sd: Dict[str, Tensor], prefix: str, x: Tensor, k: int, use_symm: bool, bc_x: int = 1,
                          bc_y: int = 1) -> Tensor:
    """Nine bias-free VALID convolutions: the main bank on the whole input, eight more on the border strips of width
    pad = k + 1 + (bc - 1) (k = 5) or k + (bc - 1) (:1025-1026), framed around the main result, plus one shared bias.
    As in the reference the strip cut from the LAST rows lands in the FIRST output rows and vice versa (:1057-1060)."""
    c_o = sd[prefix + "learnable_bias"].shape[1]
    sym = symmetry_counts(c_o)

    def conv(name, t):
        w = sd[prefix + name + ".weight"]
        if use_symm:
            w = expand_symmetric_weight(w, sym)
        return F.conv2d(t, w)

    pad_x = k + 1 + (bc_x - 1) if k == 5 else k + (bc_x - 1)
    pad_y = k + 1 + (bc_y - 1) if k == 5 else k + (bc_y - 1)
    top_left = conv("conv_top_left", x[:, :, :pad_y, :pad_x])
    bottom_left = conv("conv_bottom_left", x[:, :, -pad_y:, :pad_x])
    top_right = conv("conv_top_right", x[:, :, :pad_y, -pad_x:])
    bottom_right = conv("conv_bottom_right", x[:, :, -pad_y:, -pad_x:])
    top = conv("conv_top", x[:, :, :pad_y, :])
    left = conv("conv_left", x[:, :, :, :pad_x])
    bottom = conv("conv_bottom", x[:, :, -pad_y:, :])
    right = conv("conv_right", x[:, :, :, -pad_x:])
    y = torch.cat([left, conv("conv", x), right], dim=3)
    top = torch.cat([top_left, top, top_right], dim=3)
    bottom = torch.cat([bottom_left, bottom, bottom_right], dim=3)
    y = torch.cat([bottom, y, top], dim=2)
    return y + sd[prefix + "learnable_bias"]


# --------------------------------------------------------------------------------------
# FluidLayer (pytorch_networks_convae.py:702-799)
# --------------------------------------------------------------------------------------
def gn_groups(c_o: int) -> int:
    return int(c_o / min(4, c_o))  # :788


def fluid_layer(sd: Dict[str, Tensor], prefix: str, x: Tensor, act: str, r_p: str,
                use_symm: bool) -> Tensor:
    w = sd[prefix + "layers.0.weight"]
    b = sd[prefix + "layers.0.bias"]
    c_o = b.shape[0]
    if use_symm:
        w = expand_symmetric_weight(w, symmetry_counts(c_o))
    y = conv2d_same(x, w, b, r_p)
    y = F.group_norm(y, gn_groups(c_o), sd[prefix + "layers.1.weight"], sd[prefix + "layers.1.bias"], 1e-5)
    return activation(act, y)


# --------------------------------------------------------------------------------------
# Unet (pytorch_networks_convae.py:1765-2070)
# --------------------------------------------------------------------------------------
def unet_layer_table(levels: int, c_i: int, c_h: int, c_o: int, repeats: int) -> list:
    """(state-dict prefix, C_in, C_out, kind) for every conv, in ctor order (:1842-1983)."""
    t = []
    for r in range(repeats):
        t.append((f"conv.{r}.", c_i if r == 0 else c_h, c_h, "fluid"))
    c = c_h
    for l in range(1, levels):
        for r in range(repeats):
            cin = int(c / 2) if (r == 0 and l > 1) else c
            t.append((f"convs.{l - 1}.{r}.", cin, c, "fluid"))
        c *= 2
    c = int(c / 2)
    for li, l in enumerate(range(levels - 2, 0, -1)):
        for r in range(repeats):
            cin = c + int(c / 2) if r == 0 else int(c / 2)
            t.append((f"upconvs.{li}.{r}.", cin, int(c / 2), "fluid"))
        c = int(c / 2)
    t.append((f"conv.{repeats}.", 2 * c, c, "head_gn"))
    t.append((f"conv.{repeats + 1}.", c, c, "head_act"))
    t.append((f"conv.{repeats + 2}.", c, c_o, "head_out"))
    return t


def unet_features(sd, x, levels, repeats, act, r_p, use_symm) -> Tensor:
    """Everything up to and including the mean-subtract + crop (:1985-2024)."""
    mode = torch_pad_mode(r_p)
    x = F.pad(x, (3, 3, 0, 0)) if mode == "constant" else F.pad(x, (3, 3, 0, 0), mode=mode)
    feat = {0: x}
    for r in range(repeats):
        feat[0] = fluid_layer(sd, f"conv.{r}.", feat[0], act, r_p, use_symm)
    sizes = {0: feat[0].shape[-2:]}
    for l in range(1, levels):
        cur = F.avg_pool2d(feat[l - 1], 2, 2)
        sizes[l] = cur.shape[-2:]
        for r in range(repeats):
            cur = fluid_layer(sd, f"convs.{l - 1}.{r}.", cur, act, r_p, use_symm)
        feat[l] = cur
    xu = feat[levels - 1]
    for li, l in enumerate(range(levels - 2, 0, -1)):
        xu = F.interpolate(xu, size=tuple(sizes[l]), mode="bicubic")
        xu = torch.cat((feat[l], xu), dim=1)
        for r in range(repeats):
            xu = fluid_layer(sd, f"upconvs.{li}.{r}.", xu, act, r_p, use_symm)
    xu = F.interpolate(xu, size=tuple(sizes[0]), mode="bicubic")
    y = torch.cat((xu, feat[0]), dim=1)
    R = repeats
    y = conv2d_same(y, sd[f"conv.{R}.weight"], sd[f"conv.{R}.bias"], r_p)
    c = y.shape[1]
    y = F.group_norm(y, int(c / 4), sd["gn.0.weight"], sd["gn.0.bias"], 1e-5)
    y = activation(act, y)
    y = activation(act, conv2d_same(y, sd[f"conv.{R + 1}.weight"], sd[f"conv.{R + 1}.bias"], r_p))
    y = conv2d_same(y, sd[f"conv.{R + 2}.weight"], sd[f"conv.{R + 2}.bias"], r_p)
    return (y - y.mean(dim=(2, 3), keepdim=True))[..., 3:-3]


def unet_features_quantised(sd, x, levels, repeats, act, r_p, use_symm, q, device_gelu=True) -> Tensor:
    """unet_features with the storage rounding of the device's bf16 mode emulated: `q` (e.g. a bf16
    round-trip) is applied wherever the engine stores a tensor — packed input, filter banks, raw conv
    outputs (GroupNorm statistics are taken BEFORE that rounding, from the f32 accumulators), activated
    outputs, pooled and upsampled tensors; the last conv's output stays f32; GELU is the bf16 mode's
    polynomial unless device_gelu=False.  Forward only (tests)."""
    mode = torch_pad_mode(r_p)
    act_fn = gelu_bf16_mode if (device_gelu and act == "gelu") else (lambda t: activation(act, t))

    def conv(xin, w, b):
        return conv2d_same(xin, q(w), b, r_p)

    def layer(prefix, xin, gn_prefix=None, post="gn_act"):
        w = sd[prefix + "weight"]
        b = sd[prefix + "bias"]
        c_o = b.shape[0]
        if use_symm and prefix.endswith("layers.0."):
            w = expand_symmetric_weight(w, symmetry_counts(c_o))
        y = conv(xin, w, b)
        if post == "none":
            return y, y
        yq = q(y)
        if post == "gn_act":
            G = gn_groups(c_o) if gn_prefix.endswith("layers.1.") else int(c_o / 4)
            B = y.shape[0]
            yg = y.reshape(B, G, -1)
            mean = yg.mean(-1, keepdim=True)
            var = yg.var(-1, unbiased=False, keepdim=True)
            z = ((yq.reshape(B, G, -1) - mean) / torch.sqrt(var + 1e-5)).reshape(y.shape)
            z = z * sd[gn_prefix + "weight"].view(1, -1, 1, 1) + sd[gn_prefix + "bias"].view(1, -1, 1, 1)
        else:
            z = yq
        a = act_fn(z)
        return q(a), a

    x = F.pad(x, (3, 3, 0, 0)) if mode == "constant" else F.pad(x, (3, 3, 0, 0), mode=mode)
    cur = q(x)
    feat, raw = {}, {}
    for r in range(repeats):
        cur, un = layer(f"conv.{r}.layers.0.", cur, f"conv.{r}.layers.1.")
    feat[0], raw[0] = cur, un
    sizes = {0: cur.shape[-2:]}
    for l in range(1, levels):
        cur = q(F.avg_pool2d(raw[l - 1], 2, 2))
        sizes[l] = cur.shape[-2:]
        for r in range(repeats):
            cur, un = layer(f"convs.{l - 1}.{r}.layers.0.", cur, f"convs.{l - 1}.{r}.layers.1.")
        feat[l], raw[l] = cur, un
    xu = feat[levels - 1]
    for li, l in enumerate(range(levels - 2, 0, -1)):
        xu = q(F.interpolate(xu, size=tuple(sizes[l]), mode="bicubic"))
        xu = torch.cat((feat[l], xu), dim=1)
        for r in range(repeats):
            xu, _ = layer(f"upconvs.{li}.{r}.layers.0.", xu, f"upconvs.{li}.{r}.layers.1.")
    xu = q(F.interpolate(xu, size=tuple(sizes[0]), mode="bicubic"))
    y = torch.cat((xu, feat[0]), dim=1)
    R = repeats
    y, _ = layer(f"conv.{R}.", y, "gn.0.")
    y, _ = layer(f"conv.{R + 1}.", y, None, post="act")
    y, _ = layer(f"conv.{R + 2}.", y, None, post="none")
    return (y - y.mean(dim=(2, 3), keepdim=True))[..., 3:-3]


def curl_head(a: Tensor) -> Tuple[Tensor, Tensor]:
    """Streamfunction a [B,1,H,W] -> (u, v) [B,1,H,W] with antisymmetric wall values and
    zero corners (pytorch_networks_convae.py:2052-2068)."""
    kx = torch.tensor([-0.5, 0.0, 0.5], dtype=a.dtype).view(1, 1, 1, 3)
    ky = kx.view(1, 1, 3, 1)
    u = F.conv2d(a, ky)[:, :, :, 1:-1]
    v = -F.conv2d(a, kx)[:, :, 1:-1, :]
    u = F.pad(u, (1, 1, 1, 1), mode="replicate")
    u = u.clone()
    u[:, :, :, 0] = -u[:, :, :, 1]
    u[:, :, :, -1] = -u[:, :, :, -2]
    v = F.pad(v, (1, 1, 1, 1), mode="replicate")
    v = v.clone()
    v[:, :, 0, :] = -v[:, :, 1, :]
    v[:, :, -1, :] = -v[:, :, -2, :]
    for f in (u, v):
        f[:, :, 0, 0] = 0
        f[:, :, 0, -1] = 0
        f[:, :, -1, 0] = 0
        f[:, :, -1, -1] = 0
    return u, v


def unet_forward(sd, x, *, levels, repeats, act="gelu", r_p="replicate", loss_type="curl",
                 use_symm=False, a_bound=10.0, p_pred=False):
    """Returns (u, v, p, T) exactly as Unet.forward (:2026-2070)."""
    y = unet_features(sd, x, levels, repeats, act, r_p, use_symm)
    if loss_type in ("mae", "mass"):
        u, v, T = y[:, 0:1], y[:, 1:2], y[:, 2:3]
        p = y[:, 3:4] if p_pred else None
        return u, v, p, T
    a = y[:, 0:1] * a_bound
    T = torch.clip(y[:, 1], 0.0, 1.5)
    p = y[:, 2] if p_pred else None
    u, v = curl_head(a)
    return u[:, 0], v[:, 0], p, T


# --------------------------------------------------------------------------------------
# NewFluidNet (pytorch_networks_convae.py:1068-1390) — SURVEY.md §8(f) row N1
# --------------------------------------------------------------------------------------
def newfluidnet_layer_table(levels: int, c_i: int, c_h: int, c_o: int, repeats: int) -> list:
    """(state-dict prefix, C_in, C_out, k or None (= f), kind) in ctor order (:1215-1313)."""
    t = [("conv.0.", c_i, c_h, None, "fluid")]
    for l in range(levels):
        for r in range(repeats):
            t.append((f"convs.{l}.{r}.", c_h, c_h, None, "fluid"))
    t.append(("conv.1.", c_h * levels + c_i, c_h, 3, "head_gn"))
    t.append(("conv.2.", c_h, c_h, 3, "head_act"))
    t.append(("conv.3.", c_h, c_o, 3, "head_out"))
    return t


def newfluidnet_features(sd, x, *, levels, repeats, act, r_p, use_symm, factor=2) -> Tensor:
    """Everything up to and including the spatial zero-mean (:1315-1346).  Level l works on the input
    feature map average-pooled l times (floor mode), then is bicubically upsampled back to the INPUT size
    (the reference hard-codes 128 x 506, :1239-1244) and concatenated; the raw inputs are appended last."""
    size = tuple(x.shape[-2:])
    x_in = fluid_layer(sd, "conv.0.", x, act, r_p, use_symm)
    y = None
    pooled = x_in
    for l in range(levels):
        if l > 0:
            pooled = F.avg_pool2d(pooled, factor, factor)      # (the reference re-pools x_in l times: same values)
        cur = pooled
        for r in range(repeats):
            cur = fluid_layer(sd, f"convs.{l}.{r}.", cur, act, r_p, use_symm)
        if l > 0:
            cur = F.interpolate(cur, size=size, mode="bicubic")
            y = torch.cat((y, cur), dim=1)
        else:
            y = cur
    y = torch.cat((y, x), dim=1)
    y = conv2d_same(y, sd["conv.1.weight"], sd["conv.1.bias"], r_p, padding=(1, 1))
    c = y.shape[1]
    y = F.group_norm(y, int(c / 4), sd["gn.0.weight"], sd["gn.0.bias"], 1e-5)
    y = activation(act, y)
    y = activation(act, conv2d_same(y, sd["conv.2.weight"], sd["conv.2.bias"], r_p, padding=(1, 1)))
    y = conv2d_same(y, sd["conv.3.weight"], sd["conv.3.bias"], r_p, padding=(1, 1))
    return y - y.mean(dim=(2, 3), keepdim=True)


def newfluidnet_forward(sd, x, *, levels, repeats, act="selu", r_p="zeros", loss_type="mae", use_symm=False,
                        a_bound=4.0, p_pred=True, factor=2):
    """Returns (u, v, p) as NewFluidNet.forward (:1348-1390): u, v [B,H,W]; p is [B,1,H,W] for 'mae'/'mass' (the reference
    returns the un-squeezed slice, :1351-1358) and [B,H,W] for 'curl'; None without p_pred."""
    y = newfluidnet_features(sd, x, levels=levels, repeats=repeats, act=act, r_p=r_p, use_symm=use_symm, factor=factor)
    if loss_type in ("mae", "mass"):
        return y[:, 0], y[:, 1], (y[:, 2:3] if p_pred else None)
    a = y[:, 0:1] * a_bound
    p = y[:, 1] if p_pred else None
    u, v = curl_head(a)
    return u[:, 0], v[:, 0], p


# --------------------------------------------------------------------------------------
# ConvAE (.ipynb_checkpoints/pycold-checkpoint.py:989-1115)
# --------------------------------------------------------------------------------------
def convae_op_table(levels: int, c_i: int, c_h: int, c_o: int, repeats: int) -> list:
    """Flat op list of ConvAE.conv in ctor order (:1038-1092): ('fluid', idx, cin, cout),
    ('pool4',), ('up4',), ('final', idx, cin, cout)."""
    ops = [("fluid", 0, c_i, c_h)]
    idx = 1
    c = c_h
    for _ in range(levels):
        ops.append(("pool4", idx)); idx += 1
        cin, cout = c, c * 4
        for r in range(repeats):
            ops.append(("fluid", idx, cin if r == 0 else cout, cout)); idx += 1
        c *= 4
    c = int(c / 4)
    for r in range(repeats):
        ops.append(("fluid", idx, c * 4 if r == 0 else c, c)); idx += 1
    for _ in range(levels, 0, -1):
        ops.append(("up4", idx)); idx += 1
        cout = c / 4
        for r in range(repeats):
            ops.append(("fluid", idx, int(c if r == 0 else cout), int(cout))); idx += 1
        c = int(c / 4)
    ops.append(("final", idx, int(c), int(c_o)))
    return ops


def convae_forward(sd, x, *, levels, c_i, c_h, c_o, repeats, act="selu", r_p="zeros",
                   loss_type="mae", use_symm=False, a_bound=4.0, p_pred=True):
    for op in convae_op_table(levels, c_i, c_h, c_o, repeats):
        if op[0] == "fluid":
            x = fluid_layer(sd, f"conv.{op[1]}.", x, act, r_p, use_symm)
        elif op[0] == "pool4":
            x = F.avg_pool2d(x, 4, 4)
        elif op[0] == "up4":
            x = F.interpolate(x, scale_factor=4, mode="bicubic")
        else:
            pad = (2, 2) if loss_type == "curl" else (1, 1)
            x = conv2d_same(x, sd[f"conv.{op[1]}.weight"], sd[f"conv.{op[1]}.bias"], r_p, padding=pad)
    if loss_type == "curl":
        a = x[:, -1:] * a_bound
        kx = torch.tensor([-0.5, 0.0, 0.5], dtype=a.dtype).view(1, 1, 1, 3)
        u = F.conv2d(a, kx.view(1, 1, 3, 1))[..., :, 1:-1]
        v = -F.conv2d(a, kx)[..., 1:-1, :]
        if p_pred:
            x = torch.cat((x[:, :-2, 1:-1, 1:-1], u, v, x[:, -2:-1, 1:-1, 1:-1]), dim=1)
        else:
            x = torch.cat((x[:, :-1, 1:-1, 1:-1], u, v), dim=1)
    return x


# --------------------------------------------------------------------------------------
# fixed FD kernels ("valid", grid units; pytorch_networks_convae.py:183-260)
# --------------------------------------------------------------------------------------
def _k(vals, axis, dtype):
    t = torch.tensor(vals, dtype=dtype)
    return t.view(1, 1, 1, -1) if axis == "x" else t.view(1, 1, -1, 1)


def dx_right(f): return F.conv2d(f, _k([0, -1, 1], "x", f.dtype))
def dx_left(f): return F.conv2d(f, _k([-1, 1, 0], "x", f.dtype))
def dy_bot(f): return F.conv2d(f, _k([0, -1, 1], "y", f.dtype))
def dy_top(f): return F.conv2d(f, _k([-1, 1, 0], "y", f.dtype))
def dx_center(f): return F.conv2d(f, _k([-0.5, 0, 0.5], "x", f.dtype))
def dy_center(f): return F.conv2d(f, _k([-0.5, 0, 0.5], "y", f.dtype))
def du_dy(f): return F.conv2d(f, _k([1, -1, -1, 1], "y", f.dtype))
def dv_dx(f): return F.conv2d(f, _k([1, -1, -1, 1], "x", f.dtype))


def laplace(f):
    k = torch.tensor([[0, 1, 0], [1, -4, 1], [0, 1, 0]], dtype=f.dtype).view(1, 1, 3, 3)
    return F.conv2d(f, k)


def get_mass(u: Tensor, v: Tensor, bc: bool = False) -> Tensor:
    """Centred-difference divergence on the interior; H, W taken from the input
    (the reference hard-codes 128x506, pytorch_networks_convae.py:40-41)."""
    H, W = u.shape[-2:]
    u = u.reshape(-1, 1, H, W)
    v = v.reshape(-1, 1, H, W)
    du = dx_center(u)[..., 1:-1, :].clone()
    dv = dy_center(v)[..., :, 1:-1].clone()
    if bc:
        du[:, :, :, 0] *= 2.0 / 1.5
        du[:, :, :, -1] *= 2.0 / 1.5
        dv[:, :, 0, :] *= 2.0 / 1.5
        dv[:, :, -1, :] *= 2.0 / 1.5
    return du + dv


def pad_grad(x: Tensor, p=(1, 1, 1, 1)) -> Tensor:
    """Linear-extrapolation padding (pytorch_networks_convae.py:55-83): left, right,
    then the LAST-row side, then the FIRST-row side."""
    for _ in range(p[0]):
        x = torch.cat((2 * x[..., :, 0:1] - x[..., :, 1:2], x), dim=-1)
    for _ in range(p[1]):
        x = torch.cat((x, 2 * x[..., :, -1:] - x[..., :, -2:-1]), dim=-1)
    for _ in range(p[2]):
        x = torch.cat((x, 2 * x[..., -1:, :] - x[..., -2:-1, :]), dim=-2)
    for _ in range(p[3]):
        x = torch.cat((2 * x[..., 0:1, :] - x[..., 1:2, :], x), dim=-2)
    return x


def eta_torch(gamma, beta, z, T, Tref=0, zref=0):
    """Frank-Kamenetskii viscosity (pytorch_networks_convae.py:86-102)."""
    return torch.exp(torch.log(gamma) * (Tref - T) + torch.log(beta) * (z - zref))


def pad_uvp(u: Tensor, v: Tensor, p: Optional[Tensor] = None):
    """Wall padding: replicate tangentially, antisymmetric normal to the wall, zero
    corners (pytorch_networks_convae.py:145-178)."""
    def corners0(t):
        t[:, :, 0, 0] = 0.0
        t[:, :, 0, -1] = 0.0
        t[:, :, -1, 0] = 0.0
        t[:, :, -1, -1] = 0.0
        return t
    u = F.pad(u, (0, 0, 1, 1), mode="replicate")
    u = corners0(torch.cat((-u[..., 0:1], u, -u[..., -1:]), dim=3))
    v = F.pad(v, (1, 1, 0, 0), mode="replicate")
    v = corners0(torch.cat((-v[:, :, 0:1], v, -v[:, :, -1:]), dim=2))
    if p is not None:
        p = corners0(F.pad(p, (1, 1, 1, 1), mode="replicate"))
    return u, v, p


# --------------------------------------------------------------------------------------
# scaler.py:4-71
# --------------------------------------------------------------------------------------
def velocity_scaler(raq, fkt, fkp):
    return np.exp((raq / 10) * 1.80167667 + np.log(fkt) * 0.4330392 + np.log(fkp) * -0.46052953) * 5


def scale_var(x, raq, fkt, fkp, var):
    return x / velocity_scaler(raq, fkt, fkp) if var in ("uprev", "vprev") else x


def unscale_var(x, raq, fkt, fkp, var):
    return x * velocity_scaler(raq, fkt, fkp) if var in ("uprev", "vprev") else x


# --------------------------------------------------------------------------------------
# Trainer.loss_fn / get_loss (multigpu.py:122-134, 196-305)
# --------------------------------------------------------------------------------------
def loss_fn(x_true: Tensor, x_pred: Tensor, loss_scale: bool, norm: str = "l1"):
    """Returns (scaled loss, plain loss).  norm='l2' swaps |.| for (.)^2 (BASELINE's
    'L2 loss' wording; the reference only has L1)."""
    red = (lambda d: d.abs().mean()) if norm == "l1" else (lambda d: (d * d).mean())
    plain = red(x_true - x_pred)
    if not loss_scale:
        return plain, plain
    mx = torch.amax(x_true, dim=(1, 2), keepdim=True)
    mn = torch.amin(x_true, dim=(1, 2), keepdim=True)
    s = torch.clip(1.0 / (mx - mn), 1.0, 10.0)
    w = torch.full_like(x_true, 11.0)
    w[:, 2:-2, 2:-2] = 1.0
    return red((x_true - x_pred) * s * w), plain


def build_unet_input(gVTp: Tensor, roll_forward: int = 1) -> Tensor:
    """multigpu.py:198-248 with roll_forward == 1: the first ten channels of gVTp
    (xc, yc, dt, raq_nd, fkt_nd, fkp_nd, V, T, u, v) with xc, yc divided by 4 and dt by
    roll_forward.  An 11th channel (previous p, present when p_pred) is split off and
    dropped, exactly as the reference does."""
    x = gVTp[:, :10].clone()
    x[:, 0] = x[:, 0] / 4.0
    x[:, 1] = x[:, 1] / 4.0
    x[:, 2] = x[:, 2] / roll_forward
    return x


def unet_roll_forward(forward, gVTp: Tensor, paras: Tensor, roll_forward: int):
    """multigpu.py:207-248 for roll_forward = R > 1: the network is applied R * R times in a chain -- per outer round R - 1
    pre-steps under no_grad and one more step; only the LAST evaluation is differentiated (every earlier output enters the
    next input through a no_grad step or is overwritten).  Each input is rebuilt from the ORIGINAL xc, yc, dt and parameter
    channels and the previous evaluation's (T, u, v); the viscosity channel is re-derived from T after the PRE-steps only
    (the recomputation after a round's last step is commented out in the reference, :247: the next round starts with the V of
    the previous pre-step): V = log10(clip(exp(-ln(fkt) T + ln(fkp) (1 - yc)), 1e-8, 1)) / 8 (yc: the unscaled channel 1 of
    gVTp; paras = (raq, fkt, fkp)).  `forward(x)` -> (u, v, p, T) of the 10-channel input x.  Returns the last outputs."""
    R_ = int(roll_forward)
    x = build_unet_input(gVTp, R_)
    yc = gVTp[:, 1:2]
    fkt, fkp = paras.reshape(-1, 3)[:, 1].view(-1, 1, 1, 1), paras.reshape(-1, 3)[:, 2].view(-1, 1, 1, 1)
    B, _, H, W = gVTp.shape
    out = None
    for i in range(R_ * R_):
        last = i == R_ * R_ - 1
        with torch.set_grad_enabled(last):
            out = forward(x)
        if last:
            break
        u, v, p, T = out
        T = T.detach().reshape(B, 1, H, W)
        x = x.clone()
        if i % R_ != R_ - 1:                                 # a pre-step: the viscosity channel follows its T
            x[:, 6:7] = torch.log10(torch.clip(torch.exp(torch.log(fkt) * (0.0 - T) + torch.log(fkp) * (1.0 - yc)), 1e-8, 1.0)) / 8.0
        x[:, 7:8] = T
        x[:, 8:9], x[:, 9:10] = u.detach().reshape(B, 1, H, W), v.detach().reshape(B, 1, H, W)
    return out


def divergence_abs(u: Tensor, v: Tensor) -> Tensor:
    """|du/dx + dv/dy| on the (H-2)x(W-2) interior, [B,1,H-2,W-2] (multigpu.py:274-286)."""
    H, W = u.shape[-2:]
    u4 = u.reshape(-1, 1, H, W)
    v4 = v.reshape(-1, 1, H, W)
    return (dx_center(u4)[..., 1:-1, :] + dy_center(v4)[..., :, 1:-1]).abs()


def get_loss_unet(pred: Tuple[Tensor, Tensor, Optional[Tensor], Tensor], uvp: Tensor, *,
                  p_pred: bool, loss_type: str, loss_scale: bool = False,
                  loss_derivative: bool = False, norm: str = "l1",
                  momentum=None):
    """multigpu.py:250-305 given the network outputs.  Predictions are squeezed to
    [B,H,W] first (the reference's 4-D vs 3-D broadcast for 'mae'/'mass' is a defect,
    SURVEY.md Appendix A.6).  ``momentum`` = None or a dict(lambda_mom, yc, paras,
    scaler, T_field) enabling the build-defined Stokes momentum term.
    Returns (loss, loss_true_u, loss_true_v, loss_p, loss_T, mean mass[, mom])."""
    u, v, p, T = pred
    B = uvp.shape[0]
    H, W = uvp.shape[-2:]
    u = u.reshape(B, H, W)
    v = v.reshape(B, H, W)
    T = T.reshape(B, H, W)
    u_true, v_true = uvp[:, 0], uvp[:, 1]
    loss_u, true_u = loss_fn(u_true, u, loss_scale, norm)
    loss_v, true_v = loss_fn(v_true, v, loss_scale, norm)
    if p_pred:
        _, loss_p = loss_fn(uvp[:, 2], p.reshape(B, H, W), loss_scale, norm)
        _, loss_T = loss_fn(uvp[:, 3], T, loss_scale, norm)
    else:
        loss_p = torch.zeros((), dtype=u.dtype)
        _, loss_T = loss_fn(uvp[:, 2], T, loss_scale, norm)
    if loss_derivative:
        u4, v4 = u.reshape(B, 1, H, W), v.reshape(B, 1, H, W)
        ut4, vt4 = u_true.reshape(B, 1, H, W), v_true.reshape(B, 1, H, W)
        loss_u = loss_u + (dy_top(ut4) * 126 - dy_top(u4) * 126).abs().mean()
        loss_v = loss_v + (dx_left(vt4) * 126 - dx_left(v4) * 126).abs().mean()
        if not loss_scale:
            # reference quirk: without loss_scale, loss_fn returns the SAME tensor twice and the
            # in-place `loss_u += ...` (multigpu.py:283-284) therefore also bumps the reported
            # "true" u/v losses
            true_u, true_v = loss_u, loss_v
    mass = divergence_abs(u, v)
    if p_pred:
        loss = (loss_u + loss_v + loss_p + loss_T) / 4.0
    else:
        loss = (loss_u + loss_v + loss_T) / 3.0
    if loss_type == "mass":
        loss = loss + mass.mean()
    elif loss_type == "curl":
        loss = loss + (mass[:, :, :, 0].mean() + mass[:, :, :, -1].mean()
                       + mass[:, :, 0, :].mean() + mass[:, :, -1, :].mean())
    out = [loss, true_u, true_v, loss_p, loss_T, mass.mean()]
    if momentum is not None:
        rx, ry = momentum_residual(u, v, p.reshape(B, H, W) if p is not None else torch.zeros_like(u),
                                   momentum.get("T_field", T), momentum["yc"], momentum["paras"],
                                   momentum["scaler"])
        mom = rx.abs().mean() + ry.abs().mean()
        out[0] = out[0] + momentum["lambda_mom"] * mom
        out.append(mom)
    return tuple(out)


def get_loss_fluidnet(pred: Tuple[Tensor, Tensor, Optional[Tensor]], uvp: Tensor, *, p_pred: bool, loss_type: str,
                      loss_scale: bool = False, loss_derivative: bool = False, norm: str = "l1"):
    """The 'fluidnet' branch of Trainer.get_loss (multigpu.py:138-195, model_AD None) given the network outputs
    (u, v, p): no temperature term, averaging over 3 (2) terms, and — unlike the Unet branch — the SCALED pressure loss
    (`loss_p, _ = self.loss_fn(p_true, p)`, :146-148).  Predictions are squeezed to [B,H,W] (the 'mae' head returns p
    un-squeezed, which would broadcast against the [B,H,W] truth).  Returns the reference's 6-tuple, loss_T = 0."""
    u, v, p = pred
    B = uvp.shape[0]
    H, W = uvp.shape[-2:]
    u = u.reshape(B, H, W)
    v = v.reshape(B, H, W)
    u_true, v_true = uvp[:, 0], uvp[:, 1]
    loss_u, true_u = loss_fn(u_true, u, loss_scale, norm)
    loss_v, true_v = loss_fn(v_true, v, loss_scale, norm)
    if p_pred:
        loss_p, _ = loss_fn(uvp[:, 2], p.reshape(B, H, W), loss_scale, norm)
    else:
        loss_p = torch.zeros((), dtype=u.dtype)
    if loss_derivative:
        u4, v4 = u.reshape(B, 1, H, W), v.reshape(B, 1, H, W)
        ut4, vt4 = u_true.reshape(B, 1, H, W), v_true.reshape(B, 1, H, W)
        loss_u = loss_u + (dy_top(ut4) * 126 - dy_top(u4) * 126).abs().mean()
        loss_v = loss_v + (dx_left(vt4) * 126 - dx_left(v4) * 126).abs().mean()
        if not loss_scale:
            true_u, true_v = loss_u, loss_v            # the same in-place aliasing quirk as the Unet branch (:168-169)
    mass = divergence_abs(u, v)
    loss = (loss_u + loss_v + loss_p) / 3.0 if p_pred else (loss_u + loss_v) / 2.0
    if loss_type == "mass":
        loss = loss + mass.mean()
    elif loss_type == "curl":
        loss = loss + (mass[:, :, :, 0].mean() + mass[:, :, :, -1].mean()
                       + mass[:, :, 0, :].mean() + mass[:, :, -1, :].mean())
    return loss, true_u, true_v, loss_p, torch.zeros((), dtype=u.dtype), mass.mean()


# --------------------------------------------------------------------------------------
# ADNet + TS rollout (pytorch_networks_convae.py:266-568) — SURVEY.md §8(f) row N3
# --------------------------------------------------------------------------------------
def adnet_step(inputs: Tensor, dt=None, T_prev: Optional[Tensor] = None, CN_max: float = 0.1):
    """ADNet.forward (:522-568): one explicit upwind advection-diffusion step of the temperature on a non-uniform grid.
    inputs [B,6,H,W] = (u, v, T, RaQ/Ra, xc, yc); returns (T_next [B,1,H,W], dt).  Like the reference, the wall
    coordinates are written into xc / yc first (on a copy here)."""
    u = inputs[:, 0:1, 1:-1, 1:-1]
    v = inputs[:, 1:2, 1:-1, 1:-1]
    if T_prev is None:
        T_prev = inputs[:, 2:3]
    raq = inputs[:, 3:4, 1:-1, 1:-1]
    xc = inputs[:, 4:5].clone()
    yc = inputs[:, 5:6].clone()
    xc[:, :, :, 0] = 0.0
    xc[:, :, :, -1] = 4.0
    yc[:, :, 0, :] = 0.0
    yc[:, :, -1, :] = 1.0
    dx_l = dx_left(xc)[..., 1:-1, :]
    dx_r = dx_right(xc)[..., 1:-1, :]
    dy_t = dy_top(yc)[..., 1:-1]
    dy_b = dy_bot(yc)[..., 1:-1]
    dT_l = dx_left(T_prev)[..., 1:-1, :]
    dT_r = dx_right(T_prev)[..., 1:-1, :]
    dT_t = dy_top(T_prev)[..., 1:-1]
    dT_b = dy_bot(T_prev)[..., 1:-1]
    dT_dx = (dT_l / dx_l) * (u > 0) + (dT_r / dx_r) * (u < 0)
    dT_dy = (dT_t / dy_t) * (v > 0) + (dT_b / dy_b) * (v < 0)
    lap = (dT_r / dx_r - dT_l / dx_l) / (0.5 * dx_r + 0.5 * dx_l) + (dT_b / dy_b - dT_t / dy_t) / (0.5 * dy_b + 0.5 * dy_t)
    if dt is None:
        dx_min = torch.amin(dx_l)
        uv_mag = torch.max(torch.amax(torch.abs(u)), torch.amax(torch.abs(v)))
        dt = torch.min(0.5 * CN_max * dx_min / uv_mag, 0.5 * ((dx_min * dx_min) ** 2) / (dx_min ** 2 + dx_min ** 2))
    Tn = T_prev[..., 1:-1, 1:-1] + dt * (-u * dT_dx - v * dT_dy + lap + raq)
    Tn = F.pad(Tn, (1, 1, 1, 1), mode="replicate")
    Tn[:, :, 0, :] = 1.0
    Tn[:, :, -1, :] = 0.0
    return Tn, dt


def ts_rollout(stokes, T_prev, ycc, raq_nd, fkt_nd, fkp_nd, raq, fkt, fkp, xc, yc, ts: int, CN_max: float = 0.1):
    """TS.forward, 'newfluidnet' branch with an advection net (:354-476): ts times { build the 7-channel input
    (xc/4, yc/4, log10(clip(eta))/8, the three normalised parameters, T) -> stokes -> un-scale u, v -> ADNet step ->
    wall / side boundary conditions }.  `stokes(inp)` returns (u, v, p) (any shapes reshapeable to [B,1,H,W]).
    Returns (x dict, dts dict, u, v, p, V) like the reference."""
    B, _, H, W = T_prev.shape
    x, dts = {0: T_prev}, {}
    u = v = p = V = None
    for i in range(1, ts + 1):
        V = torch.clip(eta_torch(fkt, fkp, 1.0 - ycc, x[i - 1]), 1e-8, 1.0)
        inp = torch.cat((xc / 4.0, yc / 4.0, torch.log10(V) / 8, raq_nd.expand(1, 1, H, W), fkt_nd.expand(1, 1, H, W),
                         fkp_nd.expand(1, 1, H, W), x[i - 1]), dim=1)
        u, v, p = stokes(inp)
        sc = torch.exp((raq / 10) * 1.80167667 + torch.log(fkt) * 0.4330392 + torch.log(fkp) * -0.46052953) * 5
        u = (u * sc).reshape(-1, 1, H, W)
        v = (v * sc).reshape(-1, 1, H, W)
        if p is not None:
            p = p.reshape(-1, 1, H, W)
        ad_in = torch.cat((u, v, x[i - 1], torch.zeros_like(u) + raq, xc, yc), dim=1)
        Tn, dt = adnet_step(ad_in, CN_max=CN_max)
        Tn[:, :, 0, :] = 1
        Tn[:, :, -1, :] = 0
        Tn[:, :, :, 0:1] = Tn[:, :, :, 1:2]
        Tn[:, :, :, -1:] = Tn[:, :, :, -2:-1]
        x[i], dts[i] = Tn, dt
    return x, dts, u, v, p, V


def ts_rollout_unet(stokes, T_prev, ycc, raq_nd, fkt_nd, fkp_nd, fkt, fkp, xc, yc, u_prev, v_prev, dt, ts: int):
    """TS.forward, 'unet' branch (:411-446): the Stokes net predicts the next T itself; u_prev, v_prev and dt are the
    caller's for every step, no advection net, p = None, V = the last input's viscosity channel log10(clip(eta))/8.
    `stokes(inp)` returns (u, v, p, T)."""
    B, _, H, W = T_prev.shape
    x = {0: T_prev}
    u = v = V = None
    for i in range(1, ts + 1):
        V = torch.log10(torch.clip(eta_torch(fkt, fkp, 1.0 - ycc, x[i - 1]), 1e-8, 1.0)) / 8.0
        inp = torch.cat((xc / 4.0, yc / 4.0, dt, raq_nd.expand(1, 1, H, W), fkt_nd.expand(1, 1, H, W), fkp_nd.expand(1, 1, H, W),
                         V, x[i - 1], u_prev, v_prev), dim=1)
        u, v, _, T = stokes(inp)
        Tn = T.reshape(B, 1, H, W).clone()
        Tn[:, :, 0, :] = 1
        Tn[:, :, -1, :] = 0
        Tn[:, :, :, 0:1] = Tn[:, :, :, 1:2]
        Tn[:, :, :, -1:] = Tn[:, :, :, -2:-1]
        x[i] = Tn
    return x, {}, u.reshape(B, 1, H, W), v.reshape(B, 1, H, W), None, V


# --------------------------------------------------------------------------------------
# Stokes momentum residual — BUILD-DEFINED (no reference implementation)
# --------------------------------------------------------------------------------------
INV_H = 126.0   # 1/h, grid of 126 layers (prepare_gaia_ini.py:22-27; multigpu.py:163-166 uses x126)
RA = 1.0        # prepare_gaia_ini.py:117


def viscosity(T: Tensor, yc: Tensor, paras: Tensor) -> Tensor:
    """eta = clip(exp(-ln(FKT) T + ln(FKP) (1 - y)), 1e-8, 1) (pytorch_networks_convae.py:86-102,
    datasetio.py:616-619).  T [B,H,W], yc [H,W] or [B,H,W], paras [B,3] = (RaQ, FKT, FKP)."""
    fkt = paras[:, 1].view(-1, 1, 1)
    fkp = paras[:, 2].view(-1, 1, 1)
    return torch.clip(eta_torch(fkt, fkp, 1.0 - yc, T), 1e-8, 1.0)


def momentum_residual(u, v, p, T, yc, paras, scaler, inv_h: float = INV_H):
    """R_x, R_y on the (H-2)x(W-2) interior, flux form with face-averaged viscosity:

      R_x = -dp/dx + d/dx(2 eta dU/dx) + d/dy(eta (dU/dy + dV/dx))
      R_y = -dp/dy + d/dy(2 eta dV/dy) + d/dx(eta (dU/dy + dV/dx)) + Ra T

    U = u*scaler, V = v*scaler (physical velocities, scaler.py:6-13); one-sided face
    differences (dx_right/dx_left, dy_bot/dy_top, pytorch_networks_convae.py:183-214);
    face viscosity = arithmetic mean of the two nodes; 1/h = 126; the flux-form pattern
    follows ADNet's Laplacian (pytorch_networks_convae.py:550-552).  The viscosity is
    evaluated from T.detach() (no gradient through eta, as the reference's roll-forward
    does at multigpu.py:226-231); the buoyancy term keeps its gradient w.r.t. T.
    All inputs [B,H,W] (row 0 = bottom); scaler [B]."""
    s = scaler.view(-1, 1, 1).to(u.dtype)
    U, V = u * s, v * s
    eta = viscosity(T.detach(), yc, paras).to(u.dtype)
    ih = inv_h
    exf = 0.5 * (eta[:, :, :-1] + eta[:, :, 1:])          # x-face j+1/2  [B,H,W-1]
    eyf = 0.5 * (eta[:, :-1, :] + eta[:, 1:, :])          # y-face i+1/2  [B,H-1,W]
    # normal fluxes
    Fx = 2.0 * exf * (U[:, :, 1:] - U[:, :, :-1]) * ih    # [B,H,W-1]
    Fy = 2.0 * eyf * (V[:, 1:, :] - V[:, :-1, :]) * ih    # [B,H-1,W]
    # shear stress on y-faces (i+1/2, j), j in 1..W-2
    dVdx_c = 0.5 * (V[:, :, 2:] - V[:, :, :-2]) * ih       # centred d/dx at nodes, [B,H,W-2]
    Tyf = eyf[:, :, 1:-1] * ((U[:, 1:, 1:-1] - U[:, :-1, 1:-1]) * ih
                             + 0.5 * (dVdx_c[:, :-1] + dVdx_c[:, 1:]))       # [B,H-1,W-2]
    # shear stress on x-faces (i, j+1/2), i in 1..H-2
    dUdy_c = 0.5 * (U[:, 2:, :] - U[:, :-2, :]) * ih       # [B,H-2,W]
    Txf = exf[:, 1:-1, :] * ((V[:, 1:-1, 1:] - V[:, 1:-1, :-1]) * ih
                             + 0.5 * (dUdy_c[:, :, :-1] + dUdy_c[:, :, 1:]))  # [B,H-2,W-1]
    dpdx = 0.5 * (p[:, 1:-1, 2:] - p[:, 1:-1, :-2]) * ih
    dpdy = 0.5 * (p[:, 2:, 1:-1] - p[:, :-2, 1:-1]) * ih
    Rx = (-dpdx + (Fx[:, 1:-1, 1:] - Fx[:, 1:-1, :-1]) * ih + (Tyf[:, 1:, :] - Tyf[:, :-1, :]) * ih)
    Ry = (-dpdy + (Fy[:, 1:, 1:-1] - Fy[:, :-1, 1:-1]) * ih + (Txf[:, :, 1:] - Txf[:, :, :-1]) * ih
          + RA * T[:, 1:-1, 1:-1])
    return Rx, Ry


# --------------------------------------------------------------------------------------
# whole training step on CPU (multigpu.py:307-338 + Adam :761-763) — the cpu_baseline leg
# --------------------------------------------------------------------------------------
class CpuUnetStep:
    """Holds fp32 leaf parameters (reference state-dict layout) and runs
    zero_grad -> forward -> loss -> backward -> Adam on the CPU with ATen ops."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict, lr=1e-3, weight_decay=0.0):
        self.cfg = dict(cfg)
        self.sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.Adam(list(self.sd.values()), lr=lr, weight_decay=weight_decay)

    def loss(self, gVTp: Tensor, uvp: Tensor, **loss_kw):
        c = self.cfg
        if c.get("rebuild_input", True):
            gVTp = build_unet_input(gVTp)
        pred = unet_forward(self.sd, gVTp, levels=c["levels"], repeats=c["repeats"], act=c["act"],
                            r_p=c["r_p"], loss_type=c["loss_type"], use_symm=c["use_symm"],
                            a_bound=c.get("a_bound", 10.0), p_pred=c["p_pred"])
        return get_loss_unet(pred, uvp, p_pred=c["p_pred"], loss_type=c["loss_type"], **loss_kw), pred

    def step(self, gVTp, uvp, **loss_kw):
        self.opt.zero_grad(set_to_none=True)
        out, _ = self.loss(gVTp, uvp, **loss_kw)
        out[0].backward()
        self.opt.step()
        return [float(o.detach()) for o in out]
