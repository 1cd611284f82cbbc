"""Network front ends with the reference's names, constructor signatures and state_dict keys
(reference pytorch_networks_convae.py; ConvAE from .ipynb_checkpoints/pycold-checkpoint.py:989-1115),
executed by the HIP engine on MI355X.

Only the modules on the training hot path are provided (FluidLayer, Unet, ConvAE) plus the
small field helpers the reference exports from this module.  Everything numeric inside
forward()/backward() runs in libmantle_hip kernels; there is no CPU path.
"""
import math
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from . import _lib as L
from .engine import convae_graph, newfluidnet_graph, single_layer_graph, unet_graph
from .hipnet import HipNetMixin
from .learned_padding import BoundaryLearnedConvolution2D   # noqa: F401  (reference :802-1065)
from .symmetric_layers_torch import SymmetricConv2d

_SUPPORTED_ACTS = ("gelu", "relu", "silu", "tanh", "selu", "elu")


class color:
    PURPLE = "\033[95m"
    CYAN = "\033[96m"
    DARKCYAN = "\033[36m"
    BLUE = "\033[94m"
    GREEN = "\033[92m"
    YELLOW = "\033[93m"
    RED = "\033[91m"
    BOLD = "\033[1m"
    UNDERLINE = "\033[4m"
    END = "\033[0m"


# --------------------------------------------------------------------------------------------------
# field helpers (API surface of the reference module; host-side/data-preparation use, device-agnostic)
# --------------------------------------------------------------------------------------------------
def _dx(v, w):      # 1x3 / 1x4 cross-correlation, 'valid'
    n = len(w)
    W = v.shape[-1]
    return sum(w[i] * v[..., i:W - n + 1 + i] for i in range(n) if w[i] != 0)


def _dy(v, w):
    n = len(w)
    H = v.shape[-2]
    return sum(w[i] * v[..., i:H - n + 1 + i, :] for i in range(n) if w[i] != 0)


def dx_right(v, device=None): return _dx(v, (0, -1, 1))          # reference :183-190
def dy_bot(v, device=None): return _dy(v, (0, -1, 1))            # :193-199
def dx_left(v, device=None): return _dx(v, (-1, 1, 0))           # :202-208
def dy_top(v, device=None): return _dy(v, (-1, 1, 0))            # :211-215
def dx_center(v, device=None): return _dx(v, (-0.5, 0, 0.5))     # :218-224
def dy_center(v, device=None): return _dy(v, (-0.5, 0, 0.5))     # :227-233
def du_dy(v, device=None): return _dy(v, (1, -1, -1, 1))         # :236-242
def dv_dx(v, device=None): return _dx(v, (1, -1, -1, 1))         # :245-251


def laplace(v, device=None):                                      # :254-260 (5-point)
    return (v[..., :-2, 1:-1] + v[..., 2:, 1:-1] + v[..., 1:-1, :-2] + v[..., 1:-1, 2:] - 4 * v[..., 1:-1, 1:-1])


def get_mass(u, v, bc=False):
    """Centred-difference divergence on the interior (reference :27-52; H, W taken from the input
    instead of the hard-coded 128 x 506)."""
    H, W = u.shape[-2:]
    u = u.reshape(-1, 1, H, W)
    v = v.reshape(-1, 1, H, W)
    du_dx = dx_center(u)[..., 1:-1, :].clone()
    dv_dy = dy_center(v)[..., :, 1:-1].clone()
    if bc:
        du_dx[:, :, :, 0] *= 2.0 / 1.5
        du_dx[:, :, :, -1] *= 2.0 / 1.5
        dv_dy[:, :, 0, :] *= 2.0 / 1.5
        dv_dy[:, :, -1, :] *= 2.0 / 1.5
    return du_dx + dv_dy


def pad_grad(x, p=(1, 1, 1, 1)):
    """Linear-extrapolation padding: p = (left, right, last-row side, first-row side) (reference :55-83)."""
    for _ in range(p[0]):
        x = torch.cat((2 * x[:, :, :, 0:1] - x[:, :, :, 1:2], x), dim=-1)
    for _ in range(p[1]):
        x = torch.cat((x, 2 * x[:, :, :, -1:] - x[:, :, :, -2:-1]), dim=-1)
    for _ in range(p[2]):
        x = torch.cat((x, 2 * x[:, :, -1:, :] - x[:, :, -2:-1, :]), dim=-2)
    for _ in range(p[3]):
        x = torch.cat((2 * x[:, :, 0:1, :] - x[:, :, 1:2, :], x), dim=-2)
    return x


def eta_torch(gamma, beta, z, T, Tref=0, zref=0):
    """Frank-Kamenetskii viscosity exp(ln(gamma)(Tref-T) + ln(beta)(z-zref)) (reference :86-102)."""
    return torch.exp(torch.log(gamma) * (Tref - T) + torch.log(beta) * (z - zref))


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def exists(val):
    return val is not None


def get_lr(optimizer):
    for param_group in optimizer.param_groups:
        return param_group["lr"]


def pad_uvp(u, v, p=None):
    """Wall padding of (u, v, p): replicate along the wall, antisymmetric normal velocity, zero
    corners (reference :145-178)."""
    def zero_corners(t):
        t[:, :, 0, 0] = 0.0
        t[:, :, 0, -1] = 0.0
        t[:, :, -1, 0] = 0.0
        t[:, :, -1, -1] = 0.0
        return t
    u = torch.cat((u[:, :, 0:1], u, u[:, :, -1:]), dim=2)
    u = zero_corners(torch.cat((-u[:, :, :, 0:1], u, -u[:, :, :, -1:]), dim=3))
    v = torch.cat((v[:, :, :, 0:1], v, v[:, :, :, -1:]), dim=3)
    v = zero_corners(torch.cat((-v[:, :, 0:1, :], v, -v[:, :, -1:, :]), dim=2))
    if p is not None:
        p = torch.cat((p[:, :, 0:1], p, p[:, :, -1:]), dim=2)
        p = zero_corners(torch.cat((p[:, :, :, 0:1], p, p[:, :, :, -1:]), dim=3))
    return u, v, p


# --------------------------------------------------------------------------------------------------
# FluidLayer (reference :702-799): conv -> GroupNorm -> activation -> dropout(p)
# --------------------------------------------------------------------------------------------------
def _check_common(act_fn, r_p, dilation, drop_rate=0.0, spectral_conv=False, blurr=False):
    if act_fn not in _SUPPORTED_ACTS:
        raise NotImplementedError(f"act_fn={act_fn!r}: supported on the HIP path: {_SUPPORTED_ACTS} "
                                  "('sine' is undefined in the reference itself)")
    if r_p not in ("zeros", "replicate", "reflect", "learned"):
        raise NotImplementedError(f"r_p={r_p!r}: the HIP path implements zeros / replicate / reflect / learned padding")
    if dilation != 1:
        raise NotImplementedError("dilation != 1 is not implemented on the HIP path")
    if drop_rate not in (0, 0.0):
        raise NotImplementedError("dropout with p > 0 is not implemented on the HIP path (reference default 0)")
    if spectral_conv:
        raise NotImplementedError("spectral_conv is out of scope (FFT path)")
    if blurr:
        raise NotImplementedError("blurr is out of scope")


class FluidLayer(nn.Module, HipNetMixin):
    def __init__(self, c_i: int, c_o: int, act_fn: str = "selu", r_p="zeros", use_symm=False, dilation=1, f=3,
                 drop_rate=0.0):
        super().__init__()
        _check_common(act_fn, r_p, dilation, drop_rate)
        self.r_p = "constant" if r_p == "zeros" else r_p
        self.act_fn = act_fn
        self.layers = nn.ModuleList()
        h_s = int(c_o / 4) if c_o > 4 else int(c_o / 2)
        if r_p == "learned":
            self.layers.append(BoundaryLearnedConvolution2D(c_i, c_o, k=f, use_symm=use_symm))      # reference :760-763
        elif use_symm:
            self.layers.append(SymmetricConv2d(c_i, c_o, kernel_size=f, padding="same", dilation=dilation,
                                               padding_mode=r_p, symmetry={"h": h_s, "v": 0, "hv": 0}))
        else:
            self.layers.append(nn.Conv2d(c_i, c_o, kernel_size=f, padding="same", dilation=dilation,
                                         padding_mode=r_p))
        self.layers.append(torch.nn.GroupNorm(int(c_o / min(4, c_o)), c_o))
        self._init_hipnet(single_layer_graph(c_i, c_o, f, f // 2, "zeros" if r_p == "learned" else r_p,
                                             h_s if use_symm else 0, L.POST_GN_ACT, act_fn, int(c_o / min(4, c_o)), gn=True,
                                             learned=(r_p == "learned")))

    def forward(self, inputs, bc_x=1, bc_y=1):
        if self.r_p == "learned" and (bc_x != 1 or bc_y != 1):
            raise NotImplementedError("a stand-alone FluidLayer runs the learned padding with bc_x = bc_y = 1 (inside the Unet "
                                      "graph the first layer's bc_x = 4 is part of the graph)")
        return self._run_graph(inputs)


# --------------------------------------------------------------------------------------------------
# curl head (Unet :2038-2070) as an autograd function over the HIP kernels
# --------------------------------------------------------------------------------------------------
class _CurlHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, a_bound):
        """y: [N, C, H, W] f32 network output; channel 0 = streamfunction, channel 1 = T."""
        N, Cc, H, W = y.shape
        y = y.contiguous()
        u = torch.empty((N, H, W), dtype=torch.float32, device=y.device)
        v = torch.empty_like(u)
        Tc = torch.empty_like(u)
        L.call("mc_curl_head_fwd", L.ptr(y), y.data_ptr() + 4 * H * W, N, H, W, Cc * H * W, float(a_bound), 0.0, 1.5,
               L.ptr(u), L.ptr(v), L.ptr(Tc), L.stream())
        ctx.save_for_backward(y)
        ctx.a_bound = float(a_bound)
        return u, v, Tc

    @staticmethod
    def backward(ctx, gu, gv, gT):
        (y,) = ctx.saved_tensors
        N, Cc, H, W = y.shape
        gy = torch.zeros_like(y)
        ws = torch.empty(2 * N * (H - 2) * (W - 2), dtype=torch.float32, device=y.device)
        gu = (gu if gu is not None else torch.zeros((N, H, W), device=y.device)).contiguous().float()
        gv = (gv if gv is not None else torch.zeros((N, H, W), device=y.device)).contiguous().float()
        gT = (gT if gT is not None else torch.zeros((N, H, W), device=y.device)).contiguous().float()
        L.call("mc_curl_head_bwd", L.ptr(gu), L.ptr(gv), L.ptr(gT), y.data_ptr() + 4 * H * W, N, H, W, ctx.a_bound,
               0.0, 1.5, L.ptr(gy), gy.data_ptr() + 4 * H * W, Cc * H * W, Cc * H * W, L.ptr(ws), L.stream())
        return gy, None


# --------------------------------------------------------------------------------------------------
# Unet (reference :1700-2070)
# --------------------------------------------------------------------------------------------------
class Unet(nn.Module, HipNetMixin):
    """Symmetric-convolution U-Net for the Stokes surrogate.  Same constructor, parameters and
    state_dict keys as the reference; forward returns (u, v, p, T)."""

    def __init__(self, levels: int, c_i: int, c_h: int, c_o: int, device=torch.device("cpu"), act_fn: str = "gelu",
                 r_p="replicate", loss_type="curl", use_symm=False, dilation=1, a_bound=10.0, use_cosine=False,
                 repeats=2, use_skip=False, f=5, p_pred=False, spectral_conv=False, blurr=False, drop_rate=0.0):
        super().__init__()
        _check_common(act_fn, r_p, dilation, drop_rate, spectral_conv, blurr)
        self.levels, self.loss_type, self.a_bound = levels, loss_type, a_bound
        self.use_cosine, self.repeats, self.use_skip, self.p_pred = use_cosine, repeats, use_skip, p_pred
        self.blurrer = None
        self.r_p = "constant" if r_p == "zeros" else r_p
        self.act_fn = act_fn
        graph = unet_graph(levels, c_i, c_h, c_o, act=act_fn, r_p=r_p, use_symm=use_symm, repeats=repeats, f=f)

        def fl(cin, cout):
            return FluidLayer(cin, cout, act_fn, r_p, use_symm, dilation, f=f, drop_rate=drop_rate)

        # module tree in the reference's construction order (:1842-1983) so state_dict keys match
        self.conv = nn.ModuleList()
        self.gn = nn.ModuleList()
        for r in range(repeats):
            self.conv.append(fl(c_i if r == 0 else c_h, c_h))
        self.pool = nn.AvgPool2d((2, 2), stride=2)
        self.convs = nn.ModuleList()
        c = c_h
        for l in range(1, levels):
            self.convs.append(nn.ModuleList())
            for r in range(repeats):
                self.convs[-1].append(fl(int(c / 2) if (r == 0 and l > 1) else c, c))
            c *= 2
        c = int(c / 2)
        self.upconvs = nn.ModuleList()
        for l in range(levels - 2, 0, -1):
            self.upconvs.append(nn.ModuleList())
            for r in range(repeats):
                self.upconvs[-1].append(fl(c + int(c / 2) if r == 0 else int(c / 2), int(c / 2)))
            c = int(c / 2)
        if r_p == "learned":                                                   # reference :1946-1983
            self.conv.append(BoundaryLearnedConvolution2D(int(c * 2), c, k=f, use_symm=use_symm))
            self.gn.append(torch.nn.GroupNorm(int(c / 4), c))
            self.conv.append(BoundaryLearnedConvolution2D(c, c, k=f, use_symm=use_symm))
            self.conv.append(BoundaryLearnedConvolution2D(c, c_o, k=f, use_symm=use_symm))
        else:
            self.conv.append(nn.Conv2d(int(c * 2), c, kernel_size=f, padding="same", dilation=dilation, padding_mode=r_p))
            self.gn.append(torch.nn.GroupNorm(int(c / 4), c))
            self.conv.append(nn.Conv2d(c, c, kernel_size=f, padding="same", padding_mode=r_p))
            self.conv.append(nn.Conv2d(c, c_o, kernel_size=f, padding="same", padding_mode=r_p))
        self._init_hipnet(graph)

    def features(self, inputs):
        """(y - mean_HW(y))[..., 3:-3] — everything up to the output heads (:1985-2024)."""
        return self._run_graph(inputs)

    def forward(self, inputs):
        y = self.features(inputs)
        if self.loss_type in ("mae", "mass"):
            u, v, T = y[:, 0:1], y[:, 1:2], y[:, 2:3]
            p = y[:, 3:4] if self.p_pred else None
            return u, v, p, T
        elif self.loss_type == "curl":
            u, v, T = _CurlHead.apply(y, self.a_bound)
            p = y[:, 2] if self.p_pred else None
            return u, v, p, T
        raise ValueError(self.loss_type)


# --------------------------------------------------------------------------------------------------
# NewFluidNet (reference :1068-1390) — SURVEY.md §8(f) row N1
# --------------------------------------------------------------------------------------------------
class _CurlUV(torch.autograd.Function):
    """u, v from the streamfunction channel with antisymmetric walls and zero corners (:1360-1388); no T channel."""

    @staticmethod
    def forward(ctx, y, a_bound):
        N, Cc, H, W = y.shape
        y = y.contiguous()
        u = torch.empty((N, H, W), dtype=torch.float32, device=y.device)
        v = torch.empty_like(u)
        L.call("mc_curl_head_fwd", L.ptr(y), None, N, H, W, Cc * H * W, float(a_bound), 0.0, 0.0, L.ptr(u), L.ptr(v), None,
               L.stream())
        ctx.shape, ctx.a_bound = (N, Cc, H, W), float(a_bound)
        return u, v

    @staticmethod
    def backward(ctx, gu, gv):
        N, Cc, H, W = ctx.shape
        dev = gu.device if gu is not None else gv.device
        gy = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=dev)
        ws = torch.empty(2 * N * (H - 2) * (W - 2), dtype=torch.float32, device=dev)
        gu = (gu if gu is not None else torch.zeros((N, H, W), device=dev)).contiguous().float()
        gv = (gv if gv is not None else torch.zeros((N, H, W), device=dev)).contiguous().float()
        L.call("mc_curl_head_bwd", L.ptr(gu), L.ptr(gv), None, None, N, H, W, ctx.a_bound, 0.0, 0.0, L.ptr(gy), None,
               Cc * H * W, Cc * H * W, L.ptr(ws), L.stream())
        return gy, None


class NewFluidNet(nn.Module, HipNetMixin):
    """Multi-resolution trunk of the deployed surrogate (`-net newfluidnet -l 5 -f 16 -r 6 -k 5`): level l works on the first
    feature map average-pooled l times, is bicubically upsampled back and concatenated with the others and the raw inputs.
    Same constructor, module tree and state_dict keys as the reference; forward returns (u, v, p).  The reference
    hard-codes the 128 x 506 grid in its Upsample modules (:1239-1244); here the levels are upsampled to the input's size."""

    def __init__(self, levels: int, c_i: int, c_h: int, c_o: int, device=None, act_fn: str = "selu", r_p="zeros",
                 loss_type="mae", use_symm=False, dilation=1, a_bound=4.0, use_cosine=False, repeats=3, use_skip=False,
                 f=3, p_pred=True, spectral_conv=False, blurr=False, drop_rate=0.0, factor=2):
        super().__init__()
        _check_common(act_fn, r_p, dilation, drop_rate, spectral_conv, blurr)
        self.levels, self.loss_type, self.a_bound = levels, loss_type, a_bound
        self.use_cosine, self.repeats, self.use_skip, self.p_pred = use_cosine, repeats, use_skip, p_pred
        self.c_h, self.c_i, self.c_o = c_h, c_i, c_o
        self.blurrer = None
        self.r_p = "constant" if r_p == "zeros" else r_p
        graph = newfluidnet_graph(levels, c_i, c_h, c_o, act=act_fn, r_p=r_p, use_symm=use_symm, repeats=repeats, f=f,
                                  factor=factor)

        def fl(cin, cout):
            return FluidLayer(cin, cout, act_fn, r_p, use_symm, dilation, f=f, drop_rate=drop_rate)

        # module tree in the reference's construction order (:1148-1313) so state_dict keys match
        self.conv = nn.ModuleList()
        self.gn = nn.ModuleList()
        self.unpool = nn.ModuleList()
        self.conv.append(fl(c_i, c_h))
        self.pool = nn.AvgPool2d((factor, factor), stride=factor)
        for _ in range(1, levels):
            self.unpool.append(nn.Upsample(size=(128, 506), mode="bicubic"))
        self.convs = nn.ModuleList()
        for l in range(levels):
            self.convs.append(nn.ModuleList())
            for r in range(repeats):
                self.convs[l].append(fl(c_h, c_h))
        if r_p == "learned":                                                   # reference :1268-1313
            self.conv.append(BoundaryLearnedConvolution2D(c_h * levels + c_i, c_h, k=f, use_symm=use_symm))
            self.gn.append(torch.nn.GroupNorm(int(c_h / 4), c_h))
            self.conv.append(BoundaryLearnedConvolution2D(c_h, c_h, k=f, use_symm=use_symm))
            self.conv.append(BoundaryLearnedConvolution2D(c_h, c_o, k=f, use_symm=use_symm))
        else:
            self.conv.append(nn.Conv2d(c_h * levels + c_i, c_h, kernel_size=3, padding=(1, 1), dilation=dilation, padding_mode=r_p))
            self.gn.append(torch.nn.GroupNorm(int(c_h / 4), c_h))
            self.conv.append(nn.Conv2d(c_h, c_h, kernel_size=3, padding=(1, 1), padding_mode=r_p))
            self.conv.append(nn.Conv2d(c_h, c_o, kernel_size=3, padding=(1, 1), padding_mode=r_p))
        self._init_hipnet(graph)

    def features(self, inputs):
        """y - mean_HW(y): everything up to the output heads (:1315-1346)."""
        return self._run_graph(inputs)

    def forward(self, inputs):
        y = self.features(inputs)
        if self.loss_type in ("mae", "mass"):
            # (the reference returns p un-squeezed, [B,1,H,W], :1351-1358)
            return y[:, 0], y[:, 1], (y[:, 2:3] if self.p_pred else None)
        elif self.loss_type == "curl":
            u, v = _CurlUV.apply(y, self.a_bound)
            return u, v, (y[:, 1] if self.p_pred else None)
        raise ValueError(self.loss_type)


# --------------------------------------------------------------------------------------------------
# ADNet + TS: the inference rollout (reference :266-568) — SURVEY.md §8(f) row N3
# --------------------------------------------------------------------------------------------------
class ADNet(nn.Module):
    """Explicit upwind advection-diffusion step of the temperature field (reference :478-568) as ONE fused HIP stencil
    kernel (+ a reduction for the CFL time step).  forward(inputs [B,6,H,W] = (u, v, T, RaQ/Ra, xc, yc), dt=None,
    T_prev=None) -> (T_next [B,1,H,W], dt); the grid channels of sample 0 are used for the whole batch."""

    def __init__(self, device=None, r_p="zeros", CN_max=0.1):
        super().__init__()
        self.device, self.CN_max = device, CN_max
        self._ws = None

    def step(self, u, v, T_prev, xc, yc, *, raq_field=None, raq_scalar=None, vel_scale=None, dt=None, out=None, dt_out=None):
        """Raw form on device planes (f32, contiguous): u, v [B,H,W] (or a strided view with a batch stride), T_prev [B,H,W],
        xc / yc [H,W]; returns (T_next [B,H,W], dt device scalar)."""
        L.require_cuda(T_prev, "T_prev")
        B, H, W = T_prev.shape[0], T_prev.shape[-2], T_prev.shape[-1]
        dev = T_prev.device
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.zeros(2, dtype=torch.int32, device=dev)
        out = out if out is not None else torch.empty((B, H, W), dtype=torch.float32, device=dev)
        dt_dev = dt_out if dt_out is not None else torch.empty(1, dtype=torch.float32, device=dev)
        if dt is not None:
            dt_dev.copy_(torch.as_tensor(dt, dtype=torch.float32).reshape(1))
        uvs = u.stride(0) if u.dim() >= 3 and B > 1 else H * W
        L.call("mc_adnet_step", L.ptr(u), L.ptr(v), int(uvs), L.ptr(vel_scale), L.ptr(T_prev), L.ptr(raq_field),
               L.ptr(raq_scalar), L.ptr(xc), L.ptr(yc), B, H, W, float(self.CN_max), int(dt is None), L.ptr(dt_dev),
               L.ptr(self._ws), L.ptr(out), L.stream())
        return out, dt_dev

    def forward(self, inputs, dt=None, T_prev=None):
        inputs = inputs.float().contiguous()
        L.require_cuda(inputs, "inputs")
        B, _, H, W = inputs.shape
        Tp = (T_prev if T_prev is not None else inputs[:, 2]).float().reshape(B, H, W).contiguous()
        u, v = inputs[:, 0], inputs[:, 1]                      # planes of `inputs`: batch stride 6 H W
        out, dt_dev = self.step(u, v, Tp, inputs[0, 4].contiguous(), inputs[0, 5].contiguous(),
                                raq_field=inputs[:, 3].contiguous(), dt=dt)
        return out.view(B, 1, H, W), (dt_dev[0] if dt is None else dt)


class TS(nn.Module):
    """Evaluation wrapper for time stepping (reference :266-476), 'newfluidnet' branch with an advection net: ts times
    { input builder -> Stokes net -> un-scale u, v -> ADNet step -> boundary rows / columns }, every stage a HIP kernel on
    HBM-resident planes (no host round trip inside the loop).  Same constructor and forward signature as the reference;
    returns (x dict, dts dict, u, v, p, V)."""

    def __init__(self, stokes, ad, device, ts=8, advection_scheme=2, scale=True, p_pred=True, net="fluidnet", use_graph=False):
        super().__init__()
        self.use_graph, self._g = bool(use_graph), None        # use_graph: one rollout step captured as a HIP graph and replayed
        if net not in ("newfluidnet", "unet"):
            raise NotImplementedError("TS on the HIP path covers net='newfluidnet' (the deployed configuration) and 'unet'")
        if ad is None and net == "newfluidnet":
            raise NotImplementedError("TS needs the advection net (ADNet)")
        self.stokes, self.ad, self.ts, self.device = stokes, ad, ts, device
        self.advection_scheme, self.scale, self.p_pred, self.net = advection_scheme, scale, p_pred, net

    @torch.no_grad()
    def forward(self, T_prev, sdf, sdf2, ycc, raq_nd, fkt_nd, fkp_nd, raq, fkt, fkp, xc, yc, u_prev=None, v_prev=None,
                dt=None):
        if self.net == "unet":
            return self._forward_unet(T_prev, ycc, raq_nd, fkt_nd, fkp_nd, raq, fkt, fkp, xc, yc, u_prev, v_prev, dt)
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        f = dict(dtype=torch.float32, device=dev)
        B, _, H, W = T_prev.shape
        plane = lambda t: torch.as_tensor(t).to(**f).reshape(-1, H, W)[0].contiguous()  # noqa: E731
        xcp, ycp, yccp = plane(xc), plane(yc), plane(ycc)
        sc3 = lambda a, b, c: torch.stack([torch.as_tensor(t, dtype=torch.float32).reshape(-1)[:1].expand(B) for t in (a, b, c)],  # noqa: E731
                                          1).to(**f).contiguous()
        paras, nd = sc3(raq, fkt, fkp), sc3(raq_nd, fkt_nd, fkp_nd)
        scaler = (torch.exp(paras[:, 0] / 10 * 1.80167667 + torch.log(paras[:, 1]) * 0.4330392
                            + torch.log(paras[:, 2]) * -0.46052953) * 5).contiguous()
        raq_s = paras[:, 0].contiguous()
        x = {0: T_prev.to(**f).reshape(B, 1, H, W).contiguous()}
        dts = {}
        st = self._g if (self._g is not None and self._g["shape"] == (B, H, W)) else None
        if st is None:
            st = dict(shape=(B, H, W), inp=torch.empty((B, 7, H, W), **f), T=torch.empty((B, 1, H, W), **f),
                      Tn=torch.empty((B, H, W), **f), dt=torch.empty(1, **f), graph=None, out=None)
            self._g = st
            for k, t in dict(xc=xcp, yc=ycp, ycc=yccp, paras=paras, nd=nd, raq=raq_s, scaler=scaler).items():
                st[k] = torch.empty_like(t)
        # the captured step reads these buffers by address: refresh their CONTENTS, never the tensors
        for k, t in dict(xc=xcp, yc=ycp, ycc=yccp, paras=paras, nd=nd, raq=raq_s, scaler=scaler).items():
            st[k].copy_(t)

        def one_step():
            L.call("mc_ts_build_input", L.ptr(st["T"]), L.ptr(st["xc"]), L.ptr(st["yc"]), L.ptr(st["ycc"]), L.ptr(st["paras"]),
                   L.ptr(st["nd"]), B, H, W, L.ptr(st["inp"]), L.stream())
            uu, vv, pp = self.stokes(st["inp"])
            uu = uu.reshape(B, H, W).contiguous()
            vv = vv.reshape(B, H, W).contiguous()
            self.ad.step(uu, vv, st["T"].view(B, H, W), st["xc"], st["yc"], raq_scalar=st["raq"], vel_scale=st["scaler"],
                         out=st["Tn"], dt_out=st["dt"])
            st["out"] = (uu, vv, pp)

        u = v = p = None
        for i in range(1, self.ts + 1):
            st["T"].copy_(x[i - 1])
            if self.use_graph:
                if st["graph"] is None:
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        one_step()                                  # warm-up: allocates every buffer outside the capture
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize(dev)
                    st["graph"] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(st["graph"]):
                        one_step()
                st["graph"].replay()
            else:
                one_step()
            u, v, p = st["out"]
            x[i] = st["Tn"].clone().view(B, 1, H, W)
            dts[i] = st["dt"].clone()
        V = torch.clip(torch.exp(-torch.log(paras[:, 1]).view(B, 1, 1, 1) * x[self.ts - 1]
                                 + torch.log(paras[:, 2]).view(B, 1, 1, 1) * (1.0 - yccp)), 1e-8, 1.0) if self.ts >= 1 else None
        sv = scaler.view(B, 1, 1, 1)
        return (x, dts, u.view(B, 1, H, W) * sv, v.view(B, 1, H, W) * sv,
                p.reshape(B, 1, H, W) if (p is not None and self.p_pred) else p, V)


    def _forward_unet(self, T_prev, ycc, raq_nd, fkt_nd, fkp_nd, raq, fkt, fkp, xc, yc, u_prev, v_prev, dt):
        """The 'unet' branch (reference :411-446): the Stokes net predicts the next temperature itself — ts times
        { 10-channel input (xc/4, yc/4, dt, normalised parameters, log10(clip(eta))/8, T, u_prev, v_prev) -> Unet -> wall rows
        and side columns of T }.  As in the reference, u_prev / v_prev / dt stay the caller's for every step, no advection net
        runs, p is None and the returned V is the LAST input's viscosity channel (already log-scaled).  The reference
        hard-codes a 1 x 1 x 128 x 506 view; any [B, 1, H, W] works here."""
        if u_prev is None or v_prev is None or dt is None:
            raise ValueError("TS(net='unet') needs u_prev, v_prev and dt")
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        f = dict(dtype=torch.float32, device=dev)
        B, _, H, W = T_prev.shape
        plane = lambda t: torch.as_tensor(t).to(**f).reshape(-1, H, W)[0].contiguous()  # noqa: E731
        full = lambda t: torch.as_tensor(t).to(**f).expand(B, 1, H, W).reshape(B, H, W).contiguous()  # noqa: E731
        xcp, ycp, yccp = plane(xc), plane(yc), plane(ycc)
        sc3 = lambda a, b, c: torch.stack([torch.as_tensor(t, dtype=torch.float32).reshape(-1)[:1].expand(B) for t in (a, b, c)],  # noqa: E731
                                          1).to(**f).contiguous()
        paras, nd = sc3(raq, fkt, fkp), sc3(raq_nd, fkt_nd, fkp_nd)
        dtp, up, vp = full(dt), full(u_prev), full(v_prev)
        inp = torch.empty((B, 10, H, W), **f)
        x = {0: T_prev.to(**f).reshape(B, 1, H, W).contiguous()}
        u = v = None
        for i in range(1, self.ts + 1):
            L.call("mc_ts_build_input_unet", L.ptr(x[i - 1]), L.ptr(xcp), L.ptr(ycp), L.ptr(yccp), L.ptr(paras), L.ptr(nd), L.ptr(dtp),
                   L.ptr(up), L.ptr(vp), B, H, W, L.ptr(inp), L.stream())
            u, v, _, Tn = self.stokes(inp)
            Tn = Tn.reshape(B, H, W).to(**f).contiguous()
            x[i] = torch.empty((B, 1, H, W), **f)
            L.call("mc_ts_wall_bc", L.ptr(Tn), B, H, W, L.ptr(x[i]), L.stream())
        V = inp[:, 6:7].clone() if self.ts >= 1 else None
        return x, {}, u.reshape(B, 1, H, W), v.reshape(B, 1, H, W), None, V


# --------------------------------------------------------------------------------------------------
# ConvAE (reference .ipynb_checkpoints/pycold-checkpoint.py:989-1115)
# --------------------------------------------------------------------------------------------------
class ConvAE(nn.Module, HipNetMixin):
    def __init__(self, levels: int, c_i: int, c_h: int, c_o: int, device=None, act_fn: str = "selu", r_p="zeros",
                 loss_type="mae", use_symm=False, dilation=1, a_bound=4.0, use_cosine=False, repeats=3,
                 use_skip=False, f=3, p_pred=True, spectral_conv=False, blurr=False):
        super().__init__()
        _check_common(act_fn, r_p, dilation, 0.0, spectral_conv, blurr)
        self.levels, self.loss_type, self.a_bound = levels, loss_type, a_bound
        self.use_cosine, self.repeats, self.use_skip, self.p_pred = use_cosine, repeats, use_skip, p_pred
        self.blurrer = None
        self.r_p = "constant" if r_p == "zeros" else r_p
        graph = convae_graph(levels, c_i, c_h, c_o, act=act_fn, r_p=r_p, use_symm=use_symm, repeats=repeats, f=f,
                             loss_type=loss_type)
        factor = 4

        def fl(cin, cout):
            return FluidLayer(int(cin), int(cout), act_fn, r_p, use_symm, dilation, f=f)

        self.conv = nn.ModuleList()
        self.gn = nn.ModuleList()
        self.pool = nn.ModuleList()
        self.unpool = nn.ModuleList()
        self.conv.append(fl(c_i, c_h))
        c = c_h
        for _ in range(levels):
            self.conv.append(nn.AvgPool2d((factor, factor), stride=factor))
            cout = c * factor
            for r in range(repeats):
                self.conv.append(fl(c if r == 0 else cout, cout))
            c *= factor
        c = int(c / factor)
        for r in range(repeats):
            self.conv.append(fl(c * factor if r == 0 else c, c))
        for _ in range(levels, 0, -1):
            self.conv.append(torch.nn.Upsample(scale_factor=factor, mode="bicubic"))
            cout = c / factor
            for r in range(repeats):
                self.conv.append(fl(c if r == 0 else cout, cout))
            c = int(c / factor)
        padding = (2, 2) if loss_type == "curl" else (1, 1)
        self.conv.append(nn.Conv2d(int(c), int(c_o), kernel_size=3, padding=padding, dilation=1, padding_mode=r_p))
        self._init_hipnet(graph)

    def forward(self, x):
        x = self._run_graph(x)
        if self.loss_type == "curl":
            # streamfunction head on the (H+2)x(W+2) output of the padding-2 final conv (:1099-1113);
            # pure slicing/differences on the small output tensor
            a = x[:, -1:] * self.a_bound
            u = dy_center(a)[..., :, 1:-1]
            v = -dx_center(a)[..., 1:-1, :]
            if self.p_pred:
                x = torch.cat((x[:, :-2, 1:-1, 1:-1], u, v, x[:, -2:-1, 1:-1, 1:-1]), dim=1)
            else:
                x = torch.cat((x[:, :-1, 1:-1, 1:-1], u, v), dim=1)
        return x
