"""BoundaryLearnedConvolution2D — the reference's "learned padding" layer (pytorch_networks_convae.py:802-1065; SURVEY.md
§8(f) row N4) on the HIP path: nine bias-free VALID convolutions (one on the whole input, eight on the border strips of
width k + 1 / k) framed together, plus one shared bias.

Every convolution, its filter gradient and its input gradient run on the library's conv kernels (`mc_conv2d`,
`mc_conv2d_wgrad*`); the strips are cut and the frame is assembled with `mc_rect_copy`.  The main bank runs as a
zero-padded 'same' convolution straight into the output (its interior is the valid result), the eight strip results then
overwrite the frame.  As in the reference, the strip cut from the LAST rows lands in the FIRST output rows and vice versa
(:1057-1060).  Only bc_x = bc_y = 1 (output size = input size) is implemented.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib as L
from .symmetric_layers_torch import SymmetricConv2d

BANKS = ("conv", "conv_top_left", "conv_top_right", "conv_bottom_left", "conv_bottom_right", "conv_top", "conv_bottom",
         "conv_left", "conv_right")
_DT = {"fp32": (L.MC_F32, torch.float32), "bf16": (L.MC_BF16, torch.bfloat16)}


def _regions(H, W, k):
    """name -> (input sy, sx, sh, sw, output dy, dx) for bc = 1."""
    pad = k + 1 if k == 5 else k
    f = pad - k + 1
    mh, mw = H - k + 1, W - k + 1
    return f, {
        "conv_left": (0, 0, H, pad, f, 0), "conv_right": (0, W - pad, H, pad, f, f + mw),
        "conv_bottom": (H - pad, 0, pad, W, 0, f), "conv_top": (0, 0, pad, W, f + mh, f),
        "conv_bottom_left": (H - pad, 0, pad, pad, 0, 0), "conv_bottom_right": (H - pad, W - pad, pad, pad, 0, f + mw),
        "conv_top_left": (0, 0, pad, pad, f + mh, 0), "conv_top_right": (0, W - pad, pad, pad, f + mh, f + mw)}


class _Plan:
    def __init__(self, N, H, W, c_i, c_o, k, sym_h, precision, device):
        self.key = (N, H, W, precision, str(device))
        self.N, self.H, self.W, self.c_i, self.c_o, self.k = N, H, W, c_i, c_o, k
        self.mc, self.td = _DT[precision]
        self.f, self.regions = _regions(H, W, k)
        if H < 2 * self.f + 1 + (k - 1) or W < 2 * self.f + 1 + (k - 1):
            raise ValueError("input too small for the learned-padding strips")
        dev = device

        def cb8(c, h, w):
            return torch.empty((N, (c + 7) // 8, h, w, 8), dtype=self.td, device=dev)

        def conv_entry(h, w, pad):
            d = L.ConvDesc(N, h, w, c_i, 0, c_o, k, pad, L.PAD_MODES["zeros"], self.mc, sym_h, 0, 0)
            ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
            dd = L.ConvDesc(N, ho, wo, c_o, 0, c_i, k, k - 1, 0, self.mc, 0, 0, 0)
            if L.call("mc_conv_tiles", C.byref(d)) <= 0:
                raise L.MantleHipError("unsupported convolution configuration in BoundaryLearnedConvolution2D")
            return dict(desc=d, ddesc=dd, ho=ho, wo=wo,
                        bank=torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 0), dtype=torch.uint8, device=dev),
                        dbank=torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 1), dtype=torch.uint8, device=dev),
                        wpart=torch.empty(L.call("mc_wgrad_partial_bytes", C.byref(d)), dtype=torch.uint8, device=dev))

        self.X, self.Y = cb8(c_i, H, W), cb8(c_o, H, W)
        self.dY, self.dYm = cb8(c_o, H, W), cb8(c_o, H, W)
        self.main = conv_entry(H, W, self.f)
        self.dXP = cb8(c_i, H + 2 * self.f, W + 2 * self.f)
        self.dX = cb8(c_i, H, W)
        self.strips = {}
        for name, (sy, sx, sh, sw, dy, dx) in self.regions.items():
            e = conv_entry(sh, sw, 0)
            e.update(S=cb8(c_i, sh, sw), R=cb8(c_o, e["ho"], e["wo"]), dR=cb8(c_o, e["ho"], e["wo"]), dS=cb8(c_i, sh, sw),
                     reg=(sy, sx, sh, sw, dy, dx))
            self.strips[name] = e


class _LearnedConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, bias, *weights):
        L.require_cuda(x, "input")
        x = x.float().contiguous()
        N, Ci, H, W = x.shape
        p = mod._plan(N, H, W, x.device)
        st = L.stream()
        ws = {n: w.detach().float().contiguous() for n, w in zip(BANKS, weights)}
        b = bias.detach().float().reshape(-1).contiguous()
        L.call("mc_pack_nchw", L.ptr(x), N, p.c_i, Ci, H, W, 0, 0, None, p.mc, L.ptr(p.X), st)
        m = p.main
        L.call("mc_pack_weights", C.byref(m["desc"]), L.ptr(ws["conv"]), 0, L.ptr(m["bank"]), st)
        L.call("mc_conv2d", C.byref(m["desc"]), L.ptr(p.X), None, L.ptr(m["bank"]), L.ptr(b), L.ptr(p.Y), None, None, st)
        for name, e in p.strips.items():
            sy, sx, sh, sw, dy, dx = e["reg"]
            L.call("mc_rect_copy", L.ptr(p.X), H, W, sy, sx, L.ptr(e["S"]), sh, sw, 0, 0, sh, sw, N, p.c_i, 0, p.mc, st)
            L.call("mc_pack_weights", C.byref(e["desc"]), L.ptr(ws[name]), 0, L.ptr(e["bank"]), st)
            L.call("mc_conv2d", C.byref(e["desc"]), L.ptr(e["S"]), None, L.ptr(e["bank"]), L.ptr(b), L.ptr(e["R"]), None, None, st)
            L.call("mc_rect_copy", L.ptr(e["R"]), e["ho"], e["wo"], 0, 0, L.ptr(p.Y), H, W, dy, dx, e["ho"], e["wo"], N, p.c_o, 0,
                   p.mc, st)
        out = torch.empty((N, p.c_o, H, W), dtype=torch.float32, device=x.device)
        L.call("mc_unpack_nchw", L.ptr(p.Y), N, p.c_o, H, W, 0, None, p.mc, L.ptr(out), st)
        mod._version += 1
        ctx.mod, ctx.ws, ctx.version, ctx.plan = mod, ws, mod._version, p
        ctx.wdtypes = [w.dtype for w in weights]
        ctx.bshape, ctx.bdtype = tuple(bias.shape), bias.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        mod, p, ws = ctx.mod, ctx.plan, ctx.ws
        if ctx.version != mod._version:
            raise RuntimeError("backward() through a forward pass whose device activations were overwritten by a later "
                               "forward of the same module (one in-flight forward per module)")
        N, H, W, f, st = p.N, p.H, p.W, p.f, L.stream()
        gout = gout.float().contiguous()
        dev = gout.device
        L.call("mc_pack_nchw", L.ptr(gout), N, p.c_o, p.c_o, H, W, 0, 0, None, p.mc, L.ptr(p.dY), st)
        # the main bank only sees the interior of dY: its frame belongs to the eight strip banks
        L.call("mc_rect_copy", L.ptr(p.dY), H, W, 0, 0, L.ptr(p.dYm), H, W, 0, 0, H, W, N, p.c_o, 0, p.mc, st)
        for (zy, zx, zh, zw) in ((0, 0, f, W), (H - f, 0, f, W), (0, 0, H, f), (0, W - f, H, f)):
            L.call("mc_rect_copy", None, 0, 0, 0, 0, L.ptr(p.dYm), H, W, zy, zx, zh, zw, N, p.c_o, 0, p.mc, st)
        dws = {n: torch.zeros_like(ws[n]) for n in BANKS}
        db = torch.zeros(p.c_o, dtype=torch.float32, device=dev)
        m = p.main
        L.call("mc_conv2d_wgrad", C.byref(m["desc"]), L.ptr(p.X), None, L.ptr(p.dYm), L.ptr(m["wpart"]), st)
        L.call("mc_conv2d_wgrad_finalize", C.byref(m["desc"]), L.ptr(m["wpart"]), L.ptr(dws["conv"]), L.ptr(db), st)
        L.call("mc_pack_weights", C.byref(m["desc"]), L.ptr(ws["conv"]), 1, L.ptr(m["dbank"]), st)
        L.call("mc_conv2d", C.byref(m["ddesc"]), L.ptr(p.dYm), None, L.ptr(m["dbank"]), None, L.ptr(p.dXP), None, None, st)
        for name, e in p.strips.items():
            sy, sx, sh, sw, dy, dx = e["reg"]
            L.call("mc_rect_copy", L.ptr(p.dY), H, W, dy, dx, L.ptr(e["dR"]), e["ho"], e["wo"], 0, 0, e["ho"], e["wo"], N, p.c_o, 0,
                   p.mc, st)
            L.call("mc_conv2d_wgrad", C.byref(e["desc"]), L.ptr(e["S"]), None, L.ptr(e["dR"]), L.ptr(e["wpart"]), st)
            L.call("mc_conv2d_wgrad_finalize", C.byref(e["desc"]), L.ptr(e["wpart"]), L.ptr(dws[name]), L.ptr(db), st)
            L.call("mc_pack_weights", C.byref(e["desc"]), L.ptr(ws[name]), 1, L.ptr(e["dbank"]), st)
            L.call("mc_conv2d", C.byref(e["ddesc"]), L.ptr(e["dR"]), None, L.ptr(e["dbank"]), None, L.ptr(e["dS"]), None, None, st)
            L.call("mc_rect_copy", L.ptr(e["dS"]), sh, sw, 0, 0, L.ptr(p.dXP), H + 2 * f, W + 2 * f, sy + f, sx + f, sh, sw, N,
                   p.c_i, 1, p.mc, st)
        L.call("mc_rect_copy", L.ptr(p.dXP), H + 2 * f, W + 2 * f, f, f, L.ptr(p.dX), H, W, 0, 0, H, W, N, p.c_i, 0, p.mc, st)
        dx = torch.empty((N, p.c_i, H, W), dtype=torch.float32, device=dev)
        L.call("mc_unpack_nchw", L.ptr(p.dX), N, p.c_i, H, W, 0, None, p.mc, L.ptr(dx), st)
        gws = [dws[n] if dt == torch.float32 else dws[n].to(dt) for n, dt in zip(BANKS, ctx.wdtypes)]
        return (None, dx, db.view(ctx.bshape).to(ctx.bdtype), *gws)


class BoundaryLearnedConvolution2D(nn.Module):
    """Same constructor, sub-modules and state_dict keys as the reference (nine `nn.Conv2d` / `SymmetricConv2d` banks with
    bias=False and 'valid' padding, `learnable_bias` [1, c_o, 1, 1])."""

    def __init__(self, c_i, c_o, k, stride=1, use_symm=False):
        super().__init__()
        if k not in (3, 5) or stride != 1:
            raise NotImplementedError("HIP BoundaryLearnedConvolution2D supports 3x3 / 5x5 kernels, stride 1")
        self.c_i, self.c_o, self.k, self.use_symm = c_i, c_o, k, use_symm
        h_s = int(c_o / 4) if c_o > 4 else int(c_o / 2)
        self._sym_h = h_s if use_symm else 0
        for name in BANKS:
            if use_symm:
                mod = SymmetricConv2d(c_i, c_o, k, bias=False, padding="valid", symmetry={"h": h_s, "v": 0, "hv": 0})
            else:
                mod = nn.Conv2d(in_channels=c_i, out_channels=c_o, kernel_size=k, padding="valid", bias=False)
            setattr(self, name, mod)
        self.learnable_bias = nn.Parameter(torch.zeros(1, c_o, 1, 1))
        self._precision = os.environ.get("MANTLE_PRECISION", "fp32")
        self._plans, self._version = {}, 0

    def set_precision(self, precision: str):
        if precision not in _DT:
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._precision = precision
        return self

    def _plan(self, N, H, W, device):
        key = (N, H, W, self._precision, str(device))
        p = self._plans.get(key)
        if p is None:
            L.load()
            p = self._plans[key] = _Plan(N, H, W, self.c_i, self.c_o, self.k, self._sym_h, self._precision, device)
        return p

    def forward(self, x, bc_x=1, bc_y=1):
        if bc_x != 1 or bc_y != 1:
            raise NotImplementedError("bc_x / bc_y > 1 (field-growing strips of the Unet's first layer) are not implemented")
        return _LearnedConvFn.apply(self, x, self.learnable_bias, *[getattr(self, n).weight for n in BANKS])
