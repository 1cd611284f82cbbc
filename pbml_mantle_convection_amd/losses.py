"""Fused training loss on MI355X: Trainer.loss_fn / get_loss arithmetic (reference
multigpu.py:122-134, 250-305) plus the build-defined Stokes momentum residual, forward AND
backward in one pass over the fields (libmantle_hip: mc_loss_* / mc_momentum_* / mc_curl_head_*).

`StokesLoss.evaluate` works on raw device buffers (used by the fused trainer, graph-capturable);
`StokesLoss.__call__` is the autograd-visible form used by Trainer.get_loss.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L

FUSED_LOSS = os.environ.get("MANTLE_FUSED_LOSS", "1") != "0"      # mc_loss_fused instead of mc_loss_fwd_bwd + mc_momentum_*

OUT_NAMES = ("loss", "loss_true_u", "loss_true_v", "loss_p", "loss_T", "mass", "momentum", "_")


class StokesLoss:
    def __init__(self, p_pred: bool, loss_type: str, loss_scale: bool = False, loss_derivative: bool = False,
                 norm: str = "l1", lambda_mom: float = 0.0, inv_h: float = 126.0, ra: float = 1.0,
                 a_bound: float = 10.0, t_grad: bool = True, has_T: bool = True):
        if loss_type not in L.LOSS_TYPES:
            raise ValueError(f"loss_type must be one of {list(L.LOSS_TYPES)}")
        if norm not in ("l1", "l2"):
            raise ValueError("norm must be 'l1' (reference) or 'l2'")
        self.p_pred, self.loss_type = bool(p_pred), loss_type
        self.loss_scale, self.loss_derivative = bool(loss_scale), bool(loss_derivative)
        self.norm, self.lambda_mom, self.inv_h, self.ra = norm, float(lambda_mom), float(inv_h), float(ra)
        self.a_bound, self.t_grad = float(a_bound), bool(t_grad)
        # has_T = False: FluidNet family (reference get_loss, multigpu.py:138-195): outputs (u, v[, p]) / (streamfunction[, p]),
        # truth (u, v[, p]), no temperature term, scaled pressure loss
        self.has_T = bool(has_T)
        if not self.has_T and self.lambda_mom != 0.0:
            raise ValueError("the momentum residual needs the temperature output (Unet); use lambda_mom = 0 with has_T = False")
        self._shape = None

    # ------------------------------------------------------------------
    def freeze(self):
        """Pin the buffers: a captured HIP graph holds their device pointers (see Engine.freeze)."""
        self._frozen = True

    def _alloc(self, N, Cc, H, W, dev):
        if self._shape == (N, Cc, H, W, str(dev)):
            return
        if getattr(self, "_frozen", False):
            raise RuntimeError(f"this loss object is pinned to shape {self._shape[:4]} by a captured HIP graph; got {(N, Cc, H, W)}")
        f32 = dict(dtype=torch.float32, device=dev)
        self.sums = torch.zeros(L.LOSS_SLOTS, dtype=torch.float64, device=dev)
        self.out8 = torch.zeros(8, **f32)
        self.mm = torch.zeros((N, 3, 2), **f32)
        self.gy = torch.zeros((N, Cc, H, W), **f32)
        if self.loss_type == "curl":
            self.cu, self.cv, self.cT = (torch.empty((N, H, W), **f32) for _ in range(3))
            self.gcu, self.gcv, self.gcT = (torch.empty((N, H, W), **f32) for _ in range(3))
            self.curl_ws = torch.empty(2 * N * (H - 2) * (W - 2), **f32)
        self.gsum_blocks, self.gsum_part = 0, None
        if self.fusable() and Cc <= 4:
            # per-block spatial sums of the gradient planes (the adjoint of the network's mean subtraction takes its means
            # from them: Engine.backward(gsum=...))
            self.gsum_blocks = int(L.call("mc_loss_fused_blocks", N, H, W))
            self.gsum_part = torch.zeros((N, self.gsum_blocks, 4), **f32)
        if self.lambda_mom != 0.0 and not self.fusable():         # (the one-launch form keeps S_x, S_y and eta in LDS)
            self.sx = torch.empty((N, H, W), **f32)
            self.sy = torch.empty((N, H, W), **f32)
            self.eta = torch.empty((N, H, W), **f32)
        self._shape = (N, Cc, H, W, str(dev))

    def channels_needed(self):
        t = 1 if self.has_T else 0
        if self.loss_type == "curl":
            return 1 + t + (1 if self.p_pred else 0)
        return 2 + t + (1 if self.p_pred else 0)

    def gradient_sums(self):
        """(per-block sums [N, blocks, 4], blocks) of the gradient planes written by the last evaluate(), or None."""
        if getattr(self, "gsum_part", None) is None or not self.fusable():
            return None
        return self.gsum_part, self.gsum_blocks

    def fusable(self):
        """One-launch form (mc_loss_fused): the Unet branch without the curl head."""
        return FUSED_LOSS and self.has_T and self.loss_type != "curl"

    def evaluate(self, y: Optional[torch.Tensor], uvp: torch.Tensor, yc: Optional[torch.Tensor] = None,
                 paras: Optional[torch.Tensor] = None, scaler: Optional[torch.Tensor] = None, cb8=None):
        """y: network output [N, C, H, W] f32 (Unet.features / ConvAE output); uvp: truth
        [N, 3|4, H, W] f32.  Returns (out8, gy): out8 = (loss, true_u, true_v, loss_p, loss_T, mass,
        momentum, 0) on device; gy = d(loss)/d(y).
        cb8 (only with `fusable()`): (buf, mean, crop, (N, C, H, W)) -- the last convolution's f32 output in the CB8 layout
        [N][ceil(C / 8)][H][W + 2 crop][8] and its per-(sample, channel) spatial means; y is not read then (may be None)."""
        L.require_cuda(uvp, "uvp")
        if cb8 is not None:
            if not self.fusable():
                raise RuntimeError("the CB8 form needs the one-launch loss (Unet branch without the curl head)")
            N, Cc, H, W = cb8[3]
            dev = cb8[0].device
        else:
            L.require_cuda(y, "network output")
            if y.dtype != torch.float32 or not y.is_contiguous():
                raise RuntimeError("y must be contiguous f32")
            N, Cc, H, W = y.shape
            dev = y.device
        if uvp.dtype != torch.float32 or not uvp.is_contiguous():
            uvp = uvp.float().contiguous()
        if Cc < self.channels_needed():
            raise ValueError(f"network output has {Cc} channels, loss needs {self.channels_needed()}")
        ct = (3 if self.p_pred else 2) + (1 if self.has_T else 0)
        if tuple(uvp.shape) != (N, ct, H, W):
            raise ValueError(f"uvp must be [{N},{ct},{H},{W}], got {tuple(uvp.shape)}")
        self._alloc(N, Cc, H, W, dev)
        st = L.stream()
        HW = H * W
        d = L.LossDesc(N, H, W, int(self.p_pred), L.LOSS_TYPES[self.loss_type], int(self.loss_scale),
                       int(self.loss_derivative), int(self.norm == "l2"), self.lambda_mom, self.inv_h, self.ra,
                       int(self.t_grad) if self.has_T else -1)
        self.sums.zero_()
        if self.loss_scale:
            L.call("mc_loss_minmax", L.ptr(uvp), N, ct, H, W, L.ptr(self.mm), st)
        gb = self.gy.data_ptr()
        if self.fusable():
            # channels u, v, T, p  (:2026-2036); every term in one launch
            gu, gv, gT = gb, gb + 4 * HW, gb + 4 * 2 * HW
            gp = gb + 4 * 3 * HW if self.p_pred else None
            if Cc > ct:
                self.gy.zero_()
            mom = self.lambda_mom != 0.0
            if mom:
                if yc is None or paras is None or scaler is None:
                    raise ValueError("the momentum term needs yc [H,W], paras [N,3] and scaler [N]")
                yc = yc.reshape(-1, H, W)[0].float().contiguous()            # ONE depth grid for the whole batch (see below)
                paras = paras.reshape(N, 3).float().contiguous()
                scaler = scaler.reshape(N).float().contiguous()
            args = (L.ptr(uvp), L.ptr(self.mm), L.ptr(yc) if mom else None, L.ptr(paras) if mom else None,
                    L.ptr(scaler) if mom else None, L.ptr(self.sums), gu, gv, gp, gT, Cc * HW, Cc * HW, L.ptr(self.gsum_part), st)
            if cb8 is not None:
                buf, mean, crop = cb8[:3]
                L.call("mc_loss_fused", C.byref(d), None, None, None, None, 0, 0, L.ptr(buf), W + 2 * crop, crop, L.ptr(mean), Cc, *args)
            else:
                yb = y.data_ptr()
                L.call("mc_loss_fused", C.byref(d), yb, yb + 4 * HW, yb + 4 * 3 * HW if self.p_pred else None, yb + 4 * 2 * HW,
                       Cc * HW, Cc * HW, None, 0, 0, None, 0, *args)
            L.call("mc_loss_finalize", C.byref(d), L.ptr(self.sums), L.ptr(self.out8), st)
            return self.out8, self.gy
        yb = y.data_ptr()
        if self.loss_type == "curl" and not self.has_T:
            # NewFluidNet: channel 0 = streamfunction, 1 = p  (reference pytorch_networks_convae.py:1360-1369)
            self.gy.zero_()
            L.call("mc_curl_head_fwd", yb, None, N, H, W, Cc * HW, self.a_bound, 0.0, 0.0, L.ptr(self.cu), L.ptr(self.cv),
                   None, st)
            u, v, T, pbs = self.cu.data_ptr(), self.cv.data_ptr(), None, HW
            gu, gv, gT = self.gcu.data_ptr(), self.gcv.data_ptr(), None
            p = yb + 4 * 1 * HW if self.p_pred else None
            gp = gb + 4 * 1 * HW if self.p_pred else None
        elif self.loss_type == "curl":
            # channel 0 = streamfunction, 1 = T, 2 = p  (reference pytorch_networks_convae.py:2038-2049)
            self.gy.zero_()
            L.call("mc_curl_head_fwd", yb, yb + 4 * HW, N, H, W, Cc * HW, self.a_bound, 0.0, 1.5, L.ptr(self.cu),
                   L.ptr(self.cv), L.ptr(self.cT), st)
            u, v, T, pbs = self.cu.data_ptr(), self.cv.data_ptr(), self.cT.data_ptr(), HW
            gu, gv, gT = self.gcu.data_ptr(), self.gcv.data_ptr(), self.gcT.data_ptr()
            p = yb + 4 * 2 * HW if self.p_pred else None
            gp = gb + 4 * 2 * HW if self.p_pred else None
        elif not self.has_T:
            # NewFluidNet 'mae' / 'mass': channels u, v, p  (:1348-1358)
            u, v, T, pbs = yb, yb + 4 * HW, None, Cc * HW
            gu, gv, gT = gb, gb + 4 * HW, None
            p = yb + 4 * 2 * HW if self.p_pred else None
            gp = gb + 4 * 2 * HW if self.p_pred else None
            if Cc > ct:
                self.gy.zero_()
        else:
            # channels u, v, T, p  (:2026-2036)
            u, v, T, pbs = yb, yb + 4 * HW, yb + 4 * 2 * HW, Cc * HW
            gu, gv, gT = gb, gb + 4 * HW, gb + 4 * 2 * HW
            p = yb + 4 * 3 * HW if self.p_pred else None
            gp = gb + 4 * 3 * HW if self.p_pred else None
            if Cc > ct:
                self.gy.zero_()
        ppbs = Cc * HW
        L.call("mc_loss_fwd_bwd", C.byref(d), u, v, p, T, pbs, ppbs, L.ptr(uvp), L.ptr(self.mm), L.ptr(self.sums), gu,
               gv, gp, gT, st)
        if self.lambda_mom != 0.0:
            if yc is None or paras is None or scaler is None:
                raise ValueError("the momentum term needs yc [H,W], paras [N,3] and scaler [N]")
            # ONE depth grid for the whole batch (all samples of a data set share the mesh; a loader delivers it per sample):
            # Trainer checks once, outside the captured step, that the rows are equal (Trainer._check_single_mesh)
            yc = yc.reshape(-1, H, W)[0].float().contiguous()
            paras = paras.reshape(N, 3).float().contiguous()
            scaler = scaler.reshape(N).float().contiguous()
            L.call("mc_momentum_residual", C.byref(d), u, v, p, T, pbs, ppbs, L.ptr(yc), L.ptr(paras), L.ptr(scaler),
                   L.ptr(self.sums), L.ptr(self.sx), L.ptr(self.sy), L.ptr(self.eta), st)
            L.call("mc_momentum_adjoint", C.byref(d), T, pbs, ppbs, L.ptr(self.eta), L.ptr(paras), L.ptr(scaler),
                   L.ptr(self.sx), L.ptr(self.sy), gu, gv, gp, gT, st)
        if self.loss_type == "curl" and not self.has_T:
            L.call("mc_curl_head_bwd", gu, gv, None, None, N, H, W, self.a_bound, 0.0, 0.0, gb, None, Cc * HW, Cc * HW,
                   L.ptr(self.curl_ws), st)
        elif self.loss_type == "curl":
            L.call("mc_curl_head_bwd", gu, gv, gT, yb + 4 * HW, N, H, W, self.a_bound, 0.0, 1.5, gb, gb + 4 * HW,
                   Cc * HW, Cc * HW, L.ptr(self.curl_ws), st)
        L.call("mc_loss_finalize", C.byref(d), L.ptr(self.sums), L.ptr(self.out8), st)
        return self.out8, self.gy

    def __call__(self, y, uvp, yc=None, paras=None, scaler=None):
        """Autograd-visible: returns out8 whose element 0 back-propagates d(loss)/dy into y."""
        return _LossFn.apply(self, y, uvp, yc, paras, scaler)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss_obj, y, uvp, yc, paras, scaler):
        out8, gy = loss_obj.evaluate(y.contiguous(), uvp, yc, paras, scaler)
        ctx.gy = gy.clone()
        return out8.clone()

    @staticmethod
    def backward(ctx, gout):
        return None, ctx.gy * gout[0], None, None, None, None
