"""Shared machinery of the nn.Module front ends: parameter views, the autograd bridge to the
HIP engine and the flat parameter/gradient store used by the fused trainer."""
from __future__ import annotations

import os
from typing import Dict, List

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine, NetGraph

DEFAULT_PRECISION = os.environ.get("MANTLE_PRECISION", "fp32")


class _NetFunction(torch.autograd.Function):
    """y = engine.forward(x); backward runs the HIP backward and hands parameter gradients to autograd."""

    @staticmethod
    def forward(ctx, mod, x, *plist):
        eng = mod.engine()
        params = mod._param_dict(plist)
        out = eng.forward(x, params, getattr(mod, "_chan_scale", None))
        mod._fwd_version += 1
        ctx.mod, ctx.params, ctx.version = mod, params, mod._fwd_version
        ctx.dtypes = [p.dtype for p in plist]
        return out

    @staticmethod
    def backward(ctx, gout):
        mod = ctx.mod
        if ctx.version != mod._fwd_version:
            raise RuntimeError("backward() through a forward pass whose device activations were overwritten by a later "
                               "forward of the same module (one in-flight forward per module)")
        names = mod._pnames
        sizes = [ctx.params[n].numel() for n in names]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=gout.device)
        grads, off = {}, 0
        for n, s in zip(names, sizes):
            grads[n] = flat[off:off + s].view(ctx.params[n].shape)
            off += s
        mod.engine().backward(gout, ctx.params, grads)
        outs = [grads[n] if dt == torch.float32 else grads[n].to(dt) for n, dt in zip(names, ctx.dtypes)]
        return (None, None, *outs)


class HipNetMixin:
    """Mixed into nn.Module subclasses whose forward is one NetGraph on the HIP engine."""

    def _init_hipnet(self, graph: NetGraph, precision: str = None):
        self._graph = graph
        self._precision = precision or getattr(self, "_precision", None) or DEFAULT_PRECISION   # (keeps an earlier set_precision)
        self._engines: Dict[str, Engine] = {}
        self._fwd_version = 0
        self._pnames: List[str] = []

    @property
    def precision(self):
        return self._precision

    def set_precision(self, precision: str):
        """'fp32' (parity gate: f32 storage and arithmetic), 'bf16' (bf16 storage, f32 accumulate), 'mixed' (f16 tensors in
        the forward pass, bf16 gradient tensors: what the momentum residual needs, engine.py)."""
        if precision not in ("fp32", "bf16", "mixed"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'mixed'")
        self._precision = precision
        return self

    def engine(self) -> Engine:
        e = self._engines.get(self._precision)
        if e is None:
            e = self._engines[self._precision] = Engine(self._graph, self._precision)
        return e

    def _named_hip_params(self):
        return list(self.named_parameters())

    def _param_dict(self, plist):
        out = {}
        for n, p in zip(self._pnames, plist):
            d = p.detach()
            if d.dtype != torch.float32:
                d = d.float()
            out[n] = d.contiguous()
        return out

    def _run_graph(self, x: torch.Tensor) -> torch.Tensor:
        L.require_cuda(x, "input")
        named = self._named_hip_params()
        self._pnames = [n for n, _ in named]
        for n, p in named:
            L.require_cuda(p, f"parameter {n}")
        return _NetFunction.apply(self, x, *[p for _, p in named])


class FlatParams:
    """One flat f32 device buffer for all parameters and one for their gradients; the module's
    nn.Parameters become views, so reference-style code (state_dict, torch optimizers) keeps
    working while the fused Adam kernel and the single RCCL all-reduce see flat memory."""

    def __init__(self, module: nn.Module, device):
        named = [(n, p) for n, p in module.named_parameters()]
        self.names = [n for n, _ in named]
        self.shapes = [tuple(p.shape) for _, p in named]
        self.offsets = []
        total = 0
        for _, p in named:
            total = (total + 3) // 4 * 4      # 16-byte aligned slices
            self.offsets.append(total)
            total += p.numel()
        total = (total + 3) // 4 * 4
        self.numel = total
        self.param = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.module = module
        with torch.no_grad():
            for (n, p), off, shp in zip(named, self.offsets, self.shapes):
                view = self.param[off:off + p.numel()].view(shp)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.grad[off:off + p.numel()].view(shp)

        self._views = {}

    def views(self, flat):
        """name -> view of `flat` (cached per buffer)."""
        key = flat.data_ptr()
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = {n: flat[off:off + int(torch.Size(s).numel())].view(s)
                                    for n, off, s in zip(self.names, self.offsets, self.shapes)}
        return v

    def bound(self) -> bool:
        """True while the module's parameters are still views of the flat buffer."""
        p = next(self.module.parameters())
        return p.data_ptr() == self.param.data_ptr() + 4 * self.offsets[0]
