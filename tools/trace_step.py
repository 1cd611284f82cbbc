#!/usr/bin/env python3
"""Per-launch timing of one fused training step (eager launches, HIP events around every C-ABI call).
usage: python tools/trace_step.py [--batch 32] [--size 506] [--agg]
prints: call, shape, us, TFLOP/s (conv family) and GB/s of algorithmic bytes where known."""
import argparse
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pbml_mantle_convection_amd import _lib as L
from pbml_mantle_convection_amd.datasetio import synthetic_batch
from pbml_mantle_convection_amd.multigpu import Trainer
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=506)
ap.add_argument("--agg", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
CFG = bench.CFG
torch.manual_seed(0)
model = Unet(CFG["levels"], CFG["c_i"], CFG["c_h"], CFG["c_o"], dev, CFG["act"], CFG["r_p"], CFG["loss_type"],
             use_symm=CFG["use_symm"], repeats=CFG["repeats"], f=CFG["f"], p_pred=CFG["p_pred"])
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 9], gamma=0.5)
tr = Trainer(model, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
             loss_scale=False, loss_derivative=False, loss_type=CFG["loss_type"], lambda_mom=0.1, precision="bf16",
             use_graph=False)
H = W = a.size
data = [t.to(dev) for t in synthetic_batch(a.batch, H, W, 1234, p_pred=True, device="cpu")]
gVTp, uvp, scaler, paras, yc = data
for _ in range(2):
    tr.train_step(gVTp, uvp, yc, paras, scaler)
torch.cuda.synchronize()

rec = []
orig = L.call


def describe(fn, args):
    """(shape string, flops, algorithmic bytes)"""
    es = 2
    for x in args:
        d = getattr(x, "_obj", None)
        if isinstance(d, L.ConvDesc):
            ho, wo = d.h + 2 * d.pad - d.k + 1, d.w + 2 * d.pad - d.k + 1
            cin = d.c_in0 + d.c_in1
            fl = 2.0 * d.n * cin * d.c_out * d.k * d.k * ho * wo
            by = d.n * es * (cin * d.h * d.w + d.c_out * ho * wo)
            return f"{cin}->{d.c_out} {d.h}x{d.w} p{d.pad}", fl, by
    if fn.startswith("mc_gn_act"):
        _, n, c, h, w = args[:5]
        mult = {"mc_gn_act_fwd": 2, "mc_gn_act_bwd_reduce": 2, "mc_gn_act_bwd_apply": 3}.get(fn, 0)
        return f"C{c} {h}x{w}", 0.0, mult * n * c * h * w * es
    if fn.startswith("mc_bicubic"):
        return " ".join(str(v) for v in args[1:7] if isinstance(v, int)), 0.0, 0
    return "", 0.0, 0


def traced(fn, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(fn, *args)
    e1.record()
    rec.append((fn, describe(fn, args), e0, e1))
    return r


L.call = traced
import pbml_mantle_convection_amd.engine as E, pbml_mantle_convection_amd.losses as LS, pbml_mantle_convection_amd.multigpu as MG
tr.train_step(gVTp, uvp, yc, paras, scaler)
torch.cuda.synchronize()
L.call = orig
tot = 0.0
agg = defaultdict(lambda: [0.0, 0])
for fn, (shape, fl, by), e0, e1 in rec:
    us = 1e3 * e0.elapsed_time(e1)
    tot += us
    agg[(fn, shape)][0] += us
    agg[(fn, shape)][1] += 1
    if not a.agg:
        print(f"{fn:32s} {shape:28s} {us:8.1f} us  {fl / us / 1e6 if fl else 0:7.1f} TF/s  {by / us / 1e3 if by else 0:7.0f} GB/s")
if a.agg:
    for (fn, shape), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{fn:32s} {shape:28s} x{n:3d} {us:9.1f} us total {us / n:8.1f} avg")
print(f"total {tot / 1e3:.2f} ms over {len(rec)} calls (event-to-event, includes launch gaps)")
