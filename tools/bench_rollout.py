#!/usr/bin/env python3
"""SURVEY 8(f) N3 measurement: latency of one inference rollout step (input builder -> NewFluidNet -> ADNet) of the deployed
configuration (levels 5, c_h 16, repeats 6, k 5, 128 x 506, batch 1), eager launches vs HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pbml_mantle_convection_amd.pytorch_networks_convae import ADNet, NewFluidNet, TS
dev = torch.device("cuda:0")
H, W = 128, 506
torch.manual_seed(0)
for prec in ("fp32", "bf16"):
    m = NewFluidNet(5, 7, 16, 3, dev, "gelu", "zeros", "mae", use_symm=True, repeats=6, f=5, p_pred=True).to(dev).set_precision(prec)
    xs = np.concatenate(([0.0], (np.arange(W - 2) + 0.5) * 4.0 / (W - 2), [4.0]))
    ys = np.concatenate(([0.0], (np.arange(H - 2) + 0.5) * 1.0 / (H - 2), [1.0]))
    xc = torch.from_numpy(np.broadcast_to(xs[None, :], (H, W)).copy()).view(1, 1, H, W)
    yc = torch.from_numpy(np.broadcast_to(ys[:, None], (H, W)).copy()).view(1, 1, H, W)
    T0 = (1.0 - yc + 0.01 * torch.randn(1, 1, H, W)).clamp(0, 1)
    raq, fkt, fkp = (torch.tensor(v) for v in (2.5, 1e7, 30.0))
    nd = [torch.tensor(v).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]
    for use_graph in (False, True):
        ts = TS(m, ADNet(dev), dev, ts=20, net="newfluidnet", use_graph=use_graph)
        ts(T0, None, None, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc)        # warm-up / capture
        torch.cuda.synchronize()
        best = []
        for _ in range(5):                      # the first replays of a fresh graph pay its upload: report the median
            t0 = time.perf_counter()
            x, dts, *_ = ts(T0, None, None, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / 20 * 1e3)
        ms = sorted(best)[len(best) // 2]
        print(f"{prec} {'HIP graph' if use_graph else 'eager    '}: {ms:.3f} ms per rollout step (batch 1, 128x506), T range "
              f"[{float(x[20].min()):.3f}, {float(x[20].max()):.3f}]")
