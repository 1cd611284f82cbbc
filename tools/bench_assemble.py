#!/usr/bin/env python3
"""SURVEY 8(f) N2 measurement: on-device batch assembly (mc_assemble_adtime_batch) from an HBM-resident dataset vs the host
item builder (the mirror of ADTimeDataset.__getitem__) + pinned copy.  usage: tools/bench_assemble.py [--m 256] [--size 506 506]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbml_mantle_convection_amd.datasetio import ADTimeDataset, ResidentADTimeDataset, normalise_parameters
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=128); ap.add_argument("--size", type=int, nargs=2, default=[506, 506]); ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
H, W, M, B = a.size[0], a.size[1], a.m, a.batch
ds = ADTimeDataset.__new__(ADTimeDataset)
g = torch.Generator().manual_seed(1)
ds.x_data = [torch.rand((1, H, W), generator=g, dtype=torch.float64) for _ in range(M)]
ds.y_data = [torch.randn((3, H, W), generator=g, dtype=torch.float64) for _ in range(M)]
ds.t = [0.01 * i for i in range(M)]
ds.t_data = [torch.tensor(t, dtype=torch.float64) for t in ds.t]
par = (2.5, 1e7, 30.0)
ds.paras = [torch.tensor(par, dtype=torch.float64).view(3, 1, 1)] * M
ds.paras_nd = [torch.tensor(normalise_parameters(*par), dtype=torch.float64).view(3, 1, 1)] * M
ds.xc = torch.linspace(0, 4, W, dtype=torch.float64).view(1, 1, W).expand(1, H, W).contiguous()
ds.yc = torch.linspace(0, 1, H, dtype=torch.float64).view(1, H, 1).expand(1, H, W).contiguous()
ds.indices = [[i, i + 1] for i in range(M - 1)]
ds.indices_init = [[0, 1]]
ds.scale, ds.p_pred, ds.noise, ds.num_examples = True, True, 0.0, M - 1
rd = ResidentADTimeDataset(ds, "cuda:0")
idx = [1 + (7 * k) % (M - 2) for k in range(B)]
idx = [i for i in idx if ds.indices[i][0] % 8 != 0][:B]
out = None
x, y, sc, pa, yc = rd.assemble(idx)
out = dict(gVTp=x, uvp=y, scaler=sc, paras=pa.reshape(-1, 3))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pr = rd.pairs(idx)
e0.record()
for _ in range(20):
    rd.assemble(idx, out=out, pairs=pr)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
nb = len(idx)
byts = nb * H * W * 4 * (6 + 2 + 13)       # T x2, uv x4 read (+ xc, yc), 13 planes written
print(f"device assembly: batch {nb} at {H}x{W}: {us:.1f} us  ({byts / us / 1e3:.0f} GB/s, {nb / us * 1e6:.0f} samples/s)")
t0 = time.perf_counter()
items = [ds[i] for i in idx]
xb = torch.stack([it[0] for it in items]).float().pin_memory().to("cuda:0", non_blocking=True)
yb = torch.stack([it[1] for it in items]).float().pin_memory().to("cuda:0", non_blocking=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host items (fp64 assembly, stack, pinned copy): {dt * 1e3:.1f} ms  ({nb / dt:.0f} samples/s)")
