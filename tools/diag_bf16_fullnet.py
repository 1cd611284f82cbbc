"""bf16-mode vs fp32-mode forward of the CFG-2 network at 128x506 (SURVEY.md 8d parity gate: per-field MAE <= 5e-3)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbml_mantle_convection_amd.datasetio import synthetic_batch
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
DEV = "cuda:0"
for seed in (0, 1, 2):
    torch.manual_seed(seed)
    m = Unet(5, 10, 16, 4, torch.device(DEV), "gelu", "reflect", "mae", use_symm=True, repeats=3, f=5, p_pred=True).to(DEV)
    x = synthetic_batch(4, 128, 506, 100 + seed, p_pred=True, device="cpu")[0][:, :10].to(DEV).contiguous()
    outs = {}
    for prec in ("fp32", "bf16"):
        m.set_precision(prec)
        with torch.no_grad():
            outs[prec] = [o.double().cpu() for o in m(x) if o is not None]
    for n, a, b in zip("uvpT", outs["fp32"], outs["bf16"]):
        print(f"seed {seed} {n}: MAE {float((a - b).abs().mean()):.3e}   mean|field| {float(a.abs().mean()):.3e}   max|field| {float(a.abs().max()):.3e}")
