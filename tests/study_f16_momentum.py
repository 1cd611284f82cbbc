"""CPU study (not a test): momentum residual of the forward pass when the full-resolution level is stored in fp16 (one
16-bit tensor, 11 significant bits) instead of bf16 (8 bits) or bf16 (hi, lo) pairs (~16 bits).  fp64 arithmetic, every
storage rounding of the device emulated by the oracle (tests/study_bf16_momentum.py for the per-site study).
usage: python tests/study_f16_momentum.py [H=W]"""
import sys

import torch

sys.path.insert(0, ".")
from oracle import ref_cpu as O  # noqa: E402
from pbml_mantle_convection_amd.datasetio import synthetic_batch  # noqa: E402
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
m = Unet(5, 10, 16, 4, torch.device("cpu"), "gelu", "reflect", "mass", use_symm=True, repeats=3, f=5, p_pred=True)
sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
gVTp, uvp, scaler, paras, yc = [t.double() for t in synthetic_batch(1, H, W, 13, p_pred=True)]
x = O.build_unet_input(gVTp)[:, :10]
bf = lambda t: t.to(torch.float32).to(torch.bfloat16).to(t.dtype)  # noqa: E731
hf = lambda t: t.to(torch.float32).to(torch.float16).to(t.dtype)  # noqa: E731


def split(t):
    hi = bf(t)
    return hi + bf(t - hi)


def mom(y):
    u, v, p, T = y[:, 0], y[:, 1], y[:, 2], y[:, 3]
    Rx, Ry = O.momentum_residual(u, v, p, T, yc, paras, scaler)
    return float(Rx.abs().mean() + Ry.abs().mean())


def variant(l0, deep, wl0, wdeep):
    def q(t):
        if t.dim() == 4 and t.shape[-1] <= 5:
            return wl0(t) if t.shape[0] <= 16 else wdeep(t)
        return l0(t) if t.shape[-2] == H else deep(t)
    return O.unet_features_quantised(sd, x, 5, 3, "gelu", "reflect", True, q, device_gelu=False)


ident = lambda t: t  # noqa: E731
ref = mom(O.unet_features(sd, x, 5, 3, "gelu", "reflect", True))
print(f"{H}x{W}: exact {ref:.4e}")
for name, args in [("all bf16", (bf, bf, bf, bf)),
                   ("level 0 bf16 (hi, lo), rest bf16 [round 2]", (split, bf, bf, bf)),
                   ("level 0 fp16 (+ its banks), rest bf16", (hf, bf, hf, bf)),
                   ("level 0 fp16, level-0 banks bf16", (hf, bf, bf, bf)),
                   ("all forward tensors fp16", (hf, hf, hf, hf)),
                   ("level 0 exact, rest bf16", (ident, bf, bf, bf))]:
    v = mom(variant(*args))
    print(f"  {name:46s} {v:.4e}  ({v / ref:.3f} x)", flush=True)
