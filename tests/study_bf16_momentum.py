"""CPU study (not a test): which bf16 storage roundings make the momentum residual of the bf16 mode noisy?
Emulates the device's storage rounding site by site with the oracle (fp64 arithmetic) and prints mean |R| per variant."""
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


def mom(y):
    u, v, p, T = y[:, 0], y[:, 1], y[:, 2], y[:, 3]
    Rx, Ry = O.momentum_residual(u, v, p, T, yc, paras, scaler)
    return float(Rx.abs().mean() + Ry.abs().mean())


def variant(keep_exact):
    def q(t):
        if t.dim() == 4 and t.shape[-1] <= 5:
            kind = "weight"
        elif t.shape[-2] == H:
            kind = "in0" if t.shape[1] == 10 else "act0"
        else:
            kind = "deep"
        return t if kind in keep_exact else bf(t)
    return O.unet_features_quantised(sd, x, 5, 3, "gelu", "reflect", True, q, device_gelu=False)


ref = mom(O.unet_features(sd, x, 5, 3, "gelu", "reflect", True))
print(f"{H}x{W}: exact {ref:.4e}")
for name, keep in [("all bf16", ()), ("weights exact", ("weight",)), ("input exact", ("in0",)),
                   ("level-0 activations exact", ("act0",)), ("level-0 act + input exact", ("act0", "in0")),
                   ("deep exact", ("deep",)), ("only weights bf16", ("in0", "act0", "deep")),
                   ("only input bf16", ("weight", "act0", "deep")), ("only deep bf16", ("weight", "act0", "in0")),
                   ("only level-0 act bf16", ("weight", "in0", "deep"))]:
    v = mom(variant(keep))
    print(f"  {name:32s} {v:.4e}  ({v / ref:.3f} x)")

# per-site contributions at level 0: only ONE full-resolution tensor rounded at a time
names = ["y0", "a0", "y1", "a1", "y2", "a2", "up", "y3", "a3", "y4", "a4"]
for only in range(len(names)):
    cnt = [0]

    def q(t):
        if t.dim() == 4 and t.shape[-1] <= 5:
            return t
        if t.shape[-2] == H and t.shape[1] != 10:
            i = cnt[0]
            cnt[0] += 1
            return bf(t) if i == only else t
        return t
    v = mom(O.unet_features_quantised(sd, x, 5, 3, "gelu", "reflect", True, q, device_gelu=False))
    print(f"  only {names[only]:4s} bf16  {v:.4e}  ({v / ref:.3f} x)   sites seen {cnt[0]}")
