"""Loss tuple of the FIRST CFG-3 step (same weights, same batch) in every precision mode, against the fp32 mode: how far
each 16-bit mode's momentum residual (slot 6) and total (slot 0) are from the f32 execution.  usage: [B] [H W]"""
import sys

import torch

sys.path.insert(0, ".")
from pbml_mantle_convection_amd.datasetio import synthetic_batch  # noqa: E402
from pbml_mantle_convection_amd.multigpu import Trainer  # noqa: E402
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (506, 506)
dev = torch.device("cuda", 0)
batch = [t.to(dev) for t in synthetic_batch(B, H, W, 1234, p_pred=True, device="cpu")]
gVTp, uvp, scaler, paras, yc = batch
ref = None
for prec in ("fp32", "mixed", "bf16"):
    torch.manual_seed(0)
    m = Unet(5, 10, 16, 4, dev, "gelu", "reflect", "mass", use_symm=True, repeats=3, f=5, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 9], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                 lambda_mom=1e-6, precision=prec if prec != "bf16" else None)
    if prec == "bf16":
        m.set_precision("bf16")
    outs = []
    for step in range(3):
        outs.append(tr.train_step(gVTp, uvp, yc, paras, scaler)[:7].tolist())
    if ref is None:
        ref = outs
    for step, o in enumerate(outs):
        print(f"{prec:6s} step {step}: total {o[0]:.6f} ({o[0] / ref[step][0]:.4f} x)  data {o[1]:.5f} {o[2]:.5f} {o[3]:.5f} {o[4]:.5f}  "
              f"div {o[5]:.5f} ({o[5] / ref[step][5]:.4f} x)  mom {o[6]:.5e} ({o[6] / ref[step][6]:.4f} x)", flush=True)
    del tr, m
    torch.cuda.empty_cache()
