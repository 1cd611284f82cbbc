"""ConvAE bf16-mode diagnostics against the fp64 golden: output / global-gradient / worst-parameter relative errors."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE
DEV = "cuda:0"
def dev(a): return torch.from_numpy(np.asarray(a)).float().to(DEV).contiguous()
for tag in ("mae", "curl"):
    g = np.load(f"tests/golden/g5_convae_{tag}.npz")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    for prec in ("fp32", "bf16"):
        m = ConvAE(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", str(g["r_p"]), str(g["loss_type"]),
                   use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
        m.load_state_dict({k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")})
        m = m.to(DEV).set_precision(prec)
        y = m(dev(g["x"]))
        (y * dev(g["ct"])).sum().backward()
        yr = torch.from_numpy(g["y"]).double()
        oe = float((y.detach().double().cpu() - yr).norm() / yr.norm())
        num = den = 0.0
        worst = ("", 0.0)
        for n, p in m.named_parameters():
            r = torch.from_numpy(g["grad/" + n]).double()
            d = p.grad.detach().double().cpu() - r
            num += float(d.norm() ** 2); den += float(r.norm() ** 2)
            if float(r.abs().max()) > 1e-6:
                e = float(d.norm() / r.norm())
                if e > worst[1]: worst = (n, e, float(r.norm()))
        print(tag, prec, "out rel %.3e  grad global rel %.3e  worst %s" % (oe, (num / den) ** 0.5, worst))
