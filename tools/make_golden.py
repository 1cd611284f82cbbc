#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden.py

/root/reference never travels to the GPU box; only the vectors written here do.  Fixtures
are data (inputs, weights, expected outputs/gradients), never reference source text.
The reference runs in its native fp64; inputs are fp32-representable so that an fp32
device path can consume exactly the same numbers.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MANTLE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)
import fields  # noqa: E402

# torchvision is not installed; the reference touches it only when blurr=True
for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.v2"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision.transforms"].v2 = sys.modules["torchvision.transforms.v2"]
os.environ.setdefault("MPLBACKEND", "Agg")

import symmetric_layers_torch as S  # noqa: E402
import pytorch_networks_convae as P  # noqa: E402
import scaler as SC  # noqa: E402
import multigpu as G  # noqa: E402

spec = importlib.util.spec_from_file_location("pycold", REF + "/.ipynb_checkpoints/pycold-checkpoint.py")
pycold = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pycold)

CPU = torch.device("cpu")
f64 = torch.float64


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(f64)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def randomize_(module, seed):
    """fp32-representable random parameters (GN affine away from the trivial 1/0)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_((torch.randn(p.shape, generator=g) / fan ** 0.5).float().to(p.dtype))
            elif "layers.1.weight" in n or n.startswith("gn.") and n.endswith("weight"):
                p.copy_((1.0 + 0.2 * torch.randn(p.shape, generator=g)).float().to(p.dtype))
            else:
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).float().to(p.dtype))


def sd_np(module, prefix="sd/"):
    # parameters are fp32-representable by construction -> store them as fp32
    return {prefix + k: v.detach().float() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad/"):
    return {prefix + k: p.grad for k, p in module.named_parameters()}


# ------------------------------------------------------------------ G1 SymmetricConv2d
def g1():
    for tag, (ci, co, k, mode, sym, hw) in {
        "a": (11, 16, 5, "reflect", {"h": 4, "v": 0, "hv": 0}, (12, 14)),
        "b": (3, 16, 3, "zeros", {"h": 4, "v": 2, "hv": 4}, (9, 10)),
        "c": (5, 8, 5, "replicate", {"h": 2, "v": 0, "hv": 0}, (7, 11)),
    }.items():
        m = S.SymmetricConv2d(ci, co, k, padding="same", padding_mode=mode, symmetry=sym).double()
        randomize_(m, 10)
        x = rnd((2, ci, *hw), 11).requires_grad_(True)
        y = m(x)
        ct = rnd(y.shape, 12)
        (y * ct).sum().backward()
        npz(f"g1{tag}_symconv", x=x, w=m.weight, b=m.bias, y=y, ct=ct, dx=x.grad, dw=m.weight.grad,
            db=m.bias.grad, meta=np.array([ci, co, k, sym["h"], sym["v"], sym["hv"]]),
            mode=np.array(mode))


# ------------------------------------------------------------------ G2 FluidLayer
def g2():
    for tag, (ci, co, k, act, mode, symm, hw) in {
        "a": (16, 16, 5, "gelu", "reflect", True, (10, 12)),
        "b": (8, 4, 3, "selu", "zeros", True, (8, 8)),
        "c": (6, 8, 5, "relu", "replicate", False, (9, 13)),
        "d": (8, 8, 3, "silu", "reflect", True, (8, 9)),
        "e": (8, 8, 3, "tanh", "reflect", False, (8, 9)),
        "f": (8, 8, 3, "elu", "zeros", False, (8, 9)),
    }.items():
        m = P.FluidLayer(ci, co, act, mode, symm, 1, f=k).double()
        randomize_(m, 20)
        x = rnd((2, ci, *hw), 21).requires_grad_(True)
        y = m(x)
        ct = rnd(y.shape, 22)
        (y * ct).sum().backward()
        npz(f"g2{tag}_fluidlayer", x=x, y=y, ct=ct, dx=x.grad, meta=np.array([ci, co, k, int(symm)]),
            act=np.array(act), mode=np.array(mode), **sd_np(m), **grads_np(m))


# ------------------------------------------------------------------ G3 resampling ops
def g3():
    x = rnd((1, 2, 31, 32), 30).requires_grad_(True)
    y = torch.nn.Upsample(size=(63, 64), mode="bicubic")(x)
    ct = rnd(y.shape, 31)
    (y * ct).sum().backward()
    npz("g3a_bicubic_size", x=x, y=y, ct=ct, dx=x.grad)
    x = rnd((2, 3, 8, 8), 32).requires_grad_(True)
    y = torch.nn.Upsample(scale_factor=4, mode="bicubic")(x)
    ct = rnd(y.shape, 33)
    (y * ct).sum().backward()
    npz("g3b_bicubic_x4", x=x, y=y, ct=ct, dx=x.grad)
    x = rnd((2, 3, 13, 17), 34).requires_grad_(True)
    y = torch.nn.AvgPool2d((2, 2), stride=2)(x)
    ct = rnd(y.shape, 35)
    (y * ct).sum().backward()
    npz("g3c_avgpool2", x=x, y=y, ct=ct, dx=x.grad)
    x = rnd((1, 2, 16, 12), 36).requires_grad_(True)
    y = torch.nn.AvgPool2d((4, 4), stride=4)(x)
    ct = rnd(y.shape, 37)
    (y * ct).sum().backward()
    npz("g3d_avgpool4", x=x, y=y, ct=ct, dx=x.grad)


# ------------------------------------------------------------------ G4 tiny Unet
def g4():
    for tag, (loss_type, p_pred, c_i, c_o, r_p, symm, act) in {
        "curl": ("curl", True, 10, 3, "reflect", True, "gelu"),
        "mae": ("mae", True, 11, 4, "reflect", True, "gelu"),
        "mass_rep": ("mass", False, 10, 3, "replicate", False, "gelu"),
        "mae_zeros": ("mae", True, 11, 4, "zeros", True, "silu"),
    }.items():
        m = P.Unet(3, c_i, 8, c_o, CPU, act, r_p, loss_type, use_symm=symm, repeats=2, f=5,
                   p_pred=p_pred).double()
        randomize_(m, 40)
        x = torch.from_numpy(fields.unet_input(2, 40, 54, 41, c_i=c_i)).requires_grad_(True)
        outs = m(x)
        names = ["u", "v", "p", "T"]
        loss = 0.0
        save = {}
        for n, o in zip(names, outs):
            if o is None:
                continue
            ct = rnd(o.shape, 42 + len(save))
            loss = loss + (o * ct).sum()
            save["out/" + n] = o
            save["ct/" + n] = ct
        loss.backward()
        # x is regenerated in the tests from fields.unet_input(2, 40, 54, 41, c_i)
        npz(f"g4_unet_{tag}", dx_sample=fields.strided_sample(x.grad.numpy(), 1021), cfg=np.array([3, c_i, 8, c_o, 2, 5, int(p_pred), int(symm)]),
            loss_type=np.array(loss_type), r_p=np.array(r_p), act=np.array(act), **save, **sd_np(m),
            **grads_np(m))


# ------------------------------------------------------------------ G5 tiny ConvAE
def g5():
    for tag, (loss_type, p_pred, c_i, c_o, r_p, symm) in {
        "mae": ("mae", True, 3, 3, "reflect", True),
        "curl": ("curl", True, 3, 3, "zeros", False),
    }.items():
        m = pycold.ConvAE(2, c_i, 4, c_o, CPU, "gelu", r_p, loss_type, use_symm=symm, repeats=2, f=3,
                          p_pred=p_pred).double()
        randomize_(m, 50)
        x = rnd((2, c_i, 32, 48), 51).requires_grad_(True)
        y = m(x)
        ct = rnd(y.shape, 52)
        (y * ct).sum().backward()
        npz(f"g5_convae_{tag}", x=x, y=y, ct=ct, dx=x.grad,
            cfg=np.array([2, c_i, 4, c_o, 2, 3, int(p_pred), int(symm)]), loss_type=np.array(loss_type),
            r_p=np.array(r_p), **sd_np(m), **grads_np(m))


# ------------------------------------------------------------------ G6 Trainer.loss_fn / get_loss
class _Stub(torch.nn.Module):
    """Stands in for model_uvp: returns fixed leaf predictions [B,H,W] (3-D, as the curl
    head emits them), so get_loss's arithmetic is exercised without the 128x506-only nets."""

    def __init__(self, u, v, p, T):
        super().__init__()
        self.u, self.v, self.p, self.T = u, v, p, T

    def forward(self, _):
        return self.u, self.v, self.p, self.T


def _trainer_ns(model, p_pred, loss_scale, loss_derivative, loss_type):
    ns = types.SimpleNamespace(
        net="unet", p_pred=p_pred, model_AD=None, loss_scale=loss_scale, loss_derivative=loss_derivative,
        roll_forward=1, loss_type=loss_type, l1=torch.nn.L1Loss(), model_uvp=model,
        dx_center_kernel=torch.tensor([-0.5, 0, 0.5]).double().view(1, 1, 1, 3),
        dy_center_kernel=torch.tensor([-0.5, 0, 0.5]).double().view(1, 1, 3, 1),
        dx_left_kernel=torch.tensor([-1.0, 1, 0]).double().view(1, 1, 1, 3),
        dy_top_kernel=torch.tensor([-1.0, 1, 0]).double().view(1, 1, 3, 1))
    ns.loss_fn = lambda a, b: G.Trainer.loss_fn(ns, a, b)
    return ns


def g6():
    B, H, W = 2, 128, 506
    rows = []
    samples = {}
    case = 0
    for p_pred in (True, False):
        for loss_type in ("mae", "mass", "curl"):
            for loss_scale in (False, True):
                for loss_derivative in (False, True):
                    seed = 600 + case
                    u = torch.from_numpy(fields.smooth_field(B, H, W, seed + 1, noise=0.01)).requires_grad_(True)
                    v = torch.from_numpy(fields.smooth_field(B, H, W, seed + 2, noise=0.01)).requires_grad_(True)
                    p = torch.from_numpy(fields.smooth_field(B, H, W, seed + 3, amp=0.5)).requires_grad_(True)
                    T = torch.from_numpy(fields.temperature_field(B, H, W, seed + 4)).requires_grad_(True)
                    truth = [fields.smooth_field(B, H, W, seed + 5), fields.smooth_field(B, H, W, seed + 6)]
                    if p_pred:
                        truth.append(fields.smooth_field(B, H, W, seed + 7, amp=0.5))
                    truth.append(fields.temperature_field(B, H, W, seed + 8))
                    uvp = torch.from_numpy(np.stack(truth, 1))
                    c_i = 11 if p_pred else 10
                    gVTp = torch.from_numpy(fields.unet_input(B, H, W, seed, c_i=c_i))
                    ns = _trainer_ns(_Stub(u, v, p if p_pred else None, T), p_pred, loss_scale,
                                     loss_derivative, loss_type)
                    out = G.Trainer.get_loss(ns, gVTp, uvp, None, None, None)
                    out[0].backward()
                    rows.append([int(p_pred), {"mae": 0, "mass": 1, "curl": 2}[loss_type], int(loss_scale),
                                 int(loss_derivative), seed] + [float(o) for o in out])
                    samples[f"du/{case}"] = fields.strided_sample(u.grad.numpy())
                    samples[f"dv/{case}"] = fields.strided_sample(v.grad.numpy())
                    samples[f"dT/{case}"] = fields.strided_sample(T.grad.numpy())
                    if p_pred:
                        samples[f"dp/{case}"] = fields.strided_sample(p.grad.numpy())
                    case += 1
    npz("g6_get_loss", table=np.array(rows), **samples)
    # loss_fn alone on a small field
    xt = rnd((3, 9, 11), 690)
    xp = rnd((3, 9, 11), 691)
    o = {}
    for ls in (False, True):
        ns = types.SimpleNamespace(loss_scale=ls, l1=torch.nn.L1Loss())
        a, b = G.Trainer.loss_fn(ns, xt, xp)
        o[f"scaled_{int(ls)}"] = a
        o[f"plain_{int(ls)}"] = b
    npz("g6b_loss_fn", x_true=xt, x_pred=xp, **o)


# ------------------------------------------------------------------ G7..G9 helpers
def g7():
    B, H, W = 2, 128, 506
    u = torch.from_numpy(fields.smooth_field(B, H, W, 700, noise=0.01))
    v = torch.from_numpy(fields.smooth_field(B, H, W, 701, noise=0.01))
    o = {}
    for bc in (False, True):
        m = P.get_mass(u, v, bc=bc)
        o[f"sample_bc{int(bc)}"] = fields.strided_sample(m.numpy())
        o[f"sum_bc{int(bc)}"] = m.sum()
        o[f"abssum_bc{int(bc)}"] = m.abs().sum()
        o[f"edge_bc{int(bc)}"] = m[0, 0, :, 0]
    npz("g7_get_mass", **o)


def g8():
    ramp = (torch.arange(42, dtype=f64).view(1, 1, 6, 7) ** 1.5 + torch.arange(7, dtype=f64) * 0.25)
    o = {"x": ramp}
    for name in ("dx_right", "dx_left", "dy_bot", "dy_top", "dx_center", "dy_center", "du_dy", "dv_dx",
                 "laplace"):
        o[name] = getattr(P, name)(ramp, CPU)
    npz("g8_fd_kernels", **o)


def g9():
    T = rnd((2, 1, 5, 6), 900).abs()
    z = rnd((1, 5, 6), 901).abs()
    gamma = torch.tensor(86422511.6, dtype=f64)
    beta = torch.tensor(3.01635241, dtype=f64)
    eta = P.eta_torch(gamma, beta, z, T)
    u, v, p = rnd((2, 1, 4, 5), 902), rnd((2, 1, 4, 5), 903), rnd((2, 1, 4, 5), 904)
    pu, pv, pp = P.pad_uvp(u.clone(), v.clone(), p.clone())
    g = rnd((1, 2, 4, 5), 905)
    pg = P.pad_grad(g, (1, 2, 1, 2))
    ones = np.ones((2, 3))
    sv = SC.scale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "uprev")
    uv = SC.unscale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "vprev")
    pid = SC.scale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "pprev")
    npz("g9_helpers", T=T, z=z, gamma=gamma, beta=beta, eta=eta, u=u, v=v, p=p, pu=pu, pv=pv, pp=pp, g=g, pg=pg,
        scale_u=sv, unscale_v=uv, scale_p=pid)


def g10():
    n_newfluid = P.count_parameters(P.NewFluidNet(5, 7, 64, 1, CPU, "gelu", "zeros", "curl", use_symm=False,
                                                  a_bound=10, repeats=4, f=5, p_pred=False, factor=2))
    n_cfg2 = P.count_parameters(P.Unet(5, 11, 16, 4, CPU, "gelu", "reflect", "mae", use_symm=True, repeats=3,
                                       f=5, p_pred=True))
    n_cfg1 = P.count_parameters(pycold.ConvAE(2, 3, 16, 3, CPU, "gelu", "reflect", "mae", use_symm=True,
                                              repeats=2, f=3, p_pred=True))
    n_cfg1_plain = P.count_parameters(pycold.ConvAE(2, 3, 16, 3, CPU, "gelu", "reflect", "mae", use_symm=False,
                                                    repeats=2, f=3, p_pred=True))
    m = P.Unet(5, 11, 16, 4, CPU, "gelu", "reflect", "mae", use_symm=True, repeats=3, f=5, p_pred=True)
    keys = list(m.state_dict().keys())
    shapes = [tuple(v.shape) for v in m.state_dict().values()]
    c = pycold.ConvAE(2, 3, 16, 3, CPU, "gelu", "reflect", "mae", use_symm=True, repeats=2, f=3, p_pred=True)
    npz("g10_known_answers", newfluidnet=n_newfluid, unet_cfg2=n_cfg2, convae_cfg1=n_cfg1,
        convae_cfg1_plain=n_cfg1_plain, unet_keys=np.array(keys),
        unet_shapes=np.array([",".join(map(str, s)) for s in shapes]),
        convae_keys=np.array(list(c.state_dict().keys())),
        convae_shapes=np.array([",".join(map(str, tuple(v.shape))) for v in c.state_dict().values()]))


# ------------------------------------------------------------------ G11 two full training steps
def g11():
    """zero_grad -> get_loss -> backward -> Adam.step (multigpu.py:307-320, 761-763) twice on a
    tiny Unet at the reference's native 128x506 grid, fp64."""
    B, H, W = 2, 128, 506
    for tag, (loss_type, p_pred, c_i, c_o, ls, ld) in {
        # c_i = 10 even with p_pred: get_loss always rebuilds a 10-channel input (multigpu.py:234-248)
        "mass": ("mass", True, 10, 4, False, False),
        "curl": ("curl", False, 10, 2, True, True),
    }.items():
        m = P.Unet(3, c_i, 8, c_o, CPU, "gelu", "reflect", loss_type, use_symm=True, repeats=2, f=5,
                   p_pred=p_pred).double()
        randomize_(m, 110)
        sd0 = {("sd0/" + k): v.clone().float() for k, v in m.state_dict().items()}
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0)
        ns = _trainer_ns(m, p_pred, ls, ld, loss_type)
        losses = []
        for step in range(2):
            # get_loss splits 11 channels when p_pred (multigpu.py:198-201) but feeds the net 10
            gVTp = torch.from_numpy(fields.unet_input(B, H, W, 1100 + step, c_i=11 if p_pred else 10))
            truth = [fields.smooth_field(B, H, W, 1150 + step), fields.smooth_field(B, H, W, 1160 + step)]
            if p_pred:
                truth.append(fields.smooth_field(B, H, W, 1170 + step, amp=0.5))
            truth.append(fields.temperature_field(B, H, W, 1180 + step))
            uvp = torch.from_numpy(np.stack(truth, 1))
            if loss_type != "curl":
                # feed 3-D predictions like the curl head does (Appendix A.6 broadcast defect otherwise)
                inner = m

                class Sq(torch.nn.Module):
                    def forward(self, x):
                        u, v, p, T = inner(x)
                        return u[:, 0], v[:, 0], (p[:, 0] if p is not None else None), T[:, 0]
                ns.model_uvp = Sq()
            opt.zero_grad()
            out = G.Trainer.get_loss(ns, gVTp, uvp, None, None, None)
            out[0].backward()
            if step == 0:
                g0 = {("grad0/" + k): p.grad.clone() for k, p in m.named_parameters()}
            opt.step()
            losses.append([float(o) for o in out])
        sd2 = {("sd2/" + k): v for k, v in m.state_dict().items()}
        npz(f"g11_train_{tag}", losses=np.array(losses),
            cfg=np.array([3, c_i, 8, c_o, 2, 5, int(p_pred), 1, int(ls), int(ld)]),
            loss_type=np.array(loss_type), **sd0, **g0, **sd2)


# ------------------------------------------------------------------ G21 get_loss with roll_forward = 2 (multigpu.py:207-248)
def g21():
    """The reference's get_loss with roll_forward = 2 on a tiny Unet at its native 128 x 506 grid (the views are hard-coded),
    fp64: the six losses and every parameter gradient."""
    B, H, W, R = 1, 128, 506, 2
    m = P.Unet(3, 10, 8, 4, CPU, "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True).double()
    randomize_(m, 210)
    sd0 = {("sd0/" + k): v.clone().float() for k, v in m.state_dict().items()}
    ns = _trainer_ns(m, True, False, False, "mass")
    ns.roll_forward = R
    gVTp = torch.from_numpy(fields.unet_input(B, H, W, 2100, c_i=11))
    truth = [fields.smooth_field(B, H, W, 2150), fields.smooth_field(B, H, W, 2160), fields.smooth_field(B, H, W, 2170, amp=0.5),
             fields.temperature_field(B, H, W, 2180)]
    uvp = torch.from_numpy(np.stack(truth, 1))
    paras = torch.tensor([[5.0, 1.0e7, 10.0]], dtype=f64).view(B, 3, 1, 1)
    inner = m

    class Sq(torch.nn.Module):
        def forward(self, x):
            u, v, p, T = inner(x)
            return u[:, 0], v[:, 0], p[:, 0], T[:, 0]
    ns.model_uvp = Sq()
    out = G.Trainer.get_loss(ns, gVTp, uvp, None, paras, gVTp[:, 1:2])
    out[0].backward()
    g0 = {("grad0/" + k): p.grad.clone() for k, p in m.named_parameters()}
    npz("g21_get_loss_roll2", losses=np.array([float(o) for o in out]), paras=paras.view(B, 3),
        cfg=np.array([3, 10, 8, 4, 2, 5, 1, 1, 0, 0, R]), **sd0, **g0)


# ------------------------------------------------------------------ G12 NewFluidNet
def g12():
    for tag, (loss_type, r_p, symm, act, c_i) in {
        "mae_zeros": ("mae", "zeros", True, "gelu", 7),
        "curl_rep": ("curl", "replicate", False, "selu", 7),
    }.items():
        m = P.NewFluidNet(3, c_i, 8, 3, CPU, act, r_p, loss_type, use_symm=symm, repeats=2, f=5, p_pred=True).double()
        randomize_(m, 120)
        x = torch.from_numpy(fields.unet_input(1, 128, 506, 121, c_i=c_i)).requires_grad_(True)   # the net hard-codes 128 x 506
        outs = m(x)
        loss = 0.0
        save = {}
        for n, o in zip(["u", "v", "p"], outs):
            ct = rnd(o.shape, 122 + len(save))
            loss = loss + (o * ct).sum()
            save["out/" + n] = o
            save["ct/" + n] = ct.float()
        loss.backward()
        npz(f"g12_newfluidnet_{tag}", dx_sample=fields.strided_sample(x.grad.numpy(), 1021),
            cfg=np.array([3, c_i, 8, 3, 2, 5, 1, int(symm)]), loss_type=np.array(loss_type), r_p=np.array(r_p),
            act=np.array(act), **save, **sd_np(m), **grads_np(m))


# ------------------------------------------------------------------ G13 get_loss, FluidNet branch (multigpu.py:138-195)
class _Stub3(torch.nn.Module):
    def __init__(self, u, v, p):
        super().__init__()
        self.o = (u, v, p)

    def forward(self, x):
        return self.o


def g13():
    B, H, W = 2, 128, 506
    rows, samples, case = [], {}, 0
    for p_pred in (True, False):
        for loss_type in ("mae", "mass", "curl"):
            for loss_scale in (False, True):
                for loss_derivative in (False, True):
                    seed = 1300 + case
                    u = torch.from_numpy(fields.smooth_field(B, H, W, seed + 1, noise=0.01)).requires_grad_(True)
                    v = torch.from_numpy(fields.smooth_field(B, H, W, seed + 2, noise=0.01)).requires_grad_(True)
                    # p as [B,H,W] (what the 'curl' head returns); the 'mae' head's un-squeezed [B,1,H,W] would broadcast
                    # against the [B,H,W] truth - the build squeezes
                    p = torch.from_numpy(fields.smooth_field(B, H, W, seed + 3, amp=0.5)).requires_grad_(True)
                    truth = [fields.smooth_field(B, H, W, seed + 5), fields.smooth_field(B, H, W, seed + 6)]
                    if p_pred:
                        truth.append(fields.smooth_field(B, H, W, seed + 7, amp=0.5))
                    uvp = torch.from_numpy(np.stack(truth, 1))
                    gVTp = torch.from_numpy(fields.unet_input(B, H, W, seed, c_i=7))
                    ns = _trainer_ns(_Stub3(u, v, p if p_pred else None), p_pred, loss_scale, loss_derivative, loss_type)
                    ns.net = "newfluidnet"
                    out = G.Trainer.get_loss(ns, gVTp, uvp, None, None, None)
                    out[0].backward()
                    rows.append([int(p_pred), {"mae": 0, "mass": 1, "curl": 2}[loss_type], int(loss_scale),
                                 int(loss_derivative), seed] + [float(o) for o in out])
                    samples[f"du/{case}"] = fields.strided_sample(u.grad.numpy())
                    samples[f"dv/{case}"] = fields.strided_sample(v.grad.numpy())
                    if p_pred:
                        samples[f"dp/{case}"] = fields.strided_sample(p.grad.numpy())
                    case += 1
    npz("g13_get_loss_fluidnet", table=np.array(rows), **samples)


# ------------------------------------------------------------------ G14 ADNet step, G15 TS rollout
def _grid(H, W):
    # cell-centred non-uniform-looking grid with wall nodes (datasetio.py:401-404 pattern): interior centres, walls at 0 / 4, 0 / 1
    xs = np.concatenate(([0.0], (np.arange(W - 2) + 0.5) * 4.0 / (W - 2), [4.0]))
    ys = np.concatenate(([0.0], (np.arange(H - 2) + 0.5) * 1.0 / (H - 2), [1.0]))
    xc = torch.from_numpy(np.broadcast_to(xs[None, :], (H, W)).copy()).view(1, 1, H, W)
    yc = torch.from_numpy(np.broadcast_to(ys[:, None], (H, W)).copy()).view(1, 1, H, W)
    return xc, yc


def g14():
    H, W = 128, 506
    ad = P.ADNet(CPU)
    xc, yc = _grid(H, W)
    out = {}
    for k, seed in enumerate((1400, 1410)):
        u = torch.from_numpy(fields.smooth_field(1, H, W, seed + 1)).view(1, 1, H, W) * 400.0
        v = torch.from_numpy(fields.smooth_field(1, H, W, seed + 2)).view(1, 1, H, W) * 400.0
        T = torch.from_numpy(fields.temperature_field(1, H, W, seed + 3)).view(1, 1, H, W)
        raq = torch.full((1, 1, H, W), 2.5, dtype=f64)
        inp = torch.cat((u, v, T, raq, xc.clone(), yc.clone()), dim=1)
        Tn, dt = ad(inp.clone())
        out[f"T_next/{k}"], out[f"dt/{k}"] = (Tn if k == 0 else fields.strided_sample(Tn.numpy(), 4001)), dt
        Tn2, dt2 = ad(inp.clone(), dt=torch.tensor(3e-7, dtype=f64))
        out[f"T_next_fixed/{k}"] = fields.strided_sample(Tn2.numpy(), 4001)
    npz("g14_adnet", seeds=np.array([1400, 1410]), **out)


class _StokesStub(torch.nn.Module):
    """A deterministic stand-in for the Stokes net: smooth velocities that depend on the temperature input."""

    def forward(self, inp):
        T = inp[:, 6:7]
        a = torch.cumsum(T - T.mean(), dim=3) * 0.01
        u = (a[:, :, 2:, 1:-1] - a[:, :, :-2, 1:-1]) * 0.5
        v = -(a[:, :, 1:-1, 2:] - a[:, :, 1:-1, :-2]) * 0.5
        u = F.pad(u, (1, 1, 1, 1))
        v = F.pad(v, (1, 1, 1, 1))
        return u[:, 0], v[:, 0], T[:, 0] * 0.0


def g15():
    import torch.nn.functional as F_  # noqa: F401
    H, W = 128, 506
    xc, yc = _grid(H, W)
    T0 = torch.from_numpy(fields.temperature_field(1, H, W, 1500)).view(1, 1, H, W)
    raq, fkt, fkp = (torch.tensor(v, dtype=f64) for v in (2.5, 1e7, 30.0))
    nd = [torch.tensor(v, dtype=f64).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]
    ts = P.TS(_StokesStub(), P.ADNet(CPU), CPU, ts=3, net="newfluidnet")
    x, dts, u, v, p, V = ts(T0.clone(), None, None, yc.clone(), nd[0], nd[1], nd[2], raq, fkt, fkp, xc.clone(), yc.clone())
    smp = lambda t: fields.strided_sample(t.numpy(), 4001)  # noqa: E731
    npz("g15_ts_rollout", T1=smp(x[1]), T2=smp(x[2]), T3=x[3], dts=np.array([float(dts[i]) for i in (1, 2, 3)]), u=smp(u),
        v=smp(v), V=smp(V),
        nd=np.array([0.25, 0.26, 0.74]), paras=np.array([2.5, 1e7, 30.0]))


class _UnetStub(torch.nn.Module):
    """Stand-in for the U-Net of the TS 'unet' branch: (u, v, p, T_next) as smooth functions of the 10-channel input."""

    def forward(self, inp):
        T, up, vp, dt, V = inp[:, 7], inp[:, 8], inp[:, 9], inp[:, 2], inp[:, 6]
        Tn = T + dt * (torch.roll(T, 1, dims=2) - T) * 50.0 + 0.01 * V
        return up * 0.9 + 0.1 * T, vp * 0.8 - 0.05 * T, None, Tn


def g20():
    """TS(net='unet') of the reference (:411-446) with a stub network: pins the oracle's restatement of that branch."""
    H, W = 128, 506
    xc, yc = _grid(H, W)
    T0 = torch.from_numpy(fields.temperature_field(1, H, W, 2000)).view(1, 1, H, W)
    up = torch.from_numpy(fields.smooth_field(1, H, W, 2001)).view(1, 1, H, W)
    vp = torch.from_numpy(fields.smooth_field(1, H, W, 2002)).view(1, 1, H, W)
    dt = torch.full((1, 1, H, W), 3e-5, dtype=f64)
    raq, fkt, fkp = (torch.tensor(v, dtype=f64) for v in (2.5, 1e7, 30.0))
    nd = [torch.tensor(v, dtype=f64).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]
    ts = P.TS(_UnetStub(), None, CPU, ts=3, net="unet")
    x, dts, u, v, p, V = ts(T0.clone(), None, None, yc.clone(), nd[0], nd[1], nd[2], raq, fkt, fkp, xc.clone(), yc.clone(),
                            u_prev=up.clone(), v_prev=vp.clone(), dt=dt.clone())
    assert p is None and len(dts) == 0
    smp = lambda t: fields.strided_sample(t.numpy(), 4001)  # noqa: E731
    npz("g20_ts_rollout_unet", T1=smp(x[1]), T2=smp(x[2]), T3=x[3], u=smp(u), v=smp(v), V=smp(V),
        nd=np.array([0.25, 0.26, 0.74]), paras=np.array([2.5, 1e7, 30.0]), dt=np.array(3e-5))


# ------------------------------------------------------------------ G16 BoundaryLearnedConvolution2D
def g16():
    for tag, (c_i, c_o, k, symm, H, W) in {"k5_symm": (8, 16, 5, True, 23, 37), "k3_plain": (16, 8, 3, False, 19, 21)}.items():
        m = P.BoundaryLearnedConvolution2D(c_i, c_o, k, use_symm=symm).double()
        randomize_(m, 160)
        with torch.no_grad():
            m.learnable_bias.copy_((0.1 * torch.randn(m.learnable_bias.shape, generator=torch.Generator().manual_seed(161))).float().double())
        x = rnd((2, c_i, H, W), 162).requires_grad_(True)
        y = m(x)
        ct = rnd(y.shape, 163)
        (y * ct).sum().backward()
        npz(f"g16_learned_{tag}", meta=np.array([c_i, c_o, k, int(symm)]), x=x.detach().float(), y=y, ct=ct.float(), dx=x.grad,
            **sd_np(m), **grads_np(m))
    # the FluidLayer around it (conv -> GroupNorm -> GELU, :790-799)
    m = P.FluidLayer(8, 16, "gelu", "learned", True, 1, f=5).double()
    randomize_(m, 165)
    x = rnd((2, 8, 23, 37), 166).requires_grad_(True)
    y = m(x)
    ct = rnd(y.shape, 167)
    (y * ct).sum().backward()
    npz("g16_learned_fluidlayer", x=x.detach().float(), y=y, ct=ct.float(), dx=x.grad, **sd_np(m), **grads_np(m))


# ------------------------------------------------------------------ G17 Unet with learned padding
def g17():
    m = P.Unet(3, 10, 8, 4, CPU, "gelu", "learned", "mae", use_symm=True, repeats=2, f=5, p_pred=True).double()
    randomize_(m, 170)
    with torch.no_grad():
        g_ = torch.Generator().manual_seed(171)
        for n, p in m.named_parameters():
            if n.endswith("learnable_bias"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g_)).float().double())
    x = torch.from_numpy(fields.unet_input(2, 40, 54, 172, c_i=10)).requires_grad_(True)
    outs = m(x)
    loss, save = 0.0, {}
    for n, o in zip("uvpT", outs):
        ct = rnd(o.shape, 173 + len(save))
        loss = loss + (o * ct).sum()
        save["out/" + n] = o
        save["ct/" + n] = ct.float()
    loss.backward()
    # (no CPU restatement of the learned-padding Unet: these vectors pin the GPU path only, f32 is enough)
    save = {k_: v.detach().float() for k_, v in save.items()}
    npz("g17_unet_learned", cfg=np.array([3, 10, 8, 4, 2, 5, 1, 1]), **save, **sd_np(m),
        **{k_: v.float() for k_, v in grads_np(m).items()})
    # NewFluidNet with learned padding (two levels: 128x506 and 64x253)
    m = P.NewFluidNet(2, 7, 8, 4, CPU, "gelu", "learned", "mae", use_symm=True, repeats=1, f=5, p_pred=True).double()
    randomize_(m, 175)
    x = torch.from_numpy(fields.unet_input(1, 128, 506, 176, c_i=7)).requires_grad_(True)
    outs = m(x)
    loss, save = 0.0, {}
    for n, o in zip("uvp", outs):
        ct = rnd(o.shape, 177 + len(save))
        loss = loss + (o * ct).sum()
        save["out/" + n] = fields.strided_sample(o.detach().numpy(), 20001).astype(np.float32)
        save["ct/" + n] = ct.float()
    loss.backward()
    npz("g17_newfluidnet_learned", cfg=np.array([2, 7, 8, 4, 1, 5, 1, 1]), **save, **sd_np(m),
        **{k_: v.float() for k_, v in grads_np(m).items()})


# ------------------------------------------------------------------ G18 NewADDataset on shard files in the reference's layout
def g18():
    """The reference's own `datasetio.NewADDataset` (datasetio.py:320-654) run on a tiny data directory written in ITS file
    layout (sims.pt, <an>/sim_<n>/{times,xc,yc}.pt, e1_{u,v,p,T}prev_data_select[_init|_snaps].pt, e1_i_vec_select[_init].pt).
    The fixture holds the file contents (so the test re-creates the directory) and every item the reference returns."""
    import tempfile
    import datasetio as D
    H, W = 10, 14
    g = torch.Generator().manual_seed(180)
    sims = [(3, "train", 2.5, 1e7, 12.0, 0.0, 4.0, 0), (5, "cv", 6.0, 3e8, 40.0, 0.0, 4.0, 0), (7, "train", 8.5, 5e6, 3.0, 0.0, 4.0, 0)]
    files = {}
    root = tempfile.mkdtemp()
    torch.save(sims, root + "/sims.pt")
    yy, xx = torch.meshgrid(torch.linspace(0.02, 0.97, H, dtype=f64), torch.linspace(0.05, 3.9, W, dtype=f64), indexing="ij")
    for num, an, *_ in sims:
        d = f"{root}/{an}/sim_{num}"
        os.makedirs(d)
        content = {"times": torch.cumsum(torch.rand(9, generator=g, dtype=f64) * 1e-4, 0), "xc": xx.clone(), "yc": yy.clone()}
        for suf, m in (("_select", 5), ("_select_init", 3), ("_select_snaps", 4)):
            for k in "uvpT":
                amp = {"u": 300.0, "v": 200.0, "p": 1.0, "T": 0.5}[k]
                t = torch.rand((m, 1, H, W), generator=g, dtype=torch.float32).to(f64) * amp
                content[f"e1_{k}prev_data{suf}"] = t
            if suf != "_select_snaps":
                content[f"e1_i_vec{suf}"] = torch.tensor(sorted(torch.randperm(40, generator=g)[:m].tolist()))
        for k, v in content.items():
            torch.save(v, f"{d}/{k}.pt")
            files[f"file/{an}/sim_{num}/{k}"] = v
    out = dict(files)
    out["sims_num"] = np.array([s_[0] for s_ in sims]); out["sims_an"] = np.array([s_[1] for s_ in sims])
    out["sims_par"] = np.array([[s_[2], s_[3], s_[4], s_[5], s_[6], s_[7]] for s_ in sims])
    iv3 = files["file/train/sim_3/e1_i_vec_select"].tolist()
    iv7 = files["file/train/sim_7/e1_i_vec_select"].tolist()
    cases = {
        "all": dict(an="train", is_init=False, p_pred=True, debug=False, sims_vec=[], times_vec=[]),
        "init": dict(an="train", is_init=True, p_pred=True, debug=False, sims_vec=[], times_vec=[]),
        "snaps": dict(an="train", is_init=False, p_pred=False, debug=True, sims_vec=[], times_vec=[]),
        "cv": dict(an="cv", is_init=False, p_pred=False, debug=False, sims_vec=[], times_vec=[]),
        "filtered": dict(an="train", is_init=False, p_pred=True, debug=False, sims_vec=[3, 3, 7], times_vec=[iv3[1], iv3[3], iv7[0]]),
        "half": dict(an="train", is_init=False, p_pred=True, debug=False, sims_vec=[], times_vec=[], max_examples_percent_per_epoch=50),
    }
    for name, kw in cases.items():
        ds = D.NewADDataset(root, scale=True, load=False, noise=0.0, **kw)
        items = [ds[i] for i in range(len(ds))]
        out[f"{name}/n"] = len(ds)
        out[f"{name}/x"] = torch.stack([it[0] for it in items])
        out[f"{name}/y"] = torch.stack([it[1] for it in items])
        out[f"{name}/t"] = torch.stack([it[2].reshape(()) for it in items])
        out[f"{name}/s"] = torch.stack([torch.as_tensor(it[3]).reshape(()) for it in items])
        if name == "filtered":
            out["filtered/sims_vec"] = np.array(kw["sims_vec"]); out["filtered/times_vec"] = np.array(kw["times_vec"])
    npz("g18_newad_dataset", **out)


# ------------------------------------------------------------------ G19 ADTimeDataset on shard files in the reference's layout
def g19():
    """The reference's `datasetio.ADTimeDataset` (datasetio.py:63-280) on a tiny data directory in its file layout (full
    series e1_{u,v,p,T}prev_data.pt, the debug mode's *_select_init.pt, times / xc / yc / sims).  `__getitem__` draws a random
    initial-condition pair when idx0 % 8 == 0: the python RNG is seeded per item, here and in the test."""
    import random
    import tempfile
    import datasetio as D
    H, W, M = 8, 12, 11
    g = torch.Generator().manual_seed(190)
    sims = [(2, "train", 3.5, 2e7, 20.0, 0.0, 4.0, 0), (4, "train", 7.0, 4e8, 5.0, 0.0, 4.0, 0), (9, "cv", 1.0, 1e9, 60.0, 0.0, 4.0, 0)]
    root = tempfile.mkdtemp()
    torch.save(sims, root + "/sims.pt")
    yy, xx = torch.meshgrid(torch.linspace(0.03, 0.96, H, dtype=f64), torch.linspace(0.04, 3.95, W, dtype=f64), indexing="ij")
    out = {}
    for num, an, *_ in sims:
        d = f"{root}/{an}/sim_{num}"
        os.makedirs(d)
        content = {"times": torch.cumsum(torch.rand(M, generator=g, dtype=f64) * 1e-4, 0), "xc": xx.clone(), "yc": yy.clone()}
        for suf, m in (("", M), ("_select_init", 3)):
            for k in "uvpT":
                amp = {"u": 300.0, "v": 200.0, "p": 1.0, "T": 0.5}[k]
                content[f"e1_{k}prev_data{suf}"] = torch.rand((m, 1, H, W), generator=g, dtype=torch.float32).to(f64) * amp
        for k, v in content.items():
            torch.save(v, f"{d}/{k}.pt")
            out[f"file/{an}/sim_{num}/{k}"] = v
    out["sims_num"] = np.array([s_[0] for s_ in sims]); out["sims_an"] = np.array([s_[1] for s_ in sims])
    out["sims_par"] = np.array([[s_[2], s_[3], s_[4], s_[5], s_[6], s_[7]] for s_ in sims])
    t2 = out["file/train/sim_2/times"].tolist()
    t4 = out["file/train/sim_4/times"].tolist()
    cases = {
        "all": dict(an="train", p_pred=True, debug=False, sims_vec=[], times_vec=[], roll_forward=1),
        "roll2": dict(an="train", p_pred=False, debug=False, sims_vec=[], times_vec=[], roll_forward=2),
        "debug": dict(an="cv", p_pred=False, debug=True, sims_vec=[], times_vec=[], roll_forward=1),
        "filtered": dict(an="train", p_pred=True, debug=False, sims_vec=[2, 2, 2, 4, 4], times_vec=[t2[0], t2[3], t2[4], t4[0], t4[7]],
                         roll_forward=1),
    }
    for name, kw in cases.items():
        ds = D.ADTimeDataset(root, scale=True, load=False, noise=0.0, **kw)
        out[f"{name}/n"] = len(ds)
        out[f"{name}/indices"] = np.array(ds.indices).reshape(-1, 2)
        out[f"{name}/indices_init"] = np.array(ds.indices_init).reshape(-1, 2)
        items = []
        for i in range(len(ds)):
            random.seed(1000 + i)
            items.append(ds[i])
        out[f"{name}/x"] = torch.stack([it[0] for it in items])
        out[f"{name}/y"] = torch.stack([it[1] for it in items])
        out[f"{name}/s"] = torch.stack([torch.as_tensor(it[2]).reshape(()) for it in items])
        out[f"{name}/paras"] = torch.stack([it[3].reshape(3) for it in items])
        out[f"{name}/yc"] = items[0][4]
        if name == "filtered":
            out["filtered/sims_vec"] = np.array(kw["sims_vec"]); out["filtered/times_vec"] = np.array(kw["times_vec"])
    npz("g19_adtime_dataset", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16, g17, g18, g19, g20, g21):
        if not only or fn.__name__ in only:
            fn()
