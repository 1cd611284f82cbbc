"""GPU parity at BASELINE.json's full sizes (CFG-1 ConvAE 128x128 B=4; CFG-2/3 U-Net 506x506; CFG-5 1024x1024) against the
CPU oracle on the same seeded synthetic fields, plus size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unet(loss_type="mass", c_i=10, seed=0):
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    torch.manual_seed(seed)
    return Unet(5, c_i, 16, 4, torch.device("cpu"), "gelu", "reflect", loss_type, use_symm=True, repeats=3, f=5, p_pred=True)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_cfg2_unet_506_fp32_field_mae_below_1e5():
    """The north-star parity gate: u, v, p, T of the full symmetric U-Net at 506x506 within MAE 1e-5 of the CPU reference
    path (oracle evaluated in fp64 on the same inputs and weights)."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    m = _unet()
    sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    x = synthetic_batch(1, 506, 506, 11, p_pred=True)[0][:, :10]
    m = m.to(DEV).set_precision("fp32")
    outs = m(x.to(DEV))
    ref = O.unet_forward(sd, x.double(), levels=5, repeats=3, act="gelu", r_p="reflect", loss_type="mass", use_symm=True,
                         p_pred=True)
    for name, o, r in zip("uvpT", outs, ref):
        mae = float((o.detach().double().cpu() - r).abs().mean())
        assert mae < 1e-5, (name, mae)


def test_cfg2_unet_506_bf16_vs_quantised_oracle():
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    m = _unet()
    sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    x = synthetic_batch(1, 506, 506, 12, p_pred=True)[0][:, :10]
    m = m.to(DEV).set_precision("bf16")
    y = m.features(x.to(DEV))
    q = lambda t: t.to(torch.bfloat16).to(t.dtype)  # noqa: E731
    ref = O.unet_features_quantised(sd, x.double(), 5, 3, "gelu", "reflect", True, q)
    assert rel_l2(y, ref) < 3e-2, rel_l2(y, ref)      # 27 layers deep: rounding-boundary flips accumulate (measured 2.1e-2)
    full = O.unet_features(sd, x.double(), 5, 3, "gelu", "reflect", True)
    # and the stated bf16 bound vs the unquantised reference: MAE <= 1.5 x the reference-in-bf16 level (5e-3 on O(0.07) fields)
    assert float((y.double().cpu() - full).abs().mean()) < 5e-3 * max(1.0, float(full.abs().mean()) / 0.07)


def test_cfg3_training_step_506_fp32_vs_oracle():
    """One full CFG-3 step (data + divergence + momentum residual, Adam) at 506x506, batch 2: loss tuple and the updated
    weights against the CPU oracle's step."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    m = _unet()
    sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    gVTp, uvp, scaler, paras, yc = synthetic_batch(2, 506, 506, 13, p_pred=True)
    lam = 1e-6
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                 lambda_mom=lam, precision="fp32")
    out8 = tr.train_step(*(t.to(DEV) for t in (gVTp, uvp, yc, paras, scaler)))
    st = O.CpuUnetStep(sd, dict(levels=5, repeats=3, act="gelu", r_p="reflect", loss_type="mass", use_symm=True, p_pred=True))
    ref = st.step(gVTp.double(), uvp.double(), momentum=dict(lambda_mom=lam, yc=yc.double(), paras=paras.double(),
                                                              scaler=scaler.double()))
    got = out8[:7].tolist()
    for i in range(6):
        assert abs(got[i] - ref[i]) <= 1e-4 * max(abs(ref[i]), 1e-3), (i, got, ref)
    assert abs(got[6] - ref[6]) <= 2e-3 * abs(ref[6])              # momentum residual: |R| ~ 1e4, f32 stencils
    moved = close = 0
    for n, p in m.named_parameters():
        d = (p.detach().cpu().double() - st.sd[n].detach()).abs()
        moved += d.numel()
        close += int((d <= 2e-4).sum())
    assert close / moved > 0.999, close / moved                     # Adam moves each weight by <= lr = 1e-3


def _grad_table(tr, st):
    """per-tensor relative L2 of the HIP flat gradient against the oracle's autograd gradient, and the whole-buffer figure"""
    rows, num, den = [], 0.0, 0.0
    for n, gv in tr.flat.views(tr.flat.grad).items():
        gr = st.sd[n].grad.double()
        d = float((gv.detach().cpu().double() - gr).norm())
        num, den = num + d * d, den + float(gr.norm()) ** 2
        rows.append((n, d / max(float(gr.norm()), 1e-300), float(gr.norm())))
    return rows, (num / den) ** 0.5


@pytest.mark.parametrize("lam", [0.0, 1e-6])
def test_cfg3_training_step_506_bf16_gradients_vs_oracle(lam):
    """The benched precision at the benched resolution: one CFG-3 step in bf16 at 506x506, batch 2 -- loss tuple and EVERY
    parameter gradient against the fp64 CPU oracle (same inputs, same weights).  Bounds are the measured bf16 levels x ~1.5:
    the gradient of a 27-layer bf16 network is an O(2^-8)-noisy estimate, the per-tensor figure is worst for the first-level
    GroupNorm offsets whose true gradient nearly cancels over 506^2 pixels."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    m = _unet()
    sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    gVTp, uvp, scaler, paras, yc = synthetic_batch(2, 506, 506, 13, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                 lambda_mom=lam, precision="bf16")
    out8 = tr.train_step(*(t.to(DEV) for t in (gVTp, uvp, yc, paras, scaler)))
    st = O.CpuUnetStep(sd, dict(levels=5, repeats=3, act="gelu", r_p="reflect", loss_type="mass", use_symm=True, p_pred=True))
    mom = dict(lambda_mom=lam, yc=yc.double(), paras=paras.double(), scaler=scaler.double()) if lam else None
    ref = st.step(gVTp.double(), uvp.double(), momentum=mom)
    got = out8[:7].tolist()
    rows, whole = _grad_table(tr, st)
    worst = max((r for r in rows if r[2] > 1e-8), key=lambda r: r[1])
    print(f"\nbf16 506^2 B=2 lam={lam}: loss got {got} ref {list(ref)}\nflat gradient rel-L2 {whole:.3e}; worst tensor {worst}")
    for r in sorted(rows, key=lambda r: -r[1])[:8]:
        print("   %-28s rel %.3e  |g| %.3e" % r)
    for i in (1, 2, 3, 4):                                  # u, v, p, T data terms
        assert abs(got[i] - ref[i]) <= 1e-2 * abs(ref[i]), (i, got, ref)
    assert abs(got[5] - ref[5]) <= 5e-2 * abs(ref[5]), (got, ref)       # divergence: first differences x 126 of bf16-noisy u, v
    if lam:
        # momentum residual (second differences x 126^2): the Trainer runs the split-precision ("mixed") forward pass for
        # it; what is left is the bf16 noise of the coarser levels (4 % in the fp64 emulation, tests/study_bf16_momentum.py)
        assert m.precision == "mixed"
        assert abs(got[6] - ref[6]) <= 8e-2 * abs(ref[6]), (got, ref)
        assert abs(got[0] - ref[0]) <= 8e-2 * abs(ref[0]), (got, ref)
    big = [r for r in rows if r[2] > 1e-8]                  # conv.5.bias: the output mean is subtracted, its gradient is 0
    worst = max(big, key=lambda r: r[1])
    assert whole <= (0.10 if not lam else 0.25), whole
    assert worst[1] <= (0.15 if not lam else 0.6), worst


def test_cfg5_unet_1024_fp32_field_mae_below_1e5():
    """CFG-5 resolution through the f32 path: u, v, p, T at 1024x1024, batch 1, within MAE 1e-5 of the fp64 oracle."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    m = _unet(seed=3)
    sd = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    x = synthetic_batch(1, 1024, 1024, 14, p_pred=True)[0][:, :10]
    m = m.to(DEV).set_precision("fp32")
    outs = m(x.to(DEV))
    ref = O.unet_forward(sd, x.double(), levels=5, repeats=3, act="gelu", r_p="reflect", loss_type="mass", use_symm=True,
                         p_pred=True)
    for name, o, r in zip("uvpT", outs, ref):
        mae = float((o.detach().double().cpu() - r).abs().mean())
        assert mae < 1e-5, (name, mae)


def test_cfg1_convae_128_batch4_fp32_vs_oracle():
    from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE
    torch.manual_seed(1)
    m = ConvAE(2, 3, 16, 3, None, "gelu", "reflect", "mae", use_symm=True, repeats=2, f=3, p_pred=True)
    sd = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    x = torch.randn((4, 3, 128, 128), generator=g)
    ct = torch.randn((4, 3, 128, 128), generator=g)
    m = m.to(DEV).set_precision("fp32")
    y = m(x.to(DEV))
    ref = O.convae_forward(sd, x.double(), levels=2, c_i=3, c_h=16, c_o=3, repeats=2, act="gelu", r_p="reflect",
                           loss_type="mae", use_symm=True, p_pred=True)
    assert float((y.double().cpu() - ref.detach()).abs().mean()) < 1e-5
    (y * ct.to(DEV)).sum().backward()
    (ref * ct.double()).sum().backward()
    worst = max(rel_l2(p.grad, sd[n].grad) for n, p in m.named_parameters() if float(sd[n].grad.abs().max()) > 1e-8)
    assert worst < 2e-3, worst


def test_cfg5_1024_mixed_precision_runs_and_agrees():
    """High-resolution 1024x1024 fields (CFG-5), batch 1: the bf16 step runs, is finite, and its loss agrees with the
    fp32 mode on identical inputs and weights."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    data = synthetic_batch(1, 1024, 1024, 14, p_pred=True)
    losses = {}
    for prec in ("fp32", "bf16"):
        m = _unet(seed=3)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
        tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
                     loss_type="mass", lambda_mom=1e-6, precision=prec)
        gVTp, uvp, scaler, paras, yc = (t.to(DEV) for t in data)
        out8 = tr.train_step(gVTp, uvp, yc, paras, scaler)
        losses[prec] = out8[:7].tolist()
        assert all(np.isfinite(losses[prec]))
        assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    # data terms agree to bf16 accuracy; the momentum residual (second differences x 126^2) through the split-precision
    # forward pass the Trainer selects for it: within 8 % (bf16 noise of the coarser levels: 5.6 % at 1024^2 in the fp64
    # emulation, tests/study_bf16_momentum.py; plain bf16 storage reads 2.5 x the fp32 value here)
    print("\n1024^2 loss tuples", losses)
    for i in (1, 2, 3, 4):
        assert abs(losses["bf16"][i] - losses["fp32"][i]) <= 3e-2 * abs(losses["fp32"][i]), (i, losses)
    for i in (0, 6):
        assert abs(losses["bf16"][i] - losses["fp32"][i]) <= 8e-2 * abs(losses["fp32"][i]), (i, losses)


def test_mirror_property_full_size():
    """SymmetricConv2d mirror property at full size (size-independent): output channel U+i (bias removed) of the layer
    equals the x-flip of channel i computed on the x-flipped input."""
    from pbml_mantle_convection_amd.symmetric_layers_torch import SymmetricConv2d
    torch.manual_seed(4)
    m = SymmetricConv2d(16, 16, 5, padding="same", padding_mode="reflect", symmetry={"h": 4})
    with torch.no_grad():
        m.bias.zero_()
    m = m.to(DEV)
    x = torch.randn((1, 16, 506, 512), device=DEV)
    y = m(x)
    yf = m(x.flip(3)).flip(3)
    U = m.unique_out_channels
    assert float((y[:, U:U + 2] - yf[:, 0:2]).abs().max()) < 2e-5
