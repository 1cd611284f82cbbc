"""GPU parity of the fused training step: loss kernels (forward + backward) against the CPU
oracle's autograd, the momentum residual, fused Adam, and two full training steps against the
golden vectors captured from the reference (zero_grad -> get_loss -> backward -> Adam)."""
import numpy as np
import pytest
import torch

import fields
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
f64 = torch.float64


def dev(a):
    return torch.from_numpy(np.asarray(a)).float().to(DEV).contiguous()


def close(a, b, atol, rtol, what=""):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    err = np.abs(a - b)
    assert (err <= atol + rtol * np.abs(b)).all(), f"{what}: max err {err.max():.3e}, ref max {np.abs(b).max():.3e}"


def _case(B, H, W, seed, p_pred):
    u = fields.smooth_field(B, H, W, seed + 1, noise=0.01)
    v = fields.smooth_field(B, H, W, seed + 2, noise=0.01)
    p = fields.smooth_field(B, H, W, seed + 3, amp=0.5)
    T = fields.temperature_field(B, H, W, seed + 4)
    truth = [fields.smooth_field(B, H, W, seed + 5), fields.smooth_field(B, H, W, seed + 6)]
    if p_pred:
        truth.append(fields.smooth_field(B, H, W, seed + 7, amp=0.5))
    truth.append(fields.temperature_field(B, H, W, seed + 8))
    return u, v, p, T, np.stack(truth, 1)


@pytest.mark.parametrize("p_pred", [True, False])
@pytest.mark.parametrize("loss_type", ["mae", "mass"])
@pytest.mark.parametrize("ls,ld,norm", [(False, False, "l1"), (True, True, "l1"), (True, False, "l2")])
def test_fused_loss_matches_oracle(p_pred, loss_type, ls, ld, norm):
    _fused_loss_case(p_pred, loss_type, ls, ld, norm, 2, 37, 53)


@pytest.mark.parametrize("H,W", [(70, 131), (5, 200), (33, 65)])       # several ragged LDS tiles in x and y; minimum height
def test_fused_loss_matches_oracle_multi_tile(H, W):
    _fused_loss_case(True, "mass", True, True, "l1", 2, H, W)


def _fused_loss_case(p_pred, loss_type, ls, ld, norm, B, H, W):
    from pbml_mantle_convection_amd.losses import StokesLoss
    u, v, p, T, uvp = _case(B, H, W, 100, p_pred)
    chans = [u, v, T] + ([p] if p_pred else [])
    y = dev(np.stack(chans, 1))
    L = StokesLoss(p_pred, loss_type, ls, ld, norm=norm)
    out8, gy = L.evaluate(y, dev(uvp))
    tu, tv, tp, tT = (torch.from_numpy(a).to(f64).requires_grad_(True) for a in (u, v, p, T))
    ref = O.get_loss_unet((tu, tv, tp if p_pred else None, tT), torch.from_numpy(uvp).to(f64), p_pred=p_pred,
                          loss_type=loss_type, loss_scale=ls, loss_derivative=ld, norm=norm)
    close(out8[:6], torch.stack([r.detach() for r in ref]), atol=1e-6, rtol=2e-5, what="loss6")
    ref[0].backward()
    gref = [tu.grad, tv.grad, tT.grad] + ([tp.grad] if p_pred else [])
    for i, gr in enumerate(gref):
        close(gy[:, i], gr, atol=1e-9, rtol=1e-4, what=f"grad ch{i}")


@pytest.mark.parametrize("p_pred", [True, False])
def test_fused_loss_curl_matches_oracle(p_pred):
    """curl mode: streamfunction -> (u, v) with wall conditions, T clip, boundary-strip divergence."""
    from pbml_mantle_convection_amd.losses import StokesLoss
    B, H, W = 2, 23, 31
    a, _, p, T, uvp = _case(B, H, W, 200, p_pred)
    T = T * 1.3 - 0.1            # exercise both clip sides
    y_np = np.stack([a, T] + ([p] if p_pred else []), 1)
    y = dev(y_np)
    L = StokesLoss(p_pred, "curl", True, True, a_bound=10.0)
    out8, gy = L.evaluate(y, dev(uvp))
    ty = torch.from_numpy(y_np).to(f64).requires_grad_(True)
    cu, cv = O.curl_head(ty[:, 0:1] * 10.0)
    pred = (cu[:, 0], cv[:, 0], ty[:, 2] if p_pred else None, torch.clip(ty[:, 1], 0.0, 1.5))
    ref = O.get_loss_unet(pred, torch.from_numpy(uvp).to(f64), p_pred=p_pred, loss_type="curl", loss_scale=True,
                          loss_derivative=True)
    close(out8[:6], torch.stack([r.detach() for r in ref]), atol=1e-6, rtol=2e-5, what="loss6")
    ref[0].backward()
    close(gy, ty.grad, atol=1e-8, rtol=2e-4, what="gy")


@pytest.mark.parametrize("H,W", [(29, 41), (70, 131), (5, 200)])       # one tile; 3 x 9 ragged tiles; the minimum height
def test_momentum_residual_matches_oracle(H, W):
    from pbml_mantle_convection_amd.losses import StokesLoss
    B = 2
    u, v, p, T, uvp = _case(B, H, W, 300, True)
    paras = fields.sim_parameters(B, 5)
    paras[:, 1] = 10.0 ** np.array([2.0, 3.0])      # moderate viscosity contrast keeps the f32 sign pattern stable
    scaler = np.array([30.0, 80.0])
    yc = np.broadcast_to(np.linspace(0, 1, H)[:, None], (H, W)).copy()
    y = dev(np.stack([u, v, T, p], 1))
    lam = 1e-6
    L = StokesLoss(True, "mass", False, False, lambda_mom=lam)
    out8, gy = L.evaluate(y, dev(uvp), dev(yc), dev(paras), dev(scaler))
    tu, tv, tp, tT = (torch.from_numpy(a).to(f64).requires_grad_(True) for a in (u, v, p, T))
    mom = dict(lambda_mom=lam, yc=torch.from_numpy(yc), paras=torch.from_numpy(paras), scaler=torch.from_numpy(scaler))
    ref = O.get_loss_unet((tu, tv, tp, tT), torch.from_numpy(uvp).to(f64), p_pred=True, loss_type="mass", momentum=mom)
    close(out8[6], ref[6].detach(), atol=0, rtol=2e-4, what="momentum value")
    close(out8[0], ref[0].detach(), atol=1e-6, rtol=2e-4, what="loss")
    ref[0].backward()
    for i, gr in enumerate([tu.grad, tv.grad, tT.grad, tp.grad]):
        g = gy[:, i].double().cpu()
        # sign(R) can flip where |R| is at rounding level; require agreement on >= 99.5 % of the pixels
        bad = (g - gr).abs() > 1e-7 + 2e-3 * gr.abs()
        assert bad.double().mean() < 5e-3, (i, float(bad.double().mean()))


@pytest.mark.parametrize("H,W,p_pred,lam,ls,ld,loss_type", [
    (70, 131, True, 1e-6, False, False, "mass"),      # the benched combination, ragged tiles in x and y
    (37, 64, False, 1e-6, True, True, "mass"),        # no pressure channel, scaled + derivative terms, exact tile width
    (16, 200, True, 0.0, True, False, "mae"),         # no momentum term
    (5, 9, True, 1e-6, False, True, "mass"),          # the minimum size: every pixel within two of a wall
])
def test_one_launch_loss_equals_the_three_kernels(monkeypatch, H, W, p_pred, lam, ls, ld, loss_type):
    """mc_loss_fused (plane form and the form that reads the last convolution's CB8 output) against mc_loss_fwd_bwd +
    mc_momentum_residual + mc_momentum_adjoint: the same per-pixel expressions -> the same gradient field bit for bit, the same
    sums up to the order of the f64 atomics."""
    from pbml_mantle_convection_amd import losses
    B = 3
    u, v, p, T, uvp = _case(B, H, W, 400, p_pred)
    paras = fields.sim_parameters(B, 5)
    paras[:, 1] = 10.0 ** np.array([2.0, 3.0, 4.0])
    scaler = np.array([30.0, 80.0, 55.0])
    yc = np.broadcast_to(np.linspace(0, 1, H)[:, None], (H, W)).copy()
    chans = [u, v, T] + ([p] if p_pred else [])
    y = dev(np.stack(chans, 1))
    extra = (dev(yc), dev(paras), dev(scaler)) if lam else (None, None, None)
    res = {}
    for mode in ("three", "planes", "cb8"):
        monkeypatch.setattr(losses, "FUSED_LOSS", mode != "three")
        Lo = losses.StokesLoss(p_pred, loss_type, ls, ld, lambda_mom=lam)
        assert Lo.fusable() == (mode != "three")
        if mode == "cb8":
            crop, mean = 3, torch.randn(B, len(chans), device=DEV)
            buf = torch.randn(B, 1, H, W + 2 * crop, 8, device=DEV)          # crop columns and channels 4..7: garbage
            buf[:, 0, :, crop:crop + W, :len(chans)] = (y + mean[:, :, None, None]).permute(0, 2, 3, 1)
            # (the kernel subtracts the mean again: compare against the planes rounded the same way)
            y_eff = buf[:, 0, :, crop:crop + W, :len(chans)].permute(0, 3, 1, 2).contiguous() - mean[:, :, None, None]
            out8, gy = Lo.evaluate(None, dev(uvp), *extra, cb8=(buf, mean, crop, (B, len(chans), H, W)))
            monkeypatch.setattr(losses, "FUSED_LOSS", False)
            ref8, refg = losses.StokesLoss(p_pred, loss_type, ls, ld, lambda_mom=lam).evaluate(y_eff, dev(uvp), *extra)
            assert torch.equal(gy, refg), float((gy - refg).abs().max())
            close(out8, ref8, atol=1e-7, rtol=1e-6, what="cb8 sums")
            continue
        out8, gy = Lo.evaluate(y, dev(uvp), *extra)
        res[mode] = (out8.clone(), gy.clone())
    assert torch.equal(res["planes"][1], res["three"][1]), float((res["planes"][1] - res["three"][1]).abs().max())
    close(res["planes"][0], res["three"][0], atol=1e-7, rtol=1e-6, what="sums")


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_gradient_means_from_the_loss_kernel(monkeypatch, precision):
    """The adjoint of the network's spatial-mean subtraction takes the means of the loss gradient from the loss kernel's
    per-block sums (mc_partial_sums_finalize) instead of a pass over the gradient (mc_sum_hw): same flat gradient."""
    from pbml_mantle_convection_amd import multigpu
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    grads = []
    for on in (True, False):
        monkeypatch.setattr(multigpu, "GSUM_FROM_LOSS", on)
        torch.manual_seed(5)
        m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
        tr = multigpu.Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
                              loss_type="mass", lambda_mom=1e-6, precision=precision)
        b = [t.to(DEV) for t in synthetic_batch(3, 48, 70, 11, p_pred=True, device="cpu")]
        tr._fwd_bwd(b[0], b[1], b[4], b[3], b[2])
        assert (tr.loss.gradient_sums() is not None)
        grads.append(tr.flat.grad.clone())
    rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
    assert rel < (1e-5 if precision == "fp32" else 2e-3), rel


def test_fused_adam_matches_torch():
    from pbml_mantle_convection_amd import _lib as L
    n = 10007
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=2e-3, weight_decay=1e-2)
    npad = (n + 3) // 4 * 4
    p = torch.zeros(npad, device=DEV); p[:n] = p0.to(DEV)
    m = torch.zeros(npad, device=DEV); v = torch.zeros(npad, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    lr = torch.full((1,), 2e-3, device=DEV)
    for it in range(5):
        gr = torch.randn(n, generator=g)
        ref_p.grad = gr.clone()
        opt.step()
        gd = torch.zeros(npad, device=DEV); gd[:n] = (gr * 4.0).to(DEV)       # grad_scale 1/4 undoes the x4
        L.call("mc_adam_step_flat", L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), n, L.ptr(lr), 0.9, 0.999, 1e-8, 1e-2, 0.25,
               L.ptr(step), L.stream())
    close(p[:n], ref_p.detach(), atol=1e-6, rtol=1e-5, what="adam params")
    assert int(step.item()) == 5


@pytest.mark.parametrize("tag", ["mass", "curl"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_training_steps_golden(golden, tag, use_graph):
    """The whole fused step on the GPU vs the reference's two steps (f32 mode)."""
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden(f"g11_train_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm, ls, ld = [int(v) for v in g["cfg"]]
    B, H, W = 2, 128, 506
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "reflect", str(g["loss_type"]), use_symm=bool(symm),
             repeats=repeats, f=f, p_pred=bool(p_pred))
    m.load_state_dict({k[4:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd0/")})
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=bool(p_pred), network="unet",
                 loss_scale=bool(ls), loss_derivative=bool(ld), loss_type=str(g["loss_type"]), precision="fp32",
                 use_graph=use_graph)
    for step in range(2):
        gVTp = dev(fields.unet_input(B, H, W, 1100 + step, c_i=11 if p_pred else 10))
        truth = [fields.smooth_field(B, H, W, 1150 + step), fields.smooth_field(B, H, W, 1160 + step)]
        if p_pred:
            truth.append(fields.smooth_field(B, H, W, 1170 + step, amp=0.5))
        truth.append(fields.temperature_field(B, H, W, 1180 + step))
        uvp = dev(np.stack(truth, 1))
        if step == 0 and not use_graph:
            out8 = tr._fwd_bwd(gVTp, uvp, None, None, None, train=True)
            for n, gr in tr.flat.views(tr.flat.grad).items():
                ref = g["grad0/" + n]
                close(gr, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)
            tr._sync_lr()
            tr._optim_step()
            vals = out8[:6].tolist()
        else:
            vals = tr._run_batch(gVTp, uvp, None, True)
        close(np.array(vals), g["losses"][step], atol=1e-6, rtol=5e-5, what=f"losses step {step}")
    for n, p in m.named_parameters():
        if float(np.abs(g["grad0/" + n]).max()) < 1e-9:
            # null direction (the last conv's bias is cancelled by the spatial mean subtraction): its gradient is
            # pure rounding noise, which Adam normalises to +-lr whatever its size -> not comparable
            continue
        close(p, g["sd2/" + n], atol=1e-4, rtol=1e-4, what="param " + n)   # Adam moves each weight by <= 1e-3 per step


def test_get_loss_autograd_path(golden):
    """Reference-style usage: loss6 = trainer.get_loss(...); loss.backward(); torch optimizer step."""
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden("g11_train_mass")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm, ls, ld = [int(v) for v in g["cfg"]]
    B, H, W = 2, 128, 506
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=repeats, f=f,
             p_pred=True)
    m.load_state_dict({k[4:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd0/")})
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
                 loss_type="mass", precision="fp32")
    gVTp = dev(fields.unet_input(B, H, W, 1100, c_i=11))
    truth = [fields.smooth_field(B, H, W, 1150), fields.smooth_field(B, H, W, 1160),
             fields.smooth_field(B, H, W, 1170, amp=0.5), fields.temperature_field(B, H, W, 1180)]
    for p in m.parameters():
        p.grad = None
    loss6 = tr.get_loss(gVTp, dev(np.stack(truth, 1)), None)
    close(torch.stack([v.detach() for v in loss6]), g["losses"][0], atol=1e-6, rtol=5e-5, what="loss6")
    loss6[0].backward()
    for n, p in m.named_parameters():
        ref = g["grad0/" + n]
        close(p.grad, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)


@pytest.mark.parametrize("path", ["fwd_bwd", "graph", "get_loss"])
def test_roll_forward_2_golden(golden, path):
    """Trainer with roll_forward = 2 (reference :207-248: 2 x 2 chained network evaluations, the viscosity channel re-derived
    after the pre-steps) against the reference's own six losses and parameter gradients, on the raw-launch path, inside a
    captured step and through the autograd-visible get_loss."""
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden("g21_get_loss_roll2")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm, ls, ld, R = [int(v) for v in g["cfg"]]
    B, H, W = 1, 128, 506
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "reflect", "mass", use_symm=bool(symm), repeats=repeats, f=f,
             p_pred=bool(p_pred))
    m.load_state_dict({k[4:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd0/")})
    opt = torch.optim.Adam(m.parameters(), lr=0.0)                         # lr 0: the captured step leaves the weights alone
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=bool(p_pred), network="unet",
                 loss_type="mass", roll_forward=R, precision="fp32", use_graph=path == "graph")
    gVTp = dev(fields.unet_input(B, H, W, 2100, c_i=11))
    truth = [fields.smooth_field(B, H, W, 2150), fields.smooth_field(B, H, W, 2160), fields.smooth_field(B, H, W, 2170, amp=0.5),
             fields.temperature_field(B, H, W, 2180)]
    uvp, paras = dev(np.stack(truth, 1)), dev(g["paras"]).view(B, 3, 1, 1)
    g_in = gVTp.clone()
    if path == "get_loss":
        for p_ in m.parameters():
            p_.grad = None
        loss6 = tr.get_loss(gVTp, uvp, None, paras, gVTp[:, 1:2])
        loss6[0].backward()
        vals = torch.stack([v.detach() for v in loss6])
        grads = {n: p_.grad for n, p_ in m.named_parameters()}
    else:
        runs = 2 if path == "graph" else 1                                # the second call replays the captured chain
        for _ in range(runs):
            out8 = tr.train_step(gVTp, uvp, None, paras, None) if path == "graph" else tr._fwd_bwd(gVTp, uvp, None, paras, None)
        vals = out8[:6]
        grads = tr.flat.views(tr.flat.grad)
    assert torch.equal(gVTp, g_in), "the batch itself must not be written"
    close(vals, g["losses"], atol=1e-6, rtol=1e-4, what="losses")
    for n, gr in grads.items():
        ref = g["grad0/" + n]
        close(gr, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)


@pytest.mark.parametrize("loss_type,R", [("curl", 2), ("curl", 3), ("mae", 3)])
def test_roll_forward_matches_oracle(loss_type, R):
    """R x R chain with the curl head between the evaluations (u, v from the streamfunction, T clipped) and R = 3 (two
    pre-steps per round), two samples with different (FKT, FKP): losses and gradients against the oracle's chain."""
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    B, H, W = 2, 40, 72
    torch.manual_seed(17)
    m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", loss_type, use_symm=True, repeats=2, f=5, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type=loss_type,
                 roll_forward=R, precision="fp32")
    gVTp = fields.unet_input(B, H, W, 2300, c_i=11)
    truth = [fields.smooth_field(B, H, W, 2350), fields.smooth_field(B, H, W, 2360), fields.smooth_field(B, H, W, 2370, amp=0.5),
             fields.temperature_field(B, H, W, 2380)]
    uvp = np.stack(truth, 1)
    paras = np.array([[5.0, 1.0e7, 10.0], [3.0, 1.0e3, 100.0]])
    out8 = tr._fwd_bwd(dev(gVTp), dev(uvp), None, dev(paras), None)
    sd = {k: v.detach().double().cpu().requires_grad_(True) for k, v in m.state_dict().items()}
    fwd = lambda x: O.unet_forward(sd, x, levels=3, repeats=2, act="gelu", r_p="reflect", loss_type=loss_type,  # noqa: E731
                                   use_symm=True, p_pred=True)
    pred = O.unet_roll_forward(fwd, torch.from_numpy(gVTp).to(f64), torch.from_numpy(paras).to(f64), R)
    ref = O.get_loss_unet(pred, torch.from_numpy(uvp).to(f64), p_pred=True, loss_type=loss_type)
    close(out8[:6], np.array([float(o.detach()) for o in ref[:6]]), atol=1e-6, rtol=2e-4, what="losses")
    ref[0].backward()
    for n, gr in tr.flat.views(tr.flat.grad).items():
        r_ = sd[n].grad
        if float(r_.abs().max()) < 1e-9:          # null direction (last bias: cancelled by the spatial mean subtraction)
            continue
        close(gr, r_, atol=1e-3 * max(1e-6, float(r_.abs().max())), rtol=5e-3, what="grad " + n)


def test_roll_forward_needs_paras():
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                 roll_forward=2, precision="fp32")
    b = [t.to(DEV) for t in synthetic_batch(2, 48, 70, 11, p_pred=True, device="cpu")]
    with pytest.raises(ValueError):
        tr.train_step(b[0], b[1], None, None, None)


def test_graph_step_in_place_input_buffers():
    """A batch written straight into the captured step's input buffers gives the same step as one passed by value."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    res = []
    for in_place in (False, True):
        torch.manual_seed(3)
        m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
        tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
                     loss_type="mass", lambda_mom=0.1, precision="fp32", use_graph=True)
        b0 = [t.to(DEV) for t in synthetic_batch(2, 48, 70, 11, p_pred=True, device="cpu")]
        b1 = [t.to(DEV) for t in synthetic_batch(2, 48, 70, 12, p_pred=True, device="cpu")]
        with pytest.raises(RuntimeError):
            tr.input_buffers()
        order = lambda b: (b[0], b[1], b[4], b[3], b[2])          # (gVTp, uvp, scaler, paras, yc) -> train_step order
        tr.train_step(*order(b0))
        if in_place:
            buf = tr.input_buffers()
            for k, v in zip(("gVTp", "uvp", "yc", "paras", "scaler"), order(b1)):
                buf[k].copy_(v.reshape(buf[k].shape))
            out = tr.train_step(buf["gVTp"], buf["uvp"], buf["yc"], buf["paras"], buf["scaler"])
        else:
            out = tr.train_step(*order(b1))
        res.append((out.clone(), torch.cat([p.detach().flatten() for p in m.parameters()]).clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("use_graph", [False, True])
def test_second_batch_on_a_different_mesh_raises(use_graph):
    """The momentum residual uses ONE depth grid for the whole batch: a LATER batch (not only the first) whose samples carry
    different grids must raise -- also in the captured step, where the check runs on the caller's tensor before the staging
    copy, and also when the caching allocator hands the second batch the first one's address."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    torch.manual_seed(3)
    m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet",
                 loss_type="mass", lambda_mom=0.1, precision="fp32", use_graph=use_graph)
    gVTp, uvp, scaler, paras, yc = [t.to(DEV) for t in synthetic_batch(2, 48, 70, 11, p_pred=True, device="cpu")]
    yc = yc.reshape(-1, 48, 70)[:1].repeat(2, 1, 1).contiguous()          # one grid per sample, as a DataLoader collates them
    tr.train_step(gVTp, uvp, yc, paras, scaler)
    yc_bad = yc.clone()
    ptr = yc_bad.data_ptr()
    yc_bad[1] += 0.01
    with pytest.raises(ValueError, match="one depth grid"):
        tr.train_step(gVTp, uvp, yc_bad, paras, scaler)
    del yc_bad
    yc_ok = yc.clone()                                        # (usually the freed block again)
    tr.train_step(gVTp, uvp, yc_ok, paras, scaler)
    yc_bad2 = yc.clone()
    yc_bad2[0, 3] -= 0.5
    with pytest.raises(ValueError, match="one depth grid"):
        tr.train_step(gVTp, uvp, yc_bad2, paras, scaler)
    assert ptr                                                # (keeps the address in the failure report)


# ---------------------------------------------------------------------------------------------- NewFluidNet
@pytest.mark.parametrize("p_pred", [True, False])
@pytest.mark.parametrize("loss_type", ["mae", "mass", "curl"])
@pytest.mark.parametrize("ls,ld", [(False, False), (True, True)])
def test_fused_loss_fluidnet_mode_matches_oracle(p_pred, loss_type, ls, ld):
    """has_T = False: the FluidNet branch of get_loss (no temperature term, scaled pressure loss, curl head without T)."""
    from pbml_mantle_convection_amd.losses import StokesLoss
    B, H, W = 2, 37, 53
    u, v, p, _, uvp4 = _case(B, H, W, 300, p_pred)
    uvp = uvp4[:, :-1]                                                  # truth without the temperature channel
    a_bound = 4.0
    if loss_type == "curl":
        psi = fields.smooth_field(B, H, W, 311, noise=0.01)
        chans = [psi] + ([p] if p_pred else [])
    else:
        chans = [u, v] + ([p] if p_pred else [])
    y = dev(np.stack(chans, 1))
    Ls = StokesLoss(p_pred, loss_type, ls, ld, a_bound=a_bound, has_T=False)
    out8, gy = Ls.evaluate(y, dev(uvp))
    ty = torch.from_numpy(np.stack(chans, 1)).to(f64).requires_grad_(True)
    if loss_type == "curl":
        cu, cv = O.curl_head(ty[:, 0:1] * a_bound)
        pred = (cu[:, 0], cv[:, 0], ty[:, 1] if p_pred else None)
    else:
        pred = (ty[:, 0], ty[:, 1], ty[:, 2] if p_pred else None)
    ref = O.get_loss_fluidnet(pred, torch.from_numpy(uvp).to(f64), p_pred=p_pred, loss_type=loss_type, loss_scale=ls,
                              loss_derivative=ld)
    close(out8[:6], torch.stack([r.detach() for r in ref]), atol=1e-6, rtol=2e-5, what="loss6")
    ref[0].backward()
    close(gy, ty.grad, atol=2e-8, rtol=2e-4, what="grad")       # (f32 sums of the curl adjoint: 1e-7 of the largest entry)


def test_newfluidnet_training_step_vs_oracle():
    """One fused training step of a small NewFluidNet (Trainer, network='newfluidnet') against the same step on the CPU
    oracle: loss tuple and updated weights."""
    from pbml_mantle_convection_amd.multigpu import Trainer, build_model
    torch.manual_seed(4)
    cfg = dict(levels=3, repeats=2, act="gelu", r_p="zeros", loss_type="mass", use_symm=True, p_pred=True)
    m = build_model("newfluidnet", 3, 7, 8, 3, torch.device("cpu"), "gelu", "zeros", "mass", True, 2, 5, p_pred=True)
    sd = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    B, H, W = 2, 64, 90
    x = torch.from_numpy(fields.unet_input(B, H, W, 401, c_i=7)).float()
    uvp = torch.from_numpy(np.stack([fields.smooth_field(B, H, W, 402), fields.smooth_field(B, H, W, 403),
                                     fields.smooth_field(B, H, W, 404, amp=0.5)], 1)).float()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[100], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="newfluidnet",
                 loss_scale=True, loss_derivative=False, loss_type="mass", precision="fp32")
    out8 = tr.train_step(x.to(DEV), uvp.to(DEV))
    # the same step on the CPU oracle
    ropt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    pred = O.newfluidnet_forward(sd, x.double(), levels=3, repeats=2, act="gelu", r_p="zeros", loss_type="mass",
                                 use_symm=True, p_pred=True)
    ref = O.get_loss_fluidnet((pred[0], pred[1], pred[2]), uvp.double(), p_pred=True, loss_type="mass", loss_scale=True)
    ref[0].backward()
    ropt.step()
    close(out8[:6], torch.stack([r.detach() for r in ref]), atol=1e-6, rtol=5e-5, what="loss6")
    for n, p in m.named_parameters():
        if n == "conv.3.bias":
            continue                                                    # null direction (see test_two_training_steps_golden)
        close(p, sd[n].detach(), atol=1e-4, rtol=1e-4, what="param " + n)


def test_resident_dataset_assembles_batches_on_device():
    """batches built by mc_assemble_adtime_batch from HBM-resident fields equal the host items of
    ADTimeDataset.__getitem__ (the mirror of datasetio.py:229-280) stacked."""
    from pbml_mantle_convection_amd.datasetio import ADTimeDataset, ResidentADTimeDataset, normalise_parameters
    H, W, M = 24, 40, 19
    ds = ADTimeDataset.__new__(ADTimeDataset)                                   # no files: fill the members synthetically
    g = torch.Generator().manual_seed(9)
    ds.x_data = [torch.rand((1, H, W), generator=g, dtype=torch.float64) for _ in range(M)]
    ds.y_data = [torch.randn((3, H, W), generator=g, dtype=torch.float64) for _ in range(M)]
    ds.t = [0.01 * i + 0.001 * float(torch.rand(1, generator=g)) for i in range(M)]
    ds.t_data = [torch.tensor(t, dtype=torch.float64) for t in ds.t]
    par = [(2.5, 1e7, 30.0), (7.0, 1e9, 3.0)]
    ds.paras = [torch.tensor(par[i >= 10], dtype=torch.float64).view(3, 1, 1) for i in range(M)]
    ds.paras_nd = [torch.tensor(normalise_parameters(*par[i >= 10]), dtype=torch.float64).view(3, 1, 1) for i in range(M)]
    ds.xc = torch.linspace(0, 4, W, dtype=torch.float64).view(1, 1, W).expand(1, H, W).contiguous()
    ds.yc = torch.linspace(0, 1, H, dtype=torch.float64).view(1, H, 1).expand(1, H, W).contiguous()
    ds.indices = [[i, i + 1] for i in range(M - 1) if i != 9]
    ds.indices_init = [[0, 1], [10, 11]]
    ds.scale, ds.p_pred, ds.noise = True, True, 0.0
    ds.num_examples = len(ds.indices)
    rd = ResidentADTimeDataset(ds, DEV)
    idx = [k for k, (i0, _) in enumerate(ds.indices) if i0 % 8 != 0][:6]         # (i0 % 8 == 0 items are randomised)
    x, y, sc, pa, yc = rd.assemble(idx)
    items = [ds[k] for k in idx]
    close(x, torch.stack([it[0] for it in items]), atol=2e-6, rtol=2e-6, what="x")
    close(y, torch.stack([it[1] for it in items]), atol=2e-6, rtol=2e-6, what="y")
    close(sc, torch.stack([it[2].reshape(()) for it in items]), atol=0, rtol=2e-6, what="scaler")
    close(pa.reshape(-1, 3), torch.stack([it[3].reshape(3) for it in items]), atol=0, rtol=1e-7, what="paras")
    close(yc, items[0][4], atol=1e-7, rtol=0, what="yc")
    # in-place into preallocated buffers (what a training loop hands over: Trainer.input_buffers())
    out = dict(gVTp=torch.zeros_like(x), uvp=torch.zeros_like(y), scaler=torch.zeros_like(sc), paras=torch.zeros((len(idx), 3), device=DEV))
    rd.assemble(idx, out=out)
    assert torch.equal(out["gVTp"], x) and torch.equal(out["uvp"], y)
    # randomised initial-condition substitution keeps to the init pairs
    assert all(tuple(p) in [(0, 1), (10, 11)] for p in rd.pairs([k for k, (i0, _) in enumerate(ds.indices) if i0 % 8 == 0]))


def test_bf16_training_tracks_fp32_training():
    """Sixty fused Adam steps (lr 5e-4) on one synthetic batch: the loss falls in both modes and the bf16 run (polynomial GELU,
    bf16 activations and gradients) ends near the fp32 run.  The step is bit-reproducible (test_training_step_is_deterministic),
    so this comparison has one outcome, not a distribution."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    batch = [t.to(DEV) for t in synthetic_batch(4, 64, 122, 21, p_pred=True, device="cpu")]
    gVTp, uvp, scaler, paras, yc = batch
    hist = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(11)
        m = Unet(4, 10, 16, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
        opt = torch.optim.Adam(m.parameters(), lr=5e-4)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1000], gamma=0.5)
        tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                     precision=prec, use_graph=True)
        hist[prec] = [float(tr.train_step(gVTp, uvp, yc, paras, scaler)[0]) for _ in range(60)]
    for prec, h in hist.items():
        assert min(h[-10:]) < 0.9 * h[0], (prec, h[0], h[-10:])
        assert all(np.isfinite(h))
    # medians of the last ten steps (single steps of this chaotic toy problem jump by 10-20 %)
    a, b = float(np.median(hist["bf16"][-10:])), float(np.median(hist["fp32"][-10:]))
    assert abs(a - b) <= 0.20 * b, (a, b)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_training_step_is_deterministic(prec):
    """Two runs of the same three fused training steps (HIP-graph replay, momentum term on, several tiles / partial slabs per
    layer) leave bit-identical parameters and Adam moments: every reduction of the step is ordered (filter gradients: slab
    combine; GroupNorm parameter gradients: per-sample sums added in sample order; no float atomics on the training state)."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    batch = [t.to(DEV) for t in synthetic_batch(6, 80, 150, 5, p_pred=True, device="cpu")]
    gVTp, uvp, scaler, paras, yc = batch
    res = []
    for _ in range(2):
        torch.manual_seed(3)
        m = Unet(3, 10, 16, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1000], gamma=0.5)
        tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                     lambda_mom=1e-6, precision=prec, use_graph=True)
        for _ in range(3):
            tr.train_step(gVTp, uvp, yc, paras, scaler)
        torch.cuda.synchronize()
        res.append((tr.flat.param.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.flat.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b), float((a - b).abs().max())


def test_eval_with_another_batch_size_after_capture():
    """A captured training graph pins the engine's buffers; evaluation on a batch of another size must neither corrupt the next
    replay nor fail (it runs on an engine of its own), and re-planning the pinned engine raises instead of freeing memory."""
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    b4 = [t.to(DEV) for t in synthetic_batch(4, 48, 70, 5, p_pred=True, device="cpu")]
    b2 = [t.to(DEV) for t in synthetic_batch(2, 48, 70, 6, p_pred=True, device="cpu")]

    def make():
        torch.manual_seed(3)
        m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1000], gamma=0.5)
        return Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                       precision="fp32", use_graph=True)

    def args(b):
        g, u, sc, pa, yc = b
        return g, u, yc, pa, sc

    ref = make()
    for _ in range(3):
        ref.train_step(*args(b4))
    tr = make()
    tr.train_step(*args(b4))
    e2 = tr.eval_step(*args(b2)).clone()                     # other batch size, between two replays
    tr.train_step(*args(b4))
    e4 = tr.eval_step(*args(b4)).clone()                     # captured size: the pinned engine itself
    tr.train_step(*args(b4))
    torch.cuda.synchronize()
    assert torch.equal(tr.flat.param, ref.flat.param)
    assert torch.isfinite(e2).all() and torch.isfinite(e4).all()
    with pytest.raises(RuntimeError, match="pinned"):
        tr.model_uvp.engine().configure(2, 48, 70, torch.device(DEV))
