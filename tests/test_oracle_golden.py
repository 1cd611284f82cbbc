"""Pins the CPU oracle (oracle/ref_cpu.py) against vectors captured from the imported
reference by tools/make_golden.py.  CPU only."""
import numpy as np
import pytest
import torch

import fields
from oracle import ref_cpu as O

f64 = torch.float64
TOL = dict(rtol=1e-9, atol=1e-11)


def T(a, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(f64)
    return t.requires_grad_(True) if grad else t


def sd_from(g, prefix="sd/", grad=True):
    return {k[len(prefix):]: T(g[k], grad and g[k].dtype.kind == "f" and "num_batches" not in k)
            for k in g.files if k.startswith(prefix)}


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), **tol)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_symmetric_conv(golden, tag):
    g = golden(f"g1{tag}_symconv")
    ci, co, k, h, v, hv = g["meta"]
    x, w, b = T(g["x"], True), T(g["w"], True), T(g["b"], True)
    sym = {"h": int(h), "v": int(v), "hv": int(hv)}
    assert w.shape[0] == O.unique_filters(int(co), sym)
    y = O.conv2d_same(x, O.expand_symmetric_weight(w, sym), b, str(g["mode"]))
    close(y, g["y"])
    (y * T(g["ct"])).sum().backward()
    close(x.grad, g["dx"])
    close(w.grad, g["dw"])
    close(b.grad, g["db"])


def test_symmetric_mirror_property(golden):
    """output channel U+i (bias removed) equals the x-flip of channel i applied to
    the x-flipped input."""
    g = golden("g1a_symconv")
    x, w = T(g["x"]), T(g["w"])
    sym = {"h": 4, "v": 0, "hv": 0}
    y = O.conv2d_same(x, O.expand_symmetric_weight(w, sym), None, "reflect")
    yf = O.conv2d_same(x.flip(3), O.expand_symmetric_weight(w, sym), None, "reflect").flip(3)
    U = w.shape[0]
    close(y[:, U:U + 2], yf[:, 0:2])


@pytest.mark.parametrize("tag", list("abcdef"))
def test_fluid_layer(golden, tag):
    g = golden(f"g2{tag}_fluidlayer")
    ci, co, k, symm = g["meta"]
    sd = sd_from(g)
    x = T(g["x"], True)
    y = O.fluid_layer(sd, "", x, str(g["act"]), str(g["mode"]), bool(symm))
    close(y, g["y"])
    (y * T(g["ct"])).sum().backward()
    close(x.grad, g["dx"])
    for k_ in sd:
        close(sd[k_].grad, g["grad/" + k_], atol=1e-10)


def test_resampling(golden):
    import torch.nn.functional as F
    for name, fn in [("g3a_bicubic_size", lambda x: F.interpolate(x, size=(63, 64), mode="bicubic")),
                     ("g3b_bicubic_x4", lambda x: F.interpolate(x, scale_factor=4, mode="bicubic")),
                     ("g3c_avgpool2", lambda x: F.avg_pool2d(x, 2, 2)),
                     ("g3d_avgpool4", lambda x: F.avg_pool2d(x, 4, 4))]:
        g = golden(name)
        x = T(g["x"], True)
        y = fn(x)
        close(y, g["y"])
        (y * T(g["ct"])).sum().backward()
        close(x.grad, g["dx"])


@pytest.mark.parametrize("tag", ["curl", "mae", "mass_rep", "mae_zeros"])
def test_unet(golden, tag):
    g = golden(f"g4_unet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    sd = sd_from(g)
    tab = O.unet_layer_table(levels, c_i, c_h, c_o, repeats)
    assert len(tab) == sum(1 for k in sd if k.endswith("weight") and sd[k].dim() == 4)
    x = T(fields.unet_input(2, 40, 54, 41, c_i=c_i), True)
    outs = O.unet_forward(sd, x, levels=levels, repeats=repeats, act=str(g["act"]), r_p=str(g["r_p"]),
                          loss_type=str(g["loss_type"]), use_symm=bool(symm), p_pred=bool(p_pred))
    loss = 0.0
    for n, o in zip("uvpT", outs):
        if o is None:
            assert "out/" + n not in g.files
            continue
        close(o, g["out/" + n])
        loss = loss + (o * T(g["ct/" + n])).sum()
    loss.backward()
    close(fields.strided_sample(x.grad.numpy(), 1021), g["dx_sample"], atol=1e-10)
    for k_ in sd:
        close(sd[k_].grad, g["grad/" + k_], atol=1e-9)


@pytest.mark.parametrize("tag", ["mae_zeros", "curl_rep"])
def test_newfluidnet(golden, tag):
    """the multi-resolution trunk (pooled copies of one feature map, bicubic back to 128 x 506, six-way
    concat with the raw inputs, 3 x 3 head) against the imported reference, outputs and every gradient."""
    g = golden(f"g12_newfluidnet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    sd = sd_from(g)
    tab = O.newfluidnet_layer_table(levels, c_i, c_h, c_o, repeats)
    assert len(tab) == sum(1 for k in sd if k.endswith("weight") and sd[k].dim() == 4)
    x = T(fields.unet_input(1, 128, 506, 121, c_i=c_i), True)
    outs = O.newfluidnet_forward(sd, x, levels=levels, repeats=repeats, act=str(g["act"]), r_p=str(g["r_p"]),
                                 loss_type=str(g["loss_type"]), use_symm=bool(symm), p_pred=bool(p_pred))
    loss = 0.0
    for n, o in zip("uvp", outs):
        close(o, g["out/" + n])
        loss = loss + (o * T(g["ct/" + n])).sum()
    loss.backward()
    close(fields.strided_sample(x.grad.numpy(), 1021), g["dx_sample"], atol=1e-10)
    for k_ in sd:
        close(sd[k_].grad, g["grad/" + k_], atol=1e-9)


@pytest.mark.parametrize("tag", ["mae", "curl"])
def test_convae(golden, tag):
    g = golden(f"g5_convae_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    sd = sd_from(g)
    x = T(g["x"], True)
    y = O.convae_forward(sd, x, levels=levels, c_i=c_i, c_h=c_h, c_o=c_o, repeats=repeats, act="gelu",
                         r_p=str(g["r_p"]), loss_type=str(g["loss_type"]), use_symm=bool(symm),
                         p_pred=bool(p_pred))
    close(y, g["y"])
    (y * T(g["ct"])).sum().backward()
    close(x.grad, g["dx"], atol=1e-10)
    for k_ in sd:
        close(sd[k_].grad, g["grad/" + k_], atol=1e-9)


def test_get_loss_table(golden):
    g = golden("g6_get_loss")
    B, H, W = 2, 128, 506
    for case, row in enumerate(g["table"]):
        p_pred, lt, ls, ld, seed = [int(v) for v in row[:5]]
        loss_type = ["mae", "mass", "curl"][lt]
        u = T(fields.smooth_field(B, H, W, seed + 1, noise=0.01), True)
        v = T(fields.smooth_field(B, H, W, seed + 2, noise=0.01), True)
        p = T(fields.smooth_field(B, H, W, seed + 3, amp=0.5), True)
        Tt = T(fields.temperature_field(B, H, W, seed + 4), True)
        truth = [fields.smooth_field(B, H, W, seed + 5), fields.smooth_field(B, H, W, seed + 6)]
        if p_pred:
            truth.append(fields.smooth_field(B, H, W, seed + 7, amp=0.5))
        truth.append(fields.temperature_field(B, H, W, seed + 8))
        uvp = T(np.stack(truth, 1))
        out = O.get_loss_unet((u, v, p if p_pred else None, Tt), uvp, p_pred=bool(p_pred),
                              loss_type=loss_type, loss_scale=bool(ls), loss_derivative=bool(ld))
        close(torch.stack([o.detach() for o in out]), row[5:], rtol=1e-10)
        out[0].backward()
        close(fields.strided_sample(u.grad.numpy()), g[f"du/{case}"], atol=1e-14)
        close(fields.strided_sample(v.grad.numpy()), g[f"dv/{case}"], atol=1e-14)
        close(fields.strided_sample(Tt.grad.numpy()), g[f"dT/{case}"], atol=1e-14)
        if p_pred:
            close(fields.strided_sample(p.grad.numpy()), g[f"dp/{case}"], atol=1e-14)


def test_get_loss_fluidnet_table(golden):
    """the FluidNet branch of get_loss (no T term, scaled pressure loss), 24 cases with gradients."""
    g = golden("g13_get_loss_fluidnet")
    B, H, W = 2, 128, 506
    for case, row in enumerate(g["table"]):
        p_pred, lt, ls, ld, seed = [int(v) for v in row[:5]]
        u = T(fields.smooth_field(B, H, W, seed + 1, noise=0.01), True)
        v = T(fields.smooth_field(B, H, W, seed + 2, noise=0.01), True)
        p = T(fields.smooth_field(B, H, W, seed + 3, amp=0.5), True)
        truth = [fields.smooth_field(B, H, W, seed + 5), fields.smooth_field(B, H, W, seed + 6)]
        if p_pred:
            truth.append(fields.smooth_field(B, H, W, seed + 7, amp=0.5))
        uvp = T(np.stack(truth, 1))
        out = O.get_loss_fluidnet((u, v, p if p_pred else None), uvp, p_pred=bool(p_pred),
                                  loss_type=["mae", "mass", "curl"][lt], loss_scale=bool(ls), loss_derivative=bool(ld))
        close(torch.stack([o.detach() for o in out]), row[5:], rtol=1e-10)
        out[0].backward()
        close(fields.strided_sample(u.grad.numpy()), g[f"du/{case}"], atol=1e-14)
        close(fields.strided_sample(v.grad.numpy()), g[f"dv/{case}"], atol=1e-14)
        if p_pred:
            close(fields.strided_sample(p.grad.numpy()), g[f"dp/{case}"], atol=1e-14)


def test_loss_fn(golden):
    g = golden("g6b_loss_fn")
    for ls in (0, 1):
        a, b = O.loss_fn(T(g["x_true"]), T(g["x_pred"]), bool(ls))
        close(a, g[f"scaled_{ls}"])
        close(b, g[f"plain_{ls}"])


def test_get_mass(golden):
    g = golden("g7_get_mass")
    B, H, W = 2, 128, 506
    u = T(fields.smooth_field(B, H, W, 700, noise=0.01))
    v = T(fields.smooth_field(B, H, W, 701, noise=0.01))
    for bc in (0, 1):
        m = O.get_mass(u, v, bc=bool(bc))
        close(fields.strided_sample(m.numpy()), g[f"sample_bc{bc}"])
        close(m.sum(), g[f"sum_bc{bc}"])
        close(m.abs().sum(), g[f"abssum_bc{bc}"])
        close(m[0, 0, :, 0], g[f"edge_bc{bc}"])


def test_fd_kernels(golden):
    g = golden("g8_fd_kernels")
    x = T(g["x"])
    for name in ("dx_right", "dx_left", "dy_bot", "dy_top", "dx_center", "dy_center", "du_dy", "dv_dx",
                 "laplace"):
        close(getattr(O, name)(x), g[name])


def test_helpers(golden):
    g = golden("g9_helpers")
    close(O.eta_torch(T(g["gamma"]), T(g["beta"]), T(g["z"]), T(g["T"])), g["eta"])
    pu, pv, pp = O.pad_uvp(T(g["u"]), T(g["v"]), T(g["p"]))
    close(pu, g["pu"]); close(pv, g["pv"]); close(pp, g["pp"])
    close(O.pad_grad(T(g["g"]), (1, 2, 1, 2)), g["pg"])
    ones = np.ones((2, 3))
    close(O.scale_var(ones, 4.21479129, 86422511.6, 3.01635241, "uprev"), g["scale_u"])
    close(O.unscale_var(ones, 4.21479129, 86422511.6, 3.01635241, "vprev"), g["unscale_v"])
    close(O.scale_var(ones, 4.21479129, 86422511.6, 3.01635241, "pprev"), g["scale_p"])
    # SURVEY §4 known answer
    np.testing.assert_allclose(g["scale_u"][0, 0], 5.69095669e-05, rtol=1e-8)


def test_known_parameter_counts(golden):
    g = golden("g10_known_answers")
    assert int(g["newfluidnet"]) == 2289281          # load_fluidnet.ipynb:365
    assert int(g["unet_cfg2"]) == 1820030
    assert int(g["convae_cfg1"]) == 860301

    def count_unet(levels, c_i, c_h, c_o, repeats, k, symm):
        n = 0
        for _, cin, cout, kind in O.unet_layer_table(levels, c_i, c_h, c_o, repeats):
            u = O.unique_filters(cout, O.symmetry_counts(cout)) if (symm and kind == "fluid") else cout
            n += u * cin * k * k + cout
            if kind in ("fluid", "head_gn"):
                n += 2 * cout
        return n
    assert count_unet(5, 11, 16, 4, 3, 5, True) == 1820030

    def count_convae(levels, c_i, c_h, c_o, repeats, k, symm):
        n = 0
        for op in O.convae_op_table(levels, c_i, c_h, c_o, repeats):
            if op[0] == "fluid":
                _, _, cin, cout = op
                u = O.unique_filters(cout, O.symmetry_counts(cout)) if symm else cout
                n += u * cin * k * k + cout + 2 * cout
            elif op[0] == "final":
                n += op[3] * op[2] * 9 + op[3]
        return n
    assert count_convae(2, 3, 16, 3, 2, 3, True) == 860301
    assert count_convae(2, 3, 16, 3, 2, 3, False) == int(g["convae_cfg1_plain"])


@pytest.mark.parametrize("tag", ["mass", "curl"])
def test_two_training_steps(golden, tag):
    """zero_grad -> get_loss -> backward -> Adam twice (multigpu.py:307-320) vs the reference."""
    g = golden(f"g11_train_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm, ls, ld = [int(v) for v in g["cfg"]]
    B, H, W = 2, 128, 506
    sd0 = {k[4:]: T(g[k]) for k in g.files if k.startswith("sd0/")}
    cfg = dict(levels=levels, repeats=repeats, act="gelu", r_p="reflect", loss_type=str(g["loss_type"]),
               use_symm=bool(symm), p_pred=bool(p_pred))
    stepper = O.CpuUnetStep(sd0, cfg, lr=1e-3)
    for step in range(2):
        gVTp = T(fields.unet_input(B, H, W, 1100 + step, c_i=11 if p_pred else 10))
        truth = [fields.smooth_field(B, H, W, 1150 + step), fields.smooth_field(B, H, W, 1160 + step)]
        if p_pred:
            truth.append(fields.smooth_field(B, H, W, 1170 + step, amp=0.5))
        truth.append(fields.temperature_field(B, H, W, 1180 + step))
        uvp = T(np.stack(truth, 1))
        if step == 0:
            stepper.opt.zero_grad()
            out, _ = stepper.loss(gVTp, uvp, loss_scale=bool(ls), loss_derivative=bool(ld))
            out[0].backward()
            for k_, p_ in stepper.sd.items():
                close(p_.grad, g["grad0/" + k_], atol=1e-10, rtol=1e-7)
            stepper.opt.step()
            vals = [float(o.detach()) for o in out]
        else:
            vals = stepper.step(gVTp, uvp, loss_scale=bool(ls), loss_derivative=bool(ld))
        close(np.array(vals), g["losses"][step], rtol=1e-8)
    for k_, p_ in stepper.sd.items():
        close(p_, g["sd2/" + k_], rtol=1e-7, atol=1e-9)


def test_get_loss_roll_forward_2(golden):
    """get_loss with roll_forward = 2 (multigpu.py:207-248): the oracle's restatement of the R x R chain of network
    evaluations (only the last one differentiated) against the reference's losses and parameter gradients."""
    g = golden("g21_get_loss_roll2")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm, ls, ld, R = [int(v) for v in g["cfg"]]
    B, H, W = 1, 128, 506
    sd = {k[4:]: T(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd0/")}
    gVTp = T(fields.unet_input(B, H, W, 2100, c_i=11))
    truth = [fields.smooth_field(B, H, W, 2150), fields.smooth_field(B, H, W, 2160), fields.smooth_field(B, H, W, 2170, amp=0.5),
             fields.temperature_field(B, H, W, 2180)]
    uvp = T(np.stack(truth, 1))
    fwd = lambda x: O.unet_forward(sd, x, levels=levels, repeats=repeats, act="gelu", r_p="reflect", loss_type="mass",  # noqa: E731
                                   use_symm=bool(symm), p_pred=bool(p_pred))
    pred = O.unet_roll_forward(fwd, gVTp, T(g["paras"]), R)
    out = O.get_loss_unet(pred, uvp, p_pred=bool(p_pred), loss_type="mass")
    close(np.array([float(o.detach()) for o in out[:6]]), g["losses"], rtol=1e-8)
    out[0].backward()
    for k_, p_ in sd.items():
        close(p_.grad, g["grad0/" + k_], atol=1e-10, rtol=1e-7)


def test_momentum_residual_constant_viscosity_identity():
    """Build-defined term: with eta == 1 and p == 0 the flux form must reduce to
    126^2 (laplace(U) + d2U/dx2 + d2V/dxdy), built from the reference's own FD kernels."""
    B, H, W = 2, 12, 15
    u = T(fields.smooth_field(B, H, W, 1)); v = T(fields.smooth_field(B, H, W, 2))
    p = torch.zeros_like(u); Tz = torch.zeros_like(u)
    yc = torch.zeros(H, W, dtype=f64)
    paras = torch.ones(B, 3, dtype=f64)          # FKT = FKP = 1 -> eta = 1
    s = torch.ones(B, dtype=f64)
    rx, ry = O.momentum_residual(u, v, p, Tz, yc, paras, s)
    u4, v4 = u[:, None], v[:, None]
    dxx = (O.dx_right(u4) - O.dx_left(u4))[:, 0, 1:-1]
    dxy_v = 0.5 * (O.dx_center(v4)[:, 0, 2:] - O.dx_center(v4)[:, 0, :-2])
    close(rx, 126.0 ** 2 * (O.laplace(u4)[:, 0] + dxx + dxy_v), rtol=1e-9, atol=1e-9)
    dyy = (O.dy_bot(v4) - O.dy_top(v4))[:, 0, :, 1:-1]
    dxy_u = 0.5 * (O.dy_center(u4)[:, 0, :, 2:] - O.dy_center(u4)[:, 0, :, :-2])
    close(ry, 126.0 ** 2 * (O.laplace(v4)[:, 0] + dyy + dxy_u), rtol=1e-9, atol=1e-9)


def test_momentum_residual_manufactured_solution():
    """Isoviscous Stokes solution on [0,4]x[0,1]: psi = sin(a x) sin(b y), u = dpsi/dy,
    v = -dpsi/dx, T = 0, p chosen so the momentum equations hold; the discrete residual must
    converge at second order."""
    errs = []
    for n in (16, 32, 64):
        H, W = n + 1, 4 * n + 1
        h = 1.0 / n
        y = torch.linspace(0, 1, H, dtype=f64)[:, None].expand(H, W)
        x = torch.linspace(0, 4, W, dtype=f64)[None, :].expand(H, W)
        a, b = 1.3, 2.1
        u = (b * torch.sin(a * x) * torch.cos(b * y))[None]
        v = (-a * torch.cos(a * x) * torch.sin(b * y))[None]
        # lap u = -(a^2+b^2) u ; grad p = lap(u,v) ; p = (a^2+b^2) (b/a) cos(ax) cos(by)
        # check: dp/dy = -(a^2+b^2) (b^2/a) cos(ax) sin(by) vs lap v = (a^2+b^2) a cos sin -> add buoyancy T
        k2 = a * a + b * b
        p = (k2 * (b / a) * torch.cos(a * x) * torch.cos(b * y))[None]
        Tt = (k2 * (a + b * b / a) * torch.cos(a * x) * torch.sin(b * y))[None] * -1.0
        paras = torch.ones(1, 3, dtype=f64)
        rx, ry = O.momentum_residual(u, v, p, Tt, torch.zeros(H, W, dtype=f64), paras,
                                     torch.ones(1, dtype=f64), inv_h=1.0 / h)
        errs.append(float(torch.maximum(rx.abs().max(), ry.abs().max())))
    assert errs[1] < errs[0] / 3.0 and errs[2] < errs[1] / 3.0, errs


def test_bf16_mode_gelu_polynomials_are_within_bf16_rounding():
    """The device's bf16 mode replaces erf by an odd polynomial (csrc/common.h); the error must stay far below the
    bf16 rounding (2^-9 relative) of the tensors it produces, also when evaluated in float32."""
    z = torch.linspace(-12.0, 12.0, 480001, dtype=torch.float64)
    gelu = torch.nn.functional.gelu(z)
    cdf = 0.5 * (1.0 + torch.erf(z / 2.0 ** 0.5))
    grad = cdf + z * torch.exp(-0.5 * z * z) / (2.0 * torch.pi) ** 0.5
    for dt in (torch.float64, torch.float32):
        zz = z.to(dt)
        assert float((O.gelu_bf16_mode(zz).double() - gelu).abs().max()) < 2e-4
        assert float((O.gelu_grad_bf16_mode(zz).double() - grad).abs().max()) < 6e-4
    # exact at the origin, monotone saturation outside the fitted interval
    assert float(O.gelu_bf16_mode(torch.zeros(1))) == 0.0
    assert float(O.gelu_grad_bf16_mode(torch.zeros(1))) == 0.5


def test_mixed_mode_f16_gelu_grad_polynomial_accuracy():
    """The "mixed" mode's fused input-gradient epilogue evaluates GELU' in packed f16 (degree 4 in z^2, |z| <= 3): with every
    operation rounded to f16 the error stays at the level of the bf16 rounding (2^-9) of the dz tensor it produces."""
    z = torch.linspace(-8.0, 8.0, 320001, dtype=torch.float64)
    cdf = 0.5 * (1.0 + torch.erf(z / 2.0 ** 0.5))
    pdf = torch.exp(-0.5 * z * z) / (2.0 * torch.pi) ** 0.5
    grad = cdf + z * pdf
    err = (O.gelu_grad_f16_mode(z) - grad).abs()
    wrms = float(((err ** 2 * pdf).sum() / pdf.sum()).sqrt())              # z ~ N(0, 1): what GroupNorm hands to the activation
    print(f"f16 GELU' polynomial: max |err| {float(err.max()):.2e}, N(0,1)-weighted rms {wrms:.2e}")
    assert float(err.max()) < 1.5e-2 and wrms < 2.5e-3
    assert float(err[z.abs() < 2.0].max()) < 5e-3
    assert float(O.gelu_grad_f16_mode(torch.zeros(1, dtype=torch.float64))) == 0.5


def _n3_grid(H, W):
    xs = np.concatenate(([0.0], (np.arange(W - 2) + 0.5) * 4.0 / (W - 2), [4.0]))
    ys = np.concatenate(([0.0], (np.arange(H - 2) + 0.5) * 1.0 / (H - 2), [1.0]))
    xc = torch.from_numpy(np.broadcast_to(xs[None, :], (H, W)).copy()).view(1, 1, H, W)
    yc = torch.from_numpy(np.broadcast_to(ys[:, None], (H, W)).copy()).view(1, 1, H, W)
    return xc, yc


def _stokes_stub(inp):
    """the deterministic stand-in used by tools/make_golden.py g15"""
    import torch.nn.functional as F
    Tt = inp[:, 6:7]
    a = torch.cumsum(Tt - Tt.mean(), dim=3) * 0.01
    u = F.pad((a[:, :, 2:, 1:-1] - a[:, :, :-2, 1:-1]) * 0.5, (1, 1, 1, 1))
    v = F.pad(-(a[:, :, 1:-1, 2:] - a[:, :, 1:-1, :-2]) * 0.5, (1, 1, 1, 1))
    return u[:, 0], v[:, 0], Tt[:, 0] * 0.0


def test_adnet_step(golden):
    """the explicit upwind advection-diffusion step (ADNet.forward) incl. its CFL time step."""
    g = golden("g14_adnet")
    H, W = 128, 506
    xc, yc = _n3_grid(H, W)
    for k, seed in enumerate(g["seeds"]):
        seed = int(seed)
        u = T(fields.smooth_field(1, H, W, seed + 1)).view(1, 1, H, W) * 400.0
        v = T(fields.smooth_field(1, H, W, seed + 2)).view(1, 1, H, W) * 400.0
        Tp = T(fields.temperature_field(1, H, W, seed + 3)).view(1, 1, H, W)
        inp = torch.cat((u, v, Tp, torch.full((1, 1, H, W), 2.5, dtype=torch.float64), xc, yc), dim=1)
        Tn, dt = O.adnet_step(inp)
        close(dt, g[f"dt/{k}"], rtol=1e-12)
        close(Tn if k == 0 else fields.strided_sample(Tn.numpy(), 4001), g[f"T_next/{k}"], atol=1e-12)
        Tn2, _ = O.adnet_step(inp, dt=torch.tensor(3e-7, dtype=torch.float64))
        close(fields.strided_sample(Tn2.numpy(), 4001), g[f"T_next_fixed/{k}"], atol=1e-12)


def test_ts_rollout(golden):
    """three steps of TS.forward ('newfluidnet' branch: input builder, un-scaling, ADNet, boundary rows)."""
    g = golden("g15_ts_rollout")
    H, W = 128, 506
    xc, yc = _n3_grid(H, W)
    T0 = T(fields.temperature_field(1, H, W, 1500)).view(1, 1, H, W)
    raq, fkt, fkp = (torch.tensor(float(v), dtype=torch.float64) for v in g["paras"])
    nd = [torch.tensor(float(v), dtype=torch.float64).view(1, 1, 1, 1) for v in g["nd"]]
    x, dts, u, v, p, V = O.ts_rollout(_stokes_stub, T0, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc, ts=3)
    close(torch.stack([dts[i] for i in (1, 2, 3)]), g["dts"], rtol=1e-10)
    smp = lambda t: fields.strided_sample(t.numpy(), 4001)  # noqa: E731
    close(smp(x[1]), g["T1"], atol=1e-11)
    close(smp(x[2]), g["T2"], atol=1e-11)
    close(x[3], g["T3"], atol=1e-11)
    close(smp(u), g["u"], atol=1e-11)
    close(smp(v), g["v"], atol=1e-11)
    close(smp(V), g["V"], atol=1e-12)


def test_ts_rollout_unet_branch(golden):
    """TS.forward 'unet' branch (:411-446): the net predicts T itself; pinned with a stub network run through the reference."""
    g = golden("g20_ts_rollout_unet")
    H, W = 128, 506
    xc, yc = _n3_grid(H, W)
    T0 = T(fields.temperature_field(1, H, W, 2000)).view(1, 1, H, W)
    up = T(fields.smooth_field(1, H, W, 2001)).view(1, 1, H, W)
    vp = T(fields.smooth_field(1, H, W, 2002)).view(1, 1, H, W)
    dt = torch.full((1, 1, H, W), float(g["dt"]), dtype=torch.float64)
    _, fkt, fkp = (torch.tensor(float(v), dtype=torch.float64) for v in g["paras"])
    nd = [torch.tensor(float(v), dtype=torch.float64).view(1, 1, 1, 1) for v in g["nd"]]

    def stub(inp):
        Tc, u0, v0, d, V = inp[:, 7], inp[:, 8], inp[:, 9], inp[:, 2], inp[:, 6]
        return u0 * 0.9 + 0.1 * Tc, v0 * 0.8 - 0.05 * Tc, None, Tc + d * (torch.roll(Tc, 1, dims=2) - Tc) * 50.0 + 0.01 * V

    x, dts, u, v, p, V = O.ts_rollout_unet(stub, T0, yc, nd[0], nd[1], nd[2], fkt, fkp, xc, yc, up, vp, dt, ts=3)
    assert p is None and dts == {}
    smp = lambda t: fields.strided_sample(t.numpy(), 4001)  # noqa: E731
    close(smp(x[1]), g["T1"], atol=1e-12)
    close(smp(x[2]), g["T2"], atol=1e-12)
    close(x[3], g["T3"], atol=1e-12)
    close(smp(u), g["u"], atol=1e-12)
    close(smp(v), g["v"], atol=1e-12)
    close(smp(V), g["V"], atol=1e-12)


@pytest.mark.parametrize("tag", ["k5_symm", "k3_plain"])
def test_boundary_learned_conv(golden, tag):
    """the "learned padding" layer (nine valid convolutions framed together) vs the imported reference."""
    g = golden(f"g16_learned_{tag}")
    c_i, c_o, k, symm = [int(v) for v in g["meta"]]
    sd = sd_from(g)
    x = T(g["x"], True)
    y = O.boundary_learned_conv(sd, "", x, k, bool(symm))
    close(y, g["y"])
    (y * T(g["ct"])).sum().backward()
    close(x.grad, g["dx"], atol=1e-10)
    for k_ in sd:
        close(sd[k_].grad, g["grad/" + k_], atol=1e-10)
