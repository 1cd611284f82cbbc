"""GPU parity of the fused ("normalise on load") execution against the unfused chain of the same library.

The consumer of a raw conv output applies the producer's GroupNorm affine + activation while it stages its tile
(mc_conv2d_fused / mc_conv2d_wgrad_fused / mc_bicubic_fwd_act), and the GroupNorm-backward reduction rides in the
input-gradient launch's epilogue (mc_conv_epilogue + mc_fold_padded_dz + mc_gn_bwd_apply_dz).  The stand-alone passes
(mc_gn_act_fwd, mc_gn_act_bwd_reduce / _apply) stay in the library for tensors with several consumers, so the two
executions can be compared directly:

  * forward: the fused prologue evaluates the SAME f32 expressions on the same stored values, so network outputs must be
    bit-identical in bf16 mode and equal to rounding (1e-6) in fp32 mode;
  * backward: fp32 mode to 1e-5 relative L2 per parameter gradient; bf16 mode: the fused gradient is as close to the fp32
    execution's as the unfused one is (error norm <= 2 x the unfused error + 1 % of the gradient norm; the fused path
    rounds dz = dA act'(z) once where the unfused path rounds dA and then dy: a different, not a larger, rounding).
Both executions are separately pinned against the reference's golden vectors by the other GPU test files (the default
execution there is the unfused one, MANTLE_FUSE=0, except for the "mixed" mode's dz epilogue); reference call sites: FluidLayer.forward pytorch_networks_convae.py:790-799,
Unet.forward :1985-2024, ConvAE.forward pycold-checkpoint.py:1094-1115."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _build(kind, r_p, seed):
    from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE, Unet
    torch.manual_seed(seed)
    if kind == "unet":
        return Unet(3, 10, 8, 4, torch.device(DEV), "gelu", r_p, "mae", use_symm=True, repeats=2, f=5, p_pred=True), (2, 10, 44, 70)
    if kind == "unet16":
        return Unet(3, 10, 16, 4, torch.device(DEV), "gelu", r_p, "mae", use_symm=True, repeats=3, f=5, p_pred=True), (2, 10, 70, 90)
    if kind == "unet16w":      # wide enough (150 + 6 + 4 >= 140 columns) for the row-reuse kernel's input-gradient launches
        return Unet(2, 10, 16, 4, torch.device(DEV), "gelu", r_p, "mae", use_symm=True, repeats=3, f=5, p_pred=True), (2, 10, 40, 150)
    return ConvAE(2, 3, 16, 3, torch.device(DEV), "gelu", r_p, "mae", use_symm=True, repeats=2, f=3, p_pred=True), (2, 3, 64, 48)


def _run(monkeypatch, fuse, kind, r_p, precision, seed=3):
    """One forward + backward of a fresh model (same seed -> same weights) with MANTLE_FUSE = fuse."""
    monkeypatch.setenv("MANTLE_FUSE", str(fuse))
    monkeypatch.setenv("MANTLE_FUSE_MAXPIX", str(1 << 30))       # (the default restricts fusion to the low-resolution levels)
    m, shape = _build(kind, r_p, seed)
    m = m.to(DEV)
    with torch.no_grad():                       # non-trivial GroupNorm affine parameters and biases
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in m.named_parameters():
            if n.endswith("layers.1.weight") or n.endswith("gn.0.weight"):
                p.copy_((1.0 + 0.3 * torch.randn(p.shape, generator=g)).to(DEV))
            elif n.endswith("layers.1.bias") or n.endswith("gn.0.bias"):
                p.copy_((0.2 * torch.randn(p.shape, generator=g)).to(DEV))
    m.set_precision(precision)
    assert m.engine().fuse == fuse
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(shape, generator=g).to(DEV)
    y = m._run_graph(x)
    ct = torch.randn(y.shape, generator=g).to(DEV)
    (y * ct).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}, m


@pytest.mark.parametrize("kind,r_p", [("unet", "reflect"), ("unet", "zeros"), ("unet16", "replicate"), ("unet16", "reflect"),
                                      ("convae", "reflect"), ("convae", "zeros"), ("unet16w", "reflect"), ("unet16w", "zeros")])
@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed"])
def test_fused_network_matches_unfused(monkeypatch, kind, r_p, precision):
    monkeypatch.setenv("MANTLE_FUSE_DZ_RR", "0")                 # (the unfused reference run must be unfused in "mixed" too)
    y0, g0, m0 = _run(monkeypatch, 0, kind, r_p, precision)
    y3, g3, m3 = _run(monkeypatch, 3, kind, r_p, precision)
    fused = [t for t in m3.engine().T.values() if t.fused]
    assert fused and not any(t.fused for t in m0.engine().T.values())
    assert any("epi" in e for e in m3.engine().plan), "no input-gradient launch carries the GroupNorm-backward epilogue"
    def check(gk, what):
        if precision == "fp32":
            worst = max((rel_l2(gk[n], g0[n]), n) for n in g0)
            assert worst[0] <= 1e-5, (what, worst)
            return
        # bf16: both executions are roundings of the same gradient; the fp32 execution is the yardstick.  (Relative
        # comparison of the two bf16 results with each other is meaningless for parameters whose true gradient is zero:
        # the bias of a conv in front of GroupNorm.)
        for n in g0:
            e0, ek = float((g0[n] - gref[n]).norm()), float((gk[n] - gref[n]).norm())
            assert ek <= 2.0 * e0 + 1e-2 * float(gref[n].norm()), (what, n, ek, e0, float(gref[n].norm()))

    if precision != "fp32":
        _, gref, _ = _run(monkeypatch, 0, kind, r_p, "fp32")
        assert torch.equal(y0, y3), float((y0 - y3).abs().max())
    else:
        assert float((y0 - y3).abs().max()) <= 1e-6 * max(1.0, float(y0.abs().max()))
    check(g3, 3)
    # forward-only fusion (bit 0) and epilogue-only fusion (bit 1) are valid executions on their own
    for fuse in (1, 2):
        yk, gk, _ = _run(monkeypatch, fuse, kind, r_p, precision)
        if precision != "fp32":
            assert torch.equal(y0, yk)
        check(gk, fuse)
    if precision == "mixed":
        # the "mixed" mode's default: the dz epilogue (packed-f16 GELU') on the row-reuse input-gradient launches only
        monkeypatch.setenv("MANTLE_FUSE_DZ_RR", "1")
        yk, gk, mk = _run(monkeypatch, 0, kind, r_p, precision)
        assert torch.equal(y0, yk)
        assert any("epi" in e for e in mk.engine().plan) == (kind == "unet16w")
        check(gk, "dz_rr")


# ------------------------------------------------------------------------------------------------------------------
# kernel level, straight through the C ABI
# ------------------------------------------------------------------------------------------------------------------
def _cb8(t, dtype):
    """NCHW f32 -> CB8 [N][C8][H][W][8] tensor of `dtype` on the device."""
    n, c, h, w = t.shape
    c8 = (c + 7) // 8
    o = torch.zeros((n, c8 * 8, h, w), dtype=torch.float32)
    o[:, :c] = t
    return o.view(n, c8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous().to(DEV).to(dtype)


def _from_cb8(t, c):
    n, c8, h, w, _ = t.shape
    return t.float().permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)[:, :c].cpu()


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("mode,ci0,ci1,co,k,hw", [("reflect", 16, 0, 16, 5, (37, 45)), ("zeros", 16, 8, 32, 5, (21, 33)),
                                                  ("replicate", 24, 0, 64, 3, (18, 20)), ("reflect", 8, 16, 16, 3, (40, 35))])
def test_conv_prologue_and_filter_gradient_prologue(dt, mode, ci0, ci1, co, k, hw):
    """mc_conv2d_fused / mc_conv2d_wgrad_fused on raw sources == mc_gn_act_fwd followed by the plain entry points
    (source 0: GroupNorm + GELU, source 1: used as it is — the skip / upsampled pair of a U-Net decoder conv)."""
    from pbml_mantle_convection_amd import _lib as L
    lib = L.load()
    mcd, tdt = (L.MC_BF16, torch.bfloat16) if dt == "bf16" else (L.MC_F32, torch.float32)
    N, (H, W) = 2, hw
    g = torch.Generator().manual_seed(ci0 * 100 + co)
    y0 = _cb8(torch.randn((N, ci0, H, W), generator=g) * 1.5 + 0.3, tdt)
    x1 = _cb8(torch.randn((N, ci1, H, W), generator=g), tdt) if ci1 else None
    groups = ci0 // 4
    stats = torch.stack([0.3 + 0.1 * torch.randn((N, groups), generator=g), 0.7 + 0.1 * torch.rand((N, groups), generator=g)], -1).to(DEV)
    gamma = (1 + 0.2 * torch.randn(ci0, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(ci0, generator=g)).to(DEV)
    # the coefficient table exactly as mc_gn_finalize_coef forms it
    cpad = (ci0 + 7) // 8 * 8
    coef = torch.zeros((N, cpad, 4), device=DEV)
    mean_c, rstd_c = stats[..., 0].repeat_interleave(4, 1), stats[..., 1].repeat_interleave(4, 1)
    coef[:, :ci0, 0] = rstd_c * gamma
    coef[:, :ci0, 1] = beta - mean_c * rstd_c * gamma
    coef[:, :ci0, 2], coef[:, :ci0, 3] = mean_c, rstd_c
    st = torch.cuda.current_stream().cuda_stream
    a0 = torch.empty_like(y0)
    L.call("mc_gn_act_fwd", y0.data_ptr(), N, ci0, H, W, groups, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
           L.POST_GN_ACT, L.ACTS["gelu"], 1, mcd, a0.data_ptr(), None, st)
    d = L.ConvDesc(N, H, W, ci0, ci1, co, k, k // 2, L.PAD_MODES[mode], mcd, 0, 0, 0)
    wt = (torch.randn((co, ci0 + ci1, k, k), generator=g) / ((ci0 + ci1) * k * k) ** 0.5).to(DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    bank = torch.empty(lib.mc_packed_weight_bytes(C.byref(d), 0), dtype=torch.uint8, device=DEV)
    L.call("mc_pack_weights", C.byref(d), wt.data_ptr(), 0, bank.data_ptr(), st)
    tiles = lib.mc_conv_tiles(C.byref(d))
    cop = (co + 7) // 8 * 8
    outs, parts = [], []
    for fused in (False, True):
        yo = torch.zeros((N, cop // 8, H, W, 8), dtype=tdt, device=DEV)
        part = torch.zeros((N, tiles, cop, 2), device=DEV)
        if fused:
            pro = L.ConvPrologue(coef.data_ptr(), None, L.ACTS["gelu"], 0)
            L.call("mc_conv2d_fused", C.byref(d), y0.data_ptr(), L.ptr(x1), C.byref(pro), bank.data_ptr(), bias.data_ptr(),
                   yo.data_ptr(), None, part.data_ptr(), None, st)
        else:
            L.call("mc_conv2d", C.byref(d), a0.data_ptr(), L.ptr(x1), bank.data_ptr(), bias.data_ptr(), yo.data_ptr(), None,
                   part.data_ptr(), st)
        outs.append(yo)
        parts.append(part)
    torch.cuda.synchronize()
    if dt == "bf16":
        assert torch.equal(outs[0], outs[1])
        assert torch.equal(parts[0], parts[1])
    else:
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[0].abs().max())
    # filter gradient
    dy = _cb8(torch.randn((N, co, H, W), generator=g), tdt)
    dws = []
    for fused in (False, True):
        wpart = torch.empty(lib.mc_wgrad_partial_bytes(C.byref(d)), dtype=torch.uint8, device=DEV)
        dw, db = torch.zeros_like(wt), torch.zeros_like(bias)
        if fused:
            pro = L.ConvPrologue(coef.data_ptr(), None, L.ACTS["gelu"], 0)
            L.call("mc_conv2d_wgrad_fused", C.byref(d), y0.data_ptr(), L.ptr(x1), C.byref(pro), dy.data_ptr(), wpart.data_ptr(), st)
        else:
            L.call("mc_conv2d_wgrad", C.byref(d), a0.data_ptr(), L.ptr(x1), dy.data_ptr(), wpart.data_ptr(), st)
        L.call("mc_conv2d_wgrad_finalize", C.byref(d), wpart.data_ptr(), dw.data_ptr(), db.data_ptr(), st)
        dws.append((dw, db))
    torch.cuda.synchronize()
    if dt == "bf16":
        assert torch.equal(dws[0][0], dws[1][0]) and torch.equal(dws[0][1], dws[1][1])
    else:
        assert rel_l2(dws[1][0], dws[0][0]) <= 1e-6 and rel_l2(dws[1][1], dws[0][1]) <= 1e-6


@pytest.mark.parametrize("dt", ["bf16", "fp32", "mixed"])
@pytest.mark.parametrize("mode,c,co,k,hw,gn", [("reflect", 16, 16, 5, (37, 45), True), ("zeros", 16, 32, 5, (21, 33), True),
                                               ("replicate", 32, 16, 3, (18, 20), True), ("reflect", 64, 64, 5, (17, 19), False),
                                               ("reflect", 16, 16, 5, (4, 5), True),
                                               # >= 140 columns on the padded domain: the row-reuse kernel's epilogue (loader waves)
                                               ("reflect", 16, 16, 5, (37, 150), True), ("zeros", 16, 16, 5, (20, 140), True),
                                               ("replicate", 16, 32, 5, (33, 141), False), ("reflect", 32, 16, 3, (18, 200), True)])
def test_input_gradient_epilogue(dt, mode, c, co, k, hw, gn):
    """Input-gradient launch with the GroupNorm-backward epilogue + mc_fold_padded_dz, against an fp64 evaluation from the
    same stored operands: dz = fold(conv^T(dY)) * act'(scale y + shift) on the interior of the padded buffer, and
    (sum dz, sum dz yhat) per (sample, channel) from the partial table.  The layer is conv(c -> co); its source tensor is
    a = act(GN(y)) with c channels (gn = False: activation only)."""
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    lib = L.load()
    # "mixed": the layer is MC_MIX16 (y is f16), its input-gradient launch and every gradient tensor are bf16
    mcd, tdt = (L.MC_F32, torch.float32) if dt == "fp32" else (L.MC_BF16, torch.bfloat16)
    mcl, ydt = (L.MC_MIX16, torch.float16) if dt == "mixed" else (mcd, tdt)
    N, (H, W), p = 2, hw, k // 2
    g = torch.Generator().manual_seed(c * 100 + co + k)
    y = _cb8(torch.randn((N, c, H, W), generator=g) * 1.2 + 0.2, ydt)
    dY = _cb8(torch.randn((N, co, H, W), generator=g), tdt)
    wt = (torch.randn((co, c, k, k), generator=g) / (c * k * k) ** 0.5).to(DEV)
    groups = c // 4
    coef = None
    if gn:
        mean = 0.2 + 0.1 * torch.randn((N, groups), generator=g)
        rstd = 0.8 + 0.1 * torch.rand((N, groups), generator=g)
        gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
        mean_c, rstd_c = mean.repeat_interleave(4, 1), rstd.repeat_interleave(4, 1)
        coef = torch.stack([rstd_c * gamma, beta - mean_c * rstd_c * gamma, mean_c, rstd_c], -1).float().to(DEV).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    d = L.ConvDesc(N, H, W, c, 0, co, k, p, L.PAD_MODES[mode], mcl, 0, 0, 0)
    dd = L.ConvDesc(N, H, W, co, 0, c, k, k - 1, 0, mcd, 0, 0, 0)
    dbank = torch.empty(lib.mc_packed_weight_bytes(C.byref(d), 1), dtype=torch.uint8, device=DEV)
    L.call("mc_pack_weights", C.byref(d), wt.data_ptr(), 1, dbank.data_ptr(), st)
    tiles = lib.mc_conv_tiles(C.byref(dd))
    fb = lib.mc_fold_blocks(H, W, p, L.PAD_MODES[mode])
    part = torch.full((N, tiles + fb, c, 2), float("nan"), device=DEV)
    dxp = torch.zeros((N, c // 8, H + 2 * p, W + 2 * p, 8), dtype=tdt, device=DEV)
    epi = L.ConvEpilogue(y.data_ptr(), L.ptr(coef), L.ACTS["gelu"], p, L.PAD_MODES[mode], H, W, part.data_ptr(), tiles + fb,
                         int(dt == "mixed"))
    L.call("mc_conv2d_fused", C.byref(dd), dY.data_ptr(), None, None, dbank.data_ptr(), None, dxp.data_ptr(), None, None,
           C.byref(epi), st)
    L.call("mc_fold_padded_dz", dxp.data_ptr(), N, c, H, W, p, L.PAD_MODES[mode], mcl, y.data_ptr(), L.ptr(coef), L.ACTS["gelu"],
           part.data_ptr(), tiles + fb, tiles, st)
    torch.cuda.synchronize()
    # fp64 reference from the stored (rounded) operands and the bank's rounded weights
    y64, dY64 = _from_cb8(y, c).double(), _from_cb8(dY, co).double()
    w64 = (wt.to(tdt).double().cpu()).requires_grad_(False)
    a = y64.clone().requires_grad_(True)          # stands for the activated tensor: only the adjoint of pad + conv matters
    ap = F.pad(a, (p, p, p, p), mode={"zeros": "constant", "reflect": "reflect", "replicate": "replicate"}[mode])
    out = F.conv2d(ap, w64)
    (dA,) = torch.autograd.grad(out, a, dY64)
    if gn:
        c64 = coef.double().cpu()
        sc, sh, me, rs = (c64[:, :, i].view(N, c, 1, 1) for i in range(4))
    else:
        sc, sh, me, rs = (torch.full((N, c, 1, 1), v, dtype=torch.float64) for v in (1.0, 0.0, 0.0, 0.0))
    z = (y64 * sc + sh).requires_grad_(True)
    (gp,) = torch.autograd.grad(F.gelu(z).sum(), z)
    dz = dA * gp
    got = _from_cb8(dxp, c).double()[:, :, p:p + H, p:p + W]
    # ("mixed" on the row-reuse kernel: GELU' is the packed-f16 polynomial, max error 1.2e-2 in the tails, rms 1.6e-3)
    tol = {"bf16": 1e-2, "fp32": 2e-5, "mixed": 2.5e-2}[dt]
    stol = {"bf16": 2e-3, "fp32": 1e-5, "mixed": 4e-3}[dt]
    print(f"\n{dt} {mode} {c}->{co} k{k} {hw}: max |dz err| / max |dz| = {float((got - dz).abs().max()) / float(dz.abs().max()):.3e}, "
          f"rms {float((got - dz).pow(2).mean().sqrt()) / float(dz.pow(2).mean().sqrt()):.3e}")
    if os.environ.get("DBG_DZ"):
        e = (got - dz)
        print("per-channel rms err / rms dz:", [round(float(e[:, ch].pow(2).mean().sqrt() / dz[:, ch].pow(2).mean().sqrt()), 3) for ch in range(c)])
        print("err vs dA:", float((got - dA).abs().max()), " err vs 0.5 dA:", float((got - 0.5 * dA).abs().max()), " |dz|max", float(dz.abs().max()))
        gg = got / dA
        print("implied g sample (got/dA) vs true g:", gg[0, 0, 10, 60:68].tolist(), gp[0, 0, 10, 60:68].tolist())
        print("implied g ch1:", gg[0, 1, 10, 60:68].tolist(), gp[0, 1, 10, 60:68].tolist())
        print("z sample:", z[0, 0, 10, 60:68].tolist())
    assert float((got - dz).abs().max()) <= tol * float(dz.abs().max()), float((got - dz).abs().max())
    s = part.double().cpu().sum(1)                 # [N, c, 2]
    assert torch.isfinite(s).all()
    ref1, ref2 = dz.sum((2, 3)), (dz * (y64 - me) * rs).sum((2, 3))
    scale = float(dz.abs().sum((2, 3)).max())
    print(f"   sums: S1 err / sum|dz| {float((s[..., 0] - ref1).abs().max()) / scale:.3e}")
    assert float((s[..., 0] - ref1).abs().max()) <= stol * scale
    if gn:
        assert float((s[..., 1] - ref2).abs().max()) <= stol * float((dz * (y64 - me) * rs).abs().sum((2, 3)).max())
    # last phase from dz: dy = scale dz - rstd (m1 + yhat m2)
    m12 = torch.stack([0.01 * torch.randn((N, groups), generator=g), 0.02 * torch.randn((N, groups), generator=g)], -1).to(DEV)
    dy = torch.zeros((N, c // 8, H, W, 8), dtype=tdt, device=DEV)
    gs = L.GradSrc(dxp.data_ptr(), L.GSRC_PADFOLD, p, L.PAD_MODES[mode], 1, H, W, 0, 0)
    L.call("mc_gn_bwd_apply_dz", C.byref(gs), y.data_ptr(), N, c, H, W, groups, L.ptr(coef), m12.data_ptr() if gn else None,
           mcl, dy.data_ptr(), st)
    torch.cuda.synchronize()
    if gn:
        m1 = m12[..., 0].double().cpu().repeat_interleave(4, 1).view(N, c, 1, 1)
        m2 = m12[..., 1].double().cpu().repeat_interleave(4, 1).view(N, c, 1, 1)
        ref = sc * got - rs * (m1 + (y64 - me) * rs * m2)
    else:
        ref = got
    err = float((_from_cb8(dy, c).double() - ref).abs().max())
    assert err <= (2e-5 if dt == "fp32" else 1e-2) * float(ref.abs().max()), err


@pytest.mark.parametrize("kind", ["unet", "unet16", "convae", "newfluidnet"])
def test_mixed_precision_mode_on_every_graph(kind):
    """precision='mixed' (f16 forward tensors, bf16 gradient tensors) on every graph family -- pooled / concatenated /
    upsampled tensors, 3 x 3 and 5 x 5 layers: forward and every parameter gradient are at least as close to the fp32 execution
    as the plain bf16 mode is, and the network output is closer (11 instead of 8 significant bits in every forward tensor)."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
    res = {}
    for prec in ("fp32", "bf16", "mixed"):
        if kind == "newfluidnet":
            torch.manual_seed(11)
            m = NewFluidNet(3, 7, 16, 3, torch.device(DEV), "gelu", "zeros", "mae", use_symm=True, repeats=2, f=5, p_pred=True)
            shape = (2, 7, 64, 90)
        else:
            m, shape = _build(kind, "reflect", 11)
        m = m.to(DEV).set_precision(prec)
        g = torch.Generator().manual_seed(12)
        x = torch.randn(shape, generator=g).to(DEV)
        outs = m(x)
        outs = [o for o in (outs if isinstance(outs, (tuple, list)) else [outs]) if o is not None]
        loss = sum((o * torch.randn(o.shape, generator=g).to(DEV)).sum() for o in outs)
        loss.backward()
        res[prec] = (torch.cat([o.detach().flatten() for o in outs]), torch.cat([p.grad.flatten() for p in m.parameters()]))
    for j, what in enumerate(("output", "gradient")):
        e_bf = rel_l2(res["bf16"][j], res["fp32"][j])
        e_mx = rel_l2(res["mixed"][j], res["fp32"][j])
        assert e_mx <= 1.25 * e_bf + 1e-3, (kind, what, e_mx, e_bf)
    assert rel_l2(res["mixed"][0], res["fp32"][0]) < 0.5 * rel_l2(res["bf16"][0], res["fp32"][0])
