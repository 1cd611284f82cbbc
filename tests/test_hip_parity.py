"""GPU parity: the HIP path (through the C ABI, via the reference-named modules) against the
golden vectors captured from the reference and against the CPU oracle on the same seeded
inputs.  f32 mode tolerance: per-field MAE < 1e-5 (BASELINE north_star); element-wise checks
use atol/rtol stated per test."""
import numpy as np
import pytest
import torch

import fields
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype).to(DEV)


def mae(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).mean())


def assert_close(a, b, atol, rtol=1e-4, what=""):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (tol {atol}+{rtol}*|ref|), MAE {err.mean():.3e}"


def load_sd(module, g, prefix="sd/"):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    return sd


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_symmetric_conv_golden(golden, tag, precision):
    """SymmetricConv2d against the reference's own layer (golden g1a / g1b / g1c): 'h' pairs (what FluidLayer uses), and
    -- g1b -- 'h' + 'v' pairs + 'hv' quadruples together; output, unique-filter gradient and bias gradient.  16-bit modes at
    their rounding level."""
    from pbml_mantle_convection_amd.symmetric_layers_torch import SymmetricConv2d
    g = golden(f"g1{tag}_symconv")
    ci, co, k, h, v, hv = [int(t) for t in g["meta"]]
    m = SymmetricConv2d(ci, co, k, padding="same", padding_mode=str(g["mode"]), symmetry={"h": h, "v": v, "hv": hv})
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(g["w"]).float())
        m.bias.copy_(torch.from_numpy(g["b"]).float())
    m = m.to(DEV).set_precision(precision)
    y = m(dev(g["x"]))
    lo = precision != "fp32"
    assert_close(y, g["y"], atol=2e-5 if not lo else 2e-2 * float(np.abs(g["y"]).max()), what="y")
    (y * dev(g["ct"])).sum().backward()
    assert_close(m.weight.grad, g["dw"], atol=2e-4 if not lo else 2e-2 * float(np.abs(g["dw"]).max()), rtol=1e-4, what="dw")
    assert_close(m.bias.grad, g["db"], atol=2e-4 if not lo else 2e-2 * float(np.abs(g["db"]).max()), rtol=1e-4, what="db")


def test_symmetric_conv_rejects_unsupported(golden):
    from pbml_mantle_convection_amd.symmetric_layers_torch import SymmetricConv2d
    m = SymmetricConv2d(3, 16, 3, padding="same", dilation=2, symmetry={"h": 4}).to(DEV)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 8, 8, device=DEV))
    m2 = SymmetricConv2d(3, 16, 3, padding="same", symmetry={"h": 4})
    with pytest.raises(RuntimeError):          # CPU tensors: no CPU fallback
        m2(torch.zeros(1, 3, 8, 8))


@pytest.mark.parametrize("tag", list("abcdef"))
def test_fluid_layer_golden(golden, tag):
    from pbml_mantle_convection_amd.pytorch_networks_convae import FluidLayer
    g = golden(f"g2{tag}_fluidlayer")
    ci, co, k, symm = [int(t) for t in g["meta"]]
    m = FluidLayer(ci, co, str(g["act"]), str(g["mode"]), bool(symm), 1, f=k)
    load_sd(m, g)
    m = m.to(DEV)
    y = m(dev(g["x"]))
    assert_close(y, g["y"], atol=2e-5, what="y")
    (y * dev(g["ct"])).sum().backward()
    for n, p in m.named_parameters():
        assert_close(p.grad, g["grad/" + n], atol=3e-4, rtol=2e-4, what=n)


@pytest.mark.parametrize("tag", ["curl", "mae", "mass_rep", "mae_zeros"])
def test_unet_golden(golden, tag):
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden(f"g4_unet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), str(g["loss_type"]),
             use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    load_sd(m, g)
    m = m.to(DEV)
    x = dev(fields.unet_input(2, 40, 54, 41, c_i=c_i))
    outs = m(x)
    loss = 0.0
    for n, o in zip("uvpT", outs):
        if o is None:
            continue
        ref = g["out/" + n]
        assert mae(o, ref) < 1e-5, (n, mae(o, ref))
        assert_close(o, ref, atol=5e-5, rtol=1e-4, what=n)
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        scale = max(1.0, float(np.abs(ref).max()))
        assert_close(p.grad, ref, atol=2e-4 * scale, rtol=1e-3, what=n)


@pytest.mark.parametrize("tag", ["mae", "curl"])
def test_convae_golden(golden, tag):
    from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE
    g = golden(f"g5_convae_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = ConvAE(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", str(g["r_p"]), str(g["loss_type"]),
               use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    load_sd(m, g)
    m = m.to(DEV)
    y = m(dev(g["x"]))
    assert mae(y, g["y"]) < 1e-5
    assert_close(y, g["y"], atol=5e-5, rtol=1e-4, what="y")
    (y * dev(g["ct"])).sum().backward()
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        scale = max(1.0, float(np.abs(ref).max()))
        assert_close(p.grad, ref, atol=2e-4 * scale, rtol=1e-3, what=n)


@pytest.mark.parametrize("kernel", ["walk", "taps"])
@pytest.mark.parametrize("hi,wi,ho,wo,c,dt", [
    (253, 256, 506, 512, 8, "bf16"),      # level 0 of the headline net: exact x2, one 512-column strip
    (126, 128, 253, 256, 16, "bf16"),     # irregular ratio in y (the row window slips once), 12-entry y lists
    (31, 32, 63, 64, 24, "f32"),          # 128-thread blocks, 3 channel blocks
    (40, 300, 80, 600, 8, "f32"),         # wider than a block: column strips with their halo
    (20, 24, 60, 72, 8, "f32"),           # x3: the window slides every third row
    (30, 63, 60, 127, 8, "bf16"),         # irregular ratio in x (12-entry x lists)
    (5, 6, 10, 12, 8, "f32"),             # every row and column clamped
    (7, 9, 7, 9, 8, "f32"),               # identity-sized resample
])
def test_bicubic_adjoint_kernels_vs_torch(kernel, hi, wi, ho, wo, c, dt):
    """The adjoint of nn.Upsample(mode='bicubic') -- the row-walk kernel (mc_bicubic_bwd_walk) and the tiled one
    (mc_bicubic_bwd_taps) -- against torch's autograd of F.interpolate on the CPU (f64), from a plain gradient tensor and from
    the interior of a padded one."""
    import ctypes as C
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    from pbml_mantle_convection_amd.engine import bicubic_tables
    L.load()
    st = L.stream()
    N = 2
    gen = torch.Generator().manual_seed(hi * 1000 + wo)
    gout = torch.randn((N, c, ho, wo), generator=gen)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    mcdt = L.MC_BF16 if dt == "bf16" else L.MC_F32
    gq = gout.to(tdt).float()                                         # what the kernel reads
    x = torch.zeros((N, c, hi, wi), dtype=torch.float64, requires_grad=True)
    F.interpolate(x, size=(ho, wo), mode="bicubic", align_corners=False).backward(gq.double())
    ref = x.grad.float()
    ty, tx = bicubic_tables(hi, ho), bicubic_tables(wi, wo)
    mt = [int(np.diff(t[2]).max()) for t in (ty, tx)]
    dy, dxt = [torch.from_numpy(a).to(DEV) for a in ty], [torch.from_numpy(a).to(DEV) for a in tx]
    c8 = c // 8
    for pad in (0, 2):
        full = torch.randn((N, c, ho + 2 * pad, wo + 2 * pad), generator=gen)          # halo: garbage that must not be read
        full[:, :, pad:pad + ho, pad:pad + wo] = gq
        buf = full.view(N, c8, 8, ho + 2 * pad, wo + 2 * pad).permute(0, 1, 3, 4, 2).contiguous().to(DEV).to(tdt)
        gs = L.GradSrc(L.ptr(buf), L.GSRC_PADFOLD if pad else L.GSRC_PLAIN, pad, 2, 1, ho, wo, 0, 0)
        out = torch.full((N, c8, hi, wi, 8), float("nan"), device=DEV).to(tdt)
        if kernel == "walk":
            L.call("mc_bicubic_bwd_walk", C.byref(gs), N, c, hi, wi, ho, wo, L.ptr(dy[0]), L.ptr(dy[1]), L.ptr(dy[2]), L.ptr(dy[3]),
                   L.ptr(dxt[2]), L.ptr(dxt[3]), L.ptr(dxt[4]), mt[1], mcdt, L.ptr(out), st)
        else:
            L.call("mc_bicubic_bwd_taps", C.byref(gs), N, c, hi, wi, ho, wo, L.ptr(dy[2]), L.ptr(dy[3]), L.ptr(dy[4]), L.ptr(dxt[2]),
                   L.ptr(dxt[3]), L.ptr(dxt[4]), mt[0], mt[1], mcdt, L.ptr(out), st)
        got = out.float().permute(0, 1, 4, 2, 3).reshape(N, c, hi, wi).cpu()
        tol = 8e-3 if dt == "bf16" else 2e-5                              # bf16: the stored result's own rounding
        torch.testing.assert_close(got, ref, rtol=tol, atol=tol * float(ref.abs().max()))


@pytest.mark.parametrize("hi,wi,ho,wo,c,dt", [
    (253, 256, 506, 512, 8, "f16"), (126, 128, 253, 256, 16, "bf16"), (31, 32, 63, 64, 24, "f32"), (40, 300, 80, 600, 8, "f32"),
    (20, 24, 60, 72, 8, "f32"), (30, 63, 60, 127, 8, "f16"), (5, 6, 10, 12, 8, "f32"), (7, 9, 7, 9, 8, "f32"), (8, 32, 128, 506, 8, "f32"),
])
def test_bicubic_forward_kernel_vs_torch(hi, wi, ho, wo, c, dt):
    """mc_bicubic_fwd (the row-walk kernel for every upsample) against F.interpolate(mode='bicubic') on the CPU in f64:
    exact and irregular ratios, column strips, clamped borders, x16 (NewFluidNet's coarsest level)."""
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    from pbml_mantle_convection_amd.engine import bicubic_tables
    L.load()
    N = 2
    gen = torch.Generator().manual_seed(hi * 1000 + wo + 1)
    tdt, mcdt = {"f16": (torch.float16, L.MC_MIX16), "bf16": (torch.bfloat16, L.MC_BF16), "f32": (torch.float32, L.MC_F32)}[dt]
    xq = torch.randn((N, c, hi, wi), generator=gen).to(tdt).float()
    ref = F.interpolate(xq.double(), size=(ho, wo), mode="bicubic", align_corners=False).float()
    ty, tx = bicubic_tables(hi, ho), bicubic_tables(wi, wo)
    dy, dxt = [torch.from_numpy(a).to(DEV) for a in ty[:2]], [torch.from_numpy(a).to(DEV) for a in tx[:2]]
    c8 = c // 8
    buf = xq.view(N, c8, 8, hi, wi).permute(0, 1, 3, 4, 2).contiguous().to(DEV).to(tdt)
    out = torch.full((N, c8, ho, wo, 8), float("nan"), device=DEV).to(tdt)
    L.call("mc_bicubic_fwd", L.ptr(buf), N, c, hi, wi, ho, wo, L.ptr(dy[0]), L.ptr(dy[1]), L.ptr(dxt[0]), L.ptr(dxt[1]), mcdt, L.ptr(out),
           L.stream())
    got = out.float().permute(0, 1, 4, 2, 3).reshape(N, c, ho, wo).cpu()
    tol = {"f16": 1e-3, "bf16": 8e-3, "f32": 2e-5}[dt]                    # 16-bit: the stored result's own rounding
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol * float(ref.abs().max()))


def test_concat_and_gradient_source_kinds():
    """mc_concat_cb8 (torch.cat of > 2 operands), mc_gsrc_sum, slices of a concatenated gradient and the AvgPool adjoint of a
    plain tensor (SURVEY 8f N1 plumbing) against torch."""
    import ctypes as C
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    L.load()
    st = L.stream()
    N, H, W = 2, 9, 13
    g = torch.Generator().manual_seed(5)
    chans = [8, 16, 8, 5]

    def to_cb8(t):                                                   # NCHW f32 -> CB8 f32 [N][C8][H][W][8]
        n, c, h, w = t.shape
        c8 = (c + 7) // 8
        p = torch.zeros((n, c8 * 8, h, w))
        p[:, :c] = t
        return p.view(n, c8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous().to(DEV)

    def from_cb8(t, c):
        n, c8, h, w, _ = t.shape
        return t.permute(0, 1, 4, 2, 3).reshape(n, c8 * 8, h, w)[:, :c].cpu()

    xs = [torch.randn((N, c, H, W), generator=g) for c in chans]
    bufs = [to_cb8(x) for x in xs]
    c8s = [(c + 7) // 8 for c in chans]
    out = torch.empty((N, sum(c8s), H, W, 8), device=DEV)
    srcs = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
    cs = (C.c_int32 * len(bufs))(*chans)
    L.call("mc_concat_cb8", srcs, cs, len(bufs), N, H, W, L.MC_F32, L.ptr(out), st)
    cat = from_cb8(out, sum(c8s) * 8)
    off = 0
    for x, c8 in zip(xs, c8s):
        assert torch.equal(cat[:, off * 8:off * 8 + x.shape[1]], x)
        off += c8
    # gradient sources: slice 1 (16 channels at block 1) of the concatenated tensor + AvgPool(2) adjoint of a plain tensor
    low = torch.randn((N, 16, H // 2, W // 2), generator=g)
    lowb = to_cb8(low)
    g0 = L.GradSrc(L.ptr(out), L.GSRC_PLAIN, 0, 0, 1, H, W, sum(c8s), 1)
    g1 = L.GradSrc(L.ptr(lowb), L.GSRC_PLAIN_POOL, 0, 0, 2, H // 2, W // 2, 0, 0)
    res = torch.empty((N, 2, H, W, 8), device=DEV)
    L.call("mc_gsrc_sum", C.byref(g0), C.byref(g1), N, 16, H, W, L.MC_F32, L.ptr(res), st)
    xp = low.clone().requires_grad_(True)
    up = torch.zeros((N, 16, H, W))
    up[:, :, :2 * (H // 2), :2 * (W // 2)] = F.interpolate(low, scale_factor=2, mode="nearest") / 4.0   # adjoint of AvgPool(2), floor mode
    ref = xs[1] + up
    torch.testing.assert_close(from_cb8(res, 16), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["mae_zeros", "curl_rep"])
def test_newfluidnet_vs_golden(golden, tag):
    """NewFluidNet forward + every parameter gradient on the HIP path (fp32 mode) against the reference."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
    g = golden(f"g12_newfluidnet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = NewFluidNet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), str(g["loss_type"]),
                    use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    sd = {k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = dev(fields.unet_input(1, 128, 506, 121, c_i=c_i))
    outs = m(x)
    loss = 0.0
    for n, o in zip("uvp", outs):
        ref = g["out/" + n]
        assert tuple(o.shape) == ref.shape
        assert_close(o, ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())), rtol=1e-4, what="out " + n)
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        if n == "conv.3.bias":
            # null direction: the spatial zero-mean cancels the last bias, its reference gradient is exactly 0 and ours
            # is the rounding residue of O(1e5) cancelling terms
            assert float(np.abs(ref).max()) < 1e-9 and float(p.grad.abs().max()) < 5e-3
            continue
        assert_close(p.grad, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)


def _n3_grid(H, W):
    xs = np.concatenate(([0.0], (np.arange(W - 2) + 0.5) * 4.0 / (W - 2), [4.0]))
    ys = np.concatenate(([0.0], (np.arange(H - 2) + 0.5) * 1.0 / (H - 2), [1.0]))
    return (np.broadcast_to(xs[None, :], (H, W)).copy().reshape(1, 1, H, W), np.broadcast_to(ys[:, None], (H, W)).copy().reshape(1, 1, H, W))


def test_adnet_step_vs_golden(golden):
    """the fused advection-diffusion stencil (+ CFL time-step reduction) against the reference's ADNet."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import ADNet
    g = golden("g14_adnet")
    H, W = 128, 506
    xc, yc = _n3_grid(H, W)
    ad = ADNet(DEV)
    for k, seed in enumerate(g["seeds"]):
        seed = int(seed)
        u = fields.smooth_field(1, H, W, seed + 1).reshape(1, 1, H, W) * 400.0
        v = fields.smooth_field(1, H, W, seed + 2).reshape(1, 1, H, W) * 400.0
        Tp = fields.temperature_field(1, H, W, seed + 3).reshape(1, 1, H, W)
        inp = dev(np.concatenate((u, v, Tp, np.full((1, 1, H, W), 2.5), xc, yc), 1))
        Tn, dt = ad(inp)
        assert abs(float(dt) - float(g[f"dt/{k}"])) <= 2e-6 * float(g[f"dt/{k}"])
        ref = g[f"T_next/{k}"]
        got = Tn if k == 0 else fields.strided_sample(Tn.cpu().numpy(), 4001)
        assert_close(got, ref, atol=2e-6, rtol=0, what="T_next")
        Tn2, _ = ad(inp, dt=3e-7)
        assert_close(fields.strided_sample(Tn2.cpu().numpy(), 4001), g[f"T_next_fixed/{k}"], atol=2e-6, rtol=0, what="T_next fixed dt")


def test_ts_rollout_newfluidnet_vs_oracle(golden):
    """three rollout steps (input builder -> NewFluidNet on the HIP path -> un-scale -> ADNet -> boundary
    conditions) against the oracle's TS restatement driving the oracle's NewFluidNet with the same weights."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import ADNet, NewFluidNet, TS
    g = golden("g12_newfluidnet_mae_zeros")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = NewFluidNet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), "mae", use_symm=bool(symm),
                    repeats=repeats, f=f, p_pred=True)
    sd32 = {k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")}
    m.load_state_dict(sd32)
    m = m.to(DEV)
    sd = {k: v.double() for k, v in sd32.items()}
    H, W = 128, 506
    xc, yc = (torch.from_numpy(a) for a in _n3_grid(H, W))
    T0 = torch.from_numpy(fields.temperature_field(1, H, W, 1500)).view(1, 1, H, W)
    raq, fkt, fkp = (torch.tensor(v, dtype=torch.float64) for v in (2.5, 1e7, 30.0))
    nd = [torch.tensor(v, dtype=torch.float64).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]

    def stokes_ref(inp):
        u, v, p = O.newfluidnet_forward(sd, inp, levels=levels, repeats=repeats, act=str(g["act"]), r_p=str(g["r_p"]),
                                        loss_type="mae", use_symm=bool(symm), p_pred=True)
        return u * 20.0, v * 20.0, p                      # (random weights: amplify so that advection matters)

    class Scaled(torch.nn.Module):
        def forward(self, inp):
            u, v, p = m(inp)
            return u * 20.0, v * 20.0, p

    xr, dtr, ur, vr, pr, Vr = O.ts_rollout(stokes_ref, T0, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc, ts=3)
    ts = TS(Scaled(), ADNet(DEV), DEV, ts=3, net="newfluidnet")
    x, dts, u, v, p, V = ts(T0, None, None, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc)
    for i in (1, 2, 3):
        assert abs(float(dts[i]) - float(dtr[i])) <= 1e-4 * float(dtr[i]), (i, float(dts[i]), float(dtr[i]))
        assert_close(x[i], xr[i].numpy(), atol=5e-5, rtol=0, what=f"T step {i}")
    assert_close(u, ur.numpy(), atol=2e-4 * float(ur.abs().max()), rtol=1e-3, what="u")
    assert_close(V, Vr.numpy(), atol=1e-6, rtol=1e-4, what="V")


def test_ts_rollout_unet_branch_vs_oracle():
    """TS(net='unet') (reference :411-446): input builder -> Unet on the HIP path -> wall conditions, three steps, against the
    oracle's restatement of that branch (pinned by g20) driving the oracle's Unet with the same weights."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import TS, Unet
    torch.manual_seed(5)
    H, W = 128, 506
    m = Unet(3, 10, 8, 4, torch.device(DEV), "gelu", "reflect", "mae", use_symm=True, repeats=2, f=5, p_pred=True).to(DEV)
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    xc, yc = (torch.from_numpy(a) for a in _n3_grid(H, W))
    T0 = torch.from_numpy(fields.temperature_field(1, H, W, 2100)).view(1, 1, H, W)
    up = torch.from_numpy(fields.smooth_field(1, H, W, 2101)).view(1, 1, H, W)
    vp = torch.from_numpy(fields.smooth_field(1, H, W, 2102)).view(1, 1, H, W)
    dt = torch.full((1, 1, H, W), 3e-5, dtype=torch.float64)
    raq, fkt, fkp = (torch.tensor(v, dtype=torch.float64) for v in (2.5, 1e7, 30.0))
    nd = [torch.tensor(v, dtype=torch.float64).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]

    def stokes_ref(inp):
        return O.unet_forward(sd, inp, levels=3, repeats=2, act="gelu", r_p="reflect", loss_type="mae", use_symm=True, p_pred=True)

    xr, _, ur, vr, _, Vr = O.ts_rollout_unet(stokes_ref, T0, yc, nd[0], nd[1], nd[2], fkt, fkp, xc, yc, up, vp, dt, ts=3)
    ts = TS(m, None, DEV, ts=3, net="unet")
    x, dts, u, v, p, V = ts(T0, None, None, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc, u_prev=up, v_prev=vp, dt=dt)
    assert p is None and dts == {}
    for i in (1, 2, 3):
        assert_close(x[i], xr[i].numpy(), atol=3e-5, rtol=0, what=f"T step {i}")
    assert_close(u, ur.numpy(), atol=3e-5, rtol=1e-4, what="u")
    assert_close(v, vr.numpy(), atol=3e-5, rtol=1e-4, what="v")
    assert_close(V, Vr.numpy(), atol=1e-6, rtol=1e-4, what="V")
    with pytest.raises(ValueError):
        ts(T0, None, None, yc, nd[0], nd[1], nd[2], raq, fkt, fkp, xc, yc)


def test_ts_rollout_graph_replay_equals_eager():
    """One rollout step captured as a HIP graph and replayed gives bit-identical fields to eager launches, also on a second
    call with fresh argument tensors (the captured step must read persistent buffers, not the caller's temporaries)."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import ADNet, NewFluidNet, TS
    torch.manual_seed(1)
    H, W = 64, 90
    m = NewFluidNet(3, 7, 8, 3, torch.device(DEV), "gelu", "zeros", "mae", use_symm=True, repeats=2, f=5, p_pred=True).to(DEV)
    xc, yc = (torch.from_numpy(a) for a in _n3_grid(H, W))
    res = {}
    for ug in (False, True):
        ts = TS(m, ADNet(DEV), DEV, ts=4, net="newfluidnet", use_graph=ug)
        for rep in range(2):
            T0 = torch.from_numpy(fields.temperature_field(1, H, W, 77 + rep)).view(1, 1, H, W)
            args = [torch.tensor(v) for v in (2.5 + rep, 1e7, 30.0)]
            nd = [torch.tensor(v).view(1, 1, 1, 1) for v in (0.25, 0.26, 0.74)]
            x, dts, *_ = ts(T0, None, None, yc.clone(), nd[0], nd[1], nd[2], args[0], args[1], args[2], xc.clone(), yc.clone())
            res[(ug, rep)] = (x[4].clone(), torch.stack([dts[i] for i in range(1, 5)]).clone())
    for rep in range(2):
        assert torch.equal(res[(False, rep)][0], res[(True, rep)][0])
        assert torch.equal(res[(False, rep)][1], res[(True, rep)][1])
        assert not torch.isnan(res[(True, rep)][0]).any()


@pytest.mark.parametrize("tag", ["k5_symm", "k3_plain"])
def test_boundary_learned_conv_vs_golden(golden, tag):
    """the learned-padding layer (nine valid convolutions framed together) on the library's conv kernels:
    output, input gradient and the gradients of all nine banks and the shared bias against the reference."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import BoundaryLearnedConvolution2D
    g = golden(f"g16_learned_{tag}")
    c_i, c_o, k, symm = [int(v) for v in g["meta"]]
    m = BoundaryLearnedConvolution2D(c_i, c_o, k, use_symm=bool(symm))
    sd = {n[3:]: torch.from_numpy(g[n]).float() for n in g.files if n.startswith("sd/")}
    assert {n: tuple(v.shape) for n, v in m.state_dict().items()} == {n: tuple(v.shape) for n, v in sd.items()}
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = dev(g["x"]).requires_grad_(True)
    y = m(x)
    assert_close(y, g["y"], atol=2e-5 * max(1.0, float(np.abs(g["y"]).max())), rtol=1e-4, what="y")
    (y * dev(g["ct"])).sum().backward()
    assert_close(x.grad, g["dx"], atol=5e-5 * max(1.0, float(np.abs(g["dx"]).max())), rtol=1e-3, what="dx")
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        assert_close(p.grad, ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)
    with pytest.raises(NotImplementedError):
        m(x, bc_x=4)


def test_fluidlayer_learned_padding_vs_golden(golden):
    """FluidLayer(r_p='learned'): learned-padding conv -> GroupNorm (statistics of the assembled output) -> GELU, through the
    engine's learned node, forward and all gradients against the reference."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import FluidLayer
    g = golden("g16_learned_fluidlayer")
    m = FluidLayer(8, 16, "gelu", "learned", True, 1, f=5)
    sd = {n[3:]: torch.from_numpy(g[n]).float() for n in g.files if n.startswith("sd/")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = dev(g["x"]).requires_grad_(True)
    y = m(x)
    assert_close(y, g["y"], atol=3e-5 * max(1.0, float(np.abs(g["y"]).max())), rtol=1e-4, what="y")
    (y * dev(g["ct"])).sum().backward()
    # (the engine does not produce the gradient w.r.t. a network's input; the stand-alone BoundaryLearnedConvolution2D does)
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        if n == "layers.0.learnable_bias":
            continue              # GroupNorm cancels a per-channel constant only per group: compared below with a loose floor
        assert_close(p.grad, ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3, what="grad " + n)
    ref = g["grad/layers.0.learnable_bias"]
    assert_close(m.layers[0].learnable_bias.grad, ref, atol=2e-3 * max(1.0, float(np.abs(ref).max())), rtol=5e-3, what="bias grad")


def test_unet_learned_padding_vs_golden(golden):
    """the U-Net with r_p='learned' (every conv a BoundaryLearnedConvolution2D, the first layer's bc_x = 4
    strips growing the field instead of F.pad, materialised concats) forward + every parameter gradient vs the reference."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden("g17_unet_learned")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "learned", "mae", use_symm=bool(symm), repeats=repeats, f=f,
             p_pred=bool(p_pred))
    sd = {n[3:]: torch.from_numpy(g[n]).float() for n in g.files if n.startswith("sd/")}
    assert {n: tuple(v.shape) for n, v in m.state_dict().items()} == {n: tuple(v.shape) for n, v in sd.items()}
    m.load_state_dict(sd)
    m = m.to(DEV)
    outs = m(dev(fields.unet_input(2, 40, 54, 172, c_i=c_i)))
    loss = 0.0
    for n, o in zip("uvpT", outs):
        ref = g["out/" + n]
        assert tuple(o.shape) == ref.shape
        assert_close(o, ref, atol=3e-5 * max(1.0, float(np.abs(ref).max())), rtol=2e-4, what="out " + n)
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        if float(np.abs(ref).max()) < 1e-6:
            continue                                    # null directions (the last layer's shared bias under the zero-mean)
        assert_close(p.grad, ref, atol=5e-4 * max(1.0, float(np.abs(ref).max())), rtol=3e-3, what="grad " + n)


def test_newfluidnet_learned_padding_vs_golden(golden):
    """SURVEY 8(f) N4 x N1: NewFluidNet with r_p='learned' (learned-padding FluidLayers and a 5 x 5 learned-padding head)."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
    g = golden("g17_newfluidnet_learned")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = NewFluidNet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "learned", "mae", use_symm=bool(symm), repeats=repeats, f=f,
                    p_pred=bool(p_pred))
    sd = {n[3:]: torch.from_numpy(g[n]).float() for n in g.files if n.startswith("sd/")}
    assert {n: tuple(v.shape) for n, v in m.state_dict().items()} == {n: tuple(v.shape) for n, v in sd.items()}
    m.load_state_dict(sd)
    m = m.to(DEV)
    outs = m(dev(fields.unet_input(1, 128, 506, 176, c_i=c_i)))
    loss = 0.0
    for n, o in zip("uvp", outs):
        ref = g["out/" + n]
        assert_close(fields.strided_sample(o.detach().cpu().numpy(), 20001), ref, atol=3e-5 * max(1.0, float(np.abs(ref).max())),
                     rtol=2e-4, what="out " + n)
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    for n, p in m.named_parameters():
        ref = g["grad/" + n]
        if float(np.abs(ref).max()) < 1e-6:
            continue
        assert_close(p.grad, ref, atol=5e-4 * max(1.0, float(np.abs(ref).max())), rtol=3e-3, what="grad " + n)
