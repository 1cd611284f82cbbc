"""GPU parity of the bf16 (MFMA) precision mode.  bf16 storage cannot meet the f32 gate (the reference itself
run in bf16 on CPU differs from fp64 by MAE 3.2e-3..3.8e-3 on O(0.07) fields), so the stated
tolerances are: single conv vs an fp64 evaluation on bf16-rounded operands: |err| <= 1e-2*max|ref| element-wise
(output rounding 2^-9 relative + f32 accumulation); whole nets: field MAE <= 5e-3 * max(1, mean|ref|-scale),
gradients: relative L2 error <= 6e-2; losses: relative 2e-2.
End-to-end against the fp64 golden vectors the bounds are the bf16 noise floor of these random-weight test nets
(12 % on fields that pass through the curl head's finite differences, 45 % relative L2 on the worst gradient); the tight
checks are the per-kernel tests (1 %) and the quantisation-emulating oracle (1.5 %)."""
import os

import numpy as np
import pytest
import torch

import fields
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.asarray(a)).float().to(DEV).contiguous()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def rel_l2(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("ci,co,k,mode,h,hw", [
    (16, 16, 5, "reflect", 4, (37, 45)),      # NT=1 config, one K-chunk, ragged tiles
    (11, 16, 5, "replicate", 4, (33, 40)),    # padded input channels
    (48, 16, 5, "reflect", 4, (20, 35)),      # three K-chunks
    (16, 32, 5, "zeros", 8, (18, 37)),        # NT=2 config
    (32, 64, 5, "reflect", 16, (17, 19)),     # NT=4 config
    (64, 128, 5, "reflect", 32, (9, 12)),     # two N-groups of NT=4
    (16, 4, 5, "reflect", 0, (21, 33)),       # plain conv, 4 output channels
    (8, 16, 3, "reflect", 4, (19, 23)),       # 3x3
    (16, 48, 3, "zeros", 0, (10, 20)),        # 3 N-tiles -> NT=1 x 3 groups
])
def test_bf16_conv_forward_and_filter_gradient(ci, co, k, mode, h, hw):
    from pbml_mantle_convection_amd.symmetric_layers_torch import SymmetricConv2d
    g = torch.Generator().manual_seed(ci * 1000 + co)
    m = SymmetricConv2d(ci, co, k, padding="same", padding_mode=mode, symmetry={"h": h} if h else {"h": 0})
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) / (ci * k * k) ** 0.5)
        m.bias.copy_(0.1 * torch.randn(co, generator=g))
    x = torch.randn((2, ci, *hw), generator=g)
    ct = torch.randn((2, co, *hw), generator=g)
    mm = m.to(DEV).set_precision("bf16") if hasattr(m, "set_precision") else m.to(DEV)
    mm._build_graph()
    mm.set_precision("bf16")
    y = mm(x.to(DEV))
    # fp64 reference on bf16-rounded operands
    xw = bf16_round(x).requires_grad_(True)
    w64 = bf16_round(m.weight.detach().cpu()).requires_grad_(True)
    b64 = m.bias.detach().cpu().double().requires_grad_(True)
    ref = O.conv2d_same(xw, O.expand_symmetric_weight(w64, {"h": h, "v": 0, "hv": 0}), b64, mode)
    err = (y.double().cpu() - ref.detach()).abs().max()
    assert float(err) <= 1e-2 * float(ref.abs().max()), float(err)
    (y * ct.to(DEV)).sum().backward()
    (ref * bf16_round(ct)).sum().backward()
    assert rel_l2(mm.weight.grad, w64.grad) < 1e-2
    assert rel_l2(mm.bias.grad, b64.grad) < 1e-2


@pytest.mark.parametrize("variant", ["0", "12", "21"])
def test_filter_gradient_kernel_variants(variant):
    """The filter-gradient kernel has tap-group variants (MC_WGRAD_RS: tens digit = one co-tile, units = two co-tiles;
    default 44).  The knob is read once per process, so the non-default forms run the conv cases in a child process:
    0 = row-at-a-time kernel, 1 / 2 = register-shift kernel with one / two tap groups."""
    import subprocess
    import sys
    env = dict(os.environ, MC_WGRAD_RS=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_bf16_conv_forward_and_filter_gradient"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "9 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("precision", ["bf16", "mixed"])
@pytest.mark.parametrize("tag", ["curl", "mae", "mass_rep", "mae_zeros"])
def test_unet_bf16_forward_vs_quantised_oracle(golden, tag, precision):
    """Forward of the 16-bit modes against the oracle with the SAME storage roundings emulated (bf16 / f16 round trips
    wherever the engine stores a tensor): what remains is accumulation order and rounding-boundary flips, so
    the bound is tight: relative L2 error <= 1.5e-2 (bf16), 5e-3 ("mixed": f16 forward tensors) on
    the (y - mean)[..., 3:-3] features."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden(f"g4_unet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), str(g["loss_type"]),
             use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    sd = {k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")}
    m.load_state_dict(sd)
    m = m.to(DEV).set_precision(precision)
    x = fields.unet_input(2, 40, 54, 41, c_i=c_i)
    y = m.features(dev(x))
    qt = torch.bfloat16 if precision == "bf16" else torch.float16
    q = lambda t: t.to(torch.float32).to(qt).to(t.dtype)  # noqa: E731
    ref = O.unet_features_quantised({k: v.double() for k, v in sd.items()}, torch.from_numpy(x), levels, repeats,
                                    str(g["act"]), str(g["r_p"]), bool(symm), q)
    print(f"\n[{tag}] {precision} forward vs quantised oracle: rel-L2 {rel_l2(y, ref):.3e}")
    assert rel_l2(y, ref) < (1.5e-2 if precision == "bf16" else 5e-3), rel_l2(y, ref)


# measured on MI355X (round 2): whole-gradient rel-L2 0.014 / 0.015 / 0.015 (mae, mass_rep, mae_zeros), 0.095 (curl: the head takes
# one-pixel differences x 126 of a bf16-noisy streamfunction); worst single tensor 0.065 / 0.071 / 0.050, curl 0.31
WHOLE_BOUND = {"curl": 0.15, "mae": 0.03, "mass_rep": 0.03, "mae_zeros": 0.03}
WORST_BOUND = {"curl": 0.45, "mae": 0.12, "mass_rep": 0.12, "mae_zeros": 0.12}
# "mixed" (round 3; f16 forward tensors): whole 0.017 / 0.007 / 0.010 / 0.009, worst tensor 0.064 / 0.028 / 0.130 / 0.056 -- the curl
# head's one-pixel differences profit most from the 11-bit forward tensors
WHOLE_BOUND_MIXED = {"curl": 0.03, "mae": 0.015, "mass_rep": 0.015, "mae_zeros": 0.015}
WORST_BOUND_MIXED = {"curl": 0.10, "mae": 0.06, "mass_rep": 0.20, "mae_zeros": 0.09}


@pytest.mark.parametrize("precision", ["bf16", "mixed"])
@pytest.mark.parametrize("tag", ["curl", "mae", "mass_rep", "mae_zeros"])
def test_unet_bf16_vs_golden(golden, tag, precision):
    """Forward and EVERY parameter gradient of both 16-bit modes against the reference's fp64 golden vectors ("mixed": f16
    forward tensors, bf16 gradient tensors -- its bounds are the bf16 mode's: it can only remove rounding)."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden(f"g4_unet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), str(g["loss_type"]),
             use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")})
    m = m.to(DEV).set_precision(precision)
    outs = m(dev(fields.unet_input(2, 40, 54, 41, c_i=c_i)))
    loss = 0.0
    for n, o in zip("uvpT", outs):
        if o is None:
            continue
        ref = g["out/" + n]
        assert float(np.abs(o.detach().double().cpu().numpy() - ref).mean()) <= 0.12 * float(np.abs(ref).mean()), n
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    named = [(n, p) for n, p in m.named_parameters() if float(np.abs(g["grad/" + n]).max()) > 1e-6]
    worst = max(rel_l2(p.grad, g["grad/" + n]) for n, p in named)
    num = sum(float((p.grad.double().cpu() - torch.from_numpy(g["grad/" + n])).norm() ** 2) for n, p in named)
    den = sum(float(np.linalg.norm(g["grad/" + n]) ** 2) for n, p in named)
    whole = (num / den) ** 0.5
    print(f"\n[{tag}] {precision} vs fp64 golden: whole-gradient rel-L2 {whole:.3f}, worst tensor {worst:.3f}")
    # Two bounds.  The gradient as a whole (the direction an optimizer step takes) carries the bf16 noise of a 14-layer
    # network on a 40 x 54 image; single small tensors (GroupNorm offsets whose true gradient nearly cancels over 2000
    # pixels) are far noisier.  At the benched size the same quantities are 6 x smaller: whole 0.033, worst 0.071
    # (tests/test_hip_fullsize.py::test_cfg3_training_step_506_bf16_gradients_vs_oracle, bounds 0.10 / 0.15).
    assert whole < (WHOLE_BOUND if precision == "bf16" else WHOLE_BOUND_MIXED)[tag], whole
    assert worst < (WORST_BOUND if precision == "bf16" else WORST_BOUND_MIXED)[tag], worst


@pytest.mark.parametrize("tag", ["mae", "curl"])
def test_convae_bf16_vs_golden(golden, tag):
    from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE
    g = golden(f"g5_convae_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = ConvAE(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", str(g["r_p"]), str(g["loss_type"]),
               use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")})
    m = m.to(DEV).set_precision("bf16")
    y = m(dev(g["x"]))
    assert rel_l2(y, g["y"]) < 0.12
    (y * dev(g["ct"])).sum().backward()
    # The random-weight ConvAE amplifies ANY rounding (a 1e-4 relative input perturbation moves its bf16 output by 4 %,
    # tools/diag_convae.py), so single small parameters can be far off while the gradient as a whole agrees: compare the
    # concatenated gradient (the direction an optimizer step takes).
    num = sum(float((p.grad.double().cpu() - torch.from_numpy(g["grad/" + n])).norm() ** 2) for n, p in m.named_parameters())
    den = sum(float(np.linalg.norm(g["grad/" + n]) ** 2) for n, _ in m.named_parameters())
    assert (num / den) ** 0.5 < 0.2, (num / den) ** 0.5


TRAIN_WHOLE_BOUND, TRAIN_WORST_BOUND = 0.06, 0.15          # measured 0.037 / 0.099


@pytest.mark.parametrize("tag", ["mass", "curl"])
def test_training_steps_bf16_vs_golden(golden, tag):
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
                 loss_scale=bool(ls), loss_derivative=bool(ld), loss_type=str(g["loss_type"]), precision="bf16")
    for step in range(2):
        gVTp = dev(fields.unet_input(B, H, W, 1100 + step, c_i=11 if p_pred else 10))
        truth = [fields.smooth_field(B, H, W, 1150 + step), fields.smooth_field(B, H, W, 1160 + step)]
        if p_pred:
            truth.append(fields.smooth_field(B, H, W, 1170 + step, amp=0.5))
        truth.append(fields.temperature_field(B, H, W, 1180 + step))
        vals = tr._run_batch(gVTp, dev(np.stack(truth, 1)), None, True)
        ref = g["losses"][step]
        if tag == "curl":
            # loss_derivative multiplies one-pixel differences by 126: bf16 storage noise dominates that term, so
            # only the plain data terms are comparable
            for i in (1, 2, 4):
                assert abs(vals[i] - ref[i]) <= 8e-2 * abs(ref[i]), (i, vals, ref)
        else:
            assert abs(vals[0] - ref[0]) <= 5e-2 * abs(ref[0]), (vals, ref)
        if step == 0 and tag != "curl":       # curl + loss_derivative: sign(noise-level differences) dominates the gradient
            grads = tr.flat.views(tr.flat.grad)
            names = [n for n in grads if float(np.abs(g["grad0/" + n]).max()) > 1e-6]
            worst = max(rel_l2(grads[n], g["grad0/" + n]) for n in names)
            num = sum(float((grads[n].double().cpu() - torch.from_numpy(g["grad0/" + n])).norm() ** 2) for n in names)
            den = sum(float(np.linalg.norm(g["grad0/" + n]) ** 2) for n in names)
            print(f"\n[train {tag}] bf16 vs fp64 golden at 128 x 506: whole-gradient rel-L2 {(num / den) ** 0.5:.3f}, worst tensor {worst:.3f}")
            assert (num / den) ** 0.5 < TRAIN_WHOLE_BOUND, (num / den) ** 0.5
            assert worst < TRAIN_WORST_BOUND, worst


@pytest.mark.parametrize("dtype_name", ["bf16", "fp32"])
@pytest.mark.parametrize("ci0,ci1,co,k,h,hw", [
    (16, 0, 16, 5, 4, (21, 30)),      # N = 16
    (16, 32, 16, 5, 4, (18, 19)),     # concat sources: N = 48 -> 3 groups, split output
    (32, 64, 32, 5, 8, (12, 17)),     # N = 96 (NT = 2 x 3 groups), K over 32 channels (2 chunks)
    (64, 128, 64, 5, 16, (9, 10)),    # N = 192 (NT = 4 x 3 groups)
    (8, 0, 4, 3, 0, (14, 15)),        # tiny channel counts, 3x3
    (32, 0, 32, 5, 8, (31, 33)),      # padded domain of 35 rows: the 12-row tile (3 x 12 instead of 3 x 16), NT = 2
    (64, 128, 64, 5, 16, (17, 18)),   # 21 rows: 12-row tiles with NT = 4, three N-groups
])
def test_input_gradient_kernel_direct(dtype_name, ci0, ci1, co, k, h, hw):
    """mc_conv2d in input-gradient mode (padded domain, rotated/transposed bank, split outputs) against
    conv_transpose2d in fp64 on identically rounded operands — exercised straight through the C ABI."""
    import ctypes as C
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    L.load()
    mc = L.MC_BF16 if dtype_name == "bf16" else L.MC_F32
    tdt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    g = torch.Generator().manual_seed(7 * ci0 + co)
    N, (H, W) = 2, hw
    cin = ci0 + ci1
    U = co - h // 2
    wu = torch.randn((U, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    dy = torch.randn((N, co, H, W), generator=g)
    pad = k // 2
    fwd = L.ConvDesc(N, H, W, ci0, ci1, co, k, pad, 2, mc, h, 0, 0)
    dd = L.ConvDesc(N, H, W, co, 0, cin, k, k - 1, 0, mc, 0, ci0 if ci1 else 0, 0)
    st = L.stream()
    dyd = dy.to(DEV).contiguous()
    dy_cb = torch.empty((N, (co + 7) // 8, H, W, 8), dtype=tdt, device=DEV)
    L.call("mc_pack_nchw", L.ptr(dyd), N, co, co, H, W, 0, 0, None, mc, L.ptr(dy_cb), st)
    bank = torch.empty(L.call("mc_packed_weight_bytes", C.byref(fwd), 1), dtype=torch.uint8, device=DEV)
    wud = wu.to(DEV).contiguous()
    L.call("mc_pack_weights", C.byref(fwd), L.ptr(wud), 1, L.ptr(bank), st)
    Hp, Wp = H + 2 * pad, W + 2 * pad
    o0 = torch.empty((N, (ci0 + 7) // 8, Hp, Wp, 8), dtype=tdt, device=DEV)
    o1 = torch.empty((N, max((ci1 + 7) // 8, 1), Hp, Wp, 8), dtype=tdt, device=DEV)
    L.call("mc_conv2d", C.byref(dd), L.ptr(dy_cb), None, L.ptr(bank), None, L.ptr(o0), L.ptr(o1) if ci1 else None, None, st)
    outs = []
    for buf, c in ((o0, ci0), (o1, ci1)):
        if c == 0:
            continue
        r = torch.empty((N, c, Hp, Wp), dtype=torch.float32, device=DEV)
        L.call("mc_unpack_nchw", L.ptr(buf), N, c, Hp, Wp, 0, None, mc, L.ptr(r), st)
        outs.append(r)
    got = torch.cat(outs, 1).double().cpu()
    rnd = bf16_round if dtype_name == "bf16" else (lambda t: t.double())
    wfull = O.expand_symmetric_weight(rnd(wu), {"h": h, "v": 0, "hv": 0})
    ref = F.conv_transpose2d(rnd(dy), wfull)               # gradient w.r.t. the PADDED input
    assert got.shape == ref.shape
    tol = 1e-2 if dtype_name == "bf16" else 1e-5
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("tag", ["mae_zeros", "curl_rep"])
def test_newfluidnet_bf16_vs_golden(golden, tag):
    """SURVEY 8(f) N1 in bf16 mode: outputs within the bf16 noise floor of the fp64 reference, gradient direction intact."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
    g = golden(f"g12_newfluidnet_{tag}")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = NewFluidNet(levels, c_i, c_h, c_o, torch.device(DEV), str(g["act"]), str(g["r_p"]), str(g["loss_type"]),
                    use_symm=bool(symm), repeats=repeats, f=f, p_pred=bool(p_pred))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith("sd/")})
    m = m.to(DEV).set_precision("bf16")
    x = fields.unet_input(1, 128, 506, 121, c_i=c_i)
    outs = m(dev(x))
    loss = 0.0
    for n, o in zip("uvp", outs):
        ref = g["out/" + n]
        if not (str(g["loss_type"]) == "curl" and n in "uv"):
            # (u, v of the curl head are one-pixel differences of the streamfunction: they amplify the white bf16 noise,
            # DESIGN.md section 4; the streamfunction itself is checked below)
            assert float(np.abs(o.detach().double().cpu().numpy() - ref).mean()) <= 0.12 * float(np.abs(ref).mean()), n
        loss = loss + (o * dev(g["ct/" + n])).sum()
    sd = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("sd/")}
    feat = O.newfluidnet_features(sd, torch.from_numpy(x), levels=levels, repeats=repeats, act=str(g["act"]),
                                  r_p=str(g["r_p"]), use_symm=bool(symm))
    loss.backward()
    assert rel_l2(m.features(dev(x)), feat) < 0.12                 # (a second forward: after the backward of the first)
    num = sum(float((p.grad.double().cpu() - torch.from_numpy(g["grad/" + n])).norm() ** 2) for n, p in m.named_parameters())
    den = sum(float(np.linalg.norm(g["grad/" + n]) ** 2) for n, _ in m.named_parameters())
    # white-noise cotangents pushed through the curl adjoint weight exactly the high-frequency content that bf16 storage
    # noise dominates (tools/diag_newfluidnet.py: fp32 mode 1e-5, bf16 mode 0.2-0.7 per parameter on this random SELU net):
    # for the 'curl' case only the direction of the concatenated gradient is asserted
    assert (num / den) ** 0.5 < (0.6 if str(g["loss_type"]) == "curl" else 0.2), (num / den) ** 0.5


def test_unet_learned_padding_bf16_vs_golden(golden):
    """SURVEY 8(f) N4 in bf16 mode (the learned-padding head stores its output in bf16: no f32 last-layer path there)."""
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    g = golden("g17_unet_learned")
    levels, c_i, c_h, c_o, repeats, f, p_pred, symm = [int(v) for v in g["cfg"]]
    m = Unet(levels, c_i, c_h, c_o, torch.device(DEV), "gelu", "learned", "mae", use_symm=bool(symm), repeats=repeats, f=f,
             p_pred=bool(p_pred))
    m.load_state_dict({n[3:]: torch.from_numpy(g[n]).float() for n in g.files if n.startswith("sd/")})
    m = m.to(DEV).set_precision("bf16")
    outs = m(dev(fields.unet_input(2, 40, 54, 172, c_i=c_i)))
    loss = 0.0
    for n, o in zip("uvpT", outs):
        ref = g["out/" + n]
        assert float(np.abs(o.detach().double().cpu().numpy() - ref).mean()) <= 0.12 * float(np.abs(ref).mean()), n
        loss = loss + (o * dev(g["ct/" + n])).sum()
    loss.backward()
    num = sum(float((p.grad.double().cpu() - torch.from_numpy(g["grad/" + n]).double()).norm() ** 2) for n, p in m.named_parameters())
    den = sum(float(np.linalg.norm(g["grad/" + n].astype(np.float64)) ** 2) for n, _ in m.named_parameters())
    assert (num / den) ** 0.5 < 0.3, (num / den) ** 0.5


@pytest.mark.parametrize("dtype_name", ["bf16", "mixed"])
@pytest.mark.parametrize("c,sc,w,pad_w,mode", [
    (10, 12, 26, 3, "reflect"),      # the benched input's shape class: 10 of 12 source channels, frame of 3
    (10, 10, 30, 1, "replicate"),
    (3, 3, 20, 2, "zeros"),
    (8, 8, 16, 0, "reflect"),        # no frame at all
    (10, 12, 21, 3, "reflect"),      # odd padded width
])
def test_input_pack_equals_padded_reference(dtype_name, c, sc, w, pad_w, mode):
    """mc_pack_nchw (NCHW f32 -> CB8 with the width frame of the network input, per-channel scale) in the 16-bit layouts.
    Every stored value must be the storage rounding of
    scale * F.pad(x) — bit for bit (the conversion is one multiplication and one rounding)."""
    import torch.nn.functional as F
    from pbml_mantle_convection_amd import _lib as L
    L.load()
    mc = L.MC_BF16 if dtype_name == "bf16" else L.MC_MIX16
    tdt = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    g = torch.Generator().manual_seed(c * 100 + w)
    N, H = 2, 5
    x = torch.randn((N, sc, H, w), generator=g)
    cs = torch.rand((c,), generator=g) + 0.5
    xd, csd = x.to(DEV).contiguous(), cs.to(DEV).contiguous()
    wp = w + 2 * pad_w
    out = torch.full((N, (c + 7) // 8, H, wp, 8), float("nan"), dtype=tdt, device=DEV)
    L.call("mc_pack_nchw", L.ptr(xd), N, c, sc, H, w, pad_w, L.PAD_MODES[mode], L.ptr(csd), mc, L.ptr(out), L.stream())
    torch.cuda.synchronize()
    ref = x[:, :c] * cs.view(1, c, 1, 1)
    if pad_w:
        ref = F.pad(ref, (pad_w, pad_w, 0, 0), mode={"zeros": "constant"}.get(mode, mode))
    ref8 = torch.zeros((N, (c + 7) // 8 * 8, H, wp))
    ref8[:, :c] = ref
    ref_cb = ref8.view(N, -1, 8, H, wp).permute(0, 1, 3, 4, 2).to(tdt)
    assert torch.equal(out.cpu().view(torch.int16), ref_cb.contiguous().view(torch.int16))
