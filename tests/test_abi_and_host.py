"""CPU-only checks: the C-ABI library loads and exports every symbol include/mantle_hip.h declares, the
ctypes signatures cover the header, module structure / state_dict keys match the reference, host logic
(graphs, bicubic tables, sharding, CLI, datasets, scaler) and the loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mantle_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pbml_mantle_convection_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mantle_hip.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert _lib.call("mc_version") >= 100
    assert _lib.load().mc_strerror(-2) == b"unsupported configuration"


def test_struct_layouts_match_header():
    import ctypes as C
    from pbml_mantle_convection_amd import _lib
    assert C.sizeof(_lib.ConvDesc) == 15 * 4
    assert C.sizeof(_lib.GradSrc) == 8 + 8 * 4
    assert C.sizeof(_lib.LossDesc) == 12 * 4


def test_argument_validation_without_gpu():
    """Descriptor validation happens before any launch, so it can be exercised on CPU."""
    import ctypes as C
    from pbml_mantle_convection_amd import _lib as L
    L.load()
    ok = L.ConvDesc(2, 40, 60, 16, 0, 16, 5, 2, 2, 0, 4, 0, 0)
    assert L.call("mc_conv_tiles", C.byref(ok)) == 3 * 4
    assert L.call("mc_packed_weight_bytes", C.byref(ok), 0) == 2 * 25 * 8 * 16 * 4
    bad_k = L.ConvDesc(2, 40, 60, 16, 0, 16, 4, 2, 2, 0, 4, 0, 0)
    assert L.call("mc_conv_tiles", C.byref(bad_k)) == -1
    odd_sym = L.ConvDesc(2, 40, 60, 16, 0, 16, 5, 2, 2, 0, 3, 0, 0)
    assert L.call("mc_conv_tiles", C.byref(odd_sym)) == -1
    with pytest.raises(L.MantleHipError):
        L.call("mc_pack_nchw", None, 1, 1, 1, 4, 4, 0, 0, None, 0, None, None)


def test_module_tree_matches_reference(golden):
    from pbml_mantle_convection_amd.pytorch_networks_convae import ConvAE, Unet, count_parameters
    g = golden("g10_known_answers")
    m = Unet(5, 11, 16, 4, torch.device("cpu"), "gelu", "reflect", "mae", use_symm=True, repeats=3, f=5, p_pred=True)
    assert count_parameters(m) == int(g["unet_cfg2"]) == 1820030
    assert list(m.state_dict().keys()) == list(g["unet_keys"])
    assert [",".join(map(str, v.shape)) for v in m.state_dict().values()] == list(g["unet_shapes"])
    c = ConvAE(2, 3, 16, 3, None, "gelu", "reflect", "mae", use_symm=True, repeats=2, f=3, p_pred=True)
    assert count_parameters(c) == int(g["convae_cfg1"]) == 860301
    assert list(c.state_dict().keys()) == list(g["convae_keys"])
    assert [",".join(map(str, v.shape)) for v in c.state_dict().values()] == list(g["convae_shapes"])
    assert count_parameters(ConvAE(2, 3, 16, 3, None, "gelu", "reflect", "mae", use_symm=False, repeats=2, f=3)) == \
        int(g["convae_cfg1_plain"])
    # the deployed multi-resolution trunk (known parameter count of the reference's own configuration)
    from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
    nf = NewFluidNet(5, 7, 64, 1, None, "gelu", "zeros", "curl", use_symm=False, a_bound=10, repeats=4, f=5, p_pred=False, factor=2)
    assert count_parameters(nf) == int(g["newfluidnet"]) == 2289281
    g12 = golden("g12_newfluidnet_mae_zeros")
    small = NewFluidNet(3, 7, 8, 3, None, "gelu", "zeros", "mae", use_symm=True, repeats=2, f=5, p_pred=True)
    ref = {k[3:]: g12[k].shape for k in g12.files if k.startswith("sd/")}
    assert {k: tuple(v.shape) for k, v in small.state_dict().items()} == ref
    with pytest.raises(NotImplementedError):
        NewFluidNet(3, 7, 12, 3, None, "gelu", "zeros", "mae")            # c_h must be a multiple of 8 on the HIP path


def test_no_cpu_fallback():
    from pbml_mantle_convection_amd.pytorch_networks_convae import FluidLayer, Unet
    m = Unet(3, 10, 8, 3, torch.device("cpu"), "gelu", "reflect", "curl", use_symm=True, repeats=2, f=5, p_pred=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 10, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FluidLayer(4, 8)(torch.zeros(1, 4, 8, 8))


def test_unsupported_options_fail_loudly():
    from pbml_mantle_convection_amd.pytorch_networks_convae import FluidLayer, Unet
    with pytest.raises(NotImplementedError):
        FluidLayer(4, 8, act_fn="sine")
    with pytest.raises(NotImplementedError):
        FluidLayer(4, 8, r_p="circular")
    with pytest.raises(NotImplementedError):
        Unet(3, 10, 8, 3, spectral_conv=True)
    with pytest.raises(NotImplementedError):
        Unet(3, 10, 8, 3, drop_rate=0.1)


def test_graph_wiring_matches_layer_table():
    from oracle import ref_cpu as O
    from pbml_mantle_convection_amd.engine import convae_graph, unet_graph
    g = unet_graph(5, 11, 16, 4, act="gelu", r_p="reflect", use_symm=True, repeats=3, f=5)
    convs = [n for n in g.nodes if n.kind == "conv"]
    tab = O.unet_layer_table(5, 11, 16, 4, 3)
    assert len(convs) == len(tab) == 27
    for n, (prefix, cin, cout, kind) in zip(convs, tab):
        assert sum(g.channels[s] for s in n.srcs) == cin and n.c_out == cout
        assert n.name.startswith(prefix)
    assert sum(1 for n in g.nodes if n.kind == "up") == 4
    assert sum(1 for n in convs if n.pool == 2) == 4
    ga = convae_graph(2, 3, 16, 3, act="gelu", r_p="reflect", use_symm=True, repeats=2, f=3, loss_type="mae")
    ops = [o for o in O.convae_op_table(2, 3, 16, 3, 2) if o[0] in ("fluid", "final")]
    cv = [n for n in ga.nodes if n.kind == "conv"]
    assert [(sum(ga.channels[s] for s in n.srcs), n.c_out) for n in cv] == [(o[2], o[3]) for o in ops]
    assert [n.name.split(".")[1] for n in cv] == [str(o[1]) for o in ops]


def test_every_launch_reads_inside_its_filter_bank_and_workspaces():
    """Host-side guard against out-of-bounds bank reads (round 2's GPU memory fault: the f32 kernel read 32 B past a bank whose
    padded output-channel count is not a multiple of 16): for EVERY convolution launch of the CFG-1 / CFG-2/3 / CFG-5 / N1 /
    N4 graphs, in every precision, the kernel family's own reach into packed_w (mc_conv_bank_read_extent, restated from the
    kernels' indexing) stays inside mc_packed_weight_bytes, for the forward and the input-gradient bank; the partial-sum and
    filter-gradient workspaces are sized from the same descriptors."""
    import ctypes as C
    from pbml_mantle_convection_amd import _lib as L
    from pbml_mantle_convection_amd import engine as E
    lib = L.load()
    graphs = {
        "cfg1 convae": (E.convae_graph(2, 3, 16, 3, act="gelu", r_p="reflect", use_symm=True, repeats=2, f=3, loss_type="mae"), 4, 128, 128),
        "cfg3 unet": (E.unet_graph(5, 10, 16, 4, act="gelu", r_p="reflect", use_symm=True, repeats=3, f=5), 32, 506, 506),
        "cfg5 unet 1024": (E.unet_graph(5, 10, 16, 4, act="gelu", r_p="reflect", use_symm=True, repeats=3, f=5), 1, 1024, 1024),
        "small unet": (E.unet_graph(3, 11, 8, 4, act="gelu", r_p="zeros", use_symm=True, repeats=2, f=5), 2, 44, 70),
        "n1 newfluidnet": (E.newfluidnet_graph(5, 7, 16, 3, act="gelu", r_p="zeros", use_symm=True, repeats=6, f=5), 32, 128, 506),
        "n4 unet learned": (E.unet_graph(3, 10, 16, 4, act="gelu", r_p="learned", use_symm=True, repeats=2, f=5), 2, 64, 96),
        "n4 newfluidnet learned": (E.newfluidnet_graph(3, 7, 16, 4, act="gelu", r_p="learned", use_symm=True, repeats=2, f=5), 2, 64, 96),
    }
    checked = 0
    for gname, (g, N, H, W) in graphs.items():
        for prec in ("fp32", "bf16", "mixed"):
            for name, d, dd in E.iter_conv_descs(g, N, H, W, prec):
                tag = (gname, prec, name)
                nbytes = lib.mc_packed_weight_bytes(C.byref(d), 0)
                ext = lib.mc_conv_bank_read_extent(C.byref(d))
                assert nbytes > 0 and 0 < ext <= nbytes, (tag, "forward", ext, nbytes)
                assert lib.mc_conv_tiles(C.byref(d)) > 0 and lib.mc_wgrad_partial_bytes(C.byref(d)) > 0, tag
                if dd is not None:
                    nb1 = lib.mc_packed_weight_bytes(C.byref(d), 1)
                    ext1 = lib.mc_conv_bank_read_extent(C.byref(dd))
                    assert nb1 > 0 and 0 < ext1 <= nb1, (tag, "input gradient", ext1, nb1)
                checked += 1
    assert checked > 600, checked


def test_bicubic_tables_match_aten(golden):
    """Host-built tap tables reproduce nn.Upsample(mode='bicubic') (golden g3a/g3b) and their transposes its adjoint."""
    from pbml_mantle_convection_amd.engine import bicubic_tables
    for name, (ho, wo) in (("g3a_bicubic_size", (63, 64)), ("g3b_bicubic_x4", (32, 32))):
        g = golden(name)
        x, y, ct, dx = g["x"], g["y"], g["ct"], g["dx"]
        iy, wy, sy, jy, twy = bicubic_tables(x.shape[2], ho)
        ix, wx, sx, jx, twx = bicubic_tables(x.shape[3], wo)
        My = np.zeros((ho, x.shape[2])); Mx = np.zeros((wo, x.shape[3]))
        for o in range(ho):
            for k in range(4):
                My[o, iy[o, k]] += wy[o, k]
        for o in range(wo):
            for k in range(4):
                Mx[o, ix[o, k]] += wx[o, k]
        got = np.einsum("oy,ncyx,px->ncop", My, x, Mx)
        np.testing.assert_allclose(got, y, atol=2e-6)
        MyT = np.zeros_like(My)
        for i in range(x.shape[2]):
            for a in range(sy[i], sy[i + 1]):
                MyT[jy[a], i] += twy[a]
        np.testing.assert_allclose(MyT, My, atol=1e-7)
        np.testing.assert_allclose(np.einsum("oy,ncop,px->ncyx", My, ct, Mx), dx, atol=2e-5)


def test_shard_range_and_cli():
    from pbml_mantle_convection_amd import multigpu as G
    assert [G.shard_range(103, 8, r) for r in (0, 7)] == [(0, 12), (84, 96)]
    spans = [G.shard_range(64, 4, r) for r in range(4)]
    assert spans == [(0, 16), (16, 32), (32, 48), (48, 64)]
    a = G.build_arg_parser().parse_args("-net unet -l 5 -f 16 -r 3 -k 5 -s 1 -p reflect -lt mass -pp 1 -b 4 -ab 10".split())
    assert (a.levels, a.c_h, a.repeats, a.kernel, a.use_symm, a.r_p, a.loss_type, a.p_pred, a.batch_size) == \
        (5, 16, 3, 5, 1, "reflect", "mass", 1, 4)
    assert a.act_fn == "gelu" and a.master_port == 366 and a.loss_scale == 1 and a.roll_forward == 1
    assert G.run_name(a) == ("unet_levels_5_gelu_16_reflect_mass_True_ab10_b4_r3_k5_fa2_adFalse_p_predTrue_l20.0_"
                             "l_scTrue_l_deFalse_debFalse_roll1_new")
    assert G.channels_for("unet", "curl", False) == (10, 2) and G.channels_for("unet", "mass", True) == (11, 4)
    assert G.channels_for("convae", "mae", True) == (3, 3) and G.channels_for("fluidnet", "curl", False) == (7, 1)


def test_restart_log_roundtrip(tmp_path):
    from pbml_mantle_convection_amd import multigpu as G
    d = str(tmp_path) + "/"
    with open(d + "fluidnet_uvpT.txt", "w") as f:
        f.write("Epoch, train loss, val loss, learning rate \n")
        f.write("0,[0.1, 0.2, 0.3, 0.4, 0.5],[0.1, 0.2, 0.3, 0.4, 0.5],0.001\n")
        f.write("25,[0.1, 0.2, 0.3, 0.4, 0.5],[0.1, 0.2, 0.3, 0.4, 0.5],0.0005\n")
    epoch, lr, ms = G.parse_restart_log(d, [20, 40, 60, 80, 120, 180])
    assert (epoch, lr, ms) == (25, 0.0005, [15, 35, 55, 95, 155])


def test_scaler_and_helpers(golden):
    from pbml_mantle_convection_amd import pytorch_networks_convae as P
    from pbml_mantle_convection_amd import scaler as SC
    g = golden("g9_helpers")
    ones = np.ones((2, 3))
    np.testing.assert_allclose(SC.scale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "uprev"), g["scale_u"])
    np.testing.assert_allclose(SC.unscale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "vprev"), g["unscale_v"])
    np.testing.assert_allclose(SC.scale_var(ones.copy(), 4.21479129, 86422511.6, 3.01635241, "Tprev"), ones)
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    np.testing.assert_allclose(P.eta_torch(t("gamma"), t("beta"), t("z"), t("T")).numpy(), g["eta"])
    pu, pv, pp = P.pad_uvp(t("u"), t("v"), t("p"))
    np.testing.assert_allclose(pu.numpy(), g["pu"]); np.testing.assert_allclose(pv.numpy(), g["pv"])
    np.testing.assert_allclose(pp.numpy(), g["pp"])
    np.testing.assert_allclose(P.pad_grad(t("g"), (1, 2, 1, 2)).numpy(), g["pg"])
    g8 = golden("g8_fd_kernels")
    x = torch.from_numpy(g8["x"])
    for name in ("dx_right", "dx_left", "dy_bot", "dy_top", "dx_center", "dy_center", "du_dy", "dv_dx", "laplace"):
        np.testing.assert_allclose(getattr(P, name)(x, torch.device("cpu")).numpy(), g8[name], atol=1e-12)
    g7 = golden("g7_get_mass")
    import fields
    u = torch.from_numpy(fields.smooth_field(2, 128, 506, 700, noise=0.01))
    v = torch.from_numpy(fields.smooth_field(2, 128, 506, 701, noise=0.01))
    for bc in (0, 1):
        np.testing.assert_allclose(fields.strided_sample(P.get_mass(u, v, bc=bool(bc)).numpy()), g7[f"sample_bc{bc}"],
                                   atol=1e-14)


def test_synthetic_dataset_layout():
    from pbml_mantle_convection_amd.datasetio import SyntheticMantleDataset, normalise_parameters, synthetic_batch
    ds = SyntheticMantleDataset(5, 24, 40, p_pred=True, seed=3)
    x, y, scaler, paras, yc = ds[2]
    assert x.shape == (11, 24, 40) and y.shape == (4, 24, 40) and paras.shape == (3, 1, 1) and yc.shape == (1, 24, 40)
    g1 = synthetic_batch(2, 24, 40, 11)
    g2 = synthetic_batch(2, 24, 40, 11)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))          # seeded
    u, v = g1[1][:, 0].double(), g1[1][:, 1].double()
    assert float(g1[0][:, 7].max()) <= 1.35 and float(g1[0][:, 7].min()) >= 0.0
    nd = normalise_parameters(9.70723344, 10 ** 9.888820429862925, 10 ** 1.9927988938926755)
    np.testing.assert_allclose(nd, (1.0, 1.0, 1.0), atol=1e-12)
