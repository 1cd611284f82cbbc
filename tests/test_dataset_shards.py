"""`datasetio.NewADDataset` on shard files in the reference's directory layout, against the items the REFERENCE's own class
returned for the same files (tests/golden/g18_newad_dataset.npz, written by tools/make_golden.py g18), and — on the GPU —
the HBM-resident form `ResidentNewADDataset` (mc_assemble_newad_batch) against the host items (SURVEY 8f row N2,
reference datasetio.py:320-654)."""
import os

import numpy as np
import pytest
import torch


def _write_tree(g, root):
    sims = [(int(n), str(a), *[float(v) for v in par[:5]], int(par[5])) for n, a, par in zip(g["sims_num"], g["sims_an"], g["sims_par"])]
    torch.save(sims, os.path.join(root, "sims.pt"))
    for k in g.files:
        if not k.startswith("file/"):
            continue
        _, an, sim, name = k.split("/")
        d = os.path.join(root, an, sim)
        os.makedirs(d, exist_ok=True)
        torch.save(torch.from_numpy(g[k]), os.path.join(d, name + ".pt"))
    return sims


CASES = {
    "all": dict(an="train", is_init=False, p_pred=True, debug=False),
    "init": dict(an="train", is_init=True, p_pred=True, debug=False),
    "snaps": dict(an="train", is_init=False, p_pred=False, debug=True),
    "cv": dict(an="cv", is_init=False, p_pred=False, debug=False),
    "filtered": dict(an="train", is_init=False, p_pred=True, debug=False),
    "half": dict(an="train", is_init=False, p_pred=True, debug=False, max_examples_percent_per_epoch=50),
}


def _dataset(golden, tmp_path, name):
    from pbml_mantle_convection_amd.datasetio import NewADDataset
    g = golden("g18_newad_dataset")
    _write_tree(g, str(tmp_path))
    kw = dict(CASES[name])
    if name == "filtered":
        kw.update(sims_vec=g["filtered/sims_vec"].tolist(), times_vec=g["filtered/times_vec"].tolist())
    return g, NewADDataset(str(tmp_path), scale=True, load=False, noise=0.0, **kw)


@pytest.mark.parametrize("name", list(CASES))
def test_newad_dataset_matches_reference_items(golden, tmp_path, name):
    g, ds = _dataset(golden, tmp_path, name)
    assert len(ds) == int(g[f"{name}/n"])
    for i in range(len(ds)):
        x, y, t, s = ds[i]
        assert x.dtype == torch.float64 and tuple(x.shape) == g[f"{name}/x"].shape[1:]
        np.testing.assert_allclose(x.numpy(), g[f"{name}/x"][i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(y.numpy(), g[f"{name}/y"][i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(float(t), g[f"{name}/t"][i], rtol=1e-14)
        np.testing.assert_allclose(float(torch.as_tensor(s).reshape(())), g[f"{name}/s"][i], rtol=1e-12)


def test_newad_dataset_debug_with_pressure_raises(golden, tmp_path):
    from pbml_mantle_convection_amd.datasetio import NewADDataset
    _write_tree(golden("g18_newad_dataset"), str(tmp_path))
    with pytest.raises(ValueError):
        NewADDataset(str(tmp_path), "train", debug=True, p_pred=True)


def test_load_train_objs_builds_newad_datasets_for_the_fluidnet_family(golden, tmp_path, monkeypatch):
    """The reference trains newfluidnet on NewADDataset shards (multigpu.py:684, 726-760); this used to raise.  (The model
    itself needs the GPU: a stand-in module keeps this a host-logic test.)"""
    from pbml_mantle_convection_amd import multigpu as G
    monkeypatch.setattr(G, "build_model", lambda *a, **k: torch.nn.Linear(2, 2))
    g = golden("g18_newad_dataset")
    os.makedirs(tmp_path / "data")
    _write_tree(g, str(tmp_path / "data"))
    os.makedirs(tmp_path / "nn")
    sv = {"train": [3, 7], "cv": [5]}
    tv = {"train": [int(g["file/train/sim_3/e1_i_vec_select"][0]), int(g["file/train/sim_7/e1_i_vec_select"][1])],
          "cv": [int(g["file/cv/sim_5/e1_i_vec_select"][2])]}
    svi = {"train": [3], "cv": [5]}
    tvi = {"train": [int(g["file/train/sim_3/e1_i_vec_select_init"][0])], "cv": [int(g["file/cv/sim_5/e1_i_vec_select_init"][1])]}
    ds, ds_init, model, _, opt, sch, epoch = G.load_train_objs(
        0, 1, str(tmp_path / "nn") + "/", str(tmp_path / "data"), 2, 7, 8, 3, "gelu", "zeros", "mass", True, 1, 3, [5],
        sv, tv, svi, tvi, p_pred=True, network="newfluidnet", debug=False)
    assert len(ds["train"]) == 2 and len(ds["cv"]) == 1
    assert len(ds_init["train"]) == 1 and len(ds_init["cv"]) == 1
    x, y, t, s = ds["train"][0]
    assert tuple(x.shape) == (7, 10, 14) and tuple(y.shape) == (3, 10, 14)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["all", "cv"])
def test_resident_newad_dataset_assembles_batches_on_device(golden, tmp_path, name):
    from pbml_mantle_convection_amd.datasetio import ResidentNewADDataset
    g, ds = _dataset(golden, tmp_path, name)
    res = ResidentNewADDataset(ds, "cuda:0")
    assert len(res) == len(ds)
    idx = [len(ds) - 1, 0, 1, 0]
    x, y, t, s = res.assemble(idx)
    torch.cuda.synchronize()
    for b, i in enumerate(idx):
        xr, yr, tr, sr = ds[i]
        np.testing.assert_allclose(x[b].cpu().numpy(), xr.numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(y[b].cpu().numpy(), yr.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(float(t[b]), float(tr), rtol=1e-6)
        np.testing.assert_allclose(float(s[b]), float(torch.as_tensor(sr).reshape(())), rtol=2e-5)
    # straight into preallocated buffers (what Trainer.input_buffers() hands out)
    out = dict(gVTp=torch.zeros_like(x), uvp=torch.zeros_like(y))
    x2, y2, _, _ = res.assemble(idx, out=out)
    assert x2.data_ptr() == out["gVTp"].data_ptr() and torch.equal(x2, x) and torch.equal(y2, y)
    with pytest.raises(IndexError):
        res.assemble([len(ds)])


AD_CASES = {
    "all": dict(an="train", p_pred=True, debug=False, roll_forward=1),
    "roll2": dict(an="train", p_pred=False, debug=False, roll_forward=2),
    "debug": dict(an="cv", p_pred=False, debug=True, roll_forward=1),
    "filtered": dict(an="train", p_pred=True, debug=False, roll_forward=1),
}


def _adtime(golden, tmp_path, name):
    from pbml_mantle_convection_amd.datasetio import ADTimeDataset
    g = golden("g19_adtime_dataset")
    _write_tree(g, str(tmp_path))
    kw = dict(AD_CASES[name])
    if name == "filtered":
        kw.update(sims_vec=g["filtered/sims_vec"].tolist(), times_vec=g["filtered/times_vec"].tolist())
    return g, ADTimeDataset(str(tmp_path), scale=True, load=False, noise=0.0, **kw)


@pytest.mark.parametrize("name", list(AD_CASES))
def test_adtime_dataset_matches_reference_items(golden, tmp_path, name):
    """ADTimeDataset (the U-Net's dataset, reference datasetio.py:63-280) read from files in the reference's layout: index
    pairs, initial-condition pairs and every item equal to what the reference's class returned (g19)."""
    import random
    g, ds = _adtime(golden, tmp_path, name)
    assert len(ds) == int(g[f"{name}/n"])
    assert np.array_equal(np.array(ds.indices).reshape(-1, 2), g[f"{name}/indices"])
    assert np.array_equal(np.array(ds.indices_init).reshape(-1, 2), g[f"{name}/indices_init"])
    for i in range(len(ds)):
        random.seed(1000 + i)
        x, y, s, paras, yc = ds[i]
        np.testing.assert_allclose(x.numpy(), g[f"{name}/x"][i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(y.numpy(), g[f"{name}/y"][i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(float(torch.as_tensor(s).reshape(())), g[f"{name}/s"][i], rtol=1e-12)
        np.testing.assert_allclose(paras.reshape(3).numpy(), g[f"{name}/paras"][i], rtol=1e-14)
        np.testing.assert_allclose(yc.numpy(), g[f"{name}/yc"], rtol=0, atol=0)


@pytest.mark.gpu
def test_resident_adtime_dataset_on_reference_layout_files(golden, tmp_path):
    """The HBM-resident form fed from shard files (not synthetic members): device batches == stacked host items."""
    import random
    from pbml_mantle_convection_amd.datasetio import ResidentADTimeDataset
    g, ds = _adtime(golden, tmp_path, "all")
    res = ResidentADTimeDataset(ds, "cuda:0")
    idx = [1, 2, 3, 5]                                   # (items whose first index is a multiple of 8 draw a random pair)
    pairs = [tuple(ds.indices[i]) for i in idx]
    x, y, sc, pa, yc = res.assemble(idx, pairs=pairs)
    torch.cuda.synchronize()
    for b, i in enumerate(idx):
        if ds.indices[i][0] % 8 == 0:
            continue
        random.seed(0)
        xr, yr, sr, pr, ycr = ds[i]
        np.testing.assert_allclose(x[b].cpu().numpy(), xr.numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(y[b].cpu().numpy(), yr.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(float(sc[b]), float(torch.as_tensor(sr).reshape(())), rtol=2e-5)
