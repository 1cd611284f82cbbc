"""Deterministic synthetic 2-D mantle-like fields shared by tools/make_golden.py (which feeds
them to the imported reference) and the tests (which feed them to the oracle / the HIP
path).  numpy's legacy RandomState stream is stable across numpy versions, so large
inputs never have to be committed — only the reference's outputs are."""
import numpy as np


def smooth_field(B, H, W, seed, amp=0.1, modes=6, noise=0.0):
    rng = np.random.RandomState(seed)
    y = np.linspace(0.0, 1.0, H)[:, None]
    x = np.linspace(0.0, 4.0, W)[None, :]
    out = np.zeros((B, H, W))
    for b in range(B):
        for _ in range(modes):
            a = rng.uniform(-1, 1)
            kx, ky = rng.uniform(0.5, 7.0), rng.uniform(0.5, 9.0)
            p1, p2 = rng.uniform(0, 6.28), rng.uniform(0, 6.28)
            out[b] += a * np.sin(kx * x + p1) * np.cos(ky * y + p2)
        if noise:
            out[b] += noise * rng.standard_normal((H, W))
    return (amp * out).astype(np.float32).astype(np.float64)


def temperature_field(B, H, W, seed):
    """Conductive profile + a few plumes, clipped to [0, 1.35]; row 0 = bottom (T=1)."""
    rng = np.random.RandomState(seed)
    y = np.linspace(0.0, 1.0, H)[:, None]
    x = np.linspace(0.0, 4.0, W)[None, :]
    out = np.zeros((B, H, W))
    for b in range(B):
        T = 1.0 - y + 0 * x
        for _ in range(4):
            a, s = rng.uniform(0.05, 0.3), rng.uniform(0.03, 0.15)
            cx, cy = rng.uniform(0, 4), rng.uniform(0, 1)
            T = T + a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s))
        T = np.clip(T, 0.0, 1.35)
        T[0, :] = 1.0
        T[-1, :] = 0.0
        out[b] = T
    return out.astype(np.float32).astype(np.float64)


def sim_parameters(B, seed):
    """(RaQ, FKT, FKP) drawn from the dataset's ranges (reference datasetio.py:124-136)."""
    rng = np.random.RandomState(seed)
    raq = rng.uniform(0.12624371, 9.70723344, B)
    fkt = 10.0 ** rng.uniform(6.00352841978384, 9.888820429862925, B)
    fkp = 10.0 ** rng.uniform(0.005251646002323797, 1.9927988938926755, B)
    return np.stack([raq, fkt, fkp], axis=1)


def unet_input(B, H, W, seed, c_i=11):
    """gVTp-like input: xc/4, yc/4, dt, 3 broadcast scalars, V, T, u, v[, p]."""
    rng = np.random.RandomState(seed + 77)
    y = np.broadcast_to(np.linspace(0.0, 1.0, H)[:, None], (H, W))
    x = np.broadcast_to(np.linspace(0.0, 4.0, W)[None, :], (H, W))
    T = temperature_field(B, H, W, seed + 1)
    u = smooth_field(B, H, W, seed + 2)
    v = smooth_field(B, H, W, seed + 3)
    p = smooth_field(B, H, W, seed + 4, amp=0.5)
    chans = []
    for b in range(B):
        sc = rng.uniform(0, 1, 4)
        V = np.log10(np.clip(np.exp(-16.0 * T[b] + 2.0 * (1 - y)), 1e-8, 1.0)) / 8.0
        c = [x / 4, y / 4, np.full((H, W), sc[0] * 1e-4), np.full((H, W), sc[1]),
             np.full((H, W), sc[2]), np.full((H, W), sc[3]), V, T[b], u[b], v[b], p[b]]
        chans.append(np.stack(c[:c_i], 0))
    return np.stack(chans, 0).astype(np.float32).astype(np.float64)


def strided_sample(a, n=257):
    """A fixed, size-independent subsample of a tensor (flattened, prime stride)."""
    f = np.asarray(a).reshape(-1)
    step = max(1, f.size // n)
    return f[::step][:n].copy()
