"""Host-side executor of the Stokes-surrogate network on libmantle_hip (MI355X).

A network (Unet, ConvAE, or a single layer) is described as a small static graph of
conv / upsample nodes (`NetGraph`).  `Engine` resolves shapes for a given input size, owns
every device buffer (activations in the CB8 layout, GroupNorm statistics, filter banks,
gradient buffers, workspaces) and issues the C-ABI kernel calls for forward and backward
on the current HIP stream — so a whole training step can be captured into one HIP graph.

Reference call sites replaced: Unet.forward (pytorch_networks_convae.py:1985-2070),
ConvAE.forward (.ipynb_checkpoints/pycold-checkpoint.py:1094-1115), FluidLayer.forward
(:790-799) and the autograd backward of all of them.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

# "mixed" = MC_MIX16: every tensor of the forward pass in f16 (11 significant bits: what the momentum residual's second
# differences need), every gradient tensor in bf16 (range), MFMA arithmetic with f32 accumulation.  (Round 2's form of the
# same idea -- bf16 everywhere, the full-resolution level of the forward pass as bf16 (hi, lo) pairs -- cost 1.1 ms per step
# more and was removed in round 3.)
DTYPES = {"fp32": (L.MC_F32, torch.float32), "f32": (L.MC_F32, torch.float32),
          "bf16": (L.MC_BF16, torch.bfloat16), "mixed": (L.MC_MIX16, torch.float16)}


# ------------------------------------------------------------------------------------------------
# static graph
# ------------------------------------------------------------------------------------------------
@dataclass
class ConvNode:
    name: str                  # state-dict prefix of the conv weight/bias ("conv.0.layers.0." ...)
    srcs: List[int]            # tensor ids (1 or 2: torch.cat order)
    out: int                   # tensor id of the activated output
    c_out: int
    k: int
    pad: int
    sym_h: int                 # 0 = plain nn.Conv2d
    post: int                  # L.POST_*
    gn_name: Optional[str]     # state-dict prefix of the GroupNorm affine or None
    groups: int
    pool: int = 1              # AvgPool factor applied to the activated output (1 = none)
    pooled: int = -1           # tensor id of the pooled output
    kind: str = "conv"
    learned: bool = False      # BoundaryLearnedConvolution2D ("learned padding"): nine valid banks + one shared bias,
    bc_x: int = 1              # name + {conv, conv_top_left, ...}.weight / name + learnable_bias; bc > 1 widens the
    bc_y: int = 1              # border strips so that the output grows (Unet's first layer, reference :1995)
    sym_v: int = 0             # y-mirrored filters / filters mirrored about both axes (SymmetricConv2d symmetry 'v' / 'hv';
    sym_hv: int = 0            # FluidLayer never sets them, reference :755-757)


@dataclass
class UpNode:
    src: int
    out: int
    size: Optional[Tuple[int, int]] = None   # resolved at plan time (size of a named tensor)
    like: int = -1                           # take H, W of this tensor id ...
    scale: int = 0                           # ... or multiply by this integer scale factor
    kind: str = "up"


@dataclass
class PoolNode:                 # standalone nn.AvgPool2d(f, stride=f) (floor mode)
    src: int
    out: int
    f: int
    kind: str = "pool"


@dataclass
class CatNode:                  # torch.cat of several tensors along channels (every operand but the last: C % 8 == 0)
    srcs: List[int]
    out: int
    kind: str = "cat"


@dataclass
class NetGraph:
    c_in: int
    c_out: int
    channels: Dict[int, int]   # tensor id -> channel count
    nodes: list
    in_pad_w: int = 0          # F.pad(inputs, (3,3,0,0)) of the Unet
    crop_w: int = 0            # [..., 3:-3]
    subtract_mean: bool = False
    pad_mode: str = "zeros"
    act: str = "gelu"
    divisor: int = 1           # H, W must be divisible by this (ConvAE: 4**levels)


def _sym_h(c_o: int) -> int:
    # FluidLayer: h = c_o/4 (c_o/2 if c_o <= 4), v = hv = 0 (reference pytorch_networks_convae.py:755-757)
    return int(c_o / 4) if c_o > 4 else int(c_o / 2)


def _groups(c_o: int) -> int:
    return int(c_o / min(4, c_o))   # :788


def unet_graph(levels, c_i, c_h, c_o, *, act, r_p, use_symm, repeats, f) -> NetGraph:
    """Layer wiring of Unet.__init__/forward (reference pytorch_networks_convae.py:1842-2024).  r_p = 'learned': every
    conv is a BoundaryLearnedConvolution2D node; the input is not padded — the first layer's bc_x = 4 strips grow the field
    by the same 3 + 3 columns (:1990-1997) — and the two-operand concats are materialised (CatNode)."""
    learned = r_p == "learned"
    ch: Dict[int, int] = {0: c_i}
    nodes = []
    nid = [0]

    def new(c):
        nid[0] += 1
        ch[nid[0]] = c
        return nid[0]

    def one_src(srcs):
        if not learned or len(srcs) == 1:
            return list(srcs)
        cat = new(sum(ch[i] for i in srcs))
        nodes.append(CatNode(list(srcs), cat))
        return [cat]

    def fluid(prefix, srcs, c_out, pool=1, bc_x=1):
        srcs = one_src(srcs)
        out = new(c_out)
        node = ConvNode(prefix + "layers.0.", list(srcs), out, c_out, f, f // 2, _sym_h(c_out) if use_symm else 0,
                        L.POST_GN_ACT, prefix + "layers.1.", _groups(c_out), pool, learned=learned, bc_x=bc_x)
        if pool > 1:
            node.pooled = new(c_out)
        nodes.append(node)
        return node

    feat = {}
    cur = 0
    for r in range(repeats):
        last = r == repeats - 1
        n = fluid(f"conv.{r}.", [cur], c_h, pool=2 if (last and levels > 1) else 1, bc_x=4 if (learned and r == 0) else 1)
        cur = n.out
    feat[0] = n
    c = c_h
    for l in range(1, levels):
        cur = feat[l - 1].pooled
        for r in range(repeats):
            last = r == repeats - 1
            n = fluid(f"convs.{l - 1}.{r}.", [cur], c, pool=2 if (last and l < levels - 1) else 1)
            cur = n.out
        feat[l] = n
        c *= 2
    c = int(c / 2)
    xu = feat[levels - 1].out
    for li, l in enumerate(range(levels - 2, 0, -1)):
        up = new(ch[xu])
        nodes.append(UpNode(xu, up, like=feat[l].out))
        srcs = [feat[l].out, up]
        for r in range(repeats):
            n = fluid(f"upconvs.{li}.{r}.", srcs, int(c / 2))
            srcs = [n.out]
        xu = n.out
        c = int(c / 2)
    if levels > 1:
        up = new(ch[xu])
        nodes.append(UpNode(xu, up, like=feat[0].out))
        head_srcs = [up, feat[0].out]
    else:
        raise ValueError("Unet needs levels >= 2")
    R = repeats

    def head(name, srcs, c_out, post, gn_name, groups):
        o = new(c_out)
        nodes.append(ConvNode(name, one_src(srcs), o, c_out, f, f // 2, (_sym_h(c_out) if use_symm else 0) if learned else 0,
                              post, gn_name, groups, learned=learned))
        return o

    o = head(f"conv.{R}.", head_srcs, c, L.POST_GN_ACT, "gn.0.", int(c / 4))
    o2 = head(f"conv.{R + 1}.", [o], c, L.POST_ACT, None, 1)
    head(f"conv.{R + 2}.", [o2], c_o, L.POST_NONE, None, 1)
    return NetGraph(c_i, c_o, ch, nodes, in_pad_w=0 if learned else 3, crop_w=3, subtract_mean=True,
                    pad_mode="zeros" if learned else r_p, act=act, divisor=1)


def convae_graph(levels, c_i, c_h, c_o, *, act, r_p, use_symm, repeats, f, loss_type) -> NetGraph:
    """ConvAE.__init__ (reference .ipynb_checkpoints/pycold-checkpoint.py:1038-1092); returns the graph and
    keeps the ModuleList indices of the reference (pool / upsample modules consume an index)."""
    ch: Dict[int, int] = {0: c_i}
    nodes = []
    nid = [0]
    idx = [0]

    def new(c):
        nid[0] += 1
        ch[nid[0]] = c
        return nid[0]

    def fluid(src, c_out):
        out = new(c_out)
        p = f"conv.{idx[0]}."
        idx[0] += 1
        node = ConvNode(p + "layers.0.", [src], out, c_out, f, f // 2, _sym_h(c_out) if use_symm else 0,
                        L.POST_GN_ACT, p + "layers.1.", _groups(c_out))
        nodes.append(node)
        return node

    n = fluid(0, c_h)
    c = c_h
    for _ in range(levels):
        n.pool = 4
        n.pooled = new(n.c_out)
        idx[0] += 1                       # the AvgPool2d module
        cur = n.pooled
        cout = c * 4
        for r in range(repeats):
            n = fluid(cur, int(cout))
            cur = n.out
        c *= 4
    c = int(c / 4)
    cur = n.out
    for r in range(repeats):
        n = fluid(cur, c)
        cur = n.out
    for _ in range(levels, 0, -1):
        up = new(ch[cur])
        nodes.append(UpNode(cur, up, scale=4))
        idx[0] += 1                       # the Upsample module
        cur = up
        cout = int(c / 4)
        for r in range(repeats):
            n = fluid(cur, cout)
            cur = n.out
        c = int(c / 4)
    o = new(c_o)
    nodes.append(ConvNode(f"conv.{idx[0]}.", [cur], o, int(c_o), 3, 2 if loss_type == "curl" else 1, 0, L.POST_NONE,
                          None, 1))
    return NetGraph(c_i, int(c_o), ch, nodes, pad_mode=r_p, act=act, divisor=4 ** levels)


def newfluidnet_graph(levels, c_i, c_h, c_o, *, act, r_p, use_symm, repeats, f, factor=2) -> NetGraph:
    """Layer wiring of NewFluidNet.__init__/forward (reference pytorch_networks_convae.py:1215-1346): level l = the input
    feature map average-pooled l times, `repeats` FluidLayers, bicubic back to the input size; six-way concat with the raw
    inputs; 3 x 3 head.  The reference re-pools the feature map from scratch for every level; the values are identical to
    pooling the previous level once more, which is what the graph does."""
    if c_h % 8:
        raise NotImplementedError("the HIP path of NewFluidNet needs c_h to be a multiple of 8 (channel-block concat)")
    learned = r_p == "learned"          # every conv a BoundaryLearnedConvolution2D; the head then uses k = f (reference :1296-1313)
    ch: Dict[int, int] = {0: c_i}
    nodes = []
    nid = [0]

    def new(c):
        nid[0] += 1
        ch[nid[0]] = c
        return nid[0]

    def fluid(prefix, src, c_out):
        out = new(c_out)
        node = ConvNode(prefix + "layers.0.", [src], out, c_out, f, f // 2, _sym_h(c_out) if use_symm else 0,
                        L.POST_GN_ACT, prefix + "layers.1.", _groups(c_out), learned=learned)
        nodes.append(node)
        return node

    x_in = fluid("conv.0.", 0, c_h).out
    pooled = x_in
    outs = []
    for l in range(levels):
        if l > 0:
            p = new(c_h)
            nodes.append(PoolNode(pooled, p, factor))
            pooled = p
        cur = pooled
        for r in range(repeats):
            cur = fluid(f"convs.{l}.{r}.", cur, c_h).out
        if l > 0:
            up = new(c_h)
            nodes.append(UpNode(cur, up, like=x_in))
            cur = up
        outs.append(cur)
    cat = new(c_h * levels + c_i)
    nodes.append(CatNode(outs + [0], cat))
    hk = f if learned else 3

    def hsym(c):
        return (_sym_h(c) if use_symm else 0) if learned else 0

    o = new(c_h)
    nodes.append(ConvNode("conv.1.", [cat], o, c_h, hk, hk // 2, hsym(c_h), L.POST_GN_ACT, "gn.0.", int(c_h / 4), learned=learned))
    o2 = new(c_h)
    nodes.append(ConvNode("conv.2.", [o], o2, c_h, hk, hk // 2, hsym(c_h), L.POST_ACT, None, 1, learned=learned))
    o3 = new(c_o)
    nodes.append(ConvNode("conv.3.", [o2], o3, c_o, hk, hk // 2, hsym(c_o), L.POST_NONE, None, 1, learned=learned))
    return NetGraph(c_i, c_o, ch, nodes, subtract_mean=True, pad_mode="zeros" if learned else r_p, act=act, divisor=1)


def single_layer_graph(c_in, c_out, k, pad, pad_mode, sym_h, post, act, groups, gn: bool, learned: bool = False,
                       sym_v: int = 0, sym_hv: int = 0) -> NetGraph:
    """One conv (+GN+act): SymmetricConv2d / FluidLayer used stand-alone."""
    ch = {0: c_in, 1: c_out}
    node = ConvNode("layers.0." if gn else "", [0], 1, c_out, k, pad, sym_h, post, "layers.1." if gn else None, groups,
                    learned=learned, sym_v=sym_v, sym_hv=sym_hv)
    return NetGraph(c_in, c_out, ch, [node], pad_mode=pad_mode, act=act)


def iter_conv_descs(g: NetGraph, N: int, H: int, W: int, precision: str):
    """(name, forward descriptor, input-gradient descriptor or None) of every convolution launch of the graph at input size
    N x H x W -- the same shape walk and descriptors as Engine.configure, without touching a device (host-side checks:
    tests/test_abi_and_host.py compares every launch's bank reach with the bank's size)."""
    mc, _ = DTYPES[precision]
    mcg = L.MC_BF16 if mc == L.MC_MIX16 else mc
    mode = L.PAD_MODES[g.pad_mode]
    size = {0: (H, W + 2 * g.in_pad_w)}
    grad = {0: False}
    for node in g.nodes:
        if node.kind == "up":
            size[node.out] = size[node.like] if node.like >= 0 else tuple(v * node.scale for v in size[node.src])
            grad[node.out] = True
            continue
        if node.kind == "pool":
            size[node.out] = tuple(v // node.f for v in size[node.src])
            grad[node.out] = grad[node.src]
            continue
        if node.kind == "cat":
            size[node.out] = size[node.srcs[0]]
            grad[node.out] = True
            continue
        h, w = size[node.srcs[0]]
        cs = [g.channels[i] for i in node.srcs]
        k = node.k
        if node.learned:
            pad_x = k + 1 + (node.bc_x - 1) if k == 5 else k + (node.bc_x - 1)
            pad_y = k + 1 + (node.bc_y - 1) if k == 5 else k + (node.bc_y - 1)
            fx, fy, mh, mw = pad_x - k + 1, pad_y - k + 1, h - k + 1, w - k + 1
            ho, wo = mh + 2 * fy, mw + 2 * fx
            for name, (sh, sw) in dict(conv=(h, w), conv_left=(h, pad_x), conv_right=(h, pad_x), conv_bottom=(pad_y, w),
                                       conv_top=(pad_y, w), conv_bottom_left=(pad_y, pad_x), conv_bottom_right=(pad_y, pad_x),
                                       conv_top_left=(pad_y, pad_x), conv_top_right=(pad_y, pad_x)).items():
                d = L.ConvDesc(N, sh, sw, cs[0], 0, node.c_out, k, 0, L.PAD_MODES["zeros"], mc, node.sym_h, 0, 0)
                dd = L.ConvDesc(N, sh - k + 1, sw - k + 1, node.c_out, 0, cs[0], k, k - 1, 0, mcg, 0, 0, 0)
                yield node.name + name, d, dd
        else:
            ho, wo = h + 2 * node.pad - k + 1, w + 2 * node.pad - k + 1
            final_f32 = node.post == L.POST_NONE and node is g.nodes[-1] and mc != L.MC_F32 and node.c_out <= 16
            d = L.ConvDesc(N, h, w, cs[0], cs[1] if len(cs) > 1 else 0, node.c_out, k, node.pad, mode, mc, node.sym_h, 0,
                           int(final_f32), node.sym_v, node.sym_hv)
            dd = None
            if any(grad[i] for i in node.srcs):
                dd = L.ConvDesc(N, ho, wo, node.c_out, 0, sum(cs), k, k - 1, 0, mcg, 0, cs[0] if len(cs) > 1 else 0, 0)
            yield node.name, d, dd
        size[node.out] = (ho, wo)
        grad[node.out] = True
        if node.pool > 1:
            size[node.pooled] = (ho // node.pool, wo // node.pool)
            grad[node.pooled] = True


# ------------------------------------------------------------------------------------------------
# bicubic tap tables (nn.Upsample(mode='bicubic', align_corners=False), A = -0.75), built in f64
# ------------------------------------------------------------------------------------------------
BICUBIC_WALK = os.environ.get("MANTLE_BICUBIC_WALK", "1") != "0"    # row-walk adjoint (mc_bicubic_bwd_walk) vs the tiled kernel


def bicubic_tables(n_in: int, n_out: int):
    A = -0.75
    scale = n_in / n_out
    o = np.arange(n_out, dtype=np.float64)
    real = scale * (o + 0.5) - 0.5
    i0 = np.floor(real)
    t = real - i0

    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    w = np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], 1)
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1).astype(np.int32)
    # transposed (CSR over input index)
    lists = [dict() for _ in range(n_in)]
    for oo in range(n_out):
        for k in range(4):
            d = lists[idx[oo, k]]
            d[oo] = d.get(oo, 0.0) + w[oo, k]
    start = np.zeros(n_in + 1, np.int32)
    tj, tw = [], []
    for i in range(n_in):
        for oo in sorted(lists[i]):
            tj.append(oo)
            tw.append(lists[i][oo])
        start[i + 1] = len(tj)
    return (idx, w.astype(np.float32), start, np.asarray(tj, np.int32), np.asarray(tw, np.float32))


# ------------------------------------------------------------------------------------------------
# engine
# ------------------------------------------------------------------------------------------------
@dataclass
class _T:            # runtime tensor
    C: int
    H: int = 0
    W: int = 0
    buf: Optional[torch.Tensor] = None
    requires_grad: bool = True
    gsrcs: list = field(default_factory=list)   # gradient sources registered during backward
    # "normalise on load": the activated tensor is never materialised; consumers read the producer's raw conv output
    # `raw` and apply act(scale * y + shift) from `coef` ([N, CP, 4] table, None = activation only) while staging
    fused: bool = False
    raw: Optional[torch.Tensor] = None
    coef: Optional[torch.Tensor] = None
    act: int = 0


class Engine:
    """Executes a NetGraph.  `params` maps state-dict names to f32 device tensors;
    `grads` maps the same names to f32 device tensors that backward ACCUMULATES into."""

    def __init__(self, graph: NetGraph, precision: str = "fp32"):
        if precision not in DTYPES:
            raise ValueError(f"precision must be one of {list(DTYPES)}")
        L.load()
        self.g = graph
        self.precision = "fp32" if precision == "f32" else precision
        self.mc_dtype, self.t_dtype = DTYPES[precision]
        # gradient tensors: bf16 in the "mixed" mode (their range), the forward type otherwise
        self.mc_gdtype, self.g_dtype = (L.MC_BF16, torch.bfloat16) if self.mc_dtype == L.MC_MIX16 else (self.mc_dtype, self.t_dtype)
        self.shape = None
        self._tables = {}
        # filter-gradient kernels on a second stream: 0 = never, 1 = every layer, 2 = only layers of <= 128 x 128 pixels
        # (their launches are latency-bound and leave most of the chip idle)
        self.wfin_per_layer = os.environ.get("MANTLE_WFIN_PER_LAYER", "1") != "0"   # A/B on MI355X: -0.14 ms/step (the combine leaves the tail of the step)
        # A/B on MI355X, round 2 (CFG-3, B = 32): 0 = 12.6 / 11.0 ms (mixed / bf16), 1 = 13.1 / 11.6, 2 = 12.8 / 11.3.  The two
        # full-chip kernels of a layer only time-slice when they run side by side (each takes twice its stand-alone time),
        # while every cross-queue dependency of the captured step costs 12-17 us (tools/step_timeline.py): one stream wins.
        # (Round 1 measured the opposite, -0.25 ms for 1, with the slower filter-gradient kernel of that time.)
        self.overlap_wgrad = int(os.environ.get("MANTLE_OVERLAP_WGRAD", "0"))
        self.overlap_pack = int(os.environ.get("MANTLE_OVERLAP_PACK", "0"))    # the three bank-packing launches beside the input pack
        # bit 0: GroupNorm + activation applied by the consumers on load (conv, filter gradient, bicubic) instead of a
        # stand-alone pass that materialises the activated tensor; bit 1: the GroupNorm-backward reduction fused into the
        # epilogue of the input-gradient launch (single-consumer tensors).  0 = the round-1 unfused chain (A/B, tests).
        self.fuse = int(os.environ.get("MANTLE_FUSE", "0"))
        # ... and only for tensors of at most this many pixels.  Measured on MI355X (CFG-3, B = 32): the MFMA kernels
        # of the two high-resolution levels are bound by vector-instruction issue, so GELU / GELU' evaluated inside them
        # costs more than the streaming pass it replaces (level-0 16->16 forward 117 -> 207 us against a 100 us pass);
        # at the deep levels the kernels are launch-latency bound and every fused pass is a launch saved.
        self.fuse_maxpix = int(os.environ.get("MANTLE_FUSE_MAXPIX", str(128 * 128)))
        # "mixed" mode, round 3: the GroupNorm-backward reduction rides in the input-gradient launch of every eligible layer
        # whose launch is a row-reuse launch (the wide levels).  y is f16 there, so dz = dA * GELU'(z) and the two sums cost
        # 8.5 packed-f16 / mixed-precision instructions per element on the loader waves, beside the next stage's MFMAs,
        # instead of the ~25 f32 instructions that made the same fusion lose in round 2.
        self.fuse_dz_rr = precision == "mixed" and os.environ.get("MANTLE_FUSE_DZ_RR", "1") != "0"
        # GroupNorm + activation of layers with at most this many pixels run as ONE launch per direction (statistics + apply
        # forward; reduce + finalize + apply backward).  MI355X, CFG-3: 64 x 64 and 32 x 32 layers 21 -> 11 us forward and
        # 38 -> 24 us backward; 128 x 128 layers break even (25 -> 24, 49 -> 53 us: 128 blocks do not fill the chip)
        self.gn_small_pix = int(os.environ.get("MANTLE_GN_SMALL_PIX", str(64 * 64 + 4)))

    # -------------------------------------------------------------- planning
    def freeze(self):
        """Pin the configured shape: a captured HIP graph holds this engine's device pointers, so re-planning for another
        batch / grid size would leave the graph replaying freed memory.  Further calls with a different shape raise."""
        self._frozen = True

    def configure(self, N: int, H: int, W: int, device):
        if self.shape == (N, H, W, str(device)):
            return
        if getattr(self, "_frozen", False):
            raise RuntimeError(f"this engine is pinned to input shape {self.shape[:3]} by a captured HIP graph; got {(N, H, W)} "
                               "(use a separate module / engine instance, or an un-captured trainer, for another shape)")
        g = self.g
        if H % g.divisor or W % g.divisor:
            raise ValueError(f"input H, W must be divisible by {g.divisor}")
        self.device = device
        self.N = N
        T: Dict[int, _T] = {tid: _T(C=c) for tid, c in g.channels.items()}
        T[0].H, T[0].W, T[0].requires_grad = H, W + 2 * g.in_pad_w, False
        mode = L.PAD_MODES[g.pad_mode]
        self.mode = mode
        f32 = dict(dtype=torch.float32, device=device)
        self.plan = []
        max_dy = 0
        max_wg = 0

        def cb8(c, h, w):
            return torch.empty((N, (c + 7) // 8, h, w, 8), dtype=self.t_dtype, device=device)

        def cb8g(c, h, w):                   # a gradient tensor
            return torch.empty((N, (c + 7) // 8, h, w, 8), dtype=self.g_dtype, device=device)

        T[0].buf = cb8(T[0].C, T[0].H, T[0].W)
        # consumers of every tensor (decides which activated tensors need not be materialised)
        cons: Dict[int, list] = {tid: [] for tid in g.channels}
        for node in g.nodes:
            for i in (node.srcs if node.kind in ("conv", "cat") else [node.src]):
                cons[i].append(node)
        self.cons = cons
        self.prod = {}                       # tensor id -> plan entry of the conv that produced it (full-resolution output)
        H0, W0 = T[0].H, T[0].W
        for node in g.nodes:
            if node.kind == "up":
                s = T[node.src]
                if node.like >= 0:
                    ho, wo = T[node.like].H, T[node.like].W
                else:
                    ho, wo = s.H * node.scale, s.W * node.scale
                o = T[node.out]
                o.H, o.W = ho, wo
                o.buf = cb8(o.C, ho, wo)
                tabs = (self._table(s.H, ho), self._table(s.W, wo))
                e = dict(node=node, tabs=tabs, dsrc=cb8g(s.C, s.H, s.W),
                         maxtaps=tuple(int(np.diff(bicubic_tables(a, b)[2]).max()) for a, b in ((s.H, ho), (s.W, wo))))
                if ho > 3 * s.H or wo > 3 * s.W:
                    # scale factor > 3: the adjoint runs as two 1-D passes through an f32 workspace (tap lists too long
                    # for the tiled kernel's window)
                    e["bws"] = torch.empty((N, (s.C + 7) // 8, s.H, wo, 8), **f32)
                self.plan.append(e)
                continue
            if node.kind == "pool":
                s, o = T[node.src], T[node.out]
                o.H, o.W = s.H // node.f, s.W // node.f
                o.buf = cb8(o.C, o.H, o.W)
                o.requires_grad = s.requires_grad
                self.plan.append(dict(node=node, dsum=cb8g(o.C, o.H, o.W)))
                continue
            if node.kind == "cat":
                srcs = [T[i] for i in node.srcs]
                o = T[node.out]
                o.H, o.W = srcs[0].H, srcs[0].W
                if any((t.H, t.W) != (o.H, o.W) for t in srcs) or any(t.C % 8 for t in srcs[:-1]):
                    raise ValueError("torch.cat operands must share H x W and all but the last need C % 8 == 0")
                o.buf = cb8(o.C, o.H, o.W)
                self.plan.append(dict(node=node))
                continue
            srcs = [T[i] for i in node.srcs]
            h, w = srcs[0].H, srcs[0].W
            for s in srcs:
                assert (s.H, s.W) == (h, w), "concat sources must agree in size"
            if node.learned:
                self.plan.append(self._plan_learned(node, srcs, T, cb8, cb8g, f32, N, device))
                e = self.plan[-1]
                max_dy = max(max_dy, N * e["coutp"] * T[node.out].H * T[node.out].W)
                continue
            final_f32 = (node.post == L.POST_NONE and node is g.nodes[-1] and self.mc_dtype != L.MC_F32
                         and node.c_out <= 16)
            ho, wo = h + 2 * node.pad - node.k + 1, w + 2 * node.pad - node.k + 1
            omode = int(final_f32)
            d = L.ConvDesc(N, h, w, srcs[0].C, srcs[1].C if len(srcs) > 1 else 0, node.c_out, node.k, node.pad,
                           mode, self.mc_dtype, node.sym_h, 0, omode, node.sym_v, node.sym_hv)
            o = T[node.out]
            o.H, o.W = ho, wo
            tiles = L.call("mc_conv_tiles", C.byref(d))
            if tiles <= 0:
                raise L.MantleHipError(f"unsupported convolution configuration for {node.name} "
                                       f"({self.precision}, c_in={srcs[0].C}+{d.c_in1}, c_out={node.c_out}, k={node.k})")
            coutp = ((node.c_out + 7) // 8) * 8
            cin_tot = sum(s.C for s in srcs)
            # dgrad = the same kernel on the padded domain: zero pad k-1, rotated/transposed bank
            dd = L.ConvDesc(N, ho, wo, node.c_out, 0, cin_tot, node.k, node.k - 1, 0, self.mc_gdtype, 0,
                            srcs[0].C if len(srcs) > 1 else 0, 0)
            need_dgrad = any(s.requires_grad for s in srcs)
            e = dict(node=node, desc=d, ddesc=dd, tiles=tiles, coutp=coutp,
                     Y=(torch.empty((N, (node.c_out + 7) // 8, ho, wo, 8), dtype=torch.float32, device=device)
                        if final_f32 else cb8(node.c_out, ho, wo)),
                     part=torch.empty((N, tiles, coutp, 2), **f32),
                     bank=torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 0), dtype=torch.uint8, device=device),
                     need_dgrad=need_dgrad)
            fusable = all(c.kind == "up" or (c.kind == "conv" and not c.learned) for c in cons[node.out])
            o.fused = bool((self.fuse & 1) and node.post != L.POST_NONE and fusable and (cons[node.out] or node.pool > 1)
                           and ho * wo <= self.fuse_maxpix)
            if o.fused:
                o.raw, o.act = e["Y"], L.ACTS[g.act]
            elif node.post != L.POST_NONE:
                o.buf = cb8(node.c_out, ho, wo)
            else:
                o.buf = e["Y"]
            self.prod[node.out] = e
            if node.post == L.POST_GN_ACT:
                e["stats"] = torch.empty((N, node.groups, 2), **f32)
                e["coef"] = torch.zeros((N, coutp, 4), **f32)       # (scale, shift, mean, rstd); padded channels stay 0
                if o.fused:
                    o.coef = e["coef"]
                blocks = L.call("mc_gn_bwd_blocks", ho, wo)
                e["gblocks"] = blocks
                e["gpart"] = torch.empty((N, blocks, coutp, 2), **f32)
                e["m12"] = torch.empty((N, node.groups, 2), **f32)
                # small layers: reduce + finalize + apply of the GroupNorm backward in one launch (mc_gn_act_bwd_small)
                if ho * wo <= self.gn_small_pix and (node.c_out // node.groups) in (1, 2, 4, 8):
                    e["pc"] = torch.empty((N, coutp, 2), **f32)
            if node.pool > 1:
                p = T[node.pooled]
                p.H, p.W = ho // node.pool, wo // node.pool
                p.buf = cb8(node.c_out, p.H, p.W)
            if need_dgrad:
                e["dbank"] = torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 1), dtype=torch.uint8,
                                         device=device)
                hp, wp = h + 2 * node.pad, w + 2 * node.pad
                e["dxp"] = [cb8g(s.C, hp, wp) for s in srcs]
                # GroupNorm-backward reduction fused into this launch's epilogue: the source is the full-resolution output
                # of a conv + (GN) + act layer and this conv is its only consumer
                pe = self.prod.get(node.srcs[0])
                dz_here = bool(self.fuse & 2) and h * w <= self.fuse_maxpix
                if (not dz_here and self.fuse_dz_rr and g.act == "gelu"
                        and L.load().mc_conv_kernel_name(C.byref(dd)).decode().startswith("k_conv_rr")):
                    dz_here = True
                if (dz_here and len(srcs) == 1 and pe is not None and pe["node"].post != L.POST_NONE
                        and not pe["node"].learned and len(cons[node.srcs[0]]) == 1 and pe["node"].pool == 1
                        ):
                    dtiles = L.call("mc_conv_tiles", C.byref(dd))
                    fblocks = L.call("mc_fold_blocks", h, w, node.pad, mode)
                    e["epi"] = pe
                    pe["dz_tiles"], pe["dz_blocks"] = dtiles, dtiles + fblocks
                    pe["dz_part"] = torch.empty((N, dtiles + fblocks, pe["coutp"], 2), **f32)
                    if "m12" not in pe and pe["node"].post == L.POST_GN_ACT:
                        pe["m12"] = torch.empty((N, pe["node"].groups, 2), **f32)
                    if (pe["node"].post == L.POST_GN_ACT and pe["dz_blocks"] > 256
                            and (pe["node"].c_out // pe["node"].groups) in (1, 2, 4, 8)):
                        pe["dz_pc"] = torch.empty((N, pe["coutp"], 2), **f32)
            max_dy = max(max_dy, N * coutp * ho * wo)
            # per-layer filter-gradient partial slabs: all layers are combined by ONE batched launch at the end of backward
            e["wpart"] = torch.empty(L.call("mc_wgrad_partial_bytes", C.byref(d)), dtype=torch.uint8, device=device)
            self.plan.append(e)
        self.T = T
        # two dY buffers: the filter gradient of layer L runs on a side stream while the main stream already
        # prepares dY of layer L-1 (see backward)
        self.dYs = [torch.empty(max_dy, dtype=self.g_dtype, device=device) for _ in range(2)]
        self.dY = self.dYs[0]
        # ... or one dY buffer per layer (default): the main chain then never waits for the side stream, the captured step is one
        # linear chain plus a side chain with fork edges only, and the graph executor keeps the chain on one hardware queue
        # (with the two alternating buffers the joins spread it over three queues: ~17 us per cross-queue hop, 2-3 per layer)
        self.dy_per_layer = os.environ.get("MANTLE_DY_PER_LAYER", "1") != "0" and self.overlap_wgrad != 0
        if self.dy_per_layer:
            for e in self.plan:
                if e["node"].kind == "conv":
                    o = T[e["node"].out]
                    e["dY"] = torch.empty(N * e["coutp"] * o.H * o.W, dtype=self.g_dtype, device=device)
        self.convs = [e for e in self.plan if e["node"].kind == "conv" and not e["node"].learned]
        self.side = torch.cuda.Stream(device=device)
        # events of the two-stream backward, created once (none is created inside a graph capture)
        self._events = [torch.cuda.Event() for _ in range(2 * len(self.plan) + 2)]
        last = self.plan[-1]
        assert last["node"].kind == "conv", "graph must end in a conv node"
        self.final_plain = last["node"].post == L.POST_NONE
        assert self.final_plain or not g.subtract_mean
        fo = T[last["node"].out]
        self.out_h, self.out_w = fo.H, fo.W - 2 * g.crop_w
        self.dOut = None if self.final_plain else cb8g(fo.C, fo.H, fo.W)
        self.chan_mean = torch.empty((N, g.c_out), **f32) if g.subtract_mean else None
        self.gmean = torch.empty((N, g.c_out), **f32) if g.subtract_mean else None
        self.shape = (N, H, W, str(device))

    # -------------------------------------------------------------- learned padding (BoundaryLearnedConvolution2D)
    LBANKS = ("conv", "conv_top_left", "conv_top_right", "conv_bottom_left", "conv_bottom_right", "conv_top", "conv_bottom",
              "conv_left", "conv_right")

    def _plan_learned(self, node, srcs, T, cb8, cb8g, f32, N, device):
        """Nine bias-free valid convolutions on the library's conv kernels, the strips cut / the frame assembled with
        mc_rect_copy (reference pytorch_networks_convae.py:1022-1065).  The gradient w.r.t. the input needs no padded
        domain: the adjoint of a valid convolution is exactly input-sized."""
        assert len(srcs) == 1, "learned padding takes one source tensor"
        s, o = srcs[0], T[node.out]
        h, w, k = s.H, s.W, node.k
        pad_x = k + 1 + (node.bc_x - 1) if k == 5 else k + (node.bc_x - 1)
        pad_y = k + 1 + (node.bc_y - 1) if k == 5 else k + (node.bc_y - 1)
        fx, fy, mh, mw = pad_x - k + 1, pad_y - k + 1, h - k + 1, w - k + 1
        if mh < 1 or mw < 1 or pad_x > w or pad_y > h:
            raise ValueError(f"{node.name}: input {h}x{w} too small for learned padding")
        ho, wo = mh + 2 * fy, mw + 2 * fx
        o.H, o.W = ho, wo
        regs = {"conv": (0, 0, h, w, fy, fx),
                "conv_left": (0, 0, h, pad_x, fy, 0), "conv_right": (0, w - pad_x, h, pad_x, fy, fx + mw),
                "conv_bottom": (h - pad_y, 0, pad_y, w, 0, fx), "conv_top": (0, 0, pad_y, w, fy + mh, fx),
                "conv_bottom_left": (h - pad_y, 0, pad_y, pad_x, 0, 0),
                "conv_bottom_right": (h - pad_y, w - pad_x, pad_y, pad_x, 0, fx + mw),
                "conv_top_left": (0, 0, pad_y, pad_x, fy + mh, 0), "conv_top_right": (0, w - pad_x, pad_y, pad_x, fy + mh, fx + mw)}
        banks = {}
        for name, (sy, sx, sh, sw, dy, dx) in regs.items():
            d = L.ConvDesc(N, sh, sw, s.C, 0, node.c_out, k, 0, L.PAD_MODES["zeros"], self.mc_dtype, node.sym_h, 0, 0)
            rh, rw = sh - k + 1, sw - k + 1
            dd = L.ConvDesc(N, rh, rw, node.c_out, 0, s.C, k, k - 1, 0, self.mc_gdtype, 0, 0, 0)
            if L.call("mc_conv_tiles", C.byref(d)) <= 0:
                raise L.MantleHipError(f"unsupported convolution configuration for {node.name}{name}")
            u8 = dict(dtype=torch.uint8, device=device)
            banks[name] = dict(desc=d, ddesc=dd, reg=(sy, sx, sh, sw, dy, dx), rh=rh, rw=rw,
                               S=None if name == "conv" else cb8(s.C, sh, sw), R=cb8(node.c_out, rh, rw),
                               dR=cb8g(node.c_out, rh, rw), dS=cb8g(s.C, sh, sw),
                               bank=torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 0), **u8),
                               dbank=torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 1), **u8),
                               wpart=torch.empty(L.call("mc_wgrad_partial_bytes", C.byref(d)), **u8))
        coutp = ((node.c_out + 7) // 8) * 8
        tiles = min(64, ho)
        e = dict(node=node, banks=banks, tiles=tiles, coutp=coutp, Y=cb8(node.c_out, ho, wo),
                 part=torch.empty((N, tiles, coutp, 2), **f32), need_dgrad=s.requires_grad, dxl=cb8g(s.C, h, w), desc=None)
        o.buf = cb8(node.c_out, ho, wo) if node.post != L.POST_NONE else e["Y"]
        if node.post == L.POST_GN_ACT:
            e["stats"] = torch.empty((N, node.groups, 2), **f32)
            e["gblocks"] = L.call("mc_gn_bwd_blocks", ho, wo)
            e["gpart"] = torch.empty((N, e["gblocks"], coutp, 2), **f32)
            e["m12"] = torch.empty((N, node.groups, 2), **f32)
        if node.pool > 1:
            p = T[node.pooled]
            p.H, p.W = ho // node.pool, wo // node.pool
            p.buf = cb8(node.c_out, p.H, p.W)
        return e

    def _learned_forward(self, e, src, params, need_part, st):
        node, N = e["node"], self.N
        bias = self._param(params, node.name + "learnable_bias")
        o_h, o_w = e["Y"].shape[2], e["Y"].shape[3]
        for name, b in e["banks"].items():
            sy, sx, sh, sw, dy, dx = b["reg"]
            w = self._param(params, node.name + name + ".weight")
            xin = src.buf
            if b["S"] is not None:
                L.call("mc_rect_copy", L.ptr(src.buf), src.H, src.W, sy, sx, L.ptr(b["S"]), sh, sw, 0, 0, sh, sw, N, src.C, 0,
                       self.mc_dtype, st)
                xin = b["S"]
            L.call("mc_pack_weights", C.byref(b["desc"]), L.ptr(w), 0, L.ptr(b["bank"]), st)
            L.call("mc_conv2d", C.byref(b["desc"]), L.ptr(xin), None, L.ptr(b["bank"]), L.ptr(bias), L.ptr(b["R"]), None, None, st)
            L.call("mc_rect_copy", L.ptr(b["R"]), b["rh"], b["rw"], 0, 0, L.ptr(e["Y"]), o_h, o_w, dy, dx, b["rh"], b["rw"], N,
                   node.c_out, 0, self.mc_dtype, st)
        if need_part:
            L.call("mc_gn_partials", L.ptr(e["Y"]), N, node.c_out, o_h, o_w, self.mc_dtype, e["tiles"], L.ptr(e["part"]), st)

    def _learned_backward(self, e, src, dY, params, grads, st):
        node, N = e["node"], self.N
        o_h, o_w = e["Y"].shape[2], e["Y"].shape[3]
        db = grads[node.name + "learnable_bias"]
        for name, b in e["banks"].items():                      # "conv" comes first: its input gradient initialises dxl
            sy, sx, sh, sw, dy, dx = b["reg"]
            L.call("mc_rect_copy", L.ptr(dY), o_h, o_w, dy, dx, L.ptr(b["dR"]), b["rh"], b["rw"], 0, 0, b["rh"], b["rw"], N, node.c_out,
                   0, self.mc_gdtype, st)
            xin = src.buf if b["S"] is None else b["S"]
            L.call("mc_conv2d_wgrad", C.byref(b["desc"]), L.ptr(xin), None, L.ptr(b["dR"]), L.ptr(b["wpart"]), st)
            L.call("mc_conv2d_wgrad_finalize", C.byref(b["desc"]), L.ptr(b["wpart"]), L.ptr(grads[node.name + name + ".weight"]),
                   L.ptr(db), st)
            if e["need_dgrad"]:
                w = self._param(params, node.name + name + ".weight")
                L.call("mc_pack_weights", C.byref(b["desc"]), L.ptr(w), 1, L.ptr(b["dbank"]), st)
                tgt = e["dxl"] if name == "conv" else b["dS"]
                L.call("mc_conv2d", C.byref(b["ddesc"]), L.ptr(b["dR"]), None, L.ptr(b["dbank"]), None, L.ptr(tgt), None, None, st)
                if name != "conv":
                    L.call("mc_rect_copy", L.ptr(b["dS"]), sh, sw, 0, 0, L.ptr(e["dxl"]), src.H, src.W, sy, sx, sh, sw, N, src.C, 1,
                           self.mc_gdtype, st)
        if e["need_dgrad"]:
            src.gsrcs.append(L.GradSrc(L.ptr(e["dxl"]), L.GSRC_PLAIN, 0, 0, 1, src.H, src.W))

    def _table(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device) for a in bicubic_tables(n_in, n_out))
        return self._tables[key]

    # -------------------------------------------------------------- forward
    def output_cb8(self):
        """The last convolution's output as the forward pass left it, for consumers that read the CB8 layout themselves
        (StokesLoss.evaluate(cb8=...)): (buffer [N][C8][H][W + 2 crop][8] f32, per-(sample, channel) spatial means or None,
        crop, (N, C, H, W)); None when the head does not end in an f32 tensor."""
        fo = self.T[self.plan[-1]["node"].out]
        if fo.buf is None or fo.buf.dtype != torch.float32 or fo.C != self.g.c_out:
            return None
        return fo.buf, (self.chan_mean if self.g.subtract_mean else None), self.g.crop_w, (self.N, self.g.c_out, self.out_h, self.out_w)

    def forward(self, x: torch.Tensor, params: Dict[str, torch.Tensor], chan_scale=None, out=None, unpack=True) -> torch.Tensor:
        """x: [N, >= c_in, H, W] f32 device tensor (extra trailing channels are ignored)
        -> [N, c_out, H', W'] f32.  chan_scale: optional [c_in] f32 per-channel input scale.  out: optional preallocated
        result (the fused trainer passes one so that a captured step allocates nothing)."""
        L.require_cuda(x, "network input")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        N, Ci, H, W = x.shape
        if Ci < self.g.c_in:
            raise ValueError(f"expected at least {self.g.c_in} input channels, got {Ci}")
        self.configure(N, H, W, x.device)
        st = L.stream()
        g, T = self.g, self.T
        act = L.ACTS[g.act]
        if self.overlap_wgrad or self.overlap_pack:
            # filter banks are packed on the side stream while the input is converted on the main one
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._pack_all_banks(params, L.stream())
        L.call("mc_pack_nchw", L.ptr(x), N, g.c_in, Ci, H, W, g.in_pad_w, self.mode, L.ptr(chan_scale), self.mc_dtype,
               L.ptr(T[0].buf), st)
        if self.overlap_wgrad or self.overlap_pack:
            main.wait_stream(self.side)
        else:
            self._pack_all_banks(params, st)
        for e in self.plan:
            node = e["node"]
            if node.kind == "pool":
                s, o = T[node.src], T[node.out]
                L.call("mc_avgpool_fwd", L.ptr(s.buf), N, s.C, s.H, s.W, node.f, self.mc_dtype, L.ptr(o.buf), st)
                continue
            if node.kind == "cat":
                srcs = [T[i] for i in node.srcs]
                o = T[node.out]
                ptrs = (C.c_void_p * len(srcs))(*[L.ptr(t.buf) for t in srcs])
                cs = (C.c_int32 * len(srcs))(*[t.C for t in srcs])
                L.call("mc_concat_cb8", ptrs, cs, len(srcs), N, o.H, o.W, self.mc_dtype, L.ptr(o.buf), st)
                continue
            if node.kind == "up":
                s, o = T[node.src], T[node.out]
                (iy, wy, *_), (ix, wx, *_) = e["tabs"]
                if s.fused:      # the producer's GroupNorm + activation are applied while the input window is staged
                    L.call("mc_bicubic_fwd_act", L.ptr(s.raw), L.ptr(s.coef), s.act, N, s.C, s.H, s.W, o.H, o.W, L.ptr(iy),
                           L.ptr(wy), L.ptr(ix), L.ptr(wx), self.mc_dtype, L.ptr(o.buf), st)
                else:
                    L.call("mc_bicubic_fwd", L.ptr(s.buf), N, s.C, s.H, s.W, o.H, o.W, L.ptr(iy), L.ptr(wy), L.ptr(ix),
                           L.ptr(wx), self.mc_dtype, L.ptr(o.buf), st)
                continue
            d = e["desc"]
            if not node.learned:
                w = self._param(params, node.name + "weight")
                b = self._param(params, node.name + "bias")
            srcs = [T[i] for i in node.srcs]
            o = T[node.out]
            final = node.post == L.POST_NONE
            need_part = node.post == L.POST_GN_ACT or (final and g.subtract_mean)
            gamma = self._param(params, node.gn_name + "weight") if node.gn_name else None
            beta = self._param(params, node.gn_name + "bias") if node.gn_name else None
            if node.learned:
                self._learned_forward(e, srcs[0], params, need_part, st)
            else:
                self._probe_begin()
                x0, x1, pro = self._sources(srcs)
                L.call("mc_conv2d_fused", C.byref(d), x0, x1, pro, L.ptr(e["bank"]), L.ptr(b), L.ptr(e["Y"]), None,
                       L.ptr(e["part"]) if need_part else None, None, st)
                self._probe_end(d, "fwd " + node.name)
            small = (node.post == L.POST_GN_ACT and "pc" in e and not o.fused and node.pool in (1, 2)
                     and not node.learned and (self.fuse == 0 or self.fuse_dz_rr) and "dz_blocks" not in e)
            if small:
                # statistics + activation (+ pooling) of a small layer in one launch
                pooled = T[node.pooled].buf if node.pool > 1 else None
                L.call("mc_gn_act_fwd_small", L.ptr(e["Y"]), L.ptr(e["part"]), e["tiles"], N, node.c_out, o.H, o.W, node.groups, 1e-5,
                       L.ptr(gamma), L.ptr(beta), act, node.pool, self.mc_dtype, L.ptr(e["stats"]), L.ptr(o.buf), L.ptr(pooled), st)
                continue
            if node.post == L.POST_GN_ACT:
                # (mean, rstd) per (sample, group) + the (scale, shift, mean, rstd) table consumers normalise on load with
                L.call("mc_gn_finalize_coef", L.ptr(e["part"]), N, e["tiles"], node.c_out, node.groups, o.H * o.W, 1e-5,
                       L.ptr(gamma), L.ptr(beta), L.ptr(e["stats"]), L.ptr(e.get("coef")), st)
            if o.fused:
                if node.pool > 1:      # only the pooled tensor is materialised
                    L.call("mc_gn_act_fwd", L.ptr(e["Y"]), N, node.c_out, o.H, o.W, node.groups, L.ptr(e.get("stats")),
                           L.ptr(gamma), L.ptr(beta), node.post, act, node.pool, self.mc_dtype, None,
                           L.ptr(T[node.pooled].buf), st)
            elif not final:
                pooled = T[node.pooled].buf if node.pool > 1 else None
                L.call("mc_gn_act_fwd", L.ptr(e["Y"]), N, node.c_out, o.H, o.W, node.groups,
                       L.ptr(e.get("stats")), L.ptr(gamma), L.ptr(beta), node.post, act, node.pool, self.mc_dtype,
                       L.ptr(o.buf), L.ptr(pooled), st)
            elif g.subtract_mean:
                L.call("mc_gn_finalize", L.ptr(e["part"]), N, e["tiles"], node.c_out, 1, o.H * o.W, 1e-5, None,
                       L.ptr(self.chan_mean), st)
        fo = T[self.plan[-1]["node"].out]
        if not unpack and self.output_cb8() is not None:      # the caller reads output_cb8() (no NCHW copy of the output)
            return None
        if out is None:
            out = torch.empty((N, g.c_out, self.out_h, self.out_w), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (N, g.c_out, self.out_h, self.out_w) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous f32 tensor of the network's output shape")
        out_dt = L.MC_F32 if fo.buf.dtype == torch.float32 else self.mc_dtype
        L.call("mc_unpack_nchw", L.ptr(fo.buf), N, fo.C, fo.H, fo.W, g.crop_w, L.ptr(self.chan_mean), out_dt,
               L.ptr(out), st)
        return out

    @staticmethod
    def _sources(srcs):
        """(x0, x1, prologue) of a conv's one or two sources: a fused tensor is read as the producer's raw conv output with
        its (scale, shift) table and activation; anything else as it is."""
        ptrs = [L.ptr(t.raw if t.fused else t.buf) for t in srcs] + [None]
        if not any(t.fused for t in srcs):
            return ptrs[0], ptrs[1], None
        t1 = srcs[1] if len(srcs) > 1 else None
        pro = L.ConvPrologue(L.ptr(srcs[0].coef) if srcs[0].fused else None,
                             L.ptr(t1.coef) if (t1 is not None and t1.fused) else None,
                             srcs[0].act if srcs[0].fused else 0, t1.act if (t1 is not None and t1.fused) else 0)
        return ptrs[0], ptrs[1], C.byref(pro)

    @staticmethod
    def _param(params, name):
        p = params[name]
        if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
            raise RuntimeError(f"parameter {name} must be a contiguous f32 device tensor")
        return p

    # -------------------------------------------------------------- backward
    def backward(self, gout: torch.Tensor, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], gsum=None):
        """gout: d(loss)/d(output) [N, c_out, H', W'] f32.  Accumulates parameter gradients into `grads`.
        gsum: (per-block sums [N, blocks, 4] of gout's channel planes, blocks) from the producer of gout
        (StokesLoss.gradient_sums()): the spatial means of gout are then taken from them instead of a pass over gout."""
        L.require_cuda(gout, "output gradient")
        gout = gout.contiguous().float()
        g, T, N = self.g, self.T, self.N
        st = L.stream()
        act = L.ACTS[g.act]
        for t in T.values():
            t.gsrcs = []
        # Two-stream backward: the filter-gradient kernels (x, dY -> dW) of a layer depend only on that layer's dY,
        # so they run on `side` while `main` continues with the input gradient and the next layer's GroupNorm
        # backward.  dY alternates between two buffers; main waits for the side stream before reusing one.
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        wg_done = [None, None]
        gp_jobs = []                                         # GroupNorm parameter gradients of the one-launch layers
        k = 0
        evi = 0
        fo = T[self.plan[-1]["node"].out]
        mean = None
        if g.subtract_mean:
            if gsum is not None and g.c_out <= 4 and tuple(gsum[0].shape) == (N, gsum[1], 4):
                L.call("mc_partial_sums_finalize", L.ptr(gsum[0]), N, gsum[1], g.c_out, 1.0 / (fo.H * fo.W), L.ptr(self.gmean), st)
            else:
                L.call("mc_sum_hw", L.ptr(gout), N * g.c_out, self.out_h * self.out_w, 1.0 / (fo.H * fo.W),
                       L.ptr(self.gmean), st)
            mean = self.gmean
        gdst = (self.plan[-1]["dY"] if self.dy_per_layer else self.dYs[0]) if self.final_plain else self.dOut
        L.call("mc_pack_grad_nchw", L.ptr(gout), N, g.c_out, fo.H, fo.W, g.crop_w, L.ptr(mean), self.mc_gdtype,
               L.ptr(gdst), st)
        if not self.final_plain:
            fo.gsrcs.append(L.GradSrc(L.ptr(self.dOut), L.GSRC_PLAIN, 0, 0, 1, fo.H, fo.W))
        for e in reversed(self.plan):
            node = e["node"]
            if node.kind == "cat":
                # the gradient of the concatenated tensor is one buffer; every operand reads its channel-block slice
                o = T[node.out]
                assert len(o.gsrcs) == 1, "a concatenated tensor feeds exactly one conv"
                q = o.gsrcs[0]
                c8_total = (o.C + 7) // 8
                off = 0
                for i in node.srcs:
                    t = T[i]
                    if t.requires_grad:
                        t.gsrcs.append(L.GradSrc(q.ptr, q.kind, q.pad, q.pad_mode, q.pool, q.hs, q.ws, c8_total, off))
                    off += (t.C + 7) // 8
                continue
            if node.kind == "pool":
                # d(src) += AvgPool adjoint of d(out); d(out) has one source (the conv it feeds) or two (+ the next pooling level)
                s, o = T[node.src], T[node.out]
                assert 1 <= len(o.gsrcs) <= 2
                if not s.requires_grad:
                    continue
                q = o.gsrcs[0]
                if len(o.gsrcs) == 1 and q.kind == L.GSRC_PADFOLD and q.c8_total == 0:
                    s.gsrcs.append(L.GradSrc(q.ptr, L.GSRC_PADFOLD_POOL, q.pad, q.pad_mode, node.f, o.H, o.W))
                else:
                    g1 = C.byref(o.gsrcs[1]) if len(o.gsrcs) > 1 else None
                    L.call("mc_gsrc_sum", C.byref(q), g1, N, o.C, o.H, o.W, self.mc_gdtype, L.ptr(e["dsum"]), st)
                    s.gsrcs.append(L.GradSrc(L.ptr(e["dsum"]), L.GSRC_PLAIN_POOL, 0, 0, node.f, o.H, o.W))
                continue
            if node.kind == "up":
                s, o = T[node.src], T[node.out]
                assert len(o.gsrcs) == 1, "an upsampled tensor feeds exactly one conv"
                (iy, wy, tys, tyj, tyw), (_, _, txs, txj, txw) = e["tabs"]
                if "bws" not in e and e["maxtaps"][1] <= 12 and BICUBIC_WALK:
                    L.call("mc_bicubic_bwd_walk", C.byref(o.gsrcs[0]), N, s.C, s.H, s.W, o.H, o.W, L.ptr(iy), L.ptr(wy), L.ptr(tys),
                           L.ptr(tyj), L.ptr(txs), L.ptr(txj), L.ptr(txw), e["maxtaps"][1], self.mc_gdtype, L.ptr(e["dsrc"]), st)
                elif "bws" in e:
                    L.call("mc_bicubic_bwd_separable", C.byref(o.gsrcs[0]), N, s.C, s.H, s.W, o.H, o.W, L.ptr(tys), L.ptr(tyj),
                           L.ptr(tyw), L.ptr(txs), L.ptr(txj), L.ptr(txw), self.mc_gdtype, L.ptr(e["bws"]), L.ptr(e["dsrc"]), st)
                else:
                    L.call("mc_bicubic_bwd_taps", C.byref(o.gsrcs[0]), N, s.C, s.H, s.W, o.H, o.W, L.ptr(tys), L.ptr(tyj),
                           L.ptr(tyw), L.ptr(txs), L.ptr(txj), L.ptr(txw), e["maxtaps"][0], e["maxtaps"][1], self.mc_gdtype,
                           L.ptr(e["dsrc"]), st)
                s.gsrcs.append(L.GradSrc(L.ptr(e["dsrc"]), L.GSRC_PLAIN, 0, 0, 1, s.H, s.W))
                continue
            d = e["desc"]
            o = T[node.out]
            srcs = [T[i] for i in node.srcs]
            dY = e["dY"] if self.dy_per_layer else self.dYs[k & 1]
            if not self.dy_per_layer and wg_done[k & 1] is not None:
                main.wait_event(wg_done[k & 1])          # the side stream has finished reading this dY buffer
            if node.post != L.POST_NONE and "dz" in e:
                # dz = dA * act'(z) and its partial sums were produced by the consumer's input-gradient launch
                assert not o.gsrcs, f"{node.name}: fused dz and separate gradient sources"
                if node.post == L.POST_GN_ACT:
                    gamma = self._param(params, node.gn_name + "weight")
                    if "dz_pc" in e:
                        # many slots per sample: one block per (group, sample); dgamma / dbeta join the batched launch below
                        L.call("mc_gn_act_bwd_finalize_n", L.ptr(e["dz_part"]), N, e["dz_blocks"], node.c_out, node.groups,
                               o.H * o.W, L.ptr(gamma), L.ptr(e["m12"]), L.ptr(e["dz_pc"]), st)
                        gp_jobs.append((L.ptr(e["dz_pc"]), node.c_out, L.ptr(grads[node.gn_name + "weight"]),
                                        L.ptr(grads[node.gn_name + "bias"])))
                    else:
                        L.call("mc_gn_act_bwd_finalize", L.ptr(e["dz_part"]), N, e["dz_blocks"], node.c_out, node.groups,
                               o.H * o.W, L.ptr(gamma), L.ptr(e["m12"]), L.ptr(grads[node.gn_name + "weight"]),
                               L.ptr(grads[node.gn_name + "bias"]), st)
                L.call("mc_gn_bwd_apply_dz", C.byref(e["dz"]), L.ptr(e["Y"]), N, node.c_out, o.H, o.W, max(node.groups, 1),
                       L.ptr(e["coef"]) if node.post == L.POST_GN_ACT else None, L.ptr(e.get("m12")), self.mc_dtype,
                       L.ptr(dY), st)
                del e["dz"]
            elif node.post != L.POST_NONE:
                gs = list(o.gsrcs)
                if node.pool > 1:
                    p = T[node.pooled]
                    for q in p.gsrcs:
                        assert q.kind in (L.GSRC_PADFOLD, L.GSRC_PLAIN) and q.c8_total == 0, "a pooled tensor feeds exactly one conv"
                        kind = L.GSRC_PADFOLD_POOL if q.kind == L.GSRC_PADFOLD else L.GSRC_PLAIN_POOL   # (PLAIN: learned-padding conv)
                        gs.append(L.GradSrc(q.ptr, kind, q.pad, q.pad_mode, node.pool, p.H, p.W))
                assert 1 <= len(gs) <= 2, f"{node.name}: {len(gs)} gradient sources"
                g0 = C.byref(gs[0])
                g1 = C.byref(gs[1]) if len(gs) > 1 else None
                gamma = self._param(params, node.gn_name + "weight") if node.gn_name else None
                beta = self._param(params, node.gn_name + "bias") if node.gn_name else None
                if node.post == L.POST_GN_ACT and "pc" in e:
                    L.call("mc_gn_act_bwd_small", L.ptr(e["Y"]), N, node.c_out, o.H, o.W, node.groups, L.ptr(e["stats"]),
                           L.ptr(gamma), L.ptr(beta), act, self.mc_dtype, g0, g1, L.ptr(dY), L.ptr(e["pc"]), st)
                    gp_jobs.append((L.ptr(e["pc"]), node.c_out, L.ptr(grads[node.gn_name + "weight"]),
                                    L.ptr(grads[node.gn_name + "bias"])))
                else:
                    if node.post == L.POST_GN_ACT:
                        L.call("mc_gn_act_bwd_reduce", L.ptr(e["Y"]), N, node.c_out, o.H, o.W, node.groups,
                               L.ptr(e["stats"]), L.ptr(gamma), L.ptr(beta), node.post, act, self.mc_dtype, g0, g1,
                               L.ptr(e["gpart"]), st)
                        L.call("mc_gn_act_bwd_finalize", L.ptr(e["gpart"]), N, e["gblocks"], node.c_out, node.groups,
                               o.H * o.W, L.ptr(gamma), L.ptr(e["m12"]), L.ptr(grads[node.gn_name + "weight"]),
                               L.ptr(grads[node.gn_name + "bias"]), st)
                    L.call("mc_gn_act_bwd_apply", L.ptr(e["Y"]), N, node.c_out, o.H, o.W, node.groups,
                           L.ptr(e.get("stats")), L.ptr(e.get("m12")), L.ptr(gamma), L.ptr(beta), node.post, act,
                           self.mc_dtype, g0, g1, L.ptr(dY), st)
            if node.learned:
                self._learned_backward(e, srcs[0], dY, params, grads, st)
                k += 1
                continue
            x0, x1, pro = self._sources(srcs)
            side = self.side if (self.overlap_wgrad == 1 or (self.overlap_wgrad == 2 and o.H * o.W <= 128 * 128)) else main
            wg_done[k & 1] = None
            if side is not main:
                ev = self._events[evi]
                evi += 1
                ev.record(main)                           # dY of this layer is complete
                side.wait_event(ev)
            with torch.cuda.stream(side):
                ss = L.stream()
                L.call("mc_conv2d_wgrad_fused", C.byref(d), x0, x1, pro, L.ptr(dY), L.ptr(e["wpart"]), ss)
                if self.wfin_per_layer and side is not main:
                    # combine this layer's partial slabs right away on the side stream (hidden under the main chain)
                    L.call("mc_conv2d_wgrad_finalize", C.byref(d), L.ptr(e["wpart"]), L.ptr(grads[node.name + "weight"]),
                           L.ptr(grads[node.name + "bias"]), ss)
                    e["_wfin_done"] = True
                else:
                    e["_wfin_done"] = False
                if side is not main:
                    wg_done[k & 1] = self._events[evi]
                    evi += 1
                    wg_done[k & 1].record(side)
            k += 1
            if e["need_dgrad"]:
                dxp = e["dxp"]
                pe = e.get("epi")
                if pe is not None:
                    # the source's GroupNorm-backward reduction rides in this launch: dz and its partial sums instead of dA
                    pn, s0 = pe["node"], srcs[0]
                    coef = L.ptr(pe["coef"]) if pn.post == L.POST_GN_ACT else None
                    epi = L.ConvEpilogue(L.ptr(pe["Y"]), coef, act, node.pad, self.mode, s0.H, s0.W, L.ptr(pe["dz_part"]),
                                         pe["dz_blocks"], int(self.mc_dtype == L.MC_MIX16))
                    self._probe_begin()
                    L.call("mc_conv2d_fused", C.byref(e["ddesc"]), L.ptr(dY), None, None, L.ptr(e["dbank"]), None,
                           L.ptr(dxp[0]), None, None, C.byref(epi), st)
                    self._probe_end(e["ddesc"], "dgrad+dz " + node.name, extra_in=s0.C * s0.H * s0.W)
                    L.call("mc_fold_padded_dz", L.ptr(dxp[0]), N, s0.C, s0.H, s0.W, node.pad, self.mode, self.mc_dtype,
                           L.ptr(pe["Y"]), coef, act, L.ptr(pe["dz_part"]), pe["dz_blocks"], pe["dz_tiles"], st)
                    pe["dz"] = L.GradSrc(L.ptr(dxp[0]), L.GSRC_PADFOLD, node.pad, self.mode, 1, s0.H, s0.W)
                    continue
                self._probe_begin()
                L.call("mc_conv2d", C.byref(e["ddesc"]), L.ptr(dY), None, L.ptr(e["dbank"]), None, L.ptr(dxp[0]),
                       L.ptr(dxp[1]) if len(dxp) > 1 else None, None, st)
                self._probe_end(e["ddesc"], "dgrad " + node.name)
                # adjoint of the padding: fold the halo onto the interior once, consumers read at an offset (the two outputs of
                # a convolution over concatenated sources in one launch)
                live = [(s, buf) for s, buf in zip(srcs, dxp) if s.requires_grad]
                if len(live) == 2 and (live[0][0].H, live[0][0].W) == (live[1][0].H, live[1][0].W):
                    L.call("mc_fold_padded2", L.ptr(live[0][1]), live[0][0].C, L.ptr(live[1][1]), live[1][0].C, N, live[0][0].H,
                           live[0][0].W, node.pad, self.mode, self.mc_gdtype, st)
                else:
                    for s, buf in live:
                        L.call("mc_fold_padded", L.ptr(buf), N, s.C, s.H, s.W, node.pad, self.mode, self.mc_gdtype, st)
                for s, buf in live:
                    s.gsrcs.append(L.GradSrc(L.ptr(buf), L.GSRC_PADFOLD, node.pad, self.mode, 1, s.H, s.W))
        main.wait_stream(self.side)
        if gp_jobs:
            n = len(gp_jobs)
            L.call("mc_gn_param_grads_batched", (C.c_void_p * n)(*[j[0] for j in gp_jobs]), (C.c_int32 * n)(*([N] * n)),
                   (C.c_int32 * n)(*[j[1] for j in gp_jobs]), (C.c_void_p * n)(*[j[2] for j in gp_jobs]),
                   (C.c_void_p * n)(*[j[3] for j in gp_jobs]), n, st)
        # one launch combines every layer's partial slabs, folds mirrored filters and accumulates into the gradients
        todo = [e for e in self.convs if not e.get("_wfin_done")]
        n = len(todo)
        if n:
            descs = (L.ConvDesc * n)(*[e["desc"] for e in todo])
            parts = (C.c_void_p * n)(*[L.ptr(e["wpart"]) for e in todo])
            dws = (C.c_void_p * n)(*[L.ptr(grads[e["node"].name + "weight"]) for e in todo])
            dbs = (C.c_void_p * n)(*[L.ptr(grads[e["node"].name + "bias"]) for e in todo])
            L.call("mc_conv2d_wgrad_finalize_batched", descs, parts, dws, dbs, n, st)

    def _pack_all_banks(self, params, st):
        """Forward and input-gradient banks of every layer in one batched launch per 24 jobs (weights are fixed
        within a step)."""
        jobs = []
        for e in self.convs:
            w = self._param(params, e["node"].name + "weight")
            jobs.append((e["desc"], L.ptr(w), 0, L.ptr(e["bank"])))
            if e["need_dgrad"]:
                jobs.append((e["desc"], L.ptr(w), 1, L.ptr(e["dbank"])))
        n = len(jobs)
        if n == 0:
            return                                           # (a graph of learned-padding layers only)
        descs = (L.ConvDesc * n)(*[j[0] for j in jobs])
        ws = (C.c_void_p * n)(*[j[1] for j in jobs])
        dg = (C.c_int32 * n)(*[j[2] for j in jobs])
        outs = (C.c_void_p * n)(*[j[3] for j in jobs])
        L.call("mc_pack_weights_batched", descs, ws, dg, outs, n, st)

    # -------------------------------------------------------------- measurement hooks (bench.py)
    _probe = None

    def enable_probe(self, passes: int = 3):
        """Record a HIP-event pair (on the launch stream) around every mc_conv2d launch.  The timing events are created and
        recorded once up front: creating them lazily stalled the host for ~80 ms when the runtime grew its signal pool in the
        middle of a pass, and that stall landed inside one event pair (round 2: a 7 ms "launch" of a 50 us kernel)."""
        self._probe = []
        n = 2 * (2 * len(getattr(self, "convs", [])) + 8) * max(passes, 1)
        self._probe_pool = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for ev in self._probe_pool:
            ev.record()
        torch.cuda.synchronize()
        return self._probe

    def disable_probe(self):
        self._probe = None
        self._probe_pool = []

    def _probe_event(self):
        ev = self._probe_pool.pop() if getattr(self, "_probe_pool", None) else torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _probe_begin(self):
        if self._probe is not None:
            self._probe_ev = self._probe_event()

    def _probe_end(self, d, label, extra_in=0):
        """extra_in: further algorithmic input elements per sample (the producer's y an epilogue-fused launch reads once)."""
        if self._probe is not None:
            ev = self._probe_event()
            es = torch.tensor([], dtype=self.t_dtype).element_size()       # (forward and gradient tensors have the same width)
            ho, wo = d.h + 2 * d.pad - d.k + 1, d.w + 2 * d.pad - d.k + 1
            cin = d.c_in0 + d.c_in1
            nbytes = d.n * es * (cin * d.h * d.w + d.c_out * ho * wo + extra_in)   # algorithmic: read inputs once, write output once
            flops = 2.0 * d.n * cin * d.c_out * d.k * d.k * ho * wo
            name = L.load().mc_conv_kernel_name(C.byref(d)).decode()
            shape = f"{label.split()[0]} {cin}->{d.c_out} k{d.k} {d.n}x{d.h}x{d.w}"
            self._probe.append((name, shape, self._probe_ev, ev, nbytes, flops))

    @staticmethod
    def kernel_family(name):
        """k_conv_rr_bf16<5,f32out> -> k_conv_rr_bf16<5>: every instantiation of a kernel for one filter size is ONE family
        (forward f16 / input gradient bf16 / dz epilogue / f32 output differ in a few instructions of the same schedule)."""
        return name.split(",")[0].rstrip(">") + ">"

    @staticmethod
    def probe_summary(probe, hbm_peak_gbs):
        """Dominant conv kernel FAMILY (largest total time): algorithmic bytes per launch / mean launch duration over ALL its
        launches of a step -- the same set of launches bench.py's `traffic` figure is weighted over."""
        probe = [(Engine.kernel_family(name), *rest) for name, *rest in probe]
        groups = {}
        for name, label, e0, e1, nbytes, flops in probe:
            g = groups.setdefault(name, dict(ms=0.0, bytes=0.0, flops=0.0, n=0))
            g["ms"] += e0.elapsed_time(e1)
            g["bytes"] += nbytes
            g["flops"] += flops
            g["n"] += 1
        if not groups:                               # (a network of learned-padding layers only: no plain mc_conv2d node probed)
            return {"bound": "hbm", "achieved": None, "peak": hbm_peak_gbs, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": None, "launches": 0}
        name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
        ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        # the same kernel serves every level of the network: its launches by layer shape (the five that take the most time)
        shapes = {}
        for nm, shape, e0, e1, nbytes, flops in probe:
            if nm == name:
                v = shapes.setdefault(shape, dict(ms=0.0, bytes=0.0, n=0))
                v["ms"] += e0.elapsed_time(e1)
                v["bytes"] += nbytes
                v["n"] += 1
        by_shape = [{"layer": k, "launches": v["n"], "avg_launch_us": 1e3 * v["ms"] / v["n"],
                     "achieved": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / hbm_peak_gbs}
                    for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])[:5]]
        return {"bound": "hbm", "achieved": ach, "peak": hbm_peak_gbs, "unit": "GB/s", "frac": ach / hbm_peak_gbs,
                "traffic": None, "kernel": name, "launches": g["n"], "avg_launch_us": 1e3 * g["ms"] / g["n"],
                "avg_algorithmic_MB_per_launch": g["bytes"] / g["n"] / 1e6,
                "tflops": g["flops"] / (g["ms"] * 1e-3) / 1e12,
                "share_of_conv_time": g["ms"] / sum(v["ms"] for v in groups.values()), "by_shape": by_shape}

    def algorithmic_bytes_per_sample(self, precision=None) -> float:
        """3 s (sum_in + sum_out over the conv layers) + s_io (C_i + 2 C_o) H W."""
        s = 2 if (precision or self.precision) in ("bf16", "mixed") else 4
        tot = 0
        for e in self.plan:
            if e["node"].kind != "conv":
                continue
            if e["node"].learned:
                src, o = self.T[e["node"].srcs[0]], self.T[e["node"].out]
                tot += src.C * src.H * src.W + o.C * o.H * o.W
                continue
            d = e["desc"]
            ho, wo = d.h + 2 * d.pad - d.k + 1, d.w + 2 * d.pad - d.k + 1
            tot += (d.c_in0 + d.c_in1) * d.h * d.w + d.c_out * ho * wo
        # (s_io = s: SURVEY §8d prices the boundary tensors in the storage type as well; round 2 used 4 bytes here, which
        # flattered hbm_roofline_frac_step by 1.7 %)
        return 3.0 * s * tot + float(s) * (self.g.c_in + 2 * self.g.c_out) * self.out_h * self.out_w

    def activation_bytes(self) -> int:
        tot = 0
        for e in self.plan:
            for k in ("Y", "dsrc"):
                if k in e:
                    tot += e[k].numel() * e[k].element_size()
            for b in e.get("dxp", []):
                tot += b.numel() * b.element_size()
        for t in self.T.values():
            if t.buf is not None:
                tot += t.buf.numel() * t.buf.element_size()
        return tot
