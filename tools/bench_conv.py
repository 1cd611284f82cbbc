#!/usr/bin/env python3
"""Micro-benchmark of single conv kernels through the C ABI (forward, input gradient, filter gradient).
usage: tools/bench_conv.py [--n 32] [--hw 506 512] [--cin 16] [--cout 16] [--k 5] [--dtype bf16] [--iters 10] [--which fwd,dgrad,wgrad]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbml_mantle_convection_amd import _lib as L

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=32); p.add_argument("--hw", type=int, nargs=2, default=[506, 512])
p.add_argument("--cin", type=int, default=16); p.add_argument("--cin1", type=int, default=0)
p.add_argument("--cout", type=int, default=16)
p.add_argument("--k", type=int, default=5); p.add_argument("--dtype", default="bf16"); p.add_argument("--iters", type=int, default=10)
p.add_argument("--which", default="fwd,dgrad,wgrad"); p.add_argument("--sym", type=int, default=4)
a = p.parse_args()
L.load()
dev = "cuda:0"
# "mixed": MC_MIX16 layer (f16 forward tensors, bf16 gradient tensors and input-gradient launch)
mc = {"bf16": L.MC_BF16, "mixed": L.MC_MIX16}.get(a.dtype, L.MC_F32)
mcg = L.MC_BF16 if a.dtype == "mixed" else mc
tdt = {"bf16": torch.bfloat16, "mixed": torch.float16}.get(a.dtype, torch.float32)
gdt = torch.bfloat16 if a.dtype == "mixed" else tdt
es = 4 if a.dtype == "fp32" else 2
N, (H, W) = a.n, a.hw
pad = a.k // 2
cin = a.cin + a.cin1
d = L.ConvDesc(N, H, W, a.cin, a.cin1, a.cout, a.k, pad, 2, mc, a.sym, 0, 0)
dd = L.ConvDesc(N, H, W, a.cout, 0, cin, a.k, a.k - 1, 0, mcg, 0, a.cin if a.cin1 else 0, 0)
st = L.stream()
cb = lambda c, h, w, t=tdt: torch.randn((N, (c + 7) // 8, h, w, 8), device=dev).to(t)
x0 = cb(a.cin, H, W); x1 = cb(a.cin1, H, W) if a.cin1 else None
y = cb(a.cout, H, W); dy = cb(a.cout, H, W, gdt)
dx0 = cb(a.cin, H + 2 * pad, W + 2 * pad, gdt); dx1 = cb(a.cin1, H + 2 * pad, W + 2 * pad, gdt) if a.cin1 else None
U = a.cout - a.sym // 2
w = torch.randn((U, cin, a.k, a.k), device=dev) / (cin * a.k * a.k) ** 0.5
b = torch.zeros(a.cout, device=dev)
bank = torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 0), dtype=torch.uint8, device=dev)
dbank = torch.empty(L.call("mc_packed_weight_bytes", C.byref(d), 1), dtype=torch.uint8, device=dev)
L.call("mc_pack_weights", C.byref(d), L.ptr(w), 0, L.ptr(bank), st)
L.call("mc_pack_weights", C.byref(d), L.ptr(w), 1, L.ptr(dbank), st)
tiles = L.call("mc_conv_tiles", C.byref(d))
part = torch.empty((N, tiles, ((a.cout + 7) // 8) * 8, 2), device=dev)
wpart = torch.empty(L.call("mc_wgrad_partial_bytes", C.byref(d)), dtype=torch.uint8, device=dev)
dw = torch.zeros_like(w); db = torch.zeros_like(b)
flops = 2.0 * N * cin * a.cout * a.k * a.k * H * W
# fused forms: the source is a raw conv output normalised on load; the input gradient carries the GroupNorm-backward epilogue
cp8 = lambda c: ((c + 7) // 8) * 8
coef0 = torch.randn((N, cp8(a.cin), 4), device=dev) * 0.1 + torch.tensor([1.0, 0.0, 0.0, 1.0], device=dev)
pro = L.ConvPrologue(L.ptr(coef0), None, L.ACTS["gelu"], 0)
dtiles = L.call("mc_conv_tiles", C.byref(dd))
fb = L.call("mc_fold_blocks", H, W, pad, 2)
epart = torch.empty((N, dtiles + fb, cp8(cin), 2), device=dev)
ecoef = torch.randn((N, cp8(cin), 4), device=dev) * 0.1 + torch.tensor([1.0, 0.0, 0.0, 1.0], device=dev)
epi = L.ConvEpilogue(L.ptr(x0), L.ptr(ecoef), L.ACTS["gelu"], pad, 2, H, W, L.ptr(epart), dtiles + fb, int(a.dtype == "mixed"))
ops = {
    "fwdn": (lambda: L.call("mc_conv2d_fused", C.byref(d), L.ptr(x0), L.ptr(x1), C.byref(pro), L.ptr(bank), L.ptr(b), L.ptr(y), None, L.ptr(part), None, st),
             N * es * (cin + a.cout) * H * W),
    "dgradz": (lambda: L.call("mc_conv2d_fused", C.byref(dd), L.ptr(dy), None, None, L.ptr(dbank), None, L.ptr(dx0), None, None, C.byref(epi), st),
               N * es * (2 * cin + a.cout) * H * W),
    "foldz": (lambda: L.call("mc_fold_padded_dz", L.ptr(dx0), N, cin, H, W, pad, 2, mc, L.ptr(x0), L.ptr(ecoef), L.ACTS["gelu"], L.ptr(epart), dtiles + fb, dtiles, st), 0),
    "wgradn": (lambda: L.call("mc_conv2d_wgrad_fused", C.byref(d), L.ptr(x0), L.ptr(x1), C.byref(pro), L.ptr(dy), L.ptr(wpart), st),
               N * es * (cin + a.cout) * H * W),
    "fwd": (lambda: L.call("mc_conv2d", C.byref(d), L.ptr(x0), L.ptr(x1), L.ptr(bank), L.ptr(b), L.ptr(y), None, L.ptr(part), st),
            N * es * (cin + a.cout) * H * W),
    "dgrad": (lambda: L.call("mc_conv2d", C.byref(dd), L.ptr(dy), None, L.ptr(dbank), None, L.ptr(dx0), L.ptr(dx1), None, st),
              N * es * (cin + a.cout) * H * W),
    "wgrad": (lambda: L.call("mc_conv2d_wgrad", C.byref(d), L.ptr(x0), L.ptr(x1), L.ptr(dy), L.ptr(wpart), st),
              N * es * (cin + a.cout) * H * W),
    "wfin": (lambda: L.call("mc_conv2d_wgrad_finalize", C.byref(d), L.ptr(wpart), L.ptr(dw), L.ptr(db), st), 0),
}
for name in a.which.split(","):
    fn, nbytes = ops[name]
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{name:6s} {a.cin}+{a.cin1}->{a.cout} k{a.k} {N}x{H}x{W} {a.dtype}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  "
          f"{nbytes / us / 1e3:7.1f} GB/s algorithmic  [{L.call('mc_conv_kernel_name', C.byref(d if name != 'dgrad' else dd)).decode()}]")
