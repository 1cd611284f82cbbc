#!/usr/bin/env python3
"""Micro-benchmark of the GroupNorm+activation kernels through the C ABI (bf16).
usage: tools/bench_gn.py --n 32 --c 64 --hw 63 64 [--iters 20]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbml_mantle_convection_amd import _lib as L
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=32); p.add_argument("--c", type=int, default=64)
p.add_argument("--hw", type=int, nargs=2, default=[63, 64]); p.add_argument("--iters", type=int, default=20)
a = p.parse_args()
L.load(); dev = "cuda:0"; st = L.stream()
N, Cc, (H, W) = a.n, a.c, a.hw
G = Cc // 4
cb = lambda: torch.randn((N, Cc // 8, H, W, 8), device=dev).to(torch.bfloat16)
y, out, dA, dY = cb(), cb(), cb(), cb()
stats = torch.stack([torch.zeros(N, G, device=dev), torch.ones(N, G, device=dev)], -1).contiguous()
gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
nb = L.call("mc_gn_bwd_blocks", H, W)
gpart = torch.zeros((N, nb, Cc, 2), device=dev); m12 = torch.zeros((N, G, 2), device=dev)
dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
g0 = L.GradSrc(L.ptr(dA), L.GSRC_PLAIN, 0, 0, 1, H, W)
ACT = L.ACTS["gelu"]
ops = {
 "fwd": lambda: L.call("mc_gn_act_fwd", L.ptr(y), N, Cc, H, W, G, L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.POST_GN_ACT, ACT, 1, L.MC_BF16, L.ptr(out), None, st),
 "reduce": lambda: L.call("mc_gn_act_bwd_reduce", L.ptr(y), N, Cc, H, W, G, L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.POST_GN_ACT, ACT, L.MC_BF16, C.byref(g0), None, L.ptr(gpart), st),
 "finalize": lambda: L.call("mc_gn_act_bwd_finalize", L.ptr(gpart), N, nb, Cc, G, H * W, L.ptr(gamma), L.ptr(m12), L.ptr(dg), L.ptr(db), st),
 "apply": lambda: L.call("mc_gn_act_bwd_apply", L.ptr(y), N, Cc, H, W, G, L.ptr(stats), L.ptr(m12), L.ptr(gamma), L.ptr(beta), L.POST_GN_ACT, ACT, L.MC_BF16, C.byref(g0), None, L.ptr(dY), st),
}
E = N * Cc * H * W * 2
mult = {"fwd": 2, "reduce": 2, "finalize": 0, "apply": 3}
for name, fn in ops.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{name:9s} C{Cc} {H}x{W}: {us:8.1f} us  {mult[name] * E / us / 1e3:8.0f} GB/s", flush=True)
