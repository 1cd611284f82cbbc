#!/usr/bin/env python3
"""Experiment: two independent half-batch training chains replayed concurrently on two streams vs one full-batch chain.
(Answers whether the batch-independent 3.5 ms of a step can be hidden behind another chain's throughput-bound kernels.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbml_mantle_convection_amd.datasetio import synthetic_batch
from pbml_mantle_convection_amd.multigpu import Trainer
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet

dev = torch.device("cuda:0")
CFG = bench.CFG


def make(B, seed):
    torch.manual_seed(0)
    m = Unet(CFG["levels"], CFG["c_i"], CFG["c_h"], CFG["c_o"], dev, CFG["act"], CFG["r_p"], CFG["loss_type"],
             use_symm=CFG["use_symm"], repeats=CFG["repeats"], f=CFG["f"], p_pred=CFG["p_pred"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 9], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type=CFG["loss_type"],
                 lambda_mom=0.1, precision="bf16", use_graph=True)
    g, u, sc, pa, yc = [t.to(dev) for t in synthetic_batch(B, 506, 506, seed, p_pred=True, device="cpu")]
    return tr, (g, u, yc, pa, sc)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full, bf = make(32, 1)
print("one chain, B=32: %.3f ms" % timeit(lambda: full.train_step(*bf)))
del full, bf
torch.cuda.empty_cache()
for halves in (2, 4):
    trs = [make(32 // halves, 10 + i) for i in range(halves)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(halves)]
    for tr, b in trs:
        tr.train_step(*b)            # capture
    torch.cuda.synchronize()

    def seq():
        for tr, b in trs:
            tr.train_step(*b)

    def conc():
        cur = torch.cuda.current_stream()
        for (tr, b), s in zip(trs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                tr.train_step(*b)
        for s in streams:
            cur.wait_stream(s)

    print("%d chains of B=%d sequential: %.3f ms   concurrent: %.3f ms" % (halves, 32 // halves, timeit(seq), timeit(conc)))
    del trs
    torch.cuda.empty_cache()

# Result on MI355X: one chain B=32 13.3 ms; 2 x B=16 sequential 16.0 / concurrent 15.9 ms; 4 x B=8 22.4 / 22.3 ms: graph
# launches on different streams do not overlap, so splitting the batch into concurrent chains does not hide the fixed cost.
