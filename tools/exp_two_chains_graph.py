#!/usr/bin/env python3
"""Experiment: two half-batch training chains captured as parallel branches of ONE HIP graph vs one full-batch chain.
(tools/exp_two_chains.py showed that separate graph launches do not overlap; branches of one graph do.)"""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pbml_mantle_convection_amd.datasetio import synthetic_batch
from pbml_mantle_convection_amd.multigpu import Trainer
from pbml_mantle_convection_amd.pytorch_networks_convae import Unet

if os.environ.get("EXP_FORCE") != "1":
    sys.exit("exp_two_chains_graph.py: known to die with SIGSEGV inside torch.cuda.CUDAGraph.capture_end() (hipStreamEndCapture /\n"
             "graph instantiation) as soon as TWO training chains are captured as parallel branches of one graph -- reproduced on\n"
             "ROCm 7.2 / torch 2.10 after every allocation and event creation was moved out of the captured region, with and without\n"
             "the engines' side streams and the optimizer.  One chain captured the same way replays fine.  Set EXP_FORCE=1 to run it\n"
             "anyway (host crash only, no GPU hang).  See DESIGN.md section 10.")

dev = torch.device("cuda:0")
CFG = bench.CFG


def make(B, seed, use_graph):
    torch.manual_seed(0)
    m = Unet(CFG["levels"], CFG["c_i"], CFG["c_h"], CFG["c_o"], dev, CFG["act"], CFG["r_p"], CFG["loss_type"],
             use_symm=CFG["use_symm"], repeats=CFG["repeats"], f=CFG["f"], p_pred=CFG["p_pred"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 9], gamma=0.5)
    tr = Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type=CFG["loss_type"],
                 lambda_mom=0.1, precision="bf16", use_graph=use_graph)
    g, u, sc, pa, yc = [t.to(dev) for t in synthetic_batch(B, 506, 506, seed, p_pred=True, device="cpu")]
    return tr, (g, u, yc.float(), pa.float(), sc.float())


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full, bf = make(32, 1, True)
print("one chain, B=32: %.3f ms" % timeit(lambda: full.train_step(*bf)), flush=True)
del full, bf
torch.cuda.empty_cache()

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
OPT = os.environ.get("EXP_OPT", "1") == "1"
trs = [make(32 // nch, 10 + i, False) for i in range(nch)]
for tr, b in trs:
    tr._sync_lr()
warm = torch.cuda.Stream(device=dev)
warm.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(warm):
    for tr, b in trs:
        tr._fwd_bwd(*b, train=True)
        tr._optim_step()
torch.cuda.current_stream().wait_stream(warm)
torch.cuda.synchronize()
print("warm-up done", flush=True)
branch = [torch.cuda.Stream(device=dev) for _ in range(nch - 1)]
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    cur = torch.cuda.current_stream()
    for s in branch:
        s.wait_stream(cur)
    for i, (tr, b) in enumerate(trs):
        if i == 0:
            tr._fwd_bwd(*b, train=True)
            if OPT:
                tr._optim_step()
        else:
            with torch.cuda.stream(branch[i - 1]):
                tr._fwd_bwd(*b, train=True)
                if OPT:
                    tr._optim_step()
    for s in branch:
        cur.wait_stream(s)
print("captured", flush=True)
print("%d chains of B=%d as branches of one graph: %.3f ms" % (nch, 32 // nch, timeit(graph.replay)), flush=True)

# Result on MI355X (ROCm 7.2, torch 2.10), rounds 1 and 2: one chain captured this way replays at 12.1 ms; with two chains the
# process dies with SIGSEGV inside CUDAGraph.capture_end (faulthandler: torch/cuda/graphs.py:130), also with
# MANTLE_OVERLAP_WGRAD=0, without the optimizer, and (round 2) with no allocation or event creation inside the capture.
