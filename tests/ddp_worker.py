"""Rank body of the two-process data-parallel GPU test (tests/test_hip_ddp.py).  Imported by name in children that the
multiprocessing fork server forks; the fork server itself is started by conftest.py before anything touches the GPU, so no
process that has initialised HIP ever forks or execs."""
import os

import numpy as np


def make_trainer(prec, seed, dev="cuda:0"):
    import torch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    torch.manual_seed(seed)
    m = Unet(3, 10, 16, 4, torch.device(dev), "gelu", "reflect", "mass", use_symm=True, repeats=2, f=5, p_pred=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1000], gamma=0.5)
    return Trainer(m, None, None, None, None, None, opt, sch, 0, 1, "/tmp/", p_pred=True, network="unet", loss_type="mass",
                   lambda_mom=1e-6, precision=prec, use_graph=True)


def global_batch(B=4, H=80, W=150, seed=5):
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    return synthetic_batch(B, H, W, seed, p_pred=True, device="cpu")


def steps(tr, batch, n):
    import torch
    g, u, sc, pa, yc = [t.to("cuda:0") for t in batch]
    first = None
    for i in range(n):
        tr.train_step(g, u, yc, pa, sc)
        if i == 0:
            torch.cuda.synchronize()
            first = tr.flat.grad.detach().cpu().numpy().copy()
    torch.cuda.synchronize()
    return dict(grad1=first, param=tr.flat.param.detach().cpu().numpy(), m=tr.exp_avg.cpu().numpy(),
                v=tr.exp_avg_sq.cpu().numpy())


def run(rank, world, port, prec, nsteps, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from pbml_mantle_convection_amd import multigpu as G
    G.ddp_setup(rank, world, port, backend="gloo")        # both ranks share the one GPU of the box: RCCL refuses that
    torch.cuda.set_device(0)
    tr = make_trainer(prec, 100 + rank)                   # rank 1 starts from other weights: the broadcast must fix it
    assert tr.world == world
    batch = global_batch()
    lo, hi = G.shard_range(batch[0].shape[0], world, rank)
    res = steps(tr, [t[lo:hi] for t in batch[:4]] + [batch[4]], nsteps)        # yc [H, W] is the mesh, shared by all samples
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def run_rccl_single(port, outdir):
    """One rank, backend 'nccl' (= RCCL): the process group and its watchdog thread are alive while the training step is
    captured and replayed, and the flat buffers go through RCCL's broadcast / all-reduce entry points once."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from pbml_mantle_convection_amd import multigpu as G
    G.ddp_setup(0, 1, port, backend="nccl")
    assert dist.get_backend() == "nccl"
    tr = make_trainer("bf16", 100)
    res = steps(tr, global_batch(), 3)
    t = tr.flat.grad.clone()
    dist.all_reduce(t)                                   # RCCL on the flat gradient buffer (world 1: identity)
    dist.broadcast(tr.flat.param, src=0)
    torch.cuda.synchronize()
    assert torch.equal(t, tr.flat.grad)
    np.savez(os.path.join(outdir, "rccl1.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def run_bench(rank, world, port, outdir):
    """`bench.py --gpus 2 --backend gloo` as the driver launches it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), on a small grid; rank 0's stdout (the ONE JSON line) goes to a file."""
    import contextlib
    import io
    import sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.argv = ["bench.py", "--gpus", str(world), "--backend", "gloo", "--steps", "3", "--warmup", "2", "--batch", "2",
                "--size", "64", "150", "--no-cpu-baseline"]
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(outdir, f"bench_rank{rank}.out"), "w") as f:
        f.write(buf.getvalue())
