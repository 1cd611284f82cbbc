"""world_size-2 rehearsal of the data-parallel host logic on CPU (gloo): parameter broadcast, the single
flat-gradient all-reduce(sum) with the 1/world folded into the optimizer, shard ranges and per-rank seeds.
The kernels themselves need the GPU; what is checked here is the N > 1 plumbing around them."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from pbml_mantle_convection_amd import multigpu as G
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    G.ddp_setup(rank, world, port, backend="gloo")
    torch.manual_seed(100 + rank)                       # deliberately different initial parameters
    flat_param = torch.randn(1003)
    G.broadcast_flat(flat_param)
    # per-rank shard of the global batch: different seeds -> different samples
    lo, hi = G.shard_range(8, world, rank)
    x = synthetic_batch(hi - lo, 16, 24, 1234 + rank)[0]
    grad = torch.full((1003,), float(rank + 1)) + x.mean()          # stand-in for this rank's flat gradient
    local = grad.clone()
    w = G.allreduce_flat(grad)
    # the Adam kernel applies grad_scale = 1/world: emulate that step with plain SGD for the check
    new_param = flat_param - 0.1 * grad * (1.0 / w)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = flat_param - 0.1 * torch.stack(gathered).mean(0)
    out[rank] = dict(param0=flat_param[:5].clone(), ok=bool(torch.allclose(new_param, expect, atol=1e-6)), world=w,
                     shard=(lo, hi), xmean=float(x.mean()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_allreduce_and_broadcast():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["world"] == r1["world"] == 2
    assert torch.equal(r0["param0"], r1["param0"])          # broadcast made the replicas identical
    assert r0["ok"] and r1["ok"]                            # sum all-reduce x 1/world == mean of the rank gradients
    assert r0["shard"] == (0, 4) and r1["shard"] == (4, 8)
    assert r0["xmean"] != r1["xmean"]                       # independent shards


def test_single_process_is_noop():
    from pbml_mantle_convection_amd import multigpu as G
    g = torch.ones(7)
    assert G.allreduce_flat(g) == 1 and torch.equal(g, torch.ones(7))
    G.broadcast_flat(g)
