"""Data-parallel training step on the GPU with world_size 2: two rank processes (forked from a fork server that was started
before the GPU was initialised) share the box's one MI355X and exchange the flat gradient through gloo -- RCCL refuses two
ranks on one device, so the RCCL transport itself stays unverified here; everything around it (parameter broadcast, shard
of the global batch, captured fwd/bwd graph -> all-reduce(sum) -> captured Adam with 1/world) is the product path."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_two_ranks_match_each_other_and_the_single_process_step(prec, tmp_path):
    import ddp_worker as W
    world, nsteps = 2, 3
    ctx = mp.get_context("forkserver")
    port = _free_port()
    procs = [ctx.Process(target=W.run, args=(r, world, port, prec, nsteps, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0, p.exitcode
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in ("grad1", "param", "m", "v"):                 # replicas stay bit-identical
        assert np.array_equal(r0[k], r1[k]), k

    # the same three steps in one process on the whole batch, from rank 0's initial weights
    one = W.steps(W.make_trainer(prec, 100), W.global_batch(), nsteps)
    g2 = r0["grad1"].astype(np.float64) / world             # all-reduce(sum) of per-rank means -> global mean
    g1 = one["grad1"].astype(np.float64)
    rel = np.linalg.norm(g2 - g1) / np.linalg.norm(g1)
    # per-sample arithmetic is identical; only the order of the f32 filter-gradient sums differs
    assert rel <= (2e-5 if prec == "fp32" else 2e-4), rel
    lr = 1e-3
    d = np.abs(r0["param"].astype(np.float64) - one["param"])
    # Adam's first steps move every weight by ~lr whatever |g| is, so an entry whose gradient is ~0 may differ by up to lr
    assert (d > 0.1 * lr).mean() <= (2e-3 if prec == "fp32" else 1e-2), (d > 0.1 * lr).mean()
    assert d.max() <= 2.5 * nsteps * lr


def test_captured_step_with_a_live_rccl_process_group(tmp_path):
    """backend='nccl' with ONE rank (all this box has): RCCL initialises, its watchdog thread runs while the step is captured
    (capture_error_mode='thread_local') and replayed, all-reduce / broadcast of the flat buffers go through RCCL; the result
    equals the run without a process group.  (Two ranks over RCCL need two GPUs: unverified here)"""
    import ddp_worker as W
    ctx = mp.get_context("forkserver")
    p = ctx.Process(target=W.run_rccl_single, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(300)
    if p.is_alive():
        p.terminate()
    assert p.exitcode == 0, p.exitcode
    r = np.load(tmp_path / "rccl1.npz")
    one = W.steps(W.make_trainer("bf16", 100), W.global_batch(), 3)
    for k in ("grad1", "param", "m", "v"):
        assert np.array_equal(r[k], one[k]), k


def test_bench_two_ranks_prints_one_json_line(tmp_path):
    """`bench.py --gpus 2` end to end (the driver's multi-GPU run, rehearsed over gloo on the box's one GPU): both ranks run
    `bench.main()` with the torchrun environment, rank 0 prints exactly one JSON line with n_gpus = 2, the whole-job value
    and the roofline object; rank 1 prints nothing."""
    import json
    import ddp_worker as W
    world = 2
    ctx = mp.get_context("forkserver")
    port = _free_port()
    procs = [ctx.Process(target=W.run_bench, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0, p.exitcode
    lines = [l for l in open(tmp_path / "bench_rank0.out").read().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["unit"] == "samples/s"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and "cpu_baseline" not in d
    assert open(tmp_path / "bench_rank1.out").read().strip() == ""
