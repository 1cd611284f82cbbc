#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the Stokes-surrogate training step (symmetric U-Net
forward + Stokes PDE-residual loss + backward + gradient all-reduce + Adam) on synthetic 2-D
506x506 mantle fields, one process per MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, ms_per_step,
`roofline` for the dominant kernel (algorithmic bytes / live HIP-event duration) and `cpu_baseline`
(the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
CFG = dict(levels=5, c_i=10, c_h=16, c_o=4, act="gelu", r_p="reflect", loss_type="mass", use_symm=True, repeats=3, f=5,
           p_pred=True)        # CFG-3 of SURVEY.md §8 (c_i = 10: get_loss feeds ten channels, reference :234-248)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None)
    p.add_argument("--warmup", type=int, default=None)
    p.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    p.add_argument("--size", type=int, nargs=2, default=[506, 506], metavar=("H", "W"))
    p.add_argument("--precision", type=str, default=os.environ.get("MANTLE_BENCH_PRECISION", "bf16"),
                   choices=["bf16", "mixed", "fp32"],
                   help="bf16 with a momentum term runs as 'mixed' (f16 forward tensors, bf16 gradient tensors, see engine.py)")
    p.add_argument("--lambda-mom", type=float, default=1e-6, help="weight of the Stokes momentum residual (CFG-3)")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-batch", type=int, default=2)
    p.add_argument("--cpu-steps", type=int, default=3)
    p.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    p.add_argument("--r-p", type=str, default=None, choices=["zeros", "replicate", "reflect", "learned"],
                   help="padding of the conv layers (default: reflect for cfg3, zeros for newfluidnet; 'learned' = SURVEY 8f N4)")
    p.add_argument("--workload", type=str, default="cfg3", choices=["cfg3", "newfluidnet"],
                   help="cfg3 = the headline benchmark; newfluidnet = SURVEY 8(f) N1, the deployed multi-resolution trunk "
                        "(-net newfluidnet -l 5 -f 16 -r 6 -k 5) on its 128 x 506 grid")
    return p.parse_args()


def host_cores():
    """CPU threads this process may really use: the cgroup CPU quota if there is one (the GPU box grants a
    16-CPU share of a 256-thread host), else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MANTLE_CPU_THREADS", "16"))))


PMC_FILE = "round3_pmc_traffic.json"


def pmc_traffic(kernel, B, H, W, precision):
    """(HBM bytes per launch of the dominant kernel, commit they were measured at) from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 + WRITE_SIZE, profiles/round3_pmc_traffic.json, collected on this same workload by tools/profile_round.sh);
    (None, None) for any other workload.  The counters cannot be read inside this process: the figure is as old as its commit."""
    if (B, H, W) != (32, 506, 506) or precision not in ("bf16", "mixed"):
        return None, None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            d = json.load(f)
        # the probe names the kernel family (k_conv_rr_bf16<5>: every instantiation for that filter size); the counters are
        # per instantiation: launch-weighted mean over the same family
        fam = kernel.replace(" ", "").rstrip(">")
        ks = [v for n, v in d["kernels"].items() if n == fam + ">" or n.startswith(fam + ",")]
        if not ks:
            return None, None
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks), d.get("commit")
    except Exception:
        return None, None


def cpu_baseline(args, H, W):
    """The CPU oracle (oracle/ref_cpu.py, pinned against the reference by golden vectors) timed on the
    host cores: same network, same loss (incl. momentum term), fp32, bounded sample."""
    from oracle import ref_cpu as O
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet
    cores = host_cores()
    torch.set_num_threads(cores)
    B = args.cpu_batch
    torch.manual_seed(0)
    m = Unet(CFG["levels"], CFG["c_i"], CFG["c_h"], CFG["c_o"], torch.device("cpu"), CFG["act"], CFG["r_p"],
             CFG["loss_type"], use_symm=CFG["use_symm"], repeats=CFG["repeats"], f=CFG["f"], p_pred=CFG["p_pred"])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    st = O.CpuUnetStep(sd, dict(levels=CFG["levels"], repeats=CFG["repeats"], act=CFG["act"], r_p=CFG["r_p"],
                                loss_type=CFG["loss_type"], use_symm=CFG["use_symm"], p_pred=CFG["p_pred"]))
    gVTp, uvp, scaler, paras, yc = synthetic_batch(B, H, W, 99, p_pred=True)
    mom = dict(lambda_mom=args.lambda_mom, yc=yc, paras=paras, scaler=scaler) if args.lambda_mom else None
    st.step(gVTp, uvp, momentum=mom)                      # warm-up
    t0 = time.time()
    for _ in range(args.cpu_steps):
        st.step(gVTp, uvp, momentum=mom)
    dt = time.time() - t0
    return dict(value=B * args.cpu_steps / dt, unit="samples/s", cores=cores, kind="port",
                sample=f"{args.cpu_steps} steps of batch {B} at {H}x{W}, fp32, torch {torch.__version__} CPU ATen, "
                       f"1 warm-up step")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)              # (rehearsals may oversubscribe one GPU with gloo)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from pbml_mantle_convection_amd import _lib as L
    from pbml_mantle_convection_amd.datasetio import synthetic_batch
    from pbml_mantle_convection_amd.multigpu import Trainer
    from pbml_mantle_convection_amd.pytorch_networks_convae import Unet

    H, W = args.size
    B = args.batch
    n1 = args.workload == "newfluidnet"
    if n1:
        H, W = 128, 506
        args.no_cpu_baseline = True                      # (the CPU baseline leg is the headline workload's)
    steps = args.steps if args.steps is not None else (20 if args.precision != "fp32" else 3)
    warmup = args.warmup if args.warmup is not None else (5 if args.precision != "fp32" else 1)

    torch.manual_seed(0)                                   # identical initial weights on every rank
    if n1:
        from pbml_mantle_convection_amd.pytorch_networks_convae import NewFluidNet
        model = NewFluidNet(5, 7, 16, 4 if args.r_p == "learned" else 3, dev, "gelu", args.r_p or "zeros", "mass", use_symm=True,
                            repeats=6, f=5, p_pred=True)
    else:
        model = Unet(CFG["levels"], CFG["c_i"], CFG["c_h"], CFG["c_o"], dev, CFG["act"], args.r_p or CFG["r_p"], CFG["loss_type"],
                     use_symm=CFG["use_symm"], repeats=CFG["repeats"], f=CFG["f"], p_pred=CFG["p_pred"])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10 ** 9], gamma=0.5)
    tr = Trainer(model, None, None, None, None, None, opt, sch, local_rank % ndev, 1, "/tmp/", p_pred=True,
                 network="newfluidnet" if n1 else "unet", loss_scale=n1, loss_derivative=False,
                 loss_type=CFG["loss_type"], lambda_mom=0.0 if n1 else args.lambda_mom, precision=args.precision,
                 use_graph=not args.no_graph)
    # synthetic fields, resident in HBM before the timed region (seed differs per rank: independent shards)
    gVTp, uvp, scaler, paras, yc = synthetic_batch(B, H, W, 1234 + rank, p_pred=True, device="cpu")
    if n1:
        gVTp, uvp = gVTp[:, :7].contiguous(), uvp[:, :3].contiguous()      # NewADDataset items: x [7,H,W], y (u, v, p)
    gVTp, uvp, scaler, paras, yc = (t.to(dev) for t in (gVTp, uvp, scaler, paras, yc))

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(warmup):
        out8 = tr.train_step(gVTp, uvp, yc, paras, scaler)
        if i == 0 and not args.no_graph:
            # the synthetic batch lives in the captured step's own input buffers from here on (resident in HBM before the
            # timed region; a loader would write each new batch into these buffers): no staging copy inside the step
            b = tr.input_buffers()
            gVTp, uvp, yc, paras, scaler = b["gVTp"], b["uvp"], b["yc"], b["paras"], b["scaler"]
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out8 = tr.train_step(gVTp, uvp, yc, paras, scaler)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out8[0].item())
    if not (loss == loss) or abs(loss) > 1e6:
        raise SystemExit(f"non-finite / diverged loss {loss}")

    # ---- roofline of the dominant kernel: an instrumented pass of the same steps, HIP events around every
    #      launch of the conv forward/input-gradient kernel family on the launch stream
    eng = model.engine()
    nprobe = min(steps, 3)
    probe = eng.enable_probe(nprobe)
    for _ in range(nprobe):
        tr._fwd_bwd(gVTp, uvp, yc, paras, scaler, train=True)
    torch.cuda.synchronize(dev)
    roof = eng.probe_summary(probe, HBM_PEAK_GBS)
    eng.disable_probe()
    roof["traffic"], roof["traffic_measured_at_commit"] = pmc_traffic(roof["kernel"], B, H, W, args.precision) if roof["kernel"] \
        else (None, None)

    line = None
    if rank == 0:
        sps = world * B * steps / elapsed
        line = {
            "metric": "training samples/sec (2-D 128x506 Stokes fields, NewFluidNet)" if n1 else
                      f"training samples/sec (2-D {H}x{W} Stokes fields)", "value": sps, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "mixed": "f16+bf16"}.get(model.precision, "bf16"), "data": "synthetic",
            "config": {"workload": (f"N1: NewFluidNet (levels 5, c_h 16, k 5, repeats 6, zeros, symmetric) + scaled L1 data loss "
                                    f"+ divergence, {H}x{W}, per-GPU batch {B}, Adam, "
                                    f"{'HIP-graph replay' if not args.no_graph else 'eager launches'}"
                                    + (f", r_p={args.r_p}" if args.r_p else "")) if n1 else
                                   f"CFG-3: symmetric U-Net (levels 5, c_h 16, k 5, repeats 3, reflect) + L1 data loss "
                                   f"+ divergence + Stokes momentum residual, {H}x{W}, per-GPU batch {B}, Adam, "
                                   f"{'HIP-graph replay' if not args.no_graph else 'eager launches'}"
                                   + (f", r_p={args.r_p}" if args.r_p else ""),
                       "precision": {"fp32": "fp32 storage and arithmetic", "bf16": "bf16 storage, bf16 MFMA, f32 accumulate",
                                     "mixed": "f16 storage and f16 MFMA in the forward pass, bf16 storage and bf16 MFMA for every "
                                              "gradient tensor, f32 accumulate: the Trainer's choice for bf16 with a momentum term"}[model.precision],
                       "global_batch": world * B, "grid": [H, W], "parallelism": f"dp{world}",
                       "loss": float(loss)},
            "roofline": roof,
            "hbm_roofline_frac_step": sps / world * eng.algorithmic_bytes_per_sample(args.precision) / (HBM_PEAK_GBS * 1e9),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, H, W)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
