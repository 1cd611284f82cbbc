"""Data-parallel trainer for the Stokes surrogate on MI355X — the reference's multigpu.py surface
(ddp_setup, Trainer, load_train_objs, prepare_dataloader, main, CLI flags; reference
multigpu.py:16-1154) on top of the HIP engine.

What differs underneath (MI355X-first, not a translation of the DDP/NCCL call pattern):
  * one process per GPU; parameters and gradients live in ONE flat f32 buffer each
    (hipnet.FlatParams), so the data-parallel exchange is a single RCCL all-reduce(sum) of
    7.3 MB over xGMI per step, with the 1/world_size folded into the fused
    Adam kernel — no DDP wrapper, no bucket reducer, no per-tensor optimizer loop;
  * forward + loss + backward run as C-ABI kernel launches on one HIP stream and can be
    captured into a HIP graph (`use_graph=True`) — the loss is evaluated forward AND backward
    by one fused kernel, and losses are accumulated on the device (no per-step .item() x 6);
  * every rank steps the LR scheduler (the reference only steps it on rank 0).
"""
from __future__ import annotations

import datetime
import os
import sys
import random
import time
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset

from . import _lib as L
from .datasetio import *  # noqa: F401,F403  (reference does the same star import)
from .hipnet import FlatParams
from .losses import StokesLoss
from .pytorch_networks_convae import ConvAE, NewFluidNet, Unet, count_parameters


def ddp_setup(rank, world_size, master_port, backend: Optional[str] = None):
    """One process per GPU; backend 'nccl' is RCCL on ROCm (reference :16-34).  `backend='gloo'`
    is accepted for CPU rehearsals of the host logic."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world_size > 1:
        os.environ.setdefault("MASTER_PORT", str(master_port))
    else:
        os.environ.setdefault("MASTER_PORT", str(random.randint(20000, 60000)))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size,
                                timeout=datetime.timedelta(seconds=36000))


# --------------------------------------------------------------------------------------------------
# flat-buffer data parallelism (host logic is backend-agnostic: exercised with gloo in the CPU tests)
# --------------------------------------------------------------------------------------------------
def broadcast_flat(flat: torch.Tensor, src: int = 0, group=None):
    """Initial parameter broadcast from rank 0 (what DDP's constructor does, reference :69)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":       # rehearsal transport, see allreduce_flat
            host = flat.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            flat.detach().copy_(host)
        else:
            dist.broadcast(flat, src=src, group=group)


def allreduce_flat(flat_grad: torch.Tensor, group=None, async_op: bool = False):
    """ONE all-reduce(sum) of the whole flat gradient buffer; the mean's 1/world is applied by
    the Adam kernel (grad_scale).  Returns the world size (and the work handle if async)."""
    if not (dist.is_available() and dist.is_initialized()):
        return (1, None) if async_op else 1
    world = dist.get_world_size(group)
    work = None
    if world > 1 and flat_grad.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal transport (several ranks on one GPU): stage through the host; RCCL takes the device buffer directly
        host = flat_grad.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat_grad.copy_(host)
    elif world > 1:
        work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return (world, work) if async_op else world


def shard_range(n_items: int, world_size: int, rank: int):
    """Contiguous equal shards, remainder dropped (reference :692-698) -> identical step counts on all ranks."""
    per = int((n_items - n_items % world_size) / world_size)
    return int(per * rank), int(per * (rank + 1))


GSUM_FROM_LOSS = os.environ.get("MANTLE_GSUM_FROM_LOSS", "1") != "0"   # spatial means of the loss gradient from the loss kernel's per-block sums
CB8_LOSS = os.environ.get("MANTLE_CB8_LOSS", "1") != "0"     # the fused loss reads the network output in its CB8 layout (no NCHW copy)


class Trainer:
    _promotion_logged = False

    def __init__(self, model_uvp: torch.nn.Module, model_AD, train_data, cv_data, train_data_init, cv_data_init,
                 optimizer: torch.optim.Optimizer, scheduler, gpu_id: int, save_every: int, nn_dir, p_pred=False,
                 debug=False, network="fluidnet", loss_scale=False, loss_derivative=False, roll_forward=1, epoch=0,
                 loss_type="curl", *, norm="l1", lambda_mom=0.0, precision=None, use_graph=False, log_every=100):
        if network not in ("unet", "iunet", "convae", "newfluidnet"):
            raise NotImplementedError(f"network={network!r}: the HIP path covers 'unet', 'convae' and 'newfluidnet' "
                                      "(SURVEY.md §8f row N1; the older FluidNet trunk is not built)")
        fluid = "fluidnet" in network          # the reference's `"fluidnet" in self.net` branch of get_loss (:138)
        if fluid and lambda_mom != 0.0:
            raise NotImplementedError("the momentum residual needs the temperature output of the Unet")
        if model_AD is not None:
            raise NotImplementedError("model_AD (advection net) is out of scope of the training hot path")
        self.gpu_id = gpu_id
        self.device = torch.device("cuda", gpu_id) if isinstance(gpu_id, int) else torch.device(gpu_id)
        self.train_data, self.cv_data = train_data, cv_data
        self.train_data_init, self.cv_data_init = train_data_init, cv_data_init
        self.optimizer, self.scheduler, self.save_every = optimizer, scheduler, save_every
        self.model_uvp = model_uvp.to(self.device)
        if precision == "bf16" and lambda_mom != 0.0 and os.environ.get("MANTLE_MIXED", "1") != "0":
            # second differences x 126^2 of a network whose forward tensors carry 8 mantissa bits are rounding noise (2 x the
            # exact value at 506^2): the momentum term needs the f16 forward pass of the "mixed" mode (11 bits; gradients
            # stay bf16).  MANTLE_MIXED=0 keeps plain bf16.
            precision = "mixed"
            if not Trainer._promotion_logged:
                Trainer._promotion_logged = True
                print(f"[mantle] precision 'bf16' with lambda_mom != 0 runs as '{precision}' (f16 forward tensors, bf16 "
                      "gradients); MANTLE_MIXED=0 keeps plain bf16", file=sys.stderr, flush=True)
        if precision is not None:
            self.model_uvp.set_precision(precision)
        self.model_AD = None
        self.p_pred, self.nn_dir, self.debug, self.net = p_pred, nn_dir, debug, network
        self.loss_scale, self.loss_derivative, self.roll_forward = loss_scale, loss_derivative, roll_forward
        self.start_epoch, self.loss_type = epoch, loss_type
        self.use_graph, self.log_every = use_graph, log_every
        # flat parameter / gradient / Adam-moment buffers; nn.Parameters become views
        self.flat = FlatParams(self.model_uvp, self.device)
        self.exp_avg = torch.zeros_like(self.flat.param)
        self.exp_avg_sq = torch.zeros_like(self.flat.param)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._lr_host = None
        broadcast_flat(self.flat.param)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        a_bound = getattr(self.model_uvp, "a_bound", 10.0)
        self.loss = StokesLoss(p_pred if network != "convae" else False,
                               loss_type if network != "convae" else "mae", loss_scale, loss_derivative, norm=norm,
                               lambda_mom=lambda_mom, a_bound=a_bound, has_T=not fluid)
        if network == "convae" or fluid:
            self.chan_scale = None                    # these nets see gVTp as it is (reference :139)
        else:
            cs = torch.ones(self.model_uvp._graph.c_in, dtype=torch.float32)
            cs[0] = 0.25
            cs[1] = 0.25
            cs[2] = 1.0 / roll_forward
            self.chan_scale = cs.to(self.device)      # xc/4, yc/4, dt/roll_forward (reference :234-248)
        self.l1 = torch.nn.L1Loss()
        self.losses = [0.0] * 6
        self.losses_cv = [0.0] * 6
        self._graph = None
        self._static = None

    # ------------------------------------------------------------------ reference-API helpers
    def get_lr(self, optimizer=None):
        for param_group in self.optimizer.param_groups:
            return param_group["lr"]

    def loss_fn(self, x_true, x_pred):
        """(scaled loss, plain L1) of one field (reference :122-134) — convenience mirror on device
        tensors; the training path evaluates all terms in the fused HIP loss kernel instead."""
        if self.loss_scale:
            maxs = torch.amax(x_true, dim=(1, 2), keepdim=True)
            mins = torch.amin(x_true, dim=(1, 2), keepdim=True)
            scaler = torch.clip(1.0 / (maxs - mins), 1.0, 10.0)
            bc_scaler = torch.full_like(x_true, 11.0)
            bc_scaler[:, 2:-2, 2:-2] = 1.0
            return torch.mean(torch.abs((x_true - x_pred) * scaler * bc_scaler)), self.l1(x_true, x_pred)
        loss = self.l1(x_true, x_pred)
        return loss, loss

    # ------------------------------------------------------------------ loss
    def _features(self, gVTp):
        m = self.model_uvp
        if self.net == "convae":
            return m._run_graph(gVTp) if m.loss_type != "curl" else m(gVTp)
        # the net sees the first c_in channels of gVTp with xc, yc / 4 and dt / roll_forward; the scaling
        # is applied inside the input-pack kernel
        m._chan_scale = self.chan_scale
        return m.features(gVTp)

    def _rolls(self):
        return self.roll_forward > 1 and self.net in ("unet", "iunet")      # the reference's other branches ignore it (:138)

    def _roll_chain(self, gVTp, paras, fwd):
        """roll_forward = R > 1 (reference :207-248): R * R network evaluations in a chain, per round R - 1 pre-steps under
        no_grad and one more; every input is rebuilt from channels 0..5 of the batch and the previous evaluation's (T, u, v),
        with the viscosity channel re-derived from T after the pre-steps only (:246-247 are commented out).  Only the LAST
        evaluation reaches the loss with a gradient path (every earlier output enters the next input through a no_grad
        step), so the first R * R - 1 run forward-only here; returns the last evaluation's input (raw channels: the
        input-pack kernel scales xc, yc, dt).  `fwd(x)` -> network output [N, c_out, H, W] f32."""
        if paras is None:
            raise ValueError("roll_forward > 1 needs paras (RaQ, FKT, FKP per sample) for the viscosity channel")
        R = int(self.roll_forward)
        N, C, H, W = gVTp.shape
        if C < 10:
            raise ValueError(f"roll_forward > 1 rebuilds a 10-channel input; gVTp has {C} channels")
        key = (N, C, H, W, str(gVTp.device))
        if getattr(self, "_roll_key", None) != key:              # (a captured step allocates these in its warm-up pass)
            self._roll_x = torch.empty((N, C, H, W), dtype=torch.float32, device=gVTp.device)
            self._roll_uvT = torch.empty((3, N, H, W), dtype=torch.float32, device=gVTp.device)
            self._roll_key = key
        x = self._roll_x
        x.copy_(gVTp)
        p3 = paras.to(torch.float32).reshape(N, 3).contiguous()
        m = self.model_uvp
        for i in range(R * R - 1):
            y = fwd(x)
            Cc = y.shape[1]
            if m.loss_type == "curl":                            # Unet.forward's head (u, v from the streamfunction; T clipped)
                u, v, T = self._roll_uvT[0], self._roll_uvT[1], self._roll_uvT[2]
                L.call("mc_curl_head_fwd", L.ptr(y), y.data_ptr() + 4 * H * W, N, H, W, Cc * H * W, float(m.a_bound), 0.0, 1.5,
                       L.ptr(u), L.ptr(v), L.ptr(T), L.stream())
                pu, pv, pT, stride = L.ptr(u), L.ptr(v), L.ptr(T), H * W
            else:
                pu, pv, pT, stride = y.data_ptr(), y.data_ptr() + 4 * H * W, y.data_ptr() + 8 * H * W, Cc * H * W
            L.call("mc_roll_forward_update", L.ptr(x), C, pu, pv, pT, stride, L.ptr(p3), int(i % R != R - 1), N, H, W,
                   L.stream())
        return x

    def get_loss(self, gVTp, uvp, scaler, paras=None, yc=None):
        """Autograd-visible 6-tuple (loss, loss_true_u, loss_true_v, loss_p, loss_T, mean mass) exactly as the
        reference combines them (:250-305); `loss.backward()` runs the HIP backward."""
        gVTp = gVTp.to(self.device, torch.float32)
        uvp = uvp.to(self.device, torch.float32).contiguous()
        if self._rolls():
            with torch.no_grad():
                gVTp = self._roll_chain(gVTp.contiguous(), None if paras is None else paras.to(self.device), self._features)
        y = self._features(gVTp)
        sc = None
        if self.loss.lambda_mom != 0.0:
            sc = scaler.to(self.device, torch.float32) if scaler is not None else None
        out8 = self.loss(y, uvp, yc.to(self.device) if yc is not None else None,
                         paras.to(self.device) if paras is not None else None, sc)
        return out8[0], out8[1].detach(), out8[2].detach(), out8[3].detach(), out8[4].detach(), out8[5].detach()

    # ------------------------------------------------------------------ fused step
    def _sync_lr(self):
        lr = float(self.get_lr())
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def _adam_args(self):
        g = self.optimizer.param_groups[0]
        b1, b2 = g.get("betas", (0.9, 0.999))
        return float(b1), float(b2), float(g.get("eps", 1e-8)), float(g.get("weight_decay", 0.0))

    def _fwd_bwd(self, gVTp, uvp, yc, paras, scaler, train=True, eng=None, loss=None):
        """pack -> forward -> fused loss fwd+bwd -> backward, all raw kernel launches (graph-capturable)."""
        m = self.model_uvp
        if eng is None:
            eng, loss = m.engine(), self.loss
        if not self.flat.bound():
            raise RuntimeError("model parameters were re-allocated (e.g. .to()/.double()) after the Trainer "
                               "flattened them")
        params = self.flat.views(self.flat.param)
        self._check_single_mesh(yc)
        key = (tuple(gVTp.shape), str(gVTp.device))
        if eng is m.engine() and getattr(self, "_ybuf_key", None) != key:   # network output buffer, allocated outside any capture
            eng.configure(gVTp.shape[0], gVTp.shape[2], gVTp.shape[3], gVTp.device)
            self._ybuf = torch.empty((gVTp.shape[0], eng.g.c_out, eng.out_h, eng.out_w), dtype=torch.float32, device=gVTp.device)
            self._ybuf_key = key
        ybuf = self._ybuf if eng is m.engine() else None
        if self._rolls():
            gVTp = self._roll_chain(gVTp, paras, lambda x: eng.forward(x, params, self.chan_scale, out=ybuf))
        # (one-launch loss: it reads the last convolution's output where it lies, and forward() returns None instead of an
        # NCHW copy when the head ends in an f32 CB8 tensor)
        y = eng.forward(gVTp, params, self.chan_scale, out=ybuf, unpack=not (loss.fusable() and CB8_LOSS))
        if y is None:
            out8, gy = loss.evaluate(None, uvp, yc, paras, scaler, cb8=eng.output_cb8())
        else:
            out8, gy = loss.evaluate(y, uvp, yc, paras, scaler)
        if train:
            self.flat.grad.zero_()
            eng.backward(gy, params, self.flat.views(self.flat.grad), gsum=loss.gradient_sums() if GSUM_FROM_LOSS else None)
        return out8

    def _check_single_mesh(self, yc):
        """The momentum residual evaluates the viscosity with ONE depth grid yc [H, W] for the whole batch (sample 0's).
        A batch whose samples carry different grids would silently get wrong residuals.  Checked on EVERY batch the caller
        hands over (one small reduction + host sync; not keyed on the tensor's address: the caching allocator hands the
        next batch the same one) -- in a captured step BEFORE the batch is copied into the static buffer, never inside a
        capture.  A loader that fills `input_buffers()` in place passes the static buffer itself: no copy, no check, and the
        loader is responsible for its batches (`validate_mesh`)."""
        if yc is None or self.loss.lambda_mom == 0.0 or yc.dim() < 3 or yc.shape[0] <= 1:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        self.validate_mesh(yc)

    @staticmethod
    def validate_mesh(yc):
        y2 = yc.reshape(yc.shape[0], -1)
        if not bool((y2 == y2[:1]).all()):
            raise ValueError("the momentum residual needs one depth grid yc for the whole batch (all samples on the same mesh)")

    def _optim_step(self):
        b1, b2, eps, wd = self._adam_args()
        L.call("mc_adam_step_flat", L.ptr(self.flat.param), L.ptr(self.flat.grad), L.ptr(self.exp_avg),
               L.ptr(self.exp_avg_sq), self.flat.numel, L.ptr(self.lr_dev), b1, b2, eps, wd, 1.0 / self.world,
               L.ptr(self.step_count), L.stream())

    def train_step(self, gVTp, uvp, yc=None, paras=None, scaler=None):
        """zero_grad -> forward -> loss -> backward -> all-reduce -> Adam (reference _run_batch :307-320),
        returning the 8 loss scalars as a DEVICE tensor (no host sync)."""
        self._sync_lr()
        if self.use_graph:
            return self._graph_step(gVTp, uvp, yc, paras, scaler)
        out8 = self._fwd_bwd(gVTp, uvp, yc, paras, scaler, train=True)
        allreduce_flat(self.flat.grad)
        self._optim_step()
        return out8

    def _graph_step(self, gVTp, uvp, yc, paras, scaler):
        if self._graph is None:
            st = dict(gVTp=gVTp.clone(), uvp=uvp.clone(),
                      yc=None if yc is None else yc.clone().float(),
                      paras=None if paras is None else paras.clone().float(),
                      scaler=None if scaler is None else scaler.clone().float())
            self._static = st
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # warm-up: allocates every engine buffer outside the capture
                self._fwd_bwd(st["gVTp"], st["uvp"], st["yc"], st["paras"], st["scaler"], train=True)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.device)
            # with a process group alive its watchdog thread polls events while we capture: only calls of THIS thread may
            # invalidate the capture (the default "global" mode would turn the watchdog's event query into a capture error)
            mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, **mode):
                self._static_out = self._fwd_bwd(st["gVTp"], st["uvp"], st["yc"], st["paras"], st["scaler"],
                                                 train=True)
            # Adam is a graph of its own: for world > 1 the flat-gradient all-reduce runs between the two replays
            self._graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_opt, **mode):
                self._optim_step()
            # the graph holds the engine's and the loss's device pointers: pin their shapes (another batch size through
            # eval_step / get_loss / model(x) would otherwise re-plan them and the next replay would touch freed memory)
            self.model_uvp.engine().freeze()
            self.loss.freeze()
        st = self._static
        # a captured step reads its inputs from fixed buffers; a caller that fills `input_buffers()` in place (a loader
        # writing the next batch straight into them) passes those very tensors and no copy is made
        if yc is not None and st["yc"] is not None and yc.data_ptr() != st["yc"].data_ptr():
            self._check_single_mesh(yc)                      # (the captured step itself cannot check anything)
        for k, v in (("gVTp", gVTp), ("uvp", uvp), ("yc", yc), ("paras", paras), ("scaler", scaler)):
            if st[k] is not None and v is not None and v.data_ptr() != st[k].data_ptr():
                st[k].copy_(v.reshape(st[k].shape), non_blocking=True)
        self._graph.replay()
        if self.world > 1:
            allreduce_flat(self.flat.grad)
        self._graph_opt.replay()
        return self._static_out

    def input_buffers(self):
        """The device tensors a captured training step reads (available after the first `train_step` with
        use_graph): dict(gVTp, uvp, yc, paras, scaler).  Writing a batch into them and passing them to `train_step`
        skips the device-to-device staging copy."""
        if self._graph is None:
            raise RuntimeError("input_buffers() is available after the first graph-captured train_step")
        return dict(self._static)

    def eval_step(self, gVTp, uvp, yc=None, paras=None, scaler=None):
        """Forward + loss only.  Once a training graph has been captured its engine is pinned to the captured shape; an
        evaluation batch of another size runs on an engine / loss pair of its own (same parameters)."""
        if self._graph is not None and tuple(gVTp.shape) != tuple(self._static["gVTp"].shape):
            if getattr(self, "_eval_pair", None) is None:
                import copy
                from .engine import Engine
                lo = copy.copy(self.loss)
                lo._shape, lo._frozen = None, False
                self._eval_pair = (Engine(self.model_uvp._graph, self.model_uvp.precision), lo)
            return self._fwd_bwd(gVTp, uvp, yc, paras, scaler, train=False, eng=self._eval_pair[0], loss=self._eval_pair[1])
        return self._fwd_bwd(gVTp, uvp, yc, paras, scaler, train=False)

    # ------------------------------------------------------------------ reference loop
    def _prep(self, gVTp, uvp, scaler, paras, yc):
        gVTp = gVTp.to(self.device, torch.float32, non_blocking=True).contiguous()
        uvp = uvp.to(self.device, torch.float32, non_blocking=True).contiguous()
        need = self.loss.lambda_mom != 0.0
        yc = yc.to(self.device, torch.float32) if (need and yc is not None) else None
        paras = paras.to(self.device, torch.float32) if ((need or self._rolls()) and paras is not None) else None
        scaler = scaler.to(self.device, torch.float32) if (need and scaler is not None) else None
        return gVTp, uvp, yc, paras, scaler

    def _run_batch(self, gVTp, uvp, scaler, is_train, paras=None, yc=None, sync=True):
        gVTp, uvp, yc, paras, scaler = self._prep(gVTp, uvp, scaler, paras, yc)
        out8 = self.train_step(gVTp, uvp, yc, paras, scaler) if is_train else self.eval_step(gVTp, uvp, yc, paras,
                                                                                             scaler)
        if not sync:
            return out8
        return [float(v) for v in out8[:6].tolist()]

    def _unpack(self, data):
        if self.net in ("unet", "iunet"):
            return data[0], data[1], data[2], data[3], data[4]      # x, y, scaler, paras, yc (ADTimeDataset)
        return data[0], data[1], data[3] if len(data) > 3 else None, None, None

    def _run_epoch(self, epoch):
        print(f"[GPU{self.gpu_id}] Epoch {epoch} | Steps: {len(self.train_data)}")
        acc = torch.zeros(8, dtype=torch.float64, device=self.device)
        i = -1
        for i, data in enumerate(self.train_data):
            t0 = time.time()
            gVTp, uvp, scaler, paras, yc = self._unpack(data)
            acc += self._run_batch(gVTp, uvp, scaler, True, paras=paras, yc=yc, sync=False).double()
            if i % self.log_every == 0:                     # the only host sync in the hot loop
                print(epoch, (acc[:6] / (i + 1)).tolist(), time.time() - t0)
        self.losses = (acc[:6] / max(i + 1, 1)).tolist()
        acc.zero_()
        i_cv = -1
        with torch.no_grad():
            print(f"[GPU{self.gpu_id}] Epoch CV {epoch} | Steps: {len(self.cv_data)}")
            for i_cv, data in enumerate(self.cv_data):
                gVTp, uvp, scaler, paras, yc = self._unpack(data)
                acc += self._run_batch(gVTp, uvp, scaler, False, paras=paras, yc=yc, sync=False).double()
        self.losses_cv = (acc[:6] / max(i_cv + 1, 1)).tolist()

    def _save_checkpoint(self, epoch):
        """File names, state_dict keys and the log line format of the reference (:412-436)."""
        ckp = {k: v.detach().clone() for k, v in self.model_uvp.state_dict().items()}
        if self.debug:
            if epoch % 10 == 0:
                torch.save(ckp, self.nn_dir + "fluidnet_uvp.pt")
        else:
            torch.save(ckp, self.nn_dir + str(epoch) + "_fluidnet_uvp.pt")
            print("-------------------------------------------")
            print(epoch, self.losses, self.losses_cv, self.get_lr(self.optimizer))
            print("-------------------------------------------")
        with open(self.nn_dir + "fluidnet_uvpT.txt", "a") as writer:
            writer.write(str(epoch) + "," + str(self.losses[1:]) + "," + str(self.losses_cv[1:]) + ","
                         + str(self.get_lr(self.optimizer)) + "\n")

    def train(self, max_epochs: int):
        for epoch in range(self.start_epoch, max_epochs):
            t0 = time.time()
            self.losses = [0.0] * 6
            self.losses_cv = [0.0] * 6
            self._run_epoch(epoch)
            t1 = time.time()
            if (self.gpu_id == 0 or self.world == 1) and epoch % self.save_every == 0:
                self._save_checkpoint(epoch)
                print(t1 - t0)
            self.scheduler.step()          # on EVERY rank (reference: rank 0 only -> diverging LRs)


# --------------------------------------------------------------------------------------------------
def build_model(network, levels, c_i, c_h, c_o, rank, act_fn, r_p, loss_type, use_symm, repeats, kernel,
                use_skip=False, p_pred=False, spectral_conv=False, dilation=1, a_bound=10, blurr=False, dropout=0.0):
    """Model construction of load_train_objs (reference :492-609) without the `.double()` (the HIP path
    computes in f32 or bf16 with f32 master weights)."""
    dev = torch.device("cuda", rank) if isinstance(rank, int) else rank
    if network in ("unet", "iunet"):
        return Unet(levels, c_i, c_h, c_o, dev, act_fn, r_p, loss_type, use_symm=use_symm, dilation=dilation,
                    a_bound=a_bound, repeats=repeats, use_skip=use_skip, f=kernel, p_pred=p_pred,
                    spectral_conv=spectral_conv, blurr=blurr, drop_rate=dropout)
    if network == "convae":
        return ConvAE(levels, c_i, c_h, c_o, dev, act_fn, r_p, loss_type, use_symm=use_symm, dilation=dilation,
                      a_bound=a_bound, repeats=repeats, use_skip=use_skip, f=kernel, p_pred=p_pred,
                      spectral_conv=spectral_conv, blurr=blurr)
    if network == "newfluidnet":
        return NewFluidNet(levels, c_i, c_h, c_o, dev, act_fn, r_p, loss_type, use_symm=use_symm, dilation=dilation,
                           a_bound=a_bound, repeats=repeats, use_skip=use_skip, f=kernel, p_pred=p_pred,
                           spectral_conv=spectral_conv, blurr=blurr, drop_rate=dropout)
    raise NotImplementedError(f"network={network!r} is outside the HIP path (unet, convae, newfluidnet)")


def parse_restart_log(nn_dir, milestones):
    """Resume bookkeeping of the reference (:621-656): last epoch and lr from the text log, milestones rebased."""
    with open(nn_dir + "fluidnet_uvpT.txt") as fw:
        lines = fw.readlines()
    epoch = int(lines[-1].split(",")[0])
    start_lr = float(lines[-1].split(",")[-1])
    if epoch > milestones[-1]:
        milestones = []
    else:
        i0 = int(np.where(np.asarray(milestones) > epoch)[0][0])
        milestones = [milestones[i0] - epoch] + [m - epoch for m in milestones[i0 + 1:]]
    return epoch, start_lr, milestones


def load_train_objs(rank, world_size, nn_dir, data_dir, levels, c_i, c_h, c_o, act_fn, r_p, loss_type, use_symm,
                    repeats, kernel, milestones, sims_vec, times_vec, sims_vec_init, times_vec_init, use_skip=False,
                    p_pred=False, spectral_conv=False, dilation=1, a_bound=10, restart=False, advect=False,
                    network="fluidnet", scale=True, noise=0.0, debug=False, blurr=False, l2_reg=0.0, dropout=0.0,
                    roll_forward=1, factor=2, multi_scales=[], synthetic=None):
    """Same signature and return tuple as the reference (:453-769).  `synthetic=dict(n=..., H=..., W=...)`
    substitutes the seeded synthetic dataset for the absent /plp_scr1 data files."""
    if advect:
        raise NotImplementedError("advect=True (ADNet) is out of scope")
    model_uvp = build_model(network, levels, c_i, c_h, c_o, rank, act_fn, r_p, loss_type, use_symm, repeats, kernel,
                            use_skip, p_pred, spectral_conv, dilation, a_bound, blurr, dropout)
    print(count_parameters(model_uvp))
    if restart:
        epoch, start_lr, milestones = parse_restart_log(nn_dir, milestones)
        sd = torch.load(nn_dir + str(epoch) + "_fluidnet_uvp.pt", map_location="cpu", weights_only=True)
        model_uvp.load_state_dict({k: v.float() for k, v in sd.items()})
        epoch += 1
        print("Restarting from epoch, lr, milestones")
        print(epoch, start_lr, milestones)
    else:
        epoch, start_lr = 0, 1e-3
        with open(nn_dir + "fluidnet_uvpT.txt", "w") as writer:
            writer.write("Epoch, train loss, val loss, learning rate \n")
    dataset, dataset_init = {}, {}
    for an in ["train", "cv"]:
        if synthetic is not None:
            n = synthetic["n"] if an == "train" else max(synthetic["n"] // 4, 1)
            lo, hi = shard_range(n * world_size, world_size, rank if isinstance(rank, int) else 0)
            dataset[an] = SyntheticMantleDataset(hi - lo, synthetic["H"], synthetic["W"], p_pred=p_pred,  # noqa: F405
                                                 seed=1234 + 1000 * (an == "cv") + lo, network=network, c_i=c_i)
            dataset_init[an] = None
            continue
        lo, hi = shard_range(len(sims_vec[an]), world_size, rank)
        print(rank, "splitting ", len(sims_vec[an]), " samples into ", world_size, " chunks of size ", hi - lo)
        if network in ("unet", "iunet"):
            dataset[an] = ADTimeDataset(data_dir, an, scale, is_init=False, p_pred=p_pred, noise=noise,  # noqa: F405
                                        debug=debug, sims_vec=sims_vec[an][lo:hi], times_vec=times_vec[an][lo:hi],
                                        roll_forward=roll_forward)
            dataset_init[an] = None
        else:
            # the FluidNet family trains on NewADDataset items (reference :684, 726-760)
            kw = dict(scale=scale, p_pred=p_pred, noise=noise, debug=debug)
            dataset[an] = NewADDataset(data_dir, an, is_init=False, sims_vec=sims_vec[an][lo:hi],  # noqa: F405
                                       times_vec=times_vec[an][lo:hi], **kw)
            if debug:
                dataset_init[an] = None
            else:
                lo_i, hi_i = shard_range(len(sims_vec_init[an]), world_size, rank)
                dataset_init[an] = NewADDataset(data_dir, an, is_init=True, sims_vec=sims_vec_init[an][lo_i:hi_i],  # noqa: F405
                                                times_vec=times_vec_init[an][lo_i:hi_i], **kw)
    optimizer = torch.optim.Adam([{"params": model_uvp.parameters(), "lr": start_lr, "weight_decay": l2_reg}])
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=milestones, gamma=0.5)
    return dataset, dataset_init, model_uvp, None, optimizer, scheduler, epoch


def prepare_dataloader(dataset: Dataset, batch_size: int, world_size, rank):
    return DataLoader(dataset, batch_size=batch_size, pin_memory=torch.cuda.is_available(), shuffle=True,
                      drop_last=True)


def main(rank: int, world_size: int, save_every: int, total_epochs: int, batch_size: int, nn_dir, data_dir, levels,
         c_i, c_h, c_o, act_fn, r_p, loss_type, use_symm, repeats, kernel, milestones, sims_vec, times_vec,
         sims_vec_init, times_vec_init, use_skip=False, p_pred=False, spectral_conv=False, dilation=1, a_bound=10,
         restart=False, advect=False, network="fluidnet", debug=False, scale=True, blurr=False, master_port=366,
         l2_reg=0.0, dropout=0.0, loss_scale=False, loss_derivative=False, roll_forward=1, factor=2, multi_scales=[],
         synthetic=None, precision=None, lambda_mom=0.0, use_graph=False):
    ddp_setup(rank, world_size, master_port)
    dataset, dataset_init, model_uvp, model_AD, optimizer, scheduler, epoch = load_train_objs(
        rank, world_size, nn_dir, data_dir, levels, c_i, c_h, c_o, act_fn, r_p, loss_type, use_symm, repeats, kernel,
        milestones, sims_vec, times_vec, sims_vec_init, times_vec_init, use_skip=use_skip, p_pred=p_pred,
        spectral_conv=spectral_conv, dilation=dilation, a_bound=a_bound, restart=restart, advect=advect,
        network=network, scale=scale, debug=debug, blurr=blurr, l2_reg=l2_reg, dropout=dropout,
        roll_forward=roll_forward, factor=factor, multi_scales=multi_scales, synthetic=synthetic)
    train_data = prepare_dataloader(dataset["train"], batch_size, world_size, rank)
    cv_data = prepare_dataloader(dataset["cv"], batch_size, world_size, rank)
    trainer = Trainer(model_uvp, model_AD, train_data, cv_data, None, None, optimizer, scheduler, rank, save_every,
                      nn_dir, p_pred, debug, network, loss_scale, loss_derivative, roll_forward, epoch=epoch,
                      loss_type=loss_type, precision=precision, lambda_mom=lambda_mom, use_graph=use_graph)
    trainer.train(total_epochs)
    if dist.is_initialized():
        dist.destroy_process_group()


def build_arg_parser():
    """The reference's flags with the same names and defaults (:915-972) plus the build's own."""
    import argparse
    p = argparse.ArgumentParser(description="Train convnet")
    p.add_argument("-a", "--act_fn", type=str, default="gelu")
    p.add_argument("-l", "--levels", type=int, default=6)
    p.add_argument("-f", "--c_h", type=int)
    p.add_argument("-fac", "--factor", type=int, default=2)
    p.add_argument("-p", "--r_p", type=str, default="replicate")
    p.add_argument("-gpu", "--gpu_nums", type=str)
    p.add_argument("-lt", "--loss_type", type=str, default="curl")
    p.add_argument("-d", "--dilation", type=int, default=1)
    p.add_argument("-b", "--batch_size", type=int)
    p.add_argument("-s", "--use_symm", type=int)
    p.add_argument("-ab", "--a_bound", type=int)
    p.add_argument("-r", "--repeats", type=int)
    p.add_argument("-rst", "--restart", type=int, default=0)
    p.add_argument("-sk", "--use_skip", type=int, default=0)
    p.add_argument("-k", "--kernel", type=int)
    p.add_argument("-sc", "--scale", type=int, default=1)
    p.add_argument("-l_sc", "--loss_scale", type=int, default=1)
    p.add_argument("-l_de", "--loss_derivative", type=int, default=0)
    p.add_argument("-blurr", "--blurr", type=int, default=0)
    p.add_argument("-pp", "--p_pred", type=int, default=0)
    p.add_argument("-ad", "--advect", type=int, default=0)
    p.add_argument("-n", "--noise", type=float, default=0.0)
    p.add_argument("-deb", "--debug", type=int)
    p.add_argument("-net", "--network", type=str, default="fluidnet")
    p.add_argument("-spectral", "--spectral_conv", type=int, default=0)
    p.add_argument("-mp", "--master_port", type=int, default=366)
    p.add_argument("-l2", "--l2_reg", type=float, default=0.0)
    p.add_argument("-d_r", "--drop_rate", type=float, default=0.0)
    p.add_argument("-roll", "--roll_forward", type=int, default=1)
    p.add_argument("-scales", "--multi_scales", type=float, nargs="+", default=[])
    # build-specific (paths are arguments instead of hard-coded; synthetic data; precision; momentum weight)
    p.add_argument("--data_dir", type=str, default="/plp_scr1/agar_sh/data/TPH/")
    p.add_argument("--nn_root", type=str, default="./trained_networks/")
    p.add_argument("--synthetic", type=int, nargs=3, metavar=("N", "H", "W"), default=None)
    p.add_argument("--precision", type=str, default=None, choices=[None, "fp32", "bf16", "mixed"])
    p.add_argument("--lambda_mom", type=float, default=0.0)
    p.add_argument("--use_graph", type=int, default=0)
    p.add_argument("--epochs", type=int, default=None)
    return p


def run_name(a):
    """Run-directory name encoding the hyper-parameters (reference :1011-1057)."""
    b = lambda v: str(v == 1)  # noqa: E731
    f_nn = (a.network + "_levels_" + str(a.levels) + "_" + a.act_fn + "_" + str(a.c_h) + "_" + a.r_p + "_" + a.loss_type
            + "_" + b(a.use_symm) + "_ab" + str(a.a_bound) + "_b" + str(a.batch_size) + "_r" + str(a.repeats) + "_k"
            + str(a.kernel) + "_fa" + str(a.factor) + "_ad" + b(a.advect) + "_p_pred" + b(a.p_pred) + "_l2"
            + str(a.l2_reg) + "_l_sc" + b(a.loss_scale) + "_l_de" + b(a.loss_derivative) + "_deb" + b(a.debug))
    if "unet" in a.network:
        f_nn += "_roll" + str(a.roll_forward) + "_new"
    if a.blurr == 1:
        f_nn += "_blurr"
    return f_nn


def channels_for(network, loss_type, p_pred):
    """c_i / c_o selection of the reference CLI (:1072-1087)."""
    if "fluidnet" in network:
        c_i, c_o = 7, 3
    elif network == "convae":
        c_i, c_o = 3, 3
    elif network == "unet":
        c_i, c_o = 11, 4
        if not p_pred:
            c_i -= 1
    else:
        raise ValueError(network)
    if loss_type == "curl":
        c_o -= 1
    if not p_pred:
        c_o -= 1
    return c_i, c_o


def cli(argv=None):
    import torch.multiprocessing as mp
    a = build_arg_parser().parse_args(argv)
    if a.gpu_nums:
        os.environ["HIP_VISIBLE_DEVICES"] = a.gpu_nums
    world_size = max(torch.cuda.device_count(), 1)
    nn_dir = os.path.join(a.nn_root, run_name(a)) + "/"
    os.makedirs(nn_dir, exist_ok=True)
    debug = a.debug == 1
    if debug:
        epochs, milestones = 1500, [20, 200, 400, 600, 800, 1000]
    else:
        epochs, milestones = 150, [20, 40, 60, 80, 120, 180]   # reference list is unsorted (App. A.9)
    if a.epochs is not None:
        epochs = a.epochs
    p_pred = a.p_pred == 1
    c_i, c_o = channels_for(a.network, a.loss_type, p_pred)
    if a.network == "unet" and p_pred:
        c_i = 10      # get_loss feeds the net ten channels even when p_pred (reference :234-248)
    sims_vec, times_vec, sims_init, times_init = {}, {}, {}, {}
    synthetic = None
    if a.synthetic is not None:
        synthetic = dict(n=a.synthetic[0], H=a.synthetic[1], W=a.synthetic[2])
        for an in ("train", "cv"):
            sims_vec[an], times_vec[an], sims_init[an], times_init[an] = [], [], None, None
    else:
        for an in ("train", "cv"):
            init = get_indices_time if a.network == "unet" else get_indices   # noqa: F405
            sims_vec[an], times_vec[an] = init(a.data_dir, an, is_init=False, debug=debug,
                                               roll_forward=a.roll_forward)
            sims_init[an], times_init[an] = None, None
    args = (world_size, 1, epochs, a.batch_size, nn_dir, a.data_dir, a.levels, c_i, a.c_h, c_o, a.act_fn, a.r_p,
            a.loss_type, a.use_symm == 1, a.repeats, a.kernel, milestones, sims_vec, times_vec, sims_init, times_init,
            a.use_skip == 1, p_pred, a.spectral_conv == 1, a.dilation, a.a_bound, a.restart == 1, a.advect == 1,
            a.network, debug, a.scale == 1, a.blurr == 1, a.master_port, a.l2_reg, a.drop_rate, a.loss_scale == 1,
            a.loss_derivative == 1, a.roll_forward, a.factor, a.multi_scales, synthetic, a.precision, a.lambda_mom,
            a.use_graph == 1)
    if world_size == 1:
        main(0, *args)
    else:
        mp.spawn(main, args=args, nprocs=world_size)


if __name__ == "__main__":
    cli()
