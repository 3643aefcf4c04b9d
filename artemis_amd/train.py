"""A Lightning-free training step: the whole ArTEMIS model (``--model artemis``, this package's
restatement on stock ops + the fused head op) or one ChronoSynth head alone (``--model head``)
+ L1 loss + Adamax + MultiStepLR, one process per GPU with ``torch.distributed`` -- the pieces of
the reference's ``main.py`` that touch the hot path
(``main.py:45`` Adamax, ``:89-93`` MultiStepLR([40,60,75,85,95,100], gamma 0.5) stepped per epoch,
``loss.py:9`` L1; Lightning DDP ``:133``).

The op itself needs no collective: the batch is sharded over ranks (``sharding.shard_batch``) and
the only exchange of a step is DDP's gradient all-reduce of the module's parameters (29.09 M fp32 for
the full model, RCCL over xGMI with backend ``nccl``; ``gloo`` works for rehearsals on fewer GPUs
than ranks -- see ``rehearsal_env``).  With ``--model head`` the encoder/decoder are stood in for by
given feature tensors.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m artemis_amd.train \
        --model artemis --batch 64 --size 256 --steps 5        # BASELINE configs[2]
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.nn as nn

from .device import add_device_argument, select_device
from .model import ArTEMIS, ChronoSynth
from .sharding import shard_bounds


def rehearsal_env(backend):
    """DistributedDataParallel over ``gloo`` with CUDA tensors needs the HIP runtime's queue-thread
    mode on ROCm 7.0 / this PyTorch build: ProcessGroupGloo stages each bucket through a pinned host
    copy on a side stream that waits on an event of the compute stream and is then synchronised
    from gloo's worker thread; in direct-dispatch mode that ``hipStreamSynchronize`` never returns
    once the event is still pending when the wait is issued (any model whose backward keeps the
    queue busy; one process is enough; none of this package's kernels need be involved --
    DESIGN.md 6, tools/ddp_probe.py, tools/dbg/hip_wait_repro.py).  ``AMD_DIRECT_DISPATCH=0`` must
    be in the environment before the first HIP call of the process.  Backend ``nccl`` (RCCL, the
    production path) is not affected and is left alone."""
    if backend != "nccl":
        os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")


def synthetic_batch(batch, num_features, size, seed, device):
    """features (B, F, 4, H/2, W/2), 4 frames (B, 3, H, W), ground truth (B, 3, H, W), times (B,)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, W = size
    feats = torch.randn(batch, num_features, 4, (H + 1) // 2, (W + 1) // 2, generator=g)
    frames = [torch.rand(batch, 3, H, W, generator=g) for _ in range(4)]
    gt = torch.rand(batch, 3, H, W, generator=g)
    t = torch.rand(batch, generator=g)
    return (feats.to(device), [f.to(device) for f in frames], gt.to(device), t.to(device))


def synthetic_frames(batch, size, seed, device):
    """4 context frames (B, 3, H, W) in [0, 1], ground truth (B, 3, H, W), times (B,) drawn from the
    septuplet's {0.25, 0.5, 0.75} (data/preprocessing/vimeo90k_septuplet_process.py)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, W = size
    frames = [torch.rand(batch, 3, H, W, generator=g) for _ in range(4)]
    gt = torch.rand(batch, 3, H, W, generator=g)
    t = torch.tensor([0.25, 0.5, 0.75])[torch.randint(0, 3, (batch,), generator=g)]
    return [f.to(device) for f in frames], gt.to(device), t.to(device)


class HeadTrainer:
    """One ChronoSynth head under Adamax + MultiStepLR, data-parallel over the process group."""

    #: main.py:89 -- MultiStepLR milestones, counted in EPOCHS (Lightning steps the scheduler once
    #: per epoch by default), gamma 0.5 (:93)
    MILESTONES = (40, 60, 75, 85, 95, 100)

    def __init__(self, device, num_features=64, lr=2e-4, betas=(0.9, 0.999), milestones=MILESTONES,
                 seed=103, fused=True):
        torch.manual_seed(seed)          # config.py:62; same init on every rank
        self.device = device
        self.module = self.build(num_features, fused).to(device)
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        self.model = self.module
        # DistributedDataParallel whenever a process group is up (also a group of one rank, so that
        # a one-GPU box can run the RCCL path): bucketed gradient all-reduce overlapped with the
        # backward -- RCCL with backend "nccl", gloo for rehearsals (see rehearsal_env).
        self.ddp = torch.distributed.is_initialized()
        if self.ddp:
            self.model = nn.parallel.DistributedDataParallel(
                self.module, device_ids=[device.index], broadcast_buffers=False)
        self.optimizer = torch.optim.Adamax(self.module.parameters(), lr=lr, betas=betas)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, list(milestones), gamma=0.5)
        self.loss = nn.L1Loss(reduction="mean")

    @staticmethod
    def build(num_features, fused):
        return ChronoSynth(4, num_features, 5, 1, apply_softmax=True, fused=fused)

    def shard(self, batch):
        """This rank's slice of a global batch (features, frames, gt, times).  The batch must
        divide evenly: DDP averages the per-rank mean losses' gradients, which is the global mean
        only for equal shards (and an empty shard would make the L1 mean NaN)."""
        feats, frames, gt, t = batch
        if feats.size(0) % self.world != 0:
            raise ValueError(f"global batch {feats.size(0)} is not a multiple of the world size "
                             f"{self.world}: DDP's gradient average would not be the global mean")
        b, e = shard_bounds(feats.size(0), self.rank, self.world)
        return feats[b:e], [f[b:e] for f in frames], gt[b:e], t[b:e]

    def step(self, global_batch):
        feats, frames, gt, t = self.shard(global_batch)
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(feats, frames, tuple(gt.shape[-2:]), t)
        loss = self.loss(out, gt)
        loss.backward()                  # DDP all-reduces (averages) the parameter gradients here
        self.optimizer.step()
        return loss.detach()

    def end_epoch(self):
        """Step the LR schedule: once per epoch, as Lightning does for main.py:89-95."""
        self.scheduler.step()


class ModelTrainer(HeadTrainer):
    """The whole ArTEMIS model (main.py:37-95 without Lightning): forward on 4 context frames and
    the output times, L1 on the full-resolution output only (loss.py:17-18), Adamax, MultiStepLR;
    DDP all-reduces the 29.09 M parameters' gradients."""

    @staticmethod
    def build(num_features, fused):
        return ArTEMIS(num_inputs=4, joinType="concat", kernel_size=5, dilation=1, fused=fused)

    def shard(self, batch):
        frames, gt, t = batch
        if gt.size(0) % self.world != 0:
            raise ValueError(f"global batch {gt.size(0)} is not a multiple of the world size "
                             f"{self.world}: DDP's gradient average would not be the global mean")
        b, e = shard_bounds(gt.size(0), self.rank, self.world)
        return [f[b:e] for f in frames], gt[b:e], t[b:e]

    def step(self, global_batch):
        frames, gt, t = self.shard(global_batch)
        self.optimizer.zero_grad(set_to_none=True)
        _, _, out = self.model(frames, t)
        loss = self.loss(out, gt)
        loss.backward()                  # DDP: the 29.09 M gradients' all-reduce, bucketed
        self.optimizer.step()
        return loss.detach()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    add_device_argument(ap)
    ap.add_argument("--model", choices=["head", "artemis"], default="head")
    ap.add_argument("--steps", type=int, default=3, help="optimizer steps per epoch")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="global batch (sharded over ranks)")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--backend", default=os.environ.get("ARTEMIS_DIST_BACKEND", "nccl"))
    ap.add_argument("--composed", action="store_true", help="reference composition instead of the fused head")
    ap.add_argument("--dump", default="", help="rank 0 writes the final parameters here (tests)")
    ap.add_argument("--cudnn_benchmark", type=int, default=1,
                    help="main.py:23-24 sets torch.backends.cudnn.benchmark (MIOpen find mode on ROCm); "
                         "0 = immediate mode (short test runs: finding costs seconds per conv shape)")
    args = ap.parse_args(argv)
    if os.environ.get("ARTEMIS_TRACE_HANG"):      # debugging aid: where is every thread after N s?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ARTEMIS_TRACE_HANG"]), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ARTEMIS_FORCE_DIST=1: bring the process group up for a single rank too (under
    # torch.distributed.run), so that DistributedDataParallel + RCCL run on a one-GPU box
    dist_on = world > 1 or os.environ.get("ARTEMIS_FORCE_DIST") == "1"
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    if dist_on:
        rehearsal_env(args.backend)
    if world > 1 and args.backend != "nccl" and args.device == "auto":
        # rehearsal on fewer GPUs than ranks: wrap around the visible devices
        args.device = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    device = select_device(args.device)
    if dist_on:
        kw = {"device_id": device} if args.backend == "nccl" else {}
        torch.distributed.init_process_group(args.backend, **kw)
    full = args.model == "artemis"
    tr = (ModelTrainer if full else HeadTrainer)(device, num_features=args.features,
                                                 fused=not args.composed)
    import time
    for i in range(args.steps * args.epochs):
        t0 = time.perf_counter()
        if full:
            batch = synthetic_frames(args.batch, (args.size, args.size), 1000 + i, device)
        else:
            batch = synthetic_batch(args.batch, args.features, (args.size, args.size), 1000 + i, device)
        loss = tr.step(batch)
        if (i + 1) % args.steps == 0:
            tr.end_epoch()
        if dist_on:
            torch.distributed.all_reduce(loss, op=torch.distributed.ReduceOp.AVG
                                         if args.backend == "nccl" else torch.distributed.ReduceOp.SUM)
            if args.backend != "nccl":
                loss /= world
        if tr.rank == 0:
            print(f"step {i}: loss {float(loss):.6f} lr {tr.scheduler.get_last_lr()[0]:.2e} "
                  f"({time.perf_counter() - t0:.3f} s)", flush=True)
    if args.dump and tr.rank == 0:
        torch.save({k: v.detach().cpu() for k, v in tr.module.state_dict().items()}, args.dump)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
