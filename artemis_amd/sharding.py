"""Batch x timestep sharding of the synthesis op over the GPUs of one node.

Every output element of the op depends only on its own sample (``intSample`` indexes every tensor
of the reference kernels, cupy_module/synth.py:19,26-28,41-44) and ChronoSynth calls the op once
per context frame and per head (model/chrono_synth.py:115-127), so the work items
``(sample, timestep)`` are independent: they are dealt to the ranks in contiguous, balanced blocks
and the data path needs NO collective.  One process per GPU (``torch.distributed``; backend
``nccl`` = RCCL on ROCm, ``gloo`` in the CPU tests); a collective appears only where a caller wants
the results of all ranks in one place (``gather_outputs``), which inference normally does not --
each rank writes its own frames -- and in the training step's gradient all-reduce, which belongs
to the model, not to this op (the op has no parameters).

Host logic only: nothing here touches a device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous block of ``n_items`` that ``rank`` owns.  Blocks differ in
    size by at most one; the first ``n_items % world`` ranks hold the larger ones."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    if n_items < 0:
        raise ValueError("n_items must be >= 0")
    q, r = divmod(n_items, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


@dataclass(frozen=True)
class WorkItem:
    sample: int    # index into the batch
    timestep: int  # index into the list of output times


def shard_items(batch: int, timesteps: int, rank: int, world: int) -> List[WorkItem]:
    """This rank's (sample, timestep) items; timestep varies fastest so that a rank's items share
    samples (and therefore encoder features) as much as possible.  Interpolate mode at 7 timesteps
    on 8 GPUs gives one timestep per GPU and one idle rank (BASELINE.json configs[3])."""
    b, e = shard_bounds(batch * timesteps, rank, world)
    return [WorkItem(i // timesteps, i % timesteps) for i in range(b, e)]


@dataclass(frozen=True)
class InterpolationShard:
    timesteps: Tuple[int, ...]  # indices into the list of output times this rank computes ...
    y0: int                     # ... for output rows [y0, y1) of the frame
    y1: int


def shard_interpolation(timesteps: int, height: int, rank: int, world: int) -> InterpolationShard:
    """Interpolate mode for ONE window (BASELINE.json configs[3]: 7 timesteps of a 1080p window).
    When the ranks divide the timesteps, every rank takes whole frames (`shard_items`).  Otherwise
    dealing whole timesteps leaves ranks idle -- 7 timesteps on 2, 4 or 8 GPUs: the busiest rank
    holds 4/7, 2/7, 1/7 of the work, 87.5 % efficiency at best -- so every rank takes a row strip of
    ALL timesteps instead: one launch of N = timesteps over rows [y0, y1), addressed through the
    explicit frame geometry (org_y = y0 - pad; the frames stay whole, weights / offsets / occlusion
    exist for the strip only).  Still no collective: the op is pointwise in its output pixel."""
    if timesteps % world == 0:
        b, e = shard_bounds(timesteps, rank, world)
        return InterpolationShard(tuple(range(b, e)), 0, height)
    y0, y1 = shard_bounds(height, rank, world)
    return InterpolationShard(tuple(range(timesteps)), y0, y1)


def shard_batch(tensors: Sequence, rank: int, world: int):
    """Slice dim 0 of each tensor to this rank's samples (views, no copy)."""
    n = tensors[0].shape[0]
    b, e = shard_bounds(n, rank, world)
    return [t[b:e] for t in tensors]


def row_strips(height: int, world: int, halo: int) -> List[Tuple[int, int, int, int]]:
    """Spatial sharding plan for one large frame (BASELINE.json configs[4]): rank r produces output
    rows [y0, y1) and needs padded-frame rows [y0 - halo, y1 + (k-1)d + 1 + halo) clamped to the
    frame, halo = tile/offset radius.  weight/offset rows need no halo (the op is pointwise in
    them).  Returns (y0, y1, in_y0, in_y1_exclusive) per rank in padded-frame coordinates for
    k = 5, d = 1."""
    out = []
    for r in range(world):
        y0, y1 = shard_bounds(height, r, world)
        lo = max(0, y0 - halo)
        hi = min(height + 4, y1 + 4 + 1 + halo)
        out.append((y0, y1, lo, hi))
    return out


def gather_outputs(local, batch: int, group=None):
    """All-gather per-rank outputs (dim 0 = this rank's samples) into the full batch order.
    Only for callers that need every rank's result in one place; uses
    ``torch.distributed.all_gather`` on padded equal-size blocks (no data-path use)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, r, world) for r in range(world)]
    longest = max(e - b for b, e in sizes)
    pad = local.new_zeros((longest,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: e - b] for p, (b, e) in zip(parts, sizes)], 0)
