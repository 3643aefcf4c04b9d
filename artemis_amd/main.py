"""Entry point of the harness: the counterpart of the reference's ``main.py`` + ``config.py`` for
the modes that exercise the hot path, Lightning- and CuPy-free, with HIP device selection
(``--device hip:N``; the reference's ``--num_gpu`` is inert, config.py:61,69-72).

    python -m artemis_amd.main --mode interpolate --timesteps 0.125,0.25,0.375,0.5,0.625,0.75,0.875 \
        --frame1_path a.png --frame2_path b.png --frame3_path c.png --frame4_path d.png \
        --model_path model.ckpt --save_path out/ --device hip:0
    python -m artemis_amd.main --mode interpolate --synthetic 1080x1920       # random frames / weights
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m artemis_amd.main \
        --mode train --batch_size 64 --crop 256 --steps 10                      # BASELINE configs[2]

Flags keep the reference's names and defaults where the reference has them (config.py:14-63).
Interpolate mode (main.py:98-124) evaluates ALL timesteps as one batch (``ArTEMIS.interpolate``):
encoder and decoder once per window, the synthesis heads at N = #timesteps -- upstream runs the whole
model once per timestep.  Under ``torchrun`` WHOLE timesteps are dealt to the ranks
(``sharding.shard_items``: 7 timesteps on 8 GPUs = one each, one rank idle; 4/3 on 2 GPUs), no
collective, and every rank runs encoder and decoder for its window.  (The balanced split of one
window -- a row strip of all timesteps per rank, ``sharding.shard_interpolation`` -- is the
library- and bench-level path, ``bench.py --workload interp7``; this CLI does not use it.)
Training data is synthetic (the Vimeo-90K septuplet set, 82 GB, is not part of this repository);
frames on disk are read with PIL when it is importable.
"""
from __future__ import annotations

import argparse
import os

import torch

from .device import add_device_argument, select_device
from .model import ArTEMIS
from .sharding import shard_items


def get_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--mode", choices=["train", "interpolate"], default="interpolate")
    ap.add_argument("--nbr_frame", type=int, default=4)
    ap.add_argument("--joinType", choices=["concat", "add", "none"], default="concat")
    ap.add_argument("--kernel_size", type=int, default=5)
    ap.add_argument("--dilation", type=int, default=1)
    ap.add_argument("--model_path", type=str, default="", help="state_dict or Lightning checkpoint")
    ap.add_argument("--save_path", type=str, default="")
    for i in range(1, 5):
        ap.add_argument(f"--frame{i}_path", type=str, default="")
    ap.add_argument("--timesteps", type=str, default="0.5",
                    help="comma-separated list of timesteps from 0-1 (e.g. '0.25, 0.5, 0.75')")
    ap.add_argument("--synthetic", type=str, default="",
                    help="HxW: random context frames instead of files (benchmarks, smoke runs)")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--beta1", type=float, default=0.9)
    ap.add_argument("--beta2", type=float, default=0.999)
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--max_epoch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="optimizer steps per epoch (synthetic data)")
    ap.add_argument("--crop", type=int, default=256, help="training crop size (RandomCrop(256) upstream)")
    ap.add_argument("--random_seed", type=int, default=103)
    ap.add_argument("--backend", default=os.environ.get("ARTEMIS_DIST_BACKEND", "nccl"))
    ap.add_argument("--composed", action="store_true",
                    help="the reference's per-frame composition instead of the fused head op")
    add_device_argument(ap)
    return ap.parse_args(argv)


def load_weights(model, path):
    """A plain state_dict, or a Lightning checkpoint whose ``state_dict`` keys carry the
    ``model.`` prefix of ``ArTEMISModel.model`` (main.py:44).  Loaded with ``weights_only=True``."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd)


def read_image(path):
    """(1, 3, H, W) float in [0, 1] (utils.py:8-15)."""
    import numpy as np
    from PIL import Image
    arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0).contiguous()


def save_image(tensor, path):
    import numpy as np
    arr = (tensor.detach().clamp(0, 1).permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    try:
        from PIL import Image
        Image.fromarray(arr).save(path)
    except ImportError:
        torch.save(tensor.detach().cpu(), os.path.splitext(path)[0] + ".pt")


def interpolate(args, device, rank=0, world=1):
    torch.manual_seed(args.random_seed)
    model = ArTEMIS(args.nbr_frame, args.joinType, args.kernel_size, args.dilation,
                    fused=not args.composed)
    if args.model_path:
        load_weights(model, args.model_path)
    model.to(device).eval()
    if args.synthetic:
        h, w = (int(v) for v in args.synthetic.lower().split("x"))
        g = torch.Generator().manual_seed(args.random_seed)
        frames = [torch.rand(1, 3, h, w, generator=g).to(device) for _ in range(args.nbr_frame)]
    else:
        paths = [getattr(args, f"frame{i}_path") for i in range(1, 5)]
        if not all(paths):
            raise SystemExit("interpolate mode needs --frame1_path .. --frame4_path (or --synthetic HxW)")
        frames = [read_image(p).to(device) for p in paths]
    timesteps = [float(t) for t in "".join(args.timesteps.split()).split(",")]
    mine = [timesteps[it.timestep] for it in shard_items(1, len(timesteps), rank, world)]
    if not mine:
        return []
    out = model.interpolate(frames, mine)                      # (len(mine), 3, H, W), one batch
    if args.save_path:
        os.makedirs(args.save_path, exist_ok=True)
        for t, img in zip(mine, out):
            save_image(img, os.path.join(args.save_path, f"frame_t={t}.png"))
    return list(zip(mine, out))


def train(args, device):
    from .train import ModelTrainer, synthetic_frames
    tr = ModelTrainer(device, lr=args.lr, betas=(args.beta1, args.beta2), seed=args.random_seed,
                      fused=not args.composed)
    step = 0
    for epoch in range(args.max_epoch):
        for _ in range(args.steps):
            batch = synthetic_frames(args.batch_size, (args.crop, args.crop), 1000 + step, device)
            loss = tr.step(batch)
            if tr.rank == 0:
                print(f"epoch {epoch} step {step}: loss {float(loss):.6f} "
                      f"lr {tr.scheduler.get_last_lr()[0]:.2e}", flush=True)
            step += 1
        tr.end_epoch()
    return tr


def main(argv=None):
    args = get_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # main.py:23-24 (`if args.cuda: torch.backends.cudnn.benchmark = True`): on ROCm this is MIOpen's
    # find mode -- the convolutions of encoder / decoder / sub-networks are benchmarked once per shape
    torch.backends.cudnn.benchmark = True
    if world > 1 and args.mode == "train":
        from .train import rehearsal_env
        rehearsal_env(args.backend)
    if world > 1 and args.backend != "nccl" and args.device == "auto":
        args.device = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    device = select_device(args.device)
    if world > 1 and args.mode == "train":
        kw = {"device_id": device} if args.backend == "nccl" else {}
        torch.distributed.init_process_group(args.backend, **kw)
    if args.mode == "interpolate":
        done = interpolate(args, device, rank, world)
        print(f"rank {rank}: timesteps {[t for t, _ in done]}", flush=True)
    else:
        train(args, device)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
