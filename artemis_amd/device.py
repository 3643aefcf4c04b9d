"""HIP device selection for harnesses around the op (the reference's ``--num_gpu`` flag is inert
and its op launches on whatever device CuPy finds: config.py:61, cupy_module/synth.py:258).

    parser = argparse.ArgumentParser(); add_device_argument(parser)
    device = select_device(parser.parse_args().device)     # "hip:1", "cuda:1", "1", "auto"

Under ``torchrun`` ("auto"), the device is LOCAL_RANK.  ``HIP_VISIBLE_DEVICES`` is honoured by the
runtime as usual (indices here count visible devices).  The op itself needs none of this: it runs
on the device of its tensors, on torch's current stream.
"""
from __future__ import annotations

import os
import re


def parse_device(spec) -> int:
    """'hip:N' | 'cuda:N' | 'gpu:N' | 'N' | 'auto' | None -> visible-device index."""
    if spec is None or str(spec).lower() in ("auto", "hip", "cuda", "gpu"):
        return int(os.environ.get("LOCAL_RANK", "0"))
    m = re.fullmatch(r"(?:(?:hip|cuda|gpu):)?(\d+)", str(spec).strip().lower())
    if not m:
        raise ValueError(f"device {spec!r}: expected hip:N, cuda:N, N or auto")
    return int(m.group(1))


def select_device(spec=None):
    """Make the chosen HIP device current and return it as a ``torch.device``."""
    import torch

    index = parse_device(spec)
    count = torch.cuda.device_count()
    if count == 0:
        raise RuntimeError("no HIP device visible (the synthesis op has no CPU path)")
    if index >= count:
        raise ValueError(f"device index {index} but only {count} HIP device(s) visible "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '')!r})")
    torch.cuda.set_device(index)
    return torch.device("cuda", index)  # torch names HIP devices "cuda"


def add_device_argument(parser, default="auto"):
    parser.add_argument("--device", type=str, default=default,
                        help="HIP device: hip:N / cuda:N / N, or auto = LOCAL_RANK (torchrun) else 0")
    return parser
