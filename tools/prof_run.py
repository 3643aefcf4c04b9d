#!/usr/bin/env python3
"""Minimal launch sequence for rocprofv3 passes: `iters` x (forward, backward) over 4 rotating
1080p operand sets (cold caches, as in bench.py), sigma = 1 by default.

    rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py
    rocprofv3 --pmc FETCH_SIZE -- python3 tools/prof_run.py
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from artemis_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--radius", type=int, default=-1)
    ap.add_argument("--gin", type=int, default=-1, help="-1 none, 0 atomic, 1 deterministic")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    frames = bench.make_frames(1, a.sigma, dev, 103)
    st = torch.cuda.current_stream().cuda_stream
    H, W = bench.H, bench.W
    gin = ws = None
    wsb = 0
    if a.gin >= 0:
        gin = torch.empty_like(frames[0]["inp"])
        wsb = lib.artemis_synth_backward_workspace_bytes(1, 3, H, W, 5, 1, a.gin)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for _ in range(a.iters):
        for f in frames:
            _lib.check(lib.artemis_synth_forward(
                f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(), f["ox"].data_ptr(),
                f["out"].data_ptr(), 1, 3, H, W, 5, 1, a.radius, st), "fwd")
        for f in frames:
            _lib.check(lib.artemis_synth_backward(
                f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(), f["ox"].data_ptr(),
                f["gout"].data_ptr(), gin.data_ptr() if gin is not None else None,
                f["gw"].data_ptr(), f["ga"].data_ptr(), f["gb"].data_ptr(), 1, 3, H, W, 5, 1,
                max(a.gin, 0), a.radius, ws.data_ptr() if ws is not None else None, wsb, st),
                "bwd")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
