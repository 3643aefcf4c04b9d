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
    ap.add_argument("--head", action="store_true", help="also run the fused ChronoSynth head pair")
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
    if a.head:
        import ctypes
        T = 4
        ptrs = (ctypes.c_void_p * T)(*[f["inp"].data_ptr() for f in frames])
        g = torch.Generator(device=dev).manual_seed(5)
        w = torch.softmax(torch.randn(1, T, 25, H, W, device=dev, generator=g), 2)
        al = torch.randn(1, T, 25, H, W, device=dev, generator=g) * a.sigma
        be = torch.randn(1, T, 25, H, W, device=dev, generator=g) * a.sigma
        occ = torch.softmax(torch.randn(1, T, H, W, device=dev, generator=g), 1)
        gw, ga, gb = (torch.empty_like(w) for _ in range(3))
        gocc = torch.empty_like(occ)
        out = frames[0]["out"]
        gout = frames[0]["gout"]
        for _ in range(a.iters):
            _lib.check(lib.artemis_synth_head_forward(
                ptrs, T, w.data_ptr(), al.data_ptr(), be.data_ptr(), occ.data_ptr(), out.data_ptr(),
                1, H, W, H + 4, W + 4, 0, 0, 0, a.radius, st), "head fwd")
            _lib.check(lib.artemis_synth_head_backward(
                ptrs, T, w.data_ptr(), al.data_ptr(), be.data_ptr(), occ.data_ptr(), gout.data_ptr(),
                gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), gocc.data_ptr(), 1, H, W, H + 4, W + 4,
                0, 0, 0, a.radius, st), "head bwd")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
