#!/usr/bin/env python3
"""Developer timing of the fused ChronoSynth head (forward, backward) over head shapes.
usage: python tools/head_sweep.py B,H,W [B,H,W ...] [--radius R]"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd import _lib  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    radius = -1
    padded = "--padded" in sys.argv[1:]
    for a in sys.argv[1:]:
        if a.startswith("--radius="):
            radius = int(a.split("=")[1])
        if a.startswith("--small="):   # small-grid threshold (tiles); default: the library's
            _lib.load().artemis_synth_set_small_grid_tiles(int(a.split("=")[1]))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    T = 4
    for spec in args:
        B, H, W = map(int, spec.split(","))
        g = torch.Generator(device=dev).manual_seed(1)
        frames = [torch.rand(B, 3, H + (4 if padded else 0), W + (4 if padded else 0), device=dev, generator=g)
                  for _ in range(T)]
        fh, fw, org = (H + 4, W + 4, 0) if padded else (H, W, -2)
        w = torch.softmax(torch.randn(B, T, 25, H, W, device=dev, generator=g), 2)
        a = torch.randn(B, T, 25, H, W, device=dev, generator=g)
        b = torch.randn(B, T, 25, H, W, device=dev, generator=g)
        occ = torch.softmax(torch.randn(B, T, H, W, device=dev, generator=g), 1)
        gout = torch.randn(B, 3, H, W, device=dev, generator=g)
        out = torch.empty(B, 3, H, W, device=dev)
        gw, ga, gb = (torch.empty_like(w) for _ in range(3))
        gocc = torch.empty_like(occ)
        ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])

        def fwd():
            _lib.check(lib.artemis_synth_head_forward(
                ptrs, T, w.data_ptr(), a.data_ptr(), b.data_ptr(), occ.data_ptr(), out.data_ptr(),
                B, H, W, fh, fw, org, org, 0, radius, st), "fwd")

        def bwd():
            _lib.check(lib.artemis_synth_head_backward(
                ptrs, T, w.data_ptr(), a.data_ptr(), b.data_ptr(), occ.data_ptr(), gout.data_ptr(),
                gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), gocc.data_ptr(), B, H, W, fh, fw, org, org,
                0, radius, st), "bwd")

        res = {}
        for name, fn, nbytes in (("fwd", fwd, 1276.0), ("bwd", bwd, 2492.0)):
            ts = []
            reps = 5  # back-to-back launches per event pair: no idle gap (clock ramp) in the timing
            for i in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                if i > 1:
                    ts.append(e0.elapsed_time(e1) / reps)
            med = statistics.median(ts)
            res[name] = med
            print(f"head B={B} {H}x{W} {name}: {med*1e3:8.1f} us  {nbytes*B*H*W/med/1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
