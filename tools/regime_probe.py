#!/usr/bin/env python3
"""Developer probe: forward / backward time of the smooth-flow regime as its amplitude and noise
vary (where does the time go when windows have to follow the motion?)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from artemis_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
H, W = bench.H, bench.W
frames = bench.make_frames(1, 1.0, dev, 103)


def run(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(7):
        e0.record()
        for f in frames:
            fn(f)
        e1.record()
        e1.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1) / len(frames))
    return statistics.median(ts) * 1e3


def fwd(f):
    _lib.check(lib.artemis_synth_forward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                         f["ox"].data_ptr(), f["out"].data_ptr(), 1, 3, H, W, 5, 1, -1, st), "f")


def bwd(f):
    _lib.check(lib.artemis_synth_backward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                          f["ox"].data_ptr(), f["gout"].data_ptr(), None, f["gw"].data_ptr(),
                                          f["ga"].data_ptr(), f["gb"].data_ptr(), 1, 3, H, W, 5, 1, 0, -1,
                                          None, 0, st), "b")


for amp, noise, spacing in [(15, 0.3, 128), (15, 0.0, 128), (5, 0.3, 128), (2, 0.3, 128), (15, 0.3, 256),
                            (15, 0.3, 512), (0, 0.3, 128), (0, 1.0, 128)]:
    g = torch.Generator(device=dev).manual_seed(1)
    for f in frames:
        rn = lambda *s: torch.randn(*s, device=dev, generator=g)
        up = lambda lo: torch.nn.functional.interpolate(lo, size=(H, W), mode="bicubic").expand(1, 25, H, W)
        f["oy"] = (up(rn(1, 1, H // spacing + 2, W // spacing + 2) * amp) + rn(1, 25, H, W) * noise).contiguous()
        f["ox"] = (up(rn(1, 1, H // spacing + 2, W // spacing + 2) * amp) + rn(1, 25, H, W) * noise).contiguous()
    far = float(((f["oy"].abs() > 15) | (f["ox"].abs() > 15)).float().mean())
    print(f"amp {amp:3d} noise {noise:.1f} spacing {spacing:4d}: fwd {run(fwd):7.1f} us  bwd {run(bwd):7.1f} us  "
          f"(|offset| > 15: {far * 100:.1f} % of taps)", flush=True)
