#!/usr/bin/env python3
"""A/B of gradInput (atomic mode) across builds in one process.
usage: gin_ab.py lib... [--radius=R] [--regime=sigma:4|smooth30|pan]"""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tools.ab import bind

radius = -1
regime = None
libs = []
for a in sys.argv[1:]:
    if a.startswith("--radius="):
        radius = int(a.split("=")[1])
    elif a.startswith("--regime="):
        regime = a.split("=")[1]
    else:
        libs.append((os.path.basename(a), bind(os.path.join(ROOT, a))))
dev = torch.device("cuda:0")
frames = bench.make_frames(1, 1.0, dev, 103, regime=regime)
st = torch.cuda.current_stream().cuda_stream
H, W = bench.H, bench.W
gin = torch.empty_like(frames[0]["inp"])
wsb = libs[0][1].artemis_synth_backward_workspace_bytes(1, 3, H, W, 5, 1, 1)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
T = {n: [] for n, _ in libs}
for rnd in range(8):
    for name, lib in libs:
        f = frames[rnd % 4]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.artemis_synth_backward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                        f["ox"].data_ptr(), f["gout"].data_ptr(), gin.data_ptr(), None, None, None,
                                        1, 3, H, W, 5, 1, 0, radius, ws.data_ptr(), wsb, st)
        e1.record(); e1.synchronize()
        assert rc == 0
        if rnd:
            T[name].append(e0.elapsed_time(e1))
for n, ts in T.items():
    print(f"{n:28s} gin atomic r={radius} {regime or 'sigma:1'}: median {statistics.median(ts)*1e3:8.1f} us  min {min(ts)*1e3:8.1f}")
