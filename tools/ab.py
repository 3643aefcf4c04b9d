#!/usr/bin/env python3
"""A/B builds of libartemis_synth in ONE process, interleaved rounds, cold operands (4 rotating
1080p frame sets as in bench.py).  usage: python tools/ab.py libA.so libB.so ... [--sigma S]
[--radius R] [--shape N,H,W]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from artemis_amd import _lib  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    return lib


def head_ab(a):
    """fused head forward / backward of every library, interleaved rounds, 3 rotating operand sets"""
    dev = torch.device("cuda:0")
    libs = [(os.path.basename(p), bind(p if os.path.isabs(p) else os.path.join(ROOT, p))) for p in a.libs]
    H, W, B, T = bench.H, bench.W, a.batch, bench.FRAMES
    sets = []
    for i in range(3):
        frames, w, al, be, occ, gout = bench.head_operands(B, dev, a.regime or f"sigma:{a.sigma}", 31 + i)
        out = torch.empty(B, 3, H, W, device=dev)
        gw, ga, gb = (torch.empty_like(w) for _ in range(3))
        gocc = torch.empty_like(occ)
        ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])
        sets.append(dict(frames=frames, w=w, a=al, b=be, occ=occ, gout=gout, out=out, gw=gw, ga=ga, gb=gb,
                         gocc=gocc, ptrs=ptrs))
    st = torch.cuda.current_stream().cuda_stream
    print(f"head regime {a.regime or 'sigma:%g' % a.sigma} radius {a.radius} B {B} size {H}x{W}")

    def fwd(lib, s):
        assert lib.artemis_synth_head_forward(s["ptrs"], T, s["w"].data_ptr(), s["a"].data_ptr(), s["b"].data_ptr(),
                                              s["occ"].data_ptr(), s["out"].data_ptr(), B, H, W, H, W, -2, -2, 0,
                                              a.radius, st) == 0

    def bwd(lib, s):
        assert lib.artemis_synth_head_backward(s["ptrs"], T, s["w"].data_ptr(), s["a"].data_ptr(), s["b"].data_ptr(),
                                               s["occ"].data_ptr(), s["gout"].data_ptr(), s["gw"].data_ptr(),
                                               s["ga"].data_ptr(), s["gb"].data_ptr(), s["gocc"].data_ptr(), B, H, W,
                                               H, W, -2, -2, 0, a.radius, st) == 0

    Tm = {(n, k): [] for n, _ in libs for k in ("fwd", "bwd")}
    for rnd in range(a.rounds + 1):
        for name, lib in libs:
            for kind, fn in (("fwd", fwd), ("bwd", bwd)):
                evs = []
                for s in sets:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(lib, s)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                if rnd:
                    Tm[(name, kind)].append(sum(x.elapsed_time(y) for x, y in evs) / len(evs))
    px = B * H * W
    for (name, kind), ts in Tm.items():
        med = statistics.median(ts)
        nb = (1276.0 if kind == "fwd" else 2492.0) * px
        print(f"{name:32s} head {kind}: median {med*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us  {nb/med/1e6:7.0f} GB/s "
              f"({nb/med/1e6/80:5.1f}% of 8 TB/s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--regime", default="", help="bench.make_offsets regime (smooth30, pan, sigma:4); default sigma:<--sigma>")
    ap.add_argument("--radius", type=int, default=-1)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", default="", help="H,W of the output (default 1080,1920)")
    ap.add_argument("--head", action="store_true", help="the fused 4-frame head (forward / backward) instead of the op")
    a = ap.parse_args()
    if a.size:
        bench.H, bench.W = (int(v) for v in a.size.split(","))
    if a.head:
        return head_ab(a)
    dev = torch.device("cuda:0")
    libs = [(os.path.basename(p), bind(p if os.path.isabs(p) else os.path.join(ROOT, p))) for p in a.libs]
    frames = bench.make_frames(a.batch, a.sigma, dev, 103, regime=a.regime or None, size=(bench.H, bench.W))
    print(f"regime {a.regime or 'sigma:%g' % a.sigma} radius {a.radius}")
    st = torch.cuda.current_stream().cuda_stream
    H, W, B = bench.H, bench.W, a.batch

    def fwd(lib, f):
        assert lib.artemis_synth_forward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                         f["ox"].data_ptr(), f["out"].data_ptr(), B, 3, H, W, 5, 1,
                                         a.radius, st) == 0

    def bwd(lib, f):
        assert lib.artemis_synth_backward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                          f["ox"].data_ptr(), f["gout"].data_ptr(), None,
                                          f["gw"].data_ptr(), f["ga"].data_ptr(), f["gb"].data_ptr(),
                                          B, 3, H, W, 5, 1, 0, a.radius, None, 0, st) == 0

    T = {(n, k): [] for n, _ in libs for k in ("fwd", "bwd")}
    for rnd in range(a.rounds + 1):
        for name, lib in libs:
            for kind, fn in (("fwd", fwd), ("bwd", bwd)):
                evs = []
                for f in frames:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(lib, f)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                if rnd:
                    T[(name, kind)].append(sum(x.elapsed_time(y) for x, y in evs) / len(evs))
    px = B * H * W
    for (name, kind), ts in T.items():
        med = statistics.median(ts)
        nb = (324.0 if kind == "fwd" else 624.0) * px
        print(f"{name:32s} {kind}: median {med*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us  {nb/med/1e6:7.0f} GB/s")
    for name, _ in libs:
        tot = statistics.median(T[(name, "fwd")]) + statistics.median(T[(name, "bwd")])
        print(f"{name:32s} step(4 frames) {4*tot:7.3f} ms -> {px/(4*tot)/1e3:8.1f} Mpix/s "
              f"({px*3792/(4*tot*1e-3)/8e12*100:5.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
