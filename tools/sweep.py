#!/usr/bin/env python3
"""Developer sweep on one GPU: per-kernel time / achieved algorithmic GB/s of forward and fused
backward over offset regimes x tile radii x shapes (interleaved rounds in one process, medians).

    python tools/sweep.py [--shape 1,1080,1920] [--sigmas 0.045,1,4,16] [--radii -1,0,2,8] [--gin]
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="1,1080,1920")
    ap.add_argument("--sigmas", default="0.045,1,4,16")
    ap.add_argument("--radii", default="-1,0,2,8")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--gin", action="store_true", help="also time gradInput (atomic, deterministic)")
    ap.add_argument("--smooth", action="store_true", help="add a smooth-flow offset regime")
    args = ap.parse_args()
    N, H, W = map(int, args.shape.split(","))
    sigmas = [float(s) for s in args.sigmas.split(",")]
    radii = [int(r) for r in args.radii.split(",")]
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(103)
    frame = torch.rand(N, 3, H, W, device=dev)
    inp = torch.nn.functional.pad(frame, (2, 2, 2, 2), mode="replicate").contiguous()
    w = torch.softmax(torch.randn(N, 25, H, W, device=dev), 1)
    gout = torch.randn(N, 3, H, W, device=dev)
    out = torch.empty(N, 3, H, W, device=dev)
    gw, ga, gb = (torch.empty(N, 25, H, W, device=dev) for _ in range(3))
    gin = torch.empty_like(inp)
    ws_bytes = lib.artemis_synth_backward_workspace_bytes(N, 3, H, W, 5, 1, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if args.gin else None
    px = N * H * W
    fwd_b, bwd_b = 324.0 * px, 624.0 * px
    # gradInput's compulsory bytes: read w, alpha, beta (300 B) + gradOutput (12), zero and write
    # gradInput (2 x 12 B) = 336 B per pixel
    gin_b = 336.0 * px

    regimes = [(f"sigma={s:g}", lambda s=s: (torch.randn(N, 25, H, W, device=dev) * s,
                                             torch.randn(N, 25, H, W, device=dev) * s))
               for s in sigmas]
    if args.smooth:
        def smooth():
            lo = torch.randn(N, 25, H // 32 + 2, W // 32 + 2, device=dev) * 3
            up = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear")
            return ((up + torch.randn_like(up) * 0.1).contiguous(),
                    (up.flip(1) + torch.randn_like(up) * 0.1).contiguous())
        regimes.append(("smooth3", smooth))

        def smooth30():  # large coherent motion with steep gradients (bench.make_offsets)
            import bench
            g = torch.Generator(device=dev).manual_seed(7)
            return bench.make_offsets("smooth30", (N, 25, H, W), dev, g)
        regimes.append(("smooth30", smooth30))

        def boundary():  # an object moving (+28, -22) px over a third of a still frame
            oy = torch.randn(N, 25, H, W, device=dev) * 0.4
            ox = torch.randn(N, 25, H, W, device=dev) * 0.4
            oy[:, :, H // 4:3 * H // 4, W // 3:2 * W // 3] += 28
            ox[:, :, H // 4:3 * H // 4, W // 3:2 * W // 3] -= 22
            return oy, ox
        regimes.append(("object(28,-22)", boundary))

        def pan():  # a camera pan: every tap displaced by (20, -35) px + N(0, 1)
            return ((torch.randn(N, 25, H, W, device=dev) + 20.0).contiguous(),
                    (torch.randn(N, 25, H, W, device=dev) - 35.0).contiguous())
        regimes.append(("pan(20,-35)", pan))

    def timeit(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    print(f"shape N={N} {H}x{W}; GB/s = algorithmic bytes (324 fwd / 624 bwd / 336 gradInput per pixel) / time")
    for name, mk in regimes:
        oy, ox = mk()
        variants = {}
        for r in radii:
            variants[f"fwd r={r}"] = (lambda r=r: _lib.check(lib.artemis_synth_forward(
                inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), out.data_ptr(),
                N, 3, H, W, 5, 1, r, st), "fwd"), fwd_b)
            variants[f"bwd r={r}"] = (lambda r=r: _lib.check(lib.artemis_synth_backward(
                inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(),
                None, gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), N, 3, H, W, 5, 1, 0, r,
                None, 0, st), "bwd"), bwd_b)
        if args.gin:
            # the whole backward in one call (tap gradients + gradInput: the scale's maxima then come
            # from the tap kernel), per radius; the printed increment over `bwd r=` is gradInput's cost
            for r in radii:
                for mode, nm in ((0, "atomic"), (1, "determ")):
                    variants[f"all {nm} r={r}"] = (lambda r=r, mode=mode: _lib.check(lib.artemis_synth_backward(
                        inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(),
                        gin.data_ptr(), gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), N, 3, H, W, 5, 1, mode, r,
                        ws.data_ptr(), ws_bytes, st), "all"), bwd_b + gin_b)
            for mode, nm in ((0, "gin atomic"), (1, "gin determ")):
                variants[nm] = (lambda mode=mode: _lib.check(lib.artemis_synth_backward(
                    inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(),
                    gin.data_ptr(), None, None, None, N, 3, H, W, 5, 1, mode, -1,
                    ws.data_ptr(), ws_bytes, st), "gin"),
                    gin_b)
        times = {k: [] for k in variants}
        for k, (fn, _) in variants.items():
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for k, (fn, _) in variants.items():
                times[k].append(timeit(fn))
        print(f"-- {name}")
        for k, (fn, nbytes) in variants.items():
            med = statistics.median(times[k])
            extra = ""
            if k.startswith("all "):
                base = statistics.median(times["bwd " + k.split()[-1]])
                extra = f"  gradInput's share {1e3 * (med - base):7.1f} us"
            print(f"   {k:18s} {med*1e3:9.1f} us  min {min(times[k])*1e3:9.1f} us  "
                  f"{nbytes/med/1e6:8.1f} GB/s  ({nbytes/med/1e6/8000*100:5.1f}% of 8 TB/s){extra}")


if __name__ == "__main__":
    main()
