#!/usr/bin/env python3
"""Per-wave timeline of the forward kernel in a developer build (-DSYNTH_DEV_TILE_CLOCK, see
synth_device.hpp): how long each workgroup holds its CU, how many windows it staged and how many
pixels it left to the per-pixel fix-up -- to see what a launch waits for in a given offset regime.

    tools/build_variants.sh clock:-DSYNTH_DEV_TILE_CLOCK        # -> artemis_amd/lib/exp/libsynth_clock.so
    python tools/tile_clock.py artemis_amd/lib/exp/libsynth_clock.so --regime smooth30
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tools.ab import bind  # noqa: E402

WAVES = 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--regime", default="smooth30")
    ap.add_argument("--radius", type=int, default=-1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = bind(a.lib if os.path.isabs(a.lib) else os.path.join(ROOT, a.lib))
    lib.artemis_synth_dev_set_tile_log.restype = ctypes.c_int
    lib.artemis_synth_dev_set_tile_log.argtypes = [ctypes.c_void_p]
    frames = bench.make_frames(1, 1.0, dev, 103, regime=a.regime)
    H, W = bench.H, bench.W
    st = torch.cuda.current_stream().cuda_stream
    ntiles = (((W + 127) // 128) * ((H + 15) // 16) + 7) // 8 * 8   # the grid is rounded up to 8 XCDs
    log = torch.zeros(ntiles * WAVES * 12, dtype=torch.int64, device=dev)

    def fwd(f):
        assert lib.artemis_synth_forward(f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(),
                                         f["ox"].data_ptr(), f["out"].data_ptr(), 1, 3, H, W, 5, 1,
                                         a.radius, st) == 0

    for f in frames:
        fwd(f)
    torch.cuda.synchronize()
    assert lib.artemis_synth_dev_set_tile_log(log.data_ptr()) == 0
    f = frames[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fwd(f)
    e1.record()
    torch.cuda.synchronize()
    assert lib.artemis_synth_dev_set_tile_log(None) == 0
    L = log.cpu().numpy().reshape(ntiles, WAVES, 12)
    xcd = np.arange(len(L)) % 8              # workgroups are dealt round-robin to the 8 XCDs
    keep = (L[:, :, 3] > 0).any(1)           # (workgroups beyond the last tile log nothing)
    L, xcd = L[keep], xcd[keep]
    ntiles = len(L)
    ok = L[:, :, 3] > 0
    tick = 0.01  # us per tick of the 100 MHz constant clock
    t_begin = L[:, :, 0][ok].min()
    start = np.where(ok, L[:, :, 0], np.iinfo(np.int64).max).min(1)
    end = np.where(ok, L[:, :, 3], 0).max(1)
    dur = (end - start) * tick
    passes = L[:, :, 4].max(1)
    left = L[:, :, 5].sum(1)
    first = np.where(ok, (L[:, :, 1] - L[:, :, 0]), 0).max(1) * tick
    loop = np.where(ok, (L[:, :, 2] - L[:, :, 0]), 0).max(1) * tick
    fix = np.where(ok, (L[:, :, 3] - L[:, :, 2]), 0).max(1) * tick
    print(f"regime {a.regime}: event time {e0.elapsed_time(e1)*1e3:.1f} us; device timeline "
          f"{(end.max() - t_begin) * tick:.1f} us; {ntiles} workgroups")
    print(f"workgroup duration us: median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  "
          f"p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}  sum/256 CUs {dur.sum()/256:.1f}")
    for p in sorted(set(passes.tolist())):
        m = passes == p
        print(f"  extra passes {p}: {m.sum():5d} workgroups, duration median {np.median(dur[m]):6.1f} max {dur[m].max():6.1f} "
              f"| first pass {np.median(first[m]):6.1f} | to end of passes {np.median(loop[m]):6.1f} "
              f"| fix-up median {np.median(fix[m]):6.1f} max {fix[m].max():6.1f} | left pixels median {np.median(left[m]):.0f} max {left[m].max()}")
    # the second pass, per wave that ran it: vote before it, fill + barrier, its taps, what follows
    W2 = L[(L[:, :, 10] > 0)]
    if len(W2):
        med = lambda v: float(np.median(v)) * tick
        print(f"second pass, {len(W2)} waves: vote {med(W2[:, 8] - W2[:, 1]):.1f} us | fill + barrier "
              f"{med(W2[:, 9] - W2[:, 8]):.1f} | taps {med(W2[:, 10] - W2[:, 9]):.1f} | after it (vote, third pass) "
              f"{med(W2[:, 2] - W2[:, 10]):.1f}")
    print("per XCD: workgroups, sum of durations / 32 CUs, last end (us):")
    for k in range(8):
        m = xcd == k
        print(f"  xcd {k}: {m.sum():4d}  {dur[m].sum() / 32:7.1f}  {(end[m].max() - t_begin) * tick:7.1f}")
    # what the launch waits for: the workgroups that end last
    order = np.argsort(end)[::-1][:12]
    print("last to finish (end us, start us, duration, extra passes, left pixels, fix-up us):")
    for i in order:
        print(f"  {(end[i]-t_begin)*tick:7.1f} {(start[i]-t_begin)*tick:7.1f} {dur[i]:6.1f} {passes[i]} {left[i]:5d} {fix[i]:6.1f}")
    # start-time histogram: how many rounds of workgroups
    s = (start - t_begin) * tick
    hist, edges = np.histogram(s, bins=12)
    print("start-time histogram (us):", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))
    # correlation of duration with leftovers
    big = left > 0
    if big.any():
        print(f"workgroups with left pixels: {big.sum()}, their duration median {np.median(dur[big]):.1f}; "
              f"us per left pixel (fix-up / left): {np.median(fix[big] / left[big]):.3f}")


if __name__ == "__main__":
    main()
