#!/usr/bin/env python3
"""Kernel-level summary of ONE steady-state train step from a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 -m artemis_amd.train --model artemis
--batch 8 --size 256 --steps 4`): the dispatches between the first fused-head forward launch of the
second-to-last step and that of the last step are exactly one step's kernels (cyclically shifted), free
of first-use compilation and of MIOpen's find-mode trials.  Writes a CSV: top kernels by time, launch
count, GPU-busy time and the wall time of the window.

    python tools/train_step_profile.py gpurun_out/r3_train_prof profiles/r3_train_step_kernel_stats.csv
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=110):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[:n] + "..."


def main():
    src, dst = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {src}")
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    heads = [s for s, _, n in rows if "synth_fwd_tapmajor" in n]
    if len(heads) < 6 or len(heads) % 3:
        raise SystemExit(f"{len(heads)} head-forward launches: expected 3 per step, at least 2 steps")
    t0, t1 = heads[-6], heads[-3]          # first head forward of the last two steps
    win = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in win:
        agg[n][0] += 1
        agg[n][1] += e - s
        busy += e - s
    wall = t1 - t0
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["# one steady-state train step (window between the first fused-head forward launches of "
                    "the last two steps)"])
        w.writerow(["# steps in trace", len(heads) // 3, "launches in window", len(win), "distinct kernels",
                    len(agg), "wall_us", round(wall / 1e3, 1), "gpu_busy_us", round(busy / 1e3, 1),
                    "busy_frac", round(busy / wall, 3)])
        w.writerow(["Name", "Calls", "TotalUs", "AverageUs", "PercentOfBusy"])
        for n, (c, t) in order[:top]:
            w.writerow([short(n), c, round(t / 1e3, 1), round(t / c / 1e3, 2), round(100 * t / busy, 2)])
        rest = order[top:]
        w.writerow([f"(the other {len(rest)} kernels)", sum(c for _, (c, _) in rest),
                    round(sum(t for _, (_, t) in rest) / 1e3, 1), "", round(100 * sum(t for _, (_, t) in rest) / busy, 2)])
        synth = [(n, c, t) for n, (c, t) in agg.items() if "synth_" in n or "asynth" in n]
        w.writerow(["(this library's kernels, all)", sum(c for _, c, _ in synth),
                    round(sum(t for _, _, t in synth) / 1e3, 1), "", round(100 * sum(t for _, _, t in synth) / busy, 2)])
    print(open(dst).read())


if __name__ == "__main__":
    main()
