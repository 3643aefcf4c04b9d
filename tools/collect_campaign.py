#!/usr/bin/env python3
"""Copy what tools/campaign.sh left under gpurun_out/r3z into profiles/ (tracked) and print the
numbers DESIGN.md quotes.  usage: python tools/collect_campaign.py [tag]   (default tag: r3)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r3z")
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
P = os.path.join(ROOT, "profiles")

subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(SRC, "pmc"),
                       os.path.join(SRC, "pmc_calib"), "round 3, final kernels", "lean"], stdout=subprocess.DEVNULL)
shutil.copy(glob.glob(os.path.join(SRC, "prof_bench", "**", "*kernel_stats.csv"), recursive=True)[0],
            os.path.join(P, f"{tag}_kernel_stats.csv"))
shutil.copy(glob.glob(os.path.join(SRC, "prof_head", "**", "*kernel_stats.csv"), recursive=True)[0],
            os.path.join(P, f"{tag}_head_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "pmc_summary.txt"), os.path.join(P, f"{tag}_pmc_summary.txt"))
shutil.copy(os.path.join(SRC, "sweep_gin.log"), os.path.join(P, f"{tag}_sweep_gin.log"))
shutil.copy(os.path.join(ROOT, "gpurun_out", "parity_errors.txt"), os.path.join(P, "parity_errors.txt"))
for f in ("bench", "bench_interp7", "bench_config2", "bench_under_rocprof"):
    shutil.copy(os.path.join(SRC, f + ".json"), os.path.join(P, f"{tag}_{f}.json"))
    d = json.loads(open(os.path.join(P, f"{tag}_{f}.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["steps"], d["warmup"], "launch", r["avg_launch_ms"], "frac", r["frac"],
          "traffic", r["traffic"])
    c = d["config"]
    for k in ("fwd_kernel_ms", "hbm_frac_of_8TBps_whole_step", "adaptive", "other_offset_regimes",
              "config2_forward_gpu", "fused_head"):
        if k in c:
            print("   ", k, json.dumps(c[k])[:1000])
    if "cpu_baseline" in d:
        print("    cpu", json.dumps(d["cpu_baseline"])[:300])
for f in (f"{tag}_kernel_stats.csv", f"{tag}_head_kernel_stats.csv"):
    print(f)
    for r in csv.DictReader(open(os.path.join(P, f))):
        n = r["Name"]
        if any(k in n for k in ("synth", "reach", "absmax", "finalize", "zero_if")):
            short = n.split("(anonymous namespace)::")[-1].split("(")[0]
            print("  %-66s calls %4s avg %9.1f us  min %8.1f" % (short[:66], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["MinNs"]) / 1e3))
d = json.load(open(os.path.join(P, "pmc_traffic.json")))
for k in ("synth_fwd", "synth_bwd", "head_fwd", "head_bwd"):
    print(k, d[k]["ratio"], d[k]["kernel"])
