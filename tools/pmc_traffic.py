#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh) into
profiles/pmc_traffic.json: HBM bytes per launch of the synth kernels, corrected as
MI355X_MICROARCH.md (HBM section) prescribes -- FETCH_SIZE counts KiB and, on gfx950, half of the
streamed bytes; the factor is calibrated on a kernel of known traffic in the product's own access
pattern (tools/pmc_calib.py under the same two passes).

usage: python tools/pmc_traffic.py <pmc dir of prof_run.py> <pmc dir of pmc_calib.py> [note] [shape]
(shape: the kernel shape prof_run.py ran -- "lean" for --radius -3, the adaptive default's choice at
sigma 1; "14" for --radius -1 -- bench.py quotes the record only for launches of that shape)
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = 1080, 1920


def means(root, needle):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if needle not in k:
                    continue
                short = k[k.index(needle):].split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    run_dir, calib_dir = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    shape = sys.argv[4] if len(sys.argv) > 4 else "lean"
    import bench
    plane_kib = 75 * H * W * 4 / 1024.0
    cal = means(calib_dir, "stream")
    # forward-like stand-in reads exactly 75 planes, backward-like reads 75 and writes 75
    reads = [v["FETCH_SIZE"] for v in cal.values() if "FETCH_SIZE" in v]
    writes = [v["WRITE_SIZE"] for v in cal.values() if v.get("WRITE_SIZE", 0) > plane_kib / 2]
    fetch_corr = plane_kib / (sum(reads) / len(reads)) if reads else 2.0
    write_corr = plane_kib / (sum(writes) / len(writes)) if writes else 1.0
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_passes.sh), means "
                  "over the launches of tools/prof_run.py (4 rotating 1080p operand sets, sigma 1, "
                  "kernel shape '" + shape + "'), MI355X. " + note,
        "source_hash": bench.source_hash(),
        "shape": shape,
        "units": "FETCH_SIZE/WRITE_SIZE count KiB; gfx950 FETCH_SIZE reads 1/2 of streamed bytes "
                 "(MI355X_MICROARCH.md, HBM section)",
        "calibration": {
            "kernel": "tools/exp/exp_stream.hip stand-ins (exactly 75 planes of 1080x1920 fp32 read / "
                      "read + written with the product's dword nt accesses)",
            "known_KiB": plane_kib,
            "fetch_correction": round(fetch_corr, 4),
            "write_correction": round(write_corr, 4),
            "kernels": {k: v for k, v in cal.items()},
        },
    }
    alg = {"fwd": bench.FWD_BYTES * H * W, "bwd": bench.BWD_BYTES * H * W,
           "head_fwd": (4 * (300 + 12.07 + 4) + 12) * H * W, "head_bwd": (4 * (600 + 12.07 + 8) + 12) * H * W}
    for name, v in sorted(means(run_dir, "synth_").items()):
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        head = ", true, float" in name or "true, true, float" in name
        is_head = ("fwd_tapmajor" in name and ", true, float" in name) or \
                  ("bwd_tapmajor" in name and name.split("float")[0].rstrip(", ").endswith("true"))
        if "fwd_tapmajor" in name:
            key = "head_fwd" if is_head else "synth_fwd"
        elif "bwd_tapmajor" in name:
            key = "head_bwd" if is_head else "synth_bwd"
        else:
            key = name.split("<")[0]
        nbytes = (v["FETCH_SIZE"] * fetch_corr + v["WRITE_SIZE"] * write_corr) * 1024
        out[key] = {"kernel": name, "FETCH_SIZE": round(v["FETCH_SIZE"], 1),
                    "WRITE_SIZE": round(v["WRITE_SIZE"], 1), "bytes_per_launch": int(nbytes)}
        a = alg.get(key.replace("synth_", ""))
        if a:
            out[key]["algorithmic_bytes_per_launch"] = int(a)
            out[key]["ratio"] = round(nbytes / a, 4)
        out[key + "_bytes_per_launch"] = int(nbytes)
        del head
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("_per_launch") or k == "calibration"},
                     indent=1)[:2000])


if __name__ == "__main__":
    main()
