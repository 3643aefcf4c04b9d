#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output directories: mean counter value per kernel launch.
usage: python tools/pmc_summary.py <dir-with-pass-subdirs>"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if "synth" not in k:
                    continue
                short = k.split("(")[0].replace("void (anonymous namespace)::", "").split("::")[-1]
                short = k[k.index("synth"):].split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")


if __name__ == "__main__":
    main()
