#!/usr/bin/env python3
"""Compile the HIP source for gfx950 with -save-temps and print, per kernel, the register /
LDS / scratch budget and an instruction-class census of its ISA (developer tool; CPU only).

usage: python tools/kres.py [filter-substring] [--radius R | --host] [--flags "..."]
(default: the tile translation unit for radius 8; --host: synth_kernels.hip with the direct kernels)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "artemis_amd", "csrc")


def main():
    filt = ""
    extra = []
    radius = 8
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--flags":
            extra = args.pop(0).split()
        elif a == "--keep":
            pass
        elif a == "--radius":
            radius = int(args.pop(0))
        elif a == "--host":
            radius = None
        else:
            filt = a
    tmp = tempfile.mkdtemp(prefix="kres_")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-c", "-munsafe-fp-atomics", "-ffp-contract=on",
           "-I" + os.path.join(ROOT, "include"), "-save-temps", "-o", "k.o"] + \
        ([os.path.join(CSRC, "synth_kernels.hip")] if radius is None else
         [os.path.join(CSRC, "synth_tiles.hip"), f"-DSYNTH_TILE_R={radius}"]) + extra
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr)
        sys.exit(1)
    for line in r.stderr.splitlines():
        if "warning" in line or "error" in line:
            print(line)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    text = open(os.path.join(tmp, asm)).read()
    # split per function
    funcs = re.split(r"\n\t\.globl\t", text)
    rows = []
    for f in funcs[1:]:
        name = f.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "")
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if filt and filt not in dem:
            continue
        body = f
        def meta(key):
            m = re.search(r"\." + key + r"\s+(\d+)", body)
            return int(m.group(1)) if m else -1
        vg = meta("amdhsa_next_free_vgpr")
        sg = meta("amdhsa_next_free_sgpr")
        lds = meta("amdhsa_group_segment_fixed_size")
        scr = meta("amdhsa_private_segment_fixed_size")
        occ = re.search(r"; Occupancy: (\d+)", body)
        ops = collections.Counter()
        code = body.split(".amdhsa_kernel")[0]
        for ln in code.splitlines():
            ln = ln.strip()
            if not ln or ln.startswith((";", ".", "s_nop")) or ln.endswith(":"):
                continue
            op = ln.split()[0]
            if op.startswith("v_"):
                ops["valu"] += 1
            elif op.startswith("s_waitcnt"):
                ops["waitcnt"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                ops["branch"] += 1
            elif op.startswith("s_"):
                ops["salu"] += 1
            elif op.startswith("ds_"):
                ops[op] += 1
            elif op.startswith(("global_", "flat_", "scratch_", "buffer_")):
                ops[op] += 1
            else:
                ops["other"] += 1
        rows.append((dem, vg, sg, lds, scr, occ.group(1) if occ else "?", dict(ops)))
    for dem, vg, sg, lds, scr, occ, ops in rows:
        print(f"{dem}\n   vgpr={vg} sgpr={sg} lds={lds} scratch={scr} occ={occ}")
        print("   " + " ".join(f"{k}={v}" for k, v in sorted(ops.items())))
    print("asm:", os.path.join(tmp, asm))


if __name__ == "__main__":
    main()
