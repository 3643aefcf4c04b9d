#!/usr/bin/env python3
"""Build (on CPU: `python tools/exp/run_exp.py build`) and time (on GPU) the experiment kernels."""
import ctypes
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "artemis_amd", "lib", "exp")


def build(name="exp_fwd"):
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-munsafe-fp-atomics",
                    "-o", so, os.path.join(HERE, name + ".hip")], check=True)
    return so


def main():
    name = os.environ.get("EXP", "exp_fwd")
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build(sys.argv[2] if len(sys.argv) > 2 else name))
        return
    import torch
    variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (list(range(12)) + [100, 101, 102] if name == "exp_fwd" else list(range(23)))
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    lib = ctypes.CDLL(os.path.join(OUT, f"lib{name}.so"))
    lib.exp_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    N, H, W = 1, int(os.environ.get('EXP_H', 1080)), int(os.environ.get('EXP_W', 1920))
    torch.manual_seed(0)
    sets = []
    for _ in range(4):
        inp = torch.nn.functional.pad(torch.rand(N, 3, H, W, device=dev), (2, 2, 2, 2), mode="replicate").contiguous()
        w = torch.softmax(torch.randn(N, 25, H, W, device=dev), 1)
        oy = torch.randn(N, 25, H, W, device=dev) * sigma
        ox = torch.randn(N, 25, H, W, device=dev) * sigma
        out = torch.empty(N, 3, H, W, device=dev)
        sets.append((inp, w, oy, ox, out))
    st = torch.cuda.current_stream().cuda_stream
    times = {v: [] for v in variants}
    for rnd in range(9):
        for v in variants:
            inp, w, oy, ox, out = sets[rnd % 4]
            if v >= 100:
                oy = oy.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.exp_run(v, inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), out.data_ptr(), N, H, W, st)
            e1.record()
            e1.synchronize()
            assert rc == 0, (v, rc)
            if rnd:
                times[v].append(e0.elapsed_time(e1))
    px = N * H * W
    for v in variants:
        med = statistics.median(times[v])
        nbytes = {100: 200.0 * px, 101: 100.0 * px, 102: 100.0 * px}.get(v, 324.0 * px)
        print(f"variant {v:3d}: {med*1e3:8.1f} us (min {min(times[v])*1e3:8.1f})  {nbytes/med/1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
