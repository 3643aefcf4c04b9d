#!/usr/bin/env python3
import ctypes, os, statistics, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.environ.get("EXP_STREAM_LIB", os.path.join(ROOT, "artemis_amd", "lib", "exp", "libexp_stream.so")))
lib.exp_runb.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
H, W = 1080, 1920
sets = [[torch.randn(1, 25, H, W, device=dev) for _ in range(6)] for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [47, 67, 60, 61, 64, 65, 66, 46, 62, 63]
T = {v: [] for v in variants}
for rnd in range(9):
    for v in variants:
        s = sets[rnd % 3]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = lib.exp_runb(v, *[t.data_ptr() for t in s], H, W, st); e1.record(); e1.synchronize()
        assert rc == 0, (v, rc)
        if rnd: T[v].append(e0.elapsed_time(e1))
for v in variants:
    med = statistics.median(T[v])
    nb = 300.0 if v in (90, 91, 92) else 312.0 if v in (35,36,37,38,39,44,45,46,54,55,56,62,63,73,74,83,97,107,108,110) else 600.0
    if hasattr(lib, "exp_pace_timeouts") and v >= 200:
        lib.exp_runb(v, *[t.data_ptr() for t in sets[0]], H, W, st); torch.cuda.synchronize()
        print(f"   (pace waits that ran into their bound in one launch: {lib.exp_pace_timeouts()})")
    print(f"stream variant {v:2d}: {med*1e3:8.1f} us (min {min(T[v])*1e3:8.1f}) {nb*H*W/med/1e6:7.0f} GB/s")
