#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration launches (MI355X_MICROARCH.md, HBM section: calibrate on a
known byte count in your own access pattern): the pure-streaming experiment kernels read exactly
75 planes (forward-like) or read 75 + write 75 planes (backward-like) of 1080 x 1920 fp32 with the
product's dword-per-lane non-temporal accesses and tap-major order.

    rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_calib.py
"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "artemis_amd", "lib", "exp", "libexp_stream.so"))
lib.exp_runb.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
H, W = 1080, 1920
sets = [[torch.randn(1, 25, H, W, device=dev) for _ in range(6)] for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream
for rnd in range(3):
    for v in (37, 32):  # forward-like (reads 75 planes, writes 3), backward-like (reads 75, writes 75)
        s = sets[rnd % 3]
        assert lib.exp_runb(v, *[t.data_ptr() for t in s], H, W, st) == 0
torch.cuda.synchronize()
