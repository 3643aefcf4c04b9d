#!/usr/bin/env python3
"""exp_glds.hip: every variant must produce the sums variant 0 produces (the stand-ins are only
worth timing if the LDS-DMA path really moves the bytes)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "artemis_amd", "lib", "exp", "libexp_glds.so"))
lib.exp_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
H, W = 1080, 1920
torch.manual_seed(1)
w, oy, ox = (torch.randn(1, 25, H, W, device=dev) for _ in range(3))
st = torch.cuda.current_stream().cuda_stream
outs = {}
for v in (0, 1, 2, 3, 11, 12, 13):
    out = torch.zeros(1, 3, H, W, device=dev)
    assert lib.exp_run(v, None, w.data_ptr(), oy.data_ptr(), ox.data_ptr(), out.data_ptr(), 1, H, W, st) == 0
    torch.cuda.synchronize()
    outs[v] = out
ref = torch.stack([w.sum(1), oy.sum(1), ox.sum(1)], 1)
for v, o in outs.items():
    print(f"variant {v:2d}: max |out - variant 0| = {float((o - outs[0]).abs().max()):.3e}, "
          f"max |out - torch sum| = {float((o - ref).abs().max()):.3e}")
