#!/usr/bin/env python3
import ctypes, os, statistics
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "artemis_amd", "lib", "exp", "libexp_gran.so"))
lib.gran_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
nbytes = 2 * 1024**3
src = torch.rand(nbytes // 4, device=dev)
dst = torch.empty(nbytes // 4, device=dev)
blocks = 2048  # 8192 waves
total = 1 << 30  # bytes touched per launch per direction
for mode, mname in ((0, "read"), (1, "write"), (2, "copy")):
    for gdw in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        waves = blocks * 4
        iters = max(1, total // (waves * gdw * 256))
        ts = []
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = lib.gran_run(gdw, mode, src.data_ptr(), dst.data_ptr(), nbytes, blocks, iters, st); e1.record(); e1.synchronize()
            assert rc == 0
            if i: ts.append(e0.elapsed_time(e1))
        med = statistics.median(ts)
        moved = waves * iters * gdw * 256 * (2 if mode == 2 else 1)
        print(f"{mname:5s} chunk {gdw*256:6d} B: {med*1e3:8.1f} us {moved/med/1e6:7.0f} GB/s")
