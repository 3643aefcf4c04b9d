#!/usr/bin/env python3
import ctypes, os, statistics, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "artemis_amd", "lib", "exp", "libexp_bw.so"))
lib.bw_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
names = {0: "read f4 x4", 1: "read f4 x8", 2: "read f1 x8", 3: "read f1 x16", 4: "copy f4 x4", 5: "copy f4 x8", 6: "write f4", 7: "copy f4 ntS", 8: "copy f4 ntLS", 9: "copy f1x8", 10: "copy f1 ntS", 11: "copy f1 ntLS", 12: "write f1", 13: "write f1 nt", 14: "chunk f1x8", 15: "chunk f1 ntLS"}
for mb in (621,):
    nbytes = mb * 1000 * 1000 // 16 * 16
    bufs = [(torch.rand(nbytes // 4, device=dev), torch.empty(nbytes // 4, device=dev)) for _ in range(2 if mb > 1000 else 3)]
    for blocks in (1024, 2048, 4096, 16384, 65536):
        for v in (2, 9, 11, 12, 14, 15):
            ts = []
            for i in range(7):
                s, d = bufs[i % len(bufs)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rc = lib.bw_run(v, s.data_ptr(), d.data_ptr(), nbytes, blocks, st); e1.record(); e1.synchronize()
                assert rc == 0
                if i: ts.append(e0.elapsed_time(e1))
            med = statistics.median(ts)
            moved = nbytes * (2 if v in (4, 5, 7, 8, 9, 10, 11, 14, 15) else 1)
            print(f"{mb:5d} MB blocks={blocks:5d} {names[v]:12s} {med*1e3:8.1f} us {moved/med/1e6:7.0f} GB/s")
    del bufs
