import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
sd = ctypes.CDLL("tools/dbg/libstackdump.so"); sd.stackdump_install(2); sd.stackdump_install_crash()
import artemis_amd
from artemis_amd import _lib
dev = torch.device("cuda:0")
N,H,W=1,64,128
inp=torch.rand(N,3,H+4,W+4,device=dev); w=torch.rand(N,25,H,W,device=dev); oy=torch.randn(N,25,H,W,device=dev); ox=torch.randn(N,25,H,W,device=dev)
print("fixed", float(artemis_amd.synth(inp,w,oy,ox).sum()), flush=True)
with artemis_amd.synth_options(tile_radius=-1):
    print("r14", float(artemis_amd.synth(inp,w,oy,ox).sum()), flush=True)
with artemis_amd.synth_options(tile_radius=-2):
    print("auto", float(artemis_amd.synth(inp,w,oy,ox).sum()), flush=True)
print(artemis_amd.adaptive_state(H,W))
