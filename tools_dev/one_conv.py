"""Runs one fpc_conv2d configuration in a loop (for rocprofv3 --pmc).  python tools_dev/one_conv.py NSPLIT [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
ns = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0"); L = nat.lib()
B, Cin, Hi, Wi, Cout, k = 4, 256, 120, 160, 128, 3
x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
out = torch.empty((B, Hi, Wi, Cout), device=dev)
ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Hi, Wi, Cin, Cout, k, k), dtype=torch.uint8, device=dev)
sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
for _ in range(iters):
    nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(), None, B, Hi, Wi,
                           Cin, Cout, k, k, 1, 1, 0, 0, 0, ns, ws.data_ptr(), ws.numel(), st), "conv")
torch.cuda.synchronize()
