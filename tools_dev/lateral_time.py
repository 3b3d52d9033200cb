"""FPN lateral shapes through fpc_conv2d (one decoder = one group): k_conv_igemm forms against k_lateral1x1 (nsplit 2000 + parts).
    python tools_dev/lateral_time.py [B]
Each call includes the weight pack kernels (a few us); times are HIP events around 10 calls / 10."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
Bs = [int(sys.argv[1])] if len(sys.argv) > 1 else [1, 32]
for B in Bs:
    for (Cin, Hi, Wi) in ((64, 120, 160), (128, 60, 80)):
        Cout = 256
        x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, 1, 1), device=dev) * 0.05
        bias = torch.randn(Cout, device=dev); up = torch.randn((B, Hi // 2, Wi // 2, Cout), device=dev)
        out = torch.empty((B, Hi, Wi, Cout), device=dev)
        ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Hi, Wi, Cin, Cout, 1, 1), dtype=torch.uint8, device=dev)
        sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
        ref = None
        for ns in (0, 1001, 2001, 2002, 2004, 2008):
            def call():
                nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, bias.data_ptr(), None, up.data_ptr(), out.data_ptr(), None,
                                       B, Hi, Wi, Cin, Cout, 1, 1, 1, 0, 0, 64 if ns == 1001 else 0, 64 if ns == 1001 else 0, ns,
                                       ws.data_ptr(), ws.numel(), st), "conv")
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 100
            if ref is None: ref = out.clone()
            err = (out - ref).abs().max().item()
            gb = (out.numel() * 4 + x.numel() * 4 + up.numel() * 4) / 1e9
            print(f"B={B} Cin={Cin} {Hi}x{Wi} nsplit={ns}: {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s  max|diff to first| {err:.2e}")
