"""Winograd forms through fpc_conv2d on the engine's shapes: python tools_dev/wino_time.py [B] [nsplit ...]   (default -2 -5: f32 / split precision, 8 waves)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FORMS = [int(v) for v in sys.argv[2:]] or [-2, -5]
for (Cin, Hi, Wi, Cout, res) in ((64, 120, 160, 64, True), (64, 120, 160, 64, False), (128, 60, 80, 128, True), (256, 30, 40, 256, True), (512, 15, 20, 512, True),
                                 (256, 120, 160, 128, False), (128, 60, 80, 128, False)):
    x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, 3, 3), device=dev) * 0.05
    sc = torch.rand(Cout, device=dev) + 0.5; sh = torch.randn(Cout, device=dev)
    r = torch.randn((B, Hi, Wi, Cout), device=dev) if res else None
    out = torch.empty((B, Hi, Wi, Cout), device=dev)
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Hi, Wi, Cin, Cout, 3, 3), dtype=torch.uint8, device=dev)
    sb, s_h, sw, s_c = x.stride(); st = torch.cuda.current_stream().cuda_stream
    ref = None
    for ns in FORMS:
        def call():
            nat.check(L.fpc_conv2d(x.data_ptr(), sb, s_h, sw, s_c, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), nat.ptr(r), None, out.data_ptr(), None,
                                   B, Hi, Wi, Cin, Cout, 3, 3, 1, 1, 1, 0, 0, ns, ws.data_ptr(), ws.numel(), st), "conv")
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        if ref is None: ref = out.clone()
        print(f"B={B} {Cin}->{Cout} {Hi}x{Wi} res={int(res)} nsplit={ns}: {us:8.1f} us (incl. ~20 us of weight packing)  equal to the first form: {bool(torch.equal(out, ref))}")
