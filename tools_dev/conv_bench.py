"""Micro-benchmark of fpc_conv2d (k_conv_igemm) per shape and tiling on the GPU box.
    python tools_dev/conv_bench.py [--iters 50]
Prints us per call and achieved f32 TFLOP/s (2*M*N*K) for each (shape, bm, bn, nsplit)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--shapes", default="all")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = nat.lib()

SHAPES = {
    # name: (B, Cin, Hi, Wi, Cout, k, stride, pad)
    "s2.0x4": (4, 256, 120, 160, 128, 3, 1, 1),
    "s3.0x4": (4, 256, 60, 80, 128, 3, 1, 1),
    "l1": (1, 64, 120, 160, 64, 3, 1, 1),
    "l2": (1, 128, 60, 80, 128, 3, 1, 1),
    "l3": (1, 256, 30, 40, 256, 3, 1, 1),
    "l4": (1, 512, 15, 20, 512, 3, 1, 1),
    "p2lat x4": (4, 64, 120, 160, 256, 1, 1, 0),
}
CONFIGS = [(64, 64, 1001), (64, 128, 1001), (128, 128, 1001), (64, 64, 1004), (0, 0, -1), (0, 0, -3), (0, 0, -2), (0, 0, -4), (0, 0, 0), (64, 64, 1), (64, 128, 1), (128, 64, 1), (128, 128, 1), (64, 64, 2), (64, 64, 4), (64, 64, 8),
           (64, 128, 2), (64, 128, 4), (128, 128, 2)]

for name, (B, Cin, Hi, Wi, Cout, k, stride, pad) in SHAPES.items():
    if a.shapes != "all" and name not in a.shapes.split(","):
        continue
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    x = torch.randn((B, Hi, Wi, Cin), device=dev)
    w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
    out = torch.empty((B, Ho, Wo, Cout), device=dev)
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Ho, Wo, Cin, Cout, k, k), dtype=torch.uint8, device=dev)
    sb, sh, sw, sc = x.stride()
    flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    st = torch.cuda.current_stream().cuda_stream
    for (bm, bn, ns) in CONFIGS:
        if Cout <= 64 and bn == 128:
            continue
        plan = (ctypes.c_int * 4)()
        L.fpc_conv2d_plan(B, Ho, Wo, Cin, Cout, k, k, bm, bn, ns, plan)
        if 0 < ns < 1000 and plan[2] != ns:
            continue
        if ns >= 1000 and (Cin % 32 or plan[2] != ns - 1000):
            continue
        if ns < 0 and (k != 3 or Cin % 8 or Cout % 64):
            continue

        def call():
            nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(),
                                   None, B, Hi, Wi, Cin, Cout, k, k, stride, pad, 0, bm, bn, ns, ws.data_ptr(),
                                   ws.numel(), st), "conv")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        # the call includes the weight re-pack kernel: subtract nothing, report as is plus plan
        print(f"{name:10s} M={B*Ho*Wo:6d} N={Cout:4d} K={Cin*k*k:5d} plan={tuple(plan)} {us:8.1f} us  {flops/us/1e6:7.1f} TF (incl. pack)")
