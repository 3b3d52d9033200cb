"""Phase stamps of k_conv_igemm (diagnostic build: -DFPC_STAMP_IGEMM).
    python -c "from fastposecnn_amd import build; build.build(force=True, extra=['-DFPC_STAMP_IGEMM'])"
    python tools_dev/igemm_stamps.py            # on the GPU box, then rebuild without the flag
Per wave: kernel entry -> first barrier (address set-up + first operands), K loop, epilogue (incl. the last store)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
CASES = [("stem 7x7/2 on NHWC4 M=76800 N=64 K=196 (MODE 2 form)", 1, 4, 480, 640, 64, 7, [(64, 64, 1)]),
         ("stem-like 3x3 Cin=32 M=76800 N=64 K=288", 1, 32, 240, 320, 64, 3, [(64, 64, 1)]),
         ("l3 3x3 M=1200 N=256 K=2304", 1, 256, 30, 40, 256, 3, [(64, 128, 6), (64, 64, 3), (64, 64, 8), (64, 64, 12), (64, 64, 1006), (64, 64, 1012)]),
         ("l2 3x3 M=4800 N=128 K=1152", 1, 128, 60, 80, 128, 3, [(64, 128, 3), (64, 64, 1), (64, 64, 4), (64, 64, 1003), (64, 64, 1001), (64, 64, 1006)]),
         ("l4 3x3 M=300 N=512 K=4608", 1, 512, 15, 20, 512, 3, [(64, 64, 8), (64, 64, 16), (64, 128, 12), (64, 64, 1012), (64, 64, 1024)]),
         ("p2 lateral x4 M=76800 N=256 K=64", 4, 64, 120, 160, 256, 1, [(64, 64, 1), (128, 128, 1)])]
for name, B, Cin, Hi, Wi, Cout, k, plans in CASES:
    x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
    Ho, Wo = ((Hi + 2 * (k // 2) - k) // 2 + 1, (Wi + 2 * (k // 2) - k) // 2 + 1) if k == 7 else (Hi, Wi)
    out = torch.empty((B, Ho, Wo, Cout), device=dev)
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Ho, Wo, Cin, Cout, k, k), dtype=torch.uint8, device=dev)
    sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
    for bm, bn, ns in plans:
        plan = (ctypes.c_int * 4)(); L.fpc_conv2d_plan(B, Ho, Wo, Cin, Cout, k, k, bm, bn, ns, plan)
        nblk = plan[3] * B * ((Cout + bn - 1) // bn) * plan[2] if False else 0
        dbg = torch.zeros((1 << 16, 4, 4), dtype=torch.int64, device=dev)
        def call(d):
            nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(), d, B, Hi, Wi,
                                   Cin, Cout, k, k, (2 if k == 7 else 1), k // 2, 77 if d else 0, bm, bn, ns, ws.data_ptr(), ws.numel(), st), "conv")
        for _ in range(3): call(dbg.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call(dbg.data_ptr())
        e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().double()
        live = d[:, :, 3].sum(1) > 0
        d = d[live]
        steps = d[:, :, 3].mean()
        print(f"{name:34s} plan {tuple(plan)}: {e0.elapsed_time(e1)/20*1e3:7.1f} us/call (incl. pack+epilogue kernels) | {int(live.sum())} WGs, {steps:.0f} K-steps: "
              f"prologue {d[:, :, 0].mean():7.0f}  K loop {d[:, :, 1].mean():7.0f} ({d[:, :, 1].mean()/steps:5.0f}/step)  epilogue {d[:, :, 2].mean():6.0f} ticks "
              f"= {(d[:, :, :3].sum(2).mean())/2400:.1f} us per workgroup")
