"""Round 6: the 128-channel split-precision Winograd form (nsplit = -6, wino128.hip) beside the 64-channel one (-5) on the
shapes of a config-3 forward: parity against each other, then a launch loop for rocprofv3 --kernel-trace --stats
(the fpc_conv2d call packs the weights first: read the k_conv_wino* rows of the kernel stats, not the wall time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
SHAPES = {      # name: (B, Cin, H, W, Cout)
    "s2.0": (32, 256, 120, 160, 128),
    "s3.0": (32, 256, 60, 80, 128),
    "s2.1": (32, 128, 120, 160, 128),
    "l2": (32, 128, 60, 80, 128),
    "l3": (32, 256, 30, 40, 256),
    "l4": (32, 512, 15, 20, 512),
    "l1": (32, 64, 120, 160, 64),
}
RES = int(os.environ.get("RES", "0"))      # 1: scale / shift / residual / ReLU epilogue (an encoder block's second convolution)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
forms = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-5, -6]
for name in names:
    B, Cin, H, W, Cout = SHAPES[name]
    torch.manual_seed(0)
    x = torch.randn((B, H, W, Cin), device=dev); w = torch.randn((Cout, Cin, 3, 3), device=dev) / (Cin * 9) ** 0.5
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, H, W, Cin, Cout, 3, 3), dtype=torch.uint8, device=dev)
    sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
    outs = {}
    res = torch.randn((B, H, W, Cout), device=dev) if RES else None
    scl = (torch.rand(Cout, device=dev) + 0.5) if RES else None
    sft = torch.randn(Cout, device=dev) if RES else None
    for ns in forms:
        out = torch.full((B, H, W, Cout), float("nan"), device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(iters + 1):
            if it == 1:
                e0.record()
            nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), scl.data_ptr() if RES else None, sft.data_ptr() if RES else None,
                                   res.data_ptr() if RES else None, None, out.data_ptr(), None, B, H, W,
                                   Cin, Cout, 3, 3, 1, 1, RES, 0, 0, ns, ws.data_ptr(), ws.numel(), st), "conv %d" % ns)
        e1.record(); torch.cuda.synchronize()
        outs[ns] = out
        print(f"{name} B{B} Cin{Cin} {H}x{W} Cout{Cout} form {ns}: {e0.elapsed_time(e1) / iters * 1e3:9.1f} us per call incl. weight pack", flush=True)
    if len(forms) == 2:
        a, b = outs[forms[0]], outs[forms[1]]
        print(f"   max |a - b| = {(a - b).abs().max().item():.3e} of max |a| = {a.abs().max().item():.3e}; nan: {torch.isnan(b).any().item()}", flush=True)
