import ctypes, sys, os
sys.path.insert(0, '/root/repo')
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
torch.manual_seed(0)
for (B, Cin, Hi, Wi, Cout, k, stride, pad, bm, bn, ns) in [(1, 64, 24, 40, 128, 3, 1, 1, 64, 64, 1), (2, 128, 17, 23, 96, 3, 2, 1, 64, 128, 2), (1, 256, 30, 40, 256, 1, 1, 0, 128, 128, 1), (1, 64, 120, 160, 256, 1, 1, 0, 64, 64, 1)]:
    Ho = (Hi + 2 * pad - k) // stride + 1; Wo = (Wi + 2 * pad - k) // stride + 1
    x = torch.randn((B, Hi, Wi, Cin), device=dev).abs_(); w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Ho, Wo, Cin, Cout, k, k), dtype=torch.uint8, device=dev)
    sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
    res = {}
    for tag, nse in (("f32", ns), ("bf16x3", 1000 + ns)):
        out = torch.zeros((B, Ho, Wo, Cout), device=dev)
        nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(), None, B, Hi, Wi,
                               Cin, Cout, k, k, stride, pad, 0, bm, bn, nse, ws.data_ptr(), ws.numel(), st), "conv")
        torch.cuda.synchronize()
        e = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        res[tag] = e
    print(f"B{B} Cin{Cin} {Hi}x{Wi} Cout{Cout} k{k}s{stride} tile {bm}x{bn} split {ns}: max err / max|ref|  f32 {res['f32']:.2e}  bf16x3 {res['bf16x3']:.2e}")
