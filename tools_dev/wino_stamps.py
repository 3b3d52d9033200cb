"""Cycle stamps of the Winograd K loop (diagnostic hook of fpc_conv2d: relu == 77 routes gn_part to the stamp buffer)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else -1      # -1: 4-wave barrier form, -4: 8-wave all-DMA form
NWAVE = 8 if VAR in (-4, -2, -5) else 4      # (-6, -7: four waves)
B, Cin, Hi, Wi, Cout, k = 4, 256, 120, 160, 128, 3
if len(sys.argv) > 6:      # python tools_dev/wino_stamps.py VAR B Cin H W Cout
    B, Cin, Hi, Wi, Cout = (int(v) for v in sys.argv[2:7])
x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, k, k), device=dev) * 0.05
out = torch.empty((B, Hi, Wi, Cout), device=dev)
ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Hi, Wi, Cin, Cout, k, k), dtype=torch.uint8, device=dev)
nblk = (-(-(Wi // 2) // 8) * (-(-(Hi // 2) // (8 if VAR in (-4, -2, -5, -7, -8, -9) else 4)))) * B * (Cout // (128 if VAR == -6 else 64))
dbg = torch.zeros((nblk, NWAVE, 8), dtype=torch.int64, device=dev)
sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(), dbg.data_ptr(), B, Hi, Wi,
                           Cin, Cout, k, k, 1, 1, 77, 0, 0, VAR, ws.data_ptr(), ws.numel(), st), "conv")
torch.cuda.synchronize()
d = dbg.cpu().double()
clk = (d[:, :, 3] / d[:, :, 4] * 100.0)
print("in-kernel clock (s_memtime / s_memrealtime * 100 MHz): mean %.0f MHz, min %.0f, max %.0f" % (clk.mean(), clk.min(), clk.max()))
print("K loop: %.0f shader ticks = %.1f us per workgroup" % (d[:, :, 3].mean(), (d[:, :, 4].mean() / 100.0)))
per = d[:, :, :3] / d[:, :, 5:6]
names = ["issue loads", "wait + barrier", "MFMA block"] if VAR != -4 else ["first half (16 MFMA)", "vmcnt + barrier", "issue + second half"]
print("cycles per K-step (s_memtime ticks), mean over waves / median / p90:")
for i, n in enumerate(names):
    v = per[:, :, i].flatten()
    print(f"  {n:16s} {v.mean():8.0f} {v.median():8.0f} {v.quantile(0.9):8.0f}")
print("  total            %8.0f" % per.sum(-1).mean())
print("  whole K loop per step: %.0f" % (d[:, :, 3] / d[:, :, 5]).mean())
print("kernel entry -> K loop: %.0f ticks; K loop end -> last store acknowledged: %.0f ticks (mean over waves; p90 %.0f)" %
      (d[:, :, 6].mean(), d[:, :, 7].mean(), d[:, :, 7].flatten().quantile(0.9)))
print("per wave (mean over workgroups): issue | wait + barrier | MFMA block")
for w in range(NWAVE):
    print("  wave %d: %6.0f %6.0f %6.0f" % (w, per[:, w, 0].mean(), per[:, w, 1].mean(), per[:, w, 2].mean()))
