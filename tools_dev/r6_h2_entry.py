import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fastposecnn_amd import _native as nat
dev = torch.device("cuda:0"); L = nat.lib()
B, Cin, Hi, Wi, Cout = 32, 256, 120, 160, 128
x = torch.randn((B, Hi, Wi, Cin), device=dev); w = torch.randn((Cout, Cin, 3, 3), device=dev) * 0.05
out = torch.empty((B, Hi, Wi, Cout), device=dev)
ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Hi, Wi, Cin, Cout, 3, 3), dtype=torch.uint8, device=dev)
nblk = (-(-(Wi // 2) // 8) * (-(-(Hi // 2) // 8))) * B * (Cout // 64)
dbg = torch.zeros((nblk, 4, 8), dtype=torch.int64, device=dev)
sb, sh, sw, sc = x.stride(); st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, None, None, None, out.data_ptr(), dbg.data_ptr(), B, Hi, Wi, Cin, Cout, 3, 3, 1, 1, 77, 0, 0, -8, ws.data_ptr(), ws.numel(), st), "conv")
torch.cuda.synchronize()
d = dbg.cpu().double()
print("entry: set-up + issue %.0f | first operands land %.0f | barrier %.0f | -> loop (total entry %.0f) | K loop %.0f | exit incl. store drain %.0f" % (
    d[:, :, 0].mean(), d[:, :, 1].mean(), d[:, :, 2].mean(), d[:, :, 6].mean(), d[:, :, 3].mean(), d[:, :, 7].mean()))
