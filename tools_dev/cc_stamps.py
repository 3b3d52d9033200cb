"""Phase times of k_cca_image (image 0) from a -DFPC_STAMP_CC build: python tools_dev/cc_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth, _native as nat
import aggregation_layer as al
dev = torch.device("cuda:0")
lib = nat.lib()
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(1))[0].items()}
cm = cat["mask"].to(torch.int64).contiguous()
B, H, W = cm.shape
al.attach_fg_bits(cm); bits = al.fg_bits_of(cm)
labels = torch.empty((B, H, W), dtype=torch.int32, device=dev); n_dev = torch.empty(1, dtype=torch.int32, device=dev)
root = torch.empty(64, dtype=torch.int32, device=dev)
nb = lib.fpc_cc_workspace_bytes(B, H, W)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
HW = H * W
bstride = -(-HW // 4096) * 64
off = 0
def up(x, a=256): return (x + a - 1) // a * a
off = up(8 * B * bstride); off = up(off + 4 * B * (HW // 64)); rstride = up(HW // 2 + 1, 64); off = up(off + 4 * B * rstride)
rows = []
for it in range(20):
    nat.check(lib.fpc_cc_label_bits(bits.data_ptr(), B, H, W, labels.data_ptr(), n_dev.data_ptr(), root.data_ptr(), 64, ws.data_ptr(), nb,
                                    torch.cuda.current_stream().cuda_stream), "cc")
    torch.cuda.synchronize()
    rows.append(ws[off:off + 72].view(torch.int64).cpu().numpy().copy())
s = np.median(np.stack(rows[5:]).astype(np.float64) - np.stack(rows[5:])[:, :1], axis=0) * 0.01
print("k_cca_image phases (us since start): load %.2f starts+scan %.2f init %.2f (4 unused) %.2f unions %.2f flatten %.2f rank %.2f out %.2f" % tuple(s[1:9]))
