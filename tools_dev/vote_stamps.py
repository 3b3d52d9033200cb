"""Phase timeline of the vote's four kernels (workgroup 0 of each) from a -DFPC_STAMP_VOTE build:
    python -c "from fastposecnn_amd import build; build.build(extra=['-DFPC_STAMP_VOTE'])"; python tools_dev/vote_stamps.py [--hn 1000] [--frames 1]
s_memrealtime ticks at 100 MHz: 10 ns resolution, one clock for all kernels."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth, _native as nat
import aggregation_layer as al

ap = argparse.ArgumentParser()
ap.add_argument("--hn", type=int, default=1000)
ap.add_argument("--frames", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(a.frames))[0].items()}
agg = al.AggregationLayer(None, 7).forward(cat)
mask, xy = agg["instance_masks"], agg["xy"]
n, H, W = mask.shape
vertex = xy.permute(0, 2, 3, 1)
sn, sh, sw, sc = vertex.stride()
lib = nat.lib()
nbytes = lib.fpc_ransac_workspace_bytes(n, H, W, a.hn)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
out = torch.empty((n, 2), device=dev)
st = torch.cuda.current_stream().cuda_stream
names = ["scan", "plan", "count", "final"]
rows = []
for it in range(30):
    nat.check(lib.fpc_ransac_voting_v3(mask.data_ptr(), vertex.data_ptr(), sn, sh, sw, sc, n, None, H, W, a.hn, None, None, it, 0.999, 5,
                                       30000, out.data_ptr(), None, None, None, None, None, None, None, ws.data_ptr(), ws.numel(), st), "vote")
    torch.cuda.synchronize()
    s = ws[nbytes - 1024:].view(torch.int64).cpu().numpy().reshape(4, 32)
    rows.append(s.copy())
s = np.median(np.stack(rows[10:]).astype(np.float64) - np.stack(rows[10:])[:, :1, :1], axis=0) * 0.01   # us since scan start
for k in range(4):
    v = [f"{x:7.2f}" for x in s[k] if x > 0 or k == 0][:8]
    print(f"{names[k]:6s}", " ".join(v))
