"""Which (instance, hypothesis) counts differ from the oracle on the small golden (dev aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd.lib
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
from oracle import oracle as orc
orc.build()
g = dict(np.load("tests/golden/vote_small.npz"))
dev = "cuda"
xy = torch.from_numpy(g["xy"]).to(dev)
vertex = xy.permute(0, 2, 3, 1).unsqueeze(3)
out, dbg = rvg.ransac_voting_layer_v3(torch.from_numpy(g["mask"]).to(dev), vertex, int(g["hn"]), idxs=torch.from_numpy(g["idxs"]).to(dev), return_debug=True)
want, wdbg = orc.ransac_voting_layer_v3(g["mask"], g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], int(g["hn"]), idxs=g["idxs"], return_debug=True)
c, wc = dbg[0]["counts"].cpu().numpy(), wdbg[0]["counts"]
print("tn", wdbg[0]["tn"], "win", dbg[0]["win_idx"].cpu().numpy(), wdbg[0]["win_idx"])
bad = np.argwhere(c != wc)
print("differing", len(bad))
for i, h in bad[:20]:
    print(i, h, "native", c[i, h], "oracle", wc[i, h], "hyp", wdbg[0]["hyp"][i, h])
