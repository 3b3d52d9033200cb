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
ap.add_argument("--prune", type=int, default=0)
ap.add_argument("--bits", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(a.frames))[0].items()}
agg = al.AggregationLayer(None, 7).forward(cat)
mask, xy = agg["instance_masks"], agg["xy"]
n, H, W = mask.shape
vertex = xy.permute(0, 2, 3, 1)
sn, sh, sw, sc = vertex.stride()
lib = nat.lib()
lib.fpc_vote_set_prune(a.prune, 0, None)
bits = al.mask_bits_of(mask) if a.bits else None
nbytes = lib.fpc_ransac_workspace_bytes(n, H, W, a.hn)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
out = torch.empty((n, 2), device=dev)
st = torch.cuda.current_stream().cuda_stream
names = ["scan", "plan", "count", "final"]
rows = []
for it in range(30):
    nat.check(lib.fpc_ransac_voting_v3_bits(mask.data_ptr(), bits.data_ptr() if bits is not None else None, vertex.data_ptr(), sn, sh, sw, sc, n, None, H, W, a.hn, None, None, it, 0.999, 5,
                                       30000, out.data_ptr(), None, None, None, None, None, None, None, ws.data_ptr(), ws.numel(), st), "vote")
    torch.cuda.synchronize()
    s = ws[nbytes - 1024:].view(torch.int64).cpu().numpy().reshape(4, 32)
    rows.append(s.copy())
s = np.median(np.stack(rows[10:]).astype(np.float64) - np.stack(rows[10:])[:, :1, :1], axis=0) * 0.01   # us since scan start
for k in range(4):
    v = [f"{x:7.2f}" for x in s[k] if x > 0 or k == 0][:8]
    print(f"{names[k]:6s}", " ".join(v))
raw = np.median(np.stack(rows[10:]).astype(np.float64), axis=0)
t0 = np.stack(rows[10:])[:, 0, 0].astype(np.float64)
rel = np.median((np.stack(rows[10:]).astype(np.float64) - t0[:, None, None]), axis=0) * 0.01
for ps in range(3):
    if raw[2][8 * ps] > 0:
        print(f"count pass {ps}: start {rel[2][8*ps]:7.2f} tables+first loads issued {rel[2][8*ps+1]:7.2f} first segment staged {rel[2][8*ps+2]:7.2f} "
              f"tiles done {rel[2][8*ps+3]:7.2f} flushed {rel[2][8*ps+4]:7.2f} end {rel[2][8*ps+5]:7.2f}  segments {raw[2][8*ps+6]:.0f} items {raw[2][8*ps+7]:.0f}")
    if ps > 0 and raw[1][8 + 8 * ps] > 0:
        print(f"lead {ps}: start {rel[1][8+8*ps]:7.2f} leader known {rel[1][8+8*ps+1]:7.2f} L known {rel[1][8+8*ps+2]:7.2f} compacted {rel[1][8+8*ps+3]:7.2f}")

if a.prune == 1:
    d = ws[nbytes - 1024 - 4 * 1024 * 4 * 8:nbytes - 1024].view(torch.int64).cpu().numpy().reshape(4, 1024, 4)
    for ps in range(3):
        t0, t1, ns, hw = d[ps, :, 0], d[ps, :, 1], d[ps, :, 2] & 0xffff, d[ps, :, 3]
        items = d[ps, :, 2] >> 16
        dur = (t1 - t0) * 0.01
        act = items > 0
        xcc = hw & 15
        cu = (hw >> 8 >> 8) & 15; se = (hw >> 8 >> 13) & 7        # HW_ID: cu_id bits 11:8, sh 12, se 15:13
        print(f"pass {ps}: active {act.sum()} dur mean {dur[act].mean():.1f} p10 {np.percentile(dur[act],10):.1f} p50 {np.percentile(dur[act],50):.1f} p90 {np.percentile(dur[act],90):.1f} max {dur[act].max():.1f}; start spread {(t0[act].max()-t0[act].min())*0.01:.1f}")
        print("   by segments:", {int(k): round(float(dur[act & (ns == k)].mean()), 1) for k in np.unique(ns[act])}, " counts", {int(k): int((act & (ns == k)).sum()) for k in np.unique(ns[act])})
        print("   by xcc:", {int(k): round(float(dur[act & (xcc == k)].mean()), 1) for k in np.unique(xcc[act])})
        order = np.argsort(dur); print("   slowest blocks", order[-8:], dur[order[-8:]].round(1), "fastest", order[:4], dur[order[:4]].round(1))
        # per-quarter of blockIdx
        print("   by block range:", [round(float(dur[i:i + 128][act[i:i + 128]].mean()), 1) for i in range(0, 1024, 128)])
