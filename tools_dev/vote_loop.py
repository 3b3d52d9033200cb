"""Tight loop over fpc_ransac_voting_v3 (and optionally cc/agg) on the bench fixture: steady-state kernel
times without Python gaps.  python tools_dev/vote_loop.py [--hn 1000] [--iters 300] [--frames 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth, _native as nat
import aggregation_layer as al

ap = argparse.ArgumentParser()
ap.add_argument("--hn", type=int, default=1000)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--frames", type=int, default=1)
ap.add_argument("--agg", action="store_true")
ap.add_argument("--bits", action="store_true", help="pass the aggregation layer's mask bit words (the scan skips the f32 planes)")
ap.add_argument("--sets", type=int, default=1, help="distinct copies of the inputs, used in rotation (cold reads, as in bench.py)")
ap.add_argument("--prune", type=int, default=0, help="fpc_vote_set_prune mode: 0 never, 1 always")
ap.add_argument("--cum", type=str, default="", help="pass schedule in 16ths, e.g. 5,10")
ap.add_argument("--info", action="store_true", help="print the survivor histogram of the last call")
a = ap.parse_args()
dev = torch.device("cuda:0")
cat_cpu, _ = synth.make_vote_batch(range(a.frames))
cat = {k: v.to(dev) for k, v in cat_cpu.items()}
layer = al.AggregationLayer(None, 7)
agg = layer.forward(cat)
mask = agg["instance_masks"]; xy = agg["xy"]
n, H, W = mask.shape
vertex = xy.permute(0, 2, 3, 1)
sn, sh, sw, sc = vertex.stride()
lib = nat.lib()
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
rvg.set_vote_prune(a.prune, tuple(int(c) for c in a.cum.split(",")) if a.cum else None)
ws = torch.empty(lib.fpc_ransac_workspace_bytes(n, H, W, a.hn), dtype=torch.uint8, device=dev)
out = torch.empty((n, 2), device=dev)
st = torch.cuda.current_stream().cuda_stream

bits = al.mask_bits_of(mask) if a.bits else None
bptr = bits.data_ptr() if bits is not None else None
sets = [(mask, vertex)] + [(mask.clone(), xy.clone().permute(0, 2, 3, 1)) for _ in range(a.sets - 1)]

def vote(seed):
    m, v = sets[seed % len(sets)]
    nat.check(lib.fpc_ransac_voting_v3_bits(m.data_ptr(), bptr, v.data_ptr(), sn, sh, sw, sc, n, None, H, W, a.hn, None, None,
                                       seed, 0.999, 5, 30000, out.data_ptr(), None, None, None, None, None, None, None,
                                       ws.data_ptr(), ws.numel(), st), "vote")

for i in range(20):
    vote(i)
    if a.agg: layer.forward(cat)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for i in range(a.iters):
    vote(100 + i)
    if a.agg: layer.forward(cat)
e1.record(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms = e0.elapsed_time(e1) / a.iters
print(f"n={n} hn={a.hn} sets={a.sets} bits={int(a.bits)} per-call {ms*1e3:.1f} us (wall {dt/a.iters*1e6:.1f} us)  alg {n*12*H*W/ms/1e6:.1f} GB/s  out0={out[0].tolist()}")
if a.info:
    import numpy as np
    info = torch.empty((n, 8), dtype=torch.int32, device=dev)
    nat.check(lib.fpc_vote_prune_info(ws.data_ptr(), ws.numel(), n, H, W, a.hn, info.data_ptr(), st), "info")
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    print("alive entering the last pass: mean %.0f min %d max %d of %d; units mean %.1f" % (inf[:, 2].mean(), inf[:, 2].min(), inf[:, 2].max(), a.hn, inf[:, 1].mean()))
