"""Tight loop over CC + aggregation on the bench fixture (profiling aid): python tools_dev/post_loop.py [--frames 1] [--iters 200]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth
import aggregation_layer as al

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(a.frames))[0].items()}
layer = al.AggregationLayer(None, 7)
cm = al.attach_fg_bits(cat["mask"].to(torch.int64).contiguous())      # as the class compression hands it over
for _ in range(10):
    labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)
    layer._aggregate(cat, cm, labels, 6 * a.frames, n_dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)
    layer._aggregate(cat, cm, labels, 6 * a.frames, n_dev)
e1.record(); torch.cuda.synchronize()
print(f"frames={a.frames} cc+agg per call {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us")
