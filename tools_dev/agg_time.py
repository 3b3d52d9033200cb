"""Device time of the aggregation (graph replay of 10 captured calls) at B = 1 and B = 32: python tools_dev/agg_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth, _native as nat
import aggregation_layer as al
dev = torch.device("cuda:0")
layer = al.AggregationLayer(None, 7)
for frames in (1, 32):
    cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(frames))[0].items()}
    cm = cat["mask"].to(torch.int64).contiguous()
    B, H, W = cm.shape
    n = 6 * frames
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(side):
        labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)
        layer._aggregate(cat, cm, labels, n, n_dev); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                keep = layer._aggregate(cat, cm, labels, n, n_dev)
    torch.cuda.current_stream().wait_stream(side)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        e0.record(); g.replay(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    ts.sort(); t = ts[len(ts) // 2]
    alg = B * 40 * H * W + n * 12 * H * W
    print(f"frames={frames} aggregate: {t:.1f} us per call; alg {alg / t / 1e3:.1f} GB/s = {alg / t / 1e3 / 8000:.3f} of 8 TB/s")
