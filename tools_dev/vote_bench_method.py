"""bench.py's vote_roofline on the 32-frame hn=128 batch with different group sizes (is the gap to the launch loop a fixed cost?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
import aggregation_layer as al
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.HV_NUM_OF_HYPOTHESES = 128
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).to(dev).eval()
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(32))[0].items()}
cat["mask"] = al.attach_fg_bits(cat["mask"].to(torch.int64).contiguous())
for calls in (10, 30):
    for bits in (True, False):
        r = bench.vote_roofline(model, cat, 192, 7, "x", calls=calls, use_bits=bits)
        print(f"calls={calls} bits={bits}: {r['launch_ms']*1e3:.1f} us  frac {r['frac']:.4f}")
