"""Autotuned engine forwards in a loop, for rocprofv3 (kernel trace or --pmc passes): python tools_dev/net_loop.py ENCODER B [iters]
The forwards after the tuning pass are identical; each starts with k_nchw3_to_nhwc4 (tools_dev/conv_pmc.py cuts one out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
enc, B = sys.argv[1], int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.BACKBONE_ARCH = "FPN"; hp.ENCODER = enc; hp.ENGINE_GRAPH = False
torch.manual_seed(0)
m = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
x = torch.stack([synth.make_image(i) for i in range(B)]).to(dev)
with torch.no_grad():
    for _ in range(2 + iters):
        lg = m.pure_model_forward(x); m.class_compression(lg)
torch.cuda.synchronize()
