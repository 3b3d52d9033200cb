"""Tilings the autotuner picks per convolution site under objective 0 (latency) and 1 (latency x sqrt(chip share)).
out5 = (bm, bn, nsplit or -winograd form, Cout, K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config
from fastposecnn_amd.engine import NetEngine
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
torch.manual_seed(0)
m = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
plans = {}
for mode in (0, 1, 2):
    e = NetEngine(m, 1, 480, 640, dev, autotune=True, tune_mode=mode, graph=True, split_precision=True)
    plans[mode] = e.conv_plans()
names = e._names
for i, (a, b, c) in enumerate(zip(plans[0], plans[1], plans[2])):
    flag = "" if a == b == c else "   <-- differs"
    print(f"site {i:2d}: latency {a[:3]}  sqrt-share {b[:3]}  share {c[:3]}  (Cout {a[3]}, K {a[4]}){flag}")
