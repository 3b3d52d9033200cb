"""HIP-event timing of the hough-vote call alone (dev aid): python tools_dev/vote_time.py [B] [hn] [reps] [bits]
(`bits`: pass the aggregation layer's mask bit words, as the model's own pipeline does)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hn = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
use_bits = len(sys.argv) > 4 and sys.argv[4] == "bits"
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.HV_NUM_OF_HYPOTHESES = hn
model = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
cat_cpu, _ = synth.make_vote_batch(range(B))
cat = {k: v.to(dev) for k, v in cat_cpu.items()}
agg = model.aggregation_layer.forward(cat)
n = agg["instance_masks"].shape[0]
vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
import aggregation_layer as al
bits = al.mask_bits_of(agg["instance_masks"]) if use_bits else None
out, dbg = rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=1, return_debug=True, mask_bits=bits)
d = dbg[0]
print("n", n, "tn", d["tn"].tolist()[:12], "win_count", d["win_count"].tolist()[:12])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(reps):
    e0.record()
    rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=1, mask_bits=bits)
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
us = ts[len(ts) // 2]
alg = n * 12 * 480 * 640
print(json.dumps({"mask_bits": use_bits, "B": B, "hn": hn, "n": n, "vote_us_median": round(us, 2), "vote_us_min": round(ts[0], 2),
                  "achieved_GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000, 4)}))
