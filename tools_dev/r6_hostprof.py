import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.streaming import FrameStreamer
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
x = synth.make_image(0)[None].to(dev)
cat_cpu, _ = synth.make_vote_batch(range(1))
cat = {k: v.to(dev) for k, v in cat_cpu.items()}
st = FrameStreamer(model, net_streams=4)
for _ in range(12): st.collect(st.submit(x, categorical_override=cat))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for rep in range(40):
    tk = [st.submit(x, categorical_override=cat) for _ in range(5)]
    [st.collect(t) for t in tk]
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats(sys.argv[1] if len(sys.argv) > 1 else "cumulative"); ps.print_stats(45)
