"""Where the time of the host-fed pipeline goes at batch 1 (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.streaming import FrameStreamer
from fastposecnn_amd.tools.dataset import FrameUploader, preprocess_frames
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
model = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
cat_cpu, _ = synth.make_vote_batch(range(1)); cat = {k: v.to(dev) for k, v in cat_cpu.items()}
x = synth.make_image(0)[None].to(dev)
st = FrameStreamer(model, net_streams=4)
st.prepare(x, categorical_override=cat)
depth = 4
up = FrameUploader(1, 480, 640, device=dev, slots=depth + 2)
frames = np.random.default_rng(0).integers(0, 256, (1, 480, 640, 3), dtype=np.uint8)
cur = torch.cuda.current_stream(dev)
pending = []
T = {"upload": 0.0, "wait": 0.0, "submit": 0.0, "collect": 0.0}
def step(timed):
    t0 = time.perf_counter(); t, ready = up.upload(frames)
    t1 = time.perf_counter()
    t2 = time.perf_counter(); pending.append(st.submit(t, categorical_override=cat, ready=ready))
    t3 = time.perf_counter()
    if len(pending) > depth: st.collect(pending.pop(0))
    t4 = time.perf_counter()
    if timed:
        T["upload"] += t1 - t0; T["wait"] += t2 - t1; T["submit"] += t3 - t2; T["collect"] += t4 - t3
for _ in range(12): step(False)
torch.cuda.synchronize(); n = 100; t0 = time.perf_counter()
for _ in range(n): step(True)
while pending: st.collect(pending.pop(0))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print({k: round(v / n * 1e6, 1) for k, v in T.items()}, "us per step; total", round(dt / n * 1e6, 1), "us ->", round(n / dt, 1), "img/s")
