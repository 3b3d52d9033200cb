"""Where the host thread's time goes in the host-fed streaming loop: python tools_dev/hostfed_breakdown.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.streaming import FrameStreamer
from fastposecnn_amd.tools.dataset import FrameUploader
import aggregation_layer as al
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
x = synth.make_image(0)[None].to(dev)
cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(1))[0].items()}
cat["mask"] = al.attach_fg_bits(cat["mask"].to(torch.int64).contiguous())
st = FrameStreamer(model)
st.prepare(x, categorical_override=cat)
depth = len(st.models)
up = FrameUploader(1, 480, 640, device=dev, slots=depth + 4)
frames = np.random.default_rng(0).integers(0, 256, (1, 480, 640, 3), dtype=np.uint8)
for mode in ("resident", "host"):
    pending = []
    acc = {"upload": 0.0, "submit": 0.0, "collect": 0.0}
    def step():
        if mode == "host":
            t0 = time.perf_counter(); t, ready = up.upload(frames); acc["upload"] += time.perf_counter() - t0
        else:
            t, ready = x, None
        t0 = time.perf_counter(); pending.append(st.submit(t, categorical_override=cat, ready=ready)); acc["submit"] += time.perf_counter() - t0
        if len(pending) > depth:
            t0 = time.perf_counter(); st.collect(pending.pop(0)); acc["collect"] += time.perf_counter() - t0
    for _ in range(40): step()
    while pending: st.collect(pending.pop(0))
    torch.cuda.synchronize()
    for k in acc: acc[k] = 0.0
    n = 600
    t0 = time.perf_counter()
    for _ in range(n): step()
    while pending: st.collect(pending.pop(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {n / dt:.1f} img/s, {dt / n * 1e3:.3f} ms per frame; host per frame: " + ", ".join(f"{k} {v / n * 1e3:.3f} ms" for k, v in acc.items()))
