"""Statement-level host timing of FrameUploader.upload at batch 1, alone and beside the frame streams (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastposecnn_amd
from fastposecnn_amd.tools.dataset import FrameUploader, preprocess_frames
dev = torch.device("cuda:0")
up = FrameUploader(1, 480, 640, device=dev, slots=6)
frames = np.random.default_rng(0).integers(0, 256, (1, 480, 640, 3), dtype=np.uint8)
src = torch.from_numpy(frames)
T = {}
def tick(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + t - t0; return t
def upload(self, src):
    k = self._i % len(self._host); self._i += 1
    t = time.perf_counter()
    if self._busy[k] is not None: self._busy[k].synchronize()
    t = tick("busy.sync", t)
    self._host[k].copy_(src)
    t = tick("to_pinned", t)
    consumed = torch.cuda.Event(); consumed.record(torch.cuda.current_stream(self.device))
    t = tick("consumed.record", t)
    with torch.cuda.stream(self.stream):
        if self._free[k] is not None: self.stream.wait_event(self._free[k])
        t = tick("wait_free", t)
        self._dev[k].copy_(self._host[k], non_blocking=True)
        t = tick("h2d", t)
        self._busy[k] = torch.cuda.Event(); self._busy[k].record()
        t = tick("busy.record", t)
        preprocess_frames(self._dev[k], self.params, out=self._out[k])
        t = tick("preprocess", t)
        done = torch.cuda.Event(); done.record()
        t = tick("done.record", t)
    self._free[k] = consumed
    return self._out[k], done
for _ in range(12): upload(up, src)
torch.cuda.synchronize(); T.clear(); n = 200; t0 = time.perf_counter()
for _ in range(n): upload(up, src)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("alone:", {k: round(v / n * 1e6, 1) for k, v in T.items()}, "total", round(dt / n * 1e6, 1), "us")
