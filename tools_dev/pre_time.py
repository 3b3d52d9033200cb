"""Times the input side on the GPU box: kernel-only (frames resident) and PCIe-inclusive (pinned host frames)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import fastposecnn_amd
from fastposecnn_amd.tools.dataset import preprocess_frames, FrameUploader
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 480, 640
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8))
xd = x.cuda(); out = torch.empty((B, 3, H, W), device="cuda")
for _ in range(20): preprocess_frames(xd, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): preprocess_frames(xd, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
up = FrameUploader(B, H, W)
xn = x.numpy()
for _ in range(10): up.upload(xn)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): up.upload(xn)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(json.dumps({"B": B, "kernel_us": round(us, 2), "GBps_15B_per_px": round(B * H * W * 15 / us / 1e3, 1),
                  "upload_us_incl_pcie": round((t1 - t0) / 200 * 1e6, 1), "upload_img_s": round(B * 200 / (t1 - t0), 1)}))
