"""Measurement of the matching row: gtf.batchwise_get_2d_iou on 640x480 masks.
    python tools_dev/iou_bench.py [--out profiles/r01_iou_bench.json]
HIP events around fpc_mask_iou (both kernels + the count memset) on its stream; algorithmic bytes =
(n1 + n2) * H*W * 4 read + 4*n1*n2 written, against the 8 TB/s HBM peak; beside it the reference's own
algorithm (the [n1,n2,H,W] logical_and / logical_or expansion) as torch-ROCm ops on the same GPU, and the C
oracle on the host (one thread, bounded sample)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastposecnn_amd.lib as L
from oracle import oracle as orc

ap = argparse.ArgumentParser(); ap.add_argument("--out"); a = ap.parse_args()
dev = torch.device("cuda:0")
H, W = 480, 640
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
g = torch.Generator().manual_seed(0)
def masks(n):
    out = []
    for _ in range(n):
        cx, cy, r = (torch.randint(lo, hi, (1,), generator=g) for lo, hi in ((60, 580), (60, 420), (30, 110)))
        out.append((((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).float())
    return torch.stack(out)
res = {"what": "gtf.batchwise_get_2d_iou (fpc_mask_iou) on 640x480 f32 masks, MI355X", "cases": []}
for n1, n2 in ((6, 6), (8, 8), (32, 32)):
    m1, m2 = masks(n1).to(dev), masks(n2).to(dev)
    f = lambda: L.gtf.batchwise_get_2d_iou(m1, m2)
    ref = lambda: ((m1[:, None].expand(n1, n2, H, W).logical_and(m2.expand(n1, n2, H, W))).sum((2, 3)) /
                   (m1[:, None].expand(n1, n2, H, W).logical_or(m2.expand(n1, n2, H, W))).sum((2, 3)))
    assert torch.equal(f(), ref())
    def timeit(fn, it):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    us, us_ref = timeit(f, 200), timeit(ref, 20)
    alg = (n1 + n2) * H * W * 4 + 4 * n1 * n2
    c = {"n1": n1, "n2": n2, "us_per_call": round(us, 2), "algorithmic_bytes": alg, "achieved_GBps": round(alg / us / 1e3, 1),
         "hbm_peak_GBps": 8000.0, "frac": round(alg / us / 1e3 / 8000.0, 4),
         "torch_rocm_reference_algorithm_us": round(us_ref, 1), "speedup_vs_torch_expansion": round(us_ref / us, 1)}
    if n1 <= 8:
        a1, a2 = m1.cpu().numpy(), m2.cpu().numpy()
        t0 = time.perf_counter(); orc.mask_iou(a1, a2); dt = time.perf_counter() - t0
        c["cpu_oracle_ms_1_thread"] = round(dt * 1e3, 1)
    res["cases"].append(c)
    print(c)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
