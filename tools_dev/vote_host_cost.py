"""Host time of one model.hough_voting(agg) enqueue (no synchronisation inside the loop) against its device time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
m = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
cat_cpu, _ = synth.make_vote_batch(range(1))
cat = {k: v.to(dev) for k, v in cat_cpu.items()}
with torch.no_grad():
    aggs = [m.aggregate(cat) for _ in range(64)]
    for _ in range(8): m.hough_voting(m.aggregate(cat))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for a in aggs: m.hough_voting(a)
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
print(f"host enqueue {1e6 * (t1 - t0) / 64:.1f} us per call, device {1e3 * e0.elapsed_time(e1) / 64:.1f} us per call")
