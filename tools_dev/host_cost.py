"""Host-side cost of one streamed frame: wall time of submit() alone (no device wait), of collect(), and of the
engine call alone.   python tools_dev/host_cost.py [--n 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.streaming import FrameStreamer

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--net-streams", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.ENGINE_GRAPH = bool(int(os.environ.get('FPC_ENGINE_GRAPH', '1')))
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
x = synth.make_image(0)[None].to(dev)
cat_cpu, _ = synth.make_vote_batch(range(1))
cat = {k: v.to(dev) for k, v in cat_cpu.items()}
st = FrameStreamer(model, net_streams=a.net_streams)
for _ in range(12):
    st.collect(st.submit(x, categorical_override=cat))
torch.cuda.synchronize()
t0 = time.perf_counter()
tk = [st.submit(x, categorical_override=cat) for _ in range(a.n)]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
outs = [st.collect(t) for t in tk]
t3 = time.perf_counter()
for rep in range(3):      # short bursts: the hardware queues never fill, so this is pure host time
    torch.cuda.synchronize()
    q0 = time.perf_counter()
    tk = [st.submit(x, categorical_override=cat) for _ in range(8)]
    q1 = time.perf_counter()
    [st.collect(t) for t in tk]
    print(f"burst of 8: host {1e3*(q1-q0)/8:.3f} ms/frame")
print(f"submit (host only) {1e3*(t1-t0)/a.n:.3f} ms/frame; device drained after {1e3*(t2-t0)/a.n:.3f} ms/frame; "
      f"collect after drain {1e3*(t3-t2)/a.n:.3f} ms/frame")
m = st.models[0]
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.n):
        lg = m.pure_model_forward(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pure_model_forward host {1e3*(t1-t0)/a.n:.3f} ms; drained {1e3*(t2-t0)/a.n:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(a.n):
        c = m.class_compression(lg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"class_compression host {1e3*(t1-t0)/a.n:.3f} ms")
    t0 = time.perf_counter()
    ps = [m.post_network_enqueue(cat) for _ in range(a.n)]
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"post_network_enqueue host {1e3*(t1-t0)/a.n:.3f} ms; drained {1e3*(t2-t0)/a.n:.3f} ms")
    t0 = time.perf_counter()
    for p in ps:
        m.post_network_finish(p)
    t1 = time.perf_counter()
    print(f"post_network_finish host {1e3*(t1-t0)/a.n:.3f} ms")
