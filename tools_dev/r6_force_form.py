"""Round 6: a config-3 forward (ResNet34, batch 32, 480 x 640) on the autotuner's plans, then with every 3x3 / stride-1 site forced
onto one Winograd form (argv[1], default 8): the forward's time by HIP events and the plans the tuner had picked."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastposecnn_amd import config, synth, lib
form = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
enc = sys.argv[3] if len(sys.argv) > 3 else "resnet34"
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.ENCODER = enc; hp.PERFORM_AGGREGATION = False
torch.manual_seed(0)
m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).eval().to(dev)
x = torch.stack([synth.make_image(i % 4, 480, 640) for i in range(B)]).to(dev)
def timed(tag, n=10):
    with torch.no_grad():
        for _ in range(3): m(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): m(x)
        e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / n:.3f} ms per forward", flush=True)
timed("autotuned")
eng = next(iter(m._engines.values()))
print("plans (nsplit): ", collections.Counter(p[2] for p in eng.conv_plans()))
print([p[2] for p in eng.conv_plans()])
n = eng.force_winograd(form)
timed(f"{n} sites forced onto form {form}")
