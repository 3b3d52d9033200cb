"""Phase stamps of k_merge_head (diagnostic build -DFPC_STAMP_MH): mean ticks per workgroup (thread 0).
    python -c "from fastposecnn_amd import build; build.build(force=True, extra=['-DFPC_STAMP_MH'])"; python tools_dev/mh_stamps.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth, _native as nat
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False; hp.ENGINE_AUTOTUNE = False
torch.manual_seed(0)
m = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.stack([synth.make_image(i) for i in range(B)]).to(dev)
lib = nat.lib(); f = lib.fpc_dbg_merge_head_stamps; f.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 6)()
with torch.no_grad():
    for _ in range(3): m.pure_model_forward(x)
    torch.cuda.synchronize(); f(buf)
    for _ in range(10): m.pure_model_forward(x)
    torch.cuda.synchronize(); f(buf)
n = buf[5]
names = ["weights + affines", "gather/GN/ReLU/merge", "barrier", "MFMA head", "partials + stores"]
print(f"{n} workgroups; mean ticks per workgroup: " + ", ".join(f"{nm} {buf[i]/n:.0f}" for i, nm in enumerate(names)) + f"; total {sum(buf[:5])/n:.0f}")
