"""Gradient at the output of every conv / norm module: native-conv path and torch f32 path against the float64 model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.lib import train_conv
dev = torch.device("cuda:0")
hp = config.HEAD_TRAINING(); hp.RUNTIME_TIMING = False
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout2d): m.p = 0.0
x = torch.stack([synth.make_image(i, 96, 128) for i in range(2)]).to(dev)
store = {}
order = []
def mk(name):
    def fh(mod, inp, out):
        if name not in order: order.append(name)
        out.register_hook(lambda g: store.setdefault(cur[0], {}).__setitem__(name, g.detach().double().clone()))
    return fh
for n, m in model.named_modules():
    if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
        m.register_forward_hook(mk(n))
cur = [None]
for tag in ("native", "torch32", "torch64"):
    cur[0] = tag
    train_conv.ENABLED = tag == "native"
    if tag == "torch64": model = model.double(); x = x.double()
    model.zero_grad(set_to_none=True)
    out = model.pure_model_forward(x)
    sum(v.square().mean() for v in out.values()).backward(); torch.cuda.synchronize()
for n in reversed(order):
    if not n.startswith("encoder"): continue
    r = store["torch64"][n]; s = max(r.abs().max().item(), 1e-30)
    en = (store["native"][n] - r).abs().max().item() / s; et = (store["torch32"][n] - r).abs().max().item() / s
    print(f"{n:42s} native {en:.2e} torch32 {et:.2e} {'<<<' if en > 5 * et else ''}")
