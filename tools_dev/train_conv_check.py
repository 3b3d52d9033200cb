"""Every training-mode convolution of one forward/backward of the model, native result against aten's on the SAME inputs."""
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
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 128)
x = torch.stack([synth.make_image(i, H, W) for i in range(2)]).to(dev)
orig_b, orig_f = train_conv._Conv2dFn.backward, train_conv._Conv2dFn.forward
rows = []
def fwd(ctx, x, w, bias, stride, pad):
    y = orig_f(ctx, x, w, bias, stride, pad)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride, pad)
    rows.append(("fwd", tuple(x.shape), tuple(w.shape), stride, (y.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)))
    return y
def bwd(ctx, gy):
    out = orig_b(ctx, gy)
    x, w = ctx.saved_tensors
    s, p = ctx.stride, ctx.pad
    ax, aw, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])
    if out[0] is not None:
        rows.append(("dx", tuple(x.shape), tuple(w.shape), s, (out[0].double() - ax).abs().max().item() / max(ax.abs().max().item(), 1e-30)))
    rows.append(("dw", tuple(x.shape), tuple(w.shape), s, (out[1].double() - aw).abs().max().item() / max(aw.abs().max().item(), 1e-30)))
    return out
train_conv._Conv2dFn.forward = staticmethod(fwd); train_conv._Conv2dFn.backward = staticmethod(bwd)
out = model.pure_model_forward(x)
sum(v.square().mean() for v in out.values()).backward(); torch.cuda.synchronize()
rows.sort(key=lambda r: -r[4])
for r in rows[:14]: print("%s x%s w%s s%d rel err %.3e" % r)
print(len(rows), "checks; plan cache:", {k[1:]: v for k, v in train_conv._plan_cache.items()})
