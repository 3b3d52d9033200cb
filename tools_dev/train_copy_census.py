"""Which training-mode convolutions receive tensors that are not channel-last (a layout copy each)?"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.lib import train_conv
dev = torch.device("cuda:0")
hp = config.HEAD_TRAINING(); hp.RUNTIME_TIMING = False
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).to(dev).train()
x = torch.stack([synth.make_image(i) for i in range(2)]).to(dev)
census = collections.Counter()
orig = train_conv._channels_last
import traceback
def cl(t):
    r = orig(t)
    if r is not t:
        who = [f.name for f in traceback.extract_stack()[-4:-1]]
        census[(tuple(t.shape), tuple(t.stride()), who[-1])] += 1
    return r
train_conv._channels_last = cl
out = model.pure_model_forward(x)
sum(v.square().mean() for v in out.values()).backward(); torch.cuda.synchronize()
tot = 0
for (shape, stride, who), n in sorted(census.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    mb = n * torch.Size(shape).numel() * 4 / 1e6; tot += mb
    print(f"{n:3d} x {shape} strides {stride} in {who}: {mb:.1f} MB")
print("total copied MB (B=2):", tot)
