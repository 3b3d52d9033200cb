import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_net as T
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(77)
x = torch.randn((1, 256, 15, 20), generator=g)
w = torch.randn((128, 256, 3, 3), generator=g) / 48.0
for bm, bn, ns in ((64, 64, 4), (64, 64, 2)):
    two, _, _ = T._conv2d(dev, x, w, 1, 1, bm=bm, bn=bn, nsplit=100 + ns)
    one, _, plan = T._conv2d(dev, x, w, 1, 1, bm=bm, bn=bn, nsplit=ns)
    d = (one - two).abs()      # NCHW [1,128,15,20]
    bad = d > 1e-6
    print(plan, "bad frac", bad.float().mean().item(), "nan", torch.isnan(one).sum().item())
    pix = bad[0].any(0).reshape(-1)       # per pixel
    ch = bad[0].reshape(128, -1).any(1)
    print(" bad pixels", pix.nonzero().flatten().tolist()[:80])
    print(" bad channels", ch.nonzero().flatten().tolist()[:80])
    r = (one / two)[bad]
    print(" ratio sample", r[:10].tolist())
