import torch
x = torch.empty((32, 67, 480, 640), device="cuda")
for _ in range(3): x.fill_(1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x.fill_(2.0)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fill %.1f MB in %.1f us = %.2f TB/s" % (x.numel() * 4 / 1e6, ms * 1e3, x.numel() * 4 / ms / 1e9))
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy %.1f MB in %.1f us = %.2f TB/s (read + write)" % (x.numel() * 4 / 1e6, ms * 1e3, 2 * x.numel() * 4 / ms / 1e9))
