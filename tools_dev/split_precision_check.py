"""Numerical check (CPU, no GPU): how accurate is a bf16 x 3 split-precision product chain with f32 accumulation —
the arithmetic `v_mfma_f32_32x32x16_bf16` would run — on a convolution-sized contraction (K = 2304), against float64?
Truncation split a = a1 + a2 + a3 (8 significant bits each: exact), products of two 8-bit parts are exact in f32, so
only the f32 accumulation and the dropped cross terms contribute.   python tools_dev/split_precision_check.py"""
import torch
torch.manual_seed(0)
M, K, N = 512, 2304, 128
a = torch.randn(M, K).abs_()                  # post-ReLU-like activations
w = torch.randn(K, N) * 0.05
ref = a.double() @ w.double()

def split3(x):
    p1 = (x.view(torch.int32) & -65536).view(torch.float32); r = x - p1
    p2 = (r.view(torch.int32) & -65536).view(torch.float32); r2 = r - p2
    p3 = (r2.view(torch.int32) & -65536).view(torch.float32)
    return p1, p2, p3

(a1, a2, a3), (w1, w2, w3) = split3(a), split3(w)
mm = lambda x, y: x @ y
cases = {"plain f32 matmul": a @ w,
         "bf16x3, 6 terms (i + j <= 4)": mm(a1, w1) + mm(a1, w2) + mm(a2, w1) + mm(a1, w3) + mm(a2, w2) + mm(a3, w1),
         "bf16x3, 3 terms (i + j <= 3)": mm(a1, w1) + mm(a1, w2) + mm(a2, w1)}
scale = ref.abs().max().item()
for name, v in cases.items():
    e = (v.double() - ref).abs()
    print(f"{name:30s} max abs err {e.max().item():.3e} = {e.max().item() / scale:.2e} of max|ref|, rms {e.pow(2).mean().sqrt().item():.3e}")
