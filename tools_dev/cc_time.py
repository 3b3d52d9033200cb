"""Device time of the connected-component labelling (graph replay of 20 captured calls): bit-word entry, i64 entry, per batch.
    python tools_dev/cc_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth, _native as nat
import aggregation_layer as al

dev = torch.device("cuda:0")
lib = nat.lib()
for frames in (1, 32):
    cat = {k: v.to(dev) for k, v in synth.make_vote_batch(range(frames))[0].items()}
    cm = cat["mask"].to(torch.int64).contiguous()
    B, H, W = cm.shape
    al.attach_fg_bits(cm)
    bits = al.fg_bits_of(cm)
    labels = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    n_dev = torch.empty(1, dtype=torch.int32, device=dev)
    root = torch.empty(64 * B, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.fpc_cc_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    for name in ("bits", "i64"):
        def call():
            st = torch.cuda.current_stream().cuda_stream
            if name == "bits":
                nat.check(lib.fpc_cc_label_bits(bits.data_ptr(), B, H, W, labels.data_ptr(), n_dev.data_ptr(), root.data_ptr(), root.numel(),
                                                ws.data_ptr(), ws.numel(), st), "cc bits")
            else:
                nat.check(lib.fpc_cc_label(cm.data_ptr(), B, H, W, labels.data_ptr(), n_dev.data_ptr(), root.data_ptr(), root.numel(),
                                           ws.data_ptr(), ws.numel(), st), "cc i64")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call(); side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    call()
        torch.cuda.current_stream().wait_stream(side)
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(15):
            e0.record(); g.replay(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        alg = B * 12 * H * W
        print(f"frames={frames} {name}: {ts[len(ts)//2]:.1f} us per call (n={int(n_dev.item())}); alg {alg/ts[len(ts)//2]/1e3:.1f} GB/s = {alg/ts[len(ts)//2]/1e3/8000:.3f} of 8 TB/s")
