"""Soak test of the frame-streaming runtime: N frames of K alternating inputs through FrameStreamer with five
frames in flight; every frame's outputs must equal, bit for bit, what the same plan produced for that input
before (the four plans are autotuned separately, so equality is per plan).  Catches cross-stream races
(shared workspaces, pinned read-back slots, allocator reuse across streams).   python tools_dev/soak.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
from fastposecnn_amd.streaming import FrameStreamer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
K = 3
xs = [synth.make_image(i)[None].to(dev) for i in range(K)]
cats = []
for i in range(K):
    c, _ = synth.make_vote_batch(range(i, i + 1))
    cats.append({k: v.to(dev) for k, v in c.items()})
st = FrameStreamer(model)
nplan = len(st.models)

def digest(out):
    a = out["aggregated"]
    return (out["categorical"]["mask"].sum().item(), out["logits"]["quaternion"].double().sum().item(),
            tuple(a["xy"].flatten().tolist()), tuple(a["class_ids"].tolist()), a["RT"].double().sum().item())

# reference digests per (plan, input): run each combination alone first (seed fixed per input so the vote's sampler repeats)
want = {}
for rep in range(nplan * K):
    k = st._n % nplan
    i = rep % K
    torch.manual_seed(100 + i)
    want[(k, i)] = digest(st.collect(st.submit(xs[i], categorical_override=cats[i])))
    # nplan and K coprime -> all combinations are visited in nplan*K submissions
assert len(want) == nplan * K, (len(want), nplan, K)
pending, bad, t0 = [], 0, time.perf_counter()
for f in range(N):
    k = st._n % nplan
    i = f % K
    torch.manual_seed(100 + i)
    pending.append((k, i, st.submit(xs[i], categorical_override=cats[i])))
    if len(pending) > nplan:
        kk, ii, t = pending.pop(0)
        if digest(st.collect(t)) != want[(kk, ii)]:
            bad += 1
while pending:
    kk, ii, t = pending.pop(0)
    if digest(st.collect(t)) != want[(kk, ii)]:
        bad += 1
dt = time.perf_counter() - t0
print(f"soak: {N} frames, {bad} mismatching, {N / dt:.0f} img/s incl. per-frame digests")
sys.exit(1 if bad else 0)
