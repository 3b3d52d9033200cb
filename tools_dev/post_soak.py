"""The whole post-network path (aggregate -> vote -> RT, deferred form) on fixed inputs on four streams at once, optionally
beside a stream that keeps the chip busy with matrix work; every output compared with the first call's.
python tools_dev/post_soak.py [N] [load]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
load = len(sys.argv) > 2 and sys.argv[2] == "load"
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
base = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
models = [copy.copy(base) for _ in range(4)]
for m in models: m._pinned_counts, m._pinned_next = [], -1
K = 3
cats = []
for i in range(K):
    c, _ = synth.make_vote_batch(range(i, i + 1))
    cats.append({k: v.to(dev) for k, v in c.items()})
KEYS = ("xy", "quaternion", "scales", "z", "class_ids", "R", "T")
def dig(a): return {k: a[k].clone() for k in KEYS}
refs = [dig(base.post_network_finish(base.post_network_enqueue(c, seed=11))) for c in cats]
streams = [torch.cuda.Stream() for _ in range(4)]
ls = torch.cuda.Stream(); A = torch.randn((4096, 4096), device=dev); Bm = torch.randn((4096, 4096), device=dev)
bad, first, done = {}, None, 0
while done < N:
    batch = []
    for j in range(min(400, N - done)):
        i = (done + j) % K; k = (done + j) % 4
        if load and j % 8 == 0:
            with torch.cuda.stream(ls):
                torch.mm(A, Bm)
        with torch.cuda.stream(streams[k]), torch.no_grad():
            batch.append((i, k, models[k].post_network_enqueue(cats[i], seed=11)))
    for n, (i, k, t) in enumerate(batch):
        with torch.cuda.stream(streams[k]):
            a = models[k].post_network_finish(t)
        for key in KEYS:
            if not torch.equal(a[key], refs[i][key]):
                bad[key] = bad.get(key, 0) + 1
                if first is None: first = (done + n, i, key, a[key].flatten().tolist()[:14], refs[i][key].flatten().tolist()[:14])
    done += len(batch)
print("post soak:", N, "frames on 4 streams", "(with matrix load)" if load else "", "mismatches per output:", bad)
if first: print("first:", first)
