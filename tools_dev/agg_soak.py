"""Repeats CC + aggregation on fixed inputs on four streams at once and compares every output (labels, masks, xy planes, means,
bit words) with the first call's.  python tools_dev/agg_soak.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth
import aggregation_layer as al
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
dev = torch.device("cuda:0")
K = 3
cats = []
for i in range(K):
    c, _ = synth.make_vote_batch(range(i, i + 1))
    cats.append({k: v.to(dev) for k, v in c.items()})
layer = al.AggregationLayer(None, 7)
def run(cat):
    agg, n_dev = layer.forward_deferred(cat, 32)
    return {**{k: agg[k] for k in ("instance_masks", "xy", "quaternion", "scales", "z", "class_ids", "sample_ids")},
            "bits": al.mask_bits_of(agg["instance_masks"]), "n": n_dev}
refs = []
for c in cats:
    r = run(c); torch.cuda.synchronize()
    n = int(r["n"].item())
    refs.append(({k: v[:n].clone() if k != "n" else v.clone() for k, v in r.items()}, n))
streams = [torch.cuda.Stream() for _ in range(4)]
bad, first, done = {}, None, 0
while done < N:
    batch = []
    for j in range(min(200, N - done)):
        i = (done + j) % K
        with torch.cuda.stream(streams[(done + j) % 4]):
            batch.append((i, run(cats[i])))
    torch.cuda.synchronize()
    for m, (i, r) in enumerate(batch):
        ref, n = refs[i]
        for k, v in r.items():
            a = v if k == "n" else v[:n]
            if not torch.equal(a, ref[k]):
                bad[k] = bad.get(k, 0) + 1
                if first is None:
                    d = (a != ref[k]).nonzero()
                    first = (done + m, i, k, d[:6].tolist(), int(d.shape[0]))
    done += len(batch)
print("agg soak:", N, "calls on 4 streams, mismatches per output:", bad)
if first: print("first mismatch (call, input, output, differing indices, how many):", first)
