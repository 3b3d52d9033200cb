"""Network + aggregation + TWO identical votes per frame on four streams: if the two votes of a frame disagree the vote is not
deterministic under the network's load; if they agree with each other but not with the reference, its inputs were different."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import config, synth
import aggregation_layer as al
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
base = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
models = [copy.copy(base) for _ in range(4)]
K = 3
xs = [synth.make_image(i)[None].to(dev) for i in range(K)]
cats = []
for i in range(K):
    c, _ = synth.make_vote_batch(range(i, i + 1))
    cats.append({k: v.to(dev) for k, v in c.items()})
layer = al.AggregationLayer(None, 7)
streams = [torch.cuda.Stream() for _ in range(4)]
def frame(k, i):
    with torch.no_grad(), torch.cuda.stream(streams[k]):
        models[k].pure_model_forward(xs[i])                      # the load (its own plan per model copy)
        agg, n_dev = layer.forward_deferred(cats[i], 32)
        masks = agg["instance_masks"]; vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
        bits = al.mask_bits_of(masks)
        a, da = rvg.ransac_voting_layer_v3(masks, vertex, 1000, seed=7, return_debug=True, mask_bits=bits, n_dev=n_dev)
        b, db = rvg.ransac_voting_layer_v3(masks, vertex, 1000, seed=7, return_debug=True, mask_bits=bits, n_dev=n_dev)
        chk = (masks[:6].double().sum((1, 2)), agg["xy"][:6].double().abs().sum((1, 2, 3)), bits[:6].sum(1))
        return a[:6], da[0]["hyp"][:6].double().sum((1, 2)), b[:6], db[0]["hyp"][:6].double().sum((1, 2)), chk, da[0], db[0]
for k in range(4):
    for i in range(K):
        frame(k, i)
torch.cuda.synchronize()
refs = [frame(0, i) for i in range(K)]
torch.cuda.synchronize()
refs = [tuple(t.clone() if torch.is_tensor(t) else (tuple(u.clone() for u in t) if isinstance(t, tuple) else None) for t in r) for r in refs]
ab = ref_bad = in_bad = 0
done = 0
while done < N:
    batch = [((done + j) % K, frame((done + j) % 4, (done + j) % K)) for j in range(min(400, N - done))]
    torch.cuda.synchronize()
    for i, (a, ha, b, hb, chk, da, db) in batch:
        r = refs[i]
        if not (torch.equal(a, b) and torch.equal(ha, hb)):
            ab += 1
            if ab <= 4:
                for key in ("tn", "win_idx", "win_count", "inlier_count"):
                    if not torch.equal(da[key][:6], db[key][:6]): print("  ", key, da[key][:6].tolist(), db[key][:6].tolist())
                dh = (da["hyp"][:6] != db["hyp"][:6]).any(2)          # [6, hn]
                for inst in range(6):
                    idx = dh[inst].nonzero().flatten().tolist()
                    if idx: print(f"   instance {inst}: {len(idx)} hypotheses differ, indices {idx[:12]} ... {idx[-4:]}; first: {da['hyp'][inst, idx[0]].tolist()} vs {db['hyp'][inst, idx[0]].tolist()}")
                dc = (da["counts"][:6] != db["counts"][:6])
                print("   counts differing per instance:", dc.sum(1).tolist())
        if not (torch.equal(a, r[0]) and torch.equal(ha, r[1])): ref_bad += 1
        if not all(torch.equal(u, v) for u, v in zip(chk, r[4])): in_bad += 1
    done += len(batch)
print(f"pipe soak: {N} frames; the two votes of a frame disagree: {ab}; first vote != reference: {ref_bad}; aggregation checksums != reference: {in_bad}")
