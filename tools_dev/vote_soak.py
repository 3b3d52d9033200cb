"""Repeats one vote call on fixed inputs on FOUR streams at once (no synchronisation inside a batch of calls) and compares
EVERY output (centres, winner, counts, inlier count) with the first call's: finds rare nondeterminism inside the vote itself.
python tools_dev/vote_soak.py [N] [bits]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastposecnn_amd.lib as L
from fastposecnn_amd import synth
import aggregation_layer as al
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
use_bits = len(sys.argv) > 2 and sys.argv[2] == "bits"
poison = len(sys.argv) > 3 and sys.argv[3] == "poison"      # fill every CU's LDS with NaN patterns before each call
dev = torch.device("cuda:0")
K = 3
sets = []
for i in range(K):
    cat_cpu, _ = synth.make_vote_batch(range(i, i + 1))
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    agg, n_dev = al.AggregationLayer(None, 7).forward_deferred(cat, 32)      # capacity rows + device-side count, as the pipeline
    masks = agg["instance_masks"]; vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    sets.append((masks, vertex, al.mask_bits_of(masks) if use_bits else None, n_dev, int(n_dev.item())))
torch.cuda.synchronize()
refs = []
for m, v, b, nd, n in sets:
    o, d = rvg.ransac_voting_layer_v3(m, v, 1000, seed=7, return_debug=True, mask_bits=b, n_dev=nd)
    refs.append((o[:n].clone(), {k: t[:n].clone() for k, t in d[0].items()}))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
if poison:
    import ctypes
    PL = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_poison.so"))
    PL.poison_lds.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
bad = {}
first = None
done = 0
while done < N:
    batch = []
    for j in range(min(2000, N - done)):
        i = (done + j) % K
        s = streams[(done + j) % 4]
        with torch.cuda.stream(s):
            if poison:
                PL.poison_lds(0x7FC12345 if j % 2 else 0xFFFFFFFF, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
            o, d = rvg.ransac_voting_layer_v3(sets[i][0], sets[i][1], 1000, seed=7, return_debug=True, mask_bits=sets[i][2], n_dev=sets[i][3])
        n = sets[i][4]
        batch.append((i, o[:n], {k: t[:n] for k, t in d[0].items()}))
    torch.cuda.synchronize()
    for n, (i, o, d) in enumerate(batch):
        ro, rd = refs[i]
        for k in list(rd) + ["xy"]:
            a, b = (o, ro) if k == "xy" else (d[k], rd[k])
            if not torch.equal(a, b):
                bad[k] = bad.get(k, 0) + 1
                if first is None:
                    first = (done + n, i, k, {kk: (d[kk] != rd[kk]).nonzero().flatten().tolist()[:10] for kk in rd},
                             o.flatten().tolist(), ro.flatten().tolist())
    done += len(batch)
print("vote soak:", N, "calls on 4 streams, mismatches per output:", bad)
if first: print("first mismatch (call, input, output, differing indices per debug output, xy got, xy want):", first)
