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
VOTE_DEBUG = bool(os.environ.get("SOAK_VOTE_DEBUG"))
if VOTE_DEBUG:      # the vote's own integers ride along in the output dict: which of them moves when xy moves?
    import hough_voting as _hv2, ransac_voting_gpu_layer.ransac_voting_gpu as _rvg2
    import torch as _t
    def _fwd(self, agg_data, n_dev=None, seed=None):
        from aggregation_layer import mask_bits_of
        uv, mask = agg_data['xy'], agg_data['instance_masks']
        out, dbg = _rvg2.ransac_voting_layer_v3(mask=mask, vertex=_t.unsqueeze(uv.permute(0, 2, 3, 1), dim=3),
                                                round_hyp_num=self.HPARAM.HV_NUM_OF_HYPOTHESES, n_dev=n_dev, seed=seed,
                                                mask_bits=mask_bits_of(mask), return_debug=True)
        d = dbg[0]
        agg_data.update({'hypothesis': out, 'pruned_hypothesis': out, 'xy': _t.squeeze(out, dim=1), 'xy_mask': uv,
                         'dbg_tn': d['tn'], 'dbg_win_idx': d['win_idx'], 'dbg_win_count': d['win_count'],
                         'dbg_inl': d['inlier_count'], 'dbg_counts_sum': d['counts'].long().sum(1),
                         'dbg_hyp_sum': d['hyp'].double().sum((1, 2)), 'dbg_mask_sum': mask.double().sum((1, 2)),
                         'dbg_uv_sum': uv.double().abs().sum((1, 2, 3))})
        return agg_data
    _hv2.HoughVotingLayer.forward = _fwd
if os.environ.get("SOAK_NO_BITS"):            # experiment: the vote reads the f32 masks instead of the bit words
    import aggregation_layer as _al, hough_voting as _hv
    _al.mask_bits_of = lambda m: None
if os.environ.get("SOAK_NO_ROOT_PIX"):        # experiment: accumulation and planes as two launches
    import aggregation_layer as _al2
    _orig = _al2.AggregationLayer.batchwise_break_segmentation_mask
    def _no_rp(self, *a, **k):
        r = _orig(self, *a, **k)
        try: del r[0]._fpc_root_pix
        except AttributeError: pass
        return r
    _al2.AggregationLayer.batchwise_break_segmentation_mask = _no_rp
dev = torch.device("cuda:0")
hp = config.INFERENCE(); hp.RUNTIME_TIMING = False
hp.ENGINE_SPLIT_PRECISION = bool(int(os.environ.get("FPC_SPLIT_PRECISION", "1")))
hp.ENGINE_GRAPH = bool(int(os.environ.get("FPC_ENGINE_GRAPH", "1")))
hp.ENCODER = os.environ.get("SOAK_ENCODER", hp.ENCODER)        # SOAK_ENCODER=resnet34 SOAK_BATCH=32: BASELINE config 3
BATCH = int(os.environ.get("SOAK_BATCH", "1"))
torch.manual_seed(0)
model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval().to(dev)
K = 3
xs = [torch.stack([synth.make_image(i * BATCH + j) for j in range(BATCH)]).to(dev) for i in range(K)]
cats = []
for i in range(K):
    c, _ = synth.make_vote_batch(range(i * BATCH, (i + 1) * BATCH))
    cats.append({k: v.to(dev) for k, v in c.items()})
    if not os.environ.get("SOAK_NO_FG_BITS"):      # as the class compression hands the mask over: with its foreground bit words
        import aggregation_layer as _al3
        cats[-1]["mask"] = _al3.attach_fg_bits(cats[-1]["mask"].to(torch.int64).contiguous())
st = FrameStreamer(model)
nplan = len(st.models)

NAMES = ("categorical mask sum", "quaternion logits sum", "xy", "class ids", "RT sum", "quaternion means", "scales", "z", "sample ids",
         "tn", "win_idx", "win_count", "inlier_count", "counts sum", "hyp sum", "mask sum", "uv sum")


def digest(out):
    a = out["aggregated"]
    return (out["categorical"]["mask"].sum().item(), out["logits"]["quaternion"].double().sum().item(),
            tuple(a["xy"].flatten().tolist()), tuple(a["class_ids"].tolist()), a["RT"].double().sum().item(),
            tuple(a["quaternion"].flatten().tolist()), tuple(a["scales"].flatten().tolist()), tuple(a["z"].flatten().tolist()),
            tuple(a["sample_ids"].tolist())) + (tuple(tuple(a[k].tolist()) for k in ("dbg_tn", "dbg_win_idx", "dbg_win_count", "dbg_inl",
                                                     "dbg_counts_sum", "dbg_hyp_sum", "dbg_mask_sum", "dbg_uv_sum")) if VOTE_DEBUG else ())


def report(got, want, f):
    for n, g, w in zip(NAMES, got, want):
        if g != w:
            print(f"frame {f}: {n} differs:\n   got  {g}\n   want {w}")
            if n == "xy":      # does the odd row belong to another input (a row the vote did not write: stale memory)?
                rows_g = [tuple(g[2 * j:2 * j + 2]) for j in range(len(g) // 2)]
                for (kk, ii), wd in WANT.items():
                    rows_w = [tuple(wd[2][2 * j:2 * j + 2]) for j in range(len(wd[2]) // 2)]
                    for j, r in enumerate(rows_g):
                        if r != tuple(w[2 * j:2 * j + 2]) and r in rows_w:
                            print(f"   row {j} of the odd frame equals row {rows_w.index(r)} of (plan {kk}, input {ii})")

# reference digests per (plan, input): run each combination alone first (seed fixed per input so the vote's sampler repeats)
want = {}
WANT = want
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
        d = digest(st.collect(t))
        if d != want[(kk, ii)]:
            bad += 1
            if bad <= 3: report(d, want[(kk, ii)], f)
while pending:
    kk, ii, t = pending.pop(0)
    d = digest(st.collect(t))
    if d != want[(kk, ii)]:
        bad += 1
        if bad <= 3: report(d, want[(kk, ii)], -1)
dt = time.perf_counter() - t0
print(f"soak: {N} frames, {bad} mismatching, {N / dt:.0f} img/s incl. per-frame digests")
sys.exit(1 if bad else 0)
