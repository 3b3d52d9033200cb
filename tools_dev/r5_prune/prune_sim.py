"""CPU study for the exact progressive pruning of k_vote_count (round 5).

For the bench fixture (synth.make_vote_frame) computes the full inlier matrix of hn = 1000 hypotheses per instance with
numpy (float32 arithmetic close to the reference's; exactness does not matter for a work estimate) and replays pruning
schedules: entries are visited in P passes (strided units of 512 ranks); after each pass the leader of the partial counts
is fully counted (L) and every hypothesis with partial + remaining < L is dropped.  Prints the work that remains.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from fastposecnn_amd import synth


def instance_matrix(xy, m, hn, rng, thresh=0.999, max_num=30000):
    ys, xs = np.nonzero(m)
    d = np.stack([xy[0][ys, xs], xy[1][ys, xs]], 1).astype(np.float32)
    c = np.stack([xs, ys], 1).astype(np.float32)
    tn = len(xs)
    if tn > max_num:
        sel = rng.random(tn) < max_num / tn
        d, c = d[sel], c[sel]; tn = len(d)
    idx = rng.integers(0, tn, (hn, 2))
    # two-line intersection (.cu:28-45)
    n0 = np.stack([d[idx[:, 0], 1], -d[idx[:, 0], 0]], 1); n1 = np.stack([d[idx[:, 1], 1], -d[idx[:, 1], 0]], 1)
    c0 = (n0 * c[idx[:, 0]]).sum(1); c1 = (n1 * c[idx[:, 1]]).sum(1)
    det = n0[:, 0] * n1[:, 1] - n0[:, 1] * n1[:, 0]
    ok = np.abs(det) > 1e-6
    det = np.where(ok, det, 1)
    hx = np.where(ok, (c0 * n1[:, 1] - c1 * n0[:, 1]) / det, 0); hy = np.where(ok, (n0[:, 0] * c1 - n1[:, 0] * c0) / det, 0)
    inl = np.zeros((hn, tn), dtype=bool)
    for h0 in range(0, hn, 100):
        gx = hx[h0:h0 + 100, None] - c[None, :, 0]; gy = hy[h0:h0 + 100, None] - c[None, :, 1]
        nn = np.sqrt(gx * gx + gy * gy)
        cos = (gx * d[None, :, 0] + gy * d[None, :, 1]) / np.maximum(nn, 1e-12)
        inl[h0:h0 + 100] = (cos > thresh) & (nn > 1e-6)
    return inl


def simulate(inl, passes, unit=512, order="strided"):
    hn, tn = inl.shape
    nu = (tn + unit - 1) // unit
    uid = np.arange(tn) // unit
    if order == "strided":
        pass_of_unit = np.arange(nu) % passes
    else:
        pass_of_unit = (np.arange(nu) * passes) // nu
    alive = np.ones(hn, bool)
    partial = np.zeros(hn, np.int64)
    full = inl.sum(1)
    work = 0.0
    seen = 0
    surv = []
    for p in range(passes):
        cols = pass_of_unit[uid] == p
        ne = int(cols.sum())
        work += alive.sum() * ne
        partial[alive] += inl[alive][:, cols].sum(1)
        seen += ne
        surv.append(int(alive.sum()))
        if p == passes - 1: break
        lead = int(np.argmax(np.where(alive, partial, -1)))
        L = full[lead]; work += (tn - seen)            # leader's full count
        ub = partial + (tn - seen)
        alive &= (ub > L) | ((ub == L) & (np.arange(hn) <= lead))
    # exactness check: the winner survives
    w = int(np.argmax(full))
    assert alive[w], "winner pruned"
    assert int(np.argmax(np.where(alive, partial, -1))) == w
    return work / (hn * tn), surv


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    hn = 1000
    rng = np.random.default_rng(0)
    tot = {}
    for f in range(frames):
        cat, centres = synth.make_vote_frame(f)
        mask = cat["mask"][0].numpy(); xy = cat["xy"][0].numpy()
        for cls in range(1, 7):
            m = mask == cls
            if m.sum() < 5: continue
            inl = instance_matrix(xy, m, hn, rng)
            rho = inl.sum(1) / inl.shape[1]
            q = np.quantile(rho, [0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
            line = f"frame {f} cls {cls} tn {inl.shape[1]:6d} rho q10..max " + " ".join(f"{v:.2f}" for v in q)
            for passes in (2, 3, 4, 6, 8):
                for order in ("strided",):
                    w, surv = simulate(inl, passes, order=order)
                    tot.setdefault((passes, order), []).append((w, inl.shape[1]))
                    line += f" | P{passes} {w:.2f}"
            print(line, flush=True)
    for k, v in tot.items():
        ws = np.array([a for a, _ in v]); tn = np.array([b for _, b in v])
        print(k, "weighted work", float((ws * tn).sum() / tn.sum()))


if __name__ == "__main__":
    main()
