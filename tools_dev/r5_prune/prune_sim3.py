"""Pruning with the 'dead entry' disc bound: unseen entries whose cone misses the disc around the leader that holds every
alive hypothesis cannot vote for any of them and leave the bound's remaining count (see prune_sim.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from fastposecnn_amd import synth


def instance(xy, m, hn, rng, thresh=0.999, max_num=30000):
    ys, xs = np.nonzero(m)
    d = np.stack([xy[0][ys, xs], xy[1][ys, xs]], 1).astype(np.float64)
    c = np.stack([xs, ys], 1).astype(np.float64)
    tn = len(xs)
    idx = rng.integers(0, tn, (hn, 2))
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
    return inl, np.stack([hx, hy], 1), c, d


def simulate(inl, hyp, c, d, fracs, disc, thresh=0.999, unit=512):
    hn, tn = inl.shape
    kappa = np.sqrt(1 - thresh ** 2) / thresh
    nu = (tn + unit - 1) // unit
    uid = np.arange(tn) // unit
    cum = np.cumsum(fracs); P = len(fracs)
    pos = (np.arange(nu) * 0.6180339887498949) % 1.0
    pass_of_unit = np.searchsorted(cum, pos, side="right").clip(0, P - 1)
    pe = pass_of_unit[uid]
    alive = np.ones(hn, bool); partial = np.zeros(hn, np.int64); full = inl.sum(1)
    work = 0.0; surv = []
    for p in range(P):
        cols = pe == p
        na = 32 * ((int(alive.sum()) + 31) // 32)
        work += na * int(cols.sum())
        partial[alive] += inl[alive][:, cols].sum(1)
        surv.append(int(alive.sum()))
        if p == P - 1: break
        unseen = pe > p
        rem = int(unseen.sum())
        lead = int(np.argmax(np.where(alive, partial, -1)))
        L = full[lead]
        def prune(rem_eff):
            ub = partial + rem_eff
            return alive & ((ub > L) | ((ub == L) & (np.arange(hn) <= lead)))
        a2 = prune(rem)
        if disc:
            D = hyp[lead][None] - c[unseen]; e = d[unseen] / np.maximum(np.linalg.norm(d[unseen], axis=1, keepdims=True), 1e-12)
            t = (D * e).sum(1); s = D[:, 0] * e[:, 1] - D[:, 1] * e[:, 0]
            m = np.abs(s) - kappa * t
            for it in range(4):
                R = np.linalg.norm(hyp[a2] - hyp[lead][None], axis=1).max()
                dead = int((m > R * np.sqrt(1 + kappa ** 2) * 1.001 + 1e-3).sum())
                a3 = prune(rem - dead)
                if a3.sum() == a2.sum(): break
                a2 = a3
        alive = a2
    w = int(np.argmax(full)); assert alive[w]
    return work / (hn * tn), surv


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rng = np.random.default_rng(0)
    scheds = [(0.3, 0.3, 0.4), (0.25, 0.35, 0.4), (0.2, 0.3, 0.5), (0.4, 0.6), (0.2, 0.2, 0.2, 0.4)]
    res = {}
    for f in range(frames):
        cat, _ = synth.make_vote_frame(f)
        mask = cat["mask"][0].numpy(); xy = cat["xy"][0].numpy()
        for cls in range(1, 7):
            inl, hyp, c, d = instance(xy, mask == cls, 1000, rng)
            for s in scheds:
                for disc in (False, True):
                    w, sv = simulate(inl, hyp, c, d, s, disc)
                    res.setdefault((s, disc), []).append((w, inl.shape[1], sv))
    for k, v in res.items():
        ws = np.array([a for a, _, _ in v]); tn = np.array([b for _, b, _ in v])
        print(k, "work %.3f" % float((ws * tn).sum() / tn.sum()), np.mean(np.array([s for _, _, s in v]), 0).round(0))


if __name__ == "__main__":
    main()
