"""Unequal pass fractions for the pruning study (see prune_sim.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from prune_sim import instance_matrix
from fastposecnn_amd import synth


def simulate_f(inl, fracs, unit=512, tilewise=False):
    hn, tn = inl.shape
    nu = (tn + unit - 1) // unit
    uid = np.arange(tn) // unit
    # assign units to passes: strided pattern approximating the fractions
    cum = np.cumsum(fracs); P = len(fracs)
    # low-discrepancy: unit k gets position (k * 0.6180339887) mod 1
    pos = (np.arange(nu) * 0.6180339887498949) % 1.0
    pass_of_unit = np.searchsorted(cum, pos, side="right").clip(0, P - 1)
    alive = np.ones(hn, bool); partial = np.zeros(hn, np.int64); full = inl.sum(1)
    work = 0.0; seen = 0; surv = []
    for p in range(P):
        cols = pass_of_unit[uid] == p
        ne = int(cols.sum())
        na = int(alive.sum())
        if tilewise: na = 32 * ((na + 31) // 32)
        work += na * ne
        partial[alive] += inl[alive][:, cols].sum(1)
        seen += ne; surv.append(int(alive.sum()))
        if p == P - 1: break
        lead = int(np.argmax(np.where(alive, partial, -1)))
        L = full[lead]
        ub = partial + (tn - seen)
        alive &= (ub > L) | ((ub == L) & (np.arange(hn) <= lead))
    w = int(np.argmax(full)); assert alive[w]
    return work / (hn * tn), surv


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rng = np.random.default_rng(0)
    scheds = [(0.5, 0.5), (0.4, 0.6), (0.35, 0.65), (0.3, 0.7), (0.3, 0.3, 0.4), (0.25, 0.25, 0.5), (0.3, 0.2, 0.5), (0.25, 0.35, 0.4),
              (0.2, 0.2, 0.2, 0.4), (0.25, 0.15, 0.2, 0.4), (0.25, 0.25, 0.25, 0.25)]
    tot = {s: [] for s in scheds}; survs = {s: [] for s in scheds}
    for f in range(frames):
        cat, _ = synth.make_vote_frame(f)
        mask = cat["mask"][0].numpy(); xy = cat["xy"][0].numpy()
        for cls in range(1, 7):
            m = mask == cls
            inl = instance_matrix(xy, m, 1000, rng)
            for s in scheds:
                w, sv = simulate_f(inl, s, tilewise=True)
                tot[s].append((w, inl.shape[1])); survs[s].append(sv)
    for s in scheds:
        ws = np.array([a for a, _ in tot[s]]); tn = np.array([b for _, b in tot[s]])
        print(s, "work %.3f" % float((ws * tn).sum() / tn.sum()), "survivors per pass", np.mean(np.array(survs[s]), 0).round(0))


if __name__ == "__main__":
    main()
