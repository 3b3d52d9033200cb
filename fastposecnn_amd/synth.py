"""Synthetic inputs for the benchmarks and the full-size tests.

* `make_image`: the reference's input recipe (F/tools/dataset.py:249-262): uniform(0,255) RGB ->
  imagenet mean/std -> divided by its own max-abs.
* `make_vote_frame`: the post-network "vote bench" fixture — per frame K non-overlapping
  elliptical instances (semi-axes U(30,110) px, classes cycling 1..6, sub-pixel centre), vote
  field rot(eps)(c-p)/|c-p| with eps ~ N(0, 0.03 rad) and 5 % uniformly random outlier
  directions, per-instance constant quaternion / scales / log-z planes + N(0, 0.01).
  CPU `torch.Generator().manual_seed(1000 + frame)`.
Everything is generated on the CPU and copied to the device by the caller.
"""
import math

import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_image(seed, H=480, W=640):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand((3, H, W), generator=g) * 255.0
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    img = (img / 255.0 - mean) / std
    return (img / img.abs().max()).float()


def _place_ellipses(K, H, W, g, rmin, rmax):
    boxes = []
    out = []
    for k in range(K):
        lo, hi = rmin, rmax
        for attempt in range(400):
            rx = float(torch.empty(1).uniform_(lo, hi, generator=g))
            ry = float(torch.empty(1).uniform_(lo, hi, generator=g))
            cx = float(torch.empty(1).uniform_(rx + 1, W - rx - 2, generator=g))
            cy = float(torch.empty(1).uniform_(ry + 1, H - ry - 2, generator=g))
            box = (cx - rx - 2, cy - ry - 2, cx + rx + 2, cy + ry + 2)
            if all(box[2] < b[0] or b[2] < box[0] or box[3] < b[1] or b[3] < box[1] for b in boxes):
                boxes.append(box)
                out.append((cx, cy, rx, ry))
                break
            if attempt % 40 == 39:      # crowded frame: allow smaller instances
                hi = max(lo + 1.0, hi * 0.85)
        else:
            raise RuntimeError("could not place the synthetic instances")
    return out


def make_vote_frame(frame, K=6, H=480, W=640, num_classes=7, rmin=30.0, rmax=110.0, noise=0.03, outlier=0.05):
    """Returns (categorical dict of CPU tensors for ONE frame [1,...], list of true centres)."""
    g = torch.Generator().manual_seed(1000 + frame)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    mask = torch.zeros((H, W), dtype=torch.int64)
    quat = torch.zeros((4, H, W)); scales = torch.zeros((3, H, W)); xy = torch.zeros((2, H, W)); z = torch.zeros((H, W))
    centres = []
    for k, (ex, ey, rx, ry) in enumerate(_place_ellipses(K, H, W, g, rmin, rmax)):
        cx = ex + float(torch.empty(1).uniform_(-0.5, 0.5, generator=g))
        cy = ey + float(torch.empty(1).uniform_(-0.5, 0.5, generator=g))
        m = (((xx - ex) / rx) ** 2 + ((yy - ey) / ry) ** 2) <= 1.0
        cls = 1 + k % (num_classes - 1)
        mask[m] = cls
        dx, dy = cx - xx, cy - yy
        nrm = torch.sqrt(dx * dx + dy * dy).clamp_min(1e-12)
        ux, uy = dx / nrm, dy / nrm
        eps = torch.randn((H, W), generator=g) * noise
        c, s = torch.cos(eps), torch.sin(eps)
        vx, vy = c * ux - s * uy, s * ux + c * uy
        o = torch.rand((H, W), generator=g) < outlier
        ang = torch.rand((H, W), generator=g) * (2 * math.pi)
        vx = torch.where(o, torch.cos(ang), vx)
        vy = torch.where(o, torch.sin(ang), vy)
        xy[0][m] = vx[m]; xy[1][m] = vy[m]
        q = torch.randn(4, generator=g); q = q / q.norm()
        sc = torch.empty(3).uniform_(0.1, 0.5, generator=g)
        lz = float(torch.empty(1).uniform_(6.2, 7.2, generator=g))      # log(depth in mm)
        n = int(m.sum())
        for a in range(4):
            quat[a][m] = q[a] + torch.randn(n, generator=g) * 0.01
        for a in range(3):
            scales[a][m] = sc[a] + torch.randn(n, generator=g) * 0.01
        z[m] = lz + torch.randn(n, generator=g) * 0.01
        centres.append((cx, cy, cls, n))
    # categorical planes are what class compression emits: unit quaternion / xy where foreground
    fg = mask != 0
    qn = quat.norm(dim=0, keepdim=True); quat = torch.where(fg, quat / qn.clamp_min(1e-12), torch.zeros(()))
    vn = xy.norm(dim=0, keepdim=True); xy = torch.where(fg, xy / vn.clamp_min(1e-12), torch.zeros(()))
    cat = {"mask": mask[None], "quaternion": quat[None].float(), "scales": scales[None].float(),
           "xy": xy[None].float(), "z": z[None].float()}
    return cat, centres


def make_vote_batch(frames, **kw):
    cats, centres = [], []
    for f in frames:
        c, ce = make_vote_frame(f, **kw)
        cats.append(c); centres.append(ce)
    cat = {k: torch.cat([c[k] for c in cats], dim=0).contiguous() for k in cats[0]}
    return cat, centres
