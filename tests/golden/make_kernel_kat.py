"""Hand-derived known-answer vectors for the two live CUDA kernels of the reference
(RV/src/ransac_voting_kernel.cu:22-48 generate_hypothesis_kernel, :100-125 voting_for_hypothesis_kernel).

The extension cannot be built here (no nvcc, removed ATen APIs) and the reference ships no test for it, so these
vectors are the pin of the 40 lines the oracle restates.  NOTHING here is produced by running the oracle or the HIP
library: every input is a small integer, a power of two, or float32(1e-6) itself, so that each product, sum, square
root and quotient of the kernel is exact in fp32 and the expected output follows from the kernel's source by the
arithmetic written out in each case's "why".  Writes kernel_kat.json.

Notation of the kernel (.cu:28-36): for pixel t with vote (dx, dy) at (cx, cy):  n = (nx, ny) = (dy, -dx).
    det_y = nx1*ny0 - nx0*ny1 ; det_x = ny1*nx0 - ny0*nx1                       (.cu:42-43: skip if fabs(.) < 1e-6, DOUBLE literal)
    y = (nx1*(nx0*cx0+ny0*cy0) - nx0*(nx1*cx1+ny1*cy1)) / det_y                 (.cu:44)
    x = (ny1*(nx0*cx0+ny0*cy0) - ny0*(nx1*cx1+ny1*cy1)) / det_x                 (.cu:45)
    a skipped pair leaves the zero-initialised output (.cu:75 at::zeros).
Vote (.cu:112-125): d = h - c; norm1 = sqrt(nx^2+ny^2) with (nx, ny) the RAW vote; norm2 = |d|; skip if norm1 < 1e-6 or
    norm2 < 1e-6 (double literal); inlier iff (d.n)/(norm1*norm2) > thresh (strict, signed); only ever writes 1.
"""
import json
import os
import struct

import numpy as np

F1E6 = float(np.float32(1e-6))                                  # 9.999999974752427e-07 < 1e-6: the fp32 neighbour BELOW the double literal
F1E6_UP = float(np.nextafter(np.float32(1e-6), np.float32(1)))  # 1.0000001111620804e-06 > 1e-6
assert F1E6 < 1e-6 < F1E6_UP and struct.pack("<f", F1E6) == bytes.fromhex("bd378635")
P20, P19 = 2.0 ** -20, 2.0 ** -19                               # 9.5367e-07 < 1e-6 < 1.9073e-06
F06 = float(np.float32(0.6))                                    # fl(3/5)
F06_DOWN = float(np.nextafter(np.float32(0.6), np.float32(0)))

hyp_cases = [
    dict(name="axes_meet", why="d0=(1,0)@(0,3): n0=(0,-1); d1=(0,1)@(4,0): n1=(1,0). det_y=1*(-1)-0*0=-1, det_x=0*0-(-1)*1=1. "
         "a0=0*0+(-1)*3=-3, a1=1*4+0*0=4. y=(1*(-3)-0*4)/(-1)=3, x=(0*(-3)-(-1)*4)/1=4.",
         direct=[[1, 0], [0, 1]], coords=[[0, 3], [4, 0]], pair=[0, 1], expect=[4, 3]),
    dict(name="scaled_votes", why="non-unit votes: d0=(2,0)@(0,3): n0=(0,-2); d1=(0,0.5)@(4,0): n1=(0.5,0). det_y=0.5*(-2)-0=-1, "
         "det_x=0-(-2)*0.5=1. a0=(-2)*3=-6, a1=0.5*4=2. y=(0.5*(-6)-0)/(-1)=3, x=(0-(-2)*2)/1=4: scale does not move the point.",
         direct=[[2, 0], [0, 0.5]], coords=[[0, 3], [4, 0]], pair=[0, 1], expect=[4, 3]),
    dict(name="det_2^-20_skipped", why="d0=(1,0): n0=(0,-1); d1=(1,2^-20): n1=(2^-20,-1). det_y=2^-20*(-1)-0*(-1)=-2^-20, "
         "|det_y|=9.54e-7 < 1e-6 -> return before the store: output stays (0,0).",
         direct=[[1, 0], [1, P20]], coords=[[0, 0], [4, 0]], pair=[0, 1], expect=[0, 0]),
    dict(name="det_2^-19_computed", why="d1=(1,2^-19): n1=(2^-19,-1). det_y=-2^-19, det_x=0-(-1)*2^-19=2^-19, both 1.9e-6 >= 1e-6. "
         "a0=0, a1=2^-19*4+(-1)*0=2^-17. y=(2^-19*0-0*2^-17)/(-2^-19)=-0 (== 0), x=((-1)*0-(-1)*2^-17)/2^-19=4: the x axis meets the "
         "line through (4,0).", direct=[[1, 0], [1, P19]], coords=[[0, 0], [4, 0]], pair=[0, 1], expect=[4, 0]),
    dict(name="det_equals_float32(1e-6)_skipped", why="d1=(1,f32(1e-6)): det_y=-f32(1e-6) exactly (product with -1). "
         "fabs(det) = 9.99999997e-7 is compared with the DOUBLE literal 1e-6: smaller -> skipped -> (0,0).  An fp32 compare "
         "`< 1e-6f` would NOT skip (equal): this vector tells the two apart.",
         direct=[[1, 0], [1, F1E6]], coords=[[0, 0], [4, 0]], pair=[0, 1], expect=[0, 0]),
    dict(name="det_next_float_above_1e-6_computed", why="d1=(1,nextafter(f32(1e-6))): |det|=1.00000011e-6 >= 1e-6 -> computed. "
         "a1=nx1*4 (exact, power of two), x=(0-(-1)*(4 nx1))/nx1=4 exactly, y=(nx1*0-0)/(-nx1)=-0.",
         direct=[[1, 0], [1, F1E6_UP]], coords=[[0, 0], [4, 0]], pair=[0, 1], expect=[4, 0]),
    dict(name="same_pixel_twice", why="t0 == t1: det_y = nx*ny - nx*ny = 0 -> skipped -> (0,0).",
         direct=[[1, 0], [0, 1]], coords=[[7, 9], [4, 0]], pair=[0, 0], expect=[0, 0]),
    dict(name="antiparallel", why="d0=(1,0), d1=(-1,0): n0=(0,-1), n1=(0,1): det_y=0*(-1)-0*1=0 -> skipped.",
         direct=[[1, 0], [-1, 0]], coords=[[0, 3], [8, 3]], pair=[0, 1], expect=[0, 0]),
]

vote_cases = [
    dict(name="points_at_it", why="n=(1,0), c=(0,0), h=(5,0): d=(5,0), norm1=1, norm2=5, cos=5/5=1 > 0.999.",
         vote=[1, 0], c=[0, 0], h=[5, 0], thresh=0.999, inlier=1),
    dict(name="points_away", why="n=(-1,0): cos=-5/5=-1: the test is signed (no fabs): not > 0.999.",
         vote=[-1, 0], c=[0, 0], h=[5, 0], thresh=0.999, inlier=0),
    dict(name="negative_cos_vs_negative_threshold", why="same pair, thresh=-1.5: -1 > -1.5 -> inlier (strict signed compare only).",
         vote=[-1, 0], c=[0, 0], h=[5, 0], thresh=-1.5, inlier=1),
    dict(name="cos_equals_thresh", why="n=(1,0), d=(3,4): norm2=sqrt(9+16)=5 exactly, cos=fl(3/5)=f32(0.6); thresh=f32(0.6): "
         "equal is NOT an inlier (strict >).", vote=[1, 0], c=[0, 0], h=[3, 4], thresh=F06, inlier=0),
    dict(name="cos_one_ulp_above_thresh", why="same pair, thresh=nextafter(f32(0.6), 0): f32(0.6) > thresh -> inlier.",
         vote=[1, 0], c=[0, 0], h=[3, 4], thresh=F06_DOWN, inlier=1),
    dict(name="cos_one_equals_thresh_one", why="cos=1 exactly, thresh=1: not an inlier.",
         vote=[1, 0], c=[0, 0], h=[2, 0], thresh=1.0, inlier=0),
    dict(name="norm1_2^-20_skipped", why="vote (2^-20,0): norm1=sqrt(2^-40)=2^-20=9.54e-7 < 1e-6 -> return, although it points at h.",
         vote=[P20, 0], c=[0, 0], h=[5, 0], thresh=0.999, inlier=0),
    dict(name="norm1_2^-19_counted", why="vote (2^-19,0): norm1=2^-19 >= 1e-6; cos=(5*2^-19)/(2^-19*5)=1 > 0.999 (all exact).",
         vote=[P19, 0], c=[0, 0], h=[5, 0], thresh=0.999, inlier=1),
    dict(name="zero_vote", why="vote (0,0): norm1=0 < 1e-6 -> skipped.", vote=[0, 0], c=[5, 5], h=[1, 1], thresh=0.999, inlier=0),
    dict(name="pixel_on_hypothesis", why="d=(0,0): norm2=0 < 1e-6 -> skipped.", vote=[1, 0], c=[3, 0], h=[3, 0], thresh=0.999, inlier=0),
    dict(name="norm2_2^-20_skipped", why="c=(1,0), h=(1+2^-20,0) (representable): d=(2^-20,0), norm2=2^-20 < 1e-6 -> skipped.",
         vote=[1, 0], c=[1, 0], h=[1 + P20, 0], thresh=0.999, inlier=0),
    dict(name="norm2_2^-19_counted", why="h=(1+2^-19,0): d=(2^-19,0), norm2=2^-19 >= 1e-6, cos=2^-19/(1*2^-19)=1 > 0.999.",
         vote=[1, 0], c=[1, 0], h=[1 + P19, 0], thresh=0.999, inlier=1),
    dict(name="unnormalised_vote", why="vote (0,3), d=(0,2): norm1=3, norm2=2, cos=(0+6)/(3*2)=1 > 0.999: |vote| cancels.",
         vote=[0, 3], c=[4, 4], h=[4, 6], thresh=0.999, inlier=1),
    dict(name="perpendicular", why="vote (0,1), d=(5,0): cos=0/(1*5)=0: not > 0.999; with thresh=-0.5: 0 > -0.5 -> inlier.",
         vote=[0, 1], c=[0, 0], h=[5, 0], thresh=0.999, inlier=0),
    dict(name="perpendicular_negative_thresh", why="as above with thresh=-0.5.", vote=[0, 1], c=[0, 0], h=[5, 0], thresh=-0.5, inlier=1),
]

for c in hyp_cases + vote_cases:                                 # every stored number must be an exact float32
    for k in ("direct", "coords", "expect", "vote", "c", "h"):
        if k in c:
            a = np.asarray(c[k], dtype=np.float64)
            assert np.array_equal(a, a.astype(np.float32).astype(np.float64)), (c["name"], k)

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_kat.json")
with open(out, "w") as f:
    json.dump({"source": "RV/src/ransac_voting_kernel.cu:22-48,100-125 — hand-derived, see make_kernel_kat.py",
               "generate_hypothesis": hyp_cases, "voting_for_hypothesis": vote_cases}, f, indent=1)
print("wrote", out, len(hyp_cases), len(vote_cases))
