"""Ground-truth pose pieces of a dataset item (tools/dataset.py: generate_agg_data): what the reference derives from the
side file's RT matrices in F/tools/data_manipulation.py:962-997 (extract_xyz_R_T_from_RTs, with :878-935 projection,
:999-1003 depth, :1017-1050 translation).  Here ONE batched evaluation over all instances of a frame, host-side numpy in
float64 like the reference; `tests/test_dataset_gt.py` pins the values against the reference's own class.

Per instance, with M = RT^-1 (camera <- object) and K the intrinsics:
  origin   o = M[:3, 3]                       the object's origin in the camera frame (the reference projects the points
                                              0.3 * {0, e_z, e_y, e_x} and keeps only the first)
  pixel    (u, v) = trunc_int32((K o)[:2] / (K o)[2])
  xy       = (v, u)                           row, column — the reference flips the projection
  z        = 1000 * M[2, 3]                   millimetres
  T        = K^-1 [f32(f32(u) z'), f32(f32(v) z'), z']   z' = z / 1000; the two products are rounded to float32 as in
                                              the reference (it scales a float32 copy of the pixel in place)
  R        = RT[:3, :3]
"""
import numpy as np


def project_origins(RTs, intrinsics):
    """RTs [n,4,4], intrinsics [3,3] -> (pixels int32 [n,2] as (u = column, v = row), z [n] in mm)."""
    M = np.linalg.inv(np.asarray(RTs, dtype=np.float64).reshape(-1, 4, 4))
    cam = M[:, :3, 3]                                           # [n,3]
    proj = cam @ np.asarray(intrinsics, dtype=np.float64).T     # rows = K o
    pix = (proj[:, :2] / proj[:, 2:3]).astype(np.int32)         # truncation toward zero, as .astype(np.int32) does
    return pix, M[:, 2, 3] * 1000.0


def translations_from_pixels(pix, z_mm, intrinsics):
    """pixels int32 [n,2] (u, v), depths [n] (mm) -> T [n,3] (metres)."""
    zs = np.asarray(z_mm, dtype=np.float64) / 1000.0
    uv = (pix.astype(np.float32).astype(np.float64) * zs[:, None]).astype(np.float32).astype(np.float64)
    rays = np.concatenate([uv, zs[:, None]], axis=1)            # [n,3]
    return rays @ np.linalg.inv(np.asarray(intrinsics)).T


def extract_xyz_R_T_from_RTs(RTs, intrinsics):
    """-> {'xy' [n,2] (row, column of the projected origin), 'z' [n,1] (mm), 'R' [n,3,3], 'T' [n,3]}, float64."""
    RTs = np.asarray(RTs, dtype=np.float64).reshape(-1, 4, 4)
    pix, z = project_origins(RTs, intrinsics)
    return {'xy': pix[:, ::-1].astype(np.float64), 'z': z[:, None], 'R': RTs[:, :3, :3].copy(),
            'T': translations_from_pixels(pix, z, intrinsics)}
