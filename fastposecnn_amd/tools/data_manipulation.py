"""The pieces of F/tools/data_manipulation.py the ground-truth half of a dataset item needs (tools/dataset.py):
extract_xyz_R_T_from_RTs (:962-997) and what it calls — the pin-hole projection of the object's origin (:878-935), z from the
inverse transform (:999-1003), the translation vector through the inverse intrinsics (:1017-1050).  Host-side numpy in
float64 like the reference; the f32 cast of the projected origin inside create_translation_vector is the reference's."""
import numpy as np


def cartesian_2_homogeneous_coord(cartesian_coord):
    """[3, N] -> [4, N]"""
    return np.vstack([cartesian_coord, np.ones((1, cartesian_coord.shape[1]), dtype=cartesian_coord.dtype)])


def homogeneous_2_cartesian_coord(homogeneous_coord):
    """[K, N] -> [K - 1, N], divided by the last row"""
    return homogeneous_coord[:-1, :] / homogeneous_coord[-1, :]


def transform_3d_camera_coords_to_2d_quantized_projections(cartesian_camera_coordinates_3d, RT, intrinsics):
    """[3, N] camera-frame points, RT [4, 4], intrinsics [3, 3] -> int32 [N, 2] pixel projections (x, y); the reference's
    "method 2": inverse transform, K [I | 0], perspective division, truncation to int32 (:925)."""
    homogeneous_camera_coordinates_3d = cartesian_2_homogeneous_coord(cartesian_camera_coordinates_3d)
    K_matrix = np.hstack([intrinsics, np.zeros((intrinsics.shape[0], 1), dtype=np.float32)])
    homogeneous_world_coordinates_3d = np.linalg.inv(RT) @ homogeneous_camera_coordinates_3d
    homogeneous_projections_2d = K_matrix @ homogeneous_world_coordinates_3d
    cartesian_projections_2d = homogeneous_2_cartesian_coord(homogeneous_projections_2d)
    cartesian_projections_2d = cartesian_projections_2d.astype(np.int32)
    return cartesian_projections_2d.transpose()


def extract_z_from_RT(RT):
    return np.linalg.inv(RT)[2, 3] * 1000


def create_translation_vector(cartesian_projections_2d_xy_origin, z, intrinsics):
    """projection [2, 1] of the origin, its depth z (mm), intrinsics -> translation vector [3, 1] (metres)"""
    p = cartesian_projections_2d_xy_origin.astype(np.float32)
    p[0, :] = p[0, :] * (z / 1000)
    p[1, :] = p[1, :] * (z / 1000)
    homogeneous = np.vstack([p, z / 1000])
    return np.linalg.inv(intrinsics) @ homogeneous


def extract_xyz_R_T_from_RTs(RTs, intrinsics):
    n = len(RTs)
    xy, z, R, T = np.zeros((n, 2)), np.zeros((n, 1)), np.zeros((n, 3, 3)), np.zeros((n, 3))
    for i in range(n):
        xyz_axis = 0.3 * np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [1, 0, 0]]).transpose()
        projected = transform_3d_camera_coords_to_2d_quantized_projections(xyz_axis, RTs[i], intrinsics)
        xy[i] = np.flip(projected[0])
        z[i] = extract_z_from_RT(RTs[i])
        origin = projected[0, :].reshape((-1, 1))
        T[i] = create_translation_vector(origin, z[i], intrinsics).T
        R[i] = np.array(RTs[i])[:3, :3]
    return {'xy': xy, 'z': z, 'R': R, 'T': T}
