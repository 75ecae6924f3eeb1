"""Deterministic synthetic cameras and Gaussian scenes for tests and bench.

Camera conventions follow the reference (gs-simp/utils/graphics_utils.py:38-71,
gs-simp/scene/cameras.py:60-63): matrices are stored transposed (row-vector convention),
`full_proj = world_view @ projection`, `camera_center = inverse(world_view)[3, :3]`.
Everything is generated on the CPU with a seeded numpy Generator and returned as float32 numpy.
"""
import math

import numpy as np


def projection_matrix(znear, zfar, fovx, fovy):
    """Perspective matrix with z in [0,1] and P[3,2]=1 (graphics_utils.py:51-71), NOT transposed."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fovy_deg=50.0, R=None, T=None, znear=0.01, zfar=100.0):
    """Returns dict(viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H) in reference layout.
    R is camera-to-world rotation (as COLMAP loaders hand it to Camera), T world-to-camera
    translation (graphics_utils.py:38-49)."""
    fovy = math.radians(fovy_deg)
    fovx = 2 * math.atan(math.tan(fovy / 2) * W / H)
    R = np.eye(3, dtype=np.float64) if R is None else np.asarray(R, np.float64)
    T = np.zeros(3, np.float64) if T is None else np.asarray(T, np.float64)
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    view = np.float32(Rt).T.copy()                       # world_view_transform (transposed W2C)
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()
    full = (view.astype(np.float32) @ proj).astype(np.float32)
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=view, projmatrix=full, campos=campos,
                tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2), W=int(W), H=int(H))


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_scene(N, cam, sh_degree=3, seed=0, log_scale_mean=math.log(0.01), log_scale_std=0.5,
               zmin=1.0, zmax=8.0):
    """SURVEY.md §8d recipe: z~U(zmin,zmax) in camera space, x,y~U(-1.1,1.1)*z*tanfov (about 17 %
    outside the frustum), log-normal scales, random unit quaternions, sigmoid(N(0,1.5^2))
    opacities, SH dc~N(0,1), rest~N(0,0.2^2). Points are placed in camera space and moved to
    world space with the camera pose so any pose sees the same distribution."""
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    z = rng.uniform(zmin, zmax, N)
    x = rng.uniform(-1.1, 1.1, N) * z * cam["tanfovx"]
    y = rng.uniform(-1.1, 1.1, N) * z * cam["tanfovy"]
    pc = np.stack([x, y, z, np.ones(N)], 1)              # camera-space homogeneous (row vectors)
    pw = pc @ np.linalg.inv(cam["viewmatrix"].astype(np.float64))
    means3D = pw[:, :3].astype(np.float32)
    scales = np.exp(rng.normal(log_scale_mean, log_scale_std, (N, 3))).astype(np.float32)
    q = rng.normal(size=(N, 4))
    rotations = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacities = (1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, (N, 1))))).astype(np.float32)
    shs = np.concatenate([rng.normal(0, 1.0, (N, 1, 3)), rng.normal(0, 0.2, (N, M - 1, 3))], 1).astype(np.float32)
    return dict(means3D=means3D, scales=scales, rotations=rotations, opacities=opacities, shs=shs,
                sh_degree=sh_degree)
