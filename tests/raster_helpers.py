"""Shared scene builders for the rasterizer tests (CPU oracle side and GPU side use the same)."""
import numpy as np

from multiview_inpaint_amd import synthetic as syn


def small_scene(seed, N=80, W=40, H=36, deg=3, pose=True, log_scale=np.log(0.15), zmax=6.0):
    rng = np.random.default_rng(seed)
    R = syn.random_rotation(rng) if pose else None
    T = rng.normal(size=3) if pose else None
    cam = syn.make_camera(W, H, 50.0, R, T)
    sc = syn.make_scene(N, cam, deg, seed, log_scale_mean=log_scale, zmin=1.0, zmax=zmax)
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    return cam, sc, bg


def oracle_params(ro, cam, sc, bg, N=None, M=None, scale_modifier=1.0):
    N = sc["means3D"].shape[0] if N is None else N
    deg = sc["sh_degree"]
    M = sc["shs"].shape[1] if M is None else M
    return ro.make_params(N, deg, M, cam["W"], cam["H"], cam["tanfovx"], cam["tanfovy"], scale_modifier,
                          cam["viewmatrix"], cam["projmatrix"], cam["campos"], bg)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
