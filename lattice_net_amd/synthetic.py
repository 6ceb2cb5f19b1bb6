"""Seeded synthetic point clouds for the BASELINE.json configs.  NumPy only."""
from __future__ import annotations

import numpy as np


def lidar_cloud(n: int, seed: int = 0) -> np.ndarray:
    """C3 / SemanticKITTI-like: range 2+58*u1*u2 m, uniform azimuth, height -1.7+0.3*N(0,1)
    (+U(0,3) for 20 % of points), capped at 60 m.  float32 [n, 3]."""
    rng = np.random.default_rng(seed)
    r = 2.0 + 58.0 * rng.random(n) * rng.random(n)
    az = rng.random(n) * 2 * np.pi
    z = -1.7 + 0.3 * rng.standard_normal(n)
    tall = rng.random(n) < 0.2
    z = z + tall * rng.random(n) * 3.0
    r = np.minimum(r, 60.0)
    return np.ascontiguousarray(np.stack([r * np.cos(az), r * np.sin(az), z], axis=1).astype(np.float32))


def cube_cloud(n: int, seed: int = 0, lo: float = -1.0, hi: float = 1.0, d: int = 3) -> np.ndarray:
    """C1: uniform cube."""
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(rng.uniform(lo, hi, (n, d)).astype(np.float32))


def box_surface_cloud(n: int, seed: int = 0) -> np.ndarray:
    """C2 / ShapeNet-like: points on the surface of a unit box + N(0, 0.001) noise."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.5, 0.5, (n, 3))
    face = rng.integers(0, 6, n)
    ax = face % 3
    p[np.arange(n), ax] = np.where(face < 3, -0.5, 0.5)
    p += 0.001 * rng.standard_normal((n, 3))
    return np.ascontiguousarray(p.astype(np.float32))


def planes_cloud(n: int, seed: int = 0) -> np.ndarray:
    """C4 / ScanNet-like: points on axis-aligned planes inside an 8 x 3 x 8 m box."""
    rng = np.random.default_rng(seed)
    ext = np.array([8.0, 3.0, 8.0])
    p = rng.random((n, 3)) * ext
    plane_axis = rng.integers(0, 3, n)
    level = rng.integers(0, 4, n) / 3.0
    p[np.arange(n), plane_axis] = level * ext[plane_axis]
    p += 0.005 * rng.standard_normal((n, 3))
    return np.ascontiguousarray(p.astype(np.float32))
