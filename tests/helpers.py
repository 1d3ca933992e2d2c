"""Shared input generators for the parity tests (seeded; the same arrays go to the oracle and to the GPU)."""
import numpy as np


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def random_rays(rng, n, origin_box=3.0, tmax_inf_fraction=0.7, target=None):
    """n x {o[3], d[3], tmax}: origins in a box, directions towards `target` points (or random) so many rays hit."""
    o = rng.uniform(-origin_box, origin_box, (n, 3))
    if target is None:
        d = unit(rng.normal(size=(n, 3)))
    else:
        d = unit(target - o)
    tmax = np.where(rng.uniform(size=n) < tmax_inf_fraction, np.inf, rng.uniform(0.0, 6.0, n))
    return np.concatenate([o, d, tmax[:, None]], 1).astype(np.float32)


def hemisphere_dirs(rng, n):
    return unit(rng.normal(size=(n, 3)))


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
