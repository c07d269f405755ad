"""Seeded inputs for the bare bicubic: the shape of the reference's TestBiCubicSimilarCeres
(pixsfm/base/src/interpolation_test.cc:327-364): 10 x 10 x 128 grids of half / float / double and positions r, c in [0, 9.9]
including the clamped border and points outside the grid.  Inputs only."""
import numpy as np


def positions():
    pos = np.stack(np.meshgrid(np.arange(0, 100, 17) / 10.0, np.arange(0, 100, 19) / 10.0, indexing="ij"), -1).reshape(-1, 2)
    return np.concatenate([pos, [[-0.7, 4.2], [9.95, 9.95], [3.0, 5.0], [0.0, 0.0], [12.3, -2.5], [9.9, 0.1]]])


def grid(name):
    dt = {"f16": np.float16, "f32": np.float32, "f64": np.float64}[name]
    rng = np.random.default_rng({"f16": 11, "f32": 12, "f64": 13}[name])
    return rng.uniform(-1, 1, (10, 10, 128)).astype(dt)
