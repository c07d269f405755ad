"""Seeded inputs for the interpolation stack: one evaluation each -- the patch, its corner / scale /
upsampling factor, the keypoint in IMAGE coordinates, the InterpolationConfig switches.  Inputs only: the expected values are
computed by the oracle at test time (tests/test_patch_interp.py)."""
import numpy as np


def cases():
    """dicts with the seeded inputs of one evaluation each: the patch, its corner / scale / upsampling factor, the keypoint
    in IMAGE coordinates, the InterpolationConfig switches."""
    rng = np.random.default_rng(1618033)
    out = []
    for k in range(96):
        dt = [np.float16, np.float32, np.float64][k % 3]
        ch = [128, 64][(k // 3) % 2]
        hw = [(16, 16), (10, 10), (8, 12)][(k // 6) % 3]
        data = rng.normal(0, 1, hw + (ch,)).astype(dt)
        corner = (int(rng.integers(0, 2000)), int(rng.integers(0, 2000)))
        scale = (float(rng.uniform(0.2, 1.6)), float(rng.uniform(0.2, 1.6))) if k % 4 else (1.0, 1.0)
        up = [1.0, 1.0, 2.0, 0.5][(k // 2) % 4]
        # local coordinates: interior, the clamped border band, exactly on a texel, outside (CheckBounds)
        kind = k % 8
        if kind < 4:
            uv = rng.uniform(1.0, min(hw) - 2.0, 2)
        elif kind == 4:
            uv = rng.uniform(-0.9, 0.9, 2)
        elif kind == 5:
            uv = np.array([hw[1] - rng.uniform(0.05, 1.0), hw[0] - rng.uniform(0.05, 1.0)])
        elif kind == 6:
            uv = np.floor(rng.uniform(1.0, min(hw) - 2.0, 2))
        else:
            uv = np.array([hw[1] + rng.uniform(0.1, 2.0), -rng.uniform(0.1, 2.0)])
        xy = np.array([(uv[0] / up + corner[0] + 0.5) / scale[0], (uv[1] / up + corner[1] + 0.5) / scale[1]])   # featurepatch.h:257-260
        out.append(dict(name="pe%02d" % k, data=data, corner=corner, scale=scale, up=up, xy=xy, uv=uv,
                        l2=bool((k // 2) % 2 == 0), float_simd=bool(k % 5 == 0), check_bounds=bool(k % 3 != 1)))
    return out
