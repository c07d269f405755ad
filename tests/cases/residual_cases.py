"""Seeded inputs for the residual functors (SURVEY 8a rows A7-A10 and the Jet bridge A4): KA edges (two patches, two keypoints,
a reference descriptor for the unary term) and BA observations (five camera models, un-normalised quaternions, scaled patches,
one projection in eight outside its patch).  Inputs only: expected values come from the oracle at test time."""
import numpy as np

NUM_PARAMS = [3, 4, 4, 5, 8]      # SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV
PS, CH = 16, 128


def _smooth_patch(rng, dtype):
    """A band-limited random field, so that the keypoint Jacobians are well scaled."""
    yy, xx = np.meshgrid(np.arange(PS), np.arange(PS), indexing="ij")
    out = np.zeros((PS, PS, CH))
    for _ in range(4):
        fx, fy = rng.uniform(0.05, 0.45, 2)
        ph = rng.uniform(0, 2 * np.pi, CH)
        out += rng.normal(0, 1, CH) * np.cos(2 * np.pi * (fx * xx[..., None] + fy * yy[..., None]) + ph)
    return out.astype(dtype)


def ka_cases(seed=577215):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(36):
        dt = [np.float16, np.float16, np.float32, np.float64][k % 4]
        c = dict(name="ka%02d" % k, l2=bool(k % 3 != 2), float_simd=bool(k % 6 == 1))
        for side in ("1", "2"):
            c["d" + side] = _smooth_patch(rng, dt)
            c["c" + side] = np.array([rng.integers(0, 1500), rng.integers(0, 1500)], np.int32)
            c["s" + side] = rng.uniform(0.25, 1.0, 2) if k % 2 else np.ones(2)
            uv = rng.uniform(-0.5, 16.5, 2) if k % 9 == 0 else rng.uniform(1.5, 14.5, 2)     # some in the clamped border band
            c["kp" + side] = (uv + c["c" + side] + 0.5) / c["s" + side]
        ref = rng.normal(0, 1, CH)
        c["ref"] = ref / np.linalg.norm(ref)
        out.append(c)
    return out


def _rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def ba_cases(seed=141421):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(40):
        model = k % 5
        f = rng.uniform(400, 1200); cx, cy = rng.uniform(300, 900, 2)
        params = {0: [f, cx, cy], 1: [f, 1.07 * f, cx, cy], 2: [f, cx, cy, 0.06], 3: [f, cx, cy, 0.06, -0.012],
                  4: [f, 1.04 * f, cx, cy, 0.05, -0.01, 1e-3, -2e-3]}[model]
        params = np.array(params)
        c = dict(name="ba%02d" % k, model=model, params=params, d=_smooth_patch(rng, np.float16), l2=bool(k % 4 != 3),
                 check_bounds=bool(k % 2))
        c["s"] = rng.uniform(0.25, 1.0, 2) if k % 3 else np.ones(2)
        pc = np.array([cx, cy])                                                              # keypoints within 250 px of the principal point
        c["c"] = np.floor((pc + rng.uniform(-250, 250, 2)) * c["s"]).astype(np.int32)
        q = rng.normal(0, 1, 4)
        c["q"] = q / np.linalg.norm(q) * (1.0 if k % 2 else rng.uniform(0.9, 1.1))       # QuaternionRotatePoint normalises
        c["X"] = rng.normal(0, 1, 3)
        uv = rng.uniform(1.5, 14.5, 2) if k % 8 else np.array([16.7, 3.2])                  # one in eight outside the patch
        xy = (uv + c["c"] + 0.5) / c["s"]
        fx, fy = params[0], (params[1] if model in (1, 4) else params[0])
        pcx, pcy = (params[2], params[3]) if model in (1, 4) else (params[1], params[2])
        z = rng.uniform(2.0, 8.0)
        un = np.array([(xy[0] - pcx) / fx, (xy[1] - pcy) / fy])
        for _ in range(20):                                                                 # undo the distortion roughly
            r2 = un @ un
            if model == 2: d = un * params[3] * r2
            elif model == 3: d = un * (params[3] * r2 + params[4] * r2 * r2)
            elif model == 4:
                k1, k2, p1, p2 = params[4:8]
                rad = k1 * r2 + k2 * r2 * r2
                d = np.array([un[0] * rad + 2 * p1 * un[0] * un[1] + p2 * (r2 + 2 * un[0] ** 2),
                              un[1] * rad + 2 * p2 * un[0] * un[1] + p1 * (r2 + 2 * un[1] ** 2)])
            else: d = np.zeros(2)
            un = np.array([(xy[0] - pcx) / fx, (xy[1] - pcy) / fy]) - d
        c["t"] = np.array([un[0] * z, un[1] * z, z]) - _rot(c["q"]) @ c["X"]
        ref = rng.normal(0, 1, CH)
        c["ref"] = ref / np.linalg.norm(ref)
        out.append(c)
    return out
