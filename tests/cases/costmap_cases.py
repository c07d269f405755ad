"""Seeded patches for cost-map extraction (bundle_adjustment/src/costmap_extractor.h:230-358): the branch without
interpolation (raw texels, storage-type central differences) and the interpolating branch (upsampling factor != 1, cross
derivative), every loss / sqrt / channel-count combination, fp16 / fp32 / fp64 storage.  Inputs only."""
import numpy as np

CH = 128


def cases():
    rng = np.random.default_rng(662607)
    out = []
    for k in range(48):                                   # the branch without interpolation
        dt = [np.float16, np.float16, np.float32, np.float64][k % 4]
        H, W = [(16, 16), (8, 8), (16, 16), (10, 12)][(k // 4) % 4]
        base = rng.normal(0, 1, CH); base /= np.linalg.norm(base)
        patch = (base + rng.normal(0, [0.3, 0.05][k % 2], (H, W, CH))).astype(dt)
        if k % 5 == 0:
            patch[2, 3] = base.astype(dt)                  # cost ~ 0: the `cost > 1e-8` gate
        out.append(dict(name="cm%02d" % k, patch=patch, ref=base.copy(),
                        loss=[("trivial", 1.0), ("cauchy", 0.25), ("huber", 0.5)][k % 3],
                        grad=bool(k % 8 != 7), sqrt=bool((k // 2) % 2), out_dtype=np.dtype(dt if k % 11 or dt is not np.float16 else np.float64),
                        up=1.0, cross=False, l2=True))
    for k in range(16):                                   # the interpolating branch
        dt = [np.float16, np.float64][k % 2]
        patch = rng.normal(0, 1, (8, 8, CH)).astype(dt)
        ref = rng.normal(0, 1, CH); ref /= np.linalg.norm(ref)
        grad = bool(k % 5 != 4)
        up = [2.0, 1.0, 1.5, 0.5][k % 4]
        cross = grad and (up == 1.0 or k % 3 == 0)
        if not grad and up == 1.0:
            up = 2.0
        out.append(dict(name="ci%02d" % k, patch=patch, ref=ref, loss=[("trivial", 1.0), ("cauchy", 0.25)][k % 2], grad=grad,
                        sqrt=bool(k % 2), out_dtype=np.dtype(np.float64 if k % 4 else dt), up=up, cross=cross, l2=bool(k % 3)))
    rng3 = np.random.default_rng(299792)                  # CHANNELS = 3, the extractor's other registered case (raw-texel branch)
    for k in range(18):
        dt = [np.float16, np.float32, np.float64][k % 3]
        H, W = [(16, 16), (8, 8), (9, 13)][(k // 3) % 3]
        base = rng3.uniform(0.1, 0.9, 3)
        patch = (base + rng3.normal(0, [0.2, 0.03][k % 2], (H, W, 3))).astype(dt)
        if k % 4 == 0:
            patch[1, 2] = base.astype(dt)
        out.append(dict(name="c3_%02d" % k, patch=patch, ref=(base.astype(dt).astype(np.float64) if k % 4 == 0 else base.copy()),
                        loss=[("trivial", 1.0), ("cauchy", 0.25), ("huber", 0.1)][(k // 2) % 3], grad=bool(k % 6 != 5),
                        sqrt=bool((k // 3) % 2), out_dtype=np.dtype(dt if k % 7 or dt is not np.float16 else np.float32),
                        up=1.0, cross=False, l2=False))
    return out


def out_shape(c):
    H, W, _ = c["patch"].shape
    co = (4 if c["cross"] else 3) if c["grad"] else 1
    return int(H * (c["up"] + 1e-6)), int(W * (c["up"] + 1e-6)), co          # costmap_extractor.h:385-390
