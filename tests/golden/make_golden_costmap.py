"""Generates tests/golden/costmap_ref.npz by running the REFERENCE's own CostMapExtractor::FillPointCostmap
(pixsfm/bundle_adjustment/src/costmap_extractor.h:230-358, compiled in place into oracle/_ref/libpxo_ref_costmap.so by
oracle/Makefile; the loss functions under it are restated from the published Ceres formulas) on seeded patches: the branch
without interpolation (raw texels, storage-type central differences) and the interpolating branch (upsampling factor != 1,
cross derivative), every loss / sqrt / channel-count combination, fp16 / fp32 / fp64 storage.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_costmap.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_costmap.so")
DT = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
LOSS = {"trivial": 0, "cauchy": 1, "huber": 2}
CH = 128


def _p(a):
    return C.c_void_p(a.ctypes.data)


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


def run_reference(c):
    lib = C.CDLL(LIB)
    H, W, _ = c["patch"].shape
    Ho, Wo, co = out_shape(c)
    out = np.zeros((Ho, Wo, co), c["out_dtype"])
    patch = np.ascontiguousarray(c["patch"])
    if patch.shape[2] == 3:
        rc = lib.pxo_ref_fill_point_costmap3(_p(patch), DT[patch.dtype], H, W, _p(c["ref"]), _p(out), DT[out.dtype], int(c["grad"]),
                                             int(c["sqrt"]), LOSS[c["loss"][0]], C.c_double(c["loss"][1]))
    else:
        rc = lib.pxo_ref_fill_point_costmap(_p(patch), DT[patch.dtype], H, W, _p(c["ref"]), _p(out), DT[out.dtype], Ho, Wo,
                                            C.c_double(c["up"]), int(c["grad"]), int(c["cross"]), int(c["sqrt"]), LOSS[c["loss"][0]],
                                            C.c_double(c["loss"][1]), int(c["l2"]))
    assert rc == 0, (patch.dtype, out.dtype)
    return out


if __name__ == "__main__":
    store = {}
    for c in cases():
        out = run_reference(c)
        store[c["name"]] = out.view(np.uint16) if out.dtype == np.float16 else out
    path = os.path.join(HERE, "costmap_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "cost maps")
