"""Generates tests/golden/residuals_ref.npz by running the REFERENCE's own residual functors -- residuals/src/featuremetric.h
(FeatureMetric2DCostFunctor: KA edge), residuals/src/feature_reference.h (FeatureReference2DCostFunctor: unary reference term,
FeatureReferenceCostFunctor / FeatureReferenceConstantPoseCostFunctor: BA), base/src/projection.h (WorldToPixel) -- compiled in
place into oracle/_ref/libpxo_ref_residual.so (oracle/ref_residual_shim.cc) and differentiated with one dual number per parameter
like ceres::AutoDiffCostFunction.  Underneath them the rotation, the camera models and the dual number are stubs restated from
the published Ceres / COLMAP definitions (oracle/ref_stubs/interp/), so the vectors pin the functors' COMPOSITION and the
interpolation stack, not COLMAP's camera models.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_residuals.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_residual.so")
DT = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
NUM_PARAMS = [3, 4, 4, 5, 8]      # SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV
PS, CH = 16, 128


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


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


def run_ka(c):
    lib = C.CDLL(LIB)
    r, J1, J2 = np.empty(CH), np.empty((CH, 2)), np.empty((CH, 2))
    ok = lib.pxo_ref_ka_edge(_p(c["d1"]), _p(c["d2"]), DT[c["d1"].dtype], PS, PS, _p(c["c1"]), _p(c["s1"]), _p(c["c2"]), _p(c["s2"]),
                             int(c["l2"]), int(c["float_simd"]), 0, _p(c["kp1"]), _p(c["kp2"]), _p(r), _p(J1), _p(J2))
    assert ok == 1
    r2, Jk = np.empty(CH), np.empty((CH, 2))
    ok = lib.pxo_ref_ref2d(_p(c["d1"]), DT[c["d1"].dtype], PS, PS, _p(c["c1"]), _p(c["s1"]), int(c["l2"]), int(c["float_simd"]), 0,
                           _p(c["kp1"]), _p(c["ref"]), _p(r2), _p(Jk))
    assert ok == 1
    return r, J1, J2, r2, Jk


def run_ba(c, const_pose):
    lib = C.CDLL(LIB)
    K = NUM_PARAMS[c["model"]]
    r, J = np.empty(CH), np.empty((CH, (3 if const_pose else 10) + K))
    ok = lib.pxo_ref_ba_residual(c["model"], int(const_pose), _p(c["d"]), PS, PS, _p(c["c"]), _p(c["s"]), int(c["l2"]), 0,
                                 int(c["check_bounds"]), _p(c["q"]), _p(c["t"]), _p(c["X"]), _p(c["params"]), _p(c["ref"]), _p(r), _p(J))
    assert ok in (0, 1)
    return r, J, ok


if __name__ == "__main__":
    store = {}
    for c in ka_cases():
        r, J1, J2, r2, Jk = run_ka(c)
        n = c["name"]
        store[n + "_r"], store[n + "_J1"], store[n + "_J2"], store[n + "_r2d"], store[n + "_J2d"] = r, J1, J2, r2, Jk
    n_false = 0
    for c in ba_cases():
        r, J, ok = run_ba(c, False)
        rc, Jc, okc = run_ba(c, True)
        K = NUM_PARAMS[c["model"]]
        assert ok == okc and np.array_equal(r, rc) and np.array_equal(J[:, 7:], Jc)      # constant pose = the point / camera columns
        n = c["name"]
        store[n + "_r"], store[n + "_J"], store[n + "_ok"] = r, J, np.array([ok])
        n_false += 1 - ok
    path = os.path.join(HERE, "residuals_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(ka_cases()), "KA edges,", len(ba_cases()), "BA residuals,", n_false, "with the functor returning false")
