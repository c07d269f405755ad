"""Generates tests/golden/patch_interp_ref.npz by running the REFERENCE's own interpolation stack -- features/src/featurepatch.h
(image -> patch coordinates), features/src/patch_interpolator.h (Evaluate, EvaluateLocal, CheckBounds), base/src/interpolation.h
(BiCubicInterpolator::EvaluateSIMD, PixelInterpolator's L2 normalisation + chain rule, the Jet bridge), util/src/math.h --
compiled in place into oracle/_ref/libpxo_ref_interp.so by oracle/Makefile (stub headers: oracle/ref_stubs/interp/), on seeded
patches and keypoints.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_patch_interp.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_interp.so")
DT = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


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


def run_patch_eval(c, want_grad=True):
    lib = C.CDLL(LIB)
    d = c["data"]
    H, W, ch = d.shape
    f = np.empty(ch)
    gx, gy = (np.empty(ch), np.empty(ch)) if want_grad else (None, None)
    xy = np.ascontiguousarray(c["xy"], dtype=np.float64)
    inside = lib.pxo_ref_patch_eval(_p(d), DT[d.dtype], H, W, ch, c["corner"][0], c["corner"][1], C.c_double(c["scale"][0]),
                                    C.c_double(c["scale"][1]), C.c_double(c["up"]), int(c["l2"]), int(c["float_simd"]),
                                    int(c["check_bounds"]), _p(xy), _p(f), _p(gx), _p(gy))
    assert inside >= 0
    return f, gx, gy, inside


def run_local_eval(c, cross=True):
    lib = C.CDLL(LIB)
    d = c["data"]
    H, W, ch = d.shape
    f, dr, dc = (np.empty(ch) for _ in range(3))
    drc = np.empty(ch) if cross else None
    uv = np.ascontiguousarray(c["uv"], dtype=np.float64)
    inside = lib.pxo_ref_patch_eval_local(_p(d), DT[d.dtype], H, W, ch, int(c["l2"]), int(c["float_simd"]), int(c["check_bounds"]),
                                          _p(uv), _p(f), _p(dr), _p(dc), _p(drc))
    assert inside >= 0
    return f, dr, dc, drc, inside


if __name__ == "__main__":
    store = {}
    n_out = 0
    for c in cases():
        f, gx, gy, inside = run_patch_eval(c)
        fv, _, _, inside_v = run_patch_eval(c, want_grad=False)
        assert np.array_equal(f, fv) and inside == inside_v          # value-only and Jet evaluation agree
        lf, ldr, ldc, ldrc, linside = run_local_eval(c)
        n = c["name"]
        store[n + "_f"], store[n + "_gx"], store[n + "_gy"], store[n + "_inside"] = f, gx, gy, np.array([inside])
        store[n + "_lf"], store[n + "_ldr"], store[n + "_ldc"], store[n + "_ldrc"] = lf, ldr, ldc, ldrc
        store[n + "_linside"] = np.array([linside])
        n_out += 1 - inside
    path = os.path.join(HERE, "patch_interp_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "evaluations,", n_out, "out of bounds")
