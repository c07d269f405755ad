"""The interpolation stack of the hot path pinned against the REFERENCE's own code:
tests/golden/patch_interp_ref.npz holds what pixsfm's featurepatch.h (image -> patch coordinates incl. the half-pixel shift,
corner, scale and upsampling factor), patch_interpolator.h (Evaluate / EvaluateLocal / CheckBounds), interpolation.h
(BiCubicInterpolator::EvaluateSIMD, PixelInterpolator's L2 normalisation and its chain rule, the Jet bridge) and
util/src/math.h return for seeded patches and keypoints, compiled in place (tests/golden/make_golden_patch_interp.py,
oracle/ref_interp_shim.cc).  Checked here: the oracle's C restatement (oracle/pxo_interp.c) on the CPU and the HIP kernel
behind pxr_interpolate on the GPU.  The in-place build replaces Eigen's norm() / dot() by left-to-right loops (Eigen is
absent), which is also the oracle's order, hence the tight bound; a build with real Eigen differs by rounding."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-13           # relative to the largest entry of the vector compared


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_patch_interp", os.path.join(HERE, "golden", "make_golden_patch_interp.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _close(a, b, tol=TOL):
    return np.abs(a - b).max() <= tol * max(1e-300, np.abs(b).max())


def test_oracle_patch_eval_matches_the_reference_vectors():
    import pxo
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "patch_interp_ref.npz"))
    n_out = n_exact = 0
    for c in gen.cases():
        n = c["name"]
        patch = pxo.make_patch(c["data"], c["corner"], c["scale"], c["up"])
        cfg = pxo.cfg(c["l2"], c["float_simd"], c["check_bounds"])
        f, gx, gy, inside = pxo.patch_eval(patch, c["xy"], cfg)
        assert int(inside) == int(gold[n + "_inside"][0]), n
        assert _close(f, gold[n + "_f"]) and _close(gx, gold[n + "_gx"]) and _close(gy, gold[n + "_gy"]), n
        n_exact += int(np.array_equal(f, gold[n + "_f"]) and np.array_equal(gx, gold[n + "_gx"]) and np.array_equal(gy, gold[n + "_gy"]))
        fv, _, _, inside_v = pxo.patch_eval(patch, c["xy"], cfg, want_grad=False)
        assert np.array_equal(fv, f) and inside_v == inside
        n_out += 1 - int(inside)
    assert n_out >= 5                                  # CheckBounds cases are in the set
    assert n_exact >= len(gen.cases()) * 9 // 10       # same operation order: bit-identical almost everywhere


def test_oracle_local_eval_with_cross_derivative_matches_the_reference_vectors():
    import pxo
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "patch_interp_ref.npz"))
    for c in gen.cases():
        n = c["name"]
        patch = pxo.make_patch(c["data"])
        cfg = pxo.cfg(c["l2"], c["float_simd"], c["check_bounds"])
        f, dr, dc, drc = pxo.pixel_interp_cross(patch, float(c["uv"][1]), float(c["uv"][0]), cfg)
        assert _close(f, gold[n + "_lf"]) and _close(dr, gold[n + "_ldr"]) and _close(dc, gold[n + "_ldc"]), n
        assert _close(drc, gold[n + "_ldrc"]), n


def test_reference_run_live_when_present():
    import pxo
    gen = _gen()
    if not os.path.exists(gen.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_interp.so not built (reference tree absent)")
    gold = np.load(os.path.join(HERE, "golden", "patch_interp_ref.npz"))
    for c in gen.cases()[:24]:                         # the committed vectors are what this build returns
        f, gx, gy, inside = gen.run_patch_eval(c)
        n = c["name"]
        assert np.array_equal(f, gold[n + "_f"]) and np.array_equal(gx, gold[n + "_gx"]) and inside == int(gold[n + "_inside"][0])
    rng = np.random.default_rng(9)
    for k in range(40):                                # fresh random draws against the oracle
        dt = [np.float16, np.float32, np.float64][k % 3]
        data = rng.normal(0, 1, (16, 16, 128)).astype(dt)
        corner = (int(rng.integers(0, 900)), int(rng.integers(0, 900)))
        scale = (float(rng.uniform(0.25, 1.0)), float(rng.uniform(0.25, 1.0)))
        uv = rng.uniform(-1.5, 17.5, 2)
        c = dict(data=data, corner=corner, scale=scale, up=1.0, uv=uv, l2=bool(k % 2), float_simd=bool(k % 7 == 0), check_bounds=True,
                 xy=np.array([(uv[0] + corner[0] + 0.5) / scale[0], (uv[1] + corner[1] + 0.5) / scale[1]]))
        f, gx, gy, inside = gen.run_patch_eval(c)
        of, ogx, ogy, oin = pxo.patch_eval(pxo.make_patch(data, corner, scale, 1.0), c["xy"], pxo.cfg(c["l2"], c["float_simd"], True))
        assert int(oin) == inside and _close(of, f) and _close(ogx, gx) and _close(ogy, gy), k


@pytest.mark.gpu
def test_hip_interpolate_matches_the_reference_vectors():
    """pxr_interpolate (descriptor + Jacobian with respect to the keypoint) against the reference-generated vectors.  Feature
    arenas have no upsampling factor (only cost maps do, costmap_extractor.h:399), so those cases are left to the oracle."""
    from pixsfm_amd import engine
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "patch_interp_ref.npz"))
    ctx = engine.Context(0)
    groups = {}
    for c in gen.cases():
        if c["up"] != 1.0:
            continue
        key = (c["data"].dtype, c["data"].shape, c["l2"], c["float_simd"], c["check_bounds"])
        groups.setdefault(key, []).append(c)
    n_checked = 0
    for (dt, shape, l2, fs, cb), cs in groups.items():
        H, W, C = shape
        arena = engine.PatchArena(ctx, len(cs), H, W, C, dt)
        arena.upload(0, np.stack([c["data"] for c in cs]), np.array([c["corner"] for c in cs], np.int32),
                     np.array([c["scale"] for c in cs], np.float64))
        cfg = engine.interp_cfg(l2_normalize=l2, use_float_simd=fs, check_bounds=cb)
        desc, J = engine.interpolate(ctx, arena, cfg, np.stack([c["xy"] for c in cs]), np.arange(len(cs)), jacobian=True)
        for i, c in enumerate(cs):
            n = c["name"]
            tol = 1e-9 if fs else 1e-11
            assert _close(desc[i], gold[n + "_f"], tol), (n, np.abs(desc[i] - gold[n + "_f"]).max())
            assert _close(J[i, :, 0], gold[n + "_gx"], tol) and _close(J[i, :, 1], gold[n + "_gy"], tol), n
            n_checked += 1
    assert n_checked >= 40
