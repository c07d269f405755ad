"""The interpolation stack of the hot path on 96 seeded evaluations (tests/cases/patch_interp_cases.py:
fp16 / fp32 / fp64 patches, 128 / 64 channels, three patch shapes, scaled patches, upsampling factors, the clamped border band,
points exactly on a texel, points outside the patch for CheckBounds).

CPU: the oracle's patch evaluation (oracle/pxo_interp.c) against the parts of it that have a closed form -- the image -> patch
coordinate map of features/src/featurepatch.h:250-255, the chain rule through it, CheckBounds of
features/src/patch_interpolator.h:160-166, unit norm after the L2 normalisation (interpolation.h:648-666).  The reference's own
known-answer cases for the bicubic itself are in tests/test_oracle_interp.py.
GPU: the HIP kernel behind pxr_interpolate against the oracle on the same inputs.

PARITY UNPINNED beyond those closed forms: the reference's featurepatch.h / patch_interpolator.h / interpolation.h cannot be
compiled in this image (Eigen, Ceres, COLMAP, HighFive, Boost are absent), so the oracle's values are a restatement
of the source, not outputs of the reference."""
import numpy as np
import pytest

from cases import patch_interp_cases as gen


def _close(a, b, tol):
    return np.abs(a - b).max() <= tol * max(1e-300, np.abs(b).max())


def test_oracle_patch_eval_is_the_local_evaluation_through_the_coordinate_map():
    import pxo
    n_out = 0
    for c in gen.cases():
        patch = pxo.make_patch(c["data"], c["corner"], c["scale"], c["up"])
        H, W, _ = c["data"].shape
        cfg = pxo.cfg(c["l2"], c["float_simd"], c["check_bounds"])
        f, gx, gy, inside = pxo.patch_eval(patch, c["xy"], cfg)
        # featurepatch.h:250-255: u = (x sx - 0.5 - x0) up, v = (y sy - 0.5 - y0) up
        u = (c["xy"][0] * c["scale"][0] - 0.5 - c["corner"][0]) * c["up"]
        v = (c["xy"][1] * c["scale"][1] - 0.5 - c["corner"][1]) * c["up"]
        lf, ldr, ldc = pxo.pixel_interp(pxo.make_patch(c["data"]), v, u, cfg)
        assert _close(f, lf, 1e-15), c["name"]
        # Jet bridge (interpolation.h:130-140): d/dx = df/dc * sx * up, d/dy = df/dr * sy * up
        assert _close(gx, ldc * c["scale"][0] * c["up"], 1e-14) and _close(gy, ldr * c["scale"][1] * c["up"], 1e-14), c["name"]
        # patch_interpolator.h:160-166: inside <=> 0 < u < W and 0 < v < H (only looked at under check_bounds)
        want_inside = 1 if not c["check_bounds"] else int(0 < u < W and 0 < v < H)
        assert int(inside) == want_inside, c["name"]
        if c["l2"]:
            assert abs(f @ f - 1.0) < 1e-12 and abs(f @ gx) < 1e-10 * max(1.0, np.abs(gx).max()), c["name"]
        fv, _, _, inside_v = pxo.patch_eval(patch, c["xy"], cfg, want_grad=False)
        assert np.array_equal(fv, f) and inside_v == inside
        n_out += 1 - int(inside)
    assert n_out >= 5                                  # CheckBounds cases are in the set


def test_oracle_cross_derivative_is_the_derivative_of_the_column_derivative():
    """PixelInterpolator::Evaluate with the cross-derivative pointer (interpolation.h:642-677) on the fp64 cases without
    normalisation: d2f/drdc against a central difference of df/dc along r."""
    import pxo
    n = 0
    for c in gen.cases():
        if c["data"].dtype != np.float64 or c["l2"] or c["float_simd"]:
            continue
        patch = pxo.make_patch(c["data"])
        cfg = pxo.cfg(False, False, False)
        u, v = float(c["uv"][0]), float(c["uv"][1])
        if abs(v - round(v)) < 1e-3 or not (1.5 < v < c["data"].shape[0] - 2.5):
            continue                                    # a knot of the spline within the stencil of the difference
        f, dr, dc, drc = pxo.pixel_interp_cross(patch, v, u, cfg)
        e = 1e-6
        dcp = pxo.pixel_interp(patch, v + e, u, cfg)[2]
        dcm = pxo.pixel_interp(patch, v - e, u, cfg)[2]
        assert np.abs((dcp - dcm) / (2 * e) - drc).max() < 1e-6 * max(1.0, np.abs(drc).max()), c["name"]
        n += 1
    assert n >= 3


@pytest.mark.gpu
def test_hip_interpolate_matches_the_oracle():
    """pxr_interpolate (descriptor + Jacobian with respect to the keypoint) against the oracle.  Feature arenas have no
    upsampling factor (only cost maps do, costmap_extractor.h:399), so those cases are left out."""
    import pxo
    from pixsfm_amd import engine
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
            f, gx, gy, _ = pxo.patch_eval(pxo.make_patch(c["data"], c["corner"], c["scale"], 1.0), c["xy"], pxo.cfg(l2, fs, cb))
            tol = 1e-9 if fs else 1e-11
            assert _close(desc[i], f, tol), (c["name"], np.abs(desc[i] - f).max())
            assert _close(J[i, :, 0], gx, tol) and _close(J[i, :, 1], gy, tol), c["name"]
            n_checked += 1
    assert n_checked >= 40
