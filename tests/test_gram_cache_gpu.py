"""GPU: the Gram-matrix cache of the BA solver (csrc/pxr_ba_gram.hip; the DEFAULT evaluation of the LM loop since round 5,
pxr_set_gram_cache(ctx, 0) / PXR_GRAM_CACHE=0 opt out).

The solver consumes the 64-byte record of a residual block, and bicubic interpolation is linear in the 16 texels of the 4 x 4
stencil: with G = T T^t (16 x 16) and D = T ref the record is a set of quadratic / linear forms in the Catmull-Rom weights.
The cache path must give pxr_ba_eval's records up to the rounding of the reference's fp32 horizontal pass
(cubic_hermite_spline_simd.h; the algebra on G is exact in fp64), rebuild exactly the observations whose projection left
their cell, and steer the LM loop to the same solution.  The inner iterations keep their matrices in the same cache from
call to call in EVERY mode: that must not change a bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# a record entry differs from the exact-order kernel's by ~1e-7 of its scale (measured: 2e-8 absolute at |r|^2 ~ 0.4): the
# fp32 pass rounds every channel at 6e-8 relative
from conftest import FP32_PASS_RECORD_ATOL as REC_ATOL


def _gauge(prob):
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return pose_const, tmask, np.full(n_cam, 0b0110, np.uint16), np.zeros(n_pt, np.uint8)


def _problem(ctx, **kw):
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena
    args = dict(n_cams=10, n_points=500, obs_per_point=5, seed=21, rot_deg=0.3, pt_sigma=0.02, noise=0.01)
    args.update(kw)
    prob = synthetic.make_ba_problem(**args)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    return prob, arena, BAProblem(ctx, arena, prob)


@pytest.mark.parametrize("dtype,channels,l2,model", [(np.float16, 128, True, 2), (np.float32, 128, True, 4), (np.float16, 64, True, 1),
                                                      (np.float32, 64, False, 0), (np.float16, 128, False, 3)])
def test_records_match_the_exact_order_kernel(ctx, dtype, channels, l2, model):
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _problem(ctx, dtype=dtype, channels=channels, model=model)
    cfg = interp_cfg(l2_normalize=l2)
    exact = ba.eval(cfg)[0].download().copy()
    rec, built = ba.eval_gram(cfg, reset=True)
    gram = rec.download().copy()
    assert built == ba.n_obs                                             # an empty cache: every observation's matrices
    assert np.array_equal(gram[:, 6:], exact[:, 6:])                     # the projection is the same code
    assert np.abs(gram[:, :6] - exact[:, :6]).max() < REC_ATOL
    assert abs(gram[:, 0].sum() - exact[:, 0].sum()) < 1e-9 * exact[:, 0].sum()      # the rounding averages out over the blocks
    arena.close()


def test_only_the_observations_that_left_their_cell_are_rebuilt(ctx):
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _problem(ctx)
    cfg = interp_cfg()
    rec, built = ba.eval_gram(cfg, reset=True)
    first = rec.download().copy()
    assert built == ba.n_obs
    rec, built = ba.eval_gram(cfg, reset=False)                           # same parameters: nothing to build, the same bits
    assert built == 0 and np.array_equal(rec.download(), first)
    # cells (floor of the patch coordinates) before and after a move of the points, from the exact kernel's projections
    xy0 = first[:, 6:8]
    ba.d["xyz"].upload(prob["xyz"] + np.random.default_rng(3).normal(0, 0.002, prob["xyz"].shape))
    exact = ba.eval(cfg)[0].download().copy()
    xy1 = exact[:, 6:8]
    cell = lambda xy: np.floor(xy * prob["scales"] - 0.5 - prob["corners"]).astype(np.int64)
    moved = int((cell(xy0) != cell(xy1)).any(axis=1).sum())
    assert 0 < moved < ba.n_obs
    rec, built = ba.eval_gram(cfg, reset=False)
    assert built == moved
    assert np.abs(rec.download()[:, :6] - exact[:, :6]).max() < REC_ATOL
    # ... and back: the cells that moved are rebuilt again, the records are those of the first call
    ba.d["xyz"].upload(prob["xyz"])
    rec, built = ba.eval_gram(cfg, reset=False)
    assert built == moved and np.array_equal(rec.download(), first)
    arena.close()


def test_a_projection_that_cannot_be_evaluated_stays_nan(ctx):
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _problem(ctx, n_points=64)
    xyz = prob["xyz"].copy()
    xyz[5] = np.nan
    ba.d["xyz"].upload(xyz)
    rec, _ = ba.eval_gram(interp_cfg(), reset=True)
    r = rec.download()
    bad = prob["obs_point"] == 5
    assert np.isnan(r[bad, 0]).all() and np.isfinite(r[~bad, 0]).all()
    arena.close()


@pytest.mark.parametrize("inner", [False, True])
def test_solve_with_the_cache_is_the_solve_without(ctx, inner):
    from pixsfm_amd.engine import Context, interp_cfg, lm_options, make_loss
    c2 = Context(0)
    c2.gram_cache = False                       # the opt-out: every evaluation by the exact-order kernel (texels)
    assert ctx.gram_cache and not c2.gram_cache
    out = []
    for c in (c2, ctx):
        prob, arena, ba = _problem(c, n_cams=12, n_points=900)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob),
                     options=lm_options(max_iterations=12, use_inner_iterations=inner))
        out.append((s, ba.params()))
        arena.close()
    (s0, p0), (s1, p1) = out
    assert s0["iterations"] == s1["iterations"]
    assert abs(s0["initial_cost"] - s1["initial_cost"]) < 1e-9 * s0["initial_cost"]
    assert abs(s0["final_cost"] - s1["final_cost"]) < 1e-6 * s0["initial_cost"]
    for a, b in zip(p0, p1):
        assert np.abs(a - b).max() < 1e-4 * max(1.0, np.abs(a).max())          # north_star: poses / points within 1e-4
    c2.close()


@pytest.mark.parametrize("inner", [False, True])
def test_use_float_simd_keeps_the_texel_kernels(ctx, inner):
    """InterpolationConfig.use_float_simd asks for the reference's all-fp32 splines (interpolation.h:177-218, float instantiation).  The
    Gram-matrix paths are exact fp64 algebra: a solve with that flag must not take them -- with the cache flag on and
    off it is the same solve to the last bit (deterministic mode)."""
    from pixsfm_amd.engine import Context, interp_cfg, lm_options, make_loss
    c2 = Context(0)
    c2.gram_cache = False
    assert ctx.gram_cache and ctx.deterministic and c2.deterministic
    out = []
    for c in (c2, ctx):
        prob, arena, ba = _problem(c, n_cams=12, n_points=900)
        s = ba.solve(interp_cfg(use_float_simd=True), make_loss("cauchy", [0.25]), *_gauge(prob),
                     options=lm_options(max_iterations=8, use_inner_iterations=inner))
        out.append((s, ba.params()))
        arena.close()
    (s0, p0), (s1, p1) = out
    assert s0["num_successful"] > 2 and s0["iterations"] == s1["iterations"] and s0["num_successful"] == s1["num_successful"]
    assert s0["initial_cost"] == s1["initial_cost"] and s0["final_cost"] == s1["final_cost"]
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)
    c2.close()


def test_the_inner_iterations_cache_does_not_change_a_bit(monkeypatch):
    """Default mode: the matrices of the inner iterations' Gram-matrix kernel live in the solve's cache: what a call would rebuild
    is rebuilt ahead of it (k_gram_flag_slots + k_gram_build), the kernel copies, and writes back what it rebuilds during its
    rounds.  The copied numbers ARE the built numbers: in deterministic mode (no floating-point atomics) the solve is the same
    to the last bit with the rebuilds inside the kernel and with no cache at all."""
    from pixsfm_amd.engine import Context, interp_cfg, lm_options, make_loss
    runs = []
    for knob in ("", "PXR_INNER_NO_PREBUILD", "PXR_INNER_NO_CACHE"):   # default / stale matrices rebuilt inside the kernel / no cache at all
        monkeypatch.delenv("PXR_INNER_NO_PREBUILD", raising=False)
        if knob:
            monkeypatch.setenv(knob, "1")
        c = Context(0)
        c.deterministic = True
        prob, arena, ba = _problem(c, n_cams=12, n_points=900)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob), options=lm_options(max_iterations=10, use_inner_iterations=True))
        runs.append((s, ba.params()))
        arena.close(); c.close()
    s0, p0 = runs[0]
    assert s0["num_successful"] > 2
    for s1, p1 in runs[1:]:
        assert s0["iterations"] == s1["iterations"] and s0["num_successful"] == s1["num_successful"]
        assert s0["final_cost"] == s1["final_cost"]
        for a, b in zip(p0, p1):
            assert np.array_equal(a, b)


def test_cost_maps_and_fp64_storage_ignore_the_flag(ctx):
    """The cache needs feature patches (128 / 64 channels, fp16 / fp32) with reference descriptors: elsewhere the flag is ignored
    by pxr_ba_solve and pxr_ba_eval_gram says so."""
    from pixsfm_amd import PixsfmHipError
    from pixsfm_amd.engine import Context, interp_cfg, lm_options, make_loss
    c2 = Context(0)
    c2.gram_cache = True
    prob, arena, ba = _problem(c2, dtype=np.float64, n_points=200)
    with pytest.raises(PixsfmHipError):
        ba.eval_gram(interp_cfg())
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob), options=lm_options(max_iterations=4))
    _, arena1, ba1 = _problem(ctx, dtype=np.float64, n_points=200)
    s1 = ba1.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob), options=lm_options(max_iterations=4))
    assert abs(s["final_cost"] - s1["final_cost"]) < 1e-8 * s["initial_cost"] and s["iterations"] == s1["iterations"]
    arena.close(); arena1.close(); c2.close()


# ---- against the oracle's residual functor (the restatement of the reference's arithmetic incl. its fp32 horizontal pass) ---------
# north_star: residuals / Jacobians within 1e-5 relative of the Ceres reference.  What the Gram-matrix path produces is the
# solver's view of a block -- |r|^2, J^t J, J^t r -- so that is what is compared, against the same quantities formed from the
# ORACLE's residual and Jacobian (pxo.ba_residual: FeatureReferenceCostFunctor of residuals/src/feature_reference.h:123-134 over
# base/src/interpolation.h:177-218, restated; the reference itself cannot be compiled here -- parity unpinned).
GRAM_VS_REFERENCE_RTOL = 1e-5          # the contract
GRAM_VS_REFERENCE_SEEN = 2e-6          # what the fp32 horizontal pass of the reference actually leaves (asserted too)


def _golden():
    import test_residuals
    return test_residuals.gen_mod, test_residuals._gold()


@pytest.mark.parametrize("storage", [np.float16, np.float32])
def test_gram_records_match_the_oracles_functor(ctx, storage):
    """All five camera models of the golden set, with / without L2 normalisation, fp16 and fp32 storage (the golden patches hold
    fp16 values: as fp32 storage the reference computes the same numbers), incl. the cases whose stencil is clamped at the
    patch border.  Per block: |r|^2, J^t J ((10+K)^2) and J^t r from the record + the projection Jacobian P (k_jac, pinned in
    tests/test_residuals.py) against the oracle's r and J."""
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg
    gen, gold = _golden()
    groups = {}
    for c in gen.ba_cases():                               # (check_bounds cases too: with a reference descriptor the functor ignores
        groups.setdefault(c["l2"], []).append(c)           #  the bounds check, feature_reference.h:128-136 -- the same vectors)
    n_checked, worst = 0, 0.0
    models = set()
    for l2, cs in groups.items():
        m = len(cs)
        cam_params = np.zeros((m, 12))
        for i, c in enumerate(cs):
            cam_params[i, :len(c["params"])] = c["params"]
        ids = np.arange(m, dtype=np.int32)
        prob = dict(obs_image=ids, obs_point=ids, obs_patch=np.arange(m, dtype=np.int64), image_camera=ids,
                    qvec=np.stack([c["q"] for c in cs]), tvec=np.stack([c["t"] for c in cs]),
                    cam_model=np.array([c["model"] for c in cs], np.int32), cam_params=cam_params,
                    xyz=np.stack([c["X"] for c in cs]), refs=np.stack([c["ref"] for c in cs]),
                    patches=np.stack([c["d"] for c in cs]).astype(storage), corners=np.stack([c["c"] for c in cs]).astype(np.int32),
                    scales=np.stack([c["s"] for c in cs]))
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(ctx, arena, prob)
        rec, built = ba.eval_gram(interp_cfg(l2_normalize=l2), reset=True)
        rec = rec.download()
        assert built == m
        P = ba.projection_jacobian().download()
        for i, c in enumerate(cs):
            n, K = c["name"], len(c["params"])
            r_ref, J_ref = gold[n + "_r"], gold[n + "_J"]
            Pi = P[i][:, :10 + K]
            M = np.array([[rec[i, 1], rec[i, 2]], [rec[i, 2], rec[i, 3]]])
            H, g, s = Pi.T @ M @ Pi, Pi.T @ rec[i, 4:6], rec[i, 0]
            H_ref, g_ref, s_ref = J_ref.T @ J_ref, J_ref.T @ r_ref, float(r_ref @ r_ref)
            # scales: |r|^2 against itself, J^t J against its largest entry, J^t r against |J| |r| (it vanishes at an optimum)
            errs = (abs(s - s_ref) / s_ref, np.abs(H - H_ref).max() / np.abs(H_ref).max(),
                    np.abs(g - g_ref).max() / (np.sqrt(np.abs(H_ref).max()) * np.sqrt(s_ref)))
            worst = max(worst, *errs)
            assert max(errs) < GRAM_VS_REFERENCE_RTOL, (n, errs)
            models.add(c["model"])
            n_checked += 1
        arena.close()
    assert n_checked == len(gen.ba_cases()) and models == {0, 1, 2, 3, 4}
    assert worst < GRAM_VS_REFERENCE_SEEN, worst


def test_solve_survives_a_failed_cache_allocation(monkeypatch):
    """the cache is on by default and costs 1.4 KB per observation; when its allocation fails the solve must fall back to
    the texel evaluation (what PXR_GRAM_CACHE=0 runs) instead of returning PXR_ENOMEM -- same bits as a context with the cache
    switched off, and no stale error message."""
    from pixsfm_amd import _lib, synthetic
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    from test_ba_solve_gpu import _gauge
    prob = synthetic.make_ba_problem(n_cams=6, n_points=80, obs_per_point=4, seed=42)
    out = []
    for fail in (False, True):
        c = Context(0)
        if fail:
            monkeypatch.setenv("PXR_GRAM_FAIL_ALLOC", "1")
            assert c.gram_cache
        else:
            c.gram_cache = False
        arena = PatchArena.from_numpy(c, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(c, arena, prob)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob), options=lm_options(max_iterations=6, use_inner_iterations=True))
        out.append((s, ba.params()))
        if fail:
            assert _lib.load().pxr_last_error() in (b"", None)
        arena.close(); c.close()
    monkeypatch.delenv("PXR_GRAM_FAIL_ALLOC")
    (s0, p0), (s1, p1) = out
    assert s1["final_cost"] == s0["final_cost"] and s1["iterations"] == s0["iterations"]
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)
