"""Seeded scenes for reference extraction: tracks of 1-12 observations, keypoints without a patch in the
view, a point with nothing visible, fp16 / fp32 / fp64 patches, 128 / 64 channels, 16 / 10 texels, several camera models, scaled
patches, Cauchy / Huber / trivial loss, 1-100 iterations, l2_normalize off, use_float_simd, closest_to_robust_mean off.
Inputs only (the scenes come from pixsfm_amd.synthetic.make_ba_problem)."""
import numpy as np

# (name, make_ba_problem arguments, extraction options, observations whose patch is withheld from the view)
SCENES = [
    ("default", dict(n_cams=10, n_points=30, obs_per_point=4, noise=0.3, seed=1), dict(), 0),
    ("single_obs", dict(n_cams=6, n_points=12, obs_per_point=1, noise=0.3, seed=2), dict(), 0),
    ("three_obs_iters1", dict(n_cams=6, n_points=20, obs_per_point=3, noise=0.4, seed=3), dict(iters=1), 0),
    ("long_tracks", dict(n_cams=14, n_points=16, obs_per_point=12, noise=0.5, seed=4), dict(iters=100), 0),
    ("nine_obs_missing", dict(n_cams=12, n_points=18, obs_per_point=9, noise=0.5, seed=5), dict(iters=20), 14),
    ("f32_64ch", dict(n_cams=8, n_points=20, obs_per_point=5, noise=0.3, seed=6, dtype=np.float32, channels=64), dict(), 0),
    ("f64_ps10", dict(n_cams=8, n_points=14, obs_per_point=5, noise=0.3, seed=7, dtype=np.float64, patch_size=10), dict(), 0),
    ("no_l2", dict(n_cams=8, n_points=20, obs_per_point=6, noise=0.4, seed=8), dict(l2_normalize=False, iters=30), 0),
    ("huber", dict(n_cams=8, n_points=20, obs_per_point=6, noise=0.4, seed=9), dict(loss=("huber", 0.5), iters=15), 0),
    ("trivial", dict(n_cams=8, n_points=20, obs_per_point=6, noise=0.4, seed=10), dict(loss=("trivial", 0.0), iters=5), 0),
    ("robust_mean", dict(n_cams=8, n_points=20, obs_per_point=7, noise=0.4, seed=11), dict(closest_to_robust_mean=False), 0),
    ("opencv_scaled", dict(n_cams=8, n_points=16, obs_per_point=5, noise=0.3, seed=12, model=4, scale=(0.5, 0.25)), dict(), 0),
    ("pinhole_float_simd", dict(n_cams=8, n_points=16, obs_per_point=5, noise=0.3, seed=13, model=1), dict(use_float_simd=True), 6),
    ("shared_camera_tight_loss", dict(n_cams=9, n_points=16, obs_per_point=8, noise=0.6, seed=14, shared_camera=True, model=3),
     dict(loss=("cauchy", 0.05), iters=50), 0),
]
DEFAULTS = dict(l2_normalize=True, use_float_simd=False, loss=("cauchy", 0.25), iters=10, closest_to_robust_mean=True)


def scene(name):
    """-> (problem dict, options, has_patch [n_obs] bool); the withheld patches leave >= 3 visible observations per track
    except for one point that loses all of them (a track of exactly two visible observations sits on an unstable fixed
    point of the IRLS)."""
    from pixsfm_amd import synthetic
    _, kw, opts, n_missing = next(s for s in SCENES if s[0] == name)
    prob = synthetic.make_ba_problem(**kw)
    has = np.ones(len(prob["obs_image"]), bool)
    if n_missing:
        rng = np.random.default_rng(1000 + kw["seed"])
        per = kw["obs_per_point"]
        has[0:per] = False                                   # point 0: nothing visible -> no reference
        pts = rng.choice(np.arange(1, kw["n_points"]), n_missing // 2, replace=False)
        for p in pts:
            has[p * per + rng.choice(per, 2, replace=False)] = False
    return prob, {**DEFAULTS, **opts}, has
