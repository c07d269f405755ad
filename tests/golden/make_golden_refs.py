"""Golden vectors for reference extraction from the REFERENCE's own code: ReferenceExtractor::RunSubset /
GetVisibleObservations / ComputeReference / FillDescriptorTrack (bundle_adjustment/src/reference_extractor.h:172-300) with
RobustMeanIRLS (base/src/irls_optim.h:24-71), the patch interpolators and WorldToPixel under them, compiled in place
(oracle/ref_refs_shim.cc -> oracle/_ref/libpxo_ref_refs.so).  Run in the build container:

    python tests/golden/make_golden_refs.py        # writes tests/golden/refs_ref.npz

The scenes are regenerated from their seeds by the tests (pixsfm_amd.synthetic.make_ba_problem); the file holds the
reference's outputs and a checksum of each scene's inputs."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_refs.so")
NPARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8}
DT = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
LOSS = {"trivial": 0, "cauchy": 1, "huber": 2}

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


def checksum(prob, has):
    h = hashlib.sha256()
    for k in ("cam_model", "cam_params", "image_camera", "qvec", "tvec", "xyz", "obs_image", "obs_point", "patches", "corners", "scales"):
        h.update(np.ascontiguousarray(prob[k]).tobytes())
    h.update(has.tobytes())
    return h.hexdigest()


def run_reference(prob, opts, has, keep_observations=True):
    """the reference's extraction on a scene; the point2D index of observation i is i"""
    lib = C.CDLL(LIB)
    n_obs, n_pts = len(prob["obs_image"]), len(prob["xyz"])
    order = np.argsort(prob["obs_point"], kind="stable")
    ptr = np.zeros(n_pts + 1, np.int64)
    np.cumsum(np.bincount(prob["obs_point"], minlength=n_pts), out=ptr[1:])
    t_img = np.ascontiguousarray(prob["obs_image"][order], np.int32)
    t_p2d = np.ascontiguousarray(order, np.int32)
    t_patch = np.ascontiguousarray(np.where(has[order], prob["obs_patch"][order], -1), np.int64)
    patches = np.ascontiguousarray(prob["patches"])
    _, H, W, ch = patches.shape
    cam_model = np.ascontiguousarray(prob["cam_model"], np.int32)
    nparams = np.array([NPARAMS[int(m)] for m in cam_model], np.int32)
    cam_params = np.zeros((len(cam_model), 8))
    cam_params[:, :prob["cam_params"].shape[1]] = prob["cam_params"][:, :8]
    arrs = dict(image_camera=np.ascontiguousarray(prob["image_camera"], np.int32), qvec=np.ascontiguousarray(prob["qvec"], np.float64),
                tvec=np.ascontiguousarray(prob["tvec"], np.float64), xyz=np.ascontiguousarray(prob["xyz"], np.float64),
                corners=np.ascontiguousarray(prob["corners"], np.int32), scales=np.ascontiguousarray(prob["scales"], np.float64))
    has_ref, src_img, src_p2d = np.zeros(n_pts, np.uint8), np.full(n_pts, -1, np.int32), np.full(n_pts, -1, np.int32)
    desc, n_kept = np.zeros((n_pts, ch)), np.zeros(n_pts, np.int32)
    obs_desc, obs_cost = np.zeros((n_obs, ch)), np.zeros(n_obs)
    lp = np.array([opts["loss"][1]], np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.pxo_ref_extract_references(
        len(cam_model), p(cam_model), p(nparams), p(cam_params), len(arrs["image_camera"]), p(arrs["image_camera"]), p(arrs["qvec"]),
        p(arrs["tvec"]), C.c_int64(n_pts), p(arrs["xyz"]), p(ptr), p(t_img), p(t_p2d), p(t_patch), DT[patches.dtype], p(patches), H, W, ch,
        p(arrs["corners"]), p(arrs["scales"]), int(opts["l2_normalize"]), int(opts["use_float_simd"]), LOSS[opts["loss"][0]], p(lp),
        int(opts["iters"]), int(opts["closest_to_robust_mean"]), int(keep_observations), p(has_ref), p(src_img), p(src_p2d), p(desc), p(n_kept),
        p(obs_desc), p(obs_cost))
    if rc != 0:
        raise RuntimeError("pxo_ref_extract_references failed: %d" % rc)
    # per-observation outputs back from track order (visible observations first) to observation order
    od, oc = np.full((n_obs, ch), np.nan), np.full(n_obs, np.nan)
    for q in range(n_pts):
        vis = [i for i in order[ptr[q]:ptr[q + 1]] if has[i]]
        assert n_kept[q] == (len(vis) if has_ref[q] else 0)
        for k, i in enumerate(vis):
            od[i], oc[i] = obs_desc[ptr[q] + k], obs_cost[ptr[q] + k]
    return dict(has_ref=has_ref.astype(bool), src_image=src_img, src_obs=src_p2d, descriptor=desc, obs_descriptor=od, obs_cost=oc)


def main():
    store = {}
    for name, *_ in SCENES:
        prob, opts, has = scene(name)
        out = run_reference(prob, opts, has)
        for k, v in out.items():
            if k != "obs_descriptor":       # the descriptors at the observations are pinned by residuals_ref.npz already; keep the file small
                store[name + "|" + k] = v
        store[name + "|checksum"] = np.array(checksum(prob, has))
        print(name, "points with a reference:", int(out["has_ref"].sum()), "/", len(out["has_ref"]))
    np.savez_compressed(os.path.join(HERE, "refs_ref.npz"), **store)


if __name__ == "__main__":
    main()
