"""Randomised CPU sweep: the ORACLE's reference extraction against the REFERENCE's own (ReferenceExtractor::RunSubset +
RobustMeanIRLS compiled in place, oracle/_ref/libpxo_ref_refs.so) on random scenes -- track lengths, storage types, channel
counts, patch sizes, camera models, losses, iteration counts, normalisation, missing patches.  Build container only (needs
the library built from /root/reference); not part of the test suite.
python tools/fuzz_refs_vs_reference.py [n_scenes] [seed]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_golden_refs.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
import pxo                      # noqa: E402
from pixsfm_amd import synthetic  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad, n_points = 0.0, 0, 0
for trial in range(n_scenes):
    opp = int(rng.choice([1, 3, 4, 5, 7, 9, 13]))
    kw = dict(n_cams=int(rng.integers(max(opp, 3), 16)), n_points=int(rng.integers(4, 25)), obs_per_point=opp, noise=float(rng.uniform(0.05, 0.7)),
              seed=int(rng.integers(1 << 30)), dtype=[np.float16, np.float32, np.float64][rng.integers(3)], channels=int(rng.choice([128, 64])),
              patch_size=int(rng.choice([16, 10, 12])), model=int(rng.integers(0, 5)), scale=[(1.0, 1.0), (0.5, 0.25), (0.75, 0.75)][rng.integers(3)])
    prob = synthetic.make_ba_problem(**kw)
    loss = [("cauchy", 0.25), ("cauchy", 0.05), ("huber", 0.3), ("trivial", 0.0)][rng.integers(4)]
    opts = dict(G.DEFAULTS, l2_normalize=bool(rng.integers(2)), use_float_simd=bool(rng.integers(2)), loss=loss, iters=int(rng.choice([1, 2, 10, 37, 100])),
                closest_to_robust_mean=bool(rng.integers(4) > 0))
    has = np.ones(len(prob["obs_image"]), bool)
    if opp >= 5 and rng.integers(2):
        for p in rng.choice(kw["n_points"], max(1, kw["n_points"] // 3), replace=False):
            has[p * opp + rng.choice(opp, 2, replace=False)] = False          # >= 3 visible observations remain
    ref = G.run_reference(prob, opts, has)
    cfg = pxo.cfg(l2_normalize=opts["l2_normalize"], use_float_simd=opts["use_float_simd"])
    ls = pxo.loss(loss[0], loss[1]) if loss[0] != "trivial" else pxo.loss("trivial")
    for p in range(len(prob["xyz"])):
        obs = np.nonzero((prob["obs_point"] == p) & has)[0]
        if len(obs) == 2:
            continue                                                          # unstable fixed point of the IRLS, DESIGN.md 2
        descs = []
        for i in obs:
            img = prob["obs_image"][i]
            cam = prob["image_camera"][img]
            patch = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
            K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
            descs.append(pxo.ba_residual(patch, cfg, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img], prob["xyz"][p],
                                         prob["cam_params"][cam][:K], None, jac=False)[0])
        idx, r, mu = pxo.compute_reference(np.array(descs), ls, opts["iters"], opts["l2_normalize"])
        out = r if opts["closest_to_robust_mean"] else mu
        n_points += 1
        err = np.abs(out - ref["descriptor"][p]).max()
        worst = max(worst, err)
        if obs[idx] != ref["src_obs"][p] or err > 1e-11:
            bad += 1
            print("MISMATCH scene", trial, "point", p, kw, opts, "chosen", obs[idx], ref["src_obs"][p], "err", err)
print("scenes %d  points %d  mismatches %d  worst abs error %.2e" % (n_scenes, n_points, bad, worst))
