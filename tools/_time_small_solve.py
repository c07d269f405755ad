"""Wall time of pxr_ba_solve on localization-sized problems (one image, a few hundred constant points): set-up vs loop.
python tools/_time_small_solve.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pixel-perfect-sfm_amd'))
import numpy as np
from pixsfm_amd import synthetic
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss

ctx = Context(0)
for n_cams, n_pts, opp in ((1, 300, 1), (1, 1000, 1), (8, 2000, 3), (50, 20000, 4)):
    prob = synthetic.make_ba_problem(n_cams=n_cams, n_points=n_pts, obs_per_point=opp, seed=3)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    n_img = n_cams
    pose_const = np.zeros(n_img, np.uint8)
    if n_cams > 1:
        pose_const[0] = 1
    tmask, cmask = np.zeros(n_img, np.uint8), np.full(len(prob["cam_model"]), 0xFFF, np.uint16)
    ptc = np.ones(n_pts, np.uint8) if n_cams == 1 else np.zeros(n_pts, np.uint8)
    best = None
    for rep in range(4):
        for name in ("qvec", "tvec", "xyz"):
            ba.d[name].upload(prob[name])
        t0 = time.perf_counter()
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pose_const, tmask, cmask, ptc, options=lm_options(max_iterations=10))
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, s)
    wall, s = best
    print("%3d images %6d observations: wall %.2f ms = set-up %.2f + loop %.2f (%d iterations, %.3f ms each)" % (
        n_cams, len(prob["obs_image"]), wall, s.get("setup_ms", float("nan")), s["total_ms"], s["iterations"], s["total_ms"] / max(1, s["iterations"])))
