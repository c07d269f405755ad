import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
from pixsfm_amd import synthetic
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
ctx = Context(0)
def gauge_of(prob):
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return [pose_const, tmask, np.full(n_cam, 0b0110, np.uint16), np.zeros(n_pt, np.uint8)]
for opp, sig in ((3, 0.03), (6, 0.01), (5, 0.02)):
    prob = synthetic.make_ba_problem(n_cams=7, n_points=50, obs_per_point=opp, seed=50 + opp, pt_sigma=sig)
    gauge = gauge_of(prob); gauge[3][::9] = 1
    for it in (1, 4):
        out = {}
        for mode in ("multi", "one", "packed"):
            for k in ("PXR_INNER_GRAM1", "PXR_INNER_PACKED"): os.environ.pop(k, None)
            if mode == "one": os.environ["PXR_INNER_GRAM1"] = "1"
            if mode == "packed": os.environ["PXR_INNER_PACKED"] = "1"
            arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
            ba = BAProblem(ctx, arena, prob)
            s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=it, use_inner_iterations=True))
            out[mode] = (s, ba.params()[3].copy()); arena.close()
        X0 = out["packed"][1]
        for mode in ("multi", "one"):
            d = np.abs(out[mode][1] - X0).max(axis=1)
            bad = np.nonzero(d > 1e-5)[0]
            print(opp, it, mode, "final", out[mode][0]["final_cost"], "packed", out["packed"][0]["final_cost"], "succ", out[mode][0]["num_successful"], out["packed"][0]["num_successful"], "max dX %.2e" % d.max(), "bad points", bad[:10], d[bad[:10]])
