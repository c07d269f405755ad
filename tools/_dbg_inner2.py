import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pixel-perfect-sfm_amd"))
import numpy as np
from pixsfm_amd import synthetic
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
def gauge_of(prob):
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return [pose_const, tmask, np.full(n_cam, 0b0110, np.uint16), np.zeros(n_pt, np.uint8)]
prob = synthetic.make_ba_problem(n_cams=7, n_points=50, obs_per_point=3, seed=53, pt_sigma=0.03)
gauge = gauge_of(prob); gauge[3][::9] = 1
for det in (False, True):
    ctx = Context(0); ctx.deterministic = det
    res = {}
    for mode in ("multi", "one", "multi1", "multi2", "packed"):
        for k in ("PXR_INNER_GRAM1", "PXR_INNER_PACKED"): os.environ.pop(k, None)
        os.environ.pop("PXR_INNER_GRAM_PPW", None)
        if mode == "one": os.environ["PXR_INNER_GRAM1"] = "1"
        if mode == "multi1": os.environ["PXR_INNER_GRAM_PPW"] = "1"
        if mode == "multi2": os.environ["PXR_INNER_GRAM_PPW"] = "2"
        if mode == "packed": os.environ["PXR_INNER_PACKED"] = "1"
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(ctx, arena, prob)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=4, use_inner_iterations=True))
        X = ba.params()[3].copy(); arena.close()
        print("det", det, mode, "final %.15e" % s["final_cost"], "X17", X[17])
    ctx.close()
