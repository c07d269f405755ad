"""Probe (GPU): records through the Gram-matrix cache vs the exact-order kernel, and LM solves with / without the cache."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pixel-perfect-sfm_amd"))
from pixsfm_amd import synthetic
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss

ctx = Context(0)
for dtype, C, l2 in [(np.float16, 128, True), (np.float32, 128, True), (np.float16, 64, True), (np.float16, 128, False)]:
    prob = synthetic.make_ba_problem(n_cams=12, n_points=900, obs_per_point=5, seed=21, rot_deg=0.3, pt_sigma=0.02, channels=C, dtype=dtype, noise=0.01)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cfg = interp_cfg(l2_normalize=l2)
    rec, *_ = ba.eval(cfg)
    exact = rec.download().copy()
    rec, built = ba.eval_gram(cfg)
    gram = rec.download().copy()
    d = np.abs(gram - exact)
    print(dtype.__name__, C, l2, "built", built, "of", ba.n_obs)
    for k, name in enumerate(["s", "gxx", "gxy", "gyy", "bx", "by", "x", "y"]):
        print("   %-4s max abs %.3e  max rel-to-max %.3e   typical %.3e" % (name, d[:, k].max(), d[:, k].max() / np.abs(exact[:, k]).max(), np.abs(exact[:, k]).mean()))
    print("   sum s: exact %.12e gram %.12e rel %.2e" % (exact[:, 0].sum(), gram[:, 0].sum(), abs(exact[:, 0].sum() - gram[:, 0].sum()) / exact[:, 0].sum()))
    # move the points: some cells change
    xyz = prob["xyz"] + np.random.default_rng(1).normal(0, 0.004, prob["xyz"].shape)
    ba.d["xyz"].upload(xyz)
    rec, built = ba.eval_gram(cfg, reset=False)
    g2 = rec.download().copy()
    rec, *_ = ba.eval(cfg)
    e2 = rec.download().copy()
    print("   after moving the points: rebuilt", built, " max abs diff s %.3e" % np.abs(g2[:, 0] - e2[:, 0]).max())
    rec, built = ba.eval_gram(cfg, reset=False)
    print("   again: rebuilt", built, "bitwise same", np.array_equal(rec.download(), g2))
    arena.close()

def gauge(prob):
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return pose_const, tmask, np.full(n_cam, 0b0110, np.uint16), np.zeros(n_pt, np.uint8)

for inner in (False, True):
    prob = synthetic.make_ba_problem(n_cams=12, n_points=900, obs_per_point=5, seed=21, rot_deg=0.3, pt_sigma=0.02, noise=0.01)
    out = []
    for on in (False, True):
        ctx.gram_cache = on
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(ctx, arena, prob)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge(prob), options=lm_options(max_iterations=10, use_inner_iterations=inner))
        out.append((s, ba.params()))
        arena.close()
    (s0, p0), (s1, p1) = out
    print("inner", inner, "iters", s0["iterations"], s1["iterations"], "succ", s0["num_successful"], s1["num_successful"])
    print("   initial %.12e %.12e  final %.12e %.12e" % (s0["initial_cost"], s1["initial_cost"], s0["final_cost"], s1["final_cost"]))
    print("   param diffs", [float(np.abs(a - b).max()) for a, b in zip(p0, p1)])
ctx.gram_cache = False
