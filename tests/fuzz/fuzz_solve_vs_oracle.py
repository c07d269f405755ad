"""Randomised parity sweep of the SOLVERS, HIP against the oracle (NOT part of the test suite; a bug hunt): pxr_ba_solve and
pxr_ka_solve on small random problems -- camera models, shared cameras, random constant poses / translation subsets / camera
parameter subsets / constant points, losses, interpolation switches, inner iterations, direct and iterative Schur solver,
tolerances; KA with random bounds, constant nodes, weights.  A short horizon (few iterations) so that the accept / reject
decisions of the two LM loops must coincide.

By default the sweep runs the REFERENCE's arithmetic -- the exact-order evaluation (PXR_GRAM_CACHE=0) and the packed
inner-iteration kernel (PXR_INNER_PACKED=1): the fp32 horizontal pass everywhere, what the oracle restates -- and holds the
solvers to the tolerances of rounds 1-3 (final cost / parameters 1e-7, initial cost 1e-9).  `--default-arithmetic` sweeps the
shipped defaults instead (Gram-matrix algebra, exact fp64), at the one looser tolerance that difference is priced at (5e-6, the
value of tests/conftest.py FP32_PASS_PARAM_RTOL rounded down).
python tests/fuzz/fuzz_solve_vs_oracle.py [n_trials] [seed] [--default-arithmetic]"""
import os
import sys
DEFAULT_ARITHMETIC = "--default-arithmetic" in sys.argv
sys.argv = [a for a in sys.argv if a != "--default-arithmetic"]
if not DEFAULT_ARITHMETIC:
    os.environ["PXR_GRAM_CACHE"] = "0"
    os.environ["PXR_INNER_PACKED"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pixel-perfect-sfm_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import pxo
import pxo_ka
from pixsfm_amd import synthetic, synthetic_ka
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
from pixsfm_amd.ka_engine import KAProblem

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = Context(0)
GROUPS = {0: ([0], [1, 2], []), 1: ([0, 1], [2, 3], []), 2: ([0], [1, 2], [3]), 3: ([0], [1, 2], [3, 4]), 4: ([0, 1], [2, 3], [4, 5, 6, 7])}
bad = 0
for trial in range(n_trials):
    model = int(rng.integers(0, 5))
    n_cams, n_pts, opp = int(rng.integers(3, 9)), int(rng.integers(20, 90)), int(rng.integers(2, 5))
    opp = min(opp, n_cams)
    shared = bool(rng.integers(2))
    dt = [np.float16, np.float32, np.float64][rng.integers(3)]
    ch = int(rng.choice([128, 64]))
    prob = synthetic.make_ba_problem(n_cams=n_cams, n_points=n_pts, obs_per_point=opp, seed=int(rng.integers(1 << 30)), model=model,
                                     shared_camera=shared, dtype=dt, channels=ch, rot_deg=float(rng.uniform(0.05, 0.4)))
    n_img, n_cam = n_cams, len(prob["cam_model"])
    pose_const = (rng.random(n_img) < 0.25).astype(np.uint8); pose_const[0] = 1
    tmask = np.where((pose_const == 0) & (rng.random(n_img) < 0.3), rng.integers(1, 8, n_img), 0).astype(np.uint8)
    f, pp, ex = GROUPS[model]
    cmask = np.zeros(n_cam, np.uint16)
    for c in range(n_cam):
        const = ([] if rng.integers(2) else f) + ([] if rng.integers(4) == 0 else pp) + ([] if rng.integers(2) else ex)
        cmask[c] = sum(1 << a for a in const)
        if rng.integers(5) == 0:
            cmask[c] = (1 << len(f + pp + ex)) - 1
    ptc = (rng.random(n_pts) < 0.15).astype(np.uint8)
    lname, lpar = [("cauchy", [0.25]), ("trivial", []), ("huber", [0.3]), ("soft_l1", [0.5])][rng.integers(4)]
    l2, fs = bool(rng.integers(2)), bool(rng.integers(4) == 0)
    kw = dict(max_iterations=int(rng.integers(2, 7)), use_inner_iterations=bool(rng.integers(2)))
    if rng.integers(3) == 0:
        kw["function_tolerance"] = 1e-3
    solver = ["direct", "iterative"][rng.integers(2)]
    try:
        loss_g, loss_o = make_loss(lname, lpar), pxo.loss(lname, lpar[0] if lpar else 1.0)
    except Exception:       # a loss one side does not know
        lname, lpar = "cauchy", [0.25]
        loss_g, loss_o = make_loss(lname, lpar), pxo.loss(lname, 0.25)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    gauge = (pose_const, tmask, cmask, ptc)
    sg = ba.solve(interp_cfg(l2_normalize=l2, use_float_simd=fs), loss_g, *gauge,
                  options=lm_options(linear_solver=solver, linear_r_tolerance=1e-13 if solver == "iterative" else -1.0, eta=1e-30 if solver == "iterative" else 0.1,
                                     max_linear_solver_iterations=2000, **kw))
    q, t, k, X = ba.params()
    so, qo, to, ko, Xo = pxo.ba_solve(prob, pxo.cfg(l2, fs, False), loss_o, *gauge, pxo.lm_options(**kw))
    tol = 1e-5 if (fs or solver == "iterative") else 1e-7
    # the shipped defaults work on Gram matrices (exact in fp64 where the oracle restates the reference's fp32 horizontal pass):
    # parameters agree to ~1e-6 there, not 1e-7; the initial cost is the exact-order kernel's in every mode
    if DEFAULT_ARITHMETIC and dt != np.float64:
        tol = max(tol, 5e-6)
    ok = sg["iterations"] == so["iterations"] and sg["num_successful"] == so["num_successful"] and sg["termination"] == so["termination"]
    ok = ok and abs(sg["initial_cost"] - so["initial_cost"]) <= 1e-9 * abs(so["initial_cost"])
    ok = ok and abs(sg["final_cost"] - so["final_cost"]) <= tol * max(abs(so["final_cost"]), 1e-12)
    err = 0.0
    for a, b in zip((q, t, k, X), (qo, to, ko, Xo)):
        b = np.asarray(b); a = a[:, :b.shape[1]] if a.ndim == 2 else a
        err = max(err, float(np.abs(a - b).max() / max(1.0, np.abs(b).max())))
    ok = ok and err < tol
    if not ok:
        bad += 1
        print("BA trial %d: model %d shared %d %s C%d loss %s l2 %d fs %d %s %s | gpu it %d ok %d term %d cost %.6e -> %.6e | cpu it %d ok %d term %d cost %.6e -> %.6e | param err %.2e" % (
            trial, model, shared, np.dtype(dt).name, ch, lname, l2, fs, solver, kw, sg["iterations"], sg["num_successful"], sg["termination"],
            sg["initial_cost"], sg["final_cost"], so["iterations"], so["num_successful"], so["termination"], so["initial_cost"], so["final_cost"], err))
    arena.close()
    # ---- KA ----
    big = rng.integers(4) == 0          # one large sub-problem: beyond the solve kernel's LDS metadata caches (96 nodes / 512 blocks)
    kprob = synthetic_ka.make_ka_problem(n_tracks=int(rng.integers(30, 60)) if big else int(rng.integers(3, 12)), track_len=int(rng.integers(2, 7)), seed=int(rng.integers(1 << 30)),
                                         dtype=dt, channels=ch, sigma=float(rng.uniform(0.3, 1.5)), max_kps_per_problem=100000 if big else int(rng.integers(6, 40)),
                                         scale=(1.0, 1.0) if rng.integers(2) else (0.5, 0.25))
    kprob["node_const"] = np.where(rng.random(len(kprob["kp"])) < 0.2, 1, kprob["node_const"]).astype(np.uint8)
    kprob["edge_w"] = rng.uniform(0.2, 1.0, len(kprob["edge_w"]))
    bound = float(rng.choice([4.0, 1.0, 0.3]))
    arena = PatchArena.from_numpy(ctx, kprob["patches"], kprob["corners"], kprob["scales"])
    ka = KAProblem(ctx, arena, kprob)
    kopt = dict(max_iterations=int(rng.integers(2, 8)), parameter_tolerance=1e-5)
    total, per = ka.solve(interp_cfg(l2_normalize=l2, use_float_simd=fs), loss_g, bound=bound, per_problem=True, options=lm_options(**kopt))
    out = ka.keypoints()
    kpo, sums = pxo_ka.ka_solve(kprob, pxo.cfg(l2, fs, False), loss_o, bound, pxo.lm_options(**kopt))
    ok = all(g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"] and g["termination"] == o["termination"] for g, o in zip(per, sums))
    kerr = float(np.abs(out - kpo).max())
    ok = ok and kerr < (1e-4 if fs else 1e-6)
    if not ok:
        bad += 1
        print("KA trial %d: %s C%d loss %s l2 %d fs %d bound %.1f %s | kp err %.2e | gpu %s | cpu %s" % (
            trial, np.dtype(dt).name, ch, lname, l2, fs, bound, kopt, kerr, [(g["iterations"], g["num_successful"], g["termination"]) for g in per],
            [(o["iterations"], o["num_successful"], o["termination"]) for o in sums]))
    arena.close()
print("trials %d  mismatches %d" % (n_trials, bad))
