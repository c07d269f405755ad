"""Randomised sweep of the reference extraction (pxr_ba_compute_references: descriptors at the projections, robust-mean IRLS,
closest observation) against the oracle -- random track lengths 1..14 (register and L2 paths), losses, iteration counts,
normalisation, storage types, channel counts, noise levels incl. identical descriptors (the early return).  NOT part of the
test suite.  python tests/fuzz/fuzz_refs_vs_oracle.py [n_trials] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("pixel-perfect-sfm_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import pxo
from pixsfm_amd import synthetic
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)
bad = 0
worst = 0.0
for trial in range(n_trials):
    opp = int(rng.integers(1, 15))
    n_cams = max(opp, int(rng.integers(3, 16)))
    dt = [np.float16, np.float32, np.float64][rng.integers(3)]
    ch = int(rng.choice([128, 64]))
    noise = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    prob = synthetic.make_ba_problem(n_cams=n_cams, n_points=int(rng.integers(5, 40)), obs_per_point=opp, seed=int(rng.integers(1 << 30)),
                                     noise=noise, dtype=dt, channels=ch, model=int(rng.integers(0, 5)))
    lname, la = [("cauchy", 0.25), ("huber", 0.5), ("trivial", 1.0), ("cauchy", 1.0)][rng.integers(4)]
    iters = int(rng.choice([1, 3, 20, 100]))
    l2 = bool(rng.integers(2))
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    chosen, mean = ba.compute_references(interp_cfg(l2_normalize=l2), make_loss(lname, [] if lname == "trivial" else [la]), iters=iters, keep_mean=True)
    refs = ba.d["refs"].download()
    cfg, ls = pxo.cfg(l2_normalize=l2), pxo.loss(lname, la)
    n_pts, C = len(prob["xyz"]), ch
    ok = True
    for p in range(n_pts):
        obs = np.nonzero(prob["obs_point"] == p)[0]
        descs = []
        for i in obs:
            img = prob["obs_image"][i]; cam = prob["image_camera"][img]
            pi = prob["obs_patch"][i]
            patch = pxo.make_patch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
            K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
            f, *_ = pxo.ba_residual(patch, cfg, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img], prob["xyz"][p],
                                    prob["cam_params"][cam][:K], None, jac=False)
            descs.append(f)
        descs = np.array(descs)
        if len(descs) == 2:
            # two observations: the midpoint is a fixed point of the IRLS by symmetry and, for weights 1 / rho that grow
            # as the distance shrinks, an UNSTABLE one -- rounding decides which observation the mean collapses onto
            # (in the reference as well); nothing to compare
            continue
        idx, ref, m = pxo.compute_reference(descs, ls, iters, l2)
        e = max(np.abs(mean[p] - m).max(), np.abs(refs[p] - ref).max())
        worst = max(worst, e)
        if chosen[p] != obs[idx]:
            # a tie between two observations at rounding level is not a bug: accept when their distances to the mean agree to 1e-12
            d = ((descs - m) ** 2).sum(1)
            j = int(np.nonzero(obs == chosen[p])[0][0]) if chosen[p] in obs else -1
            if j < 0 or abs(d[j] - d[idx]) > 1e-9 * max(1.0, d[idx]):
                ok = False
                print("   point %d: chosen %d vs %d, distances %s, mean err %.2e" % (p, chosen[p], obs[idx], d, np.abs(mean[p] - m).max()))
        elif e > 1e-9:
            ok = False
            print("   point %d: same choice, err %.2e (mean err %.2e)" % (p, e, np.abs(mean[p] - m).max()))
    if not ok:
        bad += 1
        print("trial %d: opp %d %s C%d noise %.2f loss %s iters %d l2 %d: mismatch (worst %.2e)" % (trial, opp, np.dtype(dt).name, ch, noise, lname, iters, l2, worst))
    arena.close()
print("trials %d  mismatches %d  worst abs error %.2e" % (n_trials, bad, worst))
