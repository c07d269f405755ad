"""The oracle's Schur-complement LM iteration (oracle/pxo_lm_bench.c: what bench.py times as `cpu_baseline_lm_projected`)
against the oracle's dense LM (oracle/pxo_solve.c, all unknowns in one normal matrix): the same first step."""
import numpy as np


def test_schur_iteration_equals_the_dense_first_step():
    import pxo
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=6, n_points=40, obs_per_point=3, seed=13, channels=32)
    n_img, n_pts = 6, 40
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    cmask = np.full(n_img, 0b0110, np.uint16)
    ptc = np.zeros(n_pts, np.uint8); ptc[5] = 1
    gauge = (pose_const, tmask, cmask, ptc)
    cfg, ls = pxo.cfg(), pxo.loss("cauchy", 0.25)
    out = pxo.ba_lm_iteration_schur(prob, cfg, ls, *gauge, radius=1e4, n_threads=4)
    assert out["rc"] == 0 and out["n_c"] == 6 * 5 - 1 + 2 * 6
    assert abs(out["cost"] - out["cost_check"]) < 1e-12 * out["cost"]
    s, q, t, k, X = pxo.ba_solve(prob, cfg, ls, *gauge, pxo.lm_options(max_iterations=1, jacobi_scaling=0))
    assert s["num_successful"] == 1 and abs(s["initial_cost"] - out["cost"]) < 1e-12 * out["cost"]
    # points: x1 = x0 + delta_p; free tvec components and intrinsics likewise (Euclidean blocks)
    assert np.abs((X - prob["xyz"]) - out["delta_p"]).max() < 1e-9 * np.abs(out["delta_p"]).max()
    assert np.all(out["delta_p"][5] == 0)
    dc = out["delta_c"]
    col = 0
    for i in range(n_img):
        if pose_const[i]:
            continue
        col += 3                                                     # rotation tangent
        for a in range(3):
            if (tmask[i] >> a) & 1:
                continue
            assert abs((t[i, a] - prob["tvec"][i, a]) - dc[col]) < 1e-9 * max(1e-3, abs(dc[col]))
            col += 1
    for c in range(n_img):                                            # SIMPLE_RADIAL: f and k free
        for a in (0, 3):
            assert abs((k[c, a] - prob["cam_params"][c, a]) - dc[col]) < 1e-8 * max(1e-6, abs(dc[col]))
            col += 1
    assert col == out["n_c"]
    assert all(out[k_] >= 0 for k_ in ("jacobian_eval_ms", "schur_ms", "cholesky_ms", "backsub_ms", "cost_eval_ms"))
