"""GPU, BASELINE.json full sizes (configs[2]: 200 cams / 200k points / 1M observations, 65.5 GB of
fp16 patches): the oracle cannot evaluate 1M observations in seconds, so the fused kernel is
checked through size-independent properties, plus the oracle on a random sample of the SAME
problem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(ctx):
    import torch
    from pixsfm_amd import synthetic_gpu
    from pixsfm_amd.engine import BAProblem, PatchArena
    prob, patches = synthetic_gpu.make_ba_problem_gpu("cuda:0", n_cams=200, n_points=200_000, obs_per_point=5, seed=2)
    arena = PatchArena(ctx, len(prob["obs_image"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    yield prob, patches, ba
    del ba, arena, patches
    torch.cuda.empty_cache()


def test_properties_at_one_million_observations(ctx, big):
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, patches, ba = big
    assert ba.n_obs == 1_000_000
    rec, _, _, _ = ba.eval(interp_cfg(), with_jacobian=True)
    rec = rec.download()
    s = rec[:, 0]
    # unit-norm descriptors on both sides: 0 <= |f - ref|^2 <= 4
    assert s.min() >= 0.0 and s.max() <= 4.0 + 1e-12
    # Cauchy-Schwarz on the reduced Jacobian blocks: (gx.gy)^2 <= (gx.gx)(gy.gy), |g.r|^2 <= (g.g) s
    assert (rec[:, 2] ** 2 <= rec[:, 1] * rec[:, 3] * (1 + 1e-12) + 1e-300).all()
    assert (rec[:, 4] ** 2 <= rec[:, 1] * s * (1 + 1e-9) + 1e-300).all()
    assert (rec[:, 5] ** 2 <= rec[:, 3] * s * (1 + 1e-9) + 1e-300).all()
    # cost == sum of 0.5 rho(s) recomputed on the host from the records (checksum of checksums)
    cost = ba.cost(make_loss("cauchy", [0.25]))
    want = 0.5 * 0.0625 * np.log1p(s / 0.0625).sum()
    assert abs(cost - want) < 1e-10 * want
    # value-only pass agrees with the Jacobian pass
    rec0 = ba.eval(interp_cfg(), with_jacobian=False)[0].download()
    assert np.array_equal(rec0[:, 0], s) and np.array_equal(rec0[:, 6:8], rec[:, 6:8])
    # at the ground truth every observation sits on the rendered optimum
    ba.d["qvec"].upload(prob["gt_qvec"]); ba.d["tvec"].upload(prob["gt_tvec"]); ba.d["xyz"].upload(prob["gt_xyz"])
    s_gt = ba.eval(interp_cfg(), with_jacobian=True)[0].download()
    assert s_gt[:, 0].max() < 1e-3 and np.abs(s_gt[:, 6:8] - prob["centers"]).max() < 1e-9
    ba.d["qvec"].upload(prob["qvec"]); ba.d["tvec"].upload(prob["tvec"]); ba.d["xyz"].upload(prob["xyz"])


def test_oracle_on_a_random_sample_of_the_full_problem(ctx, big):
    import pxo
    from pixsfm_amd.engine import interp_cfg
    prob, patches, ba = big
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(ba.n_obs, 512, replace=False))
    _, r, gx, gy = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    P = ba.projection_jacobian().download()[idx]
    r, gx, gy = r.download()[idx], gx.download()[idx], gy.download()[idx]
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "corners", "scales"):
        sub[k] = prob[k][idx]
    sub["obs_patch"] = np.arange(len(idx), dtype=np.int64)
    import torch
    sub["patches"] = patches[torch.as_tensor(idx, device=patches.device)].cpu().numpy()
    _, r_o, J_o = pxo.ba_eval_batch(sub, pxo.cfg(), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    assert np.abs(r - r_o).max() < 1e-10 * np.abs(r_o).max()
    J = gx[:, :, None] * P[:, None, 0, :] + gy[:, :, None] * P[:, None, 1, :]
    assert np.abs(J - J_o).max() < 1e-10 * np.abs(J_o).max()


def test_gram_matrix_records_against_the_oracle_on_a_sample_of_the_full_problem(ctx, big):
    """The LM loop's default evaluation (records from cached Gram matrices, csrc/pxr_ba_gram.hip) at configs[2]: |r|^2, J^t J and
    J^t r of 512 random blocks against the oracle's residuals and Jacobians (which restate the reference's fp32 horizontal
    pass, base/src/interpolation.h:177-218) at north_star's 1e-5 -- seen: the pass's own rounding, ~1e-7."""
    import pxo
    import torch
    from pixsfm_amd.engine import interp_cfg
    prob, patches, ba = big
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(ba.n_obs, 512, replace=False))
    rec, built = ba.eval_gram(interp_cfg(), reset=True)
    assert built == ba.n_obs
    rec = rec.download()[idx]
    P = ba.projection_jacobian().download()[idx][:, :, :14]              # SIMPLE_RADIAL: 10 + 4 columns
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "corners", "scales"):
        sub[k] = prob[k][idx]
    sub["obs_patch"] = np.arange(len(idx), dtype=np.int64)
    sub["patches"] = patches[torch.as_tensor(idx, device=patches.device)].cpu().numpy()
    _, r_o, J_o = pxo.ba_eval_batch(sub, pxo.cfg(), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    J_o = J_o[:, :, :14]
    M = np.stack([np.stack([rec[:, 1], rec[:, 2]], 1), np.stack([rec[:, 2], rec[:, 3]], 1)], 1)       # n x 2 x 2
    H = np.einsum("nai,nab,nbj->nij", P, M, P)
    g = np.einsum("nai,na->ni", P, rec[:, 4:6])
    H_o, g_o, s_o = np.einsum("nci,ncj->nij", J_o, J_o), np.einsum("nci,nc->ni", J_o, r_o), (r_o * r_o).sum(1)
    hmax = np.abs(H_o).reshape(len(idx), -1).max(1)
    e_s = np.abs(rec[:, 0] - s_o) / s_o
    e_H = np.abs(H - H_o).reshape(len(idx), -1).max(1) / hmax
    e_g = np.abs(g - g_o).max(1) / (np.sqrt(hmax) * np.sqrt(s_o))
    assert max(e_s.max(), e_H.max(), e_g.max()) < 1e-5, (e_s.max(), e_H.max(), e_g.max())
    assert max(e_s.max(), e_H.max(), e_g.max()) < 2e-6, (e_s.max(), e_H.max(), e_g.max())


def test_lm_descends_monotonically_at_full_size(ctx, big):
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, patches, ba = big
    n_img = 200
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pose_const, tmask, np.full(n_img, 0b0110, np.uint16),
                 np.zeros(200_000, np.uint8), options=lm_options(max_iterations=6))
    assert s["num_successful"] >= 5 and s["final_cost"] < 1e-4 * s["initial_cost"]
    q, t, k, X = ba.params()
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12
    assert np.array_equal(t[0], prob["tvec"][0]) and t[1][0] == prob["tvec"][1][0]
    # the summary's final cost is the cost of the parameters left on the device (the solver's from cached Gram matrices, this
    # one from the texels: they differ by the rounding of the reference's fp32 pass, conftest.FP32_PASS_COST_RTOL)
    from conftest import FP32_PASS_COST_RTOL
    ba.eval(interp_cfg(), with_jacobian=False)
    assert abs(ba.cost(make_loss("cauchy", [0.25])) - s["final_cost"]) < FP32_PASS_COST_RTOL * s["final_cost"]


def test_cost_maps_at_one_million_observations(ctx, big):
    """pxr_costmap_extract over the whole 65.5 GB arena (the persistent kernel: ~2000 patches per workgroup), then the
    cost-map BA on the 1M maps.  Properties: every cost >= 0 and (trivial loss) <= 2 for unit descriptors, the sum of
    the cost channel equals a torch reduction over the same arena, a random sample equals the oracle, and the
    cost-map LM descends."""
    import pxo_costmap
    import torch
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, patches, ba = big
    cm = ba.extract_costmaps(make_loss("trivial", []), dtype=np.float32)
    ctx.sync()
    rng = np.random.default_rng(11)
    idx = np.sort(rng.choice(ba.n_obs, 300, replace=False))
    for i in idx[:3].tolist() + [0, ba.n_obs - 1]:                       # first / last / a few single maps, full compare
        got = cm.download(i, 1)[0][0]
        want = pxo_costmap.fill_point_costmap(patches[i].cpu().numpy(), prob["refs"][prob["obs_point"][i]], out_dtype=np.float32)
        assert np.abs(got - want).max() <= 2e-7 * max(1.0, np.abs(want).max())
    # checksum of the cost channel against an independent torch reduction, in chunks of 50k maps
    maps_t = torch.empty(0)
    refs_t = torch.from_numpy(prob["refs"]).to("cuda:0")
    pts_t = torch.from_numpy(prob["obs_point"].astype(np.int64)).to("cuda:0")
    total_want, total_got, worst = 0.0, 0.0, 0.0
    for lo in range(0, ba.n_obs, 50_000):
        hi = min(ba.n_obs, lo + 50_000)
        res = patches[lo:hi].double() - refs_t[pts_t[lo:hi]][:, None, None, :]
        want = 0.5 * (res * res).sum(-1)                                 # (n, 16, 16)
        got = torch.from_numpy(cm.download(lo, hi - lo)[0][..., 0]).to("cuda:0").double()
        worst = max(worst, float((got - want).abs().max()))
        total_want += float(want.sum()); total_got += float(got.sum())
        assert float(got.min()) >= 0.0 and float(got.max()) <= 2.0 + 1e-6
    assert worst < 5e-7 and abs(total_got - total_want) < 1e-7 * total_want
    del maps_t, refs_t, pts_t
    # cost-map BA at full size
    cba = ba.costmap_problem(cm)
    n_img = 200
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    host = {k: ba.d[k].download() for k in ("qvec", "tvec", "cam_params", "xyz")}
    s = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), pose_const, tmask, np.full(n_img, 0b0110, np.uint16),
                  np.zeros(len(prob["xyz"]), np.uint8), options=lm_options(max_iterations=4, use_inner_iterations=True))
    # (the preceding LM test already refined these parameters: the maps' cost floor is close, it must still go down)
    assert s["num_successful"] >= 1 and s["final_cost"] < s["initial_cost"] and np.isfinite(s["final_cost"])
    for k, v in host.items():                                            # leave the shared parameters as they were
        ba.d[k].upload(v)
    cm.close()


def test_ka_config0_scale_matches_oracle(ctx):
    """BASELINE configs[0] scale (sacre_coeur: ~8 000 observations in ~1 900 tracks, demo.ipynb:271-274): every
    sub-problem of the KA solve against the oracle LM -- same iteration counts, termination and keypoints."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    rng = np.random.default_rng(0)
    parts, off = [], 0
    # tracks of 3 .. 6 nodes (mean ~4.2) packed into <= 50-keypoint sub-problems like keypoint_adjustment/main.py:13-57
    prob = None
    for tl, nt in ((3, 500), (4, 700), (5, 500), (6, 200)):
        p = synthetic_ka.make_ka_problem(n_tracks=nt, track_len=tl, seed=100 + tl, max_kps_per_problem=50)
        parts.append(p)
    n_nodes = sum(len(p["kp"]) for p in parts)
    assert 7500 <= n_nodes <= 8500
    cat = {}
    node_off = prob_off = 0
    for p in parts:
        q = dict(p)
        q["node_patch"] = p["node_patch"] + node_off
        q["edge_src"] = p["edge_src"] + node_off
        q["edge_dst"] = p["edge_dst"] + node_off
        q["node_problem"] = p["node_problem"] + prob_off
        for k in ("kp", "node_patch", "node_const", "node_problem", "edge_src", "edge_dst", "edge_w", "patches", "corners", "scales"):
            cat.setdefault(k, []).append(q[k])
        node_off += len(p["kp"]); prob_off += p["n_problems"]
    cat = {k: np.concatenate(v) for k, v in cat.items()}
    arena = PatchArena.from_numpy(ctx, cat["patches"], cat["corners"], cat["scales"])
    ka = KAProblem(ctx, arena, cat)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    kp = ka.keypoints()
    kpo, sums = pxo_ka.ka_solve(cat, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0)
    assert len(per) == len(sums) == prob_off
    same_traj = sum(g["iterations"] == o["iterations"] and g["termination"] == o["termination"] for g, o in zip(per, sums))
    assert same_traj >= 0.995 * len(per)            # a borderline tolerance decision may differ in a handful of problems
    assert np.abs(kp - kpo).max() < 1e-4            # north_star bar for refined parameters
    assert np.percentile(np.abs(kp - kpo).max(axis=1), 99) < 1e-7
    assert abs(total["final_cost"] - sum(o["final_cost"] for o in sums)) < 1e-6 * total["initial_cost"]


def test_ba_courtyard_shape_matches_oracle(ctx):
    """BASELINE configs[3] shape (38 images) at a size the oracle's dense LM finishes in seconds: 38 cameras,
    500 points, 3 000 observations, reduced camera system of ~300 unknowns (several Cholesky panels)."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=38, n_points=500, obs_per_point=6, seed=38)
    n_img = 38
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    gauge = (pose_const, tmask, np.full(n_img, 0b0110, np.uint16), np.zeros(500, np.uint8))
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=4))
    q, t, k, X = ba.params()
    so, qo, to, ko, Xo = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), *gauge, pxo.lm_options(max_iterations=4))
    assert s["num_camera_unknowns"] > 250 and s["num_point_unknowns"] == 1500
    assert s["iterations"] == so["iterations"] and s["num_successful"] == so["num_successful"]
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-7 * so["initial_cost"]
    assert np.abs(q - qo).max() < 1e-7 and np.abs(t - to).max() < 1e-7 and np.abs(X - Xo).max() < 1e-6
    assert np.abs(k[:, :4] - ko[:, :4]).max() < 1e-5 * 1200


def test_many_cameras_column_tiled_schur_and_long_cholesky(ctx):
    """More than 2047 camera unknowns: the LDS-privatised Schur contraction runs over several column tiles and the
    Cholesky over ~40 panels.  No oracle at this size (its dense LM is cubic in ALL unknowns); the independent
    global-atomics contraction (PXR_SCHUR_GLOBAL_ATOMICS=1) must give the same trajectory, and the solve must
    reduce the cost towards the noise floor."""
    import os
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    n_img = 320
    prob = synthetic.make_ba_problem(n_cams=n_img, n_points=2500, obs_per_point=6, seed=77)
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    gauge = (pose_const, tmask, np.full(n_img, 0b0110, np.uint16), np.zeros(2500, np.uint8))
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    out = {}
    for tag, env in (("lds", None), ("global", "1")):
        if env:
            os.environ["PXR_SCHUR_GLOBAL_ATOMICS"] = env
        else:
            os.environ.pop("PXR_SCHUR_GLOBAL_ATOMICS", None)
        try:
            ba = BAProblem(ctx, arena, prob)
            s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=4))
            out[tag] = (s, ba.params())
        finally:
            os.environ.pop("PXR_SCHUR_GLOBAL_ATOMICS", None)
    s1, p1 = out["lds"]; s2, p2 = out["global"]
    assert s1["num_camera_unknowns"] > 2047
    assert s1["iterations"] == s2["iterations"] and s1["num_successful"] == s2["num_successful"]
    assert abs(s1["final_cost"] - s2["final_cost"]) < 1e-9 * s1["initial_cost"]
    for a_, b_ in zip(p1, p2):
        assert np.abs(a_ - b_).max() < 1e-8 * max(1.0, np.abs(b_).max())
    assert s1["final_cost"] < 0.01 * s1["initial_cost"]


def test_ka_full_size_oracle_on_a_sample_of_sub_problems(ctx):
    """BASELINE configs[1] at FULL size (10 000 tracks x 10 nodes = 100 000 keypoints, 450 000 edges, 2 000
    sub-problems, 6.5 GB of patches): the whole bounded LM on the GPU, then the oracle on a random sample of the SAME
    sub-problems (they are independent: a sub-problem's solve does not depend on the others) -- iteration counts, final
    costs and refined keypoints (1e-6 px; north_star 1e-4; vs the oracle, parity unpinned w.r.t. Ceres), plus the
    size-independent properties of the full solve."""
    import os
    import sys
    import pxo
    import pxo_ka
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob, patches = bench_ka.make_problem_gpu("cuda:0", 10000, 10)
    n = len(prob["kp"])
    arena = PatchArena(ctx, n, 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    assert ka.n_edges == 450_000 and ka.n_problems == 2000
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    kp = ka.keypoints()
    # properties of the whole solve: every sub-problem converged or stopped at a smaller cost, roots did not move,
    # every keypoint stayed inside its box
    assert all(p["final_cost"] <= p["initial_cost"] * (1 + 1e-12) for p in per)
    root = prob["node_const"].astype(bool)
    assert np.array_equal(kp[root], prob["kp"][root])
    assert np.abs(kp - prob["kp"]).max() <= 4.0 + 1e-9
    assert abs(total["final_cost"] - sum(p["final_cost"] for p in per)) < 1e-9 * total["initial_cost"]
    assert total["final_cost"] < 0.01 * total["initial_cost"]
    # the oracle on 12 of the 2000 sub-problems
    rng = np.random.default_rng(3)
    chosen = np.sort(rng.choice(ka.n_problems, 12, replace=False))
    node_sel = np.nonzero(np.isin(prob["node_problem"], chosen))[0]
    renum = np.full(ka.n_problems, -1); renum[chosen] = np.arange(len(chosen))
    new_node = np.full(n, -1); new_node[node_sel] = np.arange(len(node_sel))
    edge_sel = np.nonzero(new_node[prob["edge_src"]] >= 0)[0]
    sub = dict(kp=prob["kp"][node_sel], node_patch=np.arange(len(node_sel), dtype=np.int64), node_const=prob["node_const"][node_sel],
               node_problem=renum[prob["node_problem"][node_sel]].astype(np.int32),
               edge_src=new_node[prob["edge_src"][edge_sel]].astype(np.int32),
               edge_dst=new_node[prob["edge_dst"][edge_sel]].astype(np.int32), edge_w=prob["edge_w"][edge_sel],
               patches=patches[torch.as_tensor(node_sel, device="cuda:0")].cpu().numpy(),
               corners=prob["corners"][node_sel], scales=prob["scales"][node_sel])
    assert len(edge_sel) == sum(ka.problem_sizes[c] for c in chosen)
    kpo, sums = pxo_ka.ka_solve(sub, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    for c, o in zip(chosen, sums):
        g = per[c]
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"], (c, g, o)
        assert abs(g["initial_cost"] - o["initial_cost"]) < 1e-10 * o["initial_cost"]
        assert abs(g["final_cost"] - o["final_cost"]) < 1e-7 * o["initial_cost"]
    assert np.abs(kp[node_sel] - kpo).max() < 1e-6
    del ka, arena, patches
    torch.cuda.empty_cache()


def _quat_plus(q, d):
    """[upstream Ceres QuaternionManifold::Plus] q_delta = [cos |d|, sin |d| / |d| d], result = q_delta * q (w first)."""
    nd = np.linalg.norm(d, axis=1, keepdims=True)
    s = np.where(nd > 0, np.sin(nd) / np.where(nd > 0, nd, 1.0), 1.0)
    w0, v0 = np.cos(nd[:, 0]), s * d
    w1, v1 = q[:, 0], q[:, 1:]
    return np.concatenate([(w0 * w1 - (v0 * v1).sum(1))[:, None], w0[:, None] * v1 + w1[:, None] * v0 + np.cross(v0, v1)], 1)


@pytest.mark.parametrize("solver", ["direct", "iterative"])
def test_one_lm_step_at_scale_equals_the_oracles(ctx, big, solver):
    """ONE Levenberg-Marquardt step on the first 131 075 observations of configs[2] (26 215 whole points, all 200
    cameras: what bench.py's cpu_baseline_lm_projected times) against the oracle's Schur iteration
    (oracle/pxo_lm_bench.c, OpenMP on the host cores): the cost at the linearisation point, the camera step
    (translations, intrinsics, rotations through the manifold), the point step and the cost of the candidate -- the
    solver's output at scale tied to something other than itself.  Direct: Cholesky of the 1593 x 1593 reduced system;
    iterative: the implicit-Schur conjugate gradients driven to 1e-12."""
    import os
    import pxo
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob, patches, ba_full = big
    n_img = 200
    n_pts = int(prob["obs_point"][131072 - 1]) + 1                       # whole points (observations are point-sorted)
    n_obs = int(np.searchsorted(prob["obs_point"], n_pts, side="left"))
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "obs_patch", "corners", "scales"):
        sub[k] = prob[k][:n_obs]
    sub["xyz"], sub["refs"] = prob["xyz"][:n_pts].copy(), prob["refs"][:n_pts].copy()
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    cmask, ptc = np.full(n_img, 0b0110, np.uint16), np.zeros(n_pts, np.uint8)
    # ---- the GPU step
    arena = PatchArena(ctx, n_obs, 16, 16, 128, np.float16, device_ptr=patches.data_ptr())     # the first n_obs patches of the big arena
    arena.upload(0, None, sub["corners"], sub["scales"])
    ba = BAProblem(ctx, arena, sub)
    extra = {} if solver == "direct" else dict(eta=0.0, linear_r_tolerance=1e-12, max_linear_solver_iterations=4000)
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pose_const, tmask, cmask, ptc,
                 options=lm_options(max_iterations=1, jacobi_scaling=0, use_inner_iterations=False, initial_radius=1e4,
                                    linear_solver=solver, **extra))
    assert s["iterations"] == 1 and s["num_successful"] == 1 and s["num_camera_unknowns"] == 8 * n_img - 7
    q1, t1, k1, X1 = ba.params()
    # ---- the oracle's step on the host cores
    sub["patches"] = patches[:n_obs].cpu().numpy()
    cfg, ls = pxo.cfg(), pxo.loss("cauchy", 0.25)
    out = pxo.ba_lm_iteration_schur(sub, cfg, ls, pose_const, tmask, cmask, ptc, radius=1e4, n_threads=os.cpu_count() or 1)
    assert out["rc"] == 0 and out["n_c"] == s["num_camera_unknowns"]
    tol = 1e-8 if solver == "direct" else 1e-7
    assert abs(s["initial_cost"] - out["cost"]) < 1e-10 * out["cost"]
    dp = out["delta_p"]
    assert np.abs((X1 - sub["xyz"]) - dp).max() < tol * np.abs(dp).max()
    dc, col = out["delta_c"], 0
    rot = np.zeros((n_img, 3))
    worst_t, worst_k = 0.0, 0.0
    for i in range(n_img):
        if pose_const[i]:
            continue
        rot[i] = dc[col:col + 3]; col += 3
        for a in range(3):
            if (tmask[i] >> a) & 1:
                continue
            worst_t = max(worst_t, abs((t1[i, a] - prob["tvec"][i, a]) - dc[col])); col += 1
    for c in range(n_img):
        for a in (0, 3):                                                  # SIMPLE_RADIAL: f and k free
            worst_k = max(worst_k, abs((k1[c, a] - prob["cam_params"][c, a]) - dc[col]) / max(1e-12, abs(dc[col]))); col += 1
    assert col == out["n_c"]
    assert worst_t < tol * np.abs(dc).max() and worst_k < 1e-6
    assert np.abs(q1 - _quat_plus(prob["qvec"], rot)).max() < tol
    # ---- the candidate's cost: the oracle evaluates the parameters the GPU step produced
    cand = dict(sub, qvec=q1, tvec=t1, cam_params=k1[:, :prob["cam_params"].shape[1]], xyz=X1)
    cost_c, _, _ = pxo.ba_eval_batch(cand, cfg, ls, n_threads=os.cpu_count() or 1)
    assert abs(s["final_cost"] - cost_c) < 1e-9 * cost_c and s["final_cost"] < 0.5 * s["initial_cost"]
    arena.close()
