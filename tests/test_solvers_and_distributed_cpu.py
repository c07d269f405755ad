"""CPU: (1) the oracle's LM solvers converge to the rendered optimum on small synthetic problems
(they are the parity target of the GPU solvers, so they must themselves be sane); (2) the
multi-GPU decomposition of SURVEY 8e -- points sharded, cameras replicated, all-reduce(sum) of
the partial reduced camera systems -- reproduces the single-rank reduced system, run with two
real processes over torch.distributed's gloo backend."""
import os

import numpy as np
import pytest

import pxo
import pxo_ka


def _gauge(prob):
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return pose_const, tmask, np.full(n_cam, 0b0110, np.uint16), np.zeros(n_pt, np.uint8)


def test_oracle_ba_lm_converges_and_respects_constness():
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=5, n_points=40, obs_per_point=4, seed=3, channels=32, patch_size=16)
    g = _gauge(prob)
    s, q, t, k, X = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), *g, pxo.lm_options(max_iterations=25))
    assert s["final_cost"] < 1e-3 * s["initial_cost"] and s["num_successful"] >= 4
    assert np.abs(q[0] - prob["qvec"][0]).max() < 1e-15 and np.array_equal(t[0], prob["tvec"][0])   # NormalizeQvec only
    assert t[1][0] == prob["tvec"][1][0] and t[1][1] != prob["tvec"][1][1]
    assert np.array_equal(k[:, 1:3], prob["cam_params"][:, 1:3])                  # principal point constant
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12                     # QuaternionManifold keeps |q| = 1
    # the cost the solver reports is the cost of the returned parameters
    ref = dict(prob, qvec=q, tvec=t, cam_params=k, xyz=X)
    c, _, _ = pxo.ba_eval_batch(ref, pxo.cfg(), pxo.loss("cauchy", 0.25))
    assert abs(c - s["final_cost"]) < 1e-12 * max(1.0, c)


def test_oracle_ba_lm_trivial_loss_and_tolerances():
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=4, n_points=30, obs_per_point=3, seed=9, channels=16)
    g = _gauge(prob)
    s, *_ = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("trivial", 1.0), *g,
                         pxo.lm_options(max_iterations=50, function_tolerance=1e-6))
    assert s["termination"] == 0 and s["iterations"] < 50
    s2, *_ = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("trivial", 1.0), *g, pxo.lm_options(max_iterations=3))
    assert s2["termination"] == 1 and s2["iterations"] == 3


def test_oracle_ka_lm_recovers_the_track_alignment():
    from pixsfm_amd import synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=8, track_len=5, seed=6, channels=32, max_kps_per_problem=20, sigma=0.8)
    kp, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0,
                               pxo.lm_options(parameter_tolerance=1e-8, max_iterations=50))
    root = prob["node_const"].astype(bool)
    assert np.array_equal(kp[root], prob["kp"][root])
    off = (prob["kp"][root] - prob["true_xy"][root]).repeat(5, axis=0)
    err = np.linalg.norm(kp - (prob["true_xy"] + off), axis=1)
    assert np.median(err) < 0.02 and np.median(np.linalg.norm(prob["kp"] - (prob["true_xy"] + off), axis=1)) > 0.5
    assert np.abs(kp - prob["kp"]).max() <= 4.0 + 1e-12                            # bound (main.py:78)


# ---- two-process gloo test of the point-sharded reduced camera system ---------------------------------
def _partial_reduced_system(prob, lam=1e-3):
    """numpy emulation of what one rank contributes: camera-side unknowns = tvec of every image,
    point-side = xyz; robustified blocks from the oracle; S_r = U_r - sum_p W_p (V_p + lam I)^-1 W_p^T."""
    cost, r, J = pxo.ba_eval_batch(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    n_img = len(prob["image_camera"])
    S = np.zeros((3 * n_img, 3 * n_img)); rhs = np.zeros(3 * n_img)
    ls = pxo.loss("cauchy", 0.25)
    for p in np.unique(prob["obs_point"]):
        V = lam * np.eye(3); gp = np.zeros(3); Ws = []
        for i in np.nonzero(prob["obs_point"] == p)[0]:
            s = float(r[i] @ r[i])
            rt, Jt = pxo.corrector(s, pxo.loss_eval(ls, s), r[i], J[i][:, 4:10])   # columns t(3), X(3)
            Jc, Jp = Jt[:, :3], Jt[:, 3:]
            im = prob["obs_image"][i]
            S[3 * im:3 * im + 3, 3 * im:3 * im + 3] += Jc.T @ Jc
            rhs[3 * im:3 * im + 3] += Jc.T @ rt
            V += Jp.T @ Jp; gp += Jp.T @ rt
            Ws.append((im, Jc.T @ Jp))
        Vi = np.linalg.inv(V)
        for ia, Wa in Ws:
            rhs[3 * ia:3 * ia + 3] -= Wa @ Vi @ gp
            for ib, Wb in Ws:
                S[3 * ia:3 * ia + 3, 3 * ib:3 * ib + 3] -= Wa @ Vi @ Wb.T
    return cost, S, rhs


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from pixsfm_amd import synthetic
    from pixsfm_amd.parallel import shard_ba_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synthetic.make_ba_problem(n_cams=4, n_points=14, obs_per_point=3, seed=12, channels=16, patch_size=16)
    shard, ids = shard_ba_problem(prob, rank, world)
    cost, S, rhs = _partial_reduced_system(shard)
    buf = torch.from_numpy(np.concatenate([S.reshape(-1), rhs, [cost]]))   # one packed buffer, like [S | rhs]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_of_partial_reduced_systems(tmp_path):
    import torch.multiprocessing as mp
    from pixsfm_amd import synthetic
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    prob = synthetic.make_ba_problem(n_cams=4, n_points=14, obs_per_point=3, seed=12, channels=16, patch_size=16)
    cost, S, rhs = _partial_reduced_system(prob)
    want = np.concatenate([S.reshape(-1), rhs, [cost]])
    for r in range(2):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()
