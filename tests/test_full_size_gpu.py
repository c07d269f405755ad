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
    # the summary's final cost is the cost of the parameters left on the device
    ba.eval(interp_cfg(), with_jacobian=False)
    assert abs(ba.cost(make_loss("cauchy", [0.25])) - s["final_cost"]) < 1e-9 * s["final_cost"]
