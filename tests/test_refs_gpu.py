"""GPU parity: reference extraction (pxr_ba_compute_references) vs the oracle's restatement of
ReferenceExtractor::ComputeReference + RobustMeanIRLS (reference_extractor.h:238-272,
irls_optim.h:24-71).  Tolerance 1e-10 on descriptors / robust means; chosen observation equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_refs(prob, iters, l2=True):
    import pxo
    cfg = pxo.cfg(l2_normalize=l2)
    ls = pxo.loss("cauchy", 0.25)
    n_pts = len(prob["xyz"])
    C = prob["patches"].shape[-1]
    refs, means, chosen = np.zeros((n_pts, C)), np.zeros((n_pts, C)), np.full(n_pts, -1)
    for p in range(n_pts):
        obs = np.nonzero(prob["obs_point"] == p)[0]
        if len(obs) == 0:
            continue
        descs = []
        for i in obs:
            img = prob["obs_image"][i]
            cam = prob["image_camera"][img]
            patch = pxo.make_patch(prob["patches"][prob["obs_patch"][i]], prob["corners"][prob["obs_patch"][i]],
                                   prob["scales"][prob["obs_patch"][i]])
            K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
            f, *_ = pxo.ba_residual(patch, cfg, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img],
                                    prob["xyz"][p], prob["cam_params"][cam][:K], None, jac=False)
            descs.append(f)
        idx, ref, mean = pxo.compute_reference(np.array(descs), ls, iters, l2)
        refs[p], means[p], chosen[p] = ref, mean, obs[idx]
    return refs, means, chosen


@pytest.mark.parametrize("obs_per_point,noise", [(4, 0.3), (8, 0.3), (11, 0.5)])
def test_references_match_oracle(ctx, obs_per_point, noise):
    """tracks <= 8 use the register path, longer tracks the L2 path; per-image noise makes the
    descriptors of a track differ so that the IRLS has something to do."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    prob = synthetic.make_ba_problem(n_cams=12, n_points=37, obs_per_point=obs_per_point, seed=obs_per_point, noise=noise)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    chosen, mean = ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=100, keep_mean=True)
    refs = ba.d["refs"].download()
    refs_o, means_o, chosen_o = _oracle_refs(prob, 100)
    assert np.array_equal(chosen, chosen_o)
    assert np.abs(mean - means_o).max() < 1e-10
    assert np.abs(refs - refs_o).max() < 1e-10
    assert np.abs(np.linalg.norm(refs, axis=1) - 1).max() < 1e-12
    # the refreshed references are immediately usable by the residual kernel
    rec, *_ = ba.eval(interp_cfg(), with_jacobian=False)
    assert np.isfinite(rec.download()).all()


def test_point_without_observations_and_unnormalised(ctx):
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    prob = synthetic.make_ba_problem(n_cams=5, n_points=9, obs_per_point=3, seed=2, noise=0.2)
    keep = prob["obs_point"] != 4
    for k in ("obs_image", "obs_point", "obs_patch"):
        prob[k] = prob[k][keep]
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    before = ba.d["refs"].download()[4].copy()
    chosen, mean = ba.compute_references(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), iters=20, keep_mean=True)
    refs_o, means_o, chosen_o = _oracle_refs(prob, 20, l2=False)
    assert chosen[4] == -1 and np.array_equal(ba.d["refs"].download()[4], before)
    m = np.arange(9) != 4
    assert np.array_equal(chosen[m], chosen_o[m]) and np.abs(mean[m] - means_o[m]).max() < 1e-10


def test_keep_observations_and_nearest_references(ctx):
    """ReferenceExtractor keep_observations (reference_extractor.h:60,259-265) + FindNearestReferences
    (localization/src/nearest_references.h:20-52) vs the oracle: per-observation descriptors, and for
    query keypoints near each observation the nearest of its point's observation descriptors."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss, nearest_references
    prob = synthetic.make_ba_problem(n_cams=5, n_points=60, obs_per_point=4, seed=23, noise=0.05)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=20, keep_observations=True)
    desc = ba.obs_desc.download()
    cfg = pxo.cfg()
    n_obs = len(prob["obs_image"])
    # descriptors at the current projections (value-only interpolation, A19)
    rec, r, _, _ = ba.eval(interp_cfg(), with_jacobian=False, materialize=True)
    for i in (0, 7, n_obs - 1):
        p = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        img, pt = prob["obs_image"][i], prob["obs_point"][i]
        out = pxo.ba_residual(p, cfg, int(prob["cam_model"][prob["image_camera"][img]]), prob["qvec"][img], prob["tvec"][img],
                              prob["xyz"][pt], prob["cam_params"][prob["image_camera"][img]], np.zeros(128), jac=False)
        assert np.abs(desc[i] - out[0]).max() < 1e-12
    # query: every observation's own patch with a perturbed keypoint; candidates = its point's observations
    rng = np.random.default_rng(4)
    order = np.argsort(prob["obs_point"], kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(prob["obs_point"], minlength=60))])
    kps = prob["centers"] + rng.normal(0, 0.4, (n_obs, 2))
    cand_ptr = np.concatenate([[0], np.cumsum([ptr[p + 1] - ptr[p] for p in prob["obs_point"]])])
    cand_index = np.concatenate([order[ptr[p]:ptr[p + 1]] for p in prob["obs_point"]])
    best, dist, win = nearest_references(ctx, arena, interp_cfg(), kps, np.arange(n_obs), cand_ptr, ba.obs_desc,
                                         cand_index=cand_index, want_desc=True)
    for i in range(0, n_obs, 7):
        p = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        rows = cand_index[cand_ptr[i]:cand_ptr[i + 1]]
        b, d = pxo.nearest_reference(p, cfg, kps[i], desc[rows])
        assert best[i] == rows[b] and abs(dist[i] - d) <= 1e-12 * max(d, 1e-12)
        assert np.array_equal(win[i], desc[best[i]])
    assert all(best[i] in cand_index[cand_ptr[i]:cand_ptr[i + 1]] for i in range(n_obs))


@pytest.mark.parametrize("channels,dtype,l2", [(3, np.float16, False), (3, np.float32, True), (1, np.float16, False), (1, np.float64, False)])
def test_few_channel_references_interpolation_and_nearest(ctx, channels, dtype, l2):
    """Image-intensity features (dense_features.model.name = "image": 3 or 1 channels): FeatureReferenceBundleOptimizer
    registers (3, 1) and (1, 1) (feature_reference_bundle_optimizer.h:13-16) and ReferenceExtractor runs on any channel
    count through the dynamic interpolator (reference_extractor.h:139-143), which below 8 channels is the scalar Ceres
    bicubic (interpolation.h:222-268).  Reference extraction, batched interpolation with Jacobians, nearest references and
    the residual blocks of the feature-reference BA against the oracle."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, interpolate, make_loss, nearest_references
    prob = synthetic.make_ba_problem(n_cams=7, n_points=40, obs_per_point=5, seed=channels + 10, noise=0.3, channels=channels, dtype=dtype)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cfg_g, cfg_o = interp_cfg(l2_normalize=l2), pxo.cfg(l2_normalize=l2)
    chosen, mean = ba.compute_references(cfg_g, make_loss("cauchy", [0.25]), iters=30, keep_mean=True, keep_observations=True)
    refs, desc = ba.d["refs"].download(), ba.obs_desc.download()
    ls = pxo.loss("cauchy", 0.25)
    n_obs = len(prob["obs_image"])
    od = np.zeros((n_obs, channels))
    for i in range(n_obs):
        img, pt = prob["obs_image"][i], prob["obs_point"][i]
        cam = prob["image_camera"][img]
        patch = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
        od[i] = pxo.ba_residual(patch, cfg_o, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img], prob["xyz"][pt],
                                prob["cam_params"][cam][:K], None, jac=False)[0]
    assert np.abs(desc - od).max() < 1e-12
    for p in range(len(prob["xyz"])):
        obs = np.nonzero(prob["obs_point"] == p)[0]
        idx, ref, mu = pxo.compute_reference(od[obs], ls, 30, l2)
        assert chosen[p] == obs[idx]
        assert np.abs(refs[p] - ref).max() < 1e-12 and np.abs(mean[p] - mu).max() < 1e-12
    # the residual blocks of the feature-reference BA on these features (the Schur solver only sees the six-scalar record)
    rec, r, gx, gy = ba.eval(cfg_g, with_jacobian=True, materialize=True)
    r = r.download()
    assert np.abs(r - (od - refs[prob["obs_point"]])).max() < 1e-12
    # batched interpolation with Jacobians, and nearest references, at perturbed keypoints
    rng = np.random.default_rng(1)
    kps = prob["centers"] + rng.normal(0, 0.4, (n_obs, 2))
    f, J = interpolate(ctx, arena, cfg_g, kps, np.arange(n_obs), jacobian=True)
    for i in range(0, n_obs, 9):
        patch = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        fo, gxo, gyo, _ = pxo.patch_eval(patch, kps[i], cfg_o)
        assert np.abs(f[i] - fo).max() < 1e-12
        assert np.abs(J[i] - np.stack([gxo, gyo], 1)).max() < 1e-9 * max(1.0, np.abs(gxo).max(), np.abs(gyo).max())
    ptr = np.arange(0, n_obs + 1, 5)
    cand_ptr = np.concatenate([[0], np.cumsum(np.full(n_obs, 5))])
    cand_index = np.concatenate([np.arange(ptr[p], ptr[p + 1]) for p in prob["obs_point"]])
    best, dist, win = nearest_references(ctx, arena, cfg_g, kps, np.arange(n_obs), cand_ptr, od, cand_index=cand_index, want_desc=True)
    for i in range(0, n_obs, 9):
        patch = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        rows = cand_index[cand_ptr[i]:cand_ptr[i + 1]]
        b, d = pxo.nearest_reference(patch, cfg_o, kps[i], od[rows])
        assert best[i] == rows[b] and abs(dist[i] - d) <= 1e-12 * max(d, 1e-12) and np.array_equal(win[i], od[best[i]])
