"""Reference extraction (bundle_adjustment/src/reference_extractor.h:172-300 with RobustMeanIRLS,
base/src/irls_optim.h:24-71) on the seeded scenes of tests/cases/refs_cases.py.

The oracle's extraction = per point: the VISIBLE observations (those whose keypoint has a patch in the view,
reference_extractor.h:172-214), the descriptor of each at the current projection (value only), the robust mean, the
observation closest to it (first minimum, :249-272), the squared distances as per-observation costs.
  * CPU: properties of that restatement -- no reference for a point with nothing visible, the chosen observation is a visible
    one of that point and minimises the distance to the mean, descriptors are unit vectors under l2_normalize;
  * GPU: pxr_ba_compute_references (through BAProblem.compute_references) reproduces the oracle's choice and descriptors.
PARITY UNPINNED: the reference has no expected values for this step and its headers cannot be compiled here."""
import numpy as np
import pytest

from cases import refs_cases as G

NAMES = [s[0] for s in G.SCENES]


def _visible(prob, has):
    keep = np.nonzero(has)[0]
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "obs_patch"):
        sub[k] = prob[k][keep]
    return sub, keep


def _oracle_extraction(prob, opts, has):
    import pxo
    cfg = pxo.cfg(l2_normalize=opts["l2_normalize"], use_float_simd=opts["use_float_simd"])
    ls = pxo.loss(opts["loss"][0], opts["loss"][1]) if opts["loss"][0] != "trivial" else pxo.loss("trivial")
    n_pts, ch = len(prob["xyz"]), prob["patches"].shape[3]
    out = dict(has_ref=np.zeros(n_pts, bool), src_obs=np.full(n_pts, -1), src_image=np.full(n_pts, -1),
               descriptor=np.zeros((n_pts, ch)), obs_cost=np.full(len(prob["obs_image"]), np.nan), descs={})
    for p in range(n_pts):
        obs = np.nonzero((prob["obs_point"] == p) & has)[0]
        if len(obs) == 0:
            continue
        descs = []
        for i in obs:
            img = prob["obs_image"][i]
            cam = prob["image_camera"][img]
            q = prob["obs_patch"][i]
            patch = pxo.make_patch(prob["patches"][q], prob["corners"][q], prob["scales"][q])
            K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
            f, *_ = pxo.ba_residual(patch, cfg, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img], prob["xyz"][p],
                                    prob["cam_params"][cam][:K], None, jac=False)
            descs.append(f)
        descs = np.array(descs)
        idx, ref, mean = pxo.compute_reference(descs, ls, opts["iters"], opts["l2_normalize"])
        out["has_ref"][p] = True
        out["src_obs"][p], out["src_image"][p] = obs[idx], prob["obs_image"][obs[idx]]
        out["descriptor"][p] = ref if opts["closest_to_robust_mean"] else mean
        out["obs_cost"][obs] = ((descs - mean) ** 2).sum(1)
        out["descs"][p] = (obs, descs, mean)
    return out


@pytest.mark.parametrize("name", NAMES)
def test_oracle_extraction_properties(name):
    prob, opts, has = G.scene(name)
    got = _oracle_extraction(prob, opts, has)
    for p in range(len(prob["xyz"])):
        vis = np.nonzero((prob["obs_point"] == p) & has)[0]
        assert bool(got["has_ref"][p]) == (len(vis) > 0)
        if len(vis) == 0:
            continue
        obs, descs, mean = got["descs"][p]
        assert got["src_obs"][p] in vis
        cost = got["obs_cost"][obs]
        assert cost[list(obs).index(got["src_obs"][p])] == cost.min()                   # closest to the robust mean
        if opts["l2_normalize"]:
            assert np.abs((descs ** 2).sum(1) - 1.0).max() < 1e-12 and abs(mean @ mean - 1.0) < 1e-12
        if opts["closest_to_robust_mean"]:
            assert np.array_equal(got["descriptor"][p], descs[list(obs).index(got["src_obs"][p])])
    assert np.isnan(got["obs_cost"][~has]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_the_oracles_extraction(ctx, name):
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    prob, opts, has = G.scene(name)
    gold = _oracle_extraction(prob, opts, has)
    sub, keep = _visible(prob, has)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, sub)
    loss = make_loss(opts["loss"][0], [opts["loss"][1]] if opts["loss"][0] != "trivial" else [])
    chosen, mean = ba.compute_references(interp_cfg(l2_normalize=opts["l2_normalize"], use_float_simd=opts["use_float_simd"]), loss,
                                         iters=opts["iters"], keep_mean=True)
    refs = ba.d["refs"].download()
    ok = gold["has_ref"]
    assert np.array_equal(chosen >= 0, ok)
    assert np.array_equal(keep[chosen[ok]], gold["src_obs"][ok])
    out = refs if opts["closest_to_robust_mean"] else mean
    assert np.abs(out[ok] - gold["descriptor"][ok]).max() < 1e-10
