"""Query refinement against vectors produced by the REFERENCE's own code compiled in place
(tests/golden/make_golden_localization.py, oracle/ref_loc_shim.cc): SingleQueryKeypointOptimizer::RunQuery +
ParameterizeKeypoint, SingleQueryBundleOptimizer::RunQuery + ParameterizeQuery against a recording ceres::Problem, and
FindNearestReferences.
  * CPU: the host logic of api.localization (which residual blocks a query contributes, in which order, for the three kinds of
    reference containers, inlier masks and patch indices; the constant camera parameters of the query BA) and the oracle's
    box bounds reproduce the recorded problems; pxo.nearest_reference reproduces FindNearestReferences;
  * GPU: api.localization.find_nearest_references (pxr_nearest_references) does;
  * live, when oracle/_ref/libpxo_ref_loc.so is present: the vectors are what the reference yields now."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden_localization", os.path.join(HERE, "golden", "make_golden_localization.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)


def _gold(prefix):
    z = np.load(os.path.join(HERE, "golden", "localization_ref.npz"))
    return {k.split("|", 1)[1]: z[k] for k in z.files if k.startswith(prefix + "|")}


class _Patch:
    def __init__(self, corner, scale):
        self.corner, self.scale, self.shape = corner, scale, (16, 16, 128)


class _Map:
    """stands in for api.features.FeatureMap: the set-up code only fetches patches by index"""

    def __init__(self, corners, scales):
        self.patches = [_Patch(c, s) for c, s in zip(corners, scales)]

    def fpatch(self, i):
        return self.patches[int(i)]


def _tagged(tag):
    d = np.zeros((1, 128))
    d[0, 0], d[0, 1] = tag, 0.5
    return d


def _references(mode, ref_count):
    """the containers the reference's three RunQuery overloads take, with make_golden_localization's tags"""
    from pixsfm_amd.api import features
    n = len(ref_count)
    if mode == 0:
        return [_tagged(1000.0 * i + 999) for i in range(n)]
    if mode == 1:
        return [[_tagged(1000.0 * i + r) for r in range(ref_count[i])] for i in range(n)]
    return [features.Reference(0, 0, _tagged(1000.0 * i + 999), [_tagged(1000.0 * i + r) for r in range(ref_count[i])]) for i in range(n)]


@pytest.mark.parametrize("seed", range(G.N_QKA))
def test_query_keypoint_adjustment_problem(seed):
    import pxo_ka
    from pixsfm_amd.api import localization
    c, gold = G.qka_case(seed), _gold("qka%d" % seed)
    fmap = _Map(c["corners"], c["scales"])
    kp = np.ascontiguousarray(c["kp"])
    rows, patches, prob = localization._build_problem(kp, fmap, _references(c["mode"], c["ref_count"]), c["patch_idxs"], c["inliers"])
    assert bool(gold["solved"]) == (len(prob["unary_node"]) > 0)
    # residual blocks: (keypoint, descriptor) in the order the reference adds them
    assert np.array_equal(np.asarray(rows, int)[prob["unary_node"]], gold["blk_kp"])
    assert np.array_equal(prob["unary_ref"][:, 0] if len(prob["unary_ref"]) else np.zeros(0), gold["blk_tag"])
    # box bounds of the keypoints in the problem; the others are untouched by the reference
    in_problem = np.zeros(len(kp), bool)
    in_problem[rows] = True
    has_bounds = ~np.isnan(gold["lower"][:, 0])
    if not bool(gold["solved"]):
        return
    if c["sparse"] or c["bound"] > 0:
        assert np.array_equal(has_bounds, in_problem)
        own = np.arange(len(kp)) if c["patch_idxs"] is None else c["patch_idxs"]
        b = pxo_ka.node_bounds(kp[rows], c["corners"][own][rows], c["scales"][own][rows], 16, 16, c["bound"])
        assert np.abs(b[:, :2] - gold["lower"][rows]).max() < 1e-12 and np.abs(b[:, 2:] - gold["upper"][rows]).max() < 1e-12
    else:
        assert not has_bounds.any()        # dense map and no bound: ParameterizeKeypoint sets nothing (query_keypoint_optimizer.h:145)


@pytest.mark.parametrize("seed", range(G.N_QBA))
def test_query_bundle_adjustment_problem(seed):
    from pixsfm_amd.api import localization, reconstruction
    c, gold = G.qba_case(seed), _gold("qba%d" % seed)
    fmap = _Map(c["corners"], c["scales"])
    rows, patches, xyz, refs = localization._qba_observations(c["points"], fmap, _references(c["mode"], c["ref_count"]), c["inliers"], c["patch_idxs"])
    assert bool(gold["solved"]) == (len(rows) > 0)
    assert np.array_equal(np.asarray(rows, int), gold["blk_point"])
    assert np.array_equal(np.array([r[0] for r in refs]), gold["blk_tag"])
    own = np.arange(len(c["points"])) if c["patch_idxs"] is None else c["patch_idxs"]
    assert all(p is fmap.patches[own[i]] for p, i in zip(patches, rows))
    if not bool(gold["solved"]):
        return
    # every inlier point is held constant, the pose is on the quaternion manifold.  (An inlier with an EMPTY descriptor list
    # has no residual block, and the reference still calls SetParameterBlockConstant on it, single_query_bundle_optimizer.h:
    # 169-175 -- real Ceres aborts on a block that is not in the problem, so such input is outside the contract; the product
    # simply has no observation for it.)
    inlier = np.ones(len(c["points"]), bool) if c["inliers"] is None else c["inliers"].astype(bool)
    assert np.array_equal(gold["point_const"].astype(bool), inlier) and int(gold["quaternion"]) == 1
    in_problem = np.zeros(len(c["points"]), bool)
    in_problem[rows] = True
    assert not (in_problem & ~inlier).any()
    camera = reconstruction.Camera(1, c["model"], 1000, 1000, list(c["params"]))
    options = dict(refine_focal_length=bool(c["refine"][0]), refine_principal_point=bool(c["refine"][1]), refine_extra_params=bool(c["refine"][2]))
    mask = localization._qba_camera_mask(camera, options)
    K = len(c["params"])
    want = (1 << K) - 1 if int(gold["camera_const"]) == -1 else int(gold["camera_const"])
    assert mask == want


@pytest.mark.parametrize("seed", range(G.N_NEAREST))
def test_oracle_nearest_references(seed):
    import pxo
    c, gold = G.nearest_case(seed), _gold("nearest%d" % seed)
    cfg = pxo.cfg(l2_normalize=bool(c["l2"]))
    ptr = np.concatenate([[0], np.cumsum(c["cand_count"])])
    for i in range(len(c["kp"])):
        patch = pxo.make_patch(c["patches"][i], c["corners"][i], c["scales"][i])
        out = pxo.nearest_reference(patch, cfg, c["kp"][i], c["cand"][ptr[i]:ptr[i + 1]])
        best = out[0] if isinstance(out, tuple) else out
        assert int(best) == gold["chosen"][i]
        assert np.array_equal(c["cand"][ptr[i] + int(best)], gold["descriptor"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(G.N_NEAREST))
def test_gpu_nearest_references(ctx, seed):
    from pixsfm_amd.api import base, features, localization
    c, gold = G.nearest_case(seed), _gold("nearest%d" % seed)
    n = len(c["kp"])
    fmap = features.FeatureMap.from_arrays(c["patches"], list(range(n)), c["corners"], c["scales"][0])
    ptr = np.concatenate([[0], np.cumsum(c["cand_count"])])
    refs = {100 + i: features.Reference(0, 0, np.zeros(128), [c["cand"][r] for r in range(ptr[i], ptr[i + 1])]) for i in range(n)}
    got = localization.find_nearest_references(fmap, refs, c["kp"], [100 + i for i in range(n)], base.InterpolationConfig({"l2_normalize": bool(c["l2"])}), ctx=ctx)
    for i in range(n):
        assert np.array_equal(got[i].reshape(-1), gold["descriptor"][i])


def test_golden_vectors_are_what_the_reference_yields_now():
    if not os.path.isfile(G.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_loc.so not built (reference tree absent)")
    for kind, n, case, run in (("qka", G.N_QKA, G.qka_case, G.run_qka), ("qba", G.N_QBA, G.qba_case, G.run_qba), ("nearest", G.N_NEAREST, G.nearest_case, G.run_nearest)):
        for s in range(n):
            now, gold = run(case(s)), _gold("%s%d" % (kind, s))
            for k, v in now.items():
                assert np.array_equal(v, gold[k], equal_nan=True), (kind, s, k)
