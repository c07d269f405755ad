"""Query refinement on the seeded cases of tests/cases/localization_cases.py against the oracle's Python
restatement (oracle/pxo_loc.py) of SingleQueryKeypointOptimizer::RunQuery + QueryKeypointOptimizer::ParameterizeKeypoint,
SingleQueryBundleOptimizer::RunQuery + QueryBundleOptimizer::ParameterizeQuery, and FindNearestReferences (pxo.nearest_reference):
  * CPU: the host logic of api.localization (which residual blocks a query contributes, in which order, for the three kinds of
    reference containers, inlier masks and patch indices; the constant camera parameters of the query BA) and the oracle's C
    box bounds reproduce the restated problems; pxo.nearest_reference is the argmin of the squared descriptor distances;
  * GPU: api.localization.find_nearest_references (pxr_nearest_references) reproduces the oracle's choice.
PARITY UNPINNED: the reference has no test for these optimizers and they cannot be compiled here (Ceres / COLMAP absent)."""
import numpy as np
import pytest

from cases import localization_cases as G


def _tag(i, k):
    return 1000.0 * i + (999 if k < 0 else k)


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
    """the containers the reference's three RunQuery overloads take, with the tags of _tag()"""
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
    import pxo_loc
    c = G.qka_case(seed)
    o = pxo_loc.qka_setup(c["kp"], c["corners"], c["scales"], 16, 16, c["sparse"], c["bound"], c["mode"], c["ref_count"], c["patch_idxs"],
                          c["inliers"])
    gold = dict(solved=o["solved"], blk_kp=np.array([b[0] for b in o["blocks"]], int), blk_tag=np.array([_tag(*b) for b in o["blocks"]]),
                lower=o["lower"], upper=o["upper"])
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
    import pxo_loc
    c = G.qba_case(seed)
    o = pxo_loc.qba_setup(len(c["points"]), c["model"], *c["refine"], c["mode"], c["ref_count"], c["inliers"])
    gold = dict(solved=o["solved"], blk_point=np.array([b[0] for b in o["blocks"]], int), blk_tag=np.array([_tag(*b) for b in o["blocks"]]),
                point_const=o["point_const"], camera_mask=o["camera_mask"])
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
    assert np.array_equal(gold["point_const"].astype(bool), inlier)
    in_problem = np.zeros(len(c["points"]), bool)
    in_problem[rows] = True
    assert not (in_problem & ~inlier).any()
    camera = reconstruction.Camera(1, c["model"], 1000, 1000, list(c["params"]))
    options = dict(refine_focal_length=bool(c["refine"][0]), refine_principal_point=bool(c["refine"][1]), refine_extra_params=bool(c["refine"][2]))
    mask = localization._qba_camera_mask(camera, options)
    assert mask == gold["camera_mask"]


def _oracle_nearest(c):
    import pxo
    cfg = pxo.cfg(l2_normalize=bool(c["l2"]))
    ptr = np.concatenate([[0], np.cumsum(c["cand_count"])])
    chosen, desc, query = [], [], []
    for i in range(len(c["kp"])):
        patch = pxo.make_patch(c["patches"][i], c["corners"][i], c["scales"][i])
        best, _ = pxo.nearest_reference(patch, cfg, c["kp"][i], c["cand"][ptr[i]:ptr[i + 1]])
        chosen.append(int(best)); desc.append(c["cand"][ptr[i] + int(best)])
        query.append(pxo.patch_eval(patch, c["kp"][i], cfg, want_grad=False)[0])
    return np.array(chosen), np.array(desc), np.array(query), ptr


@pytest.mark.parametrize("seed", range(G.N_NEAREST))
def test_oracle_nearest_reference_is_the_argmin_of_the_descriptor_distance(seed):
    """nearest_references.h:36-49: the candidate with the smallest squared distance to the keypoint's interpolated descriptor."""
    c = G.nearest_case(seed)
    chosen, desc, query, ptr = _oracle_nearest(c)
    for i in range(len(c["kp"])):
        d = ((c["cand"][ptr[i]:ptr[i + 1]] - query[i]) ** 2).sum(1)
        assert chosen[i] == int(np.argmin(d))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(G.N_NEAREST))
def test_gpu_nearest_references(ctx, seed):
    from pixsfm_amd.api import base, features, localization
    c = G.nearest_case(seed)
    _, want, _, _ = _oracle_nearest(c)
    n = len(c["kp"])
    fmap = features.FeatureMap.from_arrays(c["patches"], list(range(n)), c["corners"], c["scales"][0])
    ptr = np.concatenate([[0], np.cumsum(c["cand_count"])])
    refs = {100 + i: features.Reference(0, 0, np.zeros(128), [c["cand"][r] for r in range(ptr[i], ptr[i + 1])]) for i in range(n)}
    got = localization.find_nearest_references(fmap, refs, c["kp"], [100 + i for i in range(n)], base.InterpolationConfig({"l2_normalize": bool(c["l2"])}), ctx=ctx)
    for i in range(n):
        assert np.array_equal(got[i].reshape(-1), want[i])
