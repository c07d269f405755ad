"""CPU: host-side logic of the pixsfm-compatible API (no GPU calls): graph labelling
(base/src/graph.cc:126-256), edge enumeration (topological_keypoint_optimizer.h:97-175),
BA problem flattening + parameterisation (bundle_optimizer.h:139-165,247-453), setups, confs."""
import numpy as np
import pytest

from pixsfm_amd import synthetic
from pixsfm_amd.api import base, features
from pixsfm_amd.api.bundle_adjustment import (BundleAdjuster, BundleAdjustmentSetup, FeatureView, _FlatBA,
                                              default_problem_setup, find_problem_labels)
from pixsfm_amd.api.keypoint_adjustment import (KeypointAdjuster, KeypointAdjustmentSetup, build_edges,
                                                build_matching_graph)
from pixsfm_amd.api.reconstruction import reconstruction_from_flat


def _toy_graph():
    # three images, two physical tracks; a conflicting weak match tries to pull two keypoints of
    # image "a" into one track and must be rejected by the one-feature-per-image rule
    pairs = [("a", "b"), ("b", "c"), ("a", "c"), ("a", "b")]
    matches = [np.array([[0, 0], [1, 1]]), np.array([[0, 0], [1, 1]]), np.array([[0, 0]]), np.array([[1, 0]])]
    scores = [np.array([0.9, 0.8]), np.array([0.7, 0.95]), np.array([0.6]), np.array([0.1])]
    return build_matching_graph(pairs, matches, scores)


def test_graph_and_labels():
    g = _toy_graph()
    assert len(g.nodes) == 6 and g.image_name_to_id == {"a": 0, "b": 1, "c": 2}
    labels = base.compute_track_labels(g)
    key = {(g.image_id_to_name[n.image_id], n.feature_idx): labels[n.node_idx] for n in g.nodes}
    assert key[("a", 0)] == key[("b", 0)] == key[("c", 0)]
    assert key[("a", 1)] == key[("b", 1)] == key[("c", 1)] != key[("a", 0)]
    for t in set(labels):                                       # never two features of one image in a track
        imgs = [g.nodes[i].image_id for i in range(6) if labels[i] == t]
        assert len(imgs) == len(set(imgs))
    scores = base.compute_score_labels(g, labels)
    roots = base.compute_root_labels(g, labels, scores)
    assert sum(roots) == 2
    for t in set(labels):
        ids = [i for i in range(6) if labels[i] == t]
        assert roots[max(ids, key=lambda i: scores[i])]


def test_build_edges_variants():
    g = _toy_graph()
    labels = base.compute_track_labels(g)
    roots = base.compute_root_labels(g, labels, base.compute_score_labels(g, labels))
    kps = {"a": np.zeros((2, 2)), "b": np.zeros((2, 2)), "c": np.zeros((2, 2))}
    src, dst, w = build_edges(g, kps, labels, roots)
    want = [(n.node_idx, m.node_idx, m.sim) for n in g.nodes for m in n.out_matches if labels[n.node_idx] == labels[m.node_idx]]
    assert list(zip(src, dst, w)) == want and len(want) == 5      # the conflicting a1-b0 match is inter-track
    _, _, w1 = build_edges(g, kps, labels, roots, weight_by_sim=False)
    assert set(w1) == {1.0}
    s2, d2, _ = build_edges(g, kps, labels, roots, root_edges_only=True)
    assert all(roots[a] or roots[b] for a, b in zip(s2, d2)) and len(s2) < len(src)
    # topological_reference preset: star to the root, missing root edges are added with weight 1
    s3, d3, w3 = build_edges(g, kps, labels, roots, weight_by_sim=False, root_edges_only=True, root_regularize_weight=1.0)
    for t in set(labels):
        ids = {i for i in range(6) if labels[i] == t}
        r = [i for i in ids if roots[i]][0]
        touched = {a for a, b in zip(s3, d3) if b == r or a == r} | {b for a, b in zip(s3, d3) if a == r or b == r}
        assert ids <= touched


def test_keypoint_setup_and_confs():
    g = _toy_graph()
    s = KeypointAdjustmentSetup()
    s.set_masked_nodes_constant(g, [True, False, False, False, False, True])
    assert s.is_node_constant(g.nodes[0]) and not s.is_node_constant(g.nodes[1])
    s.set_image_constant(1)
    assert all(s.is_node_constant(n) for n in g.nodes if n.image_id == 1)
    with pytest.raises(ValueError):
        s.set_masked_nodes_constant(g, [True])
    adj = KeypointAdjuster.create({"strategy": "topological_reference", "optimizer": {"bound": 2.0}})
    assert adj.conf["optimizer"]["bound"] == 2.0 and adj.conf["max_kps_per_problem"] == 50
    assert adj.conf["optimizer"]["solver"]["parameter_tolerance"] == 1e-5
    with pytest.raises(ValueError):
        KeypointAdjuster.create({"optimizer": {"no_such_option": 1}})
    with pytest.raises(ValueError):
        BundleAdjuster.create({"strategy": "patch_warp"})             # outside the accelerated path
    cm = BundleAdjuster.create({"strategy": "costmaps"})             # the low-memory strategy (main.py:218-238)
    assert cm.conf["costmaps"]["loss"]["name"] == "trivial" and cm.conf["costmaps"]["as_gradientfield"] is True
    from pixsfm_amd.api import CostMapExtractor
    assert CostMapExtractor({}).get_effective_channels() == 3 and CostMapExtractor({"as_gradientfield": False}).get_effective_channels() == 1
    assert CostMapExtractor({"compute_cross_derivative": True}).get_effective_channels() == 4       # costmap_extractor.h:52-61
    with pytest.raises(ValueError):
        CostMapExtractor({"upsampling_factor": 0.0})
    ba = BundleAdjuster.create({})
    assert ba.conf["optimizer"]["solver"]["use_inner_iterations"] is True and ba.conf["references"]["iters"] == 100


def _fset(prob, rec, patch_of):
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    return features.FeatureSet(fmaps)


def test_flat_ba_matches_reference_parameterisation():
    prob = synthetic.make_ba_problem(n_cams=5, n_points=20, obs_per_point=3, seed=4, channels=8, patch_size=8, model=3)
    rec, patch_of = reconstruction_from_flat(prob)
    fset = _fset(prob, rec, patch_of)
    setup = default_problem_setup(rec)
    assert setup.has_constant_pose(1) and setup.constant_tvec(2) == [0]
    flat = _FlatBA(rec, setup, FeatureView(fset, rec), {"refine_principal_point": False})
    assert len(flat.obs_image) == 60 and sorted(flat.point_ids) == list(range(1, 21))
    assert flat.pose_const.tolist() == [1, 0, 0, 0, 0] and flat.tvec_mask.tolist() == [0, 1, 0, 0, 0]
    assert set(flat.cam_mask.tolist()) == {0b00110}              # RADIAL: cx, cy constant; f, k1, k2 refined
    assert not flat.point_const.any()
    # an image left out of the setup: its observations disappear, the affected points become constant
    setup2 = BundleAdjustmentSetup()
    setup2.add_images([1, 2, 3, 4])
    setup2.set_constant_pose(1)
    flat2 = _FlatBA(rec, setup2, FeatureView(fset, rec), {})
    touched = {int(p) + 1 for p in prob["obs_point"][prob["obs_image"] == 4]}
    const_ids = {pid for pid, c in zip(flat2.point_ids, flat2.point_const) if c}
    assert const_ids == touched & set(flat2.point_ids)
    # ... unless they are added as variable points: then the outside observations come back with a constant pose
    for pid in touched:
        setup2.add_variable_point(pid)
    flat3 = _FlatBA(rec, setup2, FeatureView(fset, rec), {})
    assert len(flat3.obs_image) == 60 and not flat3.point_const.any()
    k5 = flat3.image_ids.index(5)
    assert flat3.pose_const[k5] == 1 and flat3.cam_mask[flat3.camera_ids.index(rec.images[5].camera_id)] == 0b11111
    with pytest.raises(ValueError):
        setup2.set_constant_tvec(1, [0])                          # already constant pose
    labels = find_problem_labels(rec, 10)
    assert labels[0] == -1 and labels[1] == 0 and labels[20] == 2


def test_flat_ba_observation_order_missing_patches_and_constness():
    """Observations are point-major and, inside a point, in Track().Elements() order (what ComputeReference iterates,
    reference_extractor.h:239-247); the optimiser's SetUp raises on a missing patch (GetFeaturePatch / references.at
    throw), the extractors skip it like GetVisibleObservations and leave the caller's reconstruction untouched."""
    prob = synthetic.make_ba_problem(n_cams=5, n_points=20, obs_per_point=3, seed=4, channels=8, patch_size=8)
    rec, patch_of = reconstruction_from_flat(prob)
    # scramble one track's element order: the flat problem must follow the TRACK, not the image order
    tr = rec.points3D[7].track
    tr.elements = tr.elements[::-1]
    fset = _fset(prob, rec, patch_of)
    setup = default_problem_setup(rec)
    flat = _FlatBA(rec, setup, FeatureView(fset, rec), {})
    assert (np.diff(flat.obs_point) >= 0).all()                               # point-major
    for k, pid in enumerate(flat.point_ids):
        keys = [flat.obs_keys[i] for i in np.nonzero(flat.obs_point == k)[0]]
        assert keys == [(el.image_id, el.point2D_idx) for el in rec.points3D[pid].track.elements], pid
    # a missing patch: error for the optimiser, skipped by the extractors
    name, p2d = rec.images[3].name, next(i for i, p in enumerate(rec.images[3].points2D) if p.has_point3D())
    del fset.fmaps[name].patches[p2d]
    with pytest.raises(ValueError, match="no feature patch"):
        _FlatBA(rec, setup, FeatureView(fset, rec), {})
    q_before = [np.array(rec.images[i].qvec, copy=True) * 1.0 for i in rec.images]
    for i in rec.images:
        rec.images[i].qvec = np.asarray(rec.images[i].qvec) * 2.0              # un-normalised on purpose
    flat_x = _FlatBA(rec, setup, FeatureView(fset, rec), {}, extractor=True)
    assert len(flat_x.obs_image) == 59
    assert all(np.array_equal(rec.images[i].qvec, 2.0 * q) for i, q in zip(rec.images, q_before))   # const Reconstruction


def test_qka_problem_assembly_and_validation():
    """Host side of the localization QKA mirror: term assembly, inlier masks, argument checks."""
    from pixsfm_amd.api import QueryKeypointAdjuster, features
    from pixsfm_amd.api.localization import _build_problem, resolve_level_indices
    rng = np.random.default_rng(0)
    patches = rng.normal(size=(4, 16, 16, 128)).astype(np.float16)
    fmap = features.FeatureMap.from_arrays(patches, np.arange(4), np.zeros((4, 2), np.int32), (1.0, 1.0))
    kps = rng.uniform(4, 12, (3, 2))
    refs = [rng.normal(size=128), features.Reference(0, 1, rng.normal(size=128), observations=[rng.normal(size=128)] * 2),
            [rng.normal(size=128), rng.normal(size=128), rng.normal(size=128)]]
    rows, plist, prob = _build_problem(kps, fmap, refs, [3, 1, 0], [True, True, True])
    assert rows == [0, 1, 2] and prob["unary_node"].tolist() == [0, 1, 1, 2, 2, 2]
    assert prob["unary_ref"].shape == (6, 128) and plist[0] is fmap.fpatch(3)
    rows, _, prob = _build_problem(kps, fmap, refs, None, [True, False, True])
    assert rows == [0, 2] and prob["unary_node"].tolist() == [0, 1, 1, 1] and np.array_equal(prob["kp"], kps[[0, 2]])
    with pytest.raises(ValueError):
        _build_problem(kps, fmap, refs[:2], None, None)
    with pytest.raises(ValueError):
        _build_problem(kps, fmap, refs, [0, 1], None)
    assert resolve_level_indices(None, 3) == [2, 1, 0] and resolve_level_indices([0], 3) == [0]
    adj = QueryKeypointAdjuster({"optimizer": {"bound": 2.0}})
    assert adj.conf["optimizer"]["bound"] == 2.0 and adj.conf["optimizer"]["solver"]["parameter_tolerance"] == 1e-5
    with pytest.raises(ValueError):
        adj.refine_stacked(kps, fmap, refs, None)


def test_native_graph_labelling_matches_the_python_restatement():
    """csrc/pxr_graph.cpp (ComputeTrackLabels / ScoreLabels / RootLabels, graph.cc:126-256) vs oracle/pxo_graph.py on
    random match graphs with conflicting matches (two keypoints of one image competing for a track) and score ties."""
    import pxo_graph
    from pixsfm_amd.api import base
    rng = np.random.default_rng(7)
    for trial in range(6):
        n_img, per_img = 6, 25
        g = base.Graph()
        names = ["im%d" % i for i in range(n_img)]
        for a in range(n_img):
            for b in range(a + 1, n_img):
                m = int(rng.integers(5, 30))
                matches = np.stack([rng.integers(0, per_img, m), rng.integers(0, per_img, m)], 1)
                sims = np.round(rng.uniform(0.2, 1.0, m), 1 if trial % 2 else 6)     # coarse similarities: many ties
                g.register_matches(names[a], names[b], matches, sims)
        tl = base.compute_track_labels(g)
        assert tl == pxo_graph.compute_track_labels(g)
        sc = base.compute_score_labels(g, tl)
        assert np.allclose(sc, pxo_graph.compute_score_labels(g, tl), rtol=0, atol=1e-12)
        assert base.compute_root_labels(g, tl, sc) == pxo_graph.compute_root_labels(g, tl, sc)
        # invariants: one feature per image per track, exactly one root per track
        seen = {}
        for nd, t in zip(g.nodes, tl):
            assert (t, nd.image_id) not in seen
            seen[(t, nd.image_id)] = 1
        roots = base.compute_root_labels(g, tl, sc)
        assert sorted(t for t, r in zip(tl, roots) if r) == sorted(set(tl))


def test_track_labelling_in_parallel_over_match_graph_components():
    """pxr_graph_track_labels solves the connected components of the match graph concurrently (they are independent
    sub-problems of the reference's one sequential pass, graph.cc:126-206).  Many components + enough matches to take the
    threaded path, similarity ties, conflicting matches and cross-component outliers: labels identical to the sequential
    Python restatement."""
    import ctypes as C
    from types import SimpleNamespace as NS
    import pxo_graph
    from pixsfm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n_groups, per, n_img = 1500, 12, 40
    n = n_groups * per
    node_image = rng.integers(0, n_img, n).astype(np.int32)
    g = rng.integers(0, n_groups, 60000)
    src = (g * per + rng.integers(0, per, len(g))).astype(np.int64)
    dst = (g * per + rng.integers(0, per, len(g))).astype(np.int64)
    out_s = rng.integers(0, n, 300).astype(np.int64)                     # a few matches between groups: larger components
    out_d = rng.integers(0, n, 300).astype(np.int64)
    src, dst = np.concatenate([src, out_s]), np.concatenate([dst, out_d])
    keep = (src != dst) & (node_image[src] != node_image[dst])
    src, dst = src[keep], dst[keep]
    sim = np.round(rng.uniform(0.1, 1.0, len(src)), 2)
    assert len(src) > 20000
    got, ntr = np.empty(n, np.int64), C.c_int64()
    _lib.check(lib.pxr_graph_track_labels(n, node_image.ctypes.data, len(src), src.ctypes.data, dst.ctypes.data, sim.ctypes.data,
                                          got.ctypes.data, C.byref(ntr)), "pxr_graph_track_labels")
    nodes = [NS(node_idx=i, image_id=int(node_image[i]), out_matches=[]) for i in range(n)]
    for a, b, s in zip(src, dst, sim):
        nodes[a].out_matches.append(NS(node_idx=int(b), sim=float(s)))
    want = np.array(pxo_graph.compute_track_labels(NS(nodes=nodes)))
    assert np.array_equal(got, want) and ntr.value == want.max() + 1
    seen = set()
    for i in range(n):                                                   # one feature per image per track
        assert (got[i], node_image[i]) not in seen
        seen.add((got[i], node_image[i]))


@pytest.mark.parametrize("weight_by_sim,root_edges_only,reg", [(True, False, -1.0), (False, True, -1.0), (False, True, 1.0),
                                                               (True, False, 0.5)])
def test_native_edge_construction_matches_the_python_restatement(weight_by_sim, root_edges_only, reg):
    """pxr_ka_build_edges (SetUp + AddIntraResiduals, A12) vs oracle/pxo_graph.build_edges, incl. node subsets."""
    import pxo_graph
    from pixsfm_amd.api import base
    from pixsfm_amd.api.keypoint_adjustment import build_edges
    rng = np.random.default_rng(11)
    g = base.Graph()
    names = ["im%d" % i for i in range(5)]
    for a in range(5):
        for b in range(a + 1, 5):
            m = int(rng.integers(8, 25))
            g.register_matches(names[a], names[b], np.stack([rng.integers(0, 20, m), rng.integers(0, 20, m)], 1),
                               rng.uniform(0.2, 1.0, m))
    tl = base.compute_track_labels(g)
    sc = base.compute_score_labels(g, tl)
    roots = base.compute_root_labels(g, tl, sc)
    for subset in (None, [i for i, t in enumerate(tl) if t % 2 == 0]):
        got = build_edges(g, None, tl, roots, subset, weight_by_sim, root_edges_only, reg)
        want = pxo_graph.build_edges(g, None, tl, roots, subset, weight_by_sim, root_edges_only, reg)
        assert got[0] == list(want[0]) and got[1] == list(want[1]) and np.array_equal(got[2], want[2])
        assert len(got[0]) > 0


def test_compiled_scene_dump_equals_the_python_walk():
    """pixsfm_amd._pxr_host (pybind11, csrc/pybind/pxr_host.cpp) reads the pycolmap-style objects in C++; _SceneDump must give the
    same arrays and the same patch objects with and without it -- on the stand-in classes, with point2Ds without a 3D point in
    both spellings (-1 here, 2^64 - 1 in pycolmap), observations without a patch, a dense map and a missing feature map."""
    import pytest
    from pixsfm_amd.api import bundle_adjustment as B, features
    from pixsfm_amd.api.reconstruction import Camera, Image, Point2D, Point3D, Reconstruction
    if B._host_module() is None:
        pytest.skip("_pxr_host was not built (no pybind11 / Python headers)")

    class PycolmapStylePoint2D:                      # kInvalidPoint3DId = 2^64 - 1
        def __init__(self, pid):
            self.point3D_id = pid if pid >= 0 else 2 ** 64 - 1

        def has_point3D(self):
            return self.point3D_id != 2 ** 64 - 1

    rng = np.random.default_rng(3)
    rec = Reconstruction()
    rec.add_camera(Camera(7, "SIMPLE_RADIAL", 100, 100, [50.0, 50, 50, 0]))
    n_img, n_pts = 6, 40
    images = [Image(10 + 3 * i, "im%d.jpg" % i, 7, [1, 0, 0, 0], [0, 0, 0]) for i in range(n_img)]
    for p in range(n_pts):
        rec.add_point3D(100 + 7 * p, Point3D(rng.normal(size=3)))
    for p in range(n_pts):
        for i in rng.choice(n_img, 3, replace=False):
            im = images[i]
            cls = PycolmapStylePoint2D if i % 2 else (lambda pid: Point2D([0, 0], pid))
            if rng.random() < 0.2:
                im.points2D.append(cls(-1))          # a keypoint without a 3D point
            im.points2D.append(cls(100 + 7 * p))
            rec.points3D[100 + 7 * p].track.add_element(im.image_id, len(im.points2D) - 1)
    for im in images:
        rec.add_image(im)
    fmaps = {}
    for k, im in enumerate(images):
        if k == 4:
            continue                                  # no feature map at all for this image
        if k == 2:
            fmaps[im.name] = features.FeatureMap.dense(np.zeros((4, 4, 8), np.float16), (1.0, 1.0))
            continue
        fm = features.FeatureMap()
        for j in range(len(im.points2D)):
            if rng.random() < 0.9:
                fm.patches[j] = features.FeaturePatch(np.zeros((2, 2, 8), np.float16) + j, (j, k), (1.0, 1.0))
        fmaps[im.name] = fm
    fv = B.FeatureView(features.FeatureSet(fmaps), rec)
    a, b = B._SceneDump(rec, fv, use_compiled=True), B._SceneDump(rec, fv, use_compiled=False)
    assert a.compiled and not b.compiled
    for name in ("p2d_ptr", "p2d_point3D", "track_ptr", "track_image", "track_p2d", "has_patch", "image_camera", "cam_model"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert getattr(a, name).dtype == getattr(b, name).dtype, name
    # patch lookup on the observations that do have one
    at = np.flatnonzero((a.p2d_point3D >= 0) & (a.has_patch == 1))
    oi = (np.searchsorted(a.p2d_ptr, at, side="right") - 1).astype(np.int32)
    oj = (at - a.p2d_ptr[oi]).astype(np.int32)
    keep = oi != 4
    pa, pb = a.patches_of(oi[keep], oj[keep]), b.patches_of(oi[keep], oj[keep])
    assert len(pa) == keep.sum() and all(x is y for x, y in zip(pa, pb))
    with pytest.raises(KeyError):
        a.patches_of(np.array([4], np.int32), np.array([0], np.int32))
    # and the whole flat problem through both
    setup = B.default_problem_setup(rec)
    fa = B._FlatBA(rec, setup, fv, {}, extractor=True, scene=a)
    fb = B._FlatBA(rec, setup, fv, {}, extractor=True, scene=b)
    assert np.array_equal(fa.obs_image, fb.obs_image) and np.array_equal(fa.obs_point, fb.obs_point) and fa.obs_keys == fb.obs_keys
    assert all(x is y for x, y in zip(fa.patches, fb.patches))


def test_compiled_patch_gather_equals_the_python_walk():
    """_pxr_host.gather_patches: slots per DISTINCT patch object (a dense map is shared by all its keypoints), buffer addresses,
    corners, scales -- what features.to_arena hands to pxr_arena_upload_gather; mixed layouts are refused."""
    import pytest
    from pixsfm_amd.api import features
    host = features._host_module()
    if host is None:
        pytest.skip("_pxr_host was not built")
    rng = np.random.default_rng(1)
    base = [features.FeaturePatch(rng.normal(size=(4, 4, 8)).astype(np.float16), (3 * k, -k), (1.0 + k, 0.5)) for k in range(7)]
    order = rng.integers(0, 7, 40)
    plist = [base[k] for k in order]
    index, uniq, ptrs, corners, scales = host.gather_patches(plist)
    first = {}
    for k in order.tolist():
        first.setdefault(k, len(first))
    assert index.tolist() == [first[k] for k in order.tolist()] and index.dtype == np.int64
    want = [base[k] for k in sorted(first, key=first.get)]
    assert len(uniq) == len(want) and all(a is b for a, b in zip(uniq, want))
    assert ptrs.tolist() == [p.data.ctypes.data for p in want] and ptrs.dtype == np.uint64
    assert np.array_equal(corners, np.array([p.corner for p in want], np.int32)) and corners.dtype == np.int32
    assert np.array_equal(scales, np.array([p.scale for p in want], np.float64))
    with pytest.raises(ValueError):
        host.gather_patches(plist + [features.FeaturePatch(np.zeros((4, 4, 8), np.float32), (0, 0), (1.0, 1.0))])
    with pytest.raises(ValueError):
        host.gather_patches(plist + [features.FeaturePatch(np.zeros((2, 4, 8), np.float16), (0, 0), (1.0, 1.0))])
    import copy
    import pickle
    for clone in (pickle.loads(pickle.dumps(base[2])), copy.deepcopy(base[2])):     # a copy owns new memory: its address moved with it
        assert clone._meta[0] == clone.data.ctypes.data and clone._meta[1:] == base[2]._meta[1:]


@pytest.mark.parametrize("stride", [1, 7, 10 ** 9])
def test_scene_dump_point_ids_dense_and_sparse(stride):
    """_SceneDump maps point3D ids to indices with a lookup table when the ids are dense and with a binary search otherwise;
    a point2D that names a point3D the reconstruction lacks is a KeyError in both."""
    from pixsfm_amd.api import bundle_adjustment as B, features
    from pixsfm_amd.api.reconstruction import Camera, Image, Point2D, Point3D, Reconstruction

    def build(extra_id=None):
        rec = Reconstruction()
        rec.add_camera(Camera(1, "SIMPLE_PINHOLE", 100, 100, [50.0, 50, 50]))
        images = [Image(1 + i, "im%d.jpg" % i, 1, [1, 0, 0, 0], [0, 0, 0]) for i in range(3)]
        ids = [5 + stride * p for p in range(12)]
        for pid in ids:
            rec.add_point3D(pid, Point3D(np.zeros(3)))
        for n, pid in enumerate(ids):
            for im in (images[n % 3], images[(n + 1) % 3]):
                im.points2D.append(Point2D([0, 0], pid))
                rec.points3D[pid].track.add_element(im.image_id, len(im.points2D) - 1)
        if extra_id is not None:
            images[0].points2D.append(Point2D([0, 0], extra_id))
        for im in images:
            rec.add_image(im)
        fmaps = {im.name: features.FeatureMap() for im in images}
        return rec, B.FeatureView(features.FeatureSet(fmaps), rec), ids

    rec, fv, ids = build()
    for compiled in (True, False):
        sd = B._SceneDump(rec, fv, use_compiled=compiled)
        want = np.concatenate([[ids.index(q.point3D_id) for q in rec.images[i].points2D] for i in sorted(rec.images)])
        assert np.array_equal(sd.p2d_point3D, want)
    for missing in (5 + stride * 12, 6, 10 ** 12):
        if missing in ids:
            continue
        rec, fv, _ = build(extra_id=missing)
        with pytest.raises(KeyError):
            B._SceneDump(rec, fv)


def test_gc_is_paused_during_a_refinement_call_and_restored():
    """pixsfm_amd.api._timing.gc_paused: the collector is off inside the block, back to its previous state afterwards (also
    when the block raises, also when it was off to begin with)."""
    import gc
    from pixsfm_amd.api._timing import gc_paused
    assert gc.isenabled()
    with gc_paused():
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with gc_paused():
            raise RuntimeError("x")
    assert gc.isenabled()
    gc.disable()
    try:
        with gc_paused():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_compiled_slot_lookup_by_identity():
    """_pxr_host.slots_of: position of every item in `uniq` by object identity, -1 for strangers (the shared-arena lookup
    of features.to_arena)."""
    from pixsfm_amd.api import features
    host = features._host_module()
    if host is None or not hasattr(host, "slots_of"):
        pytest.skip("_pxr_host was not built")
    uniq = [object() for _ in range(1000)]
    rng = np.random.default_rng(0)
    pick = rng.integers(0, 1000, 5000)
    items = [uniq[k] for k in pick]
    assert np.array_equal(host.slots_of(uniq, items), pick)
    stranger = object()
    out = host.slots_of(uniq, [uniq[3], stranger, uniq[999]])
    assert out.tolist() == [3, -1, 999]
    assert host.slots_of([], [stranger]).tolist() == [-1] and len(host.slots_of(uniq, [])) == 0
    # equal but distinct objects are different patches
    a, b = (1, 2), tuple([1, 2])
    assert host.slots_of([a], [b]).tolist() == [-1]


def test_scene_dump_with_a_patch_container_that_is_not_a_dict():
    """The compiled has_patch walk takes plain dicts (PyDict_Next); any other mapping falls back to the numpy form and
    gives the same flags."""
    import collections
    from pixsfm_amd.api import bundle_adjustment as B, features
    from pixsfm_amd.api.reconstruction import Camera, Image, Point2D, Point3D, Reconstruction
    if B._host_module() is None:
        pytest.skip("_pxr_host was not built")
    rec = Reconstruction()
    rec.add_camera(Camera(1, "SIMPLE_PINHOLE", 100, 100, [50.0, 50, 50]))
    images = [Image(1 + i, "im%d.jpg" % i, 1, [1, 0, 0, 0], [0, 0, 0]) for i in range(3)]
    for p in range(9):
        rec.add_point3D(p + 1, Point3D(np.zeros(3)))
        for im in (images[p % 3], images[(p + 1) % 3]):
            im.points2D.append(Point2D([0, 0], p + 1))
            rec.points3D[p + 1].track.add_element(im.image_id, len(im.points2D) - 1)
    for im in images:
        rec.add_image(im)
    fmaps = {}
    for k, im in enumerate(images):
        fm = features.FeatureMap()
        patches = {j: features.FeaturePatch(np.zeros((2, 2, 8), np.float16), (j, k), (1.0, 1.0))
                   for j in range(len(im.points2D)) if (j + k) % 3}
        fm.patches = collections.UserDict(patches) if k == 1 else patches
        fmaps[im.name] = fm
    fv = B.FeatureView(features.FeatureSet(fmaps), rec)
    a, b = B._SceneDump(rec, fv, use_compiled=True), B._SceneDump(rec, fv, use_compiled=False)
    assert a.compiled and np.array_equal(a.has_patch, b.has_patch) and a.has_patch.dtype == b.has_patch.dtype
    assert 0 < a.has_patch.sum() < len(a.has_patch)
    # patches_of on the same containers: the compiled walk takes any mapping (__getitem__ for what is not a dict) and agrees
    # with the Python walk; a missing patch is the Python walk's KeyError on both
    have = np.flatnonzero(a.has_patch & (a.p2d_point3D >= 0))
    img_of = np.searchsorted(a.p2d_ptr, have, side="right") - 1
    j_of = have - a.p2d_ptr[img_of]
    pa = a.patches_of(img_of.astype(np.int64), j_of.astype(np.int64))
    pb = b.patches_of(img_of.astype(np.int64), j_of.astype(np.int64))
    assert len(pa) == len(pb) > 0 and all(x is y for x, y in zip(pa, pb))
    assert (img_of == 1).any()                        # the UserDict image took part
    lacking = np.flatnonzero((a.has_patch == 0) & (a.p2d_point3D >= 0))
    im_l = np.searchsorted(a.p2d_ptr, lacking, side="right") - 1
    for dump in (a, b):
        with pytest.raises(KeyError):
            dump.patches_of(im_l.astype(np.int64), (lacking - a.p2d_ptr[im_l]).astype(np.int64))


def test_flat_ba_does_not_reuse_a_scene_dump_of_another_feature_view():
    """_FlatBA takes a shared _SceneDump only for the very (reconstruction, feature_view) pair it was made from; the
    optimiser forgets the adjuster's dump and arena once run() returns."""
    from pixsfm_amd.api import bundle_adjustment as B, features
    from pixsfm_amd.api.reconstruction import Camera, Image, Point2D, Point3D, Reconstruction
    rec = Reconstruction()
    rec.add_camera(Camera(1, "SIMPLE_PINHOLE", 100, 100, [50.0, 50, 50]))
    images = [Image(1 + i, "im%d.jpg" % i, 1, [1, 0, 0, 0], [0, 0, 0]) for i in range(2)]
    for p in range(4):
        rec.add_point3D(p + 1, Point3D(np.array([0.0, 0, 5])))
        for im in images:
            im.points2D.append(Point2D([50, 50], p + 1))
            rec.points3D[p + 1].track.add_element(im.image_id, len(im.points2D) - 1)
    for im in images:
        rec.add_image(im)

    def fset(skip):
        out = {}
        for im in images:
            fm = features.FeatureMap()
            fm.patches = {j: features.FeaturePatch(np.zeros((4, 4, 8), np.float16), (48, 48), (1.0, 1.0))
                          for j in range(4) if j != skip}
            out[im.name] = fm
        return features.FeatureSet(out)
    fv_full, fv_holes = B.FeatureView(fset(None), rec), B.FeatureView(fset(2), rec)
    scene = B._SceneDump(rec, fv_full)
    setup = B.BundleAdjustmentSetup()
    setup.add_images(rec.reg_image_ids())
    same = B._FlatBA(rec, setup, fv_full, extractor=True, scene=scene)
    other = B._FlatBA(rec, setup, fv_holes, extractor=True, scene=scene)      # must re-dump: two observations lack a patch
    assert len(same.obs_image) == 8 and len(other.obs_image) == 6


def test_a_patch_replaced_in_place_drops_the_remembered_stack():
    """FeatureMap(patches, ids, corners, metadata) remembers the stacked array so that an upload can take the patches by address and
    stride (SharedArena.prefetch).  ANY write to the dict -- not only add_fpatch -- must drop that: a replaced patch of the same
    count would otherwise be uploaded from the stale slice of the array (the reference keeps no such shortcut: featuremap.cc:8-45
    builds plain FeaturePatch views)."""
    from pixsfm_amd.api.features import FeatureMap, FeaturePatch
    rng = np.random.default_rng(3)
    arr = rng.standard_normal((4, 4, 4, 8)).astype(np.float16)
    ids = [7, 9, 11, 13]
    corners = np.zeros((4, 2), np.int32)
    meta = {"is_sparse": True, "scale": [1.0, 1.0]}
    other = FeaturePatch(rng.standard_normal((4, 4, 8)).astype(np.float16), (0, 0), np.ones(2))
    edits = {
        "setitem": lambda fm: fm.patches.__setitem__(9, other),
        "fpatches": lambda fm: fm.fpatches.__setitem__(9, other),
        "update": lambda fm: fm.patches.update({9: other}),
        "pop": lambda fm: fm.patches.pop(13),
        "delitem": lambda fm: fm.patches.__delitem__(13),
        "setdefault": lambda fm: fm.patches.setdefault(15, other),
        "clear": lambda fm: fm.patches.clear(),
        "add_fpatch": lambda fm: fm.add_fpatch(9, other),
    }
    for name, edit in edits.items():
        fm = FeatureMap(arr, ids, corners, meta)
        assert fm.stacked() is not None and list(fm.patches) == ids, name
        edit(fm)
        assert fm.stacked() is None, name
    fm = FeatureMap(arr, ids, corners, meta)
    fm.patches = dict(fm.patches)                     # rebound: nothing watches the new dict
    assert fm.stacked() is None
    fm = FeatureMap(arr, ids, corners, meta)          # reading does not drop it
    _ = fm.fpatch(9), fm.keys(), fm.num_fpatches(), fm.has_fpatch(7), dict(fm.patches)
    assert fm.stacked() is not None


def test_shared_arena_slots_follow_the_stacks_layout():
    """SharedArena.slots: arena slot of an observation = first slot of its image's stack + the row of its keypoint id, computed with
    numpy from the layout the prefetch recorded -- the same answer as looking every FeaturePatch object up (to_arena / slots_of)."""
    from pixsfm_amd.api.features import FeatureMap, SharedArena
    rng = np.random.default_rng(5)
    names, maps, layout, base = ["a.jpg", "b.jpg", "c.jpg"], {}, {}, 0
    for name, n in zip(names, (5, 0, 7)):
        ids = rng.permutation(n + 3)[:n].astype(np.int64)              # keypoint ids in STACK order (not sorted, with gaps)
        if n:
            fm = FeatureMap(np.zeros((n, 2, 2, 4), np.float16), ids, np.zeros((n, 2), np.int32), {"is_sparse": True, "scale": [1.0, 1.0]})
            assert np.array_equal(fm.stacked()[3], ids)
            maps[name] = fm
            layout[name] = (base, ids)
            base += n
    class _Set:                                                         # what slots() asks of a feature set
        def has_fmap(self, name):
            return name in maps

        def fmap(self, name):
            return maps[name]
    fset = _Set()
    sa = SharedArena()
    assert sa.slots(names, [0], [0], feature_set=fset) is None          # no prefetched arena
    sa.arena, sa.layout = object(), layout
    sa._origin = (fset, {n: (m, int(m.stacked()[0].ctypes.data)) for n, m in maps.items()})       # what prefetch() records
    oi = np.array([0, 2, 2, 0, 2], np.int32)
    oj = np.array([layout["a.jpg"][1][3], layout["c.jpg"][1][0], layout["c.jpg"][1][6], layout["a.jpg"][1][0], layout["c.jpg"][1][2]])
    got = sa.slots(names, oi, oj, feature_set=fset)
    assert got.tolist() == [3, 5, 11, 0, 7]
    # the layout is only valid for the feature set that was prefetched, as it was then
    assert sa.slots(names, oi, oj) is None and sa.slots(names, oi, oj, feature_set=_Set()) is None
    # the same through the objects: position of the patch in the concatenation of the maps' dicts
    order = [id(p) for n in names if n in maps for p in maps[n].patches.values()]
    want = [order.index(id(maps[names[i]].patches[int(j)])) for i, j in zip(oi, oj)]
    assert got.tolist() == want
    missing = next(k for k in range(20) if k not in set(layout["a.jpg"][1].tolist()))
    assert sa.slots(names, [0], [missing], feature_set=fset) is None    # a keypoint the stack does not hold
    assert sa.slots(names, [1], [0], feature_set=fset) is None          # an image without patches
    assert sa.slots(names, [0], [-1], feature_set=fset) is None and sa.slots(names, [0], [10 ** 6], feature_set=fset) is None
    first = next(iter(maps["a.jpg"].patches))
    maps["a.jpg"].patches[first] = maps["a.jpg"].patches[first]         # a write drops the map's stack: the slots are stale
    assert sa.slots(names, oi, oj, feature_set=fset) is None
    sa.arena = None


def test_feature_map_from_a_stacked_array_pickles_and_copies():
    """FeatureMap.patches of a map built from ONE stacked array is a dict subclass that holds a weak reference to the
    map (a write drops the remembered stack).  It must pickle (FeaturePatch implements __setstate__ for that), and a deep copy
    must not keep pointing at the original: a write to the copy leaves the original's remembered stack alone."""
    import copy
    import pickle
    from pixsfm_amd.api import features
    rng = np.random.default_rng(0)
    arr = rng.normal(size=(5, 4, 4, 8)).astype(np.float16)
    fm = features.FeatureMap(arr, np.arange(5), rng.integers(0, 50, (5, 2)), {"is_sparse": True, "scale": np.array([0.5, 0.5]), "patch_size": 4})
    assert fm._stack is not None
    back = pickle.loads(pickle.dumps(fm))
    assert sorted(back.patches) == sorted(fm.patches) and back._stack is None
    assert all(np.array_equal(back.patches[k].data, fm.patches[k].data) for k in fm.patches)
    dup = copy.deepcopy(fm)
    assert dup._stack is None and type(dup.patches) is dict
    dup.patches[0] = dup.patches[1]
    assert fm._stack is not None                                  # the original still mirrors its array
    fm.patches[0] = fm.patches[1]
    assert fm._stack is None                                      # ... until IT is written to
