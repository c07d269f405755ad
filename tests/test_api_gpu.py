"""GPU: the pixsfm-compatible API end to end, written the way a pixsfm user would call it
(keypoint_adjustment/main.py:85-137, bundle_adjustment/main.py:65-154), checked against the
low-level engine / oracle on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ka_inputs(seed=3, n_tracks=12, track_len=5):
    """A pixsfm-style KA input: keypoints per image, a match graph, one patch per keypoint."""
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.api import features
    from pixsfm_amd.api.keypoint_adjustment import build_matching_graph
    prob = synthetic_ka.make_ka_problem(n_tracks=n_tracks, track_len=track_len, seed=seed, directed_both=False)
    # node (t, k) lives in image k with keypoint index t
    n = n_tracks * track_len
    img = np.arange(n) % track_len
    kid = np.arange(n) // track_len
    names = ["im%d" % k for k in range(track_len)]
    keypoints = {names[k]: prob["kp"][img == k].copy() for k in range(track_len)}
    pairs, matches, scores = [], [], []
    for a in range(track_len):
        for b in range(a + 1, track_len):
            sel = (img[prob["edge_src"]] == a) & (img[prob["edge_dst"]] == b)
            pairs.append((names[a], names[b]))
            matches.append(np.stack([kid[prob["edge_src"][sel]], kid[prob["edge_dst"][sel]]], 1))
            scores.append(prob["edge_w"][sel])
    graph = build_matching_graph(pairs, matches, scores)
    fmaps = {names[k]: features.FeatureMap.from_arrays(prob["patches"][img == k], kid[img == k],
                                                       prob["corners"][img == k], (1.0, 1.0)) for k in range(track_len)}
    return prob, keypoints, graph, features.FeatureManager([features.FeatureSet(fmaps)]), (img, kid, names)


def test_keypoint_adjuster_like_pixsfm(ctx):
    import pxo
    import pxo_ka
    from pixsfm_amd.api import KeypointAdjuster, base
    prob, keypoints, graph, fmanager, (img, kid, names) = _ka_inputs()
    adjuster = KeypointAdjuster.create({"strategy": "featuremetric"})
    out = adjuster.refine_multilevel(keypoints, fmanager, graph)
    summary = out["summary"][0]
    assert summary.final_cost < 0.05 * summary.initial_cost and summary.termination_type == "CONVERGENCE"
    # same problem through the oracle: graph order differs from the synthetic node order, so rebuild from the graph
    labels = base.compute_track_labels(graph)
    roots = base.compute_root_labels(graph, labels, base.compute_score_labels(graph, labels))
    from pixsfm_amd.api.keypoint_adjustment import build_edges, find_problem_labels
    src, dst, w = build_edges(graph, keypoints, labels, roots)
    node_of = [int(np.nonzero((img == n.image_id) & (kid == n.feature_idx))[0][0]) for n in graph.nodes]
    plabels, _ = find_problem_labels(labels, 50)
    oprob = dict(kp=prob["kp"][node_of], node_patch=np.arange(len(node_of), dtype=np.int64),
                 node_const=np.array(roots, np.uint8), node_problem=np.array(plabels, np.int32),
                 edge_src=np.array(src, np.int32), edge_dst=np.array(dst, np.int32), edge_w=np.array(w),
                 patches=prob["patches"][node_of], corners=prob["corners"][node_of], scales=prob["scales"][node_of])
    kpo, _ = pxo_ka.ka_solve(oprob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    got = np.array([keypoints[names[n.image_id]][n.feature_idx] for n in graph.nodes])
    assert np.abs(got - kpo).max() < 1e-6
    # an optimizer object is single-use (topological_keypoint_optimizer.h:75-77)
    from pixsfm_amd.api import FeatureMetricKeypointOptimizer, KeypointAdjustmentSetup
    opt = FeatureMetricKeypointOptimizer({}, KeypointAdjustmentSetup(), {})
    opt.run(keypoints, graph, labels, roots, fmanager.fset(0))
    with pytest.raises(ValueError):
        opt.run(keypoints, graph, labels, roots, fmanager.fset(0))


def test_topological_reference_adjuster(ctx):
    from pixsfm_amd.api import KeypointAdjuster
    prob, keypoints, graph, fmanager, _ = _ka_inputs(seed=8)
    before = {k: v.copy() for k, v in keypoints.items()}
    out = KeypointAdjuster.create({"strategy": "topological_reference"}).refine_multilevel(keypoints, fmanager, graph)
    s = out["summary"][0]
    assert s.final_cost < 0.1 * s.initial_cost
    assert any(np.abs(keypoints[k] - before[k]).max() > 0.1 for k in keypoints)


def test_bundle_adjuster_like_pixsfm(ctx):
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=6, n_points=70, obs_per_point=4, seed=17, noise=0.05)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
    conf = {"optimizer": {"solver": {"max_num_iterations": 8}}}
    out = BundleAdjuster.create(conf).refine_multilevel(rec, fmanager)
    summary, references = out["summary"][0], out["references"][0]
    # ceres::Solver::Summary::iterations: iteration 0 = the initial evaluation, then one record per LM iteration
    assert [it.iteration for it in summary.iterations] == list(range(summary.num_iterations + 1))
    assert summary.iterations[0].cost == summary.initial_cost and summary.iterations[-1].cost == summary.final_cost
    # (per-image feature noise + references taken at the perturbed projections leave a cost floor)
    assert len(references) == 70 and summary.final_cost < 0.5 * summary.initial_cost
    # the same thing through the low-level engine: references by the GPU extractor, default gauge
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=100)
    n_img = 6
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pose_const, tmask, np.full(n_img, 0b0110, np.uint16),
                 np.zeros(70, np.uint8), options=lm_options(max_iterations=8, use_inner_iterations=True))   # BA default
    q, t, k, X = ba.params()
    # observation order differs between the two flattenings (summation order, borderline inner-iteration stops)
    assert abs(s["final_cost"] - summary.final_cost) < 1e-4 * max(s["final_cost"], 1e-12)
    assert np.abs(np.array([rec.images[i + 1].qvec for i in range(n_img)]) - q).max() < 1e-4
    assert np.abs(np.array([rec.points3D[p + 1].xyz for p in range(70)]) - X).max() < 1e-4
    assert np.abs(rec.cameras[1].params - k[0, :4]).max() < 1e-4 * 1200
    # reference source is one of the point's own observations (references.h:29-72)
    for pid, ref in references.items():
        assert ref.source in [(e.image_id, e.point2D_idx) for e in rec.points3D[pid].track.elements]
        assert abs(np.linalg.norm(ref.descriptor) - 1) < 1e-12


def test_query_keypoint_adjuster_like_pixsfm(ctx):
    """localization QKA (localization/main.py:89-192): refine / stacked / feature-inlier gating
    against the oracle solving the same one-problem-per-query system."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.api import QueryKeypointAdjuster, features, find_feature_inliers
    n = 24
    base = synthetic_ka.make_ka_problem(n_tracks=n, track_len=2, seed=31, sigma=0.7)
    q = np.arange(0, 2 * n, 2)
    cfg = pxo.cfg()
    refs = []
    for nd in q + 1:           # the 3D point's reference = the descriptor of its other observation
        p = pxo.make_patch(base["patches"][nd], base["corners"][nd], base["scales"][nd])
        refs.append(pxo.ref2d_residual(p, cfg, base["true_xy"][nd], np.zeros(128))[0])
    fmap = features.FeatureMap.from_arrays(base["patches"][q], np.arange(n), base["corners"][q], (1.0, 1.0))
    kp0 = base["kp"][q].copy()
    # oracle: one problem, unary terms only
    oprob = dict(base)
    oprob.update(kp=kp0.copy(), node_patch=q.astype(np.int64), node_const=np.zeros(n, np.uint8),
                 node_problem=np.zeros(n, np.int32), edge_src=np.zeros(0, np.int32), edge_dst=np.zeros(0, np.int32),
                 edge_w=np.zeros(0), unary_node=np.arange(n, dtype=np.int32), unary_ref=np.array(refs), unary_w=None)
    kpo, sums = pxo_ka.ka_solve(oprob, cfg, pxo.loss("trivial"), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    adjuster = QueryKeypointAdjuster(ctx=ctx)
    kp = kp0.copy()
    adjuster.refine(kp, fmap, [np.asarray(r) for r in refs])
    assert np.abs(kp - kpo).max() < 1e-6
    assert adjuster.solver.last_summary["iterations"] == sums[0]["iterations"]
    assert np.median(np.linalg.norm(kp - base["true_xy"][q], axis=1)) < 0.05
    # Reference objects (descriptor / per-observation descriptors) are accepted like arrays
    kp2 = kp0.copy()
    adjuster.refine(kp2, fmap, [features.Reference(0, i, r) for i, r in enumerate(refs)])
    assert np.array_equal(kp2, kp)
    # stacked: duplicated correspondences of one 2D point collapse onto one keypoint with two terms
    idxs = list(range(n)) + [0, 1]
    kp3 = np.concatenate([kp0, kp0[:2]])
    stacked = QueryKeypointAdjuster({"stack_correspondences": True}, ctx=ctx)
    stacked.refine(kp3, fmap, [np.asarray(r) for r in refs] + [np.asarray(refs[0]), np.asarray(refs[1])], point2D_idxs=idxs)
    assert np.array_equal(kp3[-2:], kp3[:2]) and np.abs(kp3[:n] - kpo).max() < 1e-3
    # feature-inlier gating: |f - ref| at the initial keypoints, thresholded
    dist = []
    for i, nd in enumerate(q):
        p = pxo.make_patch(base["patches"][nd], base["corners"][nd], base["scales"][nd])
        dist.append(np.linalg.norm(pxo.ref2d_residual(p, cfg, kp0[i], refs[i])[0]))
    thr = float(np.median(dist))
    inl = find_feature_inliers(kp0, fmap, [np.asarray(r) for r in refs], adjuster.conf["interpolation"], thresh=thr, ctx=ctx)
    assert inl == [bool(d <= thr) for d in dist] and 0 < sum(inl) < n
    gated = QueryKeypointAdjuster({"feature_inlier_thresh": thr}, ctx=ctx)
    kp4 = kp0.copy()
    gated.refine(kp4, fmap, [np.asarray(r) for r in refs])
    out = ~np.array(inl)
    assert np.array_equal(kp4[out], kp0[out]) and np.abs(kp4[~out] - kp0[~out]).max() > 1e-3


@pytest.mark.parametrize("refine_focal", [False, True])
def test_query_bundle_adjuster_like_pixsfm(ctx, refine_focal):
    """localization QBA (localization/main.py:194-258): one image, constant 3D points, pose on the
    quaternion manifold (+ focal length when asked) -- vs the oracle LM on the same flat problem."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import QueryBundleAdjuster, features
    from pixsfm_amd.api.reconstruction import Camera
    full = synthetic.make_ba_problem(n_cams=4, n_points=120, obs_per_point=3, seed=12, model=2, rot_deg=0.3, trans=0.02)
    sel = np.nonzero(full["obs_image"] == 0)[0]
    pts = full["obs_point"][sel]
    n = len(sel)
    assert n > 40
    points3D = [full["gt_xyz"][p].copy() for p in pts]
    refs = [full["refs"][p].copy() for p in pts]
    refs[3] = [refs[3], refs[3] + 1e-3]                          # a correspondence with two references
    fmap = features.FeatureMap.from_arrays(full["patches"][sel], np.arange(n), full["corners"][sel], (1.0, 1.0))
    cam = Camera(1, 2, 1000, 1000, full["cam_params"][full["image_camera"][0], :4].copy())
    qvec, tvec = full["qvec"][0].copy(), full["tvec"][0].copy()
    inliers = [True] * n
    inliers[5] = False
    opt = {"refine_focal_length": refine_focal}
    adj = QueryBundleAdjuster({"optimizer": opt}, ctx=ctx)
    assert adj.refine(qvec, tvec, cam, points3D, fmap, refs, inliers=inliers)
    # oracle on the flat problem the reference would build
    rows = [i for i in range(n) if inliers[i]]
    o_patch, o_xyz, o_refs = [], [], []
    for i in rows:
        for d in (refs[i] if isinstance(refs[i], list) else [refs[i]]):
            o_patch.append(sel[i]); o_xyz.append(points3D[i]); o_refs.append(d)
    m = len(o_patch)
    flat = dict(obs_image=np.zeros(m, np.int32), obs_point=np.arange(m, dtype=np.int32),
                obs_patch=np.arange(m, dtype=np.int64), image_camera=np.zeros(1, np.int32),
                qvec=full["qvec"][:1].copy(), tvec=full["tvec"][:1].copy(), cam_model=np.array([2], np.int32),
                cam_params=full["cam_params"][full["image_camera"][0]][None].copy(), xyz=np.array(o_xyz),
                refs=np.array(o_refs), patches=np.ascontiguousarray(full["patches"][o_patch]),
                corners=full["corners"][o_patch], scales=full["scales"][o_patch])
    mask = 0b1110 if refine_focal else 0b1111
    so, qo, to, co, _ = pxo.ba_solve(flat, pxo.cfg(), pxo.loss("cauchy", 0.25), [0], [0], [mask], np.ones(m, np.uint8),
                                     pxo.lm_options())
    s = adj.solver.last_summary
    assert s["num_camera_unknowns"] == (7 if refine_focal else 6) and s["num_point_unknowns"] == 0
    assert abs(s["initial_cost"] - so["initial_cost"]) < 1e-10 * so["initial_cost"]
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-6 * so["initial_cost"]
    assert np.abs(qvec - qo[0]).max() < 1e-6 and np.abs(tvec - to[0]).max() < 1e-6
    assert np.abs(cam.params - co[0, :4]).max() < 1e-4 * 1200
    if not refine_focal:
        assert np.array_equal(cam.params, full["cam_params"][full["image_camera"][0], :4])
    # the refined pose is closer to the ground truth than the perturbed start
    assert np.linalg.norm(tvec - full["gt_tvec"][0]) < 0.2 * np.linalg.norm(full["tvec"][0] - full["gt_tvec"][0])


def test_reference_extractor_keep_observations_and_find_nearest(ctx):
    """ReferenceExtractor(keep_observations=True) -> Reference.observations; find_nearest_references returns, for
    a query keypoint placed on one of a point's observations, that observation's descriptor."""
    from pixsfm_amd.api import ReferenceExtractor, features, find_nearest_references
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=4, n_points=40, obs_per_point=3, seed=41)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fset = features.FeatureSet(fmaps)
    refs = ReferenceExtractor({"keep_observations": True, "iters": 10}, None, ctx=ctx).run(
        [0] * (max(rec.point3D_ids()) + 1), rec, fset)
    assert len(refs) == 40
    for pid, r in refs.items():
        assert r.has_observations() and len(r.observations) == rec.points3D[pid].track.length()
        d = np.array([np.linalg.norm(o - r.descriptor) for o in r.observations])
        assert d.min() < 1e-12                              # the reference IS one of the observations (closest to the mean)
        # ReferenceData (reference_extractor.h:256-265): the visible track and each observation's squared distance to the robust
        # mean; the chosen observation is the one with the smallest cost, and it is the source
        assert r.track == [(e.image_id, e.point2D_idx) for e in rec.points3D[pid].track.elements] and len(r.costs) == len(r.track)
        assert r.track[int(np.argmin(r.costs))] == r.source and int(np.argmin(r.costs)) == int(np.argmin(d))
    # query = observations of image 0, keypoints at their detections
    im = rec.images[min(rec.images)]
    idxs = [k for k, p in enumerate(im.points2D) if p.has_point3D()]
    kps = np.array([im.points2D[k].xy for k in idxs])
    pids = [im.points2D[k].point3D_id for k in idxs]
    near = find_nearest_references(fset.fmap(im.name), refs, kps, pids, None, patch_idxs=idxs, ctx=ctx)
    assert len(near) == len(idxs) and near[0].shape == (1, 128)
    hits = 0
    for k, pid, nd in zip(idxs, pids, near):
        d = [np.linalg.norm(o - nd) for o in refs[pid].observations]
        assert min(d) == 0.0                                # the winner is one of the candidates
        hits += 1
    assert hits == len(idxs)


def test_run_subset_only_touches_its_nodes(ctx):
    """FeatureMetricKeypointOptimizer.run_subset (featuremetric_keypoint_optimizer.h:116-137): the tracks in the
    subset end where a full run puts them (tracks are independent), every other keypoint is untouched."""
    from pixsfm_amd.api import FeatureMetricKeypointOptimizer, KeypointAdjustmentSetup, base
    prob, keypoints, graph, fmanager, (img, kid, names) = _ka_inputs(seed=5, n_tracks=10, track_len=4)
    track_labels = base.compute_track_labels(graph)
    score = base.compute_score_labels(graph, track_labels)
    roots = base.compute_root_labels(graph, track_labels, score)

    def fresh():
        setup = KeypointAdjustmentSetup()
        setup.set_masked_nodes_constant(graph, roots)
        return FeatureMetricKeypointOptimizer({"solver": {"parameter_tolerance": 1e-5}}, setup, None, ctx=ctx)

    kp_full = {k: v.copy() for k, v in keypoints.items()}
    assert fresh().run(kp_full, graph, track_labels, roots, fmanager.fset(0))
    chosen = set(sorted(set(track_labels))[:4])
    subset = {i for i, t in enumerate(track_labels) if t in chosen}
    kp_sub = {k: v.copy() for k, v in keypoints.items()}
    summary = fresh().run_subset(subset, kp_sub, graph, track_labels, roots, fmanager.fset(0))
    assert summary is not None and summary.final_cost < summary.initial_cost
    for i, nd in enumerate(graph.nodes):
        nm = graph.image_id_to_name[nd.image_id]
        if i in subset:
            assert np.abs(kp_sub[nm][nd.feature_idx] - kp_full[nm][nd.feature_idx]).max() < 1e-9
        else:
            assert np.array_equal(kp_sub[nm][nd.feature_idx], keypoints[nm][nd.feature_idx])


def test_run_subset_not_closed_over_tracks(ctx):
    """A subset that cuts through tracks: the reference enumerates the out-matches of nodes_in_problem only, the
    destination of such a match is a parameter block even when it lies outside the subset, and ParameterizeKeypoints
    visits nodes_in_problem only (topological_keypoint_optimizer.h:108-113, keypoint_optimizer.h:117) -- so the outside
    keypoint is refined too, without box bounds and without the constant flag.  Against the oracle on the same blocks."""
    import pxo
    import pxo_ka
    from pixsfm_amd.api import FeatureMetricKeypointOptimizer, KeypointAdjustmentSetup, base
    from pixsfm_amd.api.keypoint_adjustment import build_edges
    prob, keypoints, graph, fmanager, (img, kid, names) = _ka_inputs(seed=9, n_tracks=8, track_len=5)
    track_labels = base.compute_track_labels(graph)
    roots = base.compute_root_labels(graph, track_labels, base.compute_score_labels(graph, track_labels))
    n = len(graph.nodes)
    subset = sorted(i for i in range(n) if graph.nodes[i].image_id in (0, 1))     # two of the five images
    setup = KeypointAdjustmentSetup()
    setup.set_masked_nodes_constant(graph, roots)
    kp_sub = {k: v.copy() for k, v in keypoints.items()}
    summary = FeatureMetricKeypointOptimizer({"solver": {"parameter_tolerance": 1e-5}}, setup, None, ctx=ctx).run_subset(
        set(subset), kp_sub, graph, track_labels, roots, fmanager.fset(0))
    assert summary.final_cost < summary.initial_cost
    src, dst, w = build_edges(graph, keypoints, track_labels, roots, subset)
    assert len(src) and set(src) <= set(subset) and not set(dst) <= set(subset)
    touched = set(src) | set(dst)
    node_of = [int(np.nonzero((img == nd.image_id) & (kid == nd.feature_idx))[0][0]) for nd in graph.nodes]
    node_const = np.array(roots, np.uint8)
    node_const[[i for i in touched if i not in set(subset)]] = 2
    labels = np.full(n, -1, np.int32); labels[sorted(touched | set(subset))] = 0
    oprob = dict(kp=prob["kp"][node_of], node_patch=np.arange(n, dtype=np.int64), node_const=node_const,
                 node_problem=np.where(labels < 0, 1, 0).astype(np.int32),        # the oracle wants labels >= 0: untouched nodes = an empty problem 1
                 edge_src=np.array(src, np.int32), edge_dst=np.array(dst, np.int32), edge_w=np.array(w),
                 patches=prob["patches"][node_of], corners=prob["corners"][node_of], scales=prob["scales"][node_of])
    kpo, _ = pxo_ka.ka_solve(oprob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    got = np.array([kp_sub[names[nd.image_id]][nd.feature_idx] for nd in graph.nodes])
    before = np.array([keypoints[names[nd.image_id]][nd.feature_idx] for nd in graph.nodes])
    assert np.abs(got - kpo).max() < 1e-6                                         # vs oracle (parity unpinned vs Ceres)
    outside_moved = [i for i in touched if i not in set(subset) and not np.array_equal(got[i], before[i])]
    assert outside_moved, "destination keypoints outside the subset are parameters of the problem"
    for i in range(n):
        if i not in touched:
            assert np.array_equal(got[i], before[i])


def test_bundle_optimizer_set_up_solve_reset(ctx):
    """FeatureReferenceBundleOptimizer.set_up / .problem / .solve_problem / .reset (bindings.cc:36-51): the two-step
    form gives what run() gives; reset() allows a second set_up; a second run() without reset raises."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import FeatureReferenceBundleOptimizer, ReferenceExtractor, default_problem_setup, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat

    def inputs():
        prob = synthetic.make_ba_problem(n_cams=5, n_points=50, obs_per_point=3, seed=29)
        rec, patch_of = reconstruction_from_flat(prob)
        fmaps = {}
        for (image_id, p2d), pi in patch_of.items():
            fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
            fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
        return rec, features.FeatureSet(fmaps)

    opts = {"solver": {"max_num_iterations": 5}}
    rec1, fset1 = inputs()
    refs = ReferenceExtractor({"iters": 10}, None, ctx=ctx).run([0] * 51, rec1, fset1)
    o1 = FeatureReferenceBundleOptimizer(opts, default_problem_setup(rec1), None, ctx=ctx)
    assert o1.run(rec1, fset1, refs)
    with pytest.raises(ValueError):
        o1.run(rec1, fset1, refs)
    rec2, fset2 = inputs()
    o2 = FeatureReferenceBundleOptimizer(opts, default_problem_setup(rec2), None, ctx=ctx)
    o2.set_up(rec2, {"name": "cauchy", "params": [0.25]}, fset2, refs)
    cost0 = o2.problem.eval(o2.interpolation.to_engine())[0].download()[:, 0].sum()
    assert cost0 > 0 and o2.solve_problem(rec2)
    assert abs(o2.summary().final_cost - o1.summary().final_cost) < 1e-12 * max(o1.summary().final_cost, 1e-12)
    for i in rec1.images:
        assert np.abs(rec1.images[i].qvec - rec2.images[i].qvec).max() < 1e-9   # atomics: summation order varies
    o2.reset()
    assert o2.problem is None
    o2.set_up(rec2, None, fset2, refs)                     # usable again after reset()
    assert o2.solve_problem(rec2)


def test_query_keypoint_adjuster_batch_equals_sequential(ctx):
    """refine_batch: many queries in one launch, each its own sub-problem -> the same keypoints as refine() per query."""
    import pxo
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.api import QueryKeypointAdjuster, features
    queries_a, queries_b = [], []
    for qi, n in enumerate((9, 1, 17, 30)):
        base = synthetic_ka.make_ka_problem(n_tracks=n, track_len=2, seed=60 + qi, sigma=0.6)
        q = np.arange(0, 2 * n, 2)
        refs = []
        for nd in q + 1:
            p = pxo.make_patch(base["patches"][nd], base["corners"][nd], base["scales"][nd])
            refs.append(pxo.ref2d_residual(p, pxo.cfg(), base["true_xy"][nd], np.zeros(128))[0])
        fmap = features.FeatureMap.from_arrays(base["patches"][q], np.arange(n), base["corners"][q], (1.0, 1.0))
        queries_a.append((base["kp"][q].copy(), fmap, [np.asarray(r) for r in refs]))
        queries_b.append((base["kp"][q].copy(), fmap, [np.asarray(r) for r in refs]))
    seq = QueryKeypointAdjuster(ctx=ctx)
    its = []
    for pts, fmap, refs in queries_a:
        seq.refine(pts, fmap, refs)
        its.append(seq.solver.last_summary["iterations"])
    bat = QueryKeypointAdjuster(ctx=ctx)
    assert bat.refine_batch(queries_b) == [True] * 4
    for (pa, _, _), (pb, _, _), it, s in zip(queries_a, queries_b, its, bat.solver.last_summaries):
        assert np.abs(pa - pb).max() < 1e-12 and s["iterations"] == it


def test_device_resident_feature_flow_equals_host_flow(ctx):
    """SURVEY 8f row 2 end to end: dense maps stay on the device, tensor_to_arena writes the patches straight into
    an arena, FeatureMaps hold ArenaPatch handles and KeypointAdjuster indexes that arena in place -- and gives
    exactly what the host flow (numpy FeaturePatch objects, stacked and uploaded) gives on the same patches."""
    import torch
    from pixsfm_amd.api import KeypointAdjuster, features
    from pixsfm_amd.api.keypoint_adjustment import build_matching_graph
    from pixsfm_amd.engine import PatchArena
    n_img, n_kp, C, h, w = 4, 30, 128, 96, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32),
                            torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
    freq = torch.rand((C, 2), generator=g, device="cuda") * 0.25
    phase = torch.rand((C,), generator=g, device="cuda") * 6.28
    rng = np.random.default_rng(2)
    shifts = rng.uniform(-6, 6, (n_img, 2))
    pts = rng.uniform(30, 90, (n_kp, 2)) * [1.0, 0.7]
    names = ["im%d" % i for i in range(n_img)]
    arena = PatchArena(ctx, n_img * n_kp, 16, 16, C, np.float16)
    keypoints, fmaps_dev = {}, {}
    for i in range(n_img):
        fmap = torch.cos(freq[:, 0, None, None] * (xs + float(shifts[i, 0])) + freq[:, 1, None, None] * (ys + float(shifts[i, 1]))
                         + phase[:, None, None]).contiguous()                    # (C, h, w), the "dense features" of image i
        kps = pts - shifts[i] + rng.normal(0, 0.7, pts.shape)
        keypoints[names[i]] = kps.copy()
        assert features.tensor_to_arena(arena, i * n_kp, fmap, (float(w), float(h)), kps) == n_kp
        fmaps_dev[names[i]] = features.fmap_from_arena(arena, i * n_kp, range(n_kp))
    pairs = [(names[a], names[b]) for a in range(n_img) for b in range(a + 1, n_img)]
    matches = [np.stack([np.arange(n_kp), np.arange(n_kp)], 1) for _ in pairs]
    graph = build_matching_graph(pairs, matches, [rng.uniform(0.5, 1, n_kp) for _ in pairs])
    # host flow on the very same texels
    patches, corners, scales = arena.download()
    fmaps_host = {names[i]: features.FeatureMap.from_arrays(patches[i * n_kp:(i + 1) * n_kp], np.arange(n_kp),
                                                              corners[i * n_kp:(i + 1) * n_kp], tuple(scales[i * n_kp]))
                  for i in range(n_img)}
    out = {}
    for tag, fmaps in (("device", fmaps_dev), ("host", fmaps_host)):
        kp = {k: v.copy() for k, v in keypoints.items()}
        KeypointAdjuster.create({"strategy": "featuremetric"}).refine_multilevel(
            kp, features.FeatureManager([features.FeatureSet(fmaps)]), graph)
        out[tag] = kp
    moved = 0.0
    for nm in names:
        assert np.abs(out["device"][nm] - out["host"][nm]).max() < 1e-12
        moved = max(moved, np.abs(out["device"][nm] - keypoints[nm]).max())
    assert moved > 0.1
    # the arena is still alive and untouched by the adjuster (it does not own it)
    assert np.array_equal(arena.download(0, 4)[0], patches[:4])


def test_patch_interpolator_matches_oracle(ctx):
    """features.PatchInterpolator (patch_interpolator.h:86-135): descriptors and keypoint Jacobians vs the oracle."""
    import pxo
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.api import features
    base = synthetic_ka.make_ka_problem(n_tracks=5, track_len=3, seed=4, scale=(0.5, 0.25))
    fps = [features.FeaturePatch(base["patches"][i], base["corners"][i], base["scales"][i]) for i in range(15)]
    pi = features.PatchInterpolator(ctx=ctx)
    desc, J = pi.interpolate_many(fps, base["kp"], jacobian=True)
    cfg = pxo.cfg()
    for i in (0, 6, 14):
        p = pxo.make_patch(base["patches"][i], base["corners"][i], base["scales"][i])
        r, Jo = pxo.ref2d_residual(p, cfg, base["kp"][i], np.zeros(128))
        assert np.abs(desc[i] - r).max() < 1e-14 and np.abs(J[i] - Jo).max() < 1e-12 * max(1.0, np.abs(Jo).max())
    one = pi.interpolate_nodes(fps[3], base["kp"][3])
    assert one.shape == (1, 128) and np.array_equal(one[0], desc[3]) and abs(np.linalg.norm(one) - 1) < 1e-12


def test_dense_feature_maps_equal_sparse_patches(ctx):
    """Dense mode (FeatureMap.is_sparse = False, one kDenseId patch per image, featuremap.h:104-118): KA on dense maps
    gives what KA gives on 16x16 patches cropped from the same maps, as long as the stencils stay inside the crops."""
    from pixsfm_amd.api import KeypointAdjuster, features
    from pixsfm_amd.api.keypoint_adjustment import build_matching_graph
    rng = np.random.default_rng(5)
    n_img, n_kp, C, h, w = 3, 12, 128, 40, 48
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    freq, phase = rng.uniform(0, 0.3, (C, 2)), rng.uniform(0, 6.28, C)
    shifts = rng.uniform(-3, 3, (n_img, 2))
    pts = rng.uniform(14, 26, (n_kp, 2))
    names = ["im%d" % i for i in range(n_img)]
    keypoints, dense, sparse = {}, {}, {}
    for i in range(n_img):
        m = np.cos(freq[:, 0] * (xs[..., None] + shifts[i, 0]) + freq[:, 1] * (ys[..., None] + shifts[i, 1]) + phase)
        m = (m / np.linalg.norm(m, axis=-1, keepdims=True)).astype(np.float16)          # (h, w, C)
        kps = pts - shifts[i] + rng.normal(0, 0.4, pts.shape)
        keypoints[names[i]] = kps
        dense[names[i]] = features.FeatureMap.dense(m, (1.0, 1.0))
        corners = np.clip((kps - 8).astype(np.int32), 0, [w - 17, h - 17])
        sparse[names[i]] = features.FeatureMap.from_arrays(
            np.stack([m[c[1]:c[1] + 16, c[0]:c[0] + 16] for c in corners]), np.arange(n_kp), corners, (1.0, 1.0))
    pairs = [(names[a], names[b]) for a in range(n_img) for b in range(a + 1, n_img)]
    graph = build_matching_graph(pairs, [np.stack([np.arange(n_kp)] * 2, 1)] * len(pairs), [np.ones(n_kp)] * len(pairs))
    out = {}
    for tag, fm in (("dense", dense), ("sparse", sparse)):
        kp = {k: v.copy() for k, v in keypoints.items()}
        conf = {"strategy": "featuremetric", "optimizer": {"bound": 2.0}}
        KeypointAdjuster.create(conf).refine_multilevel(kp, features.FeatureManager([features.FeatureSet(fm)]), graph)
        out[tag] = kp
    assert dense[names[0]].has_fpatch(7) and dense[names[0]].fpatch(3) is dense[names[0]].fpatch(9)
    for nm in names:
        assert np.abs(out["dense"][nm] - out["sparse"][nm]).max() < 1e-9
    assert max(np.abs(out["dense"][nm] - keypoints[nm]).max() for nm in names) > 0.05     # (the root image stays put)


def test_feature_cache_to_device_flow_equals_host_flow(ctx, tmp_path):
    """a pixsfm "chunked" feature cache (extract.py:98-127) read by the native reader straight into ONE
    device arena per level (load_features_from_cache(device=True)) drives the bundle adjuster exactly like host patches."""
    import h5_writer
    if not h5_writer.available():
        pytest.skip("the image's libhdf5 is missing")
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    prob = synthetic.make_ba_problem(n_cams=5, n_points=50, obs_per_point=3, seed=29, noise=0.03)
    rec_h, patch_of = reconstruction_from_flat(prob)
    rec_d, _ = reconstruction_from_flat(prob)
    level, fmaps = {}, {}
    for (image_id, p2d), pi in patch_of.items():
        name = rec_h.images[image_id].name
        fm = level.setdefault(name, dict(keypoint_ids=[], patches=[], corners=[], scales=[], metadata={"is_sparse": True, "patch_size": 16}))
        fm["keypoint_ids"].append(p2d); fm["patches"].append(prob["patches"][pi])
        fm["corners"].append(prob["corners"][pi]); fm["scales"].append(prob["scales"][pi])
        fmaps.setdefault(name, features.FeatureMap()).patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    for fm in level.values():
        fm["patches"], fm["corners"] = np.stack(fm["patches"]), np.stack(fm["corners"])
    path = tmp_path / "features.h5"
    h5_writer.write_cache(path, [level])
    fmgr_d = features.load_features_from_cache(path, device=True, ctx=ctx)
    assert fmgr_d.fset(0).arena.n == len(prob["obs_image"])
    conf = {"optimizer": {"solver": {"max_num_iterations": 5}}}
    out_d = BundleAdjuster.create(conf).refine_multilevel(rec_d, fmgr_d)
    out_h = BundleAdjuster.create(conf).refine_multilevel(rec_h, features.FeatureManager([features.FeatureSet(fmaps)]))
    assert out_d["summary"][0].final_cost == pytest.approx(out_h["summary"][0].final_cost, rel=1e-9)
    for i in rec_h.images:
        assert np.abs(rec_h.images[i].qvec - rec_d.images[i].qvec).max() < 1e-9
    for p in rec_h.points3D:
        assert np.abs(rec_h.points3D[p].xyz - rec_d.points3D[p].xyz).max() < 1e-9


@pytest.mark.parametrize("channels", [3, 1])
def test_bundle_adjuster_on_image_intensity_features(ctx, channels):
    """BundleAdjuster (strategy feature_reference) on 3- / 1-channel maps without L2 normalisation, the "image" dense features
    of pixsfm (features/models/image.py; FeatureReferenceBundleOptimizer (3, 1), (1, 1)): references extracted and the BA
    solved through the same API objects; equal to the low-level engine on the same scene."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=6, n_points=50, obs_per_point=4, seed=31 + channels, noise=0.02, channels=channels)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
    conf = {"optimizer": {"solver": {"max_num_iterations": 6}}, "interpolation": {"l2_normalize": False}}
    out = BundleAdjuster.create(conf).refine_multilevel(rec, fmanager)
    summary, references = out["summary"][0], out["references"][0]
    assert len(references) == 50 and all(r.descriptor.shape == (1, channels) for r in references.values())
    assert summary.final_cost < summary.initial_cost
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cfg = interp_cfg(l2_normalize=False)
    ba.compute_references(cfg, make_loss("cauchy", [0.25]), iters=100)
    pose_const = np.zeros(6, np.uint8); pose_const[0] = 1
    tmask = np.zeros(6, np.uint8); tmask[1] = 1
    s = ba.solve(cfg, make_loss("cauchy", [0.25]), pose_const, tmask, np.full(6, 0b0110, np.uint16), np.zeros(50, np.uint8),
                 options=lm_options(max_iterations=6, use_inner_iterations=True))
    q, t, k, X = ba.params()
    assert abs(s["final_cost"] - summary.final_cost) < 1e-4 * max(s["final_cost"], 1e-12)
    assert np.abs(np.array([rec.points3D[p + 1].xyz for p in range(50)]) - X).max() < 1e-4


def test_query_bundle_adjuster_on_single_channel_features(ctx):
    """QueryBundleOptimizer's (1, 1) case (query_bundle_optimizer.h:33-34): a grayscale map, no L2 normalisation."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import QueryBundleAdjuster, features
    from pixsfm_amd.api.reconstruction import Camera
    full = synthetic.make_ba_problem(n_cams=4, n_points=90, obs_per_point=3, seed=5, model=2, rot_deg=0.2, trans=0.01, channels=1)
    sel = np.nonzero(full["obs_image"] == 0)[0]
    pts, n = full["obs_point"][sel], len(sel)
    points3D = [full["gt_xyz"][p].copy() for p in pts]
    # references: the unnormalised intensities at the true projections
    cfg = pxo.cfg(l2_normalize=False)
    refs = []
    for i, p in zip(sel, pts):
        patch = pxo.make_patch(full["patches"][i], full["corners"][i], full["scales"][i])
        refs.append(pxo.ba_residual(patch, cfg, 2, full["gt_qvec"][0], full["gt_tvec"][0], full["gt_xyz"][p],
                                    full["cam_params"][full["image_camera"][0]][:4], None, jac=False)[0])
    fmap = features.FeatureMap.from_arrays(full["patches"][sel], np.arange(n), full["corners"][sel], (1.0, 1.0))
    cam = Camera(1, 2, 1000, 1000, full["cam_params"][full["image_camera"][0], :4].copy())
    qvec, tvec = full["qvec"][0].copy(), full["tvec"][0].copy()
    adj = QueryBundleAdjuster({"interpolation": {"l2_normalize": False}}, ctx=ctx)
    assert adj.refine(qvec, tvec, cam, points3D, fmap, refs)
    flat = dict(obs_image=np.zeros(n, np.int32), obs_point=np.arange(n, dtype=np.int32), obs_patch=np.arange(n, dtype=np.int64),
                image_camera=np.zeros(1, np.int32), qvec=full["qvec"][:1].copy(), tvec=full["tvec"][:1].copy(),
                cam_model=np.array([2], np.int32), cam_params=full["cam_params"][full["image_camera"][0]][None].copy(),
                xyz=np.array(points3D), refs=np.array(refs), patches=np.ascontiguousarray(full["patches"][sel]),
                corners=full["corners"][sel], scales=full["scales"][sel])
    so, qo, to, co, _ = pxo.ba_solve(flat, cfg, pxo.loss("cauchy", 0.25), [0], [0], [0b1111], np.ones(n, np.uint8), pxo.lm_options())
    s = adj.solver.last_summary
    assert abs(s["initial_cost"] - so["initial_cost"]) < 1e-10 * so["initial_cost"]
    assert abs(s["final_cost"] - so["final_cost"]) < 1e-6 * so["initial_cost"]
    assert np.abs(qvec - qo[0]).max() < 1e-6 and np.abs(tvec - to[0]).max() < 1e-6
    assert s["final_cost"] < 0.5 * s["initial_cost"]


def test_iteration_callbacks_of_the_bundle_adjustment(ctx):
    """`solver.callbacks` (base/src/callbacks.h; pyceres IterationCallback objects in the reference): called after the initial
    evaluation and after every LM iteration with the iteration summary; SOLVER_ABORT (1) / SOLVER_TERMINATE_SUCCESSFULLY (2)
    end the solve like in Ceres."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=5, n_points=40, obs_per_point=4, seed=3)
    n_img = 5
    gauge = (np.r_[1, np.zeros(n_img - 1)].astype(np.uint8), np.r_[0, 1, np.zeros(n_img - 2)].astype(np.uint8), np.full(n_img, 0b0110, np.uint16), np.zeros(40, np.uint8))
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])

    def run(callbacks, max_it=6):
        ba = BAProblem(ctx, arena, prob)
        ctx.set_iteration_callbacks(callbacks)
        try:
            return ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=max_it))
        finally:
            ctx.set_iteration_callbacks(None)

    plain = run(None)
    seen = []
    s = run([lambda it: seen.append((it.iteration, it.step_is_valid, it.step_is_successful, it.cost, it.cost_change, it.trust_region_radius))])
    # observing changes nothing (the Schur sums use floating-point atomics: run-to-run equality holds to rounding only)
    assert abs(s["final_cost"] - plain["final_cost"]) <= 1e-10 * plain["final_cost"] and s["iterations"] == plain["iterations"]
    assert [x[0] for x in seen] == list(range(0, s["iterations"] + 1))
    assert seen[0][1:3] == (0, 0) and seen[0][3] == s["initial_cost"] and seen[-1][3] == s["final_cost"]
    assert sum(x[2] for x in seen) == s["num_successful"]
    for prev, cur in zip(seen, seen[1:]):
        if cur[2]:
            assert abs((prev[3] - cur[3]) - cur[4]) <= 1e-12 * prev[3] and cur[3] < prev[3]
        else:
            assert cur[3] == prev[3]
    # abort after the second iteration / declare success after the first
    s_abort = run([lambda it: 1 if it.iteration == 2 else 0])
    assert s_abort["iterations"] == 2 and s_abort["termination"] == 2                                  # PXR_TERM_FAILURE
    s_ok = run([lambda it: None, lambda it: 2 if it.iteration == 1 else 0])
    assert s_ok["iterations"] == 1 and s_ok["termination"] == 0 and abs(s_ok["final_cost"] - seen[1][3]) <= 1e-10 * seen[1][3]
    s0 = run([lambda it: 2])
    assert s0["iterations"] == 0 and s0["final_cost"] == s0["initial_cost"]

    # an exception inside a callback aborts the solve at that iteration and surfaces when the hook is removed
    def boom(it):
        if it.iteration == 1:
            raise KeyError("from the callback")
    with pytest.raises(KeyError, match="from the callback"):
        run([boom])


def test_patch_interpolator_local_coordinates(ctx):
    """PatchInterpolator.interpolate_local (dynamic_patch_interpolator.h:125-132): the keypoint in the patch's own pixel
    coordinates (column, row) -- PixelInterpolator::Evaluate(r = xy[1], c = xy[0]) without the image -> patch transform."""
    import pxo
    from pixsfm_amd.api import features
    rng = np.random.default_rng(8)
    data = rng.normal(size=(16, 16, 128)).astype(np.float16)
    fpatch = features.FeaturePatch(data, (37, 91), (0.5, 0.25))            # corner and scale must not matter
    interp = features.PatchInterpolator({"l2_normalize": True}, ctx=ctx)
    patch = pxo.make_patch(data, (37, 91), (0.5, 0.25))
    for xy in ([7.3, 4.9], [0.2, 14.6], [11.0, 3.0]):
        got = interp.interpolate_local(fpatch, xy)
        want = pxo.pixel_interp(patch, xy[1], xy[0], pxo.cfg())[0]
        assert got.shape == (128,) and np.abs(got - want).max() < 1e-12


def test_prefetched_upload_equals_the_plain_flow(ctx, monkeypatch):
    """Feature maps built by the reference's numpy constructor FeatureMap(patches [N][H][W][C], ids, corners, metadata) (what
    extract.py hands over, featuremap.cc:8-45) remember their array: BundleAdjuster.refine then starts the upload in a
    background thread from the arrays' addresses while it walks the scene objects (features.SharedArena.prefetch).  Same
    result as with maps filled patch by patch (no prefetch possible), incl. a map that holds patches of keypoints without
    a 3D point and after one map was edited (its array no longer mirrors the dict: prefetch declines, plain flow)."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    prob = synthetic.make_ba_problem(n_cams=5, n_points=60, obs_per_point=3, seed=23, noise=0.05)

    def inputs(stacked, edit=False):
        rec, patch_of = reconstruction_from_flat(prob)
        per = {}
        for (image_id, p2d), pi in sorted(patch_of.items()):
            per.setdefault(image_id, []).append((p2d, pi))
        fmaps = {}
        for image_id, items in per.items():
            ids = np.array([a for a, _ in items]); pis = np.array([b for _, b in items])
            if stacked:
                fmaps[rec.images[image_id].name] = features.FeatureMap(np.ascontiguousarray(prob["patches"][pis]), ids, prob["corners"][pis],
                                                                       {"scale": prob["scales"][pis[0]], "is_sparse": True, "patch_size": 16})
            else:
                fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
                for a, b in items:
                    fm.patches[a] = features.FeaturePatch(prob["patches"][b], prob["corners"][b], prob["scales"][b])
        if edit:
            fm = next(iter(fmaps.values()))
            k0 = next(iter(fm.patches))
            fm.add_fpatch(k0, features.FeaturePatch(fm.patches[k0].data.copy(), fm.patches[k0].corner, fm.patches[k0].scale))
            assert fm.stacked() is None
        return rec, features.FeatureManager([features.FeatureSet(fmaps)])
    started = []
    real = features.SharedArena.prefetch
    monkeypatch.setattr(features.SharedArena, "prefetch", lambda self, *a, **k: started.append(real(self, *a, **k)) or started[-1])
    conf = {"optimizer": {"solver": {"max_num_iterations": 6}}}
    results = []
    for stacked, edit in ((False, False), (True, False), (True, True)):
        rec, fm = inputs(stacked, edit)
        out = BundleAdjuster.create(conf).refine_multilevel(rec, fm)
        results.append((out["summary"][0], np.array([rec.points3D[p + 1].xyz for p in range(60)])))
    assert started == [False, True, False]
    for s, X in results[1:]:
        assert s.num_iterations == results[0][0].num_iterations
        assert abs(s.final_cost - results[0][0].final_cost) < 1e-7 * results[0][0].initial_cost
        assert np.abs(X - results[0][1]).max() < 1e-6


def test_keypoint_adjuster_prefetches_stacked_feature_maps(ctx, monkeypatch):
    """KeypointAdjuster.refine starts the upload of feature maps built by the reference's numpy constructor in a background thread
    beside the edge construction and the walk over the keypoint objects (features.SharedArena.prefetch, like BundleAdjuster);
    maps filled patch by patch take the plain flow.  Same keypoints either way -- also when a map holds patches of keypoints
    that are not in the graph."""
    from pixsfm_amd.api import KeypointAdjuster, features
    started = []
    real = features.SharedArena.prefetch
    monkeypatch.setattr(features.SharedArena, "prefetch", lambda self, *a, **k: started.append(real(self, *a, **k)) or started[-1])
    results = []
    for stacked in (False, True):
        prob, keypoints, graph, fmanager, (img, kid, names) = _ka_inputs(seed=5, n_tracks=14, track_len=4)
        if not stacked:
            fmaps = {}
            for k, nm in enumerate(names):
                fm = features.FeatureMap()
                for t in np.nonzero(img == k)[0]:
                    fm.patches[int(kid[t])] = features.FeaturePatch(prob["patches"][t], prob["corners"][t], (1.0, 1.0))
                fmaps[nm] = fm
            fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
        else:           # one more patch per map, of a keypoint the graph does not know
            fmaps = {}
            for k, nm in enumerate(names):
                sel = np.nonzero(img == k)[0]
                extra = np.concatenate([prob["patches"][sel], prob["patches"][sel[:1]]])
                fmaps[nm] = features.FeatureMap.from_arrays(extra, np.concatenate([kid[sel], [10_000]]),
                                                           np.concatenate([prob["corners"][sel], prob["corners"][sel[:1]]]), (1.0, 1.0))
            fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
        out = KeypointAdjuster.create({"strategy": "featuremetric"}).refine_multilevel(keypoints, fmanager, graph)
        results.append((out["summary"][0], np.concatenate([keypoints[nm] for nm in names])))
    assert started == [False, True]
    (s0, k0), (s1, k1) = results
    assert s0.num_iterations == s1.num_iterations
    assert abs(s0.final_cost - s1.final_cost) < 1e-9 * s0.initial_cost
    assert np.abs(k0 - k1).max() < 1e-7
