"""GPU parity for the cost-map path: pxr_costmap_extract vs oracle/pxo_costmap.py,
the 3- / 1-channel residual kernel and the cost-map BA (pxr_ba_solve with no reference) vs the oracle.

Tolerances: the cost maps are STORED in the features' dtype, so the bar is the storage type's: identical
bits except where the fp64 summation order (16-lane tree here, numpy's einsum there) moves a value across a
rounding boundary -- at most 1 ulp, on a small fraction of the entries; fp64 maps within 1e-12.  Residuals /
Jacobians within 1e-10 relative (north_star: 1e-5), refined parameters within 1e-6 (north_star: 1e-4).
The extraction kernel is compared with the oracle's restatement of FillPointCostmap in tests/test_costmap_extract.py.
The cost-map BA (a trust-region solve) is compared with the oracle only.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOSSES = [("trivial", []), ("cauchy", [0.25])]


def _setup(ctx, **kw):
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena
    prob = synthetic.make_ba_problem(**kw)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    return prob, arena, BAProblem(ctx, arena, prob)


def _ulps(a, b):
    """Distance in units of the storage type's last place (same-sign finite values)."""
    it = {2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
    ia, ib = a.view(it).astype(np.int64), b.view(it).astype(np.int64)
    sign = np.int64(1) << (8 * a.dtype.itemsize - 1)
    ia = np.where(ia < 0, -(ia + sign), ia)      # sign-magnitude -> monotone integers
    ib = np.where(ib < 0, -(ib + sign), ib)
    return np.abs(ia - ib)


def _check_maps(got, want):
    assert got.dtype == want.dtype and got.shape == want.shape
    if got.dtype == np.float64:
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        return
    d = _ulps(got, want)
    assert d.max() <= 1, "more than one ulp apart"
    # half maps: the 8-channel partial sums of a lane are fp32 (csrc/pxr_costmap.hip texel_sums; everything across lanes is fp64) --
    # the fp16 bits are the all-fp64 reference's on >= 99.9 % of the entries, one unit in the last place elsewhere (VERDICT r4 next-7;
    # measured in round 4 over 2.3 M entries: 7e-5)
    assert (d > 0).mean() < (1e-3 if got.dtype == np.float16 else 2e-3), "too many entries differ in the last place: %g" % (d > 0).mean()


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
@pytest.mark.parametrize("loss", LOSSES)
def test_costmaps_match_oracle(ctx, dtype, loss):
    import pxo_costmap
    from pixsfm_amd.engine import make_loss
    prob, arena, ba = _setup(ctx, n_cams=5, n_points=40, obs_per_point=3, seed=71, dtype=dtype)
    cm = ba.extract_costmaps(make_loss(*loss))
    got, corners, scales = cm.download()
    want = pxo_costmap.costmaps(prob["patches"], prob["obs_patch"], prob["obs_point"], prob["refs"],
                                loss=(loss[0], loss[1][0] if loss[1] else 1.0))
    _check_maps(got, want)
    # CreateShallowCostmapFSet: the cost patch sits where the feature patch sits
    assert np.array_equal(corners, prob["corners"][prob["obs_patch"]])
    assert np.array_equal(scales, prob["scales"][prob["obs_patch"]])


@pytest.mark.parametrize("kw", [dict(apply_sqrt=True), dict(as_gradientfield=False), dict(as_gradientfield=False, apply_sqrt=True)])
def test_costmap_config_variants(ctx, kw):
    import pxo_costmap
    from pixsfm_amd.engine import make_loss
    prob, arena, ba = _setup(ctx, n_cams=4, n_points=30, obs_per_point=3, seed=72)
    cm = ba.extract_costmaps(make_loss("cauchy", [0.25]), **kw)
    want = pxo_costmap.costmaps(prob["patches"], prob["obs_patch"], prob["obs_point"], prob["refs"], loss=("cauchy", 0.25), **kw)
    _check_maps(cm.download()[0], want)


@pytest.mark.parametrize("channels,patch_size", [(64, 16), (128, 8), (128, 10)])
def test_costmap_shapes(ctx, channels, patch_size):
    """C = 64 (8 lanes per texel), the low-memory configuration's 8 x 8 patches (configs/low_memory.yaml:7), and a
    width that does not fill whole wavefronts; fp32 cost maps of fp16 features."""
    import pxo_costmap
    from pixsfm_amd.engine import make_loss
    prob, arena, ba = _setup(ctx, n_cams=4, n_points=25, obs_per_point=3, seed=73, channels=channels, patch_size=patch_size)
    cm = ba.extract_costmaps(make_loss("trivial", []), dtype=np.float32)
    want = pxo_costmap.costmaps(prob["patches"], prob["obs_patch"], prob["obs_point"], prob["refs"], out_dtype=np.float32)
    _check_maps(cm.download()[0], want)


def test_costmap_is_zero_at_the_reference_texel(ctx):
    """Closed-form property: where a texel equals the reference descriptor the cost and both derivatives vanish, and the
    cost channel is 0.5 |f - ref|^2 everywhere (trivial loss)."""
    from pixsfm_amd.engine import BAProblem, PatchArena, make_loss
    rng = np.random.default_rng(5)
    patches = rng.normal(size=(3, 16, 16, 128))
    patches /= np.linalg.norm(patches, axis=-1, keepdims=True)
    patches = patches.astype(np.float16)
    refs = np.stack([patches[0, 4, 7], patches[1, 15, 0], patches[2, 0, 15]]).astype(np.float64)
    prob = dict(obs_image=np.zeros(3, np.int32), obs_point=np.arange(3, dtype=np.int32), obs_patch=np.arange(3, dtype=np.int64),
                image_camera=np.zeros(1, np.int32), qvec=np.array([[1.0, 0, 0, 0]]), tvec=np.zeros((1, 3)),
                cam_model=np.zeros(1, np.int32), cam_params=np.array([[100.0, 50, 50]]), xyz=np.array([[0, 0, 5.0]] * 3), refs=refs)
    arena = PatchArena.from_numpy(ctx, patches, np.zeros((3, 2), np.int32))
    cm = BAProblem(ctx, arena, prob).extract_costmaps(make_loss("trivial", []), dtype=np.float64).download()[0]
    for i, (y, x) in enumerate([(4, 7), (15, 0), (0, 15)]):
        assert np.all(cm[i, y, x] == 0.0)
    want = 0.5 * ((patches.astype(np.float64) - refs[:, None, None, :]) ** 2).sum(-1)
    assert np.abs(cm[..., 0] - want).max() < 1e-13


def _costmap_problem(prob, maps):
    """The flat problem the oracle sees for the cost-map BA: observation i reads map i, no references."""
    p = dict(prob)
    p["patches"], p["obs_patch"], p["refs"] = maps, np.arange(len(maps), dtype=np.int64), None
    p["corners"], p["scales"] = prob["corners"][prob["obs_patch"]], prob["scales"][prob["obs_patch"]]
    return p


@pytest.mark.parametrize("dtype", [np.float16, np.float64])
@pytest.mark.parametrize("as_gradientfield", [True, False])
def test_costmap_residual_blocks_match_oracle(ctx, dtype, as_gradientfield):
    """FeatureReferenceCostFunctor<..., CHANNELS = 3 | 1> with ref_descriptor = nullptr (costmap_bundle_optimizer.h:104-119,
    feature_reference.h:128-130): residuals = the bicubic texel of the cost map ([upstream] scalar Ceres path, C < 8)."""
    import pxo
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ba = _setup(ctx, n_cams=5, n_points=50, obs_per_point=3, seed=74, dtype=dtype)
    cm = ba.extract_costmaps(make_loss("trivial", []), as_gradientfield=as_gradientfield)
    cba = ba.costmap_problem(cm)
    rec, r, gx, gy = cba.eval(interp_cfg(l2_normalize=False), with_jacobian=True, materialize=True)
    P = cba.projection_jacobian().download()
    r, gx, gy, rec = r.download(), gx.download(), gy.download(), rec.download()
    cost_o, r_o, J_o = pxo.ba_eval_batch(_costmap_problem(prob, cm.download()[0]), pxo.cfg(l2_normalize=False),
                                         pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert rel(r, r_o) < 1e-10
    J = gx[:, :, None] * P[:, None, 0, :] + gy[:, :, None] * P[:, None, 1, :]
    assert rel(J, J_o) < 1e-10
    assert rel(rec[:, 0], (r_o ** 2).sum(1)) < 1e-10
    assert rel(rec[:, 1], (gx * gx).sum(1)) < 1e-10 and rel(rec[:, 3], (gy * gy).sum(1)) < 1e-10
    assert abs(cba.cost(make_loss("cauchy", [0.25])) - cost_o) < 1e-10 * abs(cost_o)


def _gauge(prob):
    """default_problem_setup (bundle_adjustment/main.py:12-18) with the BundleOptimizerOptions defaults (focal + extra refined)."""
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    cmask = np.full(n_cam, (1 << 1) | (1 << 2), np.uint16)    # SIMPLE_RADIAL: principal point constant
    return pose_const, tmask, cmask, np.zeros(n_pt, np.uint8)


@pytest.mark.parametrize("inner", [False, True])
def test_costmap_bundle_adjustment_matches_oracle(ctx, inner):
    """CostMapBundleAdjuster.refine (bundle_adjustment/main.py:243-286): references -> cost maps -> BA on the maps with
    l2_normalize = False; same LM loop as the feature-reference BA, trajectories must coincide with the oracle's."""
    import pxo
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, arena, ba = _setup(ctx, n_cams=6, n_points=80, obs_per_point=4, seed=75)
    cm = ba.extract_costmaps(make_loss("trivial", []))
    cba = ba.costmap_problem(cm)
    gauge = _gauge(prob)
    kw = dict(max_iterations=5, use_inner_iterations=inner)    # still descending: the accept/reject decisions must coincide
    s_gpu = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**kw))
    q, t, k, X = cba.params()
    s_cpu, qo, to, ko, Xo = pxo.ba_solve(_costmap_problem(prob, cm.download()[0]), pxo.cfg(l2_normalize=False),
                                         pxo.loss("cauchy", 0.25), *gauge, pxo.lm_options(**kw))
    assert s_gpu["iterations"] == s_cpu["iterations"] and s_gpu["num_successful"] == s_cpu["num_successful"]
    assert abs(s_gpu["initial_cost"] - s_cpu["initial_cost"]) < 1e-10 * s_cpu["initial_cost"]
    assert abs(s_gpu["final_cost"] - s_cpu["final_cost"]) < 1e-6 * s_cpu["final_cost"]
    assert s_gpu["final_cost"] < s_gpu["initial_cost"]
    for a, b in zip((q, t, k, X), (qo, to, ko, Xo)):
        b = np.asarray(b)
        a = a[:, :b.shape[1]] if a.ndim == 2 else a
        assert np.abs(a - b).max() < 1e-6 * max(1.0, np.abs(b).max())
    # the cost-map problem refined the SAME device parameters the feature problem holds
    assert np.array_equal(ba.params()[3], X)


def test_costmap_argument_errors(ctx):
    from pixsfm_amd._lib import PixsfmHipError
    from pixsfm_amd.engine import BAProblem, PatchArena, make_loss
    prob, arena, ba = _setup(ctx, n_cams=3, n_points=10, obs_per_point=2, seed=76)
    noref = BAProblem(ctx, arena, dict(prob, refs=None))
    with pytest.raises(ValueError):
        noref.extract_costmaps(make_loss("trivial", []))
    import ctypes as C
    wrong = PatchArena(ctx, ba.n_obs, 16, 16, 4)        # the 4-channel (cross derivative) layout belongs to pxr_costmap_extract_ex
    with pytest.raises(PixsfmHipError):
        from pixsfm_amd._lib import check
        check(ctx.lib.pxr_costmap_extract(ctx.handle, arena.handle, wrong.handle, 0, ba.n_obs, ba.d["obs_patch"].ptr,
                                          ba.d["obs_point"].ptr, ba.d["refs"].ptr, C.byref(make_loss("trivial", [])), 1, 0),
              "pxr_costmap_extract")


def test_costmap_bundle_adjuster_like_pixsfm(ctx):
    """BundleAdjuster.create({"strategy": "costmaps"}) the way pixsfm's low_memory configuration drives it
    (configs/low_memory.yaml:28-47, bundle_adjustment/main.py:218-286) against the same steps on the low-level engine."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, CostMapBundleAdjuster, CostMapBundleOptimizer, features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=6, n_points=70, obs_per_point=4, seed=19, noise=0.02)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
    adjuster = BundleAdjuster.create({"strategy": "costmaps", "optimizer": {"solver": {"max_num_iterations": 6}}})
    assert isinstance(adjuster, CostMapBundleAdjuster)
    out = adjuster.refine_multilevel(rec, fmanager)
    summary, references, cost_fset = out["summary"][0], out["references"][0], out["costmaps"][0]
    assert len(references) == 70 and summary.final_cost < 0.1 * summary.initial_cost
    assert cost_fset.channels == 3 and cost_fset.arena.n == len(prob["obs_image"]) and cost_fset.arena.dtype == np.float16
    assert summary.num_residuals_reduced == 3 * len(prob["obs_image"])
    # the cost map of an observation sits where its feature patch sits
    (image_id, p2d), pi = next(iter(patch_of.items()))
    cp = cost_fset.fmap(rec.images[image_id].name).fpatch(p2d)
    _, corner, scale = cost_fset.arena.download(cp.index, 1)
    assert np.array_equal(corner[0], prob["corners"][pi]) and np.array_equal(scale[0], prob["scales"][pi])
    # the same steps on the low-level engine
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=100)
    cba = ba.costmap_problem(ba.extract_costmaps(make_loss("trivial", [])))
    n_img = 6
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    s = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), pose_const, tmask, np.full(n_img, 0b0110, np.uint16),
                  np.zeros(70, np.uint8), options=lm_options(max_iterations=6, use_inner_iterations=True))
    q, t, k, X = cba.params()
    assert abs(s["initial_cost"] - summary.initial_cost) < 1e-9 * s["initial_cost"]
    assert abs(s["final_cost"] - summary.final_cost) < 1e-4 * max(s["final_cost"], 1e-12)
    assert np.abs(np.array([rec.images[i + 1].qvec for i in range(n_img)]) - q).max() < 1e-4
    assert np.abs(np.array([rec.points3D[p + 1].xyz for p in range(70)]) - X).max() < 1e-4
    # only 1- and 3-channel maps are cost maps (costmap_bundle_optimizer.h:9-14)
    from pixsfm_amd.api import default_problem_setup
    with pytest.raises(ValueError, match="Unsupported dimensions"):
        CostMapBundleOptimizer({}, default_problem_setup(rec), {"l2_normalize": False}).run(rec, fmanager.fset(0))


def test_costmap_extractor_in_chunks_equals_one_pass(ctx):
    """A scene whose feature patches exceed CostMapExtractor.chunk_bytes is processed in chunks of whole points with one
    cost-map arena: same maps (bit for bit), same references, whatever the chunking."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import CostMapExtractor, ReferenceExtractor, features
    from pixsfm_amd.api.bundle_adjustment import find_problem_labels
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    prob = synthetic.make_ba_problem(n_cams=5, n_points=37, obs_per_point=3, seed=23, noise=0.02)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fset = features.FeatureSet(fmaps)
    labels = find_problem_labels(rec, 10)
    out = {}
    for name, chunk_bytes in (("one", 16 << 30), ("many", 10 * 16 * 16 * 128 * 2)):      # 10 patches per chunk -> ~12 chunks
        ce = CostMapExtractor({"loss": {"name": "cauchy", "params": [0.25]}}, {}, chunk_bytes=chunk_bytes)
        cost_fset, refs = ce.run(labels, rec, fset, ReferenceExtractor({}, {}))
        maps = {}
        for name_img, fm in cost_fset.fmaps.items():
            for p2d, cp in fm.patches.items():
                maps[(name_img, p2d)] = cost_fset.arena.download(cp.index, 1)
        out[name] = (maps, refs)
    (m1, r1), (m2, r2) = out["one"], out["many"]
    assert m1.keys() == m2.keys() and len(m1) == len(prob["obs_image"])
    for key in m1:
        for a, b in zip(m1[key], m2[key]):
            assert np.array_equal(a, b)
    assert r1.keys() == r2.keys()
    for pid in r1:
        assert r1[pid].source == r2[pid].source and np.array_equal(r1[pid].descriptor, r2[pid].descriptor)


# pixsfm/configs/low_memory.yaml, `mapping` section (lines 11-47), as the dicts OmegaConf hands to the adjusters
LOW_MEMORY_KA = {"apply": True, "strategy": "topological_reference", "split_in_subproblems": True, "max_kps_per_problem": 1000,
                 "optimizer": {"num_threads": -1, "print_summary": False, "bound": 2.0, "solver": {"parameter_tolerance": 1.0e-5}}}
LOW_MEMORY_BA = {"apply": True, "strategy": "costmaps", "repeats": 1, "num_threads": -1, "level_indices": None,
                 "max_tracks_per_problem": 100, "references": {"keep_observations": False}, "costmaps": {"num_threads": -1},
                 "optimizer": {"loss": {"name": "cauchy", "params": [0.25]}, "print_summary": False, "refine_focal_length": False,
                               "refine_principal_point": False, "refine_extra_params": False, "refine_extrinsics": False}}


def test_low_memory_configuration_end_to_end(ctx):
    """The reference's low-memory configuration verbatim: 8 x 8 half patches, topological-reference KA with bound 2, cost-map
    BA that refines the 3D points only (all refine_* flags off)."""
    from pixsfm_amd import synthetic, synthetic_ka
    from pixsfm_amd.api import BundleAdjuster, KeypointAdjuster, features
    from pixsfm_amd.api.keypoint_adjustment import build_matching_graph
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    # ---- KA on 8 x 8 patches -------------------------------------------------------------------------------------
    n_tracks, track_len = 10, 4
    kp = synthetic_ka.make_ka_problem(n_tracks=n_tracks, track_len=track_len, seed=31, patch_size=8, sigma=0.5, directed_both=False)
    n = n_tracks * track_len
    img, kid = np.arange(n) % track_len, np.arange(n) // track_len
    names = ["im%d.jpg" % k for k in range(track_len)]
    keypoints = {names[k]: kp["kp"][img == k].copy() for k in range(track_len)}
    pairs, matches, scores = [], [], []
    for a in range(track_len):
        for b in range(a + 1, track_len):
            sel = (img[kp["edge_src"]] == a) & (img[kp["edge_dst"]] == b)
            pairs.append((names[a], names[b]))
            matches.append(np.stack([kid[kp["edge_src"][sel]], kid[kp["edge_dst"][sel]]], 1))
            scores.append(kp["edge_w"][sel])
    graph = build_matching_graph(pairs, matches, scores)
    fmaps = {names[k]: features.FeatureMap.from_arrays(kp["patches"][img == k], kid[img == k], kp["corners"][img == k], (1.0, 1.0))
             for k in range(track_len)}
    ka = KeypointAdjuster.create(LOW_MEMORY_KA)
    assert ka.conf["optimizer"]["bound"] == 2.0 and ka.conf["max_kps_per_problem"] == 1000
    s = ka.refine_multilevel(keypoints, features.FeatureManager([features.FeatureSet(fmaps)]), graph)["summary"][0]
    assert s.final_cost < 0.5 * s.initial_cost
    # ---- cost-map BA on 8 x 8 patches, points only ---------------------------------------------------------------
    prob = synthetic.make_ba_problem(n_cams=6, n_points=60, obs_per_point=4, seed=32, patch_size=8, rot_deg=0.0, trans=0.0, pt_sigma=0.01)
    rec, patch_of = reconstruction_from_flat(prob)
    fm = {}
    for (image_id, p2d), pi in patch_of.items():
        fm.setdefault(rec.images[image_id].name, features.FeatureMap()).patches[p2d] = \
            features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    q0 = {i: rec.images[i].qvec.copy() for i in rec.images}
    k0 = {c: rec.cameras[c].params.copy() for c in rec.cameras}
    x0 = np.array([rec.points3D[p + 1].xyz for p in range(60)])
    ba = BundleAdjuster.create(LOW_MEMORY_BA)
    assert ba.conf["repeats"] == 1                                         # pipeline-level keys pass through (OmegaConf.merge)
    out = ba.refine_multilevel(rec, features.FeatureManager([features.FeatureSet(fm)]))
    s = out["summary"][0]
    assert out["costmaps"][0].arena.H == 8 and out["costmaps"][0].arena.C == 3
    assert s.final_cost < 0.2 * s.initial_cost
    # refine_extrinsics: false (AddImageToProblem still normalises qvec in place, bundle_optimizer.h:255)
    assert all(np.abs(rec.images[i].qvec - q0[i] / np.linalg.norm(q0[i])).max() < 1e-15 for i in rec.images)
    assert all(np.array_equal(rec.cameras[c].params, k0[c]) for c in rec.cameras)      # no intrinsics refined
    x1 = np.array([rec.points3D[p + 1].xyz for p in range(60)])
    err0, err1 = np.abs(x0 - prob["gt_xyz"]).max(), np.abs(x1 - prob["gt_xyz"]).max()
    assert np.abs(x1 - x0).max() > 1e-4 and err1 < err0                                # the points moved towards the truth


def test_costmap_ba_check_bounds(ctx):
    """check_bounds on cost maps (the functor returns is_inside when no reference is given, feature_reference.h:128-130):
    observations projecting outside their map fail the evaluation -> NaN block norm, FAILURE at the initial point; the
    same problem without the option evaluates (border-clamped) and solves."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=4, n_points=40, obs_per_point=3, seed=3)
    prob["corners"] = prob["corners"].copy()
    prob["corners"][::5, 0] += 9
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cba = ba.costmap_problem(ba.extract_costmaps(make_loss("trivial", [])))
    rec_on = cba.eval(interp_cfg(l2_normalize=False, check_bounds=True), with_jacobian=True)[0].download()
    rec_off = cba.eval(interp_cfg(l2_normalize=False), with_jacobian=True)[0].download()
    uv = rec_off[:, 6:8] * prob["scales"] - 0.5 - prob["corners"]
    inside = (uv[:, 0] > 0) & (uv[:, 0] < 16) & (uv[:, 1] > 0) & (uv[:, 1] < 16)      # patch_interpolator.h:160-166
    assert 0 < (~inside).sum() < len(inside)
    assert np.array_equal(np.isnan(rec_on[:, 0]), ~inside) and np.array_equal(rec_on[inside], rec_off[inside])
    gauge = (np.array([1, 0, 0, 0], np.uint8), np.array([0, 1, 0, 0], np.uint8), np.full(4, 0b0110, np.uint16), np.zeros(40, np.uint8))
    for inner in (False, True):
        s = cba.solve(interp_cfg(l2_normalize=False, check_bounds=True), make_loss("cauchy", [0.25]), *gauge,
                      options=lm_options(max_iterations=5, use_inner_iterations=inner))
        assert s["termination"] == 2 and s["iterations"] == 0 and np.isnan(s["initial_cost"])
    s = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=2))
    assert s["termination"] != 2 and np.isfinite(s["final_cost"])


@pytest.mark.parametrize("H,W", [(12, 12), (12, 10), (20, 5), (1, 16), (7, 1), (1, 1)])
@pytest.mark.parametrize("channels", [128, 64])
def test_costmaps_of_odd_patch_shapes(ctx, H, W, channels):
    """The generic extraction kernel: non-square patches, the 12 x 12 windows the reference cuts out of dense maps
    (dense_cut_size, costmap_extractor.h:50), and the degenerate 1-texel rows / columns where the clamped central
    differences vanish (costmap_extractor.h:260-263)."""
    import pxo_costmap
    from pixsfm_amd.engine import BAProblem, PatchArena, make_loss
    rng = np.random.default_rng(H * 100 + W)
    n = 9
    patches = rng.normal(size=(n, H, W, channels))
    patches = (patches / np.linalg.norm(patches, axis=-1, keepdims=True)).astype(np.float16)
    refs = rng.normal(size=(4, channels)); refs /= np.linalg.norm(refs, axis=-1, keepdims=True)
    obs_point = rng.integers(0, 4, n).astype(np.int32)
    prob = dict(obs_image=np.zeros(n, np.int32), obs_point=obs_point, obs_patch=np.arange(n, dtype=np.int64),
                image_camera=np.zeros(1, np.int32), qvec=np.array([[1.0, 0, 0, 0]]), tvec=np.zeros((1, 3)),
                cam_model=np.zeros(1, np.int32), cam_params=np.array([[100.0, 50, 50]]), xyz=np.array([[0, 0, 5.0]] * 4), refs=refs)
    arena = PatchArena.from_numpy(ctx, patches, rng.integers(0, 50, (n, 2)).astype(np.int32))
    ba = BAProblem(ctx, arena, prob)
    for kw in (dict(), dict(as_gradientfield=False)):
        got = ba.extract_costmaps(make_loss("cauchy", [0.25]), **kw).download()[0]
        want = pxo_costmap.costmaps(patches, prob["obs_patch"], obs_point, refs, loss=("cauchy", 0.25), **kw)
        _check_maps(got, want)
        if H == 1 and "as_gradientfield" not in kw:
            assert np.all(got[..., 1] == 0)
        if W == 1 and "as_gradientfield" not in kw:
            assert np.all(got[..., 2] == 0)


def test_costmaps_of_dense_feature_maps(ctx):
    """Dense feature maps (FeatureMap.is_sparse = False): the reference slices a dense_cut_size window around the current
    reprojection of every observation (FeaturePatch::ToCorner / Slice, featurepatch.cc:324-359; costmap_extractor.h:210-222,
    401-431) and fills the cost map from that copy; the references come from the full maps."""
    import pxo
    import pxo_costmap
    from pixsfm_amd import synthetic
    from pixsfm_amd.api import BundleAdjuster, CostMapExtractor, ReferenceExtractor, features
    from pixsfm_amd.api.bundle_adjustment import find_problem_labels
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    prob = synthetic.make_ba_problem(n_cams=4, n_points=30, obs_per_point=3, seed=41)     # geometry only
    rec, patch_of = reconstruction_from_flat(prob)
    rng = np.random.default_rng(6)
    h = w = 50
    scale = np.array([0.05, 0.05])                                                          # 1000 x 1000 images
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    freq, phase = rng.uniform(0, 0.25, (128, 2)), rng.uniform(0, 6.28, 128)
    fmaps, dense = {}, {}
    for i, im in rec.images.items():
        m = np.cos(freq[:, 0] * (xs[..., None] + 0.3 * i) + freq[:, 1] * (ys[..., None] - 0.2 * i) + phase)
        dense[i] = (m / np.linalg.norm(m, axis=-1, keepdims=True)).astype(np.float16)
        fmaps[im.name] = features.FeatureMap.dense(dense[i], scale)
    fset = features.FeatureSet(fmaps)
    ps = 12
    cost_fset, refs = CostMapExtractor({"dense_cut_size": ps}, {}).run(find_problem_labels(rec, 10), rec, fset, ReferenceExtractor({}, {}))
    assert cost_fset.arena.H == ps and cost_fset.arena.W == ps and cost_fset.arena.n == len(prob["obs_image"])
    for (image_id, p2d), _ in patch_of.items():
        im = rec.images[image_id]
        cam = rec.cameras[im.camera_id]
        pid = im.points2D[p2d].point3D_id
        xy = pxo.world_to_pixel(cam.model_id, np.asarray(cam.params, float), im.qvec, im.tvec, rec.points3D[pid].xyz, jac=False)[0]
        c = np.trunc(xy * scale - 0.5 - ps / 2.0).astype(int)
        x0, y0 = int(np.clip(c[0], 0, w - ps)), int(np.clip(c[1], 0, h - ps))
        want = pxo_costmap.fill_point_costmap(dense[image_id][y0:y0 + ps, x0:x0 + ps], refs[pid].descriptor.reshape(-1))
        cp = cost_fset.fmap(im.name).fpatch(p2d)
        got, corner, sc = cost_fset.arena.download(cp.index, 1)
        assert np.array_equal(corner[0], [x0, y0]) and np.array_equal(sc[0], scale)
        _check_maps(got[0], want)
    # and the whole strategy on dense maps
    s = BundleAdjuster.create({"strategy": "costmaps", "optimizer": {"solver": {"max_num_iterations": 4}}}).refine_multilevel(
        rec, features.FeatureManager([fset]))["summary"][0]
    assert s.final_cost < s.initial_cost and s.num_residuals_reduced == 3 * len(prob["obs_image"])


def test_costmap_inner_iterations_on_long_tracks(ctx):
    """Tracks longer than the 8 lanes a point gets in the cost-map inner-iteration kernel (two passes per evaluation, the
    tail of the observation data read from global memory instead of the LDS staging): same trajectory as the oracle."""
    import pxo
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, arena, ba = _setup(ctx, n_cams=14, n_points=40, obs_per_point=11, seed=77)
    cm = ba.extract_costmaps(make_loss("trivial", []))
    cba = ba.costmap_problem(cm)
    gauge = _gauge(prob)
    kw = dict(max_iterations=4, use_inner_iterations=True)
    s_gpu = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**kw))
    q, t, k, X = cba.params()
    s_cpu, qo, to, ko, Xo = pxo.ba_solve(_costmap_problem(prob, cm.download()[0]), pxo.cfg(l2_normalize=False),
                                         pxo.loss("cauchy", 0.25), *gauge, pxo.lm_options(**kw))
    assert s_gpu["iterations"] == s_cpu["iterations"] and s_gpu["num_successful"] == s_cpu["num_successful"]
    assert abs(s_gpu["final_cost"] - s_cpu["final_cost"]) < 1e-6 * s_cpu["final_cost"]
    assert np.abs(X - Xo).max() < 1e-6 and np.abs(q - qo).max() < 1e-6


@pytest.mark.parametrize("dtype,channels", [(np.float16, 128), (np.float32, 64)])
@pytest.mark.parametrize("up,grad,cross,sqrt_,loss", [(2.0, True, False, False, ("trivial", [])),
                                                    (1.0, True, True, False, ("cauchy", [0.25])),
                                                    (1.0, True, True, True, ("cauchy", [0.25])),
                                                    (1.5, False, False, True, ("trivial", [])),
                                                    (2.0, True, True, False, ("huber", [0.3]))])
def test_interpolated_costmaps_match_oracle(ctx, dtype, channels, up, grad, cross, sqrt_, loss):
    """CostMapConfig.upsampling_factor / compute_cross_derivative: the INTERPOLATING branch of FillPointCostmap
    (costmap_extractor.h:280-284, 304-317, 341-345) -- bicubic evaluation at (x, y) / upsampling_factor under the
    extractor's InterpolationConfig (L2 normalisation on), 4 channels with the cross derivative -- against the oracle's
    statement-by-statement loop (vs the oracle: parity unpinned, the reference has no cost-map test)."""
    import pxo
    import pxo_costmap
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ba = _setup(ctx, n_cams=3, n_points=4, obs_per_point=2, seed=91, dtype=dtype, channels=channels, patch_size=8)
    cm = ba.extract_costmaps(make_loss(*loss), as_gradientfield=grad, apply_sqrt=sqrt_, upsampling_factor=up,
                             compute_cross_derivative=cross, cfg=interp_cfg())
    co = (4 if cross else 3) if grad else 1
    assert (cm.H, cm.W, cm.C) == (int(8 * (up + 1e-6)), int(8 * (up + 1e-6)), co) and cm.upsampling_factor == up
    got, corners, scales = cm.download()
    assert np.array_equal(corners, prob["corners"][prob["obs_patch"]])
    for i in range(len(got)):
        want = pxo_costmap.fill_point_costmap_interpolated(
            prob["patches"][prob["obs_patch"][i]], prob["refs"][prob["obs_point"][i]], pxo.cfg(),
            loss=(loss[0], loss[1][0] if loss[1] else 1.0), as_gradientfield=grad, apply_sqrt=sqrt_, upsampling_factor=up,
            compute_cross_derivative=cross)
        assert want.shape == got[i].shape and want.dtype == got[i].dtype
        scale = np.abs(want.astype(np.float64)).max(axis=(0, 1), keepdims=True) + 1e-30
        err = np.abs(got[i].astype(np.float64) - want.astype(np.float64)) / scale
        tol = {2: 2e-3, 4: 1e-6}[got.dtype.itemsize]               # storage rounding of values near zero (relative to the channel's range)
        assert err.max() < tol, (i, err.max())


def test_costmap_ba_on_upsampled_maps_matches_oracle(ctx):
    """Cost maps extracted with upsampling_factor = 2 carry that factor (SetUpsamplingFactor, costmap_extractor.h:399);
    the cost-map residual then interpolates at u = (x sx - 0.5 - x0) * 2 (featurepatch.h:250-255).  Residuals, Jacobians,
    cost and a short solve against the oracle on the same maps."""
    import pxo
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, arena, ba = _setup(ctx, n_cams=5, n_points=40, obs_per_point=3, seed=93, dtype=np.float16, patch_size=8)
    cm = ba.extract_costmaps(make_loss("trivial", []), upsampling_factor=2.0, cfg=interp_cfg())
    assert cm.H == 16 and cm.upsampling_factor == 2.0
    cba = ba.costmap_problem(cm)
    rec, r, gx, gy = cba.eval(interp_cfg(l2_normalize=False), with_jacobian=True, materialize=True)
    P = cba.projection_jacobian().download()
    oprob = _costmap_problem(prob, cm.download()[0])
    oprob["upsampling"] = 2.0
    cost_o, r_o, J_o = pxo.ba_eval_batch(oprob, pxo.cfg(l2_normalize=False), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert rel(r.download(), r_o) < 1e-10
    J = gx.download()[:, :, None] * P[:, None, 0, :] + gy.download()[:, :, None] * P[:, None, 1, :]
    assert rel(J, J_o) < 1e-10
    assert abs(cba.cost(make_loss("cauchy", [0.25])) - cost_o) < 1e-10 * abs(cost_o)
    n_img = 5
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    gauge = (pose_const, tmask, np.full(n_img, 0b1111, np.uint16), np.zeros(40, np.uint8))
    for inner in (False, True):
        for name in ("qvec", "tvec", "xyz"):
            ba.d[name].upload(prob[name])
        s = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge,
                      options=lm_options(max_iterations=4, use_inner_iterations=inner))
        so, qo, to, ko, Xo = pxo.ba_solve(oprob, pxo.cfg(l2_normalize=False), pxo.loss("cauchy", 0.25), *gauge,
                                          pxo.lm_options(max_iterations=4, use_inner_iterations=int(inner)))
        q, t, k, X = cba.params()
        assert s["iterations"] == so["iterations"] and s["num_successful"] == so["num_successful"]
        assert abs(s["final_cost"] - so["final_cost"]) < 1e-6 * so["initial_cost"]
        assert np.abs(X - Xo).max() < 1e-6 and np.abs(q - qo).max() < 1e-6
