"""Bundle-adjustment problem construction (SURVEY 8a rows A16, A17 and the solver selection of A18 / N1) on 32 seeded scenes
(tests/cases/ba_setup_cases.py): observations that become residual blocks (incl. those of extra points in images outside the
set-up and the min_track_length filter), constant poses, constant translation components, constant / partially constant cameras
(incl. cameras only seen from outside images), constant points (under-registered tracks, ConstantPoints), the inner-iteration
group, and the linear solver SolveProblem selects at 50 / 51 / 1000 / 1001 images.
Checked: the product's flattening of the scene (api.bundle_adjustment._FlatBA, what pxr_ba_solve is driven with) and its native
builder through the C-ABI (pxr_ba_build_problem) against the oracle's Python restatement of what the reference hands to Ceres
(oracle/pxo_ba_setup.py: BundleOptimizer::SetUp / AddImageToProblem / AddPointToProblem / Parameterize*,
FeatureReferenceBundleOptimizer::AddResiduals, BundleAdjustmentSetup).
PARITY UNPINNED: the reference's own test of this construction (bundle_optimizer_test.cc) needs COLMAP and Ceres; nothing of it
compiles here."""
import numpy as np

from cases import ba_setup_cases as gen_mod

NUM_PARAMS = [3, 4, 4, 5, 8]
CERES_SOLVER = {"DENSE_SCHUR": 3, "SPARSE_SCHUR": 4, "ITERATIVE_SCHUR": 5}      # ceres/types.h [upstream]
_KEYS = ("image_camera", "p2d_ptr", "p2d_point3D", "cam_model", "n_points", "in_problem", "const_pose", "tvec_mask", "var_point",
         "const_point", "const_camera", "refine_focal", "refine_pp", "refine_extra", "refine_extrinsics", "min_track_length", "use_inner")


def _gen():
    return gen_mod


def _oracle(c):
    import pxo_ba_setup
    return pxo_ba_setup.ba_setup(**{k: c[k] for k in _KEYS})


class _AnyPatch:
    """feature view stand-in: every observation has a patch"""
    def has_fpatch(self, image_id, p2d):
        return True

    def fpatch(self, image_id, p2d):
        return None


def _product(c):
    from pixsfm_amd.api import reconstruction as R
    from pixsfm_amd.api.bundle_adjustment import BundleAdjustmentSetup, _FlatBA
    rec = R.Reconstruction()
    for k, m in enumerate(c["cam_model"]):
        rec.add_camera(R.Camera(k, int(m), 1000, 1000, np.ones(NUM_PARAMS[int(m)])))
    for p in range(int(c["n_points"])):
        rec.add_point3D(p, R.Point3D(np.zeros(3)))
    for i in range(len(c["image_camera"])):
        pts = [R.Point2D((0.0, 0.0), int(v)) for v in c["p2d_point3D"][c["p2d_ptr"][i]:c["p2d_ptr"][i + 1]]]
        rec.add_image(R.Image(i, "im%d" % i, int(c["image_camera"][i]), [2.0, 0, 0, 0], np.zeros(3), pts))
        for j, pt in enumerate(pts):
            if pt.has_point3D():
                rec.points3D[pt.point3D_id].track.add_element(i, j)
    setup = BundleAdjustmentSetup()
    for i in np.flatnonzero(c["in_problem"]):
        setup.add_image(int(i))
    for i in np.flatnonzero(c["const_pose"]):
        setup.set_constant_pose(int(i))
    for i in np.flatnonzero(c["tvec_mask"]):
        setup.set_constant_tvec(int(i), [a for a in range(3) if int(c["tvec_mask"][i]) >> a & 1])
    for p in np.flatnonzero(c["var_point"]):
        setup.add_variable_point(int(p))
    for p in np.flatnonzero(c["const_point"]):
        setup.add_constant_point(int(p))
    for k in np.flatnonzero(c["const_camera"]):
        setup.set_constant_camera(int(k))
    opt = dict(min_track_length=c["min_track_length"], refine_extrinsics=c["refine_extrinsics"], refine_focal_length=c["refine_focal"],
               refine_principal_point=c["refine_pp"], refine_extra_params=c["refine_extra"])
    return rec, setup, _FlatBA(rec, setup, _AnyPatch(), opt)


def test_flattened_problem_matches_the_oracles_restatement():
    from pixsfm_amd.api.bundle_adjustment import linear_solver_for
    gen = _gen()
    seen = dict(outside=0, const_pt=0, tvec=0, cam_partial=0, cam_const=0, filtered=0, blocks=0)
    for c in gen.cases():
        g = _oracle(c)
        rec, setup, flat = _product(c)
        name = c["name"]
        # the residual blocks: one per (image, point2D) -- same SET of observations
        want_obs = sorted(zip(g["blk_image"].tolist(), g["blk_p2d"].tolist()))
        assert sorted(flat.obs_keys) == want_obs, name
        seen["blocks"] += len(want_obs)
        seen["filtered"] += int((c["p2d_point3D"] >= 0).sum()) - len(want_obs)
        # poses: 1 = both blocks constant; 2 = quaternion manifold (+ constant translation components); 0 = the image has
        # constant-pose functors only (no pose parameter block exists), which the flat problem must hold constant as well
        const_only = {i: True for i in set(g["blk_image"].tolist())}
        for i, cp in zip(g["blk_image"].tolist(), g["blk_const_pose"].tolist()):
            const_only[i] = const_only[i] and bool(cp)
        for k, i in enumerate(flat.image_ids):
            role = int(g["image_role"][i])
            if role == 2:
                assert flat.pose_const[k] == 0 and flat.tvec_mask[k] == g["tvec_mask_out"][i], (name, i)
                seen["tvec"] += int(g["tvec_mask_out"][i] != 0)
            else:
                assert role == 1 or const_only[i], (name, i)
                assert flat.pose_const[k] == 1, (name, i)
                seen["outside"] += int(not c["in_problem"][i])
        # cameras: bit mask of constant parameters
        assert sorted(flat.camera_ids) == np.flatnonzero(g["camera_mask"] >= 0).tolist(), name
        for k, cam in enumerate(flat.camera_ids):
            assert int(flat.cam_mask[k]) == int(g["camera_mask"][cam]), (name, cam, int(flat.cam_mask[k]), int(g["camera_mask"][cam]))
            full = (1 << NUM_PARAMS[int(c["cam_model"][cam])]) - 1
            seen["cam_const"] += int(g["camera_mask"][cam] == full)
            seen["cam_partial"] += int(0 < g["camera_mask"][cam] < full)
        # points
        assert sorted(flat.point_ids) == np.flatnonzero(g["point_role"] >= 0).tolist(), name
        for k, pid in enumerate(flat.point_ids):
            assert int(flat.point_const[k]) == int(g["point_role"][pid]), (name, pid)
            seen["const_pt"] += int(g["point_role"][pid] == 1)
        # inner iterations: every variable point is in group 0 (what pxr_ba_solve's use_inner_iterations assumes)
        if c["use_inner"]:
            # (a ConstantPoints entry with a fully registered track is first put into the group and then held constant,
            # bundle_optimizer.h:348-363: Ceres drops constant blocks from the ordering)
            assert np.all(g["inner_group"][g["point_role"] == 0] == 1), name
            extra = (g["inner_group"] == 1) & (g["point_role"] != 0)
            assert np.all(c["const_point"][extra]), name
        else:
            assert not g["inner_group"].any(), name
        # the linear solver, by the number of images of the SETUP
        assert CERES_SOLVER[linear_solver_for(setup.num_images())] == int(g["solver"][0]), (name, g["solver"])
        if int(g["solver"][0]) == 5:
            assert int(g["solver"][1]) == 2          # SCHUR_JACOBI
    assert seen["blocks"] > 2000 and seen["filtered"] > 50 and seen["outside"] > 10 and seen["const_pt"] > 30
    assert seen["tvec"] > 3 and seen["cam_partial"] > 10 and seen["cam_const"] > 5, seen


def test_native_builder_through_the_c_abi_matches_the_oracles_restatement():
    """pxr_ba_build_problem on the flat scene arrays directly (no Python scene objects in between)."""
    from pixsfm_amd.api.bundle_adjustment import build_problem
    gen = _gen()
    for c in gen.cases():
        g = _oracle(c)
        n_img, n_pt = len(c["image_camera"]), int(c["n_points"])
        tracks = [[] for _ in range(n_pt)]                     # Track().Elements() order = the shim's: by image, then point2D
        for i in range(n_img):
            for j, p in enumerate(c["p2d_point3D"][c["p2d_ptr"][i]:c["p2d_ptr"][i + 1]]):
                if p >= 0:
                    tracks[p].append((i, j))
        track_ptr = np.concatenate([[0], np.cumsum([len(t) for t in tracks])]).astype(np.int64)
        flat = [e for t in tracks for e in t]
        r = build_problem(c["image_camera"], c["p2d_ptr"], c["p2d_point3D"], c["cam_model"], n_pt, track_ptr,
                          [e[0] for e in flat], [e[1] for e in flat], c["in_problem"], c["const_pose"], c["tvec_mask"], c["var_point"],
                          c["const_point"], c["const_camera"], c["refine_focal"], c["refine_pp"], c["refine_extra"], c["refine_extrinsics"],
                          c["min_track_length"])
        assert sorted(zip(r["obs_image"].tolist(), r["obs_p2d"].tolist())) == sorted(zip(g["blk_image"].tolist(), g["blk_p2d"].tolist())), c["name"]
        assert np.all(np.diff(r["obs_point"]) >= 0)                                       # ordered by point
        assert np.array_equal(r["camera_mask"], g["camera_mask"]), c["name"]
        assert np.array_equal(r["point_role"], g["point_role"]), c["name"]
        for i in np.flatnonzero(r["image_in_problem"]):
            if g["image_role"][i] == 2:
                assert r["pose_const"][i] == 0 and r["tvec_mask"][i] == g["tvec_mask_out"][i], (c["name"], i)
            else:
                assert r["pose_const"][i] == 1, (c["name"], i)
