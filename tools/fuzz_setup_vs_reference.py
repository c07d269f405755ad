"""Randomised sweep of the problem CONSTRUCTION against the reference's own set-up code compiled in place (build container only:
needs oracle/_ref/libpxo_ref_ba_setup.so and libpxo_ref_ka_setup.so): random scenes / match graphs and option sets through
the product's api.bundle_adjustment._FlatBA, pxr_ka_build_edges and api.keypoint_adjustment.node_roles, compared exactly like
tests/test_ba_setup_golden.py and tests/test_ka_setup_golden.py do for the committed vectors.  NOT part of the test suite.
python tools/fuzz_setup_vs_reference.py [n] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("pixel-perfect-sfm_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import test_ba_setup_golden as tb
import test_ka_setup_golden as tk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
gb, gk = tb._gen(), tk._gen()


def ba_case(k):
    n_images = int(rng.integers(3, 14))
    n_cameras = int(rng.integers(1, min(5, n_images) + 1))
    n_points = int(rng.integers(5, 60))
    image_camera, ptr, p3 = gb._scene(rng, n_images, n_cameras, n_points, int(rng.integers(1, 8)), orphan=float(rng.uniform(0, 0.4)))
    in_problem = rng.random(n_images) < rng.uniform(0.3, 1.0)
    in_problem[rng.choice(n_images, 2, replace=False)] = True
    const_pose = in_problem & (rng.random(n_images) < 0.3)
    tvec_mask = np.where(in_problem & ~const_pose & (rng.random(n_images) < 0.3), rng.integers(1, 8, n_images), 0).astype(np.uint8)
    var_point = rng.random(n_points) < rng.choice([0.0, 0.3, 0.8])
    const_point = ~var_point & (rng.random(n_points) < rng.choice([0.0, 0.2]))
    return dict(name="fz%d" % k, image_camera=image_camera, p2d_ptr=ptr, p2d_point3D=p3, cam_model=rng.integers(0, 5, n_cameras).astype(np.int32),
                n_points=n_points, in_problem=in_problem, const_pose=const_pose, tvec_mask=tvec_mask, var_point=var_point,
                const_point=const_point, const_camera=rng.random(n_cameras) < 0.2, refine_focal=bool(rng.integers(2)),
                refine_pp=bool(rng.integers(2)), refine_extra=bool(rng.integers(2)), refine_extrinsics=bool(rng.integers(5) != 0),
                min_track_length=int(rng.choice([-1, -1, 2, 3, 4])), use_inner=bool(rng.integers(2)))


def check_ba(c):
    """the comparison of tests/test_ba_setup_golden.py on one live case; returns an error string or None"""
    from pixsfm_amd.api.bundle_adjustment import linear_solver_for
    # scenes the reference itself cannot run (SetParameterBlockConstant on a block that is not in the problem) are skipped
    try:
        g = gb.run_reference(c)
    except AssertionError:
        return "SKIP"
    rec, setup, flat = tb._product(c)
    if sorted(flat.obs_keys) != sorted(zip(g["blk_image"].tolist(), g["blk_p2d"].tolist())):
        return "observations"
    const_only = {}
    for i, cp in zip(g["blk_image"].tolist(), g["blk_const_pose"].tolist()):
        const_only[i] = const_only.get(i, True) and bool(cp)
    for k_, i in enumerate(flat.image_ids):
        role = int(g["image_role"][i])
        if role == 2:
            if flat.pose_const[k_] != 0 or flat.tvec_mask[k_] != g["tvec_mask_out"][i]:
                return "pose of image %d" % i
        elif not (flat.pose_const[k_] == 1 and (role == 1 or const_only[i])):
            return "constant pose of image %d" % i
    if sorted(flat.camera_ids) != np.flatnonzero(g["camera_mask"] >= 0).tolist():
        return "camera set"
    for k_, cam in enumerate(flat.camera_ids):
        if int(flat.cam_mask[k_]) != int(g["camera_mask"][cam]):
            return "camera %d mask %d vs %d" % (cam, int(flat.cam_mask[k_]), int(g["camera_mask"][cam]))
    if sorted(flat.point_ids) != np.flatnonzero(g["point_role"] >= 0).tolist():
        return "point set"
    for k_, pid in enumerate(flat.point_ids):
        if int(flat.point_const[k_]) != int(g["point_role"][pid]):
            return "point %d" % pid
    if len(g["blk_image"]) and tb.CERES_SOLVER[linear_solver_for(setup.num_images())] != int(g["solver"][0]):
        return "solver"
    return None


bad = skipped = 0
for k in range(n):
    c = ba_case(k)
    e = check_ba(c)
    if e == "SKIP":
        skipped += 1
        continue
    if e:
        bad += 1
        print("BA set-up case %d: %s  (%s)" % (k, e, {a: c[a] for a in ("min_track_length", "refine_extrinsics", "use_inner")}))
print("BA set-ups %d (%d the reference itself rejects), mismatches %d" % (n, skipped, bad))

# ---- KA set-ups on fresh random graphs ----
graphs = gk._graphs()
bad = 0
for k in range(n // 4):
    n_img, per = int(rng.integers(3, 9)), int(rng.choice([5, 12, 30]))
    pairs, mm = [], []
    for a in range(n_img):
        for b in range(a + 1, n_img):
            if rng.random() < 0.2:
                continue
            m = int(rng.integers(1, 2 * per))
            pairs.append((a, b))
            mm.append((np.stack([rng.integers(0, per, m), rng.integers(0, per, m)], 1).astype(np.int64), np.round(rng.uniform(0.2, 1.0, m), int(rng.choice([1, 3, 6])))))
    if not pairs:
        continue
    pairs = np.array(pairs, np.int32)
    ref = graphs.run_reference(pairs, mm)
    nn = len(ref["node_image"])
    n_images = int(pairs.max()) + 1
    n_feat = np.zeros(n_images, np.int64)
    for (a, b), (m_, _) in zip(pairs, mm):
        n_feat[a] = max(n_feat[a], m_[:, 0].max() + 1); n_feat[b] = max(n_feat[b], m_[:, 1].max() + 1)
    kp_ptr = np.concatenate([[0], np.cumsum(n_feat)]).astype(np.int64)
    kp = rng.uniform(40, 900, (int(kp_ptr[-1]), 2))
    scale = np.tile(rng.uniform(0.25, 1.0, 2) if rng.integers(2) else np.ones(2), (nn, 1))
    node_kp = kp[kp_ptr[ref["node_image"]] + ref["node_feature"]]
    corner = (np.floor(node_kp * scale - 8.0) + rng.integers(-5, 6, (nn, 2))).astype(np.int32)
    mode = int(rng.integers(3))
    sub = None if mode == 0 else np.flatnonzero(rng.random(nn) < 0.5).astype(np.int64)
    if sub is not None and len(sub) == 0:
        sub = None
    c = dict(name="fz%d" % k, pairs=pairs, mm=mm, n_images=n_images, kp_ptr=kp_ptr, kp=kp, corner=corner, scale=scale, nodes_in_problem=sub,
             node_image=ref["node_image"], node_feature=ref["node_feature"], labels=ref["labels"], roots=ref["roots"],
             weight_by_sim=bool(rng.integers(2)), root_edges_only=bool(rng.integers(3) == 0), root_regularize_weight=float(rng.choice([-1.0, 0.3])),
             bound=float(rng.choice([4.0, -1.0, 1.5])), const_roots=bool(rng.integers(2)), const_images=np.array([0] if rng.integers(4) == 0 else [], np.int32))
    try:
        tk._check_case(c, gk.run_reference(c))
    except AssertionError as e:
        bad += 1
        print("KA set-up case %d: %r" % (k, str(e)[:200]))
print("KA set-ups %d, mismatches %d" % (n // 4, bad))
