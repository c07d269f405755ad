"""Pure-Python restatement of the reference's bundle-adjustment problem construction (TEST INFRASTRUCTURE ONLY -- never imported
by the product package), on flat scene arrays:
  BundleOptimizer::SetUp                       bundle_adjustment/src/bundle_optimizer.h:138-163
  ::AddImageToProblem / ::AddPointToProblem    :246-314
  ::RegisterPoint3DObservation                 :316-332
  ::ParameterizePoints / Images / Cameras      :336-443
  ::SolveProblem's solver choice               :179-191
  FeatureReferenceBundleOptimizer::AddResiduals  feature_reference_bundle_optimizer.h:90-149
  BundleAdjustmentSetup                        bundle_adjustment_options.h:28-42 (colmap::BundleAdjustmentConfig [upstream COLMAP 3.8]:
                                               Images, VariablePoints, ConstantPoints, constant poses / tvec indices / cameras)
Camera parameter index groups [upstream COLMAP 3.8 camera_models.h]: SIMPLE_PINHOLE f,cx,cy; PINHOLE fx,fy,cx,cy;
SIMPLE_RADIAL f,cx,cy,k; RADIAL f,cx,cy,k1,k2; OPENCV fx,fy,cx,cy,k1,k2,p1,p2.
The product's builder is native (csrc/pxr_ba_setup.cpp); tests compare the two.
PARITY UNPINNED: the reference's only test of this construction (bundle_optimizer_test.cc) compares GEOMETRIC bundle adjustment
with COLMAP's and needs COLMAP + Ceres; nothing of it can be compiled here."""
import numpy as np

FOCAL = {0: [0], 1: [0, 1], 2: [0], 3: [0], 4: [0, 1]}
PRINCIPAL = {0: [1, 2], 1: [2, 3], 2: [1, 2], 3: [1, 2], 4: [2, 3]}
EXTRA = {0: [], 1: [], 2: [3], 3: [3, 4], 4: [4, 5, 6, 7]}
NUM_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8}
DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR = 3, 4, 5          # ceres/types.h LinearSolverType [upstream]
JACOBI, SCHUR_JACOBI = 1, 2                                    # ceres/types.h PreconditionerType [upstream]


def linear_solver(num_images):
    """bundle_optimizer.h:179-191 -> (linear_solver_type, preconditioner_type)."""
    if num_images <= 50:
        return DENSE_SCHUR, JACOBI
    if num_images <= 1000:
        return SPARSE_SCHUR, JACOBI
    return ITERATIVE_SCHUR, SCHUR_JACOBI


def ba_setup(image_camera, p2d_ptr, p2d_point3D, cam_model, n_points, in_problem, const_pose, tvec_mask, var_point, const_point,
             const_camera, refine_focal, refine_pp, refine_extra, refine_extrinsics, min_track_length, use_inner):
    """-> dict(blk_image, blk_p2d, blk_const_pose (canonical order: by image, then point2D), image_role [n_img]: 2 = pose blocks
    with manifold, 1 = pose blocks held constant, 0 = no varying-pose residual (constant-pose functors only, or none);
    tvec_mask_out; camera_mask [n_cam]: -1 = no residual, else bit mask of constant parameters; point_role [n_pt]: -1 = no
    residual, 1 = constant, 0 = variable; inner_group [n_pt]; solver (type, preconditioner))."""
    n_img, n_cam = len(image_camera), len(cam_model)
    p3 = [np.asarray(p2d_point3D[p2d_ptr[i]:p2d_ptr[i + 1]]) for i in range(n_img)]
    tracks = [[] for _ in range(n_points)]                       # Track().Elements(): by image, then point2D index
    for i in range(n_img):
        for j, p in enumerate(p3[i]):
            if p >= 0:
                tracks[p].append((i, j))
    const_camera = np.array(const_camera, bool).copy()           # AddPointToProblem may add to the setup's constant cameras
    blocks = []
    image_num_residuals, camera_num_residuals, reg = {}, {}, {}

    def add_residuals(i, j):                                     # feature_reference_bundle_optimizer.h:90-149
        cp = (not refine_extrinsics) or bool(const_pose[i])
        p = int(p3[i][j])
        if p < 0:
            return 0
        blocks.append((i, j, int(cp)))
        if not cp:
            image_num_residuals[i] = image_num_residuals.get(i, 0) + 1
        reg.setdefault(p, set()).add(tracks[p].index((i, j)))    # RegisterPoint3DObservation :316-332
        cam = int(image_camera[i])
        camera_num_residuals[cam] = camera_num_residuals.get(cam, 0) + 1
        return 1

    for i in np.flatnonzero(in_problem):                         # AddImageToProblem :246-276
        for j, p in enumerate(p3[i]):
            if p < 0:
                continue
            if len(tracks[p]) < min_track_length:
                continue
            add_residuals(int(i), j)
    for group in (var_point, const_point):                       # AddPointToProblem :279-314
        for p in np.flatnonzero(group):
            p = int(p)
            if len(reg.setdefault(p, set())) == len(tracks[p]):  # operator[]: the point gets an (empty) entry either way
                continue
            for (i, j) in tracks[p]:
                if in_problem[i]:
                    continue
                cam = int(image_camera[i])
                if camera_num_residuals.setdefault(cam, 0) == 0:
                    const_camera[cam] = True
                add_residuals(i, j)
    # ParameterizePoints :336-364
    point_role = np.full(n_points, -1, np.int8)
    inner_group = np.zeros(n_points, np.uint8)
    has_block = np.zeros(n_points, bool)
    for (i, j, _) in blocks:
        has_block[p3[i][j]] = True
    for p, idxs in reg.items():
        tl = len(tracks[p])
        mtl = min(min_track_length, tl) if min_track_length > 0 else tl
        if mtl > len(idxs):
            if has_block[p]:
                point_role[p] = 1
        else:
            if has_block[p]:
                point_role[p] = 0
            if use_inner:
                inner_group[p] = 1
    for p in np.flatnonzero(const_point):
        if has_block[p]:
            point_role[p] = 1
    # ParameterizeImages :366-397
    image_role = np.zeros(n_img, np.int8)
    tvec_mask_out = np.zeros(n_img, np.uint8)
    for i, cnt in image_num_residuals.items():
        if cnt <= 0:
            continue
        constant = (not refine_extrinsics) or bool(const_pose[i]) or not in_problem[i]
        if constant:
            image_role[i] = 1
        else:
            image_role[i] = 2
            tvec_mask_out[i] = tvec_mask[i]
    # ParameterizeCameras :399-443
    camera_mask = np.full(n_cam, -1, np.int32)
    all_const = not refine_focal and not refine_pp and not refine_extra
    for cam, cnt in camera_num_residuals.items():
        if cnt <= 0:
            continue
        m = int(cam_model[cam])
        if all_const or const_camera[cam]:
            camera_mask[cam] = (1 << NUM_PARAMS[m]) - 1
            continue
        idx = ([] if refine_focal else FOCAL[m]) + ([] if refine_pp else PRINCIPAL[m]) + ([] if refine_extra else EXTRA[m])
        camera_mask[cam] = sum(1 << k for k in idx)
    blocks.sort()
    return dict(blk_image=np.array([b[0] for b in blocks], np.int32), blk_p2d=np.array([b[1] for b in blocks], np.int32),
                blk_const_pose=np.array([b[2] for b in blocks], np.uint8), image_role=image_role, tvec_mask_out=tvec_mask_out,
                camera_mask=camera_mask, point_role=point_role, inner_group=inner_group,
                solver=np.array(linear_solver(int(np.count_nonzero(in_problem))), np.int32))
