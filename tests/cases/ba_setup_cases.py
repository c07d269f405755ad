"""Seeded scenes for the bundle-adjustment problem construction (SURVEY 8a rows A16, A17 and the solver selection of A18 / N1):
images inside / outside the set-up, constant poses, constant translation components, extra variable / constant points whose
outside observations are added, constant cameras, the refine_* switches, min_track_length, and the 50 / 51 / 1000 / 1001-image
solver thresholds.  Inputs only."""
import numpy as np


def _scene(rng, n_images, n_cameras, n_points, max_track, orphan=0.15):
    image_camera = rng.integers(0, n_cameras, n_images).astype(np.int32)
    image_camera[:n_cameras] = np.arange(n_cameras)                       # every camera is used
    p2d = [[] for _ in range(n_images)]
    for p in range(n_points):
        tl = int(rng.integers(1, max_track + 1))
        for i in rng.choice(n_images, min(tl, n_images), replace=False):
            p2d[i].append(p)
    for i in range(n_images):                                             # keypoints without a 3D point, shuffled in
        p2d[i] += [-1] * int(rng.binomial(len(p2d[i]) + 1, orphan))
        rng.shuffle(p2d[i])
    ptr = np.concatenate([[0], np.cumsum([len(x) for x in p2d])]).astype(np.int64)
    return image_camera, ptr, np.array([v for x in p2d for v in x], np.int64)


def cases():
    rng = np.random.default_rng(602214)
    out = []
    for k in range(28):
        n_images = int(rng.integers(4, 11))
        n_cameras = int(rng.integers(1, min(4, n_images) + 1))
        n_points = int(rng.integers(10, 41))
        image_camera, ptr, p3 = _scene(rng, n_images, n_cameras, n_points, 6)
        in_problem = rng.random(n_images) < (1.0 if k % 4 == 0 else 0.65)
        in_problem[rng.choice(n_images, 2, replace=False)] = True
        const_pose = in_problem & (rng.random(n_images) < 0.25)
        tvec_mask = np.where(in_problem & ~const_pose & (rng.random(n_images) < 0.25), rng.integers(1, 8, n_images), 0).astype(np.uint8)
        var_point = rng.random(n_points) < (0.5 if k % 2 else 0.0)         # extra points: their outside observations are added
        const_point = ~var_point & (rng.random(n_points) < (0.2 if k % 3 == 0 else 0.0))
        const_camera = rng.random(n_cameras) < (0.3 if k % 5 == 1 else 0.0)
        opt = dict(refine_focal=bool(k % 7 != 3), refine_pp=bool(k % 3 == 1), refine_extra=bool(k % 5 != 2),
                   refine_extrinsics=bool(k % 9 != 8), min_track_length=[-1, 2, 3, -1][k % 4], use_inner=bool(k % 2 == 0))
        out.append(dict(name="ba%02d" % k, image_camera=image_camera, p2d_ptr=ptr, p2d_point3D=p3,
                        cam_model=rng.integers(0, 5, n_cameras).astype(np.int32), n_points=n_points, in_problem=in_problem,
                        const_pose=const_pose, tvec_mask=tvec_mask, var_point=var_point, const_point=const_point,
                        const_camera=const_camera, **opt))
    for name, n_images in (("dense50", 50), ("sparse51", 51), ("sparse1000", 1000), ("iter1001", 1001)):   # solver thresholds
        image_camera, ptr, p3 = _scene(rng, n_images, 1, 8, 3, orphan=0.0)
        z = np.zeros
        out.append(dict(name=name, image_camera=image_camera, p2d_ptr=ptr, p2d_point3D=p3, cam_model=np.array([2], np.int32),
                        n_points=8, in_problem=np.ones(n_images, bool), const_pose=z(n_images, bool), tvec_mask=z(n_images, np.uint8),
                        var_point=z(8, bool), const_point=z(8, bool), const_camera=z(1, bool), refine_focal=True, refine_pp=False,
                        refine_extra=True, refine_extrinsics=True, min_track_length=-1, use_inner=True))
    return out
