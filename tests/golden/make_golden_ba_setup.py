"""Generates tests/golden/ba_setup_ref.npz by running the REFERENCE's own bundle-adjustment problem construction --
BundleOptimizer::Run / SetUp / AddImageToProblem / AddPointToProblem / Parameterize{Points,Images,Cameras} / SolveProblem's
solver selection and FeatureReferenceBundleOptimizer::AddResiduals, with BundleAdjustmentSetup -- compiled in place against a
RECORDING ceres::Problem and functional stand-ins for the COLMAP scene classes (oracle/ref_ba_setup_shim.cc ->
oracle/_ref/libpxo_ref_ba_setup.so) on seeded scenes: which observations become residual blocks, which poses / translation
components / camera parameters / points are constant, which points go into the inner-iteration group, which linear solver.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_ba_setup.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_ba_setup.so")


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


def run_reference(c):
    lib = C.CDLL(LIB)
    lib.pxo_ref_ba_setup.restype = C.c_int64
    n_img, n_cam, n_pt = len(c["image_camera"]), len(c["cam_model"]), int(c["n_points"])
    cap = len(c["p2d_point3D"]) + 8
    u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
    p = lambda a: C.c_void_p(a.ctypes.data)
    bi, bp, bc = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.uint8)
    role, tm = np.empty(n_img, np.int8), np.empty(n_img, np.uint8)
    cm, pr, ig, sv = np.empty(n_cam, np.int32), np.empty(n_pt, np.int8), np.empty(n_pt, np.uint8), np.empty(2, np.int32)
    keep = [np.ascontiguousarray(c["image_camera"], np.int32), np.ascontiguousarray(c["p2d_ptr"], np.int64),
            np.ascontiguousarray(c["p2d_point3D"], np.int64), np.ascontiguousarray(c["cam_model"], np.int32), u8(c["in_problem"]),
            u8(c["const_pose"]), u8(c["tvec_mask"]), u8(c["var_point"]), u8(c["const_point"]), u8(c["const_camera"])]
    m = lib.pxo_ref_ba_setup(n_img, p(keep[0]), p(keep[1]), p(keep[2]), n_cam, p(keep[3]), C.c_int64(n_pt), p(keep[4]), p(keep[5]),
                             p(keep[6]), p(keep[7]), p(keep[8]), p(keep[9]), int(c["refine_focal"]), int(c["refine_pp"]),
                             int(c["refine_extra"]), int(c["refine_extrinsics"]), int(c["min_track_length"]), int(c["use_inner"]),
                             C.c_int64(cap), p(bi), p(bp), p(bc), p(role), p(tm), p(cm), p(pr), p(ig), p(sv))
    assert m >= 0, m
    order = np.lexsort((bp[:m], bi[:m]))                       # the set-up walks unordered containers: store a canonical order
    return dict(blk_image=bi[:m][order].copy(), blk_p2d=bp[:m][order].copy(), blk_const_pose=bc[:m][order].copy(), image_role=role,
                tvec_mask_out=tm, camera_mask=cm, point_role=pr, inner_group=ig, solver=sv)


if __name__ == "__main__":
    store = {}
    n_blocks = 0
    for c in cases():
        r = run_reference(c)
        for k, v in r.items():
            store[c["name"] + "_" + k] = v
        n_blocks += len(r["blk_image"])
    path = os.path.join(HERE, "ba_setup_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "set-ups,", n_blocks, "residual blocks")
