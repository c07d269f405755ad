"""Light stand-ins for the pycolmap (v0.4) objects pixsfm's bundle adjustment touches
(colmap::Reconstruction / Image / Camera / Point3D / Track, used at
pixsfm/bundle_adjustment/src/bundle_optimizer.h:247-275).  pycolmap is not installable in this
environment; a real `pycolmap.Reconstruction` exposes the same attribute names, so the flattening
code in bundle_adjustment.py works on either.
"""
import numpy as np

# [upstream COLMAP 3.8 camera_models.h] model id, name, parameter groups (focal, principal point, extra)
CAMERA_MODELS = {
    0: ("SIMPLE_PINHOLE", 3, [0], [1, 2], []),
    1: ("PINHOLE", 4, [0, 1], [2, 3], []),
    2: ("SIMPLE_RADIAL", 4, [0], [1, 2], [3]),
    3: ("RADIAL", 5, [0], [1, 2], [3, 4]),
    4: ("OPENCV", 8, [0, 1], [2, 3], [4, 5, 6, 7]),
    5: ("OPENCV_FISHEYE", 8, [0, 1], [2, 3], [4, 5, 6, 7]),
    6: ("FULL_OPENCV", 12, [0, 1], [2, 3], [4, 5, 6, 7, 8, 9, 10, 11]),
    7: ("FOV", 5, [0, 1], [2, 3], [4]),
    8: ("SIMPLE_RADIAL_FISHEYE", 4, [0], [1, 2], [3]),
    9: ("RADIAL_FISHEYE", 5, [0], [1, 2], [3, 4]),
    10: ("THIN_PRISM_FISHEYE", 12, [0, 1], [2, 3], [4, 5, 6, 7, 8, 9, 10, 11]),
}
CAMERA_MODEL_NAME_TO_ID = {v[0]: k for k, v in CAMERA_MODELS.items()}


class Camera:
    def __init__(self, camera_id, model, width, height, params):
        self.camera_id = int(camera_id)
        self.model_id = CAMERA_MODEL_NAME_TO_ID[model] if isinstance(model, str) else int(model)
        if self.model_id not in CAMERA_MODELS:
            raise ValueError("camera model id %d is not supported by the accelerated path" % self.model_id)
        self.width, self.height = int(width), int(height)
        self.params = np.array(params, dtype=np.float64)
        if len(self.params) != CAMERA_MODELS[self.model_id][1]:
            raise ValueError("wrong number of parameters for %s" % self.model_name)

    @property
    def model_name(self):
        return CAMERA_MODELS[self.model_id][0]

    def focal_length_idxs(self):
        return list(CAMERA_MODELS[self.model_id][2])

    def principal_point_idxs(self):
        return list(CAMERA_MODELS[self.model_id][3])

    def extra_params_idxs(self):
        return list(CAMERA_MODELS[self.model_id][4])


class Point2D:
    __slots__ = ("xy", "point3D_id")
    INVALID = -1

    def __init__(self, xy, point3D_id=-1):
        self.xy = np.array(xy, dtype=np.float64)
        self.point3D_id = int(point3D_id)

    def has_point3D(self):
        return self.point3D_id != self.INVALID


class Image:
    def __init__(self, image_id, name, camera_id, qvec, tvec, points2D=()):
        self.image_id, self.name, self.camera_id = int(image_id), name, int(camera_id)
        self.qvec = np.array(qvec, dtype=np.float64)
        self.tvec = np.array(tvec, dtype=np.float64)
        self.points2D = list(points2D)

    def normalize_qvec(self):
        self.qvec = self.qvec / np.linalg.norm(self.qvec)


class TrackElement:
    __slots__ = ("image_id", "point2D_idx")

    def __init__(self, image_id, point2D_idx):
        self.image_id, self.point2D_idx = int(image_id), int(point2D_idx)


class Track:
    def __init__(self, elements=()):
        self.elements = list(elements)

    def length(self):
        return len(self.elements)

    def add_element(self, image_id, point2D_idx):
        self.elements.append(TrackElement(image_id, point2D_idx))


class Point3D:
    def __init__(self, xyz, track=None):
        self.xyz = np.array(xyz, dtype=np.float64)
        self.track = track if track is not None else Track()


class Reconstruction:
    def __init__(self):
        self.cameras, self.images, self.points3D = {}, {}, {}

    def add_camera(self, cam):
        self.cameras[cam.camera_id] = cam

    def add_image(self, im):
        self.images[im.image_id] = im

    def add_point3D(self, point3D_id, p):
        self.points3D[int(point3D_id)] = p

    def reg_image_ids(self):
        return sorted(self.images.keys())

    def point3D_ids(self):
        return sorted(self.points3D.keys())

    def num_observations(self):
        return sum(p.track.length() for p in self.points3D.values())


def reconstruction_from_flat(problem, keypoint_noise=None):
    """Build a Reconstruction + per-observation patch lookup from the flat synthetic dict of
    pixsfm_amd.synthetic.make_ba_problem (test/demo helper).  Returns (reconstruction, patches) where
    patches[(image_id, point2D_idx)] is the index of the observation's patch in problem['patches']."""
    rec = Reconstruction()
    for c in range(len(problem["cam_model"])):
        m = int(problem["cam_model"][c])
        rec.add_camera(Camera(c + 1, m, 1000, 1000, problem["cam_params"][c][:CAMERA_MODELS[m][1]]))
    n_img = len(problem["image_camera"])
    images = [Image(i + 1, "image%04d.jpg" % i, int(problem["image_camera"][i]) + 1, problem["qvec"][i], problem["tvec"][i])
              for i in range(n_img)]
    for p in range(len(problem["xyz"])):
        rec.add_point3D(p + 1, Point3D(problem["xyz"][p]))
    patch_of = {}
    for i in range(len(problem["obs_image"])):
        im = images[int(problem["obs_image"][i])]
        pid = int(problem["obs_point"][i]) + 1
        xy = problem["centers"][i] if "centers" in problem else np.zeros(2)
        im.points2D.append(Point2D(xy, pid))
        rec.points3D[pid].track.add_element(im.image_id, len(im.points2D) - 1)
        patch_of[(im.image_id, len(im.points2D) - 1)] = int(problem["obs_patch"][i])
    for im in images:
        rec.add_image(im)
    return rec, patch_of
