"""Mirror of pixsfm's featuremetric bundle-adjustment surface for the accelerated path:
`BundleAdjuster.create(conf).refine_multilevel(reconstruction, feature_manager)`
(pixsfm/bundle_adjustment/main.py:30-154) and the `_bundle_adjustment` pybind classes it drives
(pixsfm/bundle_adjustment/bindings.cc:21-141): BundleAdjustmentSetup, ReferenceExtractor,
FeatureReferenceBundleOptimizer.  Same names, argument meaning and defaults; problem
construction follows BundleOptimizer::SetUp / Parameterize (bundle_optimizer.h:139-165,247-453);
the Ceres solve is replaced by pxr_ba_solve on the GPU.

`use_inner_iterations` (default True, bundle_adjustment/main.py:43) is honoured: every variable point is
re-optimised on its own after each trust-region step (pxr_ba_inner.hip).
"""
import ctypes as C
from copy import deepcopy

import numpy as np

from .. import _lib
from ..engine import BAProblem, lm_options, make_loss
from . import base, features
from .keypoint_adjustment import Summary, default_context
from .reconstruction import CAMERA_MODELS


def default_problem_setup(reconstruction):                    # main.py:12-18
    reg_image_ids = reconstruction.reg_image_ids()
    ba_setup = BundleAdjustmentSetup()
    ba_setup.add_images(set(reg_image_ids))
    ba_setup.set_constant_pose(reg_image_ids[0])
    ba_setup.set_constant_tvec(reg_image_ids[1], [0])
    return ba_setup


def find_problem_labels(reconstruction, max_tracks_per_problem):   # main.py:21-27
    problem_labels = [-1 for _ in range(max(reconstruction.point3D_ids()) + 1)]
    for p3D_id in reconstruction.point3D_ids():
        problem_labels[p3D_id] = int(p3D_id // max_tracks_per_problem)
    return problem_labels


class BundleAdjustmentSetup:
    """colmap::BundleAdjustmentConfig [upstream] + pixsfm's overrides
    (bundle_adjustment/src/bundle_adjustment_options.h:28-42)."""

    def __init__(self):
        self.image_ids = set()
        self.constant_poses = set()
        self.constant_tvecs = {}
        self.constant_cameras = set()
        self.variable_points = set()
        self.constant_points = set()

    def add_image(self, image_id):
        self.image_ids.add(int(image_id))

    def add_images(self, image_ids):
        for i in image_ids:
            self.add_image(i)

    def has_image(self, image_id):
        return int(image_id) in self.image_ids

    def num_images(self):
        return len(self.image_ids)

    def set_constant_pose(self, image_id):
        if not self.has_image(image_id):
            raise ValueError("image %d is not part of the problem" % image_id)
        if int(image_id) in self.constant_tvecs:
            raise ValueError("image %d already has a constant tvec subset" % image_id)
        self.constant_poses.add(int(image_id))

    def has_constant_pose(self, image_id):
        return int(image_id) in self.constant_poses

    def set_constant_tvec(self, image_id, idxs):
        idxs = [int(i) for i in idxs]
        if not (0 < len(idxs) <= 3) or len(set(idxs)) != len(idxs) or any(i < 0 or i > 2 for i in idxs):
            raise ValueError("constant tvec indices must be a non-empty duplicate-free subset of {0,1,2}")
        if not self.has_image(image_id):
            raise ValueError("image %d is not part of the problem" % image_id)
        if self.has_constant_pose(image_id):
            raise ValueError("image %d already has a constant pose" % image_id)
        self.constant_tvecs[int(image_id)] = idxs

    def has_constant_tvec(self, image_id):
        return int(image_id) in self.constant_tvecs

    def constant_tvec(self, image_id):
        return self.constant_tvecs[int(image_id)]

    def set_constant_camera(self, camera_id):
        self.constant_cameras.add(int(camera_id))

    def is_constant_camera(self, camera_id):
        return int(camera_id) in self.constant_cameras

    def add_variable_point(self, point3D_id):
        if int(point3D_id) in self.constant_points:
            raise ValueError("point %d is already constant" % point3D_id)
        self.variable_points.add(int(point3D_id))

    def add_constant_point(self, point3D_id):
        if int(point3D_id) in self.variable_points:
            raise ValueError("point %d is already variable" % point3D_id)
        self.constant_points.add(int(point3D_id))


class FeatureView:
    """features.FeatureView(feature_set, reconstruction) (main.py:128-131): resolves the patch of
    (image_id, point2D_idx) through the image name, like featureview.cc:192-195."""

    def __init__(self, feature_set, reconstruction):
        self.feature_set, self.reconstruction = feature_set, reconstruction

    def has_fpatch(self, image_id, point2D_idx):
        name = self.reconstruction.images[image_id].name
        return self.feature_set.has_fmap(name) and self.feature_set.fmap(name).has_fpatch(point2D_idx)

    def fpatch(self, image_id, point2D_idx):
        return self.feature_set.fmap(self.reconstruction.images[image_id].name).fpatch(point2D_idx)

    @property
    def channels(self):
        return self.feature_set.channels

    # -- the rest of the pybind surface (features/bindings.cc:163-190) --
    def mapping(self):                           # image id -> image name of the images with 3D points (featureview.cc:66-75)
        rec = self.reconstruction
        return {i: rec.images[i].name for i in rec.reg_image_ids() if any(p.has_point3D() for p in rec.images[i].points2D)}

    def find_image_id(self, image_name):         # featureview.cc:316-323
        for i, name in self.mapping().items():
            if name == image_name:
                return i
        raise ValueError("image_name not found")

    def fmap(self, image):                       # by image id or by name
        return self.feature_set.fmap(image if isinstance(image, str) else self.reconstruction.images[image].name)

    def reserved_memory(self):                   # featureview.cc:294-313: the patches this view requires
        total = 0
        for i, name in self.mapping().items():
            if not self.feature_set.has_fmap(name):
                continue
            fm = self.feature_set.fmap(name)
            if fm.is_sparse:
                total += sum(fm.fpatch(k).num_bytes() for k, p in enumerate(self.reconstruction.images[i].points2D)
                             if p.has_point3D() and fm.has_fpatch(k))
            else:
                total += fm.num_bytes()
        return total


def linear_solver_for(num_images):
    """SolveProblem's choice (bundle_optimizer.h:180-191), by the number of images OF THE SETUP (images that enter through
    AddPointToProblem do not count): DENSE_SCHUR up to 50, SPARSE_SCHUR up to 1000, ITERATIVE_SCHUR + SCHUR_JACOBI above.
    The two direct variants are the same algebra here (Schur complement + Cholesky)."""
    if num_images <= 50:
        return "DENSE_SCHUR"
    if num_images <= 1000:
        return "SPARSE_SCHUR"
    return "ITERATIVE_SCHUR"


def build_problem(image_camera, p2d_ptr, p2d_point3D, cam_model, n_points, track_ptr, track_image, track_p2d, in_setup, const_pose,
                  tvec_mask, variable_point, constant_point, constant_camera, refine_focal_length=True, refine_principal_point=False,
                  refine_extra_params=True, refine_extrinsics=True, min_track_length=-1, has_patch=None, skip_missing_patches=False):
    """BundleOptimizer::SetUp + Parameterize* on a flat scene -- native host code (pxr_ba_build_problem, csrc/pxr_ba_setup.cpp;
    arrays as documented in include/pixsfm_hip.h).  Returns a dict: obs_image / obs_p2d / obs_point (scene ids, ordered by point
    and inside a point like its track), image_in_problem, pose_const, tvec_mask, camera_mask (-1 = not in the problem),
    point_role (-1 / 0 variable / 1 constant)."""
    lib = _lib.load()
    a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
    image_camera, p2d_ptr, p2d_point3D = a(image_camera, np.int32), a(p2d_ptr, np.int64), a(p2d_point3D, np.int64)
    cam_model, track_ptr, track_image, track_p2d = a(cam_model, np.int32), a(track_ptr, np.int64), a(track_image, np.int32), a(track_p2d, np.int32)
    in_setup, const_pose, tvec_mask = a(in_setup, np.uint8), a(const_pose, np.uint8), a(tvec_mask, np.uint8)
    variable_point, constant_point, constant_camera = a(variable_point, np.uint8), a(constant_point, np.uint8), a(constant_camera, np.uint8)
    hp = None if has_patch is None else a(has_patch, np.uint8)
    n_img, n_cam, n_p2d = len(image_camera), len(cam_model), len(p2d_point3D)
    o_img, o_p2d, o_pt = np.empty(n_p2d, np.int32), np.empty(n_p2d, np.int32), np.empty(n_p2d, np.int64)
    in_prob, pose_c, tm = np.empty(n_img, np.uint8), np.empty(n_img, np.uint8), np.empty(n_img, np.uint8)
    cmask, role = np.empty(n_cam, np.int32), np.empty(int(n_points), np.int8)
    n = C.c_int64()
    p = lambda x: None if x is None else x.ctypes.data
    _lib.check(lib.pxr_ba_build_problem(n_img, p(image_camera), p(p2d_ptr), p(p2d_point3D), n_cam, p(cam_model), int(n_points), p(track_ptr),
                                        p(track_image), p(track_p2d), p(hp), p(in_setup), p(const_pose), p(tvec_mask), p(variable_point),
                                        p(constant_point), p(constant_camera), int(bool(refine_focal_length)), int(bool(refine_principal_point)),
                                        int(bool(refine_extra_params)), int(bool(refine_extrinsics)), int(min_track_length),
                                        int(bool(skip_missing_patches)), C.byref(n), p(o_img), p(o_p2d), p(o_pt), p(in_prob), p(pose_c), p(tm),
                                        p(cmask), p(role)), "pxr_ba_build_problem")
    k = n.value
    return dict(obs_image=o_img[:k], obs_p2d=o_p2d[:k], obs_point=o_pt[:k], image_in_problem=in_prob, pose_const=pose_c, tvec_mask=tm,
                camera_mask=cmask, point_role=role)


_HOST = []


def _host_module():
    """pixsfm_amd._pxr_host, the compiled (pybind11) scene dump, or None where it was not built (csrc/Makefile target `host`)."""
    if not _HOST:
        try:
            from .. import _pxr_host
            _HOST.append(_pxr_host)
        except ImportError:
            _HOST.append(None)
    return _HOST[0]


class _SceneDump:
    """The Python objects of a scene read ONCE into flat arrays: images / cameras / points in ascending id, every point2D's
    point index, the tracks, and which (image, point2D) has a feature patch.  Independent of the BundleAdjustmentSetup and
    of the extractor / optimiser role, so BundleAdjuster.refine hands the same dump to the reference extraction and to the
    optimiser (the dump is the per-observation Python work of the drop-in path; pxr_ba_build_problem is native)."""

    def __init__(self, reconstruction, feature_view, use_compiled=True):
        rec = reconstruction
        self.reconstruction, self.feature_view = rec, feature_view
        self.img_ids, self.cam_ids, self.pt_ids = sorted(rec.images), sorted(rec.cameras), sorted(rec.points3D)
        img_ids, pt_ids = self.img_ids, self.pt_ids
        self.img_of = {i: k for k, i in enumerate(img_ids)}
        self.cam_of = {c: k for k, c in enumerate(self.cam_ids)}
        self.pt_of = pt_of = {p: k for k, p in enumerate(pt_ids)}
        images = [rec.images[i] for i in img_ids]
        self.image_camera = np.array([self.cam_of[im.camera_id] for im in images], np.int32)
        self.counts = counts = [len(im.points2D) for im in images]
        self.p2d_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        host = _host_module() if use_compiled else None
        self.compiled = host is not None
        if host is not None:         # the per-observation walk in C++ (csrc/pybind/pxr_host.cpp)
            points = [rec.points3D[p] for p in pt_ids]
            p2d_ptr, ids, self.track_ptr, self.track_image, self.track_p2d = host.scene_arrays(images, points, self.img_of)
            assert np.array_equal(p2d_ptr, self.p2d_ptr)
        else:
            try:    # the attribute alone (-1 = no 3D point here); pycolmap marks it with 2^64 - 1, which does not fit: ask has_point3D()
                ids = np.fromiter((q.point3D_id for im in images for q in im.points2D), dtype=np.int64, count=int(self.p2d_ptr[-1]))
            except OverflowError:
                ids = np.fromiter((q.point3D_id if q.has_point3D() else -1 for im in images for q in im.points2D), dtype=np.int64,
                                  count=int(self.p2d_ptr[-1]))
            ids = np.where(ids < 0, -1, ids)
            points = [rec.points3D[p] for p in pt_ids]
            tl = [pt.track.length() for pt in points]
            self.track_ptr = np.concatenate([[0], np.cumsum(tl)]).astype(np.int64)
            n_el = int(self.track_ptr[-1])
            img_of = self.img_of
            self.track_image = np.fromiter((img_of[e.image_id] for pt in points for e in pt.track.elements), dtype=np.int32, count=n_el)
            self.track_p2d = np.fromiter((e.point2D_idx for pt in points for e in pt.track.elements), dtype=np.int32, count=n_el)
        # point3D id -> index in ascending-id order: a lookup table when the ids are dense (COLMAP numbers points from 1),
        # else a binary search in the sorted id list (ids are arbitrary integers)
        pt_arr = np.asarray(pt_ids, dtype=np.int64)
        if len(pt_arr) and 0 <= pt_arr[0] and pt_arr[-1] < 4 * len(pt_arr) + 1024:
            lut = np.full(int(pt_arr[-1]) + 2, -1, np.int64)
            lut[pt_arr] = np.arange(len(pt_arr), dtype=np.int64)
            pos = lut[np.clip(ids, 0, len(lut) - 1)]
            known = (ids >= 0) & (ids < len(lut) - 1) & (pos >= 0)
            pos = np.maximum(pos, 0)
        else:
            pos = np.searchsorted(pt_arr, np.maximum(ids, 0)) if len(pt_arr) else np.zeros(len(ids), np.int64)
            pos = np.minimum(pos, max(len(pt_arr) - 1, 0))
            known = (ids >= 0) & (pt_arr[pos] == ids) if len(pt_arr) else np.zeros(len(ids), bool)
        if ((ids >= 0) & ~known).any():
            raise KeyError(int(ids[(ids >= 0) & ~known][0]))           # a point2D refers to a point3D the reconstruction lacks
        self.p2d_point3D = np.where(known, pos, -1).astype(np.int64)
        self.cam_model = np.array([rec.cameras[c].model_id for c in self.cam_ids], np.int32)
        # which observations have a feature patch -- per image one set lookup over its keypoint ids, not one call per point2D
        fs = getattr(feature_view, "feature_set", None)
        self.has_patch = np.zeros(len(ids), np.uint8)
        self._patch_dicts = []
        if fs is None:                           # any object with has_fpatch / fpatch (image_id, point2D_idx): asked one by one
            self._patch_dicts = None
            for k, i in enumerate(img_ids):
                for j in range(counts[k]):
                    at = int(self.p2d_ptr[k]) + j
                    self.has_patch[at] = 1 if (self.p2d_point3D[at] < 0 or feature_view.has_fpatch(i, j)) else 0
            images = []
        for k, im in enumerate(images):
            self._patch_dicts.append(fs.fmap(im.name) if fs.has_fmap(im.name) else None)
        flags_done = False
        if host is not None and hasattr(host, "patch_flags") and images:   # the sparse maps' keys in one C++ walk per image
            try:
                self.has_patch = host.patch_flags([fm.patches if (fm is not None and fm.is_sparse) else None
                                                   for fm in self._patch_dicts], self.p2d_ptr)
                flags_done = True
            except TypeError:                    # `patches` is not a plain dict: the numpy form below
                self.has_patch = np.zeros(len(ids), np.uint8)
        for k, fm in enumerate(self._patch_dicts if images else []):
            lo, hi = int(self.p2d_ptr[k]), int(self.p2d_ptr[k + 1])
            if fm is None or hi == lo:
                continue
            if fm.is_sparse:
                if not flags_done:
                    keys = np.fromiter(fm.patches.keys(), dtype=np.int64, count=len(fm.patches))
                    self.has_patch[lo:hi] = np.isin(np.arange(hi - lo, dtype=np.int64), keys)
            else:
                self.has_patch[lo:hi] = 1 if fm.has_fpatch(0) else 0
        self.has_patch[self.p2d_point3D < 0] = 1                        # never asked (only observations of 3D points are)

    def patches_of(self, obs_image_idx, obs_p2d):
        """The FeaturePatch / ArenaPatch of each (image index, point2D index), image by image (one dict per image)."""
        if self._patch_dicts is None:
            return [self.feature_view.fpatch(self.img_ids[a], b) for a, b in zip(obs_image_idx.tolist(), obs_p2d.tolist())]
        if self.compiled:
            try:
                # a dense map without its patch (an image no observation refers to) stands as None like a missing map
                dicts = [None if fm is None else (fm.patches if fm.is_sparse else (fm.fpatch(0) if fm.has_fpatch(0) else None))
                         for fm in self._patch_dicts]
                dense = [fm is not None and not fm.is_sparse for fm in self._patch_dicts]
                return _host_module().patches_of(dicts, dense, np.ascontiguousarray(obs_image_idx, dtype=np.int32),
                                                 np.ascontiguousarray(obs_p2d, dtype=np.int32))[0]
            except (KeyError, TypeError):
                pass                     # the pure-Python walk below names the missing image / patch
        out = np.empty(len(obs_image_idx), dtype=object)
        order = np.argsort(obs_image_idx, kind="stable")
        bounds = np.searchsorted(obs_image_idx[order], np.arange(len(self.img_ids) + 1))
        for k in range(len(self.img_ids)):
            sel = order[bounds[k]:bounds[k + 1]]
            if len(sel) == 0:
                continue
            fm = self._patch_dicts[k]
            if fm is None:
                raise KeyError(self.reconstruction.images[self.img_ids[k]].name)
            if fm.is_sparse:
                pd = fm.patches
                out[sel] = [pd[j] for j in obs_p2d[sel].tolist()]
            else:
                out[sel] = fm.fpatch(0)
        return out.tolist()


class _FlatBA:
    """Flat arrays of the residual blocks BundleOptimizer::SetUp would add (bundle_optimizer.h:139-165) and of the
    parameterisation (:335-453).  The walk over the scene is native host code (build_problem -> pxr_ba_build_problem); this
    class dumps the pycolmap-style objects into the flat scene arrays it takes and compacts its answer into the arrays of
    pxr_ba_view (images / cameras / points that take part, in ascending id)."""

    def __init__(self, reconstruction, setup, feature_view, options=None, point_filter=None, extractor=False, scene=None):
        """extractor=True: the read-only use by ReferenceExtractor / CostMapExtractor -- the reconstruction is const
        there (no NormalizeQvec; the rotation normalises q itself) and an observation without a patch is skipped like
        GetVisibleObservations does (reference_extractor.h:171-213).  extractor=False: the optimiser's SetUp, where a
        missing patch is an error (feature_view.GetFeaturePatch / references.at throw,
        feature_reference_bundle_optimizer.h:100-108).  point_filter: only these points' observations are walked.
        scene: a _SceneDump of (reconstruction, feature_view) made earlier (BundleAdjuster.refine shares one)."""
        from ._timing import phase
        self._phase = phase
        with phase("dump"):
            if scene is None or scene.reconstruction is not reconstruction or scene.feature_view is not feature_view:
                scene = _SceneDump(reconstruction, feature_view)
            self._dump(reconstruction, setup, scene, options, point_filter, extractor)

    def _dump(self, reconstruction, setup, scene, options, point_filter, extractor):
        phase = self._phase
        rec = reconstruction
        opt = options or {}
        img_ids, cam_ids, pt_ids, pt_of = scene.img_ids, scene.cam_ids, scene.pt_ids, scene.pt_of
        image_camera, p2d_ptr, cam_model = scene.image_camera, scene.p2d_ptr, scene.cam_model
        track_ptr, track_image, track_p2d = scene.track_ptr, scene.track_image, scene.track_p2d
        p2d_point3D = scene.p2d_point3D
        if point_filter is not None:
            keep = np.zeros(len(pt_ids) + 1, bool)
            keep[[pt_of[p] for p in point_filter if p in pt_of]] = True
            p2d_point3D = np.where(keep[p2d_point3D], p2d_point3D, -1)      # (index -1 reads the spare False slot)
        n_img, n_cam, n_pt = len(img_ids), len(cam_ids), len(pt_ids)
        in_setup = np.array([setup.has_image(i) for i in img_ids], np.uint8)
        const_pose = np.array([setup.has_constant_pose(i) for i in img_ids], np.uint8)
        tvm = np.array([sum(1 << a for a in setup.constant_tvec(i)) if setup.has_constant_tvec(i) else 0 for i in img_ids], np.uint8)
        var_pt = np.zeros(n_pt, np.uint8); const_pt = np.zeros(n_pt, np.uint8)
        for p in setup.variable_points:
            var_pt[pt_of[p]] = 1
        for p in setup.constant_points:
            const_pt[pt_of[p]] = 1
        const_cam = np.array([setup.is_constant_camera(c) for c in cam_ids], np.uint8)
        # which observations have a feature patch (only those that can enter the problem are asked)
        has_patch = np.where(p2d_point3D >= 0, scene.has_patch, 1).astype(np.uint8)
        try:
            with phase("build_problem"):         # (nested in "dump": subtracted by the report)
                r = build_problem(image_camera, p2d_ptr, p2d_point3D, cam_model, n_pt, track_ptr, track_image, track_p2d, in_setup, const_pose, tvm,
                                  var_pt, const_pt, const_cam, opt.get('refine_focal_length', True), opt.get('refine_principal_point', False),
                                  opt.get('refine_extra_params', True), opt.get('refine_extrinsics', True), opt.get('min_track_length', -1),
                                  has_patch, skip_missing_patches=extractor)
        except _lib.PixsfmHipError as e:
            raise ValueError(str(e).replace("pxr_ba_build_problem: ", "")) from None
        if not extractor:                                                # NormalizeQvec in AddImageToProblem, :255
            for i in setup.image_ids:
                im = rec.images[i]
                im.qvec = np.asarray(im.qvec, dtype=np.float64) / np.linalg.norm(im.qvec)
        # ---- compact: the images / cameras / points that take part, in ascending id --------------------------------------------
        used_img = np.flatnonzero(r["image_in_problem"])
        used_cam = np.flatnonzero(r["camera_mask"] >= 0)
        used_pt = np.flatnonzero(r["point_role"] >= 0)
        img_new = np.full(n_img, -1, np.int32); img_new[used_img] = np.arange(len(used_img), dtype=np.int32)
        cam_new = np.full(n_cam, -1, np.int32); cam_new[used_cam] = np.arange(len(used_cam), dtype=np.int32)
        pt_new = np.full(n_pt, -1, np.int32); pt_new[used_pt] = np.arange(len(used_pt), dtype=np.int32)
        self.image_ids = [img_ids[k] for k in used_img]
        self.camera_ids = [cam_ids[k] for k in used_cam]
        self.point_ids = [pt_ids[k] for k in used_pt]
        self.outside_images = {img_ids[k] for k in used_img if not in_setup[k]}
        self.obs_image = img_new[r["obs_image"]]
        self.obs_point = pt_new[r["obs_point"]]
        self._key_image = np.asarray(img_ids, dtype=np.int64)[r["obs_image"]] if len(img_ids) else np.zeros(0, np.int64)
        self._key_p2d = r["obs_p2d"]                 # obs_keys (a list of 1M tuples at configs[2]) is built only when somebody asks
        self._obs_keys = None
        # the patch OBJECT of every observation is looked up only when somebody asks (`patches`): with stacked feature maps uploaded
        # by SharedArena.prefetch the arena slots follow from the stacks' layout (arena_index) and no object is touched
        self._scene, self._scene_obs_image, self._patches = scene, r["obs_image"], None
        n_i, n_c, n_p = len(used_img), len(used_cam), len(used_pt)
        self.image_camera = cam_new[image_camera[used_img]]
        self.qvec = np.array([rec.images[i].qvec for i in self.image_ids], np.float64).reshape(n_i, 4)
        self.tvec = np.array([rec.images[i].tvec for i in self.image_ids], np.float64).reshape(n_i, 3)
        self.cam_model = cam_model[used_cam]
        self.cam_params = np.zeros((n_c, 12))
        for k, c in enumerate(self.camera_ids):
            self.cam_params[k, :len(rec.cameras[c].params)] = rec.cameras[c].params
        self.xyz = np.array([rec.points3D[p].xyz for p in self.point_ids], np.float64).reshape(n_p, 3)
        self.pose_const = r["pose_const"][used_img].astype(np.uint8)
        self.tvec_mask = r["tvec_mask"][used_img].astype(np.uint8)
        self.cam_mask = r["camera_mask"][used_cam].astype(np.uint16)
        self.point_const = r["point_role"][used_pt].astype(np.uint8)

    @property
    def patches(self):
        """The FeaturePatch / ArenaPatch of every residual block (one dict lookup per observation, in C++ where the host binding is built)."""
        if self._patches is None:
            self._patches = self._scene.patches_of(self._scene_obs_image, self._key_p2d)
        return self._patches

    def first_patch(self):
        if self._patches is not None:
            return self._patches[0]
        return self._scene.patches_of(self._scene_obs_image[:1], self._key_p2d[:1])[0]

    def arena_index(self, cache):
        """Arena slot of every residual block from the layout of a prefetched features.SharedArena, or None (-> features.to_arena on
        the patch objects)."""
        if cache is None or not hasattr(cache, "slots"):
            return None
        sc = self._scene
        if getattr(sc, "_image_names", None) is None:
            sc._image_names = [sc.reconstruction.images[i].name for i in sc.img_ids]
        return cache.slots(sc._image_names, self._scene_obs_image, self._key_p2d,
                           feature_set=getattr(sc.feature_view, "feature_set", None))

    @property
    def obs_keys(self):
        """(image_id, point2D_idx) of every residual block."""
        if self._obs_keys is None:
            self._obs_keys = list(zip(self._key_image.tolist(), self._key_p2d.tolist()))
        return self._obs_keys

    def key_arrays(self, obs):
        """(len(obs), 2) int64: image_id, point2D_idx of the given residual blocks -- without the list of tuples."""
        obs = np.asarray(obs, dtype=np.int64)
        return np.stack([self._key_image[obs], self._key_p2d[obs].astype(np.int64)], 1)

    def problem_dict(self, refs, patch_index=None):
        """patch_index: arena patch of each observation (features.to_arena(...).index); default 0 .. n_obs - 1."""
        obs_patch = np.arange(len(self.obs_image), dtype=np.int64) if patch_index is None else patch_index
        return dict(obs_image=self.obs_image, obs_point=self.obs_point,
                    obs_patch=obs_patch, image_camera=self.image_camera,
                    qvec=self.qvec, tvec=self.tvec, cam_model=self.cam_model, cam_params=self.cam_params,
                    xyz=self.xyz, refs=refs)


class ReferenceExtractor:
    """_bundle_adjustment.ReferenceExtractor(ref_conf, interp_conf).run(problem_labels, reconstruction,
    feature_set) -> {point3D_id: Reference} (bindings.cc:28-34,172-177; reference_extractor.h:125-162)."""
    # ReferenceConfig's own defaults (reference_extractor.h:55-67: TEN IRLS iterations); BundleAdjuster.default_conf['references']
    # carries the Python-level ones (iters 100) and passes them in full
    default_conf = {'loss': {'name': 'cauchy', 'params': [0.25]}, 'iters': 10, 'keep_observations': False,
                    'compute_offsets3D': False, 'num_threads': -1}     # (closest_to_robust_mean is not exposed by bindings.cc:69-78: always true)

    def __init__(self, config=None, interpolation_config=None, ctx=None):
        self.config = base.merge_conf(self.default_conf, config)
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx = ctx
        if self.config['compute_offsets3D']:
            raise ValueError("compute_offsets3D is outside the accelerated path (N_NODES = 1: offsets are zero)")

    def run(self, problem_labels, reconstruction, feature_set, arena_cache=None, scene=None):
        """arena_cache: a features.SharedArena -- the patches uploaded here stay on the device for the optimiser that follows;
        scene: a _SceneDump of (reconstruction, FeatureView(feature_set)) shared with it."""
        ctx = self.ctx or default_context()
        wanted = {p for p in reconstruction.point3D_ids() if p < len(problem_labels) and problem_labels[p] >= 0}
        setup = BundleAdjustmentSetup()
        setup.add_images(reconstruction.reg_image_ids())
        flat = _FlatBA(reconstruction, setup, scene.feature_view if scene is not None else FeatureView(feature_set, reconstruction),
                       point_filter=wanted, extractor=True, scene=scene)
        if len(flat.obs_image) == 0:
            return {}
        # several ranks: every rank extracts the references of its share of the points (independent per point,
        # reference_extractor.h:216-237) from its own patches; the small per-point results are gathered
        from .. import parallel
        rank, world = parallel.world()
        part = flat if world == 1 else _rank_share(flat, rank, world)
        out = {}
        if len(part.obs_image):
            from ._timing import phase
            with phase("upload"):
                arena = _arena_of(ctx, part, arena_cache)
            with phase("references"):
                ba = BAProblem(ctx, arena, part.problem_dict(np.zeros((len(part.point_ids), arena.C)), arena.index))
                out = self._references_of(ba, part)
            arena.close()
        return out if world == 1 else parallel.gather_references(out)

    def _references_of(self, ba, flat):
        """References of the flat problem `ba` (its device `refs` are filled in place) as {point3D_id: Reference}."""
        keep = bool(self.config['keep_observations'])
        chosen, mean = ba.compute_references(self.interpolation.to_engine(),
                                             make_loss(self.config['loss']['name'], self.config['loss']['params']),
                                             iters=self.config['iters'], keep_observations=keep, keep_mean=keep)
        refs = ba.d["refs"].download()
        obs_desc = ba.obs_desc.download() if keep else None                 # reference_extractor.h:259-265
        obs_of_point = {}
        if keep:
            for i, k in enumerate(flat.obs_point):
                obs_of_point.setdefault(int(k), []).append(i)
        if not keep:        # the default: ids + source observation + descriptor row per point, objects on demand
            sel = np.flatnonzero(chosen >= 0)
            return features.ReferenceMap([flat.point_ids[k] for k in sel], flat.key_arrays(chosen[sel]), refs[sel])
        out = {}
        for k, pid in enumerate(flat.point_ids):
            if chosen[k] >= 0:
                image_id, p2d_idx = flat.obs_keys[int(chosen[k])]
                out[pid] = features.Reference(image_id, p2d_idx, refs[k],
                                              observations=[obs_desc[i] for i in obs_of_point[k]] if keep else None)
                if keep:   # ReferenceData (reference_extractor.h:256-265): the visible track and each observation's distance to the robust mean
                    out[pid].track = [flat.obs_keys[i] for i in obs_of_point[k]]
                    out[pid].costs = [float(((obs_desc[i] - mean[k]) ** 2).sum()) for i in obs_of_point[k]]
        return out


class CostMapExtractor:
    """_bundle_adjustment.CostMapExtractor(costmap_conf, interp_conf).run(problem_labels, reconstruction, feature_set,
    ref_extractor) -> (cost-map FeatureSet, {point3D_id: Reference}) (bindings.cc:20-26,179-184;
    costmap_extractor.h:118-174): references by IRLS, then per observation the map of the featuremetric error of every
    texel of its patch against the reference of its 3D point.  Here both steps run on the device on ONE flat problem
    and the maps stay there: the returned FeatureSet holds features.ArenaPatch entries of a device arena
    (`.arena` of the set; 3 channels [cost, dcost/dr, dcost/dc] or 1), which CostMapBundleOptimizer consumes in place."""
    default_conf = {'loss': {'name': 'trivial', 'params': []}, 'as_gradientfield': True, 'compute_cross_derivative': False,
                    'upsampling_factor': 1.0, 'apply_sqrt': False, 'dense_cut_size': 12, 'num_threads': -1}

    def __init__(self, config=None, interpolation_config=None, ctx=None, chunk_bytes=16 << 30):
        """chunk_bytes: device memory the FEATURE patches of one pass may take.  A scene whose patches exceed it is
        processed in chunks of whole 3D points in problem-label order (the unit of the reference's RunSubset,
        costmap_extractor.h:177-228): upload the chunk's patches, references, cost maps into the ONE cost-map arena,
        release -- the low-memory point of the strategy (configs/low_memory.yaml)."""
        self.config = base.merge_conf(self.default_conf, config)
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx, self.chunk_bytes = ctx, int(chunk_bytes)
        if float(self.config['upsampling_factor']) <= 0.0:
            raise ValueError("upsampling_factor must be positive")

    def get_effective_channels(self):                                            # costmap_extractor.h:52-61
        if not self.config['as_gradientfield']:
            return 1
        return 4 if self.config['compute_cross_derivative'] else 3

    def run(self, problem_labels, reconstruction, feature_set, ref_extractor):
        from ..engine import PatchArena
        if ref_extractor is None:
            raise ValueError("a ReferenceExtractor is required")                 # Run dereferences it, costmap_extractor.h:141
        ctx = self.ctx or default_context()
        dense = any(not fm.is_sparse for fm in feature_set.fmaps.values())
        if dense and any(fm.is_sparse for fm in feature_set.fmaps.values()):
            raise ValueError("a feature set mixing sparse and dense maps is outside the accelerated path")
        wanted = {p for p in reconstruction.point3D_ids() if p < len(problem_labels) and problem_labels[p] >= 0}
        setup = BundleAdjustmentSetup()
        setup.add_images(reconstruction.reg_image_ids())
        flat = _FlatBA(reconstruction, setup, FeatureView(feature_set, reconstruction), point_filter=wanted, extractor=True)   # host lists only
        cost_fset = features.FeatureSet(channels=self.get_effective_channels())
        n_obs = len(flat.obs_image)
        if n_obs == 0:
            return cost_fset, {}
        first = flat.patches[0]
        H, W, C = first.shape
        if dense:                                      # cost maps of dense_cut_size windows (costmap_extractor.h:50,401-431)
            if isinstance(first, features.ArenaPatch):
                raise ValueError("cost maps of device-resident dense maps are not supported: pass host FeatureMap.dense maps")
            H = W = int(self.config['dense_cut_size'])
            if any(p.shape[0] < H or p.shape[1] < W for p in flat.patches):
                raise ValueError("dense_cut_size exceeds a dense feature map")           # THROW_CHECK_LE, featurepatch.cc:342-343
        dtype = first.arena.dtype if isinstance(first, features.ArenaPatch) else first.data.dtype
        up = float(self.config['upsampling_factor'])
        cross = bool(self.config['compute_cross_derivative'])
        costmaps = PatchArena(ctx, n_obs, int(H * (up + 1.0e-6)), int(W * (up + 1.0e-6)),     # costmap_extractor.h:385-390
                              self.get_effective_channels(), dtype)
        costmaps.upsampling_factor = up
        loss = make_loss(self.config['loss']['name'], self.config['loss']['params'])
        # chunks of whole points
        obs_of_point = [[] for _ in flat.point_ids]
        for i, k in enumerate(flat.obs_point):
            obs_of_point[int(k)].append(i)
        order = sorted(range(len(flat.point_ids)), key=lambda k: (problem_labels[flat.point_ids[k]], flat.point_ids[k]))
        budget = max(1, self.chunk_bytes // (H * W * C * np.dtype(dtype).itemsize))
        chunks, cur, cur_obs = [], [], 0
        for k in order:
            if cur and cur_obs + len(obs_of_point[k]) > budget:
                chunks.append(cur)
                cur, cur_obs = [], 0
            cur.append(k)
            cur_obs += len(obs_of_point[k])
        chunks.append(cur)
        map_index, offset, references = np.empty(n_obs, np.int64), 0, {}
        for pts in chunks:
            sub = _PointSubset(flat, pts, sorted(i for k in pts for i in obs_of_point[k]))
            arena = features.to_arena(ctx, sub.patches)
            ba = BAProblem(ctx, arena, sub.problem_dict(np.zeros((len(pts), C)), arena.index))
            references.update(ref_extractor._references_of(ba, sub))             # on the FULL maps, like the reference
            if dense:
                # the reference slices a dense_cut_size window around the current reprojection of every observation
                # (FeaturePatch::Slice / ToCorner, featurepatch.cc:324-359; costmap_extractor.h:210-222) and fills the
                # cost map from that copy: central differences clamp at the WINDOW border
                xy = ba.eval(self.interpolation.to_engine(), with_jacobian=False)[0].download()[:, 6:8]
                windows = []
                for i, P in enumerate(sub.patches):
                    uv = xy[i] * P.scale - 0.5 - P.corner
                    c = np.trunc(uv - H / 2.0).astype(np.int64)                  # Eigen cast<int>: towards zero
                    x0 = int(min(max(c[0], 0), P.shape[1] - W))
                    y0 = int(min(max(c[1], 0), P.shape[0] - H))
                    windows.append(features.FeaturePatch(P.data[y0:y0 + H, x0:x0 + W], (x0, y0), P.scale))
                refs_host = ba.d["refs"].download()
                arena.close()
                arena = features.to_arena(ctx, windows)
                ba = BAProblem(ctx, arena, sub.problem_dict(refs_host, arena.index))
            ba.extract_costmaps(loss, as_gradientfield=self.config['as_gradientfield'], apply_sqrt=self.config['apply_sqrt'],
                                out=costmaps, first_out=offset, upsampling_factor=up, compute_cross_derivative=cross,
                                cfg=self.interpolation.to_engine())
            arena.close()                                                        # synchronises the stream first
            map_index[sub.obs] = offset + np.arange(len(sub.obs))
            offset += len(sub.obs)
        cost_fset.arena = costmaps
        for i, (image_id, p2d_idx) in enumerate(flat.obs_keys):                 # CreateShallowCostmapFSet, :360-435
            name = reconstruction.images[image_id].name
            if not cost_fset.has_fmap(name):
                cost_fset.fmaps[name] = features.FeatureMap()
            cost_fset.fmaps[name].patches[int(p2d_idx)] = features.ArenaPatch(costmaps, map_index[i])
        return cost_fset, references


def _rank_share(flat, rank, world):
    """This rank's contiguous share of the points of a _FlatBA (balanced by observation count) with all their
    observations -- the partition of SURVEY 8e: points sharded, images and cameras replicated."""
    from ..parallel import balanced_ranges
    counts = np.bincount(flat.obs_point, minlength=len(flat.point_ids))
    lo, hi = balanced_ranges(counts, world)[rank]
    obs = np.nonzero((flat.obs_point >= lo) & (flat.obs_point < hi))[0]
    sub = _PointSubset(flat, list(range(lo, hi)), obs.tolist())
    sub.lo = lo                     # index (in the parent's point order) of the share's first point
    return sub


def _arena_of(ctx, part, cache):
    """The device arena + slot index of the residual blocks of `part` (a _FlatBA or _PointSubset): straight from the layout of a
    prefetched features.SharedArena where there is one, else through the patch objects (features.to_arena)."""
    idx = part.arena_index(cache)
    if idx is not None:
        return features.ArenaRef(cache.arena, idx, owned=False)
    return features.to_arena(ctx, part.patches, cache=cache)


class _PointSubset:
    """The observations `obs` (indices into a _FlatBA) of the points `pts` as a flat problem of their own: images and
    cameras stay those of the parent, points and observations are renumbered."""

    def __init__(self, flat, pts, obs):
        self.flat, self.obs = flat, np.asarray(obs, dtype=np.int64)
        local = {k: j for j, k in enumerate(pts)}
        self.point_ids = [flat.point_ids[k] for k in pts]
        self.obs_keys = [flat.obs_keys[i] for i in obs]
        self.obs_image = flat.obs_image[self.obs]
        self.obs_point = np.array([local[int(flat.obs_point[i])] for i in obs], np.int32)
        self._patches = None
        self.xyz = flat.xyz[np.asarray(pts, dtype=np.int64)]

    @property
    def patches(self):
        if self._patches is None:
            fp = self.flat.patches
            self._patches = [fp[i] for i in self.obs.tolist()]
        return self._patches

    def arena_index(self, cache):
        idx = self.flat.arena_index(cache)
        return None if idx is None else idx[self.obs]

    def key_arrays(self, obs):
        return self.flat.key_arrays(self.obs[np.asarray(obs, dtype=np.int64)])

    def problem_dict(self, refs, patch_index):
        f = self.flat
        return dict(obs_image=self.obs_image, obs_point=self.obs_point, obs_patch=patch_index, image_camera=f.image_camera,
                    qvec=f.qvec, tvec=f.tvec, cam_model=f.cam_model, cam_params=f.cam_params, xyz=self.xyz, refs=refs)


class FeatureReferenceBundleOptimizer:
    """_bundle_adjustment.FeatureReferenceBundleOptimizer(options, setup, interp_conf)
    .run(reconstruction, feature_view, references) (bindings.cc:36-51,137-141)."""
    option_defaults = {
        'loss': {'name': 'cauchy', 'params': [0.25]},
        'solver': {**base.solver_default_conf, 'callbacks': []},
        'print_summary': True,
        'refine_focal_length': True, 'refine_principal_point': False, 'refine_extra_params': True,
        'refine_extrinsics': True, 'min_track_length': -1,
    }

    def __init__(self, options=None, setup=None, interpolation_config=None, ctx=None, allreduce=None):
        self.options = base.merge_conf(self.option_defaults, options)
        if setup is None:
            raise ValueError("a BundleAdjustmentSetup is required")
        self.setup = setup
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx, self.allreduce = ctx, allreduce
        self._summary = None
        self._used = False

    def set_up(self, reconstruction, loss_function, feature_view, references):
        """SetUp (feature_reference_bundle_optimizer.h:73-87, bundle_optimizer.h:113-166): builds the problem --
        here the flat device arrays -- without solving it.  loss_function: {'name', 'params'} (the dict form of
        the reference's ceres::LossFunction argument) or None for the optimizer's own loss."""
        if reconstruction is None:
            raise ValueError("reconstruction cannot be NULL.")                  # bundle_optimizer.h:117-118
        if self._used:
            raise ValueError("Cannot use the same BundleOptimizer multiple times")   # :120-121
        self._used = True
        ctx = self.ctx or default_context()
        if isinstance(feature_view, features.FeatureSet):
            feature_view = FeatureView(feature_view, reconstruction)
        self._loss = loss_function or self.options['loss']
        flat = _FlatBA(reconstruction, self.setup, feature_view, self.options, scene=getattr(self, "scene", None))
        self._flat, self._ba, self._arena = flat, None, None
        if len(flat.obs_image) == 0:
            return
        C = flat.first_patch().shape[2]
        refs = None                                                              # cost maps: "just minimize"
        if isinstance(references, features.ReferenceMap):
            refs = references.descriptor_matrix(flat.point_ids)                  # references.at(point3D_id), all rows at once
        elif references is not None:
            refs = np.zeros((len(flat.point_ids), C))
            for k, pid in enumerate(flat.point_ids):
                refs[k] = references[pid].descriptor.reshape(-1)                 # references.at(point3D_id)
        from .. import parallel
        rank, world = parallel.world()
        self._share = None
        if world > 1:          # points sharded over the ranks, cameras replicated
            self._share = _rank_share(flat, rank, world)
            self._share_lo = self._share.lo
            if len(self._share.obs_image) == 0:
                raise ValueError("rank %d received no observations: fewer points than ranks" % rank)
            self._arena = _arena_of(ctx, self._share, getattr(self, "arena_cache", None))
            lo, n_loc = self._share_lo, len(self._share.point_ids)
            self._ba = BAProblem(ctx, self._arena, self._share.problem_dict(None if refs is None else refs[lo:lo + n_loc],
                                                                           self._arena.index))
            return
        from ._timing import phase
        with phase("upload"):
            self._arena = _arena_of(ctx, flat, getattr(self, "arena_cache", None))
        with phase("problem_to_device"):
            self._ba = BAProblem(ctx, self._arena, flat.problem_dict(refs, self._arena.index))

    @property
    def problem(self):
        """The reference exposes its ceres::Problem here; the accelerated path exposes the device-resident
        BAProblem (evaluate / inspect it with its eval(), cost() and params())."""
        return self._ba

    def solve_problem(self, reconstruction):
        """SolveProblem (bundle_optimizer.h:172-245): False when the problem has no residuals."""
        if getattr(self, "_flat", None) is None:
            raise ValueError("set_up() has not been called")
        flat, ba = self._flat, self._ba
        if ba is None:
            return False                                                         # NumResiduals() == 0, :174-176
        C = self._arena.C
        s = self.options['solver']
        lm = lm_options(max_iterations=s['max_num_iterations'], function_tolerance=s['function_tolerance'],
                        gradient_tolerance=s['gradient_tolerance'], parameter_tolerance=s['parameter_tolerance'],
                        max_consecutive_invalid_steps=s['max_num_consecutive_invalid_steps'],
                        use_inner_iterations=s['use_inner_iterations'],
                        max_linear_solver_iterations=s['max_linear_solver_iterations'],
                        linear_solver=linear_solver_for(self.setup.num_images()))
        point_const, allreduce = flat.point_const, self.allreduce
        if getattr(self, "_share", None) is not None:
            from .. import parallel
            lo, n_loc = self._share_lo, len(self._share.point_ids)
            point_const = flat.point_const[lo:lo + n_loc]
            if allreduce is None:
                allreduce = parallel.ensure_collective(ba.ctx)      # native RCCL (None) or the callback form
        # solver.callbacks (base/src/callbacks.h): one call per LM iteration; the first hook records Summary::iterations
        from types import SimpleNamespace
        history = []
        fields = ("iteration", "step_is_valid", "step_is_successful", "cost", "cost_change", "relative_decrease", "trust_region_radius", "step_norm")

        def record(it):
            history.append(SimpleNamespace(**{f: getattr(it, f) for f in fields}))
        ba.ctx.set_iteration_callbacks([record] + list(s.get('callbacks') or []))
        from ._timing import phase
        try:
            with phase("solve"):
                summ = ba.solve(self.interpolation.to_engine(), make_loss(self._loss['name'], self._loss['params']),
                                flat.pose_const, flat.tvec_mask, flat.cam_mask, point_const, options=lm,
                                allreduce=allreduce)
        finally:
            ba.ctx.set_iteration_callbacks(None)
        with phase("write_back"):
            self._write_back(reconstruction, flat, ba, lo if getattr(self, "_share", None) is not None else None,
                             n_loc if getattr(self, "_share", None) is not None else None)
        self._summary = Summary(summ, num_residuals=len(flat.obs_image) * C)
        self._summary.iterations = history
        return True

    def _write_back(self, reconstruction, flat, ba, lo, n_loc):
        q, t, k, X = ba.params()
        if lo is not None:                                          # every rank ends up with all refined points
            from .. import parallel
            X = parallel.gather_rows(X, np.arange(lo, lo + n_loc), len(flat.point_ids))
        for n, i in enumerate(flat.image_ids):       # in place, like feature_reference_bundle_optimizer.h:111-114
            reconstruction.images[i].qvec = q[n].copy()
            reconstruction.images[i].tvec = t[n].copy()
        for n, c in enumerate(flat.camera_ids):
            cam = reconstruction.cameras[c]
            cam.params = k[n, :len(cam.params)].copy()
        for n, pid in enumerate(flat.point_ids):
            reconstruction.points3D[pid].xyz = X[n].copy()

    def reset(self):
        """Reset (bundle_optimizer.h:124-129): drop the problem; the optimizer can be set up again."""
        if getattr(self, "_arena", None) is not None:
            self._arena.close()
        self._flat = self._ba = self._arena = None
        self._used = False

    def run(self, reconstruction, feature_view, references):
        """Run = SetUp + SolveProblem (feature_reference_bundle_optimizer.h:54-71)."""
        try:
            self.set_up(reconstruction, None, feature_view, references)
            ok = self.solve_problem(reconstruction)
        finally:
            # a scene dump / shared arena handed in by the adjuster describes THIS call's objects only: a second run on
            # another feature set (or an edited reconstruction) must not find them
            self.scene = self.arena_cache = None
        if self._arena is not None:
            self._arena.close()
            self._arena = None
        return ok

    def summary(self):
        return self._summary


class CostMapBundleOptimizer(FeatureReferenceBundleOptimizer):
    """_bundle_adjustment.CostMapBundleOptimizer(options, setup, interp_conf).run(reconstruction, costmap_view)
    (bindings.cc:143-146; costmap_bundle_optimizer.h:17-132): the same residual functor on 1- or 3-channel cost maps,
    no reference descriptor -- the residual block is the interpolated texel of the map."""

    def set_up(self, reconstruction, loss_function, feature_view):
        channels = (feature_view.feature_set if isinstance(feature_view, FeatureView) else feature_view).channels
        if channels not in (1, 3):
            raise ValueError("Unsupported dimensions (CHANNELS).")               # costmap_bundle_optimizer.h:9-14
        super().set_up(reconstruction, loss_function, feature_view, None)

    def run(self, reconstruction, feature_view):
        self.set_up(reconstruction, None, feature_view)
        ok = self.solve_problem(reconstruction)
        if self._arena is not None:
            self._arena.close()                                                  # a no-op for device-resident cost maps
            self._arena = None
        return ok


class BundleAdjuster:
    """pixsfm/bundle_adjustment/main.py:30-102."""
    default_conf = {
        'apply': True,
        'interpolation': base.interpolation_default_conf,
        'level_indices': None,
        'max_tracks_per_problem': 10,
        'optimizer': {
            'loss': {'name': 'cauchy', 'params': [0.25]},
            'solver': {**base.solver_default_conf, 'use_inner_iterations': True},
            'print_summary': False,
            'refine_focal_length': True, 'refine_principal_point': False, 'refine_extra_params': True,
            'refine_extrinsics': True,
        },
        'references': {'loss': {'name': 'cauchy', 'params': [0.25]}, 'iters': 100, 'keep_observations': False,
                       'compute_offsets3D': False, 'num_threads': -1},
        'strategy': 'feature_reference',
    }

    @classmethod
    def create(cls, conf):
        strategy = conf.get("strategy", cls.default_conf["strategy"])
        strategies = {"feature_reference": FeatureReferenceBundleAdjuster, "costmaps": CostMapBundleAdjuster}   # main.py:67-73
        if strategy not in strategies:
            raise ValueError("strategy %r is outside the accelerated path (feature_reference, costmaps)" % strategy)
        return strategies[strategy](conf)

    def refine(self, reconstruction, feature_set, problem_setup=None):
        raise NotImplementedError()

    def refine_multilevel(self, reconstruction, feature_manager, problem_setup=None):
        levels = self.conf['level_indices'] if self.conf['level_indices'] not in [None, "all"] else \
            list(reversed(range(feature_manager.num_levels)))
        from ._timing import gc_paused
        outputs = {}
        with gc_paused():
            for level_index in levels:
                out = self.refine(reconstruction, feature_manager.fset(level_index), problem_setup)
                for k, v in out.items():
                    outputs.setdefault(k, []).append(v)
        return outputs


class FeatureReferenceBundleAdjuster(BundleAdjuster):
    """main.py:105-154."""
    default_conf = deepcopy(BundleAdjuster.default_conf)

    def __init__(self, conf):
        self.conf = base.merge_adjuster_conf(self.default_conf, conf)

    def refine(self, reconstruction, feature_set, problem_setup=None):
        if problem_setup is None:
            problem_setup = default_problem_setup(reconstruction)
        feature_view = FeatureView(feature_set, reconstruction)
        from ._timing import phase
        with phase("problem_labels"):
            problem_labels = find_problem_labels(reconstruction, self.conf['max_tracks_per_problem'])
        ref_extractor = ReferenceExtractor(deepcopy(self.conf['references']), self.conf['interpolation'])
        with features.SharedArena() as shared:       # host patches cross PCIe once for the extraction AND the optimiser
            from .. import parallel
            if parallel.world()[1] == 1:             # (several ranks: every rank uploads its share only)
                with phase("prefetch_start"):        # the upload starts now and runs beside the walk over the scene objects
                    shared.prefetch(default_context(), feature_set,
                                    [reconstruction.images[i].name for i in reconstruction.reg_image_ids()])
            with phase("dump"):
                scene = _SceneDump(reconstruction, feature_view)      # the Python objects are read once for both steps
            references = ref_extractor.run(problem_labels, reconstruction, feature_set, arena_cache=shared, scene=scene)
            solver = FeatureReferenceBundleOptimizer(deepcopy(self.conf['optimizer']), problem_setup,
                                                     self.conf['interpolation'])
            solver.arena_cache, solver.scene = shared, scene
            solver.run(reconstruction, feature_view, references)
        return {"references": references, "summary": solver.summary()}


class CostMapBundleAdjuster(BundleAdjuster):
    """main.py:218-286: the reference's low-memory strategy ("costmaps").  References by IRLS, one cost map per
    observation (the feature patches are not needed afterwards: 1.5 KB instead of 64 KB per observation at 16 x 16 fp16),
    then BA on the maps with l2_normalize = False (main.py:269-270)."""
    default_conf = {
        **deepcopy(BundleAdjuster.default_conf),
        'costmaps': {'loss': {'name': 'trivial', 'params': []}, 'as_gradientfield': True,
                     'compute_cross_derivative': False, 'num_threads': -1},
    }   # ('strategy' stays the inherited 'feature_reference', like in the reference: the key only steers create())

    def __init__(self, conf):
        self.conf = base.merge_adjuster_conf(self.default_conf, conf)

    def refine(self, reconstruction, feature_set, problem_setup=None):
        if problem_setup is None:
            problem_setup = default_problem_setup(reconstruction)
        problem_labels = find_problem_labels(reconstruction, self.conf['max_tracks_per_problem'])
        interp_conf = deepcopy(self.conf['interpolation'])
        ref_extractor = ReferenceExtractor(deepcopy(self.conf['references']), interp_conf)
        ce = CostMapExtractor(deepcopy(self.conf['costmaps']), interp_conf)
        costmap_fset, references = ce.run(problem_labels, reconstruction, feature_set, ref_extractor)
        interp_conf = dict(interp_conf, l2_normalize=False)        # "Make sure l2_normalize is set to false before optim!"
        costmap_view = FeatureView(costmap_fset, reconstruction)
        solver = CostMapBundleOptimizer(deepcopy(self.conf['optimizer']), problem_setup, interp_conf)
        solver.run(reconstruction, costmap_view)
        return {"costmaps": costmap_fset, "references": references, "summary": solver.summary()}
