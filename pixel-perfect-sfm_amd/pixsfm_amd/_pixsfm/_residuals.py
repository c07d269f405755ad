"""`_pixsfm._residuals` (pixsfm/residuals/bindings.cc:14-30): single residual blocks for power users.

The reference hands out ceres::CostFunction* objects for a pyceres.Problem.  There is no Ceres here: the two functors that
are on the accelerated path come back as objects with the ceres::CostFunction surface -- num_residuals(),
parameter_block_sizes(), evaluate(*parameter_blocks) -> (ok, residuals, jacobians) with jacobians[b] row-major
num_residuals x block_size_b -- evaluated by the same batched kernels as everything else (a batch of one: pxr_ba_eval in
materialise mode and pxr_ba_projection_jacobian, J = gx P0 + gy P1).  The patch-warp (two-view) and geometric functors are
outside the accelerated path and say so."""
import numpy as np

from ..api import base
from ..api.reconstruction import CAMERA_MODELS


class BlockCostFunction:
    """FeatureReferenceCostFunctor / FeatureReferenceConstantPoseCostFunctor (residuals/src/feature_reference.h:83-215)."""

    def __init__(self, camera_model_id, patch, reference_descriptor, interpolation_config, constant_pose=None, ctx=None):
        from ..api import features
        self.model_id = int(camera_model_id)
        if self.model_id not in CAMERA_MODELS:
            raise ValueError("Camera model does not exist")
        self.patch = patch
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        channels = patch.shape[2]
        if len(self.interpolation.nodes) != 1 or channels not in (128, 1):      # feature_reference.h:273-279, :308-314
            raise ValueError("Unsupported dimensions (CHANNELS,N_NODES).")
        ref = np.asarray(reference_descriptor, dtype=np.float64)
        if ref.shape != (1, channels):                                          # THROW_CHECK_EQ rows / cols, :264-265
            raise ValueError("reference_descriptor must be %d x %d" % (1, channels))
        self.channels, self.ref = channels, ref.reshape(-1).copy()
        self.constant_pose = constant_pose
        self.ctx = ctx
        self._features = features

    def num_residuals(self):
        return self.channels

    def parameter_block_sizes(self):
        K = CAMERA_MODELS[self.model_id][1]
        return [3, K] if self.constant_pose is not None else [4, 3, 3, K]

    def evaluate(self, *parameters, jacobians=True):
        """parameters: (qvec, tvec, xyz, camera_params), or (xyz, camera_params) for the constant-pose functor.  Returns
        (ok, residuals [C], [J_b [C][size_b]] | None); ok is False when check_bounds is set, no reference is subtracted and
        the projection leaves the patch (feature_reference.h:128-130)."""
        from ..engine import BAProblem, KPAD
        from ..api.keypoint_adjustment import default_context
        sizes = self.parameter_block_sizes()
        if len(parameters) != len(sizes) or any(np.size(p) != s for p, s in zip(parameters, sizes)):
            raise ValueError("expected parameter blocks of sizes %r" % (sizes,))
        if self.constant_pose is not None:
            (q, t), (X, k) = self.constant_pose, parameters
        else:
            q, t, X, k = parameters
        K = sizes[-1]
        cam = np.zeros((1, KPAD))
        cam[0, :K] = np.asarray(k, dtype=np.float64).reshape(-1)
        ctx = self.ctx or default_context()
        arena = self._features.to_arena(ctx, [self.patch])
        prob = dict(obs_image=np.zeros(1, np.int32), obs_point=np.zeros(1, np.int32), obs_patch=arena.index, image_camera=np.zeros(1, np.int32),
                    qvec=np.asarray(q, dtype=np.float64).reshape(1, 4), tvec=np.asarray(t, dtype=np.float64).reshape(1, 3),
                    cam_model=np.array([self.model_id], np.int32), cam_params=cam, xyz=np.asarray(X, dtype=np.float64).reshape(1, 3),
                    refs=None if self.ref is None else self.ref.reshape(1, -1))
        ba = BAProblem(ctx, arena, prob)
        rec, r, gx, gy = ba.eval(self.interpolation.to_engine(), with_jacobian=bool(jacobians), materialize=True)
        ok = bool(np.isfinite(rec.download()[0, 0]))
        res = r.download()[0]
        out = None
        if jacobians:
            P = ba.projection_jacobian().download()[0]                                  # 2 x (4 | 3 | 3 | KPAD)
            J = gx.download()[0][:, None] * P[0][None, :] + gy.download()[0][:, None] * P[1][None, :]
            blocks = [J[:, 0:4], J[:, 4:7], J[:, 7:10], J[:, 10:10 + K]]
            out = [np.ascontiguousarray(b) for b in (blocks[2:] if self.constant_pose is not None else blocks)]
        arena.close()
        return ok, res, out


def FeatureReferenceCostFunctor(camera_model_id, patch, reference_descriptor, interpolation_config, ctx=None):
    """CreateFeatureReferenceCostFunctor<dtype> (feature_reference.h:255-283).  Like the reference's binding, this overload
    checks the shape of `reference_descriptor` and then does NOT use it (it forwards nullptr, :267-271): the residual is the
    interpolated descriptor itself.  Parameter blocks: qvec, tvec, point3D, camera parameters."""
    f = BlockCostFunction(camera_model_id, patch, reference_descriptor, interpolation_config, ctx=ctx)
    f.ref = None
    return f


def FeatureReferenceConstantPoseCostFunctor(camera_model_id, qvec, tvec, patch, reference_descriptor, interpolation_config, ctx=None):
    """CreateFeatureReferenceConstantPoseCostFunctor<dtype> (feature_reference.h:287-316): pose captured at construction
    (by value here; the reference keeps the pointers), residual = descriptor - reference.  Parameter blocks: point3D, camera
    parameters."""
    pose = (np.array(qvec, dtype=np.float64).reshape(4), np.array(tvec, dtype=np.float64).reshape(3))
    return BlockCostFunction(camera_model_id, patch, reference_descriptor, interpolation_config, constant_pose=pose, ctx=ctx)


def _outside(name, why):
    def factory(*a, **k):
        raise NotImplementedError("%s is outside the accelerated path (%s)" % (name, why))
    factory.__name__ = name
    return factory


FeatureMetricCostFunctor = _outside("FeatureMetricCostFunctor", "two-view patch-warp residual, DESIGN.md section 7")
GeometricCostFunctor = _outside("GeometricCostFunctor", "reprojection error: COLMAP's, not featuremetric")
GeometricConstantPoseCostFunctor = _outside("GeometricConstantPoseCostFunctor", "reprojection error: COLMAP's, not featuremetric")
