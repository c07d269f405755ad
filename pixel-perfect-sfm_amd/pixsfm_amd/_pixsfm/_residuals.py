"""`_pixsfm._residuals` (pixsfm/residuals/bindings.cc:14-30) hands single ceres::CostFunction objects to pyceres.
The accelerated path evaluates all residual blocks of a problem in one kernel (pxr_ba_eval / pxr_ka_eval) and has
no per-block object to offer; the names exist so that `from pixsfm._pixsfm._residuals import *` works."""


def _per_block(name):
    def factory(*a, **k):
        raise NotImplementedError("%s: per-block ceres cost functions are outside the accelerated path; evaluate the "
                                  "problem with pixsfm_amd.engine.BAProblem.eval / KAProblem.eval instead" % name)
    factory.__name__ = name
    return factory


FeatureReferenceCostFunctor = _per_block("FeatureReferenceCostFunctor")
FeatureReferenceConstantPoseCostFunctor = _per_block("FeatureReferenceConstantPoseCostFunctor")
FeatureMetricCostFunctor = _per_block("FeatureMetricCostFunctor")
GeometricCostFunctor = _per_block("GeometricCostFunctor")
GeometricConstantPoseCostFunctor = _per_block("GeometricConstantPoseCostFunctor")
