"""`_pixsfm._features` (pixsfm/features/bindings.cc:38-300).  The reference exports one class per storage type
(`FeaturePatch_f16` ...) plus factory functions that dispatch on the numpy dtype; the accelerated containers are
dtype-generic, so every suffixed name maps to the same class."""
from ..api.bundle_adjustment import FeatureView  # noqa: F401
from ..api.features import (FeatureManager, FeatureMap, FeaturePatch, FeatureSet, PatchInterpolator,  # noqa: F401
                            Reference, kDenseId)

for _sfx in ("_f16", "_f32", "_f64"):
    globals()["FeaturePatch" + _sfx] = FeaturePatch
    globals()["FeatureMap" + _sfx] = FeatureMap
    globals()["FeatureSet" + _sfx] = FeatureSet
    globals()["FeatureView" + _sfx] = FeatureView
    globals()["FeatureManager" + _sfx] = FeatureManager


class Map_IdReference(dict):
    """Opaque std::unordered_map<point3D_t, Reference> of the reference: a dict here."""


from ..api.features import _PatchStatus as PatchStatus  # noqa: E402,F401
