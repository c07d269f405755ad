"""`pixsfm._pixsfm`-shaped adapter over the MI355X engine.

The reference's L4 Python (pixsfm/keypoint_adjustment/main.py:8, bundle_adjustment/main.py:7,
base/__init__.py, features/__init__.py, localization/main.py) imports its native code as
`from .._pixsfm import _keypoint_adjustment as ka` etc.; `pixsfm/_pixsfm/bindings.cc:34-63` defines that module
and its sub-modules `_base`, `_features`, `_keypoint_adjustment`, `_bundle_adjustment`, `_localization`,
`_residuals`, `_util`.  This package has the same sub-module names and exports the same class / function names,
implemented by `pixsfm_amd.api` on libpixsfm_hip.so.  Two ways to use it:

  * a pixsfm maintainer drops this directory into the pixsfm package as `pixsfm/_pixsfm/` instead of the
    compiled extension (it is pure Python over the C-ABI), or
  * `pixsfm_amd._pixsfm.install_as("pixsfm._pixsfm")` registers it in `sys.modules` under that name before
    `pixsfm` is imported (tests/test_pixsfm_shim.py executes the reference's unmodified main.py files that way).

Outside the accelerated path (and therefore absent or raising): `_residuals` (per-block ceres::CostFunction
objects for pyceres -- the batched kernels replace them), PatchWarpBundleOptimizer, GeometricBundleOptimizer.
"""
import sys
import types

from . import _base, _bundle_adjustment, _features, _keypoint_adjustment, _localization, _residuals, _util  # noqa: F401


class CppLogConfig:
    """structlog of the reference (`_pixsfm.cpplog`, bindings.cc:37-43): the engine logs through Python."""
    silence = False
    headers = True
    level = 1


cpplog = CppLogConfig()


class ostream_redirect:                      # py::add_ostream_redirect(m, "ostream_redirect")
    def __init__(self, stdout=True, stderr=True):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_SUBMODULES = ("_base", "_features", "_keypoint_adjustment", "_bundle_adjustment", "_localization", "_residuals", "_util")


def install_as(name="pixsfm._pixsfm"):
    """Register this package and its sub-modules in sys.modules under `name` (default: where pixsfm's Python expects
    its native module).  Returns the module object registered."""
    this = sys.modules[__name__]
    sys.modules[name] = this
    for sub in _SUBMODULES:
        sys.modules[name + "." + sub] = getattr(this, sub)
    parent = name.rpartition(".")[0]
    if parent and parent in sys.modules and isinstance(sys.modules[parent], types.ModuleType):
        setattr(sys.modules[parent], name.rpartition(".")[2], this)
    return this
