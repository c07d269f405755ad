"""pixsfm_amd -- MI355X-native featuremetric refinement engine (KA/BA hot path of pixsfm).

Host layer above the C-ABI of libpixsfm_hip.so (include/pixsfm_hip.h).  Importing this
package does not load the shared library; the first call that needs the GPU does, and it
raises PixsfmHipError when the library is missing (no CPU fallback).
"""
from ._lib import PixsfmHipError, KPAD, OBS_REC  # noqa: F401

__version__ = "0.1.0"
