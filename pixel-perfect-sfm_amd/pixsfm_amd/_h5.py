"""ctypes loader of libpixsfm_h5.so (include/pixsfm_h5.h): the native reader of pixsfm's dense-feature cache.
Separate from libpixsfm_hip.so because it links the HDF5 C library; missing library -> PixsfmHipError, no fallback."""
import ctypes as C
import os

import numpy as np

from ._lib import F16, F32, F64, PixsfmHipError

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpixsfm_h5.so")
_DT2NP = {F16: np.float16, F32: np.float32, F64: np.float64}

_SIGNATURES = {
    "pxr_h5_last_error": (C.c_char_p, []),
    "pxr_h5_open": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "pxr_h5_close": (C.c_int, [C.c_void_p]),
    "pxr_h5_num_levels": (C.c_int, [C.c_void_p]),
    "pxr_h5_level_channels": (C.c_int, [C.c_void_p, C.c_int]),
    "pxr_h5_dtype": (C.c_int, [C.c_void_p]),
    "pxr_h5_num_images": (C.c_int, [C.c_void_p, C.c_int]),
    "pxr_h5_image_name": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int]),
    "pxr_h5_map_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pxr_h5_map_meta": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pxr_h5_read_patches": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]),
}
_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PixsfmHipError("libpixsfm_h5.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                                 "g.build()'` (needs the HDF5 C library of the image, /opt/conda)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load().pxr_h5_last_error()
        raise PixsfmHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


class FeatureCache:
    """An open cache file: FeatureManager(h5_path, fill, level_prefix) of the reference (featuremanager.cc:20-40)."""

    def __init__(self, path, level_prefix=""):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib.pxr_h5_open(str(path).encode(), level_prefix.encode(), C.byref(h)), "pxr_h5_open")
        self.handle = h
        self.num_levels = self.lib.pxr_h5_num_levels(h)
        self.channels_per_level = [self.lib.pxr_h5_level_channels(h, l) for l in range(self.num_levels)]
        self.dtype = np.dtype(_DT2NP[self.lib.pxr_h5_dtype(h)])

    def image_names(self, level):
        return [self.lib.pxr_h5_image_name(self.handle, level, i).decode()
                for i in range(self.lib.pxr_h5_num_images(self.handle, level))]

    def map_info(self, level, image):
        fmt, sparse, n, H, W, Cc = C.c_int(), C.c_int(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.pxr_h5_map_info(self.handle, level, image.encode(), C.byref(fmt), C.byref(sparse), C.byref(n),
                                        C.byref(H), C.byref(W), C.byref(Cc)), "pxr_h5_map_info")
        return dict(format=fmt.value, is_sparse=bool(sparse.value), n=n.value, shape=(H.value, W.value, Cc.value))

    def map_meta(self, level, image, n):
        ids, corners, scales = np.empty(n, np.int32), np.empty((n, 2), np.int32), np.empty((n, 2), np.float64)
        _check(self.lib.pxr_h5_map_meta(self.handle, level, image.encode(), ids.ctypes.data, corners.ctypes.data,
                                        scales.ctypes.data), "pxr_h5_map_meta")
        return ids, corners, scales

    def read_patches(self, level, image, which=None, out=None):
        """Patches `which` (positions in the stored order; None = all) as an (n, H, W, C) array of the file's dtype."""
        info = self.map_info(level, image)
        which = None if which is None else np.ascontiguousarray(which, dtype=np.int64)
        n = info["n"] if which is None else len(which)
        if out is None:
            out = np.empty((n,) + info["shape"], self.dtype)
        assert out.shape == (n,) + info["shape"] and out.dtype == self.dtype and out.flags.c_contiguous
        _check(self.lib.pxr_h5_read_patches(self.handle, level, image.encode(), n, None if which is None else which.ctypes.data,
                                            out.ctypes.data), "pxr_h5_read_patches")
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pxr_h5_close(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
