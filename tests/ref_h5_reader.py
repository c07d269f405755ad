"""ctypes face of oracle/_ref/libpxo_ref_h5.so: the REFERENCE's cache reader (features/src/featuremanager.cc, featureset.cc,
featuremap.cc, featurepatch.cc compiled in place by oracle/Makefile, on a stand-in HighFive over the image's libhdf5).
Test infrastructure: the checker for libpixsfm_h5.so."""
import ctypes as C
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libpxo_ref_h5.so")
DTYPES = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
_lib = None


def available():
    return os.path.isfile(LIB)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        l.pxo_ref_h5_last_error.restype = C.c_char_p
        l.pxo_ref_h5_open.restype = C.c_void_p
        l.pxo_ref_h5_open.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        l.pxo_ref_h5_close.argtypes = [C.c_void_p]
        l.pxo_ref_h5_close.restype = None
        for name in ("num_levels", "channels", "num_images", "image_name", "map_info", "patch_ids", "patch", "unload"):
            getattr(l, "pxo_ref_h5_" + name).restype = C.c_int
        l.pxo_ref_h5_num_levels.argtypes = [C.c_void_p]
        l.pxo_ref_h5_channels.argtypes = [C.c_void_p, C.c_int]
        l.pxo_ref_h5_num_images.argtypes = [C.c_void_p, C.c_int]
        l.pxo_ref_h5_image_name.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        l.pxo_ref_h5_map_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p] + [C.POINTER(C.c_int)] * 3
        l.pxo_ref_h5_patch_ids.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
        l.pxo_ref_h5_patch.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        l.pxo_ref_h5_load.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        l.pxo_ref_h5_load.restype = C.c_longlong
        l.pxo_ref_h5_unload.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int]
        l.pxo_ref_h5_flush.argtypes = [C.c_void_p, C.c_int]
        l.pxo_ref_h5_flush.restype = C.c_longlong
        _lib = l
    return _lib


class ReferenceError_(RuntimeError):
    pass


class ReferenceCache:
    """FeatureManager<dtype>(h5_path, fill, level_prefix) of the reference (featuremanager.cc:20-40)."""

    def __init__(self, path, dtype, fill=True, level_prefix=""):
        self.dtype = np.dtype(dtype)
        self.h = lib().pxo_ref_h5_open(os.fsencode(str(path)), int(fill), level_prefix.encode(), DTYPES[self.dtype])
        if not self.h:
            raise ReferenceError_(lib().pxo_ref_h5_last_error().decode())

    def _ok(self, rc):
        if rc < 0:
            raise ReferenceError_(lib().pxo_ref_h5_last_error().decode())
        return rc

    def close(self):
        if self.h:
            lib().pxo_ref_h5_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def num_levels(self):
        return self._ok(lib().pxo_ref_h5_num_levels(self.h))

    def channels(self, level):
        return self._ok(lib().pxo_ref_h5_channels(self.h, level))

    def image_names(self, level):
        out = []
        buf = C.create_string_buffer(4096)
        for i in range(self._ok(lib().pxo_ref_h5_num_images(self.h, level))):
            self._ok(lib().pxo_ref_h5_image_name(self.h, level, i, buf, 4096))
            out.append(buf.value.decode())
        return out

    def map_info(self, level, image):
        s, n, c = C.c_int(), C.c_int(), C.c_int()
        self._ok(lib().pxo_ref_h5_map_info(self.h, level, image.encode(), C.byref(s), C.byref(n), C.byref(c)))
        return dict(is_sparse=bool(s.value), n=n.value, channels=c.value)

    def patch_ids(self, level, image):
        ids = np.zeros(self.map_info(level, image)["n"], np.uint32)
        self._ok(lib().pxo_ref_h5_patch_ids(self.h, level, image.encode(), ids.ctypes.data))
        return ids

    def patch(self, level, image, pid):
        shape, corner, scale = np.zeros(3, np.int32), np.zeros(2, np.int32), np.zeros(2, np.float64)
        has, ref = C.c_int(), C.c_int()
        args = (self.h, level, image.encode(), int(pid), shape.ctypes.data, corner.ctypes.data, scale.ctypes.data, C.byref(has), C.byref(ref))
        self._ok(lib().pxo_ref_h5_patch(*args, None, 0))
        data = None
        if has.value:
            data = np.zeros(tuple(shape), self.dtype)
            self._ok(lib().pxo_ref_h5_patch(*args, data.ctypes.data, data.nbytes))
        return dict(shape=tuple(int(x) for x in shape), corner=corner, scale=scale, data=data, reference_count=ref.value)

    def load(self, level, image, ids=None, fill=True):
        if ids is None:
            return self._ok(lib().pxo_ref_h5_load(self.h, level, image.encode(), None, -1, int(fill)))
        ids = np.ascontiguousarray(ids, np.uint32)
        return self._ok(lib().pxo_ref_h5_load(self.h, level, image.encode(), ids.ctypes.data, len(ids), int(fill)))

    def unload(self, level, image, ids=None):
        if ids is None:
            return self._ok(lib().pxo_ref_h5_unload(self.h, level, image.encode(), None, -1))
        ids = np.ascontiguousarray(ids, np.uint32)
        return self._ok(lib().pxo_ref_h5_unload(self.h, level, image.encode(), ids.ctypes.data, len(ids)))

    def flush(self, level):
        return self._ok(lib().pxo_ref_h5_flush(self.h, level))
