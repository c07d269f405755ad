"""Probe (GPU box): what does pinning the caller's arrays in place cost against the staged upload?  hipHostRegister of an
8.6 GB numpy array, a direct hipMemcpy from it, hipHostUnregister -- next to PatchArena.upload (gather into pinned staging
buffers with streaming stores, then DMA)."""
import ctypes as C
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pixel-perfect-sfm_amd"))
import numpy as np
from pixsfm_amd.engine import Context, PatchArena

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipDeviceSynchronize.argtypes = []
ctx = Context(0)
n = 131072
host = np.empty((n, 16, 16, 128), np.float16); host[:] = 1.0
nb = host.nbytes
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), nb) == 0
for rep in range(2):
    t0 = time.perf_counter(); rc = hip.hipHostRegister(host.ctypes.data, nb, 0); t1 = time.perf_counter()
    assert rc == 0, rc
    rc = hip.hipMemcpy(d, host.ctypes.data, nb, 1); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    assert rc == 0
    rc = hip.hipMemcpy(d, host.ctypes.data, nb, 1); hip.hipDeviceSynchronize(); t2b = time.perf_counter()
    hip.hipHostUnregister(host.ctypes.data); t3 = time.perf_counter()
    print("register %.3f s (%.1f GB/s)  direct copy %.3f s (%.1f GB/s)  again %.3f s (%.1f GB/s)  unregister %.3f s" % (
        t1 - t0, nb / (t1 - t0) / 1e9, t2 - t1, nb / (t2 - t1) / 1e9, t2b - t2, nb / (t2b - t2) / 1e9, t3 - t2b))
t0 = time.perf_counter(); rc = hip.hipMemcpy(d, host.ctypes.data, nb, 1); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
print("pageable hipMemcpy %.3f s (%.1f GB/s)" % (t1 - t0, nb / (t1 - t0) / 1e9))
corners = np.zeros((n, 2), np.int32); scales = np.ones((n, 2))
a = PatchArena(ctx, n, 16, 16, 128, np.float16)
for rep in range(3):
    t0 = time.perf_counter(); a.upload(0, host, corners, scales); ctx.sync(); dt = time.perf_counter() - t0
    print("staged upload %.3f s (%.1f GB/s)" % (dt, nb / dt / 1e9))
