"""Times pxr_graph_track_labels at BASELINE configs[1] scale (10k tracks x 10 nodes, 450k matches); host code only.
PXR_GRAPH_THREADS=n overrides the worker count, PXR_VERBOSE=1 prints the phase split."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pixel-perfect-sfm_amd'))
import numpy as np, ctypes as C
from pixsfm_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
n_tracks, tl, n_img = 10000, 10, 1000
n = n_tracks * tl
node_image = np.empty(n, np.int32)
for t in range(n_tracks):
    node_image[t*tl:(t+1)*tl] = rng.choice(n_img, tl, replace=False)
iu = np.array([(a, b) for a in range(tl) for b in range(a+1, tl)])
base = (np.arange(n_tracks) * tl)[:, None]
src = (base + iu[:, 0][None]).ravel().astype(np.int64); dst = (base + iu[:, 1][None]).ravel().astype(np.int64)
sim = rng.uniform(0.5, 1, len(src))
out = np.empty(n, np.int64); ntr = C.c_int64()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    lib.pxr_graph_track_labels(n, node_image.ctypes.data, len(src), src.ctypes.data, dst.ctypes.data, sim.ctypes.data, out.ctypes.data, C.byref(ntr))
    best = min(best, (time.perf_counter() - t0) * 1e3)

print("host: track labels best %.2f ms (%d tracks)" % (best, ntr.value))

# the device version (labels + scores + roots in one call, graph resident in HBM), if a GPU is present
try:
    from pixsfm_amd.engine import Context
    ctx = Context(0)
except Exception as e:  # noqa: BLE001
    print("no GPU: device labelling not timed (%s)" % (e,))
    sys.exit(0)
order = np.argsort(src, kind="stable")
src, dst, sim = src[order], dst[order], sim[order]
d = [ctx.to_device(a, dt) for a, dt in ((node_image, np.int32), (src, np.int64), (dst, np.int64), (sim, np.float64))]
dl, ds, dr = ctx.empty((n,), np.int64), ctx.empty((n,), np.float64), ctx.empty((n,), np.uint8)
ntr_d = C.c_int64()
for what, sc, rt in (("labels", None, None), ("labels + scores + roots", ds.ptr, dr.ptr)):
    best_d = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        _lib.check(lib.pxr_graph_labels_device(ctx.handle, n, d[0].ptr, len(src), d[1].ptr, d[2].ptr, d[3].ptr, dl.ptr, sc, rt,
                                               C.byref(ntr_d)), "device")
        best_d = min(best_d, (time.perf_counter() - t0) * 1e3)
    print("device: %-24s best %.2f ms (%d tracks)" % (what, best_d, ntr_d.value))
lib.pxr_graph_track_labels(n, node_image.ctypes.data, len(src), src.ctypes.data, dst.ctypes.data, sim.ctypes.data, out.ctypes.data, C.byref(ntr))
print("identical labels:", bool(np.array_equal(dl.download(), out)))
