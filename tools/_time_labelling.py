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
print("best %.1f ms" % best)
