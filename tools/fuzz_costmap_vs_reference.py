"""Randomised sweep of the cost-map extraction kernels against the REFERENCE's own FillPointCostmap compiled in place
(oracle/_ref/libpxo_ref_costmap.so -- travels to the GPU box with the tree): storage types, patch shapes incl. the 16x16 / 8x8
fast paths, losses, sqrt, channel counts, upsampling / cross derivative.  Reports how many entries differ in the last place.
NOT part of the test suite.  python tools/fuzz_costmap_vs_reference.py [n_groups] [seed]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pixel-perfect-sfm_amd'))
import numpy as np
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss

LIB = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_costmap.so"))
DT = {np.dtype(np.float16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
LOSS = {"trivial": 0, "cauchy": 1, "huber": 2}
n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = Context(0)


def ulps(a, b):
    it = {2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
    ia, ib = a.view(it).astype(np.int64), b.view(it).astype(np.int64)
    sign = np.int64(1) << (8 * a.dtype.itemsize - 1)
    return np.abs(np.where(ia < 0, -(ia + sign), ia) - np.where(ib < 0, -(ib + sign), ib))


tot = off = worst = 0
for g in range(n_groups):
    dt = np.dtype([np.float16, np.float16, np.float32, np.float64][rng.integers(4)])
    H, W = [(16, 16), (8, 8), (16, 16), (8, 8), (12, 12), (10, 7), (5, 16)][rng.integers(7)]
    m = int(rng.integers(3, 400))
    loss = [("trivial", 1.0), ("cauchy", 0.25), ("huber", 0.5)][rng.integers(3)]
    grad, sq = bool(rng.integers(4) != 0), bool(rng.integers(2))
    interp = rng.integers(5) == 0
    up = float(rng.choice([2.0, 1.5, 0.5])) if interp and rng.integers(2) else 1.0
    cross = bool(interp and grad and (up == 1.0 or rng.integers(2)))
    if interp and up == 1.0 and not cross:
        up = 2.0
    if interp:
        H, W = min(H, 8), min(W, 8)
        m = min(m, 40)
    od = dt if (dt != np.float16 or rng.integers(4)) else np.dtype(np.float64)
    l2 = bool(rng.integers(2))
    base = rng.normal(0, 1, (m, 128)); base /= np.linalg.norm(base, axis=1, keepdims=True)
    patches = (base[:, None, None, :] + rng.normal(0, [0.3, 0.03][rng.integers(2)], (m, H, W, 128))).astype(dt)
    co = (4 if cross else 3) if grad else 1
    Ho, Wo = int(H * (up + 1e-6)), int(W * (up + 1e-6))
    want = np.zeros((m, Ho, Wo, co), od)
    for i in range(m):
        p, r, o = np.ascontiguousarray(patches[i]), np.ascontiguousarray(base[i]), want[i]
        rc = LIB.pxo_ref_fill_point_costmap(C.c_void_p(p.ctypes.data), DT[dt], H, W, C.c_void_p(r.ctypes.data), C.c_void_p(o.ctypes.data),
                                            DT[od], Ho, Wo, C.c_double(up), int(grad), int(cross), int(sq), LOSS[loss[0]], C.c_double(loss[1]), int(l2))
        assert rc == 0
    ids = np.arange(m, dtype=np.int32)
    prob = dict(obs_image=ids, obs_point=ids, obs_patch=np.arange(m, dtype=np.int64), image_camera=ids, qvec=np.tile([1.0, 0, 0, 0], (m, 1)),
                tvec=np.zeros((m, 3)), cam_model=np.zeros(m, np.int32), cam_params=np.tile([500.0, 8, 8] + [0.0] * 9, (m, 1)),
                xyz=np.tile([0.0, 0, 2.0], (m, 1)), refs=base, patches=patches, corners=np.zeros((m, 2), np.int32), scales=np.ones((m, 2)))
    arena = PatchArena.from_numpy(ctx, patches, prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cm = ba.extract_costmaps(make_loss(loss[0], [] if loss[0] == "trivial" else [loss[1]]), as_gradientfield=grad, apply_sqrt=sq, dtype=od,
                             upsampling_factor=up, compute_cross_derivative=cross, cfg=interp_cfg(l2_normalize=l2))
    got = np.ascontiguousarray(cm.download()[0])
    if od == np.float64:
        d = np.abs(got - want) > 1e-12 * max(1.0, np.abs(want).max())
        n_off, mx = int(d.sum()), 0
    else:
        u = ulps(got, want)
        u = np.where(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= 1e-12 * np.abs(want).max(), 0, u)
        n_off, mx = int((u > 0).sum()), int(u.max())
    tot += want.size; off += n_off; worst = max(worst, mx)
    if mx > 1 or (od == np.float64 and n_off):
        print("group %d: %s -> %s %dx%d m=%d loss %s grad %d sqrt %d up %.1f cross %d l2 %d: %d entries off, worst %d ulp" % (
            g, dt.name, od.name, H, W, m, loss[0], grad, sq, up, cross, l2, n_off, mx))
    arena.close()
print("entries %d, differing in the last place %d (%.2e), worst %d ulp" % (tot, off, off / max(1, tot), worst))
