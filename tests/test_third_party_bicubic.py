"""The bicubic VALUES against third-party code: Pillow's BICUBIC filter.

The reference interpolates with Ceres' `CubicHermiteSpline` (`base/src/interpolation.h:183-217`, `cubic_hermite_spline_simd.h`): the
Catmull-Rom cubic, i.e. Keys' cubic convolution kernel with a = -1/2.  Pillow's `Image.BICUBIC` is that kernel (`Resample.c`:
`bicubic_filter`, a = -0.5; OpenCV and torch use -0.75).  Upscaling an 'F' image by an integer factor k evaluates it, separably and in
float32, at the input coordinates u = (x + 0.5) / k - 0.5; away from the border (all four taps inside the image) that is the very
number `pxo_bicubic` / `pxr_interpolate` must give at (row, col) = (u_y, u_x).  (At the border Pillow drops the taps that fall outside
and renormalises, the reference clamps the index: not compared.)  Code the builder did not write; pins the interpolation VALUES
of A2 to float32 rounding -- the derivatives are pinned by torch.autograd (test_third_party_autodiff.py)."""
import numpy as np
import pytest
from PIL import Image

K = 4          # upscaling factor: sample offsets 0.125, 0.375, 0.625, 0.875 inside every cell


def _pillow_samples(img):
    """img (H, W) float32 -> (values (K H, K W) float32, rows (K H,), cols (K W,)) in input pixel coordinates"""
    H, W = img.shape
    big = np.asarray(Image.fromarray(img, mode="F").resize((K * W, K * H), resample=Image.BICUBIC), dtype=np.float32)
    rows = (np.arange(K * H) + 0.5) / K - 0.5
    cols = (np.arange(K * W) + 0.5) / K - 0.5
    return big, rows, cols


def _interior(coords, n):
    f = np.floor(coords)
    return (f - 1 >= 0) & (f + 2 <= n - 1)


@pytest.mark.parametrize("C", [1, 8])
def test_oracle_bicubic_equals_pillow(C):
    import pxo
    rng = np.random.default_rng(5 + C)
    H, W = 12, 16
    data = rng.normal(size=(H, W, C)).astype(np.float32)
    patch64 = pxo.make_patch(np.ascontiguousarray(data.astype(np.float64)))
    worst, n = 0.0, 0
    for ch in range(C):
        big, rows, cols = _pillow_samples(np.ascontiguousarray(data[:, :, ch]))
        ri = np.nonzero(_interior(rows, H))[0]
        ci = np.nonzero(_interior(cols, W))[0]
        for a in ri[:: 3]:
            for b in ci[:: 3]:
                f, _, _ = pxo.bicubic(patch64, float(rows[a]), float(cols[b]))
                worst = max(worst, abs(f[ch] - float(big[a, b])))
                n += 1
    assert n > 200
    assert worst < 5e-6, worst          # float32 separable passes in Pillow against doubles (values of order 1-3)


def test_pillow_kernel_is_the_catmull_rom_one_and_not_opencvs():
    """the test above has teeth: Keys' kernel with a = -0.75 (OpenCV, torch) differs from Pillow by 1e-2 on the same samples"""
    rng = np.random.default_rng(3)
    img = rng.normal(size=(10, 10)).astype(np.float32)
    big, rows, cols = _pillow_samples(img)

    def keys(x, a):
        x = abs(x)
        if x < 1:
            return (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
        if x < 2:
            return a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
        return 0.0

    def sample(a, r, c):
        r0, c0 = int(np.floor(r)), int(np.floor(c))
        return sum(keys(r - (r0 + j), a) * keys(c - (c0 + i), a) * float(img[r0 + j, c0 + i]) for j in range(-1, 3) for i in range(-1, 3))
    a_i, b_i = 4 * 4 + 1, 5 * 4 + 2
    assert abs(sample(-0.5, rows[a_i], cols[b_i]) - float(big[a_i, b_i])) < 5e-6
    assert abs(sample(-0.75, rows[a_i], cols[b_i]) - float(big[a_i, b_i])) > 1e-3


@pytest.mark.gpu
def test_hip_interpolation_equals_pillow(ctx):
    """pxr_interpolate (the SIMD contract from 8 channels up, fp32 storage) against Pillow on the same interior samples"""
    from pixsfm_amd.engine import PatchArena, interp_cfg, interpolate
    rng = np.random.default_rng(11)
    H = W = 16
    C = 128
    data = rng.normal(size=(1, H, W, C)).astype(np.float32)
    arena = PatchArena.from_numpy(ctx, data, np.zeros((1, 2), np.int32), np.ones((1, 2)))
    chans = [0, 7, 64, 127]
    bigs = {}
    for ch in chans:
        bigs[ch], rows, cols = _pillow_samples(np.ascontiguousarray(data[0, :, :, ch]))
    ri = np.nonzero(_interior(rows, H))[0][::2]
    ci = np.nonzero(_interior(cols, W))[0][::2]
    rr, cc = np.meshgrid(ri, ci, indexing="ij")
    # patch coordinates: pixel index u = x * scale - 0.5 - corner (featurepatch.h:250-255) with scale 1, corner 0
    kps = np.stack([cols[cc.ravel()] + 0.5, rows[rr.ravel()] + 0.5], axis=1)
    desc, _ = interpolate(ctx, arena, interp_cfg(l2_normalize=False), kps, np.zeros(len(kps), np.int64))
    for ch in chans:
        want = bigs[ch][rr.ravel(), cc.ravel()].astype(np.float64)
        assert np.abs(desc[:, ch] - want).max() < 5e-6, (ch, np.abs(desc[:, ch] - want).max())
    arena.close()


def _scipy_bicubic(P, r, c):
    """value, d/dr, d/dc at (r, c) of the Catmull-Rom surface over P (H, W) with clamped indices (grid2d.h:64-73), evaluated by
    scipy.interpolate.CubicHermiteSpline: knots 0 and 1 of the cell, tangents = central differences of the neighbours."""
    from scipy.interpolate import CubicHermiteSpline
    H, W = P.shape
    r0, c0 = int(np.floor(r)), int(np.floor(c))
    rows = [min(max(r0 - 1 + j, 0), H - 1) for j in range(4)]
    cols = [min(max(c0 - 1 + i, 0), W - 1) for i in range(4)]

    def seg(p):       # p: four samples -> the spline of the middle interval
        return CubicHermiteSpline([0.0, 1.0], [p[1], p[2]], [0.5 * (p[2] - p[0]), 0.5 * (p[3] - p[1])])
    h, hc = [], []
    for j in range(4):
        s = seg([P[rows[j], cols[i]] for i in range(4)])
        h.append(float(s(c - c0))); hc.append(float(s.derivative()(c - c0)))
    sv = seg(h)
    return float(sv(r - r0)), float(sv.derivative()(r - r0)), float(seg(hc)(r - r0))


def test_oracle_bicubic_and_derivatives_equal_scipys_hermite_spline():
    """Values AND both derivatives, border cells included, against scipy's CubicHermiteSpline (the scalar fp64 path of
    interpolation.h:222-268 for C < 8: 1e-12; the SIMD contract for C >= 8 -- fp32 horizontal pass -- 2e-6)."""
    import pxo
    rng = np.random.default_rng(21)
    H, W = 9, 11
    for C, tol in ((1, 1e-12), (8, 2e-6)):
        data = np.ascontiguousarray(rng.normal(size=(H, W, C)))
        patch = pxo.make_patch(data)
        pts = np.stack([rng.uniform(0.0, H - 1.001, 60), rng.uniform(0.0, W - 1.001, 60)], 1)
        pts[:6] = [[0.3, 0.4], [0.2, W - 1.3], [H - 1.4, 0.6], [H - 1.2, W - 1.1], [0.5, 5.5], [4.5, 0.25]]     # border cells
        for r, c in pts:
            f, dr, dc = pxo.bicubic(patch, float(r), float(c))
            for ch in range(C):
                v, vr, vc = _scipy_bicubic(data[:, :, ch], r, c)
                scale = max(1.0, abs(v), abs(vr), abs(vc))
                assert abs(f[ch] - v) < tol * scale and abs(dr[ch] - vr) < tol * scale and abs(dc[ch] - vc) < tol * scale, (C, r, c, ch)
