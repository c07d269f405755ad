"""CPU: pins the oracle's interpolation against
 (1) the KNOWN-ANSWER cases of the reference's own interpolation tests, re-stated here
     (pixsfm/base/src/interpolation_test.cc:21-185 polynomial reproduction at 1e-8, :187-207 unit norm at 1e-10,
      :272-311 Jet chain rule, :327-364 agreement of the SIMD path with ceres::BiCubicInterpolator at 1e-5 -- the latter
      against a restatement of Ceres' published cubic_interpolation.h, Ceres itself being absent),
 (2) finite differences of every analytic derivative (untested in the reference).
These are the only pins the bicubic has: the reference's spline / grid headers include Eigen and Ceres and cannot be compiled in
this image.  In particular the SPLIT of the arithmetic (fp32 horizontal pass for fp16 / fp32 storage, fp64 vertical
pass; cubic_hermite_spline_simd.h:123-175, interpolation.h:183-217) is read from the source and is NOT pinned by any output of
the reference: the 1e-5 / 1e-8 bounds above do not see it."""

import numpy as np
import pytest

import pxo

@pytest.mark.parametrize("dt", [np.float32, np.float64, np.float16])
def test_similar_to_ceres_bicubic(dt):
    """interpolation_test.cc:327-364 (TestSimilarToCeres<dtype,128>): |simd - ceres| < 1e-5."""
    rng = np.random.default_rng(0)
    data = rng.uniform(-1, 1, (10, 10, 128)).astype(dt)
    p = pxo.make_patch(data)
    for r in np.arange(0, 100, 4) / 10.0:
        for c in np.arange(0, 100, 6) / 10.0:
            a = np.stack(pxo.bicubic(p, r, c))
            b = np.stack(pxo.bicubic_ceres(p, r, c))
            assert np.abs(a - b).max() < 1e-5


@pytest.mark.parametrize("coeff", [np.zeros((3, 3)), np.diag([0, 0, 1.0]),
                                   np.array([[0, 0, 0.5], [0, 0, 0], [0.5, 0, 0]]),      # degree 10
                                   np.array([[0, 0, 0], [0, 0, 0.5], [0, 0.5, 0]]),      # degree 01
                                   np.array([[0, 0.5, 0], [0.5, 0, 0], [0, 0, 0]]),      # degree 11
                                   np.array([[1.0, 2, 3], [2, 4, 5], [3, 5, 6]])])       # full biquadratic
@pytest.mark.parametrize("ch", [1, 2, 3, 8])
def test_polynomial_reproduction(coeff, ch):
    """RunPolynomialInterpolationTest (interpolation_test.cc:21-58): bicubic reproduces x^T A x and its
    r / c derivatives on a 10x10 grid for r, c in [1, 8]; per-channel scale dim^2 + 1; tol 1e-8."""
    rr, cc = np.meshgrid(np.arange(10.0), np.arange(10.0), indexing="ij")
    X = np.stack([rr, cc, np.ones_like(rr)], -1)
    F = np.einsum("...i,ij,...j->...", X, coeff, X)
    data = np.ascontiguousarray(F[..., None] * (np.arange(ch) ** 2 + 1))
    p = pxo.make_patch(data)
    cfg = pxo.cfg(l2_normalize=False)
    for r in np.linspace(1, 8, 23):
        for c in np.linspace(1, 8, 19):
            f, dr, dc = pxo.pixel_interp(p, r, c, cfg)
            x = np.array([r, c, 1.0])
            scale = np.arange(ch) ** 2 + 1
            assert np.abs(f - scale * (x @ coeff @ x)).max() < 1e-8
            assert np.abs(dr - scale * ((coeff[0] + coeff[:, 0]) @ x)).max() < 1e-8
            assert np.abs(dc - scale * ((coeff[1] + coeff[:, 1]) @ x)).max() < 1e-8


def test_l2_normalize_unit_norm():
    """TestL2Normalize (interpolation_test.cc:187-207), same 2x4x2 grid and positions, tol 1e-10."""
    values = np.array([1.0, 5.0, 2.0, 10.0, 2.0, 6.0, 3.0, 5.0, 1.0, 2.0, 2.0, 2.0, 2.0, 2.0, 3.0, 1.0]).reshape(2, 4, 2)
    p = pxo.make_patch(np.ascontiguousarray(values))
    for r, c in ((0.5, 2.5), (1.5, 1.5), (0.0, 3.0)):
        f, _, _ = pxo.pixel_interp(p, r, c, pxo.cfg(l2_normalize=True))
        assert abs(1.0 - f @ f) < 1e-10


@pytest.mark.parametrize("l2", [False, True])
@pytest.mark.parametrize("scale,corner", [((1.0, 1.0), (0, 0)), ((0.5, 0.25), (100, 200)), ((2.0, 3.0), (-5, 7))])
def test_patch_coordinates_and_gradient_finite_differences(l2, scale, corner):
    """FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255) + Jet bridge (interpolation.h:130-140)
    + L2 chain rule (interpolation.h:648-666): analytic d/dx, d/dy vs central differences on fp64
    patches (fp16 patches make the fp32 horizontal pass too noisy for FD)."""
    rng = np.random.default_rng(3)
    data = rng.uniform(-1, 1, (16, 16, 128))
    p = pxo.make_patch(data, corner, scale)
    cfg = pxo.cfg(l2)
    # image point that lands at patch coords (u, v) = (7.3, 8.7)
    xy = np.array([(7.3 + 0.5 + corner[0]) / scale[0], (8.7 + 0.5 + corner[1]) / scale[1]])
    f, gx, gy, inside = pxo.patch_eval(p, xy, cfg)
    f_direct, dr, dc = pxo.pixel_interp(p, 8.7, 7.3, cfg)
    assert np.abs(f - f_direct).max() < 1e-12
    assert np.abs(gx - dc * scale[0]).max() < 1e-12 and np.abs(gy - dr * scale[1]).max() < 1e-12
    e = 1e-5
    gxf = (pxo.patch_eval(p, xy + [e, 0], cfg)[0] - pxo.patch_eval(p, xy - [e, 0], cfg)[0]) / (2 * e)
    gyf = (pxo.patch_eval(p, xy + [0, e], cfg)[0] - pxo.patch_eval(p, xy - [0, e], cfg)[0]) / (2 * e)
    assert np.abs(gxf - gx).max() < 1e-7 * max(1.0, np.abs(gx).max())
    assert np.abs(gyf - gy).max() < 1e-7 * max(1.0, np.abs(gy).max())
    if l2:   # gradients of a unit vector are orthogonal to it
        assert abs(f @ gx) < 1e-12 and abs(f @ gy) < 1e-12


def test_border_clamp_and_check_bounds():
    """Grid2D clamping (grid2d.h:64-73): far outside the patch the value is the corner texel and the
    gradient vanishes; PatchInterpolator::CheckBounds (patch_interpolator.h:160-166)."""
    rng = np.random.default_rng(1)
    data = rng.uniform(-1, 1, (16, 16, 128)).astype(np.float16)
    p = pxo.make_patch(data)
    f, dr, dc = pxo.bicubic(p, -10.0, -10.0)
    assert np.array_equal(f, data[0, 0].astype(np.float64)) and not dr.any() and not dc.any()
    f, dr, dc = pxo.bicubic(p, 40.0, 3.2)
    assert not dr.any() and dc.any()
    cfg = pxo.cfg(check_bounds=True)
    assert pxo.patch_eval(p, [8.0, 8.0], cfg)[3] == 1
    assert pxo.patch_eval(p, [0.4, 8.0], cfg)[3] == 0      # u = -0.1 <= 0
    assert pxo.patch_eval(p, [8.0, 16.6], cfg)[3] == 0     # v = 16.1 >= H
    assert pxo.patch_eval(p, [0.4, 8.0], pxo.cfg())[3] == 1


def test_half_conversion_matches_numpy():
    bits = np.arange(0, 65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([pxo.lib().pxo_half_to_float(int(b)) for b in bits[::7]], dtype=np.float32)
    w = want[::7]
    assert np.array_equal(got[~np.isnan(w)], w[~np.isnan(w)]) and np.isnan(got[np.isnan(w)]).all()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-5, 2000), [0.0, -0.0, 65504.0, 1e6, -1e6, 6e-8]]).astype(np.float32)
    got = np.array([pxo.lib().pxo_float_to_half(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, x.astype(np.float16).view(np.uint16))


def test_half_widening_matches_the_references_half_hpp():
    """Every one of the 65536 binary16 patterns widened to float by the reference's vendored third-party/half.hpp (compiled
    from its own source, oracle/ref_half_shim.cc) and by the oracle's pxo_half_to_float (what the interpolation reads fp16
    texels through; the reference's SIMD path uses F16C's _mm256_cvtph_ps, cubic_hermite_spline_simd.h:51-54, which is the
    same IEEE widening)."""
    import ctypes as C
    r = pxo.ref()
    if r is None:
        pytest.skip("oracle/_ref/libpxo_ref_half.so not built (needs /root/reference at build time)")
    bits = np.arange(65536, dtype=np.uint16)
    want = np.empty(65536, np.float32)
    r.pxo_ref_half_to_float(bits.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_int64(65536))
    got = np.array([pxo.lib().pxo_half_to_float(int(b)) for b in bits], dtype=np.float32)
    nan = np.isnan(want)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32)) and np.isnan(got[nan]).all()
