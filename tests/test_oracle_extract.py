"""CPU: the patch-producer restatement (oracle/pxo_extract.py) against the golden fixture produced by
the REFERENCE's own extract_patches.py + torch normalize (tests/golden/make_golden_extract.py)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_ref.npz")


def _ulp_diff_f16(a, b):
    return np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32))


def test_restatement_matches_reference_gather():
    import pxo_extract
    g = np.load(GOLD)
    fmap, kps, size = pxo_extract.golden_inputs()
    patches, corners, scale = pxo_extract.sparse_patches(fmap, kps, size)
    assert np.array_equal(corners, g["corners"]) and np.array_equal(scale, g["scale"])
    # corners hit both clip bounds and the interior
    assert corners.min() == 0 and (corners[:, 0].max(), corners[:, 1].max()) == (26 - 17, 22 - 17)
    assert patches.dtype == g["patches"].dtype == np.float16 and patches.shape == g["patches"].shape
    # the L2 norm is an fp32 reduction whose summation order differs between numpy and torch:
    # values agree to 1 fp16 ulp, almost all bit-exactly
    d = _ulp_diff_f16(patches, g["patches"])
    assert d.max() <= 1 and (d == 0).mean() > 0.995


def test_without_normalisation_the_gather_is_bit_exact():
    import pxo_extract
    fmap, kps, size = pxo_extract.golden_inputs()
    patches, corners, _ = pxo_extract.sparse_patches(fmap, kps, size, l2_normalize=False, dtype=np.float32)
    for k, (x0, y0) in enumerate(corners):
        assert np.array_equal(patches[k], fmap[:, y0:y0 + 16, x0:x0 + 16].transpose(1, 2, 0))
    # unit norm after normalisation (what the KA/BA kernels assume of stored descriptors, extractor.py:173)
    p16, _, _ = pxo_extract.sparse_patches(fmap, kps, size)
    assert np.abs(np.linalg.norm(p16.astype(np.float64), axis=-1) - 1).max() < 2e-3
