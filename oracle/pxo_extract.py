"""CPU restatement of the sparse patch producer (TEST INFRASTRUCTURE ONLY -- never imported by the
product package).

Follows FeatureExtractor.tensor_to_fmap, sparse branch (pixsfm/features/extractor.py:152-199) and
extract_patches_torch / extract_patches_numpy (pixsfm/features/extract_patches.py:13-44):
  extractor.py:173-175  featuremap = normalize(featuremap, dim=1); featuremap.to(dtype)
  extractor.py:177      scale = (fw / w, fh / h)
  extractor.py:192-193  corners = (keypoints * scale - ps / 2).astype(int32); clip to [0, (fw, fh) - ps - 1]
  extract_patches.py    patches[k] = featuremap[:, y0:y0+ps, x0:x0+ps] permuted to (ps, ps, C)
Pinned by tests/golden/extract_ref.npz, generated with the reference's own extract_patches.py and
torch.nn.functional.normalize (tests/golden/make_golden_extract.py).
"""
import numpy as np


def sparse_patches(featuremap, keypoints, image_size, ps=16, l2_normalize=True, dtype=np.float16):
    """featuremap (C, fh, fw) float32/float16; keypoints (n, 2) image coords; image_size (w, h).
    Returns (patches (n, ps, ps, C) dtype, corners (n, 2) int32, scale (2,) float64)."""
    fm = np.asarray(featuremap).astype(np.float32)
    c, fh, fw = fm.shape
    if l2_normalize:   # torch.nn.functional.normalize: v / max(||v||_2, eps = 1e-12), float32
        norm = np.sqrt(np.sum(fm * fm, axis=0, dtype=np.float32), dtype=np.float32)
        fm = fm / np.maximum(norm, np.float32(1e-12))
    fm = fm.astype(dtype)
    w, h = image_size
    scale = np.array((fw / w, fh / h))
    corners = (np.asarray(keypoints, dtype=np.float64) * scale - ps / 2.0).astype(np.int32)
    corners = np.clip(corners, [0, 0], np.array([fw, fh]) - ps - 1).astype(np.int32)
    patches = np.empty((len(corners), ps, ps, c), dtype=dtype)
    for k, (x0, y0) in enumerate(corners):
        patches[k] = fm[:, y0:y0 + ps, x0:x0 + ps].transpose(1, 2, 0)
    return patches, corners, scale


def golden_inputs(seed=7, channels=128, fh=22, fw=26, image_size=(104.0, 66.0)):
    """The seeded inputs of tests/golden/extract_ref.npz (PCG64 streams are stable across numpy versions)."""
    rng = np.random.default_rng(seed)
    fmap = rng.normal(0, 1, (channels, fh, fw)).astype(np.float32)
    w, h = image_size
    kps = np.array([[0.3, 0.4], [w - 0.2, h - 0.1], [w / 2, h / 2], [w / 2 + 0.37, 5.0], [3.0, h / 2 - 0.49],
                    [w - 40.0, h - 30.0]])
    return fmap, kps, image_size
