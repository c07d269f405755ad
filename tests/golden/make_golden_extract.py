"""Generates tests/golden/extract_ref.npz by running the REFERENCE's own patch gather
(pixsfm/features/extract_patches.py, loaded by file path -- it only needs numpy + torch) and
torch.nn.functional.normalize on the seeded inputs of oracle/pxo_extract.golden_inputs, following
FeatureExtractor.tensor_to_fmap's sparse branch (pixsfm/features/extractor.py:173-199).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_extract.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pxo_extract  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_extract_patches", "/root/reference/pixsfm/features/extract_patches.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

fmap, kps, (w, h) = pxo_extract.golden_inputs()
ps = 16
featuremap = torch.from_numpy(fmap)[None]                                   # (1, C, fh, fw)
featuremap = torch.nn.functional.normalize(featuremap, dim=1)               # extractor.py:173-174
featuremap = featuremap.to(torch.float16)                                   # :175
scale = np.array((featuremap.shape[3] / w, featuremap.shape[2] / h))        # :177
_, c, fh, fw = featuremap.shape
corners = (kps * scale - ps / 2.0).astype(np.int32)                         # :192
corners = np.clip(corners, [0, 0], np.array([fw, fh]) - ps - 1)             # :193
patches = ref.extract_patches_numpy(featuremap.squeeze(0), corners, ps)     # :194-195
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "extract_ref.npz"),
                    patches=patches, corners=corners.astype(np.int32), scale=scale)
print("wrote extract_ref.npz", patches.shape, patches.dtype, corners.tolist())
