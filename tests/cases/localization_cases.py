"""Seeded inputs for the query refinements' problem construction: query keypoint adjustment, query bundle
adjustment (three kinds of reference containers, inlier masks, patch indices, camera refinement switches) and nearest-reference
selection.  Inputs only."""
import numpy as np

N_QKA, N_QBA, N_NEAREST = 40, 40, 6


def qka_case(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(1, 13))
    n_patches = n + int(rng.integers(0, 4))
    scales = rng.choice([1.0, 0.5, 0.25], size=(n_patches, 2)) * rng.uniform(0.9, 1.1, (n_patches, 1))
    corners = rng.integers(0, 400, size=(n_patches, 2)).astype(np.int32)
    patch_idxs = rng.permutation(n_patches)[:n].astype(np.int32) if seed % 3 == 0 else None
    own = np.arange(n) if patch_idxs is None else patch_idxs
    # keypoints inside their patch, image coordinates
    kp = (corners[own] + 0.5 + rng.uniform(2, 14, (n, 2))) / scales[own]
    mode = seed % 3 if seed % 7 else 2
    ref_count = {0: np.ones(n), 1: rng.integers(0, 4, n), 2: rng.choice([0, 0, 1, 3, 4], n)}[mode].astype(np.int32)
    inliers = None
    if seed % 4 == 1:
        inliers = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    if seed == 9:
        inliers = np.zeros(n, np.uint8)                       # nothing left: RunQuery returns false
    return dict(kp=kp, corners=corners, scales=scales, sparse=int(seed % 5 != 4), bound=float(rng.choice([-1.0, 0.0, 4.0, 0.5, 40.0])),
                mode=mode, ref_count=ref_count, patch_idxs=patch_idxs, inliers=inliers)


def qba_case(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 13))
    model = int(rng.integers(0, 5))
    base = {0: [1200.0, 500, 500], 1: [1200.0, 1180.0, 500, 500], 2: [1200.0, 500, 500, 0.02], 3: [1200.0, 500, 500, 0.02, -0.01],
            4: [1200.0, 1180.0, 500, 500, 0.02, -0.01, 1e-3, -5e-4]}[model]
    points = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), rng.uniform(3, 5, (n, 1))], 1)
    n_patches = n + int(rng.integers(0, 3))
    patch_idxs = rng.permutation(n_patches)[:n].astype(np.int32) if seed % 3 == 0 else None
    mode = seed % 3 if seed % 7 else 2
    ref_count = {0: np.ones(n), 1: rng.integers(0, 4, n), 2: rng.choice([0, 0, 1, 3, 4], n)}[mode].astype(np.int32)
    inliers = (rng.uniform(size=n) < 0.6).astype(np.uint8) if seed % 4 == 1 else None
    if seed == 9:
        inliers = np.zeros(n, np.uint8)
    return dict(points=points, model=model, params=np.array(base), qvec=np.array([1.0, 0, 0, 0]), tvec=np.zeros(3),
                corners=np.full((n_patches, 2), 490, np.int32), scales=np.ones((n_patches, 2)),
                refine=[int(x) for x in rng.integers(0, 2, 3)] if seed % 2 else [0, 0, 0], mode=mode, ref_count=ref_count,
                patch_idxs=patch_idxs, inliers=inliers)


def nearest_case(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 9))
    patches = rng.normal(size=(n, 16, 16, 128)).astype(np.float16)
    corners = rng.integers(0, 300, size=(n, 2)).astype(np.int32)
    scales = np.tile(rng.choice([1.0, 0.5], 2), (n, 1))
    kp = (corners + 0.5 + rng.uniform(3, 13, (n, 2))) / scales
    cand_count = rng.integers(1, 6, n).astype(np.int32)
    cand = rng.normal(size=(int(cand_count.sum()), 128))
    cand /= np.linalg.norm(cand, axis=1, keepdims=True)
    return dict(patches=patches, corners=corners, scales=scales, kp=kp, cand_count=cand_count, cand=cand, l2=int(seed % 2 == 0))
