"""Pure-Python restatement of the reference's query-refinement problem construction (TEST INFRASTRUCTURE ONLY --
never imported by the product package):
  qka_setup   SingleQueryKeypointOptimizer::RunQuery, the three reference-container overloads
              (localization/src/single_query_keypoint_optimizer.h:82-196) + QueryKeypointOptimizer::ParameterizeKeypoint
              (query_keypoint_optimizer.h:137-165) + SolveProblem's empty-problem return (:67-70)
  qba_setup   SingleQueryBundleOptimizer::RunQuery, the three overloads (single_query_bundle_optimizer.h:91-215) +
              QueryBundleOptimizer::ParameterizeQuery (query_bundle_optimizer.h:114-150) + SolveProblem (:72-75)
FindNearestReferences (nearest_references.h:20-52) is pxo.nearest_reference.
A descriptor is identified by a (correspondence, k) pair: k = -1 the correspondence's own descriptor, k >= 0 its k-th
per-observation descriptor.
PARITY UNPINNED: the reference has no test for these optimizers and they cannot be compiled here (Ceres / COLMAP absent)."""
import numpy as np

import pxo_ba_setup


def _walk(n, mode, ref_count, inliers):
    """(idx, k) of every residual block in the order the reference adds them, and the correspondences that get parameterised."""
    blocks, visited = [], []
    for idx in range(n):
        if inliers is not None and not inliers[idx]:
            continue
        if mode == 0:                                   # vector<Ref<DescriptorMatrixXd>>: one descriptor per correspondence
            blocks.append((idx, -1)); visited.append(idx)
        elif mode == 1:                                 # vector<vector<DescriptorMatrixXd>>: parameterised only if one was added
            for k in range(int(ref_count[idx])):
                blocks.append((idx, k))
            if ref_count[idx] > 0:
                visited.append(idx)
        else:                                           # vector<Reference>: own descriptor unless it has per-observation ones
            if ref_count[idx] == 0:
                blocks.append((idx, -1))
            else:
                for k in range(int(ref_count[idx])):
                    blocks.append((idx, k))
            visited.append(idx)
    return blocks, visited


def qka_setup(kp, corners, scales, width, height, sparse, bound, mode, ref_count, patch_idxs=None, inliers=None):
    """-> dict(solved, blocks [(idx, k)], lower [n, 2], upper [n, 2] (NaN where ParameterizeKeypoint set nothing))."""
    n = len(kp)
    blocks, visited = _walk(n, mode, ref_count, inliers)
    lower, upper = np.full((n, 2), np.nan), np.full((n, 2), np.nan)
    for idx in visited:
        if not (bound > 0.0 or sparse):
            continue
        p = idx if patch_idxs is None else int(patch_idxs[idx])
        sx, sy = float(scales[p][0]), float(scales[p][1])
        lx, ly = (corners[p][0] + 0.5) / sx, (corners[p][1] + 0.5) / sy
        ux, uy = lx + width / sx, ly + height / sy
        if bound > 0.0:
            ux, uy = min(kp[idx][0] + bound / sx, ux), min(kp[idx][1] + bound / sy, uy)
            lx, ly = max(kp[idx][0] - bound / sx, lx), max(kp[idx][1] - bound / sy, ly)
        lower[idx], upper[idx] = (lx, ly), (ux, uy)
    return dict(solved=len(blocks) > 0, blocks=blocks, lower=lower, upper=upper)


def qba_setup(n, model, refine_focal, refine_pp, refine_extra, mode, ref_count, inliers=None):
    """-> dict(solved, blocks [(idx, k)], point_const [n] (every inlier point is held constant, with or without a residual),
    camera_mask: bit mask of constant camera parameters (all bits when nothing is refined))."""
    blocks, _ = _walk(n, mode, ref_count, inliers)
    point_const = np.ones(n, bool) if inliers is None else np.asarray(inliers).astype(bool)
    if not refine_focal and not refine_pp and not refine_extra:
        mask = (1 << pxo_ba_setup.NUM_PARAMS[model]) - 1
    else:
        idx = (([] if refine_focal else pxo_ba_setup.FOCAL[model]) + ([] if refine_pp else pxo_ba_setup.PRINCIPAL[model]) +
               ([] if refine_extra else pxo_ba_setup.EXTRA[model]))
        mask = sum(1 << k for k in idx)
    return dict(solved=len(blocks) > 0, blocks=blocks, point_const=point_const, camera_mask=mask)
