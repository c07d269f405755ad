"""ORACLE (test infrastructure only): numpy restatement of the reference's cost-map extraction,
CostMapExtractor::FillPointCostmap (pixsfm/bundle_adjustment/src/costmap_extractor.h:230-358), for the
case the reference takes without interpolation (cost patch of the feature patch's size, no cross
derivative, :253-279 / :330-340): per texel (y, x)

    f      = patch[y, x, :] as double                       (raw texel: NOT l2-normalised, :257-259)
    res    = f - reference descriptor                        (:286-288)
    cost   = 0.5 * rho(|res|^2)[0]                           (:290-294, CostMapConfig.loss, default Trivial)
    dfdr   = 0.5 * (patch[min(H-1, y+1), x] - patch[max(0, y-1), x])      central differences, the
    dfdc   = 0.5 * (patch[y, min(W-1, x+1)] - patch[y, max(0, x-1)])      subtraction in the STORAGE type
                                                             (Eigen expression on Map<Matrix<dtype>>, :266-279;
                                                             half 2.2.0 rounds it correctly, like one rounding
                                                             of the exact difference)
    dcost/dr = rho' * <res, dfdr>, dcost/dc = rho' * <res, dfdc>   only where cost > 1e-8 (:300-302)
    apply_sqrt: cost <- sqrt(cost), d <- d * 0.5 / cost       (inside the same cost > 1e-8 branch, :309-317;
                                                             always in the 1-channel branch, :351-353)
    entries are cast to the cost patch's dtype (FeaturePatch::SetEntry, featurepatch.h:246-248: `dtype(value)`;
    for half that is half(float(value)) -- half.hpp has no constructor from double -- i.e. TWO roundings).

The cost patch copies corner and scale of the feature patch (CreateShallowCostmapFSet, :382-399).
"parity unpinned": the reference has no test or golden vector for cost maps; the two storage-type rounding
rules above are pinned against the reference's vendored half.hpp (oracle/_ref, tests/test_oracle_costmap.py).
"""
import numpy as np


def _rho(loss, s):
    """[upstream ceres/loss_function.cc] rho(s), rho'(s) for (name, a)."""
    name, a = loss
    if name == "trivial":
        return s.copy(), np.ones_like(s)
    b = a * a
    if name == "cauchy":
        return b * np.log1p(s / b), 1.0 / (1.0 + s / b)
    if name == "huber":
        r = np.sqrt(s)
        big = s > b
        rho0 = np.where(big, 2.0 * a * r - b, s)
        rho1 = np.where(big, np.maximum(np.finfo(np.float64).tiny, a / np.where(big, r, 1.0)), 1.0)
        return rho0, rho1
    if name == "soft_l1":
        t = 1.0 + s / b
        return 2.0 * b * (np.sqrt(t) - 1.0), np.maximum(np.finfo(np.float64).tiny, 1.0 / np.sqrt(t))
    raise ValueError(name)


def storage_diff(a, b):
    """a - b in the storage type: one correct rounding of the exact difference (exact in double for fp16 / fp32)."""
    if a.dtype == np.float64:
        return a - b
    return (a.astype(np.float64) - b.astype(np.float64)).astype(a.dtype)


def store(values, out_dtype):
    """FeaturePatch::SetEntry's dtype(double): through float for half storage (see the module docstring)."""
    out_dtype = np.dtype(out_dtype)
    if out_dtype == np.float16:
        with np.errstate(over="ignore"):
            return values.astype(np.float32).astype(np.float16)
    return values.astype(out_dtype)


def fill_point_costmap(patch, ref, loss=("trivial", 1.0), as_gradientfield=True, apply_sqrt=False, out_dtype=None):
    """patch: (H, W, C) fp16/fp32/fp64; ref: (C,) double -> (H, W, 3 or 1) cost map of dtype out_dtype
    (default: the patch's, like the reference's Run<dtype, dtype> binding, bundle_adjustment/bindings.cc:21)."""
    patch = np.asarray(patch)
    H, W, _ = patch.shape
    out_dtype = patch.dtype if out_dtype is None else np.dtype(out_dtype)
    res = patch.astype(np.float64) - np.asarray(ref, dtype=np.float64)
    rho0, rho1 = _rho(loss, np.einsum("hwc,hwc->hw", res, res))
    cost = 0.5 * rho0
    if not as_gradientfield:
        if apply_sqrt:
            cost = np.sqrt(cost)
        return store(cost[:, :, None], out_dtype)
    yy, xx = np.arange(H), np.arange(W)
    top, bottom = np.minimum(H - 1, yy + 1), np.maximum(0, yy - 1)
    right, left = np.minimum(W - 1, xx + 1), np.maximum(0, xx - 1)
    dfdr = 0.5 * storage_diff(patch[top], patch[bottom]).astype(np.float64)
    dfdc = 0.5 * storage_diff(patch[:, right], patch[:, left]).astype(np.float64)
    on = cost > 1.0e-8
    dr = np.where(on, rho1 * np.einsum("hwc,hwc->hw", res, dfdr), 0.0)
    dc = np.where(on, rho1 * np.einsum("hwc,hwc->hw", res, dfdc), 0.0)
    if apply_sqrt:
        root = np.sqrt(np.where(on, cost, 1.0))
        dr = np.where(on, dr * 0.5 / root, dr)
        dc = np.where(on, dc * 0.5 / root, dc)
        cost = np.where(on, root, cost)
    return store(np.stack([cost, dr, dc], axis=-1), out_dtype)


def costmaps(patches, obs_patch, obs_point, refs, **kw):
    """One cost map per observation (CostMapExtractor::RunSubset, costmap_extractor.h:192-224):
    observation i -> fill_point_costmap(patches[obs_patch[i]], refs[obs_point[i]])."""
    return np.stack([fill_point_costmap(patches[p], refs[q], **kw) for p, q in zip(obs_patch, obs_point)])
