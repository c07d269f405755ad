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
Parity: the two storage-type rounding rules above are pinned against the reference's vendored half.hpp compiled from its own
source (oracle/ref_half_shim.cc -> oracle/_ref/libpxo_ref_half.so, tests/test_oracle_costmap.py).  Everything else is PARITY
UNPINNED: the reference has no test or vector for the extractor and costmap_extractor.h cannot be compiled here; the
vectorised code below is checked against a statement-by-statement loop over the same source lines and by finite differences.
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


def cost_patch_shape(H, W, upsampling_factor, channels):
    """CreateShallowCostmapFSet (costmap_extractor.h:385-390): int(side * (upsampling_factor + 1e-6))."""
    return int(H * (upsampling_factor + 1.0e-6)), int(W * (upsampling_factor + 1.0e-6)), channels


def fill_point_costmap_interpolated(patch, ref, config, loss=("trivial", 1.0), as_gradientfield=True, apply_sqrt=False,
                                    upsampling_factor=1.0, compute_cross_derivative=False, out_dtype=None):
    """The branch of FillPointCostmap that INTERPOLATES (costmap_extractor.h:280-284,341-345): taken when the cost
    patch is not of the feature patch's size (CostMapConfig.upsampling_factor != 1) or compute_cross_derivative is set.
    Per output texel (y, x): xy = (x, y) / upsampling_factor in LOCAL patch coordinates, PatchInterpolator::EvaluateLocal
    = PixelInterpolator::Evaluate(r = xy[1], c = xy[0]) with the extractor's InterpolationConfig (l2_normalize applies
    here, unlike the no-interpolation branch), then the same cost / loss / derivative rules; the cross term follows
    :304-316.  Statement-by-statement loop over the oracle's C interpolation (small inputs only)."""
    import pxo
    patch = np.ascontiguousarray(patch)
    H, W, C = patch.shape
    out_dtype = patch.dtype if out_dtype is None else np.dtype(out_dtype)
    co = (4 if compute_cross_derivative else 3) if as_gradientfield else 1
    Ho, Wo, _ = cost_patch_shape(H, W, upsampling_factor, co)
    P = pxo.make_patch(patch)
    ref = np.asarray(ref, dtype=np.float64)
    out = np.zeros((Ho, Wo, co))
    scale = 1.0 / upsampling_factor
    for y in range(Ho):
        for x in range(Wo):
            if as_gradientfield:
                f, dfdr, dfdc, dfdrc = pxo.pixel_interp_cross(P, y * scale, x * scale, config)
            else:
                f = pxo.pixel_interp(P, y * scale, x * scale, config)[0]
            res = f - ref
            rho0, rho1 = _rho(loss, np.array([res @ res]))
            cost = 0.5 * rho0[0]
            if not as_gradientfield:
                out[y, x, 0] = np.sqrt(cost) if apply_sqrt else cost
                continue
            dr = dc = drc = 0.0
            if cost > 1.0e-8:
                dr, dc = rho1[0] * (res @ dfdr), rho1[0] * (res @ dfdc)
                if compute_cross_derivative:
                    rho2 = _rho2(loss, res @ res)
                    drc = rho2 * 2.0 * (res @ dfdr) * (res @ dfdc) + rho1[0] * (dfdr @ dfdc + dfdrc @ res)
                if apply_sqrt:
                    cost = np.sqrt(cost)
                    if compute_cross_derivative:
                        drc = drc * 0.5 / cost - 0.25 / (cost ** 3) * dr * dc
                    dr, dc = dr * 0.5 / cost, dc * 0.5 / cost
            out[y, x, :3] = cost, dr, dc
            if compute_cross_derivative:
                out[y, x, 3] = drc
    return store(out, out_dtype)


def _rho2(loss, s):
    """[upstream ceres/loss_function.cc] rho''(s)."""
    name, a = loss
    b = a * a
    if name == "trivial":
        return 0.0
    if name == "cauchy":
        return -1.0 / (b * (1.0 + s / b) ** 2)
    if name == "huber":
        return -(a / np.sqrt(s)) / (2.0 * s) if s > b else 0.0
    if name == "soft_l1":
        t = 1.0 + s / b
        return -1.0 / (2.0 * b * t * np.sqrt(t))
    raise ValueError(name)


def costmaps(patches, obs_patch, obs_point, refs, **kw):
    """One cost map per observation (CostMapExtractor::RunSubset, costmap_extractor.h:192-224):
    observation i -> fill_point_costmap(patches[obs_patch[i]], refs[obs_point[i]])."""
    return np.stack([fill_point_costmap(patches[p], refs[q], **kw) for p, q in zip(obs_patch, obs_point)])
