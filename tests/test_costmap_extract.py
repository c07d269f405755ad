"""Cost-map extraction (CostMapExtractor::FillPointCostmap, bundle_adjustment/src/costmap_extractor.h:230-358) on
the seeded patches of tests/cases/costmap_cases.py -- raw-texel branch and interpolating branch, trivial / Cauchy / Huber loss,
sqrt variants, 1 / 3 / 4 channels, fp16 / fp32 / fp64 storage incl. FeaturePatch::SetEntry's cast through float for half.
GPU: pxr_costmap_extract / pxr_costmap_extract_ex against the oracle's numpy restatement (oracle/pxo_costmap.py), which is
itself checked on the CPU in tests/test_oracle_costmap.py (the reference's vendored half.hpp compiled from its own source for
the two fp16 rounding rules; an independent per-texel loop; finite differences).
Bar: the storage type's -- identical bits except where the fp64 summation order moves a value across a rounding boundary
(at most 1 ulp, rarely); fp64 maps within 1e-12.
PARITY UNPINNED beyond the fp16 rules: the reference has no test or vector for this extractor and it cannot be compiled here."""
import numpy as np
import pytest

from cases import costmap_cases as gen_mod


def _gen():
    return gen_mod


def _ulps(a, b):
    it = {2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
    ia, ib = a.view(it).astype(np.int64), b.view(it).astype(np.int64)
    sign = np.int64(1) << (8 * a.dtype.itemsize - 1)
    ia = np.where(ia < 0, -(ia + sign), ia)
    ib = np.where(ib < 0, -(ib + sign), ib)
    return np.abs(ia - ib)


def _check(got, want, name):
    assert got.dtype == want.dtype and got.shape == want.shape, name
    if got.dtype == np.float64:
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), name
        return 0
    d = _ulps(got, want)
    # entries that are differences of nearly equal sums (a texel that IS the reference) carry no relative accuracy
    d = np.where(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= 1e-12 * np.abs(want).max(), 0, d)
    assert d.max() <= 1, (name, "more than one ulp apart")
    return int((d > 0).sum())


def _oracle(c):
    import pxo
    import pxo_costmap
    if c["up"] == 1.0 and not c["cross"]:
        return pxo_costmap.fill_point_costmap(c["patch"], c["ref"], c["loss"], c["grad"], c["sqrt"], c["out_dtype"])
    return pxo_costmap.fill_point_costmap_interpolated(c["patch"], c["ref"], pxo.cfg(c["l2"], False, False), c["loss"], c["grad"],
                                                       c["sqrt"], c["up"], c["cross"], c["out_dtype"])


@pytest.mark.gpu
def test_hip_costmaps_match_the_oracle():
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss
    gen = _gen()
    ctx = Context(0)
    groups = {}
    for c in gen.cases():
        key = (c["patch"].dtype, c["patch"].shape, c["loss"], c["grad"], c["sqrt"], c["out_dtype"], c["up"], c["cross"], c["l2"])
        groups.setdefault(key, []).append(c)
    n_off = n_entries = n_maps = 0
    for (dt, shape, loss, grad, sq, od, up, cross, l2), cs in groups.items():
        m = len(cs)
        ids = np.arange(m, dtype=np.int32)
        q = np.tile([1.0, 0, 0, 0], (m, 1))
        prob = dict(obs_image=ids, obs_point=ids, obs_patch=np.arange(m, dtype=np.int64), image_camera=ids, qvec=q,
                    tvec=np.zeros((m, 3)), cam_model=np.zeros(m, np.int32), cam_params=np.tile([500.0, 8, 8] + [0.0] * 9, (m, 1)),
                    xyz=np.tile([0.0, 0, 2.0], (m, 1)), refs=np.stack([c["ref"] for c in cs]),
                    patches=np.stack([c["patch"] for c in cs]), corners=np.zeros((m, 2), np.int32), scales=np.ones((m, 2)))
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(ctx, arena, prob)
        cm = ba.extract_costmaps(make_loss(loss[0], [] if loss[0] == "trivial" else [loss[1]]), as_gradientfield=grad, apply_sqrt=sq,
                                 dtype=od, upsampling_factor=up, compute_cross_derivative=cross, cfg=interp_cfg(l2_normalize=l2))
        got = cm.download()[0]
        for i, c in enumerate(cs):
            want = _oracle(c)
            n_off += _check(np.ascontiguousarray(got[i]), want, c["name"])
            n_entries += want.size
            n_maps += 1
    assert n_maps == len(gen.cases())
    assert n_off <= 2e-3 * n_entries, (n_off, n_entries)
