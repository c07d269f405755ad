"""Cost-map extraction pinned against the REFERENCE's own code: tests/golden/costmap_ref.npz holds what
CostMapExtractor::FillPointCostmap (bundle_adjustment/src/costmap_extractor.h:230-358, compiled in place:
tests/golden/make_golden_costmap.py, oracle/ref_costmap_shim.cc) writes for seeded patches -- raw-texel branch and interpolating
branch, trivial / Cauchy / Huber loss (the rho formulas under it are restated from the published Ceres definitions), sqrt
variants, 1 / 3 / 4 channels, fp16 / fp32 / fp64 storage incl. FeaturePatch::SetEntry's cast through float for half.
Checked here: the oracle's numpy restatement (CPU) and pxr_costmap_extract / pxr_costmap_extract_ex (GPU).
Bar: the storage type's -- identical bits except where the fp64 summation order moves a value across a rounding boundary
(at most 1 ulp, rarely); fp64 maps within 1e-12."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_costmap", os.path.join(HERE, "golden", "make_golden_costmap.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _gold(c, gold):
    g = gold[c["name"]]
    return g.view(np.float16) if c["out_dtype"] == np.float16 else g


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


def test_oracle_costmaps_match_the_reference_vectors():
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "costmap_ref.npz"))
    n_off = n_entries = 0
    for c in gen.cases():
        got, want = _oracle(c), _gold(c, gold)
        n_off += _check(got, want, c["name"])
        n_entries += want.size
    assert n_off <= 2e-3 * n_entries, (n_off, n_entries)


def test_reference_run_live_when_present():
    gen = _gen()
    if not os.path.exists(gen.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_costmap.so not built (reference tree absent)")
    gold = np.load(os.path.join(HERE, "golden", "costmap_ref.npz"))
    for c in gen.cases()[::5]:
        out = gen.run_reference(c)
        want = _gold(c, gold)
        assert out.tobytes() == want.tobytes(), c["name"]


@pytest.mark.gpu
def test_hip_costmaps_match_the_reference_vectors():
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "costmap_ref.npz"))
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
            want = _gold(c, gold)
            n_off += _check(np.ascontiguousarray(got[i]), want, c["name"])
            n_entries += want.size
            n_maps += 1
    assert n_maps == len(gen.cases())
    assert n_off <= 2e-3 * n_entries, (n_off, n_entries)
