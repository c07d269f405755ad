"""CPU: libpixsfm_h5.so (the native reader of pixsfm's dense-feature cache) against the REFERENCE's own
reader -- features/src/featuremanager.cc, featureset.cc, featuremap.cc, featurepatch.cc and util/src/misc.h compiled in
place (oracle/Makefile -> oracle/_ref/libpxo_ref_h5.so; HighFive, an empty submodule of the checkout, is replaced by a
stand-in over the image's libhdf5).
  * golden: committed cache files + what the reference's reader handed out for them (tests/golden/make_golden_h5.py);
  * live (when the library is present): both readers on freshly written random caches, the subset / on-demand path
    (FeatureSet::Load(required_patches)), and the error behaviour.
What stays unpinned: the WRITER is tests/h5_writer.py's restatement of store_features.py (h5py is absent offline)."""
import os

import numpy as np
import pytest

import h5_writer
import ref_h5_reader

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
pytestmark = pytest.mark.skipif(not h5_writer.available(), reason="the image's libhdf5 is missing")
live = pytest.mark.skipif(not ref_h5_reader.available(), reason="oracle/_ref/libpxo_ref_h5.so not built")
NAMES = ["sparse_%s_%s" % (f, d) for f in ("chunked", "grouped") for d in ("half", "float", "double")] + ["dense_as_sparse", "dense"]


def _mine(path, level_prefix):
    """the product reader's view of a file, in the shape make_golden_h5.dump_reference() gives the reference's"""
    from pixsfm_amd import _h5
    out = {}
    with _h5.FeatureCache(path, level_prefix) as cache:
        out["channels_per_level"] = np.array(cache.channels_per_level, np.int32)
        out["dtype"] = cache.dtype
        for l in range(cache.num_levels):
            names = sorted(cache.image_names(l))
            out["%d/images" % l] = np.array(names)
            for im in names:
                info = cache.map_info(l, im)
                ids, corners, scales = cache.map_meta(l, im, info["n"])
                patches = cache.read_patches(l, im)
                order = np.argsort(ids, kind="stable")
                key = "%d/%s/" % (l, im)
                out[key + "is_sparse"] = np.array(info["is_sparse"])
                out[key + "channels"] = np.array(info["shape"][2])
                out[key + "ids"] = np.asarray(ids)[order].astype(np.uint32)
                out[key + "corners"] = np.asarray(corners)[order]
                out[key + "scales"] = np.asarray(scales)[order]
                out[key + "patches"] = patches[order]
    return out


def _same(ref, mine, grouped):
    """grouped (format 1) maps: the reference's FeatureMap::Channels() is 0 -- store_features.py:24-26 writes a THREE-element
    "shape" attribute (H, W, C) and featuremap.cc:100-104 reads element [3] of the vector it was read into (left at the 0 it
    was constructed with); the patches carry their own shape, so nothing downstream notices.  The product reports the real
    channel count there; everything else must be equal."""
    assert sorted(k for k in mine if k != "dtype") == sorted(ref)
    for k, v in ref.items():
        m = mine[k]
        if grouped and k.endswith("/channels"):
            assert int(v) == 0 and int(m) == ref[k[:-len("channels")] + "patches"].shape[-1], k
            continue
        assert np.asarray(m).shape == np.asarray(v).shape, k
        if k.endswith("patches"):
            assert m.dtype == v.dtype, k
        assert np.array_equal(np.asarray(m), np.asarray(v)), k


@pytest.mark.parametrize("name", NAMES)
def test_committed_cache_files_read_like_the_reference_reads_them(name):
    gold = np.load(os.path.join(GOLD, "h5_cache_ref.npz"))
    ref = {k.split("|", 1)[1]: gold[k] for k in gold.files if k.startswith(name + "|")}
    prefix = str(ref.pop("level_prefix"))
    _same(ref, _mine(os.path.join(GOLD, "h5_cache_%s.h5" % name), prefix), "grouped" in name)


def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_h5", os.path.join(GOLD, "make_golden_h5.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@live
def test_golden_vectors_are_what_the_reference_reader_yields_now():
    g = _golden_module()
    gold = np.load(os.path.join(GOLD, "h5_cache_ref.npz"))
    for name, _, kw in g.cases():
        now = g.dump_reference(os.path.join(GOLD, "h5_cache_%s.h5" % name), g.DT[kw["dtype_name"]], kw["level_prefix"])
        for k, v in now.items():
            assert np.array_equal(v, gold[name + "|" + k]), (name, k)


def _random_map(rng, n, ps, ch, dtype):
    ids = rng.choice(100000, n, replace=False)
    scale = rng.uniform(0.1, 1.0, 2)
    return dict(keypoint_ids=[int(k) for k in ids], patches=rng.normal(size=(n, ps, ps, ch)).astype(dtype),
                corners=rng.integers(0, 4000, size=(n, 2)), scales=[rng.uniform(0.1, 1.0, 2) for _ in range(n)],
                metadata={"is_sparse": True, "scale": scale, "patch_size": ps})


@live
@pytest.mark.parametrize("seed", range(6))
def test_random_caches_both_readers(tmp_path, seed):
    g = _golden_module()
    rng = np.random.default_rng(100 + seed)
    name = ["half", "float", "double"][seed % 3]
    dtype = g.DT[name]
    fmt = ["chunked", "grouped"][seed // 3 % 2]
    images = ["%s%d%s" % (rng.choice(["", "a/", "a/b/", "mapping/"]), i, rng.choice([".jpg", ".png", ".JPG", ".jpeg", ".JPEG"])) for i in range(int(rng.integers(1, 7)))]
    n_levels = int(rng.integers(1, 4))
    chans = [int(rng.choice([1, 3, 16, 64, 128])) for _ in range(n_levels)]
    sizes = [int(rng.choice([1, 4, 10, 16])) for _ in range(n_levels)]
    counts = {im: int(rng.integers(1, 30)) for im in images}
    levels = [{im: _random_map(rng, counts[im], sizes[l], chans[l], dtype) for im in images} for l in range(n_levels)]
    prefix = str(rng.choice(["", "lvl", "s"]))
    path = tmp_path / "c.h5"
    h5_writer.write_cache(path, levels, dtype_name=name, cache_format=fmt, level_prefix=prefix)
    mine = _mine(path, prefix)
    assert mine["dtype"] == np.dtype(dtype)
    _same(g.dump_reference(path, dtype, prefix), mine, fmt == "grouped")


@live
@pytest.mark.parametrize("cache_format", ["chunked", "grouped"])
def test_subset_of_patches_on_demand(tmp_path, cache_format):
    """FeatureManager(fill=False) then FeatureSet::Load(required_patches, fill=True) (featureset.cc:90-143,
    featuremap.cc:217-267 / :92-132) == pxr_h5_read_patches(which) == load_features_from_cache(required=...)."""
    from pixsfm_amd import _h5
    from pixsfm_amd.api import features
    rng = np.random.default_rng(7)
    level = {"im0.jpg": _random_map(rng, 12, 8, 16, np.float16), "im1.jpg": _random_map(rng, 5, 8, 16, np.float16)}
    path = tmp_path / "c.h5"
    h5_writer.write_cache(path, [level], cache_format=cache_format)
    want = {"im0.jpg": [level["im0.jpg"]["keypoint_ids"][i] for i in (7, 0, 3)], "im1.jpg": [level["im1.jpg"]["keypoint_ids"][4]]}
    with ref_h5_reader.ReferenceCache(path, np.float16, fill=False) as ref, _h5.FeatureCache(path) as cache:
        fmgr = features.load_features_from_cache(path, required=want)
        for im, ids in want.items():
            if cache_format == "chunked":      # metadata is there before any data is (InitFromH5GroupChunked with fill = false)
                assert ref.map_info(0, im)["n"] == len(level[im]["keypoint_ids"]) and ref.patch(0, im, ids[0])["data"] is None
            nbytes = ref.load(0, im, ids, fill=True)
            assert nbytes == len(ids) * 8 * 8 * 16 * 2
            stored, corners, scales = cache.map_meta(0, im, len(level[im]["keypoint_ids"]))
            pos = {int(k): i for i, k in enumerate(stored)}
            got = cache.read_patches(0, im, np.array([pos[k] for k in ids]))
            for j, k in enumerate(ids):
                rp = ref.patch(0, im, k)
                assert np.array_equal(rp["data"], got[j]) and np.array_equal(rp["corner"], corners[pos[k]]) and np.array_equal(rp["scale"], scales[pos[k]])
                fp = fmgr.fset(0).fmap(im).fpatch(k)
                assert np.array_equal(fp.data, rp["data"]) and np.array_equal(fp.corner, rp["corner"]) and np.array_equal(fp.scale, rp["scale"])
            assert sorted(fmgr.fset(0).fmap(im).keys()) == sorted(ids)
            if cache_format == "chunked":      # the others stay without data in the reference, and are absent in the product's manager
                other = next(k for k in level[im]["keypoint_ids"] if k not in ids)
                assert ref.patch(0, im, other)["data"] is None


@live
def test_both_readers_refuse_the_same_malformed_files(tmp_path):
    from pixsfm_amd import _h5
    from pixsfm_amd._lib import PixsfmHipError
    rng = np.random.default_rng(9)
    with pytest.raises(ref_h5_reader.ReferenceError_):
        ref_h5_reader.ReferenceCache(tmp_path / "nope.h5", np.float16)
    with pytest.raises(PixsfmHipError):
        _h5.FeatureCache(tmp_path / "nope.h5")
    # level group missing: channels_per_level announces two levels, only one is stored
    path = tmp_path / "short.h5"
    h5_writer.write_cache(path, [{"a.jpg": _random_map(rng, 2, 4, 8, np.float16)}], channels_per_level=[8, 8])
    with pytest.raises(ref_h5_reader.ReferenceError_):
        ref_h5_reader.ReferenceCache(path, np.float16)
    with pytest.raises(PixsfmHipError):
        _h5.FeatureCache(path)
    # unknown storage format (featuremap.cc:60-75 "Unknown featuremap format.")
    path = tmp_path / "fmt.h5"
    h5_writer.write_cache(path, [{"a.jpg": _random_map(rng, 2, 4, 8, np.float16)}], format_override=3)
    with pytest.raises(ref_h5_reader.ReferenceError_, match="Unknown featuremap format"):
        ref_h5_reader.ReferenceCache(path, np.float16)
    with pytest.raises(PixsfmHipError, match="format"):
        with _h5.FeatureCache(path) as cache:
            cache.map_info(0, "a.jpg")


@live
@pytest.mark.parametrize("n_ids", [1, 2, 5])
@pytest.mark.parametrize("dtype,name", [(np.float16, "half"), (np.float32, "float")])
def test_dense_maps_both_readers(tmp_path, n_ids, dtype, name):
    """ONE dense map stored per image; several keypoint ids => loaded as patch_size windows at the stored corners
    (featuremap.cc:157-165,246-256), a single id => a true dense map under kDensePatchId."""
    g = _golden_module()
    rng = np.random.default_rng(40 + n_ids)
    ps, h, w, c = 6, 23, 31, 4
    dense = rng.normal(size=(1, h, w, c)).astype(dtype)
    ids = [1000000] if n_ids == 1 else [int(k) for k in rng.choice(500, n_ids, replace=False)]
    corners = np.stack([rng.integers(0, w - ps + 1, n_ids), rng.integers(0, h - ps + 1, n_ids)], 1) if n_ids > 1 else np.array([[0, 0]])
    fm = dict(keypoint_ids=ids, patches=dense, corners=corners, scales=[np.array([0.5, 0.25])] * n_ids,
              metadata={"is_sparse": False, "patch_size": ps, "scale": np.array([0.5, 0.25])})
    path = tmp_path / "d.h5"
    h5_writer.write_cache(path, [{"a.jpg": fm, "b.png": fm}], dtype_name=name)
    ref = g.dump_reference(path, dtype, "")
    assert bool(ref["0/a.jpg/is_sparse"]) == (n_ids > 1) and ref["0/a.jpg/patches"].shape[1:] == ((ps, ps, c) if n_ids > 1 else (h, w, c))
    _same(ref, _mine(path, ""), False)
