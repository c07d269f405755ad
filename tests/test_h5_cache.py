"""CPU: the native reader of pixsfm's dense-feature cache (libpixsfm_h5.so) against files written the
way the reference's Python writes them (tests/h5_writer.py restates extract.py:98-127 + store_features.py on the image's
libhdf5; h5py itself is absent).  Reader and writer are independent implementations of the two sides of the documented
layout; more reader cases (committed files, h5dump, subsets, malformed files) in tests/test_h5_reader.py."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import h5_writer

pytestmark = pytest.mark.skipif(not h5_writer.available(), reason="the image's libhdf5 is missing")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fmap(rng, n, ps, ch, dtype, ids=None):
    patches = rng.normal(size=(n, ps, ps, ch)).astype(dtype)
    ids = rng.choice(5000, n, replace=False) if ids is None else ids
    corners = rng.integers(0, 900, size=(n, 2))
    scale = np.array([0.25, 0.5])
    return dict(keypoint_ids=[int(k) for k in ids], patches=patches, corners=corners, scales=[scale for _ in range(n)],
                metadata={"is_sparse": True, "scale": scale, "patch_size": ps})


def test_c_abi_exports_every_declared_symbol():
    from pixsfm_amd import _h5
    header = open(os.path.join(ROOT, "include", "pixsfm_h5.h")).read()
    declared = set(re.findall(r"\b(pxr_h5_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_h5.declared_symbols())
    lib = _h5.load()
    for name in declared:
        assert hasattr(lib, name)


@pytest.mark.parametrize("cache_format", ["chunked", "grouped"])
@pytest.mark.parametrize("dtype,name", [(np.float16, "half"), (np.float32, "float"), (np.float64, "double")])
def test_sparse_cache_round_trip(tmp_path, cache_format, dtype, name):
    from pixsfm_amd import _h5
    rng = np.random.default_rng(0)
    levels = [{"im0.jpg": _fmap(rng, 7, 16, 128, dtype), "seq/a/im1.png": _fmap(rng, 3, 16, 128, dtype), "z.JPG": _fmap(rng, 1, 16, 128, dtype)},
              {"im0.jpg": _fmap(rng, 7, 8, 64, dtype), "seq/a/im1.png": _fmap(rng, 3, 8, 64, dtype), "z.JPG": _fmap(rng, 1, 8, 64, dtype)}]
    path = tmp_path / "cache.h5"
    h5_writer.write_cache(path, levels, dtype_name=name, cache_format=cache_format, level_prefix="lvl")
    with _h5.FeatureCache(path, "lvl") as cache:
        assert cache.num_levels == 2 and cache.channels_per_level == [128, 64] and cache.dtype == np.dtype(dtype)
        for l, level in enumerate(levels):
            assert sorted(cache.image_names(l)) == sorted(level)          # GetImageKeys walks nested groups
            for image, fm in level.items():
                info = cache.map_info(l, image)
                n = len(fm["keypoint_ids"])
                assert info == dict(format=2 if cache_format == "chunked" else 1, is_sparse=True, n=n, shape=fm["patches"].shape[1:])
                ids, corners, scales = cache.map_meta(l, image, n)
                order = [fm["keypoint_ids"].index(int(k)) for k in ids]   # grouped: datasets come back in name order
                assert sorted(order) == list(range(n))
                assert np.array_equal(corners, fm["corners"][order]) and np.array_equal(scales, np.array(fm["scales"])[order])
                got = cache.read_patches(l, image)
                assert got.dtype == dtype and np.array_equal(got, fm["patches"][order])
                sub = np.array([n - 1, 0, n // 2])
                assert np.array_equal(cache.read_patches(l, image, sub), fm["patches"][order][sub])


def test_dense_map_stored_once_loaded_as_sparse_windows(tmp_path):
    """extractor.py:213-217 / featuremap.cc:157-165,246-256: ONE dense map per image in the file, several keypoint ids,
    loaded as patch_size windows at the stored corners."""
    from pixsfm_amd import _h5
    rng = np.random.default_rng(1)
    dense = rng.normal(size=(1, 60, 80, 32)).astype(np.float16)
    ids = [3, 10, 11, 40]
    corners = np.array([[0, 0], [64, 44], [13, 7], [30, 44]])            # (x, y), windows of 16
    fm = dict(keypoint_ids=ids, patches=dense, corners=corners, scales=[np.array([0.5, 0.5])] * 4,
              metadata={"is_sparse": False, "patch_size": 16, "scale": np.array([0.5, 0.5])})
    path = tmp_path / "dense.h5"
    h5_writer.write_cache(path, [{"a.jpg": fm}])
    with _h5.FeatureCache(path) as cache:
        info = cache.map_info(0, "a.jpg")
        assert info == dict(format=2, is_sparse=True, n=4, shape=(16, 16, 32))
        got = cache.read_patches(0, "a.jpg")
        for k, (x, y) in enumerate(corners):
            assert np.array_equal(got[k], dense[0, y:y + 16, x:x + 16])


def test_true_dense_map_and_errors(tmp_path):
    from pixsfm_amd import _h5
    from pixsfm_amd._lib import PixsfmHipError
    rng = np.random.default_rng(2)
    dense = rng.normal(size=(1, 30, 40, 16)).astype(np.float32)
    fm = dict(keypoint_ids=[1000000], patches=dense, corners=np.array([[0, 0]]), scales=[np.array([0.25, 0.25])],
              metadata={"is_sparse": False, "scale": np.array([0.25, 0.25])})
    path = tmp_path / "d.h5"
    h5_writer.write_cache(path, [{"a.jpg": fm}], dtype_name="float")
    with _h5.FeatureCache(path) as cache:
        info = cache.map_info(0, "a.jpg")
        assert info == dict(format=2, is_sparse=False, n=1, shape=(30, 40, 16))
        assert np.array_equal(cache.read_patches(0, "a.jpg")[0], dense[0])
        with pytest.raises(PixsfmHipError, match="no feature map"):
            cache.map_info(0, "missing.jpg")
        with pytest.raises(PixsfmHipError, match="outside"):
            cache.read_patches(0, "a.jpg", [3])
    with pytest.raises(PixsfmHipError, match="cannot open"):
        _h5.FeatureCache(tmp_path / "nope.h5")


def test_load_features_from_cache_host(tmp_path):
    """pixsfm.extract.load_features_from_cache -> FeatureManager of host patches; `required` loads a subset."""
    from pixsfm_amd.api import features
    rng = np.random.default_rng(3)
    level = {"im0.jpg": _fmap(rng, 6, 16, 128, np.float16), "im1.jpg": _fmap(rng, 4, 16, 128, np.float16)}
    path = tmp_path / "c.h5"
    h5_writer.write_cache(path, [level])
    fmgr = features.load_features_from_cache(path)
    assert fmgr.num_levels == 1 and fmgr.fset(0).channels == 128
    for image, fm in level.items():
        m = fmgr.fset(0).fmap(image)
        assert sorted(m.keys()) == sorted(fm["keypoint_ids"]) and m.is_sparse
        for i, k in enumerate(fm["keypoint_ids"]):
            p = m.fpatch(k)
            assert np.array_equal(p.data, fm["patches"][i]) and np.array_equal(p.corner, fm["corners"][i])
            assert np.array_equal(p.scale, fm["scales"][i])
    want = {"im1.jpg": level["im1.jpg"]["keypoint_ids"][1:3]}
    sub = features.load_features_from_cache(path, required=want)
    assert list(sub.fset(0).fmaps) == ["im1.jpg"] and sorted(sub.fset(0).fmap("im1.jpg").keys()) == sorted(want["im1.jpg"])
    with pytest.raises(KeyError):
        features.load_features_from_cache(path, required={"im1.jpg": [123456]})
    with pytest.raises(ValueError):
        features.load_features_from_cache(path, fill=False)
