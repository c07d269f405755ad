"""CPU: libpixsfm_h5.so (the native reader of pixsfm's dense-feature cache; the reference's side is
features/src/featuremanager.cc, featureset.cc, featuremap.cc:60-267, featurepatch.cc) hands back what was written:
  * the committed cache files tests/golden/h5_cache_*.h5 against the seeded content they were written from
    (tests/golden/make_golden_h5.py), both storage formats x three dtypes, a dense map stored once and loaded as patch_size
    windows at the stored corners (featuremap.cc:157-165,246-256), a true dense map under kDensePatchId;
  * freshly written random caches, the subset / on-demand path (FeatureSet::Load(required_patches), featureset.cc:90-143),
    malformed files;
  * one committed file against what the THIRD-PARTY `h5dump` of the image's HDF5 distribution prints for it.
PARITY UNPINNED with respect to the reference's reader: it needs HighFive (an empty submodule of the checkout), Eigen and
COLMAP headers and cannot be compiled here; the WRITER is tests/h5_writer.py's restatement of store_features.py (h5py absent)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import h5_writer

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
pytestmark = pytest.mark.skipif(not h5_writer.available(), reason="the image's libhdf5 is missing")
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



def _written(levels):
    """what a reader must hand back for `levels` (the argument of h5_writer.write_cache), keyed like _mine()"""
    out = {"channels_per_level": np.array([next(iter(lv.values()))["patches"].shape[-1] for lv in levels], np.int32)}
    for l, level in enumerate(levels):
        out["%d/images" % l] = np.array(sorted(level))
        for im, fm in level.items():
            ids = np.asarray(fm["keypoint_ids"])
            order = np.argsort(ids, kind="stable")
            corners, scales = np.asarray(fm["corners"]), np.stack([np.asarray(s, np.float64) for s in fm["scales"]])
            patches = fm["patches"]
            stored_sparse = bool(fm["metadata"]["is_sparse"])
            if not stored_sparse and len(ids) > 1:        # ONE dense map, several keypoints: windows of patch_size at the corners
                ps = int(fm["metadata"]["patch_size"])
                patches = np.stack([patches[0][y:y + ps, x:x + ps] for x, y in corners])
            key = "%d/%s/" % (l, im)
            out[key + "is_sparse"] = np.array(stored_sparse or len(ids) > 1)
            out[key + "channels"] = np.array(patches.shape[-1])
            out[key + "ids"] = ids[order].astype(np.uint32)
            out[key + "corners"] = corners[order]
            out[key + "scales"] = scales[order]
            out[key + "patches"] = patches[order]
    return out


def _same(want, mine):
    assert sorted(k for k in mine if k != "dtype") == sorted(want)
    for k, v in want.items():
        m = mine[k]
        assert np.asarray(m).shape == np.asarray(v).shape, k
        if k.endswith("patches"):
            assert m.dtype == v.dtype, k
        assert np.array_equal(np.asarray(m), np.asarray(v)), k


def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_h5", os.path.join(GOLD, "make_golden_h5.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", NAMES)
def test_committed_cache_files_read_back_as_written(name):
    g = _golden_module()
    _, levels, kw = next(c for c in g.cases() if c[0] == name)
    mine = _mine(os.path.join(GOLD, "h5_cache_%s.h5" % name), kw["level_prefix"])
    assert mine["dtype"] == np.dtype(g.DT[kw["dtype_name"]])
    _same(_written(levels), mine)


def test_committed_file_against_h5dump():
    """Third-party view of one committed file: `h5dump` (HDF5 tools of the image) prints the patch data set of the first map;
    the product's reader returns the same numbers (floats printed with 9 significant digits)."""
    tool = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(tool):
        pytest.skip("h5dump not in the image")
    from pixsfm_amd import _h5
    path = os.path.join(GOLD, "h5_cache_sparse_chunked_float.h5")
    listing = subprocess.run([tool, "-n", path], capture_output=True, text=True, check=True).stdout
    dsets = [ln.split()[1] for ln in listing.splitlines() if ln.strip().startswith("dataset") and "patches" in ln]
    assert dsets, listing
    d = dsets[0]                                                     # e.g. /lvl0/im0.jpg/patches
    txt = subprocess.run([tool, "-d", d, "-y", "-w", "0", "-m", "%.9g", path], capture_output=True, text=True, check=True).stdout
    body = txt[txt.index("DATA {") + 6:txt.rindex("}")]
    body = body[:body.rindex("}")] if body.rstrip().endswith("}") else body
    vals = np.array([float(t) for t in body.replace("\n", " ").replace(",", " ").split() if t not in ("{", "}")], np.float32)
    parts = d.strip("/").split("/")
    level, im = int(parts[0].replace("lvl", "")), "/".join(parts[1:-1])
    with _h5.FeatureCache(path, "lvl") as cache:
        got = cache.read_patches(level, im)
    assert vals.size == got.size and np.array_equal(vals, got.reshape(-1).astype(np.float32))


def _random_map(rng, n, ps, ch, dtype):
    ids = rng.choice(100000, n, replace=False)
    scale = rng.uniform(0.1, 1.0, 2)
    return dict(keypoint_ids=[int(k) for k in ids], patches=rng.normal(size=(n, ps, ps, ch)).astype(dtype),
                corners=rng.integers(0, 4000, size=(n, 2)), scales=[rng.uniform(0.1, 1.0, 2) for _ in range(n)],
                metadata={"is_sparse": True, "scale": scale, "patch_size": ps})


@pytest.mark.parametrize("seed", range(6))
def test_random_caches_read_back_as_written(tmp_path, seed):
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
    _same(_written(levels), mine)


@pytest.mark.parametrize("cache_format", ["chunked", "grouped"])
def test_subset_of_patches_on_demand(tmp_path, cache_format):
    """FeatureManager(fill=False) then FeatureSet::Load(required_patches, fill=True) (featureset.cc:90-143,
    featuremap.cc:217-267 / :92-132): pxr_h5_read_patches(which) and load_features_from_cache(required=...) hand out exactly
    the requested patches of what was written."""
    from pixsfm_amd import _h5
    from pixsfm_amd.api import features
    rng = np.random.default_rng(7)
    level = {"im0.jpg": _random_map(rng, 12, 8, 16, np.float16), "im1.jpg": _random_map(rng, 5, 8, 16, np.float16)}
    path = tmp_path / "c.h5"
    h5_writer.write_cache(path, [level], cache_format=cache_format)
    want = {"im0.jpg": [level["im0.jpg"]["keypoint_ids"][i] for i in (7, 0, 3)], "im1.jpg": [level["im1.jpg"]["keypoint_ids"][4]]}
    with _h5.FeatureCache(path) as cache:
        fmgr = features.load_features_from_cache(path, required=want)
        for im, ids in want.items():
            fm = level[im]
            src = {int(k): i for i, k in enumerate(fm["keypoint_ids"])}
            stored, corners, scales = cache.map_meta(0, im, len(fm["keypoint_ids"]))
            pos = {int(k): i for i, k in enumerate(stored)}
            got = cache.read_patches(0, im, np.array([pos[k] for k in ids]))
            for j, k in enumerate(ids):
                w = src[k]
                assert np.array_equal(fm["patches"][w], got[j]) and np.array_equal(fm["corners"][w], corners[pos[k]])
                assert np.array_equal(np.asarray(fm["scales"][w]), scales[pos[k]])
                fp = fmgr.fset(0).fmap(im).fpatch(k)
                assert np.array_equal(fp.data, fm["patches"][w]) and np.array_equal(fp.corner, fm["corners"][w])
                assert np.array_equal(fp.scale, np.asarray(fm["scales"][w]))
            assert sorted(fmgr.fset(0).fmap(im).keys()) == sorted(ids)        # the others are absent in the product's manager


def test_malformed_files_are_refused(tmp_path):
    from pixsfm_amd import _h5
    from pixsfm_amd._lib import PixsfmHipError
    rng = np.random.default_rng(9)
    with pytest.raises(PixsfmHipError):
        _h5.FeatureCache(tmp_path / "nope.h5")
    # level group missing: channels_per_level announces two levels, only one is stored
    path = tmp_path / "short.h5"
    h5_writer.write_cache(path, [{"a.jpg": _random_map(rng, 2, 4, 8, np.float16)}], channels_per_level=[8, 8])
    with pytest.raises(PixsfmHipError):
        _h5.FeatureCache(path)
    # unknown storage format (featuremap.cc:60-75 "Unknown featuremap format.")
    path = tmp_path / "fmt.h5"
    h5_writer.write_cache(path, [{"a.jpg": _random_map(rng, 2, 4, 8, np.float16)}], format_override=3)
    with pytest.raises(PixsfmHipError, match="format"):
        with _h5.FeatureCache(path) as cache:
            cache.map_info(0, "a.jpg")


@pytest.mark.parametrize("n_ids", [1, 2, 5])
@pytest.mark.parametrize("dtype,name", [(np.float16, "half"), (np.float32, "float")])
def test_dense_maps(tmp_path, n_ids, dtype, name):
    """ONE dense map stored per image; several keypoint ids => loaded as patch_size windows at the stored corners
    (featuremap.cc:157-165,246-256), a single id => a true dense map under kDensePatchId."""
    rng = np.random.default_rng(40 + n_ids)
    ps, h, w, c = 6, 23, 31, 4
    dense = rng.normal(size=(1, h, w, c)).astype(dtype)
    ids = [1000000] if n_ids == 1 else [int(k) for k in rng.choice(500, n_ids, replace=False)]
    corners = np.stack([rng.integers(0, w - ps + 1, n_ids), rng.integers(0, h - ps + 1, n_ids)], 1) if n_ids > 1 else np.array([[0, 0]])
    fm = dict(keypoint_ids=ids, patches=dense, corners=corners, scales=[np.array([0.5, 0.25])] * n_ids,
              metadata={"is_sparse": False, "patch_size": ps, "scale": np.array([0.5, 0.25])})
    path = tmp_path / "d.h5"
    h5_writer.write_cache(path, [{"a.jpg": fm, "b.png": fm}], dtype_name=name)
    mine = _mine(path, "")
    assert bool(mine["0/a.jpg/is_sparse"]) == (n_ids > 1) and mine["0/a.jpg/patches"].shape[1:] == ((ps, ps, c) if n_ids > 1 else (h, w, c))
    _same(_written([{"a.jpg": fm, "b.png": fm}]), mine)
