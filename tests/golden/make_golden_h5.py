"""Golden vectors for the dense-feature cache reader: small cache files in both storage formats
(tests/h5_writer.py = the layout pixsfm/features/store_features.py writes through h5py) and, beside them, what the
REFERENCE's own reader hands out for each file -- features/src/featuremanager.cc, featureset.cc, featuremap.cc,
featurepatch.cc compiled in place (oracle/Makefile -> oracle/_ref/libpxo_ref_h5.so).  Run in the build container:

    python tests/golden/make_golden_h5.py

writes tests/golden/h5_cache_*.h5 and tests/golden/h5_cache_ref.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import h5_writer            # noqa: E402
import ref_h5_reader        # noqa: E402

DT = {"half": np.float16, "float": np.float32, "double": np.float64}


def fmap(rng, n, ps, ch, dtype, sparse=True):
    ids = rng.choice(400, n, replace=False)
    scale = np.array([0.25, 0.5])
    return dict(keypoint_ids=[int(k) for k in ids], patches=rng.normal(size=(n, ps, ps, ch)).astype(dtype),
                corners=rng.integers(0, 300, size=(n, 2)), scales=[scale * (1 + k) for k in range(n)],
                metadata={"is_sparse": sparse, "scale": scale, "patch_size": ps})


def cases():
    rng = np.random.default_rng(20240924)
    out = []
    for fmt in ("chunked", "grouped"):
        for name, dtype in DT.items():
            levels = [{"im0.jpg": fmap(rng, 5, 4, 8, dtype), "seq/a/im1.png": fmap(rng, 2, 4, 8, dtype), "z.JPG": fmap(rng, 1, 4, 8, dtype)},
                      {"im0.jpg": fmap(rng, 5, 2, 3, dtype), "seq/a/im1.png": fmap(rng, 2, 2, 3, dtype), "z.JPG": fmap(rng, 1, 2, 3, dtype)}]
            out.append(("sparse_%s_%s" % (fmt, name), levels, dict(dtype_name=name, cache_format=fmt, level_prefix="lvl")))
    # ONE dense map in the file, several keypoint ids: loaded as patch_size windows (featuremap.cc:157-165,246-256)
    dense = rng.normal(size=(1, 20, 28, 6)).astype(np.float16)
    fm = dict(keypoint_ids=[3, 10, 11, 40], patches=dense, corners=np.array([[0, 0], [20, 12], [13, 7], [5, 12]]),
              scales=[np.array([0.5, 0.5])] * 4, metadata={"is_sparse": False, "patch_size": 8, "scale": np.array([0.5, 0.5])})
    out.append(("dense_as_sparse", [{"a.jpg": fm}], dict(dtype_name="half", cache_format="chunked", level_prefix="")))
    dense = rng.normal(size=(1, 9, 11, 5)).astype(np.float32)
    fm = dict(keypoint_ids=[1000000], patches=dense, corners=np.array([[0, 0]]), scales=[np.array([0.25, 0.25])],
              metadata={"is_sparse": False, "scale": np.array([0.25, 0.25])})
    out.append(("dense", [{"a.jpg": fm}], dict(dtype_name="float", cache_format="chunked", level_prefix="")))
    return out


def dump_reference(path, dtype, level_prefix):
    """everything the reference's FeatureManager holds after a filled load, keyed 'level/image/...'"""
    out = {}
    with ref_h5_reader.ReferenceCache(path, dtype, True, level_prefix) as ref:
        out["channels_per_level"] = np.array([ref.channels(l) for l in range(ref.num_levels)], np.int32)
        for l in range(ref.num_levels):
            names = ref.image_names(l)
            out["%d/images" % l] = np.array(names)
            for im in names:
                info = ref.map_info(l, im)
                ids = ref.patch_ids(l, im)
                ps = [ref.patch(l, im, k) for k in ids]
                key = "%d/%s/" % (l, im)
                out[key + "is_sparse"] = np.array(info["is_sparse"])
                out[key + "channels"] = np.array(info["channels"])
                out[key + "ids"] = ids
                out[key + "corners"] = np.stack([p["corner"] for p in ps])
                out[key + "scales"] = np.stack([p["scale"] for p in ps])
                out[key + "patches"] = np.stack([p["data"] for p in ps])
    return out


def main():
    store = {}
    for name, levels, kw in cases():
        path = os.path.join(HERE, "h5_cache_%s.h5" % name)
        if os.path.exists(path):
            os.remove(path)
        h5_writer.write_cache(path, levels, **kw)
        for k, v in dump_reference(path, DT[kw["dtype_name"]], kw["level_prefix"]).items():
            store[name + "|" + k] = v
        store[name + "|level_prefix"] = np.array(kw["level_prefix"])
        print(name, os.path.getsize(path), "bytes")
    np.savez_compressed(os.path.join(HERE, "h5_cache_ref.npz"), **store)


if __name__ == "__main__":
    main()
