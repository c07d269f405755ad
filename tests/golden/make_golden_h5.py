"""Small dense-feature cache files in both storage formats for the reader tests: written by tests/h5_writer.py
(the layout pixsfm/extract.py:98-127 and pixsfm/features/store_features.py write through h5py, issued here through ctypes on
the image's libhdf5 because h5py is absent) from the seeded content of cases().  The tests regenerate that content from the
seed and require the product's reader to hand it back (tests/test_h5_reader.py).

    python tests/golden/make_golden_h5.py       # writes tests/golden/h5_cache_*.h5"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import h5_writer            # noqa: E402

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


def main():
    for name, levels, kw in cases():
        path = os.path.join(HERE, "h5_cache_%s.h5" % name)
        if os.path.exists(path):
            os.remove(path)
        h5_writer.write_cache(path, levels, **kw)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
