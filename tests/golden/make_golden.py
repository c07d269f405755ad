#!/usr/bin/env python
"""Generates tests/golden/bicubic_ref.npz from the REFERENCE's own code.

Run in the build container (needs /root/reference): oracle/Makefile compiles the reference's
pixsfm/base/src/cubic_hermite_spline_simd.h and grid2d.h in place into oracle/_ref/libpxo_ref.so;
this script evaluates BiCubicInterpolator::EvaluateSIMD's call sequence
(pixsfm/base/src/interpolation.h:177-218) through it on seeded random grids -- the cases of the
reference's TestBiCubicSimilarCeres (pixsfm/base/src/interpolation_test.cc:327-364: 10x10x128
grids of half / float / double, r, c in {0, 0.1, ..., 9.9} incl. the clamped border) -- and stores
inputs + outputs.  The fixture travels to the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pxo  # noqa: E402


def main():
    assert pxo.ref() is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260923)
    out = {}
    pos = np.stack(np.meshgrid(np.arange(0, 100, 17) / 10.0, np.arange(0, 100, 19) / 10.0, indexing="ij"), -1).reshape(-1, 2)
    pos = np.concatenate([pos, [[-0.7, 4.2], [9.95, 9.95], [3.0, 5.0], [0.0, 0.0], [12.3, -2.5], [9.9, 0.1]]])
    out["positions_rc"] = pos
    for name, dt in (("f16", np.float16), ("f32", np.float32), ("f64", np.float64)):
        data = rng.uniform(-1, 1, (10, 10, 128)).astype(dt)
        out["grid_" + name] = data
        for fs in ((0, 1) if name == "f16" else (0,)):
            res = np.empty((len(pos), 3, 128))
            for i, (r, c) in enumerate(pos):
                res[i] = np.stack(pxo.ref_bicubic(data, float(r), float(c), bool(fs)))
            out["out_%s_fs%d" % (name, fs)] = res
    np.savez_compressed(os.path.join(HERE, "bicubic_ref.npz"), **out)
    print("wrote", os.path.join(HERE, "bicubic_ref.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
