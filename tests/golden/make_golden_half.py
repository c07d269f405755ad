"""Generates tests/golden/half_rules.npz: the two half-precision rules the cost-map extraction leans on, evaluated by
the REFERENCE's vendored third-party/half.hpp (compiled from its own source into oracle/_ref/libpxo_ref_half.so, oracle/ref_half_shim.cc -- the one piece of /root/reference that builds here without stand-ins):
  sub[i]  = half(a[i]) - half(b[i])     (central differences of costmap_extractor.h:266-276 are taken in the storage type)
  cast[i] = half(v[i]) for double v     (FeaturePatch::SetEntry, featurepatch.h:246-248: through float, two roundings)
Seeded inputs incl. exact cancellations, subnormal results and exact half-way points.

Run in the build container only (needs /root/reference to build oracle/_ref):  python tests/golden/make_golden_half.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pxo  # noqa: E402

r = pxo.ref()
assert r is not None and hasattr(r, "pxo_ref_half_sub"), "build oracle/_ref first (make -C oracle)"
rng = np.random.default_rng(2024)
n = 20000
a = (rng.normal(size=n) * 10.0 ** rng.uniform(-7, 1, n)).astype(np.float16)
b = (rng.normal(size=n) * 10.0 ** rng.uniform(-7, 1, n)).astype(np.float16)
b[:200] = a[:200]
sub = np.empty_like(a)
r.pxo_ref_half_sub(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), sub.ctypes.data_as(C.c_void_p), C.c_int64(n))
v = rng.normal(size=n) * 10.0 ** rng.uniform(-9, 3, n)
h = rng.normal(size=2000).astype(np.float16)
ties = h.astype(np.float64) + np.spacing(h).astype(np.float64) / 2
v = np.concatenate([v, ties, ties * (1 + 2e-16), ties * (1 - 2e-16), ties * (1 + 1e-9), ties * (1 - 1e-9)])
cast = np.empty(v.size, np.uint16)
r.pxo_ref_half_from_double(v.ctypes.data_as(C.c_void_p), cast.ctypes.data_as(C.c_void_p), C.c_int64(v.size))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "half_rules.npz")
np.savez_compressed(out, a=a.view(np.uint16), b=b.view(np.uint16), sub=sub.view(np.uint16), v=v, cast=cast)
print("wrote", out, os.path.getsize(out), "bytes")
