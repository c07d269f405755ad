"""Generates tests/golden/packing_ref.npz by running the REFERENCE's own `find_problem_labels`
(pixsfm/keypoint_adjustment/main.py:13-57) on seeded track-size histograms.  The module itself cannot be
imported here (omegaconf / the pybind module are absent), so the function's source is cut out of the file
with `ast` and executed on its own -- nothing of it is copied into the repository.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_packing.py
"""
import ast
import os
import sys
from collections import Counter
from typing import List, Optional

import numpy as np

SRC = "/root/reference/pixsfm/keypoint_adjustment/main.py"


def load_reference_function():
    tree = ast.parse(open(SRC).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "find_problem_labels")

    class _Logger:
        def warning(self, *a, **k):
            pass
    ns = {"Counter": Counter, "np": np, "sys": sys, "List": List, "Optional": Optional, "logger": _Logger()}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC, "exec"), ns)
    return ns["find_problem_labels"]


def cases():
    """(name, track_labels, max_per_problem, track_edge_counts | None) -- seeded, so the test can rebuild them."""
    rng = np.random.default_rng(20260924)
    out = []
    for i in range(40):
        n_tracks = int(rng.integers(1, 400))
        hi = int(rng.choice([3, 8, 30, 80]))
        sizes = rng.integers(1, hi + 1, n_tracks)
        labels = np.repeat(np.arange(n_tracks), sizes)
        if i % 3 == 0:
            labels = rng.permutation(labels)           # node order interleaves the tracks (graph order)
        cap = int(rng.choice([-1, 5, 10, 50, 50, 200]))
        edge_counts = None
        if i % 4 == 1:                                 # the edge-count form (weights per track id)
            edge_counts = (sizes * (sizes - 1)).astype(np.int64)
        out.append(("case%02d" % i, labels.astype(np.int64), cap, edge_counts))
    out.append(("single", np.zeros(7, np.int64), 50, None))
    out.append(("oversized", np.repeat([0, 1, 2], [70, 5, 5]).astype(np.int64), 50, None))
    out.append(("equal", np.repeat(np.arange(100), 10).astype(np.int64), 50, None))
    return out


if __name__ == "__main__":
    ref = load_reference_function()
    store = {}
    for name, labels, cap, ec in cases():
        pl, bins = ref([int(v) for v in labels], cap, None if ec is None else [int(v) for v in ec])
        store[name + "_labels"] = np.asarray(pl, dtype=np.int32)
        store[name + "_bins"] = np.asarray(bins, dtype=np.int64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "packing_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(store) // 2, "cases")
