"""Generates tests/golden/graph_ref.npz by running the REFERENCE's own match-graph code (pixsfm/base/src/graph.cc,
compiled in place into oracle/_ref/libpxo_ref_graph.so by oracle/Makefile) on seeded random match graphs:
Graph::RegisterMatches -> ComputeTrackLabels -> ComputeScoreLabels -> ComputeRootLabels -> CountTrackEdges.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_graph.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_graph.so")


def cases():
    """Seeded match graphs: (name, pairs (P, 2) image indices, list of (matches (m, 2), sims (m,))).  Random matches
    between few features force the conflict case of graph.cc:126-206 (two keypoints of one image competing for a track);
    coarse similarities force ties in the edge order and in the root scores."""
    rng = np.random.default_rng(314159)
    out = []
    for i in range(24):
        n_img = int(rng.integers(3, 9))
        per_img = int(rng.choice([6, 15, 40]))
        digits = int(rng.choice([1, 2, 6]))
        pairs, mm = [], []
        for a in range(n_img):
            for b in range(a + 1, n_img):
                if rng.random() < 0.15:
                    continue
                m = int(rng.integers(1, 2 * per_img))
                matches = np.stack([rng.integers(0, per_img, m), rng.integers(0, per_img, m)], 1).astype(np.int64)
                sims = np.round(rng.uniform(0.2, 1.0, m), digits)
                if i % 5 == 4 and rng.random() < 0.5:          # reversed pair order (b, a): out-matches of later images
                    pairs.append((b, a)); matches = matches[:, ::-1].copy()
                else:
                    pairs.append((a, b))
                mm.append((matches, sims))
        out.append(("graph%02d" % i, np.array(pairs, np.int32), mm))
    # a chain that must split: features 0 of images 0..3 matched in a cycle with one image twice
    pairs = np.array([(0, 1), (1, 2), (2, 0), (0, 2)], np.int32)
    mm = [(np.array([[0, 0]]), np.array([0.9])), (np.array([[0, 0]]), np.array([0.8])),
          (np.array([[0, 1]]), np.array([0.7])), (np.array([[1, 0]]), np.array([0.95]))]
    out.append(("conflict", pairs, mm))
    return out


def run_reference(pairs, mm):
    lib = C.CDLL(LIB)
    lib.pxo_ref_graph_labels.restype = C.c_int64
    ptr = np.concatenate([[0], np.cumsum([len(m) for m, _ in mm])]).astype(np.int64)
    matches = np.ascontiguousarray(np.concatenate([m for m, _ in mm]).astype(np.int64))
    sims = np.ascontiguousarray(np.concatenate([s for _, s in mm]).astype(np.float64))
    cap = 2 * len(matches) + 1
    node_image, node_feature = np.empty(cap, np.int32), np.empty(cap, np.int32)
    labels, scores, roots = np.empty(cap, np.int64), np.empty(cap, np.float64), np.empty(cap, np.uint8)
    track_edges, n_tracks = np.empty(cap, np.int64), C.c_int64()
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    n = lib.pxo_ref_graph_labels(C.c_int64(len(pairs)), C.c_void_p(pairs.ctypes.data), C.c_void_p(ptr.ctypes.data),
                                 C.c_void_p(matches.ctypes.data), C.c_void_p(sims.ctypes.data), C.c_int64(cap),
                                 C.c_void_p(node_image.ctypes.data), C.c_void_p(node_feature.ctypes.data),
                                 C.c_void_p(labels.ctypes.data), C.c_void_p(scores.ctypes.data),
                                 C.c_void_p(roots.ctypes.data), C.byref(n_tracks), C.c_void_p(track_edges.ctypes.data))
    assert n >= 0
    return dict(node_image=node_image[:n].copy(), node_feature=node_feature[:n].copy(), labels=labels[:n].copy(),
                scores=scores[:n].copy(), roots=roots[:n].copy(), track_edges=track_edges[:n_tracks.value].copy())


if __name__ == "__main__":
    store = {}
    for name, pairs, mm in cases():
        for k, v in run_reference(pairs, mm).items():
            store[name + "_" + k] = v
    path = os.path.join(HERE, "graph_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "graphs,", sum(len(store[n + "_labels"]) for n, _, _ in cases()), "nodes")
