"""Generates tests/golden/ka_setup_ref.npz by running the REFERENCE's own keypoint-adjustment problem construction --
TopologicalKeypointOptimizer::Run / SetUp, FeatureMetricKeypointOptimizer::AddIntraResiduals,
KeypointOptimizerBase::ParameterizeKeypoints, KeypointAdjustmentSetup -- compiled in place against a RECORDING ceres::Problem
(oracle/ref_ka_setup_shim.cc -> oracle/_ref/libpxo_ref_ka_setup.so) on the seeded match graphs of make_golden_graph.py:
which residual blocks are added with which ScaledLoss weight, which keypoints are held constant, which get box bounds.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_ka_setup.py
"""
import ctypes as C
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_ka_setup.so")
PS = 16


def _graphs():
    spec = importlib.util.spec_from_file_location("make_golden_graph", os.path.join(HERE, "make_golden_graph.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def cases():
    """One set-up per seeded graph; node order / labels / roots come from tests/golden/graph_ref.npz (the reference's)."""
    gold = np.load(os.path.join(HERE, "graph_ref.npz"))
    rng = np.random.default_rng(299792)
    out = []
    for k, (name, pairs, mm) in enumerate(_graphs().cases()):
        node_image, node_feature = gold[name + "_node_image"], gold[name + "_node_feature"]
        labels, roots = gold[name + "_labels"], gold[name + "_roots"]
        n = len(node_image)
        n_images = int(pairs.max()) + 1
        n_feat = np.zeros(n_images, np.int64)
        for (a, b), (m, _) in zip(pairs, mm):
            n_feat[a] = max(n_feat[a], m[:, 0].max() + 1); n_feat[b] = max(n_feat[b], m[:, 1].max() + 1)
        kp_ptr = np.concatenate([[0], np.cumsum(n_feat)]).astype(np.int64)
        kp = rng.uniform(40, 900, (int(kp_ptr[-1]), 2))
        scale = np.tile(rng.uniform(0.25, 1.0, 2) if k % 2 else np.ones(2), (n, 1))
        node_kp = kp[kp_ptr[node_image] + node_feature]
        corner = (np.floor(node_kp * scale - PS / 2.0) + rng.integers(-5, 6, (n, 2))).astype(np.int32)   # some keypoints near a patch edge
        opt = dict(weight_by_sim=bool(k % 2 == 0), root_edges_only=bool(k % 4 == 1), root_regularize_weight=[-1.0, 0.3][k % 3 == 2],
                   bound=[4.0, -1.0, 1.5][k % 3], const_roots=bool(k % 2), const_images=np.array([0] if k % 5 == 3 else [], np.int32))
        if k % 3 == 1:        # RunSubset on the nodes of about half of the tracks (a ParallelOptimizer group)
            tracks = np.unique(labels)
            pick = tracks[rng.random(len(tracks)) < 0.5]
            sub = np.flatnonzero(np.isin(labels, pick)).astype(np.int64)
            if len(sub) == 0:
                sub = np.arange(n, dtype=np.int64)
        elif k % 6 == 2:      # an arbitrary node subset: matches leave it, their destinations are never parameterised
            sub = np.flatnonzero(rng.random(n) < 0.5).astype(np.int64)
            if len(sub) == 0:
                sub = np.arange(n, dtype=np.int64)
        else:
            sub = None
        out.append(dict(name=name, pairs=pairs, mm=mm, n_images=n_images, kp_ptr=kp_ptr, kp=kp, corner=corner, scale=scale,
                        nodes_in_problem=sub, node_image=node_image, node_feature=node_feature, labels=labels, roots=roots, **opt))
    return out


def run_reference(c):
    lib = C.CDLL(LIB)
    lib.pxo_ref_ka_setup.restype = C.c_int64
    mm = c["mm"]
    ptr = np.concatenate([[0], np.cumsum([len(m) for m, _ in mm])]).astype(np.int64)
    matches = np.ascontiguousarray(np.concatenate([m for m, _ in mm]).astype(np.int64))
    sims = np.ascontiguousarray(np.concatenate([s for _, s in mm]).astype(np.float64))
    pairs = np.ascontiguousarray(c["pairs"], dtype=np.int32)
    n = len(c["node_image"])
    cap = 3 * 2 * len(matches) + 8
    bs, bd, bw = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.float64)
    nconst, nb = np.empty(n, np.uint8), np.empty((n, 4))
    sub = c["nodes_in_problem"]
    p = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    corner, scale = np.ascontiguousarray(c["corner"], np.int32), np.ascontiguousarray(c["scale"], np.float64)
    kp, kp_ptr = np.ascontiguousarray(c["kp"]), np.ascontiguousarray(c["kp_ptr"])
    ci = np.ascontiguousarray(c["const_images"], np.int32)
    m = lib.pxo_ref_ka_setup(C.c_int64(len(pairs)), p(pairs), p(ptr), p(matches), p(sims), int(c["n_images"]), p(kp_ptr), p(kp), None,
                             PS, PS, p(corner), p(scale), p(sub), C.c_int64(0 if sub is None else len(sub)), p(ci), len(ci),
                             int(c["const_roots"]), int(c["weight_by_sim"]), int(c["root_edges_only"]),
                             C.c_double(c["root_regularize_weight"]), C.c_double(c["bound"]), C.c_int64(cap), p(bs), p(bd), p(bw),
                             p(nconst), p(nb))
    assert m >= 0
    order = np.lexsort((bw[:m], bd[:m], bs[:m]))          # insertion order follows an unordered_set: store a canonical order
    return dict(src=bs[:m][order].copy(), dst=bd[:m][order].copy(), w=bw[:m][order].copy(), const=nconst, bounds=nb)


if __name__ == "__main__":
    store = {}
    n_blocks = 0
    for c in cases():
        r = run_reference(c)
        for k, v in r.items():
            store[c["name"] + "_" + k] = v
        n_blocks += len(r["src"])
    path = os.path.join(HERE, "ka_setup_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "set-ups,", n_blocks, "residual blocks")
