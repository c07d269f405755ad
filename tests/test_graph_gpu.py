"""GPU: match-graph labelling on the device (pxr_graph_labels_device) against the oracle's pure-Python
restatement of graph.cc (oracle/pxo_graph.py) on the seeded graphs and, at size, against the native host implementation
(itself compared with the oracle in tests/test_graph_labelling.py): track labels, scores (bit-identical: same summation
order), roots."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_graph_labelling import _build, _gen, oracle_labels  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_labelling_matches_the_oracle(ctx):
    from pixsfm_amd.api import base
    gen = _gen()
    for name, pairs, mm in gen.cases():
        g = _build(base, pairs, mm)
        want_l, want_s, want_r = oracle_labels(g)
        labels, scores, roots = base.compute_labels_on_device(g, ctx)
        assert labels == want_l, name
        assert np.array_equal(scores, want_s), name
        assert [int(r) for r in roots] == want_r, name


def _flat_random_graph(rng, n_groups, per, n_img, n_matches, n_cross):
    n = n_groups * per
    node_image = rng.integers(0, n_img, n).astype(np.int32)
    grp = rng.integers(0, n_groups, n_matches)
    src = (grp * per + rng.integers(0, per, n_matches)).astype(np.int64)
    dst = (grp * per + rng.integers(0, per, n_matches)).astype(np.int64)
    src = np.concatenate([src, rng.integers(0, n, n_cross)])          # a few matches between groups: larger components
    dst = np.concatenate([dst, rng.integers(0, n, n_cross)])
    keep = node_image[src] != node_image[dst]
    keep[:50] = True                                                  # keep some same-image / self matches: refused merges
    src, dst = src[keep], dst[keep]
    order = np.argsort(src, kind="stable")                            # Graph order: grouped by ascending source node
    src, dst = src[order], dst[order]
    sim = np.round(rng.uniform(0.1, 1.0, len(src)), 2)                # coarse similarities: ties everywhere
    dup = rng.integers(0, len(src), 200)                              # duplicated matches (identical tuples)
    src, dst, sim = np.concatenate([src, src[dup]]), np.concatenate([dst, dst[dup]]), np.concatenate([sim, sim[dup]])
    order = np.argsort(src, kind="stable")
    return node_image, src[order], dst[order], sim[order]


@pytest.mark.parametrize("n_groups,per,n_matches", [(1500, 12, 60000), (40, 150, 30000)])
def test_device_labelling_equals_host_labelling_at_size(ctx, n_groups, per, n_matches):
    """Many small components (the KA regime) and a few large ones (hundreds of nodes, thousands of matches per
    component: the wavefront's rank sort and the sequential merge run long)."""
    from pixsfm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n_groups)
    node_image, src, dst, sim = _flat_random_graph(rng, n_groups, per, 40, n_matches, 300)
    n, m = len(node_image), len(src)
    want_l, want_s, want_r = np.empty(n, np.int64), np.empty(n, np.float64), np.empty(n, np.uint8)
    ntr = C.c_int64()
    _lib.check(lib.pxr_graph_track_labels(n, node_image.ctypes.data, m, src.ctypes.data, dst.ctypes.data, sim.ctypes.data,
                                          want_l.ctypes.data, C.byref(ntr)), "host labels")
    _lib.check(lib.pxr_graph_score_labels(n, m, src.ctypes.data, dst.ctypes.data, sim.ctypes.data, want_l.ctypes.data,
                                          want_s.ctypes.data), "host scores")
    _lib.check(lib.pxr_graph_root_labels(n, want_l.ctypes.data, want_s.ctypes.data, want_r.ctypes.data), "host roots")
    d = [ctx.to_device(a, dt) for a, dt in ((node_image, np.int32), (src, np.int64), (dst, np.int64), (sim, np.float64))]
    got_l, got_s, got_r = ctx.empty((n,), np.int64), ctx.empty((n,), np.float64), ctx.empty((n,), np.uint8)
    ntr_d = C.c_int64()
    _lib.check(lib.pxr_graph_labels_device(ctx.handle, n, d[0].ptr, m, d[1].ptr, d[2].ptr, d[3].ptr, got_l.ptr, got_s.ptr,
                                           got_r.ptr, C.byref(ntr_d)), "device labels")
    assert ntr_d.value == ntr.value
    assert np.array_equal(got_l.download(), want_l)
    assert np.array_equal(got_s.download(), want_s)
    assert np.array_equal(got_r.download(), want_r)
    # labels only (scores / roots optional), and the argument checks
    only = ctx.empty((n,), np.int64)
    _lib.check(lib.pxr_graph_labels_device(ctx.handle, n, d[0].ptr, m, d[1].ptr, d[2].ptr, d[3].ptr, only.ptr, None, None, None), "labels only")
    assert np.array_equal(only.download(), want_l)
    bad = ctx.to_device(src[::-1].copy(), np.int64)                   # not in Graph order
    with pytest.raises(_lib.PixsfmHipError):
        _lib.check(lib.pxr_graph_labels_device(ctx.handle, n, d[0].ptr, m, bad.ptr, d[2].ptr, d[3].ptr, only.ptr, None, None, None), "order")


def test_a_giant_component_is_refused_by_the_device_and_labelled_on_the_host(ctx):
    """One connected component with more matches than the device kernel's per-component rank sort takes: the C entry
    point answers PXR_EUNSUPPORTED (no silent wrong labels, no minutes-long single wavefront) and
    base.compute_labels_on_device falls back to the native host labelling -- same three results."""
    from pixsfm_amd import _lib
    from pixsfm_amd.api import base
    lib = _lib.load()
    rng = np.random.default_rng(77)
    n, m = 3000, 40000
    node_image = rng.integers(0, 60, n).astype(np.int32)
    src = np.sort(rng.integers(0, n, m)).astype(np.int64)
    dst = rng.integers(0, n, m).astype(np.int64)
    sim = np.round(rng.uniform(0.1, 1.0, m), 3)
    d = [ctx.to_device(a, dt) for a, dt in ((node_image, np.int32), (src, np.int64), (dst, np.int64), (sim, np.float64))]
    out = ctx.empty((n,), np.int64)
    rc = lib.pxr_graph_labels_device(ctx.handle, n, d[0].ptr, m, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr, None, None, None)
    assert rc == _lib.PXR_EUNSUPPORTED
    g = base.Graph()
    for i in range(n):
        g.add_node(int(node_image[i]), i)
    for a, b, s in zip(src.tolist(), dst.tolist(), sim.tolist()):
        g.add_edge(g.nodes[a], g.nodes[b], s)
    labels, scores, roots = base.compute_labels_on_device(g, ctx)
    want = base.compute_track_labels(g)
    assert labels == want
    assert np.array_equal(scores, np.asarray(base.compute_score_labels(g, want)))
    assert roots == base.compute_root_labels(g, want, scores)
