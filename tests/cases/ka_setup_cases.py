"""Seeded keypoint-adjustment set-ups (the RunSubset semantics of A15): the match graphs of
tests/cases/graph_cases.py with keypoints, patch corners / scales and varying options.  Inputs only."""
import numpy as np

from cases import graph_cases

PS = 16


def cases():
    """One set-up per seeded graph; node order / labels / roots come from the oracle's restatement of graph.cc (oracle/pxo_graph.py)."""
    import pxo_graph
    rng = np.random.default_rng(299792)
    out = []
    for k, (name, pairs, mm) in enumerate(graph_cases.cases()):
        g = pxo_graph.Graph()
        for (a, b), (matches, sims) in zip(pairs, mm):
            g.register_matches("im%d" % a, "im%d" % b, matches, sims)
        node_image = np.array([int(g.image_id_to_name[nd.image_id][2:]) for nd in g.nodes], np.int32)
        node_feature = np.array([nd.feature_idx for nd in g.nodes], np.int32)
        labels = np.array(pxo_graph.compute_track_labels(g), np.int64)
        roots = np.array(pxo_graph.compute_root_labels(g, labels, pxo_graph.compute_score_labels(g, labels)), np.uint8)
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
