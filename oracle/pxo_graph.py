"""Pure-Python restatement of the match-graph labelling (TEST INFRASTRUCTURE ONLY -- never imported by the
product package): ComputeTrackLabels / ComputeScoreLabels / ComputeRootLabels, pixsfm/base/src/graph.cc:126-256.
The product calls the native host implementation (csrc/pxr_graph.cpp); tests compare the two."""


def compute_track_labels(graph):
    """ComputeTrackLabels (graph.cc:126-206): maximum-spanning-forest union-find over edges sorted by
    descending (sim, src, dst), never merging two components that share an image."""
    n = len(graph.nodes)
    edges = sorted(((m.sim, nd.node_idx, m.node_idx) for nd in graph.nodes for m in nd.out_matches), reverse=True)
    parent = [-1] * n
    images = [{nd.image_id} for nd in graph.nodes]

    def root(i):
        path = []
        while parent[i] != -1:
            path.append(i)
            i = parent[i]
        for p in path:
            parent[p] = i
        return i

    for _, a, b in edges:
        ra, rb = root(a), root(b)
        if ra == rb or (images[ra] & images[rb]):
            continue
        if len(images[ra]) < len(images[rb]):
            parent[ra] = rb; images[rb] |= images[ra]; images[ra] = set()
        else:
            parent[rb] = ra; images[ra] |= images[rb]; images[rb] = set()
    labels = [-1] * n
    n_tracks = 0
    for i in range(n):
        if parent[i] == -1:
            labels[i] = n_tracks
            n_tracks += 1
    for i in range(n):
        if labels[i] == -1:
            labels[i] = labels[root(i)]
    return labels


def compute_score_labels(graph, track_labels):                       # graph.cc:208-223
    s = [0.0] * len(graph.nodes)
    for nd in graph.nodes:
        for m in nd.out_matches:
            if track_labels[nd.node_idx] == track_labels[m.node_idx]:
                s[nd.node_idx] += m.sim
                s[m.node_idx] += m.sim
    return s


def compute_root_labels(graph, track_labels, score_labels):          # graph.cc:225-256
    n = len(graph.nodes)
    order = sorted(((score_labels[i], i) for i in range(n)), reverse=True)
    is_root = [False] * n
    has_root = set()
    for _, i in order:
        t = track_labels[i]
        if t in has_root:
            continue
        is_root[i] = True
        has_root.add(t)
    return is_root


def build_edges(graph, keypoints, track_labels, root_labels, nodes_in_problem=None, weight_by_sim=True,
                root_edges_only=False, root_regularize_weight=-1.0):
    """TopologicalKeypointOptimizer::SetUp + FeatureMetricKeypointOptimizer::AddIntraResiduals
    (topological_keypoint_optimizer.h:97-175, featuremetric_keypoint_optimizer.h:158-202).
    Returns (src, dst, weight) lists of residual blocks, in the reference's insertion order."""
    nodes = graph.nodes
    node_ids = range(len(nodes)) if nodes_in_problem is None else nodes_in_problem
    regularize = root_regularize_weight > 0.0
    connected_to_root = {}
    track_root = {}
    cand = []
    for i in node_ids:
        for m in nodes[i].out_matches:
            j = m.node_idx
            if track_labels[i] != track_labels[j]:
                continue                                     # inter-track: TODO in the reference too (:140-142)
            cand.append((i, j, m.sim))
            if regularize:
                for r in (i, j):
                    if root_labels[r]:
                        track_root[track_labels[r]] = r
                        connected_to_root[i] = connected_to_root[j] = True
    src, dst, w = [], [], []

    def same_keypoint(a, b):                                 # "avoid optimizing a keypoint to itself" (:147-150)
        na, nb = nodes[a], nodes[b]
        return na.image_id == nb.image_id and na.feature_idx == nb.feature_idx

    def add(a, b, weight):                                   # AddIntraResiduals
        if track_labels[a] != track_labels[b]:
            return
        if root_edges_only and not root_labels[a] and not root_labels[b]:
            return
        src.append(a); dst.append(b); w.append(weight)

    for i, j, sim in cand:
        if same_keypoint(i, j):
            continue
        add(i, j, sim if weight_by_sim else 1.0)
        if regularize:
            for k in (i, j):
                if not connected_to_root.get(k, False):
                    add(k, track_root[track_labels[k]], root_regularize_weight)
                    connected_to_root[k] = True
    return src, dst, w
