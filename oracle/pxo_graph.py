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
