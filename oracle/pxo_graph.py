"""Pure-Python restatement of the match graph and of the keypoint-adjustment problem construction (TEST INFRASTRUCTURE ONLY --
never imported by the product package):
  Graph.register_matches                      Graph::RegisterMatches / FindOrCreateNode / AddEdge, pixsfm/base/src/graph.cc:38-79
  compute_track_labels / _score_ / _root_     graph.cc:126-256
  build_edges                                 TopologicalKeypointOptimizer::SetUp + FeatureMetricKeypointOptimizer::
                                              AddIntraResiduals (keypoint_adjustment/src/topological_keypoint_optimizer.h:97-175,
                                              featuremetric_keypoint_optimizer.h:158-202)
  parameterize_keypoints                      KeypointOptimizerBase::ParameterizeKeypoints (keypoint_optimizer.h:110-157) with
                                              KeypointAdjustmentSetup (keypoint_adjustment_options.h:24-45)
The product calls its native host implementation (csrc/pxr_graph.cpp); tests compare the two.
PARITY UNPINNED: these reference files include COLMAP / Ceres headers and cannot be compiled here, and the reference has no test
for them; this is a restatement read from the source."""


class Match:
    __slots__ = ("node_idx", "sim")

    def __init__(self, node_idx, sim):
        self.node_idx, self.sim = node_idx, sim


class FeatureNode:
    __slots__ = ("image_id", "feature_idx", "node_idx", "out_matches")

    def __init__(self, image_id, feature_idx):
        self.image_id, self.feature_idx, self.node_idx, self.out_matches = image_id, feature_idx, -1, []


class Graph:
    """graph.h:47-97 -- nodes in order of creation, image ids in order of first appearance."""

    def __init__(self):
        self.nodes, self.node_map, self.image_name_to_id, self.image_id_to_name = [], {}, {}, {}

    def find_or_create_node(self, image_name, feature_idx):                      # graph.cc:38-57
        image_id = self.image_name_to_id.setdefault(image_name, len(self.image_name_to_id))
        self.image_id_to_name.setdefault(image_id, image_name)
        key = (image_id, int(feature_idx))
        if key in self.node_map:
            return self.nodes[self.node_map[key]]
        node = FeatureNode(image_id, int(feature_idx))
        self.nodes.append(node)
        node.node_idx = len(self.nodes) - 1
        self.node_map[key] = node.node_idx
        return node

    def register_matches(self, imname1, imname2, matches, similarities=None):   # graph.cc:66-79
        for k, (f1, f2) in enumerate(matches):
            n1 = self.find_or_create_node(imname1, f1)
            n2 = self.find_or_create_node(imname2, f2)
            n1.out_matches.append(Match(n2.node_idx, 1.0 if similarities is None else float(similarities[k])))   # AddEdge :59-64


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
                    add(k, track_root.get(track_labels[k], 0), root_regularize_weight)   # operator[] of the map: node 0 when the root never showed up
                    connected_to_root[k] = True
    return src, dst, w


def parameterize_keypoints(graph, src, dst, nodes_in_problem, is_node_constant, node_kp, corner, scale, width, height, bound,
                           sparse=True):
    """KeypointOptimizerBase::ParameterizeKeypoints (keypoint_optimizer.h:110-157) after the residual blocks (src, dst) were added.
    Returns (const [n] bool, bounds [n, 4] = lower x, lower y, upper x, upper y, NaN where no bound is set).  Only nodes of
    nodes_in_problem that some residual block touched (will_be_optimized_) are visited; a node that is touched but not visited
    stays a free, unbounded parameter block."""
    import numpy as np
    n = len(graph.nodes)
    touched = np.zeros(n, bool)
    touched[list(src)] = True
    touched[list(dst)] = True
    const = np.zeros(n, bool)
    bounds = np.full((n, 4), np.nan)
    for i in (range(n) if nodes_in_problem is None else nodes_in_problem):
        if not touched[i]:
            continue
        if is_node_constant(graph.nodes[i]):
            const[i] = True
        elif bound > 0.0 or sparse:
            sx, sy = float(scale[i][0]), float(scale[i][1])
            lowerx, lowery = (corner[i][0] + 0.5) / sx, (corner[i][1] + 0.5) / sy
            upperx, uppery = lowerx + width / sx, lowery + height / sy
            if bound > 0.0:
                upperx = min(node_kp[i][0] + bound / sx, upperx)
                uppery = min(node_kp[i][1] + bound / sy, uppery)
                lowerx = max(node_kp[i][0] - bound / sx, lowerx)
                lowery = max(node_kp[i][1] - bound / sy, lowery)
            bounds[i] = (lowerx, lowery, upperx, uppery)
    return const, bounds
