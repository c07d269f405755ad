"""Mirror of `pixsfm._pixsfm._base` (pixsfm/base/bindings.cc:29-154) for the accelerated path:
Graph / FeatureNode / Match, track / score / root labelling, InterpolationConfig, default confs.

Graph labelling is host pre-processing (component 6): it is restated in plain
Python so that the adjusters can be driven exactly like pixsfm's; it is not on the GPU path.
"""
from copy import deepcopy

import numpy as np

# pixsfm/base/main.py:1-22
interpolation_default_conf = {
    'nodes': [[0.0, 0.0]],
    'mode': 'BICUBIC',
    'l2_normalize': True,
    'ncc_normalize': False,
    'use_float_simd': False,
}

solver_default_conf = {
    'function_tolerance': 0.0,
    'gradient_tolerance': 0.0,
    'parameter_tolerance': 0.0,
    'minimizer_progress_to_stdout': False,
    'max_num_iterations': 100,
    'max_linear_solver_iterations': 200,
    'max_num_consecutive_invalid_steps': 10,
    'max_consecutive_nonmonotonic_steps': 10,
    'use_inner_iterations': False,
    'use_nonmonotonic_steps': False,
    'update_state_every_iteration': False,
    'num_threads': -1,
}


def merge_conf(default, override):
    """OmegaConf.merge stand-in: recursive dict merge, unknown keys are an error like the
    reference's strict make_dataclass merge (_pixsfm/src/helpers.h:149-232)."""
    out = deepcopy(default)
    for k, v in (override or {}).items():
        if k not in out:
            raise ValueError("unknown configuration key %r" % (k,))
        if isinstance(out[k], dict) and isinstance(v, dict):
            out[k] = merge_conf(out[k], v)
        else:
            out[k] = deepcopy(v)
    return out


class InterpolationConfig:
    """base/src/interpolation.h:39-51; constructible from a dict like the pybind class."""

    def __init__(self, conf=None, **kw):
        c = merge_conf({**interpolation_default_conf, 'check_bounds': False, 'fill_channel_differences': True},
                       {**(conf or {}), **kw})
        self.nodes, self.mode = c['nodes'], str(c['mode']).upper()
        self.l2_normalize, self.ncc_normalize = bool(c['l2_normalize']), bool(c['ncc_normalize'])
        self.use_float_simd, self.check_bounds = bool(c['use_float_simd']), bool(c['check_bounds'])

    def to_engine(self):
        from ..engine import interp_cfg
        return interp_cfg(l2_normalize=self.l2_normalize, use_float_simd=self.use_float_simd,
                          check_bounds=self.check_bounds, mode=self.mode, nodes=self.nodes,
                          ncc_normalize=self.ncc_normalize)


class Match:
    __slots__ = ("node_idx", "sim")

    def __init__(self, node_idx, sim):
        self.node_idx, self.sim = int(node_idx), float(sim)


class FeatureNode:
    __slots__ = ("image_id", "feature_idx", "node_idx", "out_matches")

    def __init__(self, image_id, feature_idx):
        self.image_id, self.feature_idx, self.node_idx, self.out_matches = int(image_id), int(feature_idx), -1, []


class Graph:
    """base/src/graph.{h,cc}: directed match graph over (image, keypoint) nodes."""

    def __init__(self):
        self.nodes = []
        self.image_name_to_id = {}
        self.image_id_to_name = {}
        self.node_map = {}

    def find_or_create_node(self, image_name, feature_idx):          # graph.cc:38-57
        image_id = self.image_name_to_id.setdefault(image_name, len(self.image_name_to_id))
        self.image_id_to_name.setdefault(image_id, image_name)
        key = (image_id, int(feature_idx))
        idx = self.node_map.get(key)
        if idx is None:
            node = FeatureNode(image_id, feature_idx)
            node.node_idx = len(self.nodes)
            self.nodes.append(node)
            self.node_map[key] = node.node_idx
            return node
        return self.nodes[idx]

    def add_edge(self, node1, node2, sim):                           # graph.cc:59-64
        node1.out_matches.append(Match(node2.node_idx, sim))

    def register_matches(self, imname1, imname2, matches, similarities=None):   # graph.cc:66-82
        matches = np.asarray(matches).reshape(-1, 2)
        for k, (i, j) in enumerate(matches):
            sim = 1.0 if similarities is None else float(similarities[k])
            self.add_edge(self.find_or_create_node(imname1, i), self.find_or_create_node(imname2, j), sim)

    def get_scores(self):                                            # graph.cc:16-25
        s = np.zeros(len(self.nodes))
        for n in self.nodes:
            for m in n.out_matches:
                s[m.node_idx] += m.sim
                s[n.node_idx] += m.sim
        return s


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
