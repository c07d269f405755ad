"""Mirror of `pixsfm._pixsfm._base` (pixsfm/base/bindings.cc:29-154) for the accelerated path:
Graph / FeatureNode / Match, track / score / root labelling, InterpolationConfig, default confs.

Graph labelling is host pre-processing (component 6; section 8f row 3): native host
code in libpixsfm_hip.so (csrc/pxr_graph.cpp), like the reference's graph.cc; it is not on the GPU path.
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


def merge_adjuster_conf(default, override):
    """The adjusters' own `OmegaConf.merge(self.default_conf, conf)` (keypoint_adjustment/main.py:87, bundle_adjustment/
    main.py:66-74) accepts keys the adjuster does not know -- pixsfm's YAML files carry pipeline-level entries such as
    `repeats` / `num_threads` next to the adjuster's (configs/low_memory.yaml:24-31) -- while the nested sections that
    become C++ option classes (`optimizer`, `references`, `costmaps`, `interpolation`) are strict."""
    override = dict(override or {})
    extra = {k: deepcopy(override.pop(k)) for k in list(override) if k not in default}
    out = merge_conf(default, override)
    out.update(extra)
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

    def __init__(self, node_idx, similarity=None, sim=None):         # base/bindings.cc:46-55: (node_idx, similarity)
        self.node_idx, self.sim = int(node_idx), float(similarity if similarity is not None else sim)

    similarity = property(lambda self: self.sim, lambda self, v: setattr(self, "sim", float(v)))


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

    def add_node(self, image, feature_idx):                          # graph.cc:98-113 (both overloads)
        """Appends a node WITHOUT registering it in the (image, keypoint) lookup of find_or_create_node -- like the
        reference's AddNode; `image` is an image name (registered on first use) or an image id.  Returns the node index."""
        if isinstance(image, str):
            image_id = self.image_name_to_id.setdefault(image, len(self.image_name_to_id))
            self.image_id_to_name.setdefault(image_id, image)
        else:
            image_id = int(image)
        node = FeatureNode(image_id, feature_idx)
        node.node_idx = len(self.nodes)
        self.nodes.append(node)
        return node.node_idx

    def degrees(self):                                               # graph.cc:5-14
        d = [0] * len(self.nodes)
        for n in self.nodes:
            d[n.node_idx] += len(n.out_matches)
            for m in n.out_matches:
                d[m.node_idx] += 1
        return d

    def scores(self):                                                # graph.cc:16-25 (the bound name; get_scores: same)
        return [float(x) for x in self.get_scores()]

    def edges(self):                                                 # graph.cc:27-36
        return [(n.node_idx, m.node_idx, m.sim) for n in self.nodes for m in n.out_matches]

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


def _flat_graph(graph):
    """Flat host arrays of the graph in the reference's enumeration order (for node in nodes: for match in
    node.out_matches, graph.cc:133-139)."""
    node_image = np.array([nd.image_id for nd in graph.nodes], dtype=np.int32)
    src, dst, sim = [], [], []
    for nd in graph.nodes:
        for m in nd.out_matches:
            src.append(nd.node_idx); dst.append(m.node_idx); sim.append(m.sim)
    return (node_image, np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64), np.array(sim, dtype=np.float64))


def compute_track_labels(graph):
    """_base.compute_track_labels (base/bindings.cc:29-131 -> ComputeTrackLabels, graph.cc:126-206): maximum-spanning-
    forest union-find over the matches in descending (sim, src, dst) order, never merging two components that share
    an image.  Native host code (pxr_graph_track_labels), like the reference's."""
    from .. import _lib
    lib = _lib.load()
    node_image, src, dst, sim = _flat_graph(graph)
    labels = np.empty(len(node_image), dtype=np.int64)
    _lib.check(lib.pxr_graph_track_labels(len(node_image), node_image.ctypes.data, len(src), src.ctypes.data,
                                          dst.ctypes.data, sim.ctypes.data, labels.ctypes.data, None),
               "pxr_graph_track_labels")
    return labels.tolist()


def compute_labels_on_device(graph, ctx=None):
    """Track labels, score labels and root labels in one call on the GPU (pxr_graph_labels_device): the flat graph is
    uploaded once, the three results come back as (track_labels list, scores array, root_labels list of bool) --
    identical to compute_track_labels / compute_score_labels / compute_root_labels."""
    import ctypes as C
    from .. import _lib
    from .keypoint_adjustment import default_context
    ctx = ctx or default_context()
    node_image, src, dst, sim = _flat_graph(graph)
    n = len(node_image)
    d_img, d_src, d_dst, d_sim = (ctx.to_device(a, dt) for a, dt in ((node_image, np.int32), (src, np.int64), (dst, np.int64),
                                                                    (sim, np.float64)))
    labels, scores, roots = ctx.empty((max(n, 1),), np.int64), ctx.empty((max(n, 1),), np.float64), ctx.empty((max(n, 1),), np.uint8)
    n_tracks = C.c_int64()
    try:
        _lib.check(ctx.lib.pxr_graph_labels_device(ctx.handle, n, d_img.ptr, len(src), d_src.ptr, d_dst.ptr, d_sim.ptr, labels.ptr,
                                                   scores.ptr, roots.ptr, C.byref(n_tracks)), "pxr_graph_labels_device")
    except _lib.PixsfmHipError as e:
        if e.code != _lib.PXR_EUNSUPPORTED:
            raise
        # one giant connected component (bad matches chaining tracks): the device kernel's per-component rank sort is
        # quadratic in ONE wavefront there -- the native host labelling (the reference's own shape, graph.cc) takes over
        track_labels = compute_track_labels(graph)
        score_labels = compute_score_labels(graph, track_labels)
        return track_labels, np.asarray(score_labels), compute_root_labels(graph, track_labels, score_labels)
    return labels.download()[:n].tolist(), scores.download()[:n], [bool(r) for r in roots.download()[:n]]


def compute_score_labels(graph, track_labels):                       # graph.cc:208-223
    from .. import _lib
    lib = _lib.load()
    node_image, src, dst, sim = _flat_graph(graph)
    tl = np.ascontiguousarray(track_labels, dtype=np.int64)
    if len(tl) != len(node_image):
        raise ValueError("track_labels must have one entry per graph node")
    scores = np.empty(len(node_image), dtype=np.float64)
    _lib.check(lib.pxr_graph_score_labels(len(node_image), len(src), src.ctypes.data, dst.ctypes.data, sim.ctypes.data,
                                          tl.ctypes.data, scores.ctypes.data), "pxr_graph_score_labels")
    return scores.tolist()


def compute_root_labels(graph, track_labels, score_labels):          # graph.cc:225-256
    from .. import _lib
    lib = _lib.load()
    n = len(graph.nodes)
    tl = np.ascontiguousarray(track_labels, dtype=np.int64)
    sc = np.ascontiguousarray(score_labels, dtype=np.float64)
    if len(tl) != n or len(sc) != n:
        raise ValueError("track_labels / score_labels must have one entry per graph node")
    is_root = np.empty(n, dtype=np.uint8)
    _lib.check(lib.pxr_graph_root_labels(n, tl.ctypes.data, sc.ctypes.data, is_root.ctypes.data), "pxr_graph_root_labels")
    return [bool(v) for v in is_root]
