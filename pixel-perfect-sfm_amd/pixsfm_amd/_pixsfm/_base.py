"""`_pixsfm._base` (pixsfm/base/bindings.cc:29-154)."""
import os

from ..api.base import (FeatureNode, Graph, InterpolationConfig, Match, compute_root_labels,  # noqa: F401
                        compute_score_labels, compute_track_labels)


class InterpolatorType(str):
    """String-constructible enum of the reference; only BICUBIC is on the accelerated path."""
    BICUBIC = "BICUBIC"


class Map_NameKeypoints(dict):
    """Opaque std::unordered_map<std::string, Eigen::Matrix<double, -1, 2, RowMajor>> of the reference
    (base/bindings.cc:118-131): a dict of (n, 2) float64 arrays here, refined in place."""


def count_track_edges(graph, track_labels):
    """CountTrackEdges (base/src/graph.cc:283-303): intra-track matches per track."""
    n = len(set(track_labels))
    out = [0] * n
    for nd in graph.nodes:
        for m in nd.out_matches:
            if track_labels[nd.node_idx] == track_labels[m.node_idx]:
                out[track_labels[nd.node_idx]] += 1
    return out


def count_edges_AB(graph, track_labels, is_root):
    """CountEdgesAB (base/src/graph.cc:258-281): per track (intra-track edges touching the root, the others)."""
    n = len(set(track_labels))
    out = [[0, 0] for _ in range(n)]
    for nd in graph.nodes:
        for m in nd.out_matches:
            if track_labels[nd.node_idx] == track_labels[m.node_idx]:
                touches_root = bool(is_root[nd.node_idx]) or bool(is_root[m.node_idx])
                out[track_labels[nd.node_idx]][0 if touches_root else 1] += 1
    return [tuple(x) for x in out]


def get_effective_num_threads(n):
    return (os.cpu_count() or 1) if n <= 0 else int(n)
