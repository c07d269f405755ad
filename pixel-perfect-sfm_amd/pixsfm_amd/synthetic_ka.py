"""Synthetic keypoint-adjustment instances (BASELINE.json configs[1]):
tracks of `track_len` nodes, one per distinct image, complete intra-track match graph with
similarities ~ U(0.5, 1), one 16x16xC fp16 patch per node rendered from a per-track smooth
field so that the featuremetric optimum is the true location; initial keypoints = true
location + N(0, sigma) px.  Mirrors what pixsfm's extractor produces for sparse patches
(pixsfm/features/extractor.py:179-236).
"""
import numpy as np

from . import synthetic
from .ka_engine import pack_tracks_into_problems


def make_ka_problem(n_tracks=20, track_len=5, channels=128, patch_size=16, seed=1, dtype=np.float16,
                    sigma=1.0, max_kps_per_problem=50, scale=(1.0, 1.0), noise=0.0, image_size=1000.0,
                    directed_both=True):
    """Returns a dict with the flat arrays of pxr_ka_view (+ ground truth)."""
    rng = np.random.default_rng(seed)
    n_nodes = n_tracks * track_len
    track_of_node = np.repeat(np.arange(n_tracks), track_len)
    true_xy = rng.uniform(50, image_size - 50, (n_nodes, 2)) + rng.uniform(-0.5, 0.5, (n_nodes, 2))
    scales = np.tile(np.asarray(scale, dtype=np.float64), (n_nodes, 1))
    # the detector's (noisy) keypoint is what the patch is cropped around (extractor.py:192-193)
    kp0 = true_xy + rng.normal(0, sigma, (n_nodes, 2))
    corners = np.floor(kp0 * scales - patch_size / 2.0).astype(np.int32)
    A = rng.normal(0, 1, (n_tracks, channels, synthetic.N_BASIS))
    patches = synthetic.render_patches(A[track_of_node], true_xy, corners, scales, patch_size, dtype, noise, rng)
    # complete intra-track graph; a match i->j (and j->i when directed_both) like
    # Graph::RegisterMatches on mutual matches (pixsfm/base/src/graph.cc)
    src, dst, w = [], [], []
    for t in range(n_tracks):
        ids = np.arange(t * track_len, (t + 1) * track_len)
        for a in range(track_len):
            for b in range(a + 1, track_len):
                sim = rng.uniform(0.5, 1.0)
                src.append(ids[a]); dst.append(ids[b]); w.append(sim)
                if directed_both:
                    src.append(ids[b]); dst.append(ids[a]); w.append(sim)
    edge_src, edge_dst = np.array(src, dtype=np.int32), np.array(dst, dtype=np.int32)
    edge_w = np.array(w, dtype=np.float64)
    # root = node with the highest summed similarity in its track (compute_root_labels semantics)
    score = np.zeros(n_nodes)
    np.add.at(score, edge_src, edge_w)
    np.add.at(score, edge_dst, edge_w)
    node_const = np.zeros(n_nodes, dtype=np.uint8)
    for t in range(n_tracks):
        ids = np.arange(t * track_len, (t + 1) * track_len)
        node_const[ids[np.argmax(score[ids])]] = 1
    problem_of_node, bins = pack_tracks_into_problems(track_of_node, max_kps_per_problem)
    problem_of_node = np.array(problem_of_node, dtype=np.int32)
    return dict(kp=kp0.copy(), node_patch=np.arange(n_nodes, dtype=np.int64), node_const=node_const,
                node_problem=problem_of_node, edge_src=edge_src, edge_dst=edge_dst, edge_w=edge_w,
                patches=patches, corners=corners, scales=scales, true_xy=true_xy,
                track_of_node=track_of_node, n_problems=len(bins))
