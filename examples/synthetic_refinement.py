#!/usr/bin/env python
"""End-to-end example on a synthetic scene, written the way a pixsfm user drives the reference
(pixsfm/keypoint_adjustment/main.py, pixsfm/bundle_adjustment/main.py): featuremetric keypoint adjustment, then
featuremetric bundle adjustment, then the same BA through the low-memory cost-map strategy.

    python examples/synthetic_refinement.py            # needs an MI355X and the built libpixsfm_hip.so
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))

from pixsfm_amd import synthetic, synthetic_ka                                    # noqa: E402
from pixsfm_amd.api import BundleAdjuster, KeypointAdjuster, features             # noqa: E402
from pixsfm_amd.api.keypoint_adjustment import build_matching_graph              # noqa: E402
from pixsfm_amd.api.reconstruction import reconstruction_from_flat               # noqa: E402


def keypoint_adjustment():
    n_tracks, track_len = 200, 6
    prob = synthetic_ka.make_ka_problem(n_tracks=n_tracks, track_len=track_len, seed=1, directed_both=False)
    n = n_tracks * track_len
    img, kid = np.arange(n) % track_len, np.arange(n) // track_len                # node (t, k): keypoint t of image k
    names = ["image%02d.jpg" % k for k in range(track_len)]
    keypoints = {names[k]: prob["kp"][img == k].copy() for k in range(track_len)}
    pairs, matches, scores = [], [], []
    for a in range(track_len):
        for b in range(a + 1, track_len):
            sel = (img[prob["edge_src"]] == a) & (img[prob["edge_dst"]] == b)
            pairs.append((names[a], names[b]))
            matches.append(np.stack([kid[prob["edge_src"][sel]], kid[prob["edge_dst"][sel]]], 1))
            scores.append(prob["edge_w"][sel])
    graph = build_matching_graph(pairs, matches, scores)
    fmaps = {names[k]: features.FeatureMap.from_arrays(prob["patches"][img == k], kid[img == k], prob["corners"][img == k], (1.0, 1.0))
             for k in range(track_len)}
    truth = {names[k]: prob["gt_kp"][img == k] for k in range(track_len)} if "gt_kp" in prob else None
    out = KeypointAdjuster.create({"strategy": "featuremetric"}).refine_multilevel(
        keypoints, features.FeatureManager([features.FeatureSet(fmaps)]), graph)
    s = out["summary"][0]
    print("KA : %d keypoints, cost %.4g -> %.4g, %s" % (n, s.initial_cost, s.final_cost, s.termination_type))
    if truth is not None:
        err = np.concatenate([np.linalg.norm(keypoints[k] - truth[k], axis=1) for k in names])
        print("     median keypoint error after KA: %.4f px" % np.median(err))


def bundle_adjustment(strategy):
    prob = synthetic.make_ba_problem(n_cams=12, n_points=1500, obs_per_point=5, seed=3, noise=0.02)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fmaps.setdefault(rec.images[image_id].name, features.FeatureMap()).patches[p2d] = \
            features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    out = BundleAdjuster.create({"strategy": strategy}).refine_multilevel(rec, features.FeatureManager([features.FeatureSet(fmaps)]))
    s = out["summary"][0]
    # (the references are the observations closest to the robust mean at the PERTURBED projections and every image carries
    # its own feature noise, so the optimum is not the ground truth; the cost is what the refinement minimises)
    print("BA (%s): %d observations, cost %.4g -> %.4g in %d iterations (%.1f ms), %s"
          % (strategy, len(prob["obs_image"]), s.initial_cost, s.final_cost, s.num_iterations, s.total_time_in_seconds * 1e3,
             s.termination_type))


if __name__ == "__main__":
    keypoint_adjustment()
    bundle_adjustment("feature_reference")
    bundle_adjustment("costmaps")
