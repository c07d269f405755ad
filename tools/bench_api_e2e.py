#!/usr/bin/env python
"""End-to-end wall time of the DROP-IN calls on host-resident inputs -- what a pixsfm user pays, set-up included (the
reference's `BA Time` / `KA Time` log lines cover SetUp + Solve too: bundle_optimizer.h:236-241,
featuremetric_keypoint_optimizer.h:100-108):

    BundleAdjuster.create(conf).refine_multilevel(reconstruction, feature_manager)       BASELINE configs[2] shape
    KeypointAdjuster.create(conf).refine_multilevel(keypoints, feature_manager, graph)   BASELINE configs[1] shape

split into the phases of pixsfm_amd.api._timing (dump of the Python objects / native problem construction / upload /
references / solve / write-back).  Inputs are synthetic, rendered on the GPU and brought to HOST memory first (FeaturePatch
objects over numpy views, a Python Reconstruction) -- building them is not timed.  `--device-resident` also runs the
MI355X-native flow: the same problem with the patches left in a device arena (features.ArenaPatch), no PCIe crossing.

One JSON line.  bench.py --api-e2e embeds it as `api_e2e`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _report(wall, phases):
    ph = dict(phases)
    if "build_problem" in ph and "dump" in ph:
        ph["dump"] -= ph["build_problem"]          # the native construction runs inside the dump
    known = sum(ph.values())
    ph["other"] = max(0.0, wall - known)
    return {"wall_s": round(wall, 4), "phases_s": {k: round(v, 4) for k, v in sorted(ph.items())}}


def ba_inputs(dev, n_cams, n_points, opp, device_resident, ctx):
    from pixsfm_amd import synthetic_gpu
    from pixsfm_amd.api import features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    from pixsfm_amd.engine import PatchArena
    prob, patches = synthetic_gpu.make_ba_problem_gpu(dev, n_cams=n_cams, n_points=n_points, obs_per_point=opp, seed=2,
                                                      channels=128, patch_size=16)
    prob["obs_patch"] = np.arange(len(prob["obs_image"]), dtype=np.int64)
    rec, patch_of = reconstruction_from_flat(prob)
    names = {i: rec.images[i].name for i in rec.images}
    fmaps = {names[i]: features.FeatureMap() for i in rec.images}
    keep = None
    if device_resident:
        arena = PatchArena(ctx, len(prob["obs_image"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
        arena.upload(0, None, prob["corners"], prob["scales"])
        for (image_id, p2d), pi in patch_of.items():
            fmaps[names[image_id]].patches[p2d] = features.ArenaPatch(arena, pi)
        keep = (arena, patches)
    else:
        # what a pixsfm extractor hands over (extract.py:131-139): per image ONE N x H x W x C array + keypoint ids + corners,
        # wrapped by the reference's numpy constructor FeatureMap(patches, point2D_ids, corners, metadata).  The synthetic scene
        # is rendered in observation (point-major) order, so it is brought into image-major order on the device first.
        ids = np.array([(image_id, p2d, pi) for (image_id, p2d), pi in patch_of.items()], dtype=np.int64)
        ids = ids[np.lexsort((ids[:, 1], ids[:, 0]))]
        host = patches[torch.as_tensor(ids[:, 2], device=patches.device)].cpu().numpy()      # image-major, host memory
        del patches
        torch.cuda.empty_cache()
        corners, scales = prob["corners"][ids[:, 2]], prob["scales"][ids[:, 2]]
        bounds = np.flatnonzero(np.diff(ids[:, 0], prepend=-1, append=-2))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            fmaps[names[int(ids[lo, 0])]] = features.FeatureMap(host[lo:hi], ids[lo:hi, 1], corners[lo:hi],
                                                                {"scale": scales[lo], "is_sparse": True, "patch_size": 16})
        keep = host
    return rec, features.FeatureManager([features.FeatureSet(fmaps)]), keep, len(prob["obs_image"])


def run_ba(dev, ctx, n_cams, n_points, opp, iters, device_resident):
    from pixsfm_amd.api import BundleAdjuster, _timing
    from pixsfm_amd.api import bundle_adjustment, keypoint_adjustment
    keypoint_adjustment._default_ctx = ctx
    rec, manager, keep, n_obs = ba_inputs(dev, n_cams, n_points, opp, device_resident, ctx)
    ba = BundleAdjuster.create({"optimizer": {"solver": {"max_num_iterations": iters}}, "references": {"iters": 100}})
    torch.cuda.synchronize()
    _timing.start()
    t0 = time.perf_counter()
    out = ba.refine_multilevel(rec, manager)
    wall = time.perf_counter() - t0
    rep = _report(wall, _timing.stop())
    s = out["summary"][0]
    rep.update(n_obs=int(n_obs), n_points=int(n_points), n_cams=int(n_cams), lm_iterations=int(s.num_iterations),
               initial_cost=float(s.initial_cost), final_cost=float(s.final_cost),
               inputs="device arena (ArenaPatch)" if device_resident else "host FeaturePatch objects (numpy), Python Reconstruction")
    del keep
    return rep


def run_ka(dev, ctx, n_tracks, track_len, device_resident):
    import bench_ka
    from pixsfm_amd.api import KeypointAdjuster, _timing, features
    from pixsfm_amd.api import keypoint_adjustment
    from pixsfm_amd.api.keypoint_adjustment import build_matching_graph
    from pixsfm_amd.engine import PatchArena
    keypoint_adjustment._default_ctx = ctx
    prob, patches = bench_ka.make_problem_gpu(dev, n_tracks, track_len)
    n = n_tracks * track_len
    # node k of track t is keypoint t of image k: `track_len` images with n_tracks keypoints each
    img, kid = np.arange(n) % track_len, np.arange(n) // track_len
    names = ["im%03d" % k for k in range(track_len)]
    keypoints = {names[k]: prob["kp"][img == k].copy() for k in range(track_len)}
    if device_resident:
        arena = PatchArena(ctx, n, 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
        arena.upload(0, None, prob["corners"], prob["scales"])
        fmaps = {}
        for k in range(track_len):
            fm = features.FeatureMap()
            for node in np.flatnonzero(img == k):
                fm.patches[int(kid[node])] = features.ArenaPatch(arena, int(node))
            fmaps[names[k]] = fm
        keep = (arena, patches)
    else:
        host = patches.cpu().numpy()
        del patches
        torch.cuda.empty_cache()
        fmaps = {names[k]: features.FeatureMap.from_arrays(host[img == k], kid[img == k], prob["corners"][img == k], (1.0, 1.0))
                 for k in range(track_len)}
        keep = host
    pairs, matches, scores = [], [], []
    es, ed, ew = prob["edge_src"], prob["edge_dst"], prob["edge_w"]
    for a in range(track_len):
        for b in range(a + 1, track_len):
            sel = (img[es] == a) & (img[ed] == b)
            pairs.append((names[a], names[b]))
            matches.append(np.stack([kid[es[sel]], kid[ed[sel]]], 1))
            scores.append(ew[sel])
    graph = build_matching_graph(pairs, matches, scores)
    ka = KeypointAdjuster.create({"strategy": "featuremetric"})
    torch.cuda.synchronize()
    _timing.start()
    t0 = time.perf_counter()
    out = ka.refine_multilevel(keypoints, features.FeatureManager([features.FeatureSet(fmaps)]), graph)
    wall = time.perf_counter() - t0
    rep = _report(wall, _timing.stop())
    s = out["summary"][0]
    rep.update(n_nodes=int(n), n_tracks=int(n_tracks), residual_blocks=int(s.num_residuals_reduced // 128),
               initial_cost=float(s.initial_cost), final_cost=float(s.final_cost),
               inputs="device arena (ArenaPatch)" if device_resident else "host FeaturePatch objects (numpy), Python Graph")
    del keep
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=200)
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--obs-per-point", type=int, default=5)
    ap.add_argument("--lm-iters", type=int, default=10)
    ap.add_argument("--ka-tracks", type=int, default=10_000)
    ap.add_argument("--ka-track-len", type=int, default=10)
    ap.add_argument("--device-resident", action="store_true", help="also run the flows with the patches left in a device arena")
    ap.add_argument("--skip-ba", action="store_true")
    ap.add_argument("--skip-ka", action="store_true")
    args = ap.parse_args()
    from pixsfm_amd.engine import Context
    dev = "cuda:0"
    torch.cuda.set_device(0)
    ctx = Context(0)
    out = {"host_cores": os.cpu_count()}
    if not args.skip_ba:
        out["ba_host"] = run_ba(dev, ctx, args.cams, args.points, args.obs_per_point, args.lm_iters, False)
        if args.device_resident:
            out["ba_device"] = run_ba(dev, ctx, args.cams, args.points, args.obs_per_point, args.lm_iters, True)
    if not args.skip_ka:
        out["ka_host"] = run_ka(dev, ctx, args.ka_tracks, args.ka_track_len, False)
        if args.device_resident:
            out["ka_device"] = run_ka(dev, ctx, args.ka_tracks, args.ka_track_len, True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
