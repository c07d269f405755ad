"""GPU: REAL image content through the hot path (VERDICT r3 missing-2 / next-6; BASELINE configs[0]'s photographs).

The synthetic scenes are smooth cosine fields; here the patches are cut from feature maps of resampled photographs of the
reference's own demo set (tests/real_scene.py; fixtures tests/golden/real_image_tiles.npz made by
tests/golden/make_golden_real_images.py -- the GPU box has no /root/reference): descriptors that turn by ~0.9 of a unit
vector per texel, bicubic overshoot at edges, fp16 quantisation of real feature values, a 0.5 image-to-map scale.
  * pxr_arena_extract from the dense maps on the device == the restatement of extractor.py:152-199 / extract_patches.py:13-44;
  * keypoint adjustment and bundle adjustment (with and without inner iterations) on those patches == the oracle's, to
    north_star's tolerances; the refinement must also move the keypoints / points TOWARDS the known truth.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _ulp16(a, b):
    return np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32))


@pytest.fixture(scope="module")
def scene():
    import real_scene
    return real_scene.make_scene()


@pytest.fixture(scope="module")
def arena_and_patches(ctx, scene):
    """The producer: one pxr_arena_extract per view from the view's dense map on the device (observations are grouped by
    view for the call, the arena keeps observation order)."""
    import torch
    from pixsfm_amd.engine import PatchArena
    n_obs, C = len(scene["obs_image"]), scene["channels"]
    order = np.argsort(scene["obs_image"], kind="stable")
    arena = PatchArena(ctx, n_obs, 16, 16, C, np.float16)
    first = 0
    for v in range(len(scene["fmaps"])):
        sel = order[scene["obs_image"][order] == v]
        t = torch.from_numpy(scene["fmaps"][v]).cuda().contiguous()
        assert arena.extract(first, t, scene["detected"][sel], scene["image_size"]) == len(sel)
        first += len(sel)
    patches, corners, scales = arena.download()
    inv = np.empty(n_obs, np.int64); inv[order] = np.arange(n_obs)          # arena slot of observation i
    return arena, patches, corners, scales, order, inv


def test_patches_of_real_feature_maps_equal_the_reference_gather(ctx, scene, arena_and_patches):
    import torch
    import pxo_extract
    from pixsfm_amd.engine import PatchArena
    arena, patches, corners, scales, order, inv = arena_and_patches
    for v in range(len(scene["fmaps"])):
        sel = np.nonzero(scene["obs_image"] == v)[0]
        want, c, s = pxo_extract.sparse_patches(scene["fmaps"][v], scene["detected"][sel], scene["image_size"])
        got = patches[inv[sel]]
        assert np.array_equal(corners[inv[sel]], c) and np.allclose(scales[inv[sel]], s) and s[0] == 0.5
        d = _ulp16(got, want)                       # fp32 norm summed in another order: 1 ulp of fp16 at most
        assert d.max() <= 1 and (d == 0).mean() > 0.99
    # the gather itself, without normalisation, on the fp16 map and on the reference's 3-channel `image` model: bit-identical
    import real_scene
    img = real_scene.make_scene(n_views=2, n_points=60, kind="image")
    for fm, C, dt in ((scene["fmaps"][0].astype(np.float16), 128, np.float16), (img["fmaps"][1], 3, np.float32)):
        sc = scene if C == 128 else img
        v = 0 if C == 128 else 1
        sel = np.nonzero(sc["obs_image"] == v)[0]
        want, c, s = pxo_extract.sparse_patches(fm, sc["detected"][sel], sc["image_size"], l2_normalize=False, dtype=dt)
        a = PatchArena(ctx, len(sel), 16, 16, C, dt)
        a.extract(0, torch.from_numpy(fm).cuda().contiguous(), sc["detected"][sel], sc["image_size"], l2_normalize=False)
        got, gc, gs = a.download()
        assert np.array_equal(got, want) and np.array_equal(gc, c)
        a.close()


def _truth_error(scene, kp):
    """Distance of every non-root keypoint to its true position, up to the (fixed) root's own detection offset."""
    root = scene["node_const"].astype(bool)
    err = np.zeros(len(kp))
    for p in range(len(scene["xyz"])):
        ids = np.nonzero(scene["obs_point"] == p)[0]
        r = ids[root[ids]][0]
        off = scene["detected"][r] - scene["centers"][r]
        err[ids] = np.linalg.norm(kp[ids] - (scene["centers"][ids] + off), axis=1)
    return err[~root]


def test_keypoint_adjustment_on_real_texture_equals_the_oracle(ctx, scene, arena_and_patches):
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    arena, patches, corners, scales, order, inv = arena_and_patches
    prob = dict(kp=scene["detected"].copy(), node_patch=inv.astype(np.int64), node_const=scene["node_const"],
                node_problem=scene["node_problem"], edge_src=scene["edge_src"], edge_dst=scene["edge_dst"], edge_w=scene["edge_w"])
    ka = KAProblem(ctx, arena, prob)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, options=lm_options(parameter_tolerance=1e-5),
                          per_problem=True)
    kp = ka.keypoints()
    oracle = dict(prob, patches=patches, corners=corners, scales=scales)      # the SAME texels (downloaded from the arena)
    kpo, sums = pxo_ka.ka_solve(oracle, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    same = [g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"] for g, o in zip(per, sums)]
    # a rough cost landscape (13 .. 33 LM iterations per sub-problem): a borderline accept / reject may flip between two
    # implementations that agree to 1e-13 per residual; the sub-problems with the same trajectory must agree to 1e-4 px
    assert np.mean(same) >= 0.8, same
    for pidx, ok in enumerate(same):
        nodes = np.nonzero(scene["node_problem"] == pidx)[0]
        assert abs(per[pidx]["initial_cost"] - sums[pidx]["initial_cost"]) < 1e-10 * sums[pidx]["initial_cost"]
        if ok:
            assert np.abs(kp[nodes] - kpo[nodes]).max() < 1e-4
            assert abs(per[pidx]["final_cost"] - sums[pidx]["final_cost"]) < 1e-6 * sums[pidx]["initial_cost"]
        else:                                                                   # another local trajectory: same quality
            assert per[pidx]["final_cost"] < 1.05 * sums[pidx]["final_cost"] + 1e-9
    # and the refinement is real: featuremetric consistency pulls the detections towards the true correspondences
    e0, e1 = _truth_error(scene, scene["detected"]), _truth_error(scene, kp)
    assert total["final_cost"] < 0.5 * total["initial_cost"]
    assert np.median(e1) < 0.75 * np.median(e0), (np.median(e0), np.median(e1))
    assert np.abs(kp - scene["detected"]).max() <= 4.0 / 0.5 + 1e-9            # the box bound: 4 map texels = 8 image pixels at scale 0.5


@pytest.mark.parametrize("inner", [False, True])
def test_bundle_adjustment_on_real_texture_equals_the_oracle(ctx, scene, arena_and_patches, inner):
    import pxo
    from pixsfm_amd.engine import BAProblem, interp_cfg, lm_options, make_loss
    arena, patches, corners, scales, order, inv = arena_and_patches
    n_img, n_pts = len(scene["qvec"]), len(scene["xyz"])
    prob = {k: scene[k] for k in ("obs_image", "obs_point", "image_camera", "qvec", "tvec", "cam_model", "cam_params", "xyz")}
    prob["obs_patch"] = inv.astype(np.int64)
    prob["refs"] = np.zeros((n_pts, scene["channels"]))
    ba = BAProblem(ctx, arena, prob)
    ref_obs, _ = ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=100)   # at the initial parameters
    prob["refs"] = ba.d["refs"].download()
    assert (ref_obs >= 0).all() and np.allclose(np.linalg.norm(prob["refs"], axis=1), 1.0, atol=1e-9)
    # a planar scene leaves focal length and distance coupled: intrinsics stay fixed, camera 0 and one coordinate fix the gauge
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    gauge = (pose_const, tmask, np.full(n_img, 0b1111, np.uint16), np.zeros(n_pts, np.uint8))
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=6, use_inner_iterations=inner))
    q, t, k, X = ba.params()
    oracle = dict(prob, patches=patches, corners=corners, scales=scales)
    so, qo, to, ko, Xo = pxo.ba_solve(oracle, pxo.cfg(), pxo.loss("cauchy", 0.25), *gauge,
                                      pxo.lm_options(max_iterations=6, use_inner_iterations=int(inner)))
    assert abs(s["initial_cost"] - so["initial_cost"]) < 1e-10 * so["initial_cost"]
    assert s["iterations"] == so["iterations"] and s["num_successful"] == so["num_successful"]
    tol = 1e-4 if inner else 1e-7               # (inner iterations: nested LMs with their own 1e-6 tolerances, north_star's 1e-4)
    assert abs(s["final_cost"] - so["final_cost"]) < tol * so["initial_cost"]
    assert np.abs(q - qo).max() < 1e-4 and np.abs(t - to).max() < 1e-4 and np.abs(X - Xo).max() < 1e-4
    assert s["final_cost"] < 0.9 * s["initial_cost"]
