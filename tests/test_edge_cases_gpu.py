"""GPU: empty and degenerate inputs through the C-ABI (the reference guards these with THROW_CHECKs /
early returns: bundle_optimizer.h:174-176, query_keypoint_optimizer.h:56-59, reference_extractor.h:216-237)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_and_degenerate_inputs(ctx):
    from pixsfm_amd import PixsfmHipError, synthetic, synthetic_ka
    from pixsfm_amd.engine import (BAProblem, PatchArena, interp_cfg, interpolate, lm_options, make_loss,
                                   nearest_references)
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic.make_ba_problem(n_cams=3, n_points=12, obs_per_point=2, seed=1)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    # --- BA with no observations: evaluation is a no-op, the solve refuses (NumResiduals() == 0)
    empty = dict(prob)
    for k in ("obs_image", "obs_point", "obs_patch"):
        empty[k] = prob[k][:0]
    ba0 = BAProblem(ctx, arena, empty)
    ba0.eval(interp_cfg(), with_jacobian=True)
    gauge = (np.array([1, 0, 0], np.uint8), np.zeros(3, np.uint8), np.full(3, 0b0110, np.uint16), np.zeros(12, np.uint8))
    with pytest.raises(PixsfmHipError):
        ba0.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge)
    # --- every parameter block constant: rejected, nothing to optimise
    ba = BAProblem(ctx, arena, prob)
    with pytest.raises(PixsfmHipError):
        ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), np.ones(3, np.uint8), np.zeros(3, np.uint8),
                 np.full(3, 0b1111, np.uint16), np.ones(12, np.uint8))
    # --- a point without observations gets no reference (ref_obs = -1), the others are unaffected
    ragged = dict(prob)
    keep = prob["obs_point"] != 5
    for k in ("obs_image", "obs_point", "obs_patch"):
        ragged[k] = prob[k][keep]
    bar = BAProblem(ctx, arena, ragged)
    chosen, _ = bar.compute_references(interp_cfg(), make_loss("cauchy", [0.25]), iters=5)
    assert chosen[5] == -1 and (np.delete(chosen, 5) >= 0).all()
    # --- only the points vary (cameras constant): no reduced camera system at all
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), np.ones(3, np.uint8), np.zeros(3, np.uint8),
                 np.full(3, 0b1111, np.uint16), np.zeros(12, np.uint8), options=lm_options(max_iterations=3))
    assert s["num_camera_unknowns"] == 0 and s["num_point_unknowns"] == 36 and s["final_cost"] <= s["initial_cost"]
    # --- KA: no sub-problems / no edges at all / a problem of one isolated node
    kprob = synthetic_ka.make_ka_problem(n_tracks=2, track_len=3, seed=2)
    karena = PatchArena.from_numpy(ctx, kprob["patches"], kprob["corners"], kprob["scales"])
    none = dict(kprob)
    none.update(edge_src=np.zeros(0, np.int32), edge_dst=np.zeros(0, np.int32), edge_w=np.zeros(0))
    ka = KAProblem(ctx, karena, none)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), per_problem=True)
    assert total["initial_cost"] == 0.0 and all(p["iterations"] == 0 for p in per)
    assert np.array_equal(ka.keypoints(), kprob["kp"])
    ka.eval(interp_cfg(), make_loss("cauchy", [0.25]))              # zero edges: no-op
    # --- batched helpers with n = 0
    d, J = interpolate(ctx, karena, interp_cfg(), np.zeros((0, 2)), np.zeros(0, np.int64), jacobian=True)
    assert d.shape == (0, 128) and J.shape == (0, 128, 2)
    assert karena.extract.__self__ is karena                         # bound method exists; n = 0 extract is a no-op:
    best, dist, _ = nearest_references(ctx, karena, interp_cfg(), kprob["kp"][:2], [0, 1], [0, 0, 1],
                                       np.ones((1, 128)) / np.sqrt(128))
    assert best.tolist() == [-1, 0]                                  # no candidates -> -1


@pytest.mark.parametrize("stage_bytes", [4096, 20480, 1 << 20])
def test_upload_with_patches_larger_than_the_staging_buffer(stage_bytes, monkeypatch):
    """pxr_arena_upload / pxr_arena_upload_gather cut the upload into staging-buffer-sized pieces of ONE byte stream: a
    patch larger than the pinned buffer (a dense 1600 x 1200 x 128 fp16 map is 491 MB against 256 MiB) spans several
    pieces, a piece may begin and end inside a patch.  PXR_UPLOAD_STAGE_BYTES shrinks the buffer so that 36 KB patches
    play the dense map's part."""
    from pixsfm_amd.engine import Context, PatchArena
    monkeypatch.setenv("PXR_UPLOAD_STAGE_BYTES", str(stage_bytes))
    monkeypatch.setenv("PXR_UPLOAD_THREADS", "3")
    c = Context(0)                                   # a fresh context: the staging buffers are sized at first use
    try:
        rng = np.random.default_rng(stage_bytes)
        n, H, W, ch = 150, 12, 12, 128               # 36 864-byte patches, 5.5 MB in total (> the 4 MB threading threshold)
        patches = rng.standard_normal((n, H, W, ch)).astype(np.float16)
        corners = rng.integers(0, 100, (n, 2)).astype(np.int32)
        scales = rng.uniform(0.5, 2.0, (n, 2))
        a = PatchArena.from_numpy(c, patches, corners, scales)
        got, gc, gs = a.download()
        assert np.array_equal(got.view(np.uint16), patches.view(np.uint16))
        assert np.array_equal(gc, corners) and np.array_equal(gs, scales)
        # separate host patches (FeaturePatch objects), into the middle of an arena
        sep = [np.ascontiguousarray(p) for p in patches[::-1]]
        ptrs = np.array([p.ctypes.data for p in sep], np.uint64)
        b = PatchArena.from_patch_pointers(c, ptrs, (H, W, ch), np.float16, corners, scales)
        got2, _, _ = b.download()
        assert np.array_equal(got2.view(np.uint16), patches[::-1].view(np.uint16))
        a.upload(7, patches[100:120], corners[100:120], scales[100:120])
        got3, _, _ = a.download(7, 20)
        assert np.array_equal(got3.view(np.uint16), patches[100:120].view(np.uint16))
        # a later, larger upload on the same context grows the buffers
        monkeypatch.setenv("PXR_UPLOAD_STAGE_BYTES", str(4 * stage_bytes))
        a.upload(0, patches, corners, scales)
        got4, _, _ = a.download()
        assert np.array_equal(got4.view(np.uint16), patches.view(np.uint16))
    finally:
        c.close()
