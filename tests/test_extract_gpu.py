"""GPU: pxr_arena_extract (dense map on the device -> arena patches) vs the oracle restatement and the
golden fixture made with the reference's own gather.  Bit-exact without normalisation; with it the
fp32 norm's summation order differs, so fp16 values agree to 1 ulp (written below)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_ref.npz")


def _ulp16(a, b):
    return np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32))


def test_extract_matches_reference_fixture(ctx):
    import torch
    import pxo_extract
    from pixsfm_amd.engine import PatchArena
    g = np.load(GOLD)
    fmap, kps, size = pxo_extract.golden_inputs()
    arena = PatchArena(ctx, len(kps) + 2, 16, 16, 128, np.float16)
    t = torch.from_numpy(fmap).cuda()[None].contiguous()
    assert arena.extract(1, t, kps, size) == len(kps)          # offset 1: ranges inside the arena
    patches, corners, scales = arena.download(1, len(kps))
    assert np.array_equal(corners, g["corners"])
    assert np.array_equal(scales, np.tile(g["scale"], (len(kps), 1)))
    d = _ulp16(patches, g["patches"])
    assert d.max() <= 1 and (d == 0).mean() > 0.995


@pytest.mark.parametrize("src,dst,channels", [("float32", np.float32, 128), ("float16", np.float16, 128),
                                               ("float32", np.float16, 64), ("float16", np.float64, 64)])
def test_plain_gather_is_bit_exact(ctx, src, dst, channels):
    import torch
    import pxo_extract
    from pixsfm_amd.engine import PatchArena
    rng = np.random.default_rng(5)
    fmap = rng.normal(0, 1, (channels, 60, 75)).astype(src)
    kps = rng.uniform(-5, 310, (200, 2))
    size = (300.0, 240.0)
    want, corners, scale = pxo_extract.sparse_patches(fmap, kps, size, l2_normalize=False, dtype=dst)
    arena = PatchArena(ctx, len(kps), 16, 16, channels, dst)
    arena.extract(0, torch.from_numpy(fmap).cuda(), kps, size, l2_normalize=False)
    patches, c, s = arena.download()
    assert np.array_equal(c, corners) and np.array_equal(s, np.tile(scale, (len(kps), 1)))
    assert np.array_equal(patches, want)


def test_normalised_patches_feed_the_ka_kernels(ctx):
    """End to end on the device: produce patches with pxr_arena_extract and evaluate KA edges on
    them; the same edges on oracle-produced patches uploaded from the host agree to 1e-3 (1-ulp
    fp16 texel differences), and are identical when the producer output is fed to both."""
    import torch
    import pxo
    import pxo_extract
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    rng = np.random.default_rng(9)
    fmap = rng.normal(0, 1, (128, 48, 64)).astype(np.float32)
    size = (256.0, 192.0)
    kps = rng.uniform(20, 170, (40, 2))
    arena = PatchArena(ctx, len(kps), 16, 16, 128, np.float16)
    arena.extract(0, torch.from_numpy(fmap).cuda(), kps, size)
    patches, corners, scales = arena.download()
    want, _, _ = pxo_extract.sparse_patches(fmap, kps, size)
    assert _ulp16(patches, want).max() <= 1
    prob = dict(kp=kps + rng.normal(0, 0.3, kps.shape), node_patch=np.arange(40, dtype=np.int64),
                node_const=np.zeros(40, np.uint8), node_problem=np.zeros(40, np.int32),
                edge_src=np.arange(0, 39, dtype=np.int32), edge_dst=np.arange(1, 40, dtype=np.int32), edge_w=np.ones(39))
    cost, r, _, _ = KAProblem(ctx, arena, prob).eval(interp_cfg(), make_loss("cauchy", [0.25]), materialize=True)
    r = r.download()
    # oracle on the SAME texels (downloaded from the arena): exact-level agreement
    cfg = pxo.cfg()
    for e in (0, 17, 38):
        p1 = pxo.make_patch(patches[e], corners[e], scales[e])
        p2 = pxo.make_patch(patches[e + 1], corners[e + 1], scales[e + 1])
        ro, _, _ = pxo.ka_residual(p1, p2, cfg, prob["kp"][e], prob["kp"][e + 1])
        assert np.abs(r[e] - ro).max() < 1e-12


def test_full_size_extract_properties(ctx):
    """BASELINE config-2 scale: 100k keypoints from one 128 x 256 x 320 map (6.5 GB of patches).
    Size-independent checks on the device: every sampled output texel equals the normalised source
    texel at corner + offset (<= 1 fp16 ulp), descriptors have unit norm, corners respect the clip."""
    import torch
    from pixsfm_amd.engine import PatchArena
    g = torch.Generator(device="cuda").manual_seed(3)
    fmap = torch.randn((128, 256, 320), generator=g, device="cuda", dtype=torch.float32)
    n = 100_000
    kps = (torch.rand((n, 2), generator=g, device="cuda", dtype=torch.float64) *
           torch.tensor([1300.0, 1040.0], device="cuda", dtype=torch.float64) - 10.0).cpu().numpy()
    arena = PatchArena(ctx, n, 16, 16, 128, np.float16)
    d_kp = ctx.to_device(kps, np.float64)
    arena.extract(0, fmap, d_kp, (1280.0, 1024.0))            # warm-up
    ctx.timer_start()
    arena.extract(0, fmap, d_kp, (1280.0, 1024.0))
    ms = ctx.timer_stop()

    class _Raw:   # view the arena memory as a torch tensor (no copy)
        __cuda_array_interface__ = {"shape": (n, 16, 16, 128), "typestr": "<f2", "data": (arena.data_ptr, False),
                                    "version": 2}
    out = torch.as_tensor(_Raw(), device="cuda")
    _, corners_np, scales = arena.download(0, 0)
    corners_np = np.empty((n, 2), np.int32)
    import ctypes
    from pixsfm_amd._lib import check
    check(ctx.lib.pxr_memcpy_d2h(ctx.handle, corners_np.ctypes.data, ctypes.c_void_p(ctx.lib.pxr_arena_corners(arena.handle)),
                                 corners_np.nbytes), "d2h")
    want_c = np.clip((kps * np.array([0.25, 0.25]) - 8.0).astype(np.int32), [0, 0], [320 - 17, 256 - 17])
    assert np.array_equal(corners_np, want_c)
    corners = torch.from_numpy(corners_np).cuda().long()
    norm = torch.nn.functional.normalize(fmap[None], dim=1)[0].to(torch.float16)
    sel = torch.randint(0, n, (4096,), generator=g, device="cuda")
    oy = torch.randint(0, 16, (4096,), generator=g, device="cuda")
    ox = torch.randint(0, 16, (4096,), generator=g, device="cuda")
    want = norm[:, corners[sel, 1] + oy, corners[sel, 0] + ox].T.contiguous()
    got = out[sel, oy, ox]
    diff = (want.view(torch.int16).int() - got.view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1
    assert float((out[sel].float().norm(dim=-1) - 1).abs().max()) < 2e-3
    gbps = n * 16 * 16 * 128 * (4 + 2) / (ms * 1e-3) / 1e9   # fp32 read + fp16 write per texel
    print("extract: %.3f ms for %d patches, %.0f GB/s algorithmic" % (ms, n, gbps))
    assert ms < 50.0


@pytest.mark.parametrize("ps", [8, 10])
def test_other_patch_sizes_through_producer_and_solvers(ctx, ps):
    """Patch sides other than 16 (the reference's lighter configurations use 10 / 8): the producer against the
    restatement, then KA edges and a BA evaluation on such patches against the oracle."""
    import torch
    import pxo
    import pxo_extract
    from pixsfm_amd import synthetic, synthetic_ka
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    rng = np.random.default_rng(ps)
    fmap = rng.normal(0, 1, (128, 40, 50)).astype(np.float32)
    kps = rng.uniform(-3, 210, (60, 2))
    want, corners, scale = pxo_extract.sparse_patches(fmap, kps, (200.0, 160.0), ps=ps)
    arena = PatchArena(ctx, len(kps), ps, ps, 128, np.float16)
    arena.extract(0, torch.from_numpy(fmap).cuda(), kps, (200.0, 160.0))
    patches, c, s = arena.download()
    assert np.array_equal(c, corners) and patches.shape == (60, ps, ps, 128)
    assert _ulp16(patches, want).max() <= 1
    # KA edges on ps x ps patches
    kprob = synthetic_ka.make_ka_problem(n_tracks=5, track_len=3, seed=ps, patch_size=ps, sigma=0.5)
    karena = PatchArena.from_numpy(ctx, kprob["patches"], kprob["corners"], kprob["scales"])
    cost, r, J1, J2 = KAProblem(ctx, karena, kprob).eval(interp_cfg(), make_loss("cauchy", [0.25]), materialize=True)
    r, J1 = r.download(), J1.download()
    for e in (0, len(kprob["edge_src"]) - 1):
        a_, b_ = kprob["edge_src"][e], kprob["edge_dst"][e]
        p1 = pxo.make_patch(kprob["patches"][a_], kprob["corners"][a_], kprob["scales"][a_])
        p2 = pxo.make_patch(kprob["patches"][b_], kprob["corners"][b_], kprob["scales"][b_])
        ro, J1o, _ = pxo.ka_residual(p1, p2, pxo.cfg(), kprob["kp"][a_], kprob["kp"][b_])
        assert np.abs(r[e] - ro).max() < 1e-12 and np.abs(J1[e] - J1o).max() < 1e-10
    # BA evaluation on ps x ps patches
    bprob = synthetic.make_ba_problem(n_cams=3, n_points=20, obs_per_point=2, seed=ps, patch_size=ps)
    barena = PatchArena.from_numpy(ctx, bprob["patches"], bprob["corners"], bprob["scales"])
    rec, rr, _, _ = BAProblem(ctx, barena, bprob).eval(interp_cfg(), with_jacobian=True, materialize=True)
    rr = rr.download()
    for i in (0, 17, 39):
        p = pxo.make_patch(bprob["patches"][i], bprob["corners"][i], bprob["scales"][i])
        img, pt = bprob["obs_image"][i], bprob["obs_point"][i]
        cam = bprob["image_camera"][img]
        out = pxo.ba_residual(p, pxo.cfg(), int(bprob["cam_model"][cam]), bprob["qvec"][img], bprob["tvec"][img],
                              bprob["xyz"][pt], bprob["cam_params"][cam], bprob["refs"][pt], jac=False)
        assert np.abs(rr[i] - out[0]).max() < 1e-12


def test_map_tile_ordered_extraction_of_many_patches(ctx):
    """>= 4096 keypoints: the workgroups walk the patches in map-tile order (counting sort on the device) -- every
    patch still lands in the slot of its keypoint, bit-identical to the plain gather of the oracle."""
    import torch
    import pxo_extract
    from pixsfm_amd.engine import PatchArena
    rng = np.random.default_rng(17)
    fmap = rng.normal(0, 1, (64, 96, 120)).astype(np.float32)
    n = 6000
    kps = rng.uniform(-8, 500, (n, 2))
    kps[:50] = kps[50:100]                                       # duplicates: several patches per corner
    size = (480.0, 384.0)
    want, corners, scale = pxo_extract.sparse_patches(fmap, kps, size, l2_normalize=False, dtype=np.float16)
    t = torch.from_numpy(fmap).cuda()
    arena = PatchArena(ctx, n, 16, 16, 64, np.float16)
    assert arena.extract(0, t, kps, size, l2_normalize=False) == n
    patches, c, s = arena.download()
    assert np.array_equal(c, corners) and np.array_equal(patches, want)
    # the small-call path (no reordering) writes the same bytes
    small = PatchArena(ctx, 1000, 16, 16, 64, np.float16)
    small.extract(0, t, kps[:1000], size, l2_normalize=False)
    assert np.array_equal(small.download()[0], patches[:1000])
    # with normalisation: the same values whatever the order
    arena.extract(0, t, kps, size, l2_normalize=True)
    small.extract(0, t, kps[:1000], size, l2_normalize=True)
    assert np.array_equal(small.download()[0], arena.download(0, 1000)[0])
    arena.close(); small.close()
