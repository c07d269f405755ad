"""GPU-side generator of the large synthetic BA instance (BASELINE.json config 3:
200 cams / 200k points / 1M observations, 65.5 GB of fp16 16x16x128 patches).

Same construction as synthetic.make_ba_problem (per-track smooth field rendered around the
true projection) but rendered with torch on the device in chunks, because 65 GB cannot be
staged through the host.  torch is used here only to fabricate bench input.
"""
import numpy as np
import torch

from . import synthetic

KPAD = synthetic.KPAD
TEXTURE_BLOCK = 25_000     # points per independently seeded block of texture coefficients


def make_ba_problem_gpu(device, n_cams=200, n_points=200_000, obs_per_point=5, channels=128, patch_size=16,
                        seed=2, point_range=None, chunk=16384, rot_deg=0.2, trans=0.01, pt_sigma=0.01):
    """Returns (problem dict of numpy arrays WITHOUT 'patches', patches torch.half tensor on device).

    point_range=(lo, hi): render only the observations of points [lo, hi) (multi-GPU shards);
    the camera set and all random draws are identical on every rank.
    """
    rng = np.random.default_rng(seed)
    q_gt, t_gt = synthetic.ring_cameras(n_cams, rng=rng)
    cam_params = np.zeros((n_cams, KPAD))
    cam_params[:, :4] = [1200.0, 500.0, 500.0, 0.0]          # bundle_optimizer_test.cc:81-91 (k = 0)
    cam_model = np.full(n_cams, 2, dtype=np.int32)
    image_camera = np.arange(n_cams, dtype=np.int32)
    X_gt = rng.uniform(-1, 1, (n_points, 3))
    # obs_per_point distinct cameras per point (vectorised: random keys, take the smallest k)
    # (drawn in row blocks -- the same random stream as one (n_points, n_cams) draw, without its memory: 16 GB of keys
    # at 4000 cameras x 1M points)
    obs_image_all = np.empty((n_points, obs_per_point), dtype=np.int32)
    rows = max(1, (1 << 27) // max(1, n_cams))
    for r0 in range(0, n_points, rows):
        keys = rng.random((min(rows, n_points - r0), n_cams), dtype=np.float32)
        obs_image_all[r0:r0 + len(keys)] = np.argpartition(keys, obs_per_point, axis=1)[:, :obs_per_point]
    del keys
    # perturbed initial parameters (identical on all ranks)
    qvec, tvec = q_gt.copy(), t_gt.copy()
    for i in range(n_cams):
        ax = rng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
        ang = np.deg2rad(rot_deg) * rng.uniform(0.5, 1.0)
        w0, v0 = np.cos(ang / 2), np.sin(ang / 2) * ax
        w1, v1 = qvec[i, 0], qvec[i, 1:]
        qvec[i] = np.concatenate([[w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)])
        tvec[i] += rng.normal(0, trans, 3)
    xyz = X_gt + rng.normal(0, pt_sigma, X_gt.shape)

    lo, hi = (0, n_points) if point_range is None else point_range
    n_loc = hi - lo
    obs_image = obs_image_all[lo:hi].reshape(-1)
    obs_point = np.repeat(np.arange(n_loc, dtype=np.int32), obs_per_point)   # local point index
    n_obs = len(obs_image)

    dev = torch.device(device)
    f64 = torch.float64
    Rs = torch.tensor(np.stack([synthetic.qvec_to_rotmat(q) for q in q_gt]), dtype=f64, device=dev)
    ts = torch.tensor(t_gt, dtype=f64, device=dev)
    Xg = torch.tensor(X_gt[lo:hi], dtype=f64, device=dev)
    oi = torch.tensor(obs_image.astype(np.int64), device=dev)
    op = torch.tensor(obs_point.astype(np.int64), device=dev)
    p = torch.einsum("nij,nj->ni", Rs[oi], Xg[op]) + ts[oi]
    centers = 1200.0 * p[:, :2] / p[:, 2:3] + 500.0
    corners = torch.floor(centers - patch_size / 2.0).to(torch.int32)

    wx, wy, th = (torch.tensor(a, dtype=torch.float32, device=dev) for a in synthetic._basis())
    # texture coefficients: drawn per FIXED block of TEXTURE_BLOCK points (seeded by the block's index, always at the full
    # block shape), so that a point's texture does not depend on the shard it falls into -- the scene of a run on N ranks is
    # bit for bit the scene of the one-rank run (round 3 seeded by the shard's offset: initial costs 32273.44 vs 32265.32)
    A = torch.empty((n_loc, channels, synthetic.N_BASIS), device=dev, dtype=torch.float16)
    g = torch.Generator(device=dev)
    for blk in range(lo // TEXTURE_BLOCK, (max(hi, lo + 1) - 1) // TEXTURE_BLOCK + 1):
        g.manual_seed(seed * 7919 + blk)
        b0 = blk * TEXTURE_BLOCK
        full = torch.randn((TEXTURE_BLOCK, channels, synthetic.N_BASIS), generator=g, device=dev, dtype=torch.float32)
        s0, s1 = max(lo, b0), min(hi, b0 + TEXTURE_BLOCK)
        if s1 > s0:
            A[s0 - lo:s1 - lo] = full[s0 - b0:s1 - b0].half()
    del full
    patches = torch.empty((n_obs, patch_size, patch_size, channels), dtype=torch.float16, device=dev)
    ii = torch.arange(patch_size, device=dev, dtype=torch.float32)
    for s in range(0, n_obs, chunk):
        e = min(n_obs, s + chunk)
        cx = (corners[s:e, 0:1].float() + ii[None, :] + 0.5) - centers[s:e, 0:1].float()    # (m, W)
        cy = (corners[s:e, 1:2].float() + ii[None, :] + 0.5) - centers[s:e, 1:2].float()    # (m, H)
        ph = torch.cos(cy[:, :, None, None] * wy + cx[:, None, :, None] * wx + th)          # (m, H, W, NB)
        val = torch.bmm(ph.reshape(e - s, patch_size * patch_size, -1),
                        A[op[s:e]].float().transpose(1, 2))                                   # (m, HW, C)
        val = val / val.norm(dim=-1, keepdim=True)
        patches[s:e] = val.reshape(e - s, patch_size, patch_size, channels).half()
    ph0 = torch.cos(th)[None, :]
    refs = (A.float() * ph0[0]).sum(-1).double()     # (not einsum: its batched-GEMM path fails at n_loc * channels = 1.28e8 rows)
    refs = refs / refs.norm(dim=-1, keepdim=True)
    torch.cuda.synchronize(dev)
    prob = dict(obs_image=obs_image, obs_point=obs_point, obs_patch=np.arange(n_obs, dtype=np.int64),
                image_camera=image_camera, qvec=qvec, tvec=tvec, cam_model=cam_model, cam_params=cam_params,
                xyz=xyz[lo:hi], refs=refs.cpu().numpy(), corners=corners.cpu().numpy(),
                scales=np.ones((n_obs, 2)), gt_qvec=q_gt, gt_tvec=t_gt, gt_xyz=X_gt[lo:hi],
                centers=centers.cpu().numpy())
    return prob, patches
