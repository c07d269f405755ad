"""A featuremetric KA / BA scene made of REAL image content (test helper; numpy + scipy only).

tests/golden/real_image_tiles.npz holds 384 x 384 RGB crops of the reference's own demo photographs (sacre_coeur, BASELINE
configs[0]; generator tests/golden/make_golden_real_images.py).  A crop is laid on the plane z = 0 and photographed by
`n_views` pin-hole cameras (known poses -> known homographies): every view is a cubic resampling of the photograph, so its
content is real texture -- edges, fine detail, sensor noise, JPEG blocking -- not the smooth cosine fields of
pixsfm_amd.synthetic.  3D points sit at the strongest-gradient locations of the photograph; their projections are the
keypoints.  Feature maps per view:
  "image"  the reference's weight-free `image` model (pixsfm/features/models/image.py:8-34): RGB / 255, 3 channels, float32;
  "conv"   a fixed-seed bank of 128 3 x 3 filters (zero-mean, reflect padding) over the 2 x 2-pooled RGB view -- a stand-in for
           the CNN with the same data path: dense C x h/2 x w/2 map (scale 0.5) -> per-texel L2 normalisation -> fp16 ->
           16 x 16 patches around the keypoints (pixsfm/features/extractor.py:152-199, extract_patches.py:13-44).
"""
import os

import numpy as np

from pixsfm_amd import synthetic

TILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_image_tiles.npz")
VIEW = 320            # views are VIEW x VIEW pixels
FOCAL = 420.0


def load_tile(index=0):
    return np.load(TILES)["tiles"][index]                       # (384, 384, 3) uint8


def _look_at(c):
    z = -c / np.linalg.norm(c)
    up = np.array([0.0, 1.0, 0.0])
    x = np.cross(up, z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    return R, -R @ c


def cameras(n_views, seed):
    """World-to-camera (R, t) of cameras ~2.6 plane-units in front of the plane z = 0, within ~14 degrees of its normal."""
    rng = np.random.default_rng(seed)
    Rs, ts = [], []
    for k in range(n_views):
        ang = 2 * np.pi * k / n_views
        c = np.array([0.65 * np.cos(ang), 0.65 * np.sin(ang), -2.6]) + rng.normal(0, 0.05, 3)
        R, t = _look_at(c)
        Rs.append(R); ts.append(t)
    return np.array(Rs), np.array(ts)


def render_view(tile, R, t):
    """The photograph on the plane z = 0 (plane coordinates X, Y in [-1, 1] <-> pixel centres of the crop) seen by camera
    (R, t): per view pixel the ray's intersection with the plane, then cubic resampling (scipy.ndimage.map_coordinates)."""
    from scipy.ndimage import map_coordinates
    n = tile.shape[0]
    v, u = np.meshgrid(np.arange(VIEW) + 0.5, np.arange(VIEW) + 0.5, indexing="ij")     # COLMAP pixel centres
    d_cam = np.stack([(u - VIEW / 2) / FOCAL, (v - VIEW / 2) / FOCAL, np.ones_like(u)], -1)
    d = d_cam @ R                                              # world direction = R^T d_cam
    c = -R.T @ t
    s = -c[2] / d[..., 2]
    X = c[0] + s * d[..., 0]
    Y = c[1] + s * d[..., 1]
    px, py = (X + 1) * 0.5 * n - 0.5, (Y + 1) * 0.5 * n - 0.5  # plane -> crop texel indices
    out = np.stack([map_coordinates(tile[..., ch].astype(np.float32), [py, px], order=3, mode="nearest") for ch in range(3)])
    return np.clip(out, 0, 255).astype(np.float32)             # (3, VIEW, VIEW)


def conv_bank(rgb01, channels=128, seed=11):
    """(3, h, w) float32 in [0, 1] -> (channels, h, w) float32: 3 x 3 zero-mean filters, reflect padding."""
    rng = np.random.default_rng(seed)
    Wt = rng.normal(0, 1, (channels, 3, 3, 3)).astype(np.float32)
    Wt -= Wt.mean(axis=(1, 2, 3), keepdims=True)               # no response to flat colour
    pad = np.pad(rgb01, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    h, w = rgb01.shape[1:]
    out = np.zeros((channels, h, w), np.float32)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("ck,khw->chw", Wt[:, :, dy, dx], pad[:, dy:dy + h, dx:dx + w], optimize=True)
    return out


def strong_gradient_points(tile, n_points, cell=14, margin=40):
    """One location per cell x cell block of the photograph: its strongest-gradient texel; the n_points strongest of those.
    Returns plane coordinates (n, 2) in [-1, 1]."""
    g = tile.astype(np.float32).mean(axis=2)
    gy, gx = np.gradient(g)
    e = gx * gx + gy * gy
    n = g.shape[0]
    cand = []
    for y0 in range(margin, n - margin - cell, cell):
        for x0 in range(margin, n - margin - cell, cell):
            blk = e[y0:y0 + cell, x0:x0 + cell]
            k = int(np.argmax(blk))
            cand.append((float(blk.reshape(-1)[k]), y0 + k // cell, x0 + k % cell))
    cand.sort(reverse=True)
    sel = np.array([(x, y) for _, y, x in cand[:n_points]], dtype=np.float64)
    return (sel + 0.5) / n * 2.0 - 1.0


def make_scene(n_views=6, n_points=160, tile_index=0, seed=3, kind="conv", kp_sigma=0.7, rot_deg=0.15, trans=0.004,
               pt_sigma=0.004):
    """Returns a dict: per-view dense feature maps (float32, C x VIEW x VIEW), the BA problem arrays (perturbed initial
    parameters, ground truth under gt_*), per-observation true / detected keypoints, and the KA graph (complete graph per
    track, one sub-problem per group of tracks)."""
    from pixsfm_amd.ka_engine import pack_tracks_into_problems
    rng = np.random.default_rng(seed)
    tile = load_tile(tile_index)
    Rs, ts = cameras(n_views, seed)
    views = [render_view(tile, Rs[k], ts[k]) for k in range(n_views)]
    if kind == "image":
        fmaps = [v / np.float32(255.0) for v in views]                      # models/image.py:29-32
    else:
        # the bank runs on the 2 x 2 average-pooled view: a feature map at HALF the image resolution (scale 0.5 between image
        # and map coordinates, like a CNN with stride 2), receptive field 6 x 6 pixels
        fmaps = [conv_bank((v / np.float32(255.0)).reshape(3, VIEW // 2, 2, VIEW // 2, 2).mean(axis=(2, 4))) for v in views]
    plane = strong_gradient_points(tile, n_points)
    X_gt = np.concatenate([plane, np.zeros((len(plane), 1))], 1)
    q_gt = np.array([synthetic.rotmat_to_qvec(R) for R in Rs])
    k = np.array([FOCAL, VIEW / 2, VIEW / 2, 0.0])
    obs_image, obs_point, centers = [], [], []
    for p, X in enumerate(X_gt):
        for v in range(n_views):
            xy = synthetic.project(2, k, q_gt[v], ts[v], X)
            if 26 <= xy[0] <= VIEW - 26 and 26 <= xy[1] <= VIEW - 26:
                obs_image.append(v); obs_point.append(p); centers.append(xy)
    obs_image, obs_point, centers = np.array(obs_image, np.int32), np.array(obs_point, np.int32), np.array(centers)
    # keep points seen at least twice, renumber
    cnt = np.bincount(obs_point, minlength=len(X_gt))
    keep_pt = cnt >= 2
    renum = np.cumsum(keep_pt) - 1
    sel = keep_pt[obs_point]
    obs_image, obs_point, centers = obs_image[sel], renum[obs_point[sel]].astype(np.int32), centers[sel]
    X_gt = X_gt[keep_pt]
    n_obs, n_pts = len(obs_image), len(X_gt)
    detected = centers + rng.normal(0, kp_sigma, centers.shape)              # what a detector hands to KA
    cam_params = np.zeros((n_views, synthetic.KPAD)); cam_params[:, :4] = k
    qvec, tvec = q_gt.copy(), ts.copy()
    for i in range(n_views):
        ax = rng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
        ang = np.deg2rad(rot_deg) * rng.uniform(0.5, 1.0)
        w0, v0 = np.cos(ang / 2), np.sin(ang / 2) * ax
        w1, v1 = qvec[i, 0], qvec[i, 1:]
        qvec[i] = np.concatenate([[w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)])
        tvec[i] += rng.normal(0, trans, 3)
    xyz = X_gt + rng.normal(0, pt_sigma, X_gt.shape)
    # KA graph: complete graph inside a track, root = highest summed similarity
    edge_src, edge_dst = [], []
    for p in range(n_pts):
        ids = np.nonzero(obs_point == p)[0]
        for a in range(len(ids)):
            for b in range(a + 1, len(ids)):
                edge_src.append(ids[a]); edge_dst.append(ids[b])
    edge_src, edge_dst = np.array(edge_src, np.int32), np.array(edge_dst, np.int32)
    edge_w = rng.uniform(0.5, 1.0, len(edge_src))
    score = np.zeros(n_obs); np.add.at(score, edge_src, edge_w); np.add.at(score, edge_dst, edge_w)
    node_const = np.zeros(n_obs, np.uint8)
    for p in range(n_pts):
        ids = np.nonzero(obs_point == p)[0]
        node_const[ids[np.argmax(score[ids])]] = 1
    labels, bins = pack_tracks_into_problems(obs_point, 50)
    return dict(fmaps=fmaps, image_size=(float(VIEW), float(VIEW)), obs_image=obs_image, obs_point=obs_point,
                obs_patch=np.arange(n_obs, dtype=np.int64), image_camera=np.arange(n_views, dtype=np.int32),
                qvec=qvec, tvec=tvec, cam_model=np.full(n_views, 2, np.int32), cam_params=cam_params, xyz=xyz,
                gt_qvec=q_gt, gt_tvec=ts, gt_xyz=X_gt, centers=centers, detected=detected,
                edge_src=edge_src, edge_dst=edge_dst, edge_w=edge_w, node_const=node_const,
                node_problem=np.array(labels, np.int32), n_problems=len(bins), channels=fmaps[0].shape[0])
