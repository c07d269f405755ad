"""Synthetic featuremetric problems, numpy flavour.

The generator is modelled on the reference's only problem generator
(pixsfm/bundle_adjustment/src/bundle_optimizer_test.cc:61-134: SIMPLE_RADIAL f=1200 on
1000^2 images, cameras around the origin) and on how the extractor crops sparse patches
(pixsfm/features/extractor.py:179-236: corner = int(kp*scale - ps/2), per-texel
L2-normalised, cast to fp16).  Feature content is a smooth per-track field
F_t(dx,dy) = normalize(A_t . phi(dx,dy)) rendered around the TRUE projection, so that the
featuremetric optimum coincides with the true geometry.
"""
import numpy as np

KPAD = 12
N_BASIS = 16


def _basis(seed=1234):
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, N_BASIS)
    mag = rng.uniform(0.08, 0.45, N_BASIS)     # rad / px: smooth across a 16 px patch
    theta = rng.uniform(0, 2 * np.pi, N_BASIS)
    return mag * np.cos(ang), mag * np.sin(ang), theta


def field(A, dx, dy):
    """A: (..., C, N_BASIS); dx, dy: broadcastable offsets from the true location (px).
    Returns the un-normalised field value (..., C)."""
    wx, wy, th = _basis()
    ph = np.cos(dx[..., None] * wx + dy[..., None] * wy + th)   # (..., NB)
    return np.einsum("...k,...ck->...c", ph, A)


def render_patches(A_per_patch, centers, corners, scales, patch_size, dtype=np.float16, noise=0.0, rng=None,
                   chunk=1024):
    """Render (n, ps, ps, C) patches.  centers: true image-space location of the track in each
    patch; texel (i, j) of a patch sits at image coords ((x0 + i + .5)/sx, (y0 + j + .5)/sy)
    (FeaturePatch::ToImageCoordinates, pixsfm/features/src/featurepatch.h:257-262)."""
    n, C, _ = A_per_patch.shape
    ps = patch_size
    out = np.empty((n, ps, ps, C), dtype=dtype)
    wx, wy, th = _basis()
    ii = np.arange(ps)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        xs = (corners[s:e, 0:1] + ii[None, :] + 0.5) / scales[s:e, 0:1] - centers[s:e, 0:1]   # (m, ps) image px
        ys = (corners[s:e, 1:2] + ii[None, :] + 0.5) / scales[s:e, 1:2] - centers[s:e, 1:2]
        ph = np.cos(ys[:, :, None, None] * wy + xs[:, None, :, None] * wx + th)                # (m, H, W, NB)
        val = np.einsum("mhwk,mck->mhwc", ph, A_per_patch[s:e])
        if noise > 0:
            val = val + rng.normal(0, noise, val.shape) * np.linalg.norm(val, axis=-1, keepdims=True) / np.sqrt(C)
        val /= np.linalg.norm(val, axis=-1, keepdims=True)
        out[s:e] = val.astype(dtype)
    return out


def rotmat_to_qvec(R):
    """COLMAP RotationMatrixToQuaternion (w-first), via the symmetric-matrix eigenvector."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = R.flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
                  [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    w, V = np.linalg.eigh(K)
    q = V[[3, 0, 1, 2], np.argmax(w)]
    return q * (-1 if q[0] < 0 else 1)


def qvec_to_rotmat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def project(model, k, q, t, X):
    """numpy WorldToPixel for the generator (SIMPLE_RADIAL / PINHOLE family only)."""
    p = qvec_to_rotmat(q) @ X + t
    u, v = p[0] / p[2], p[1] / p[2]
    r2 = u * u + v * v
    if model == 2:
        rad = k[3] * r2
        return np.array([k[0] * u * (1 + rad) + k[1], k[0] * v * (1 + rad) + k[2]])
    if model == 0:
        return np.array([k[0] * u + k[1], k[0] * v + k[2]])
    if model == 1:
        return np.array([k[0] * u + k[2], k[1] * v + k[3]])
    if model == 3:
        rad = k[3] * r2 + k[4] * r2 * r2
        return np.array([k[0] * u * (1 + rad) + k[1], k[0] * v * (1 + rad) + k[2]])
    if model == 4:
        rad = k[4] * r2 + k[5] * r2 * r2
        du = u * rad + 2 * k[6] * u * v + k[7] * (r2 + 2 * u * u)
        dv = v * rad + 2 * k[7] * u * v + k[6] * (r2 + 2 * v * v)
        return np.array([k[0] * (u + du) + k[2], k[1] * (v + dv) + k[3]])
    raise ValueError(model)


def ring_cameras(n_cams, radius=10.0, rng=None, jitter=0.3):
    """Cameras on a circle looking at the origin; returns world-to-camera (qvec, tvec)."""
    qs, ts = [], []
    for i in range(n_cams):
        th = 2 * np.pi * i / n_cams
        c = np.array([radius * np.cos(th), 0.0, radius * np.sin(th)])
        if rng is not None:
            c = c + rng.normal(0, jitter, 3)
        z = -c / np.linalg.norm(c)
        up = np.array([0.0, 1.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        qs.append(rotmat_to_qvec(R))
        ts.append(-R @ c)
    return np.array(qs), np.array(ts)


def make_ba_problem(n_cams=8, n_points=200, obs_per_point=4, channels=128, patch_size=16, seed=2,
                    dtype=np.float16, model=2, scale=(1.0, 1.0), noise=0.0, perturb=True,
                    rot_deg=0.2, trans=0.01, pt_sigma=0.01, shared_camera=False):
    """A small synthetic featuremetric BA instance (config 3 of BASELINE.json at reduced size).

    Returns a dict with the flat arrays pxr_ba_view expects (initial = perturbed parameters),
    the patch arena (numpy), reference descriptors, and the ground truth under 'gt_*'.
    """
    rng = np.random.default_rng(seed)
    q_gt, t_gt = ring_cameras(n_cams, rng=rng)
    base = {0: [1200.0, 500, 500], 1: [1200.0, 1180.0, 500, 500], 2: [1200.0, 500, 500, 0.02],
            3: [1200.0, 500, 500, 0.02, -0.01], 4: [1200.0, 1180.0, 500, 500, 0.02, -0.01, 1e-3, -5e-4]}[model]
    n_phys_cams = 1 if shared_camera else n_cams
    cam_params = np.zeros((n_phys_cams, KPAD))
    cam_params[:, :len(base)] = base
    cam_model = np.full(n_phys_cams, model, dtype=np.int32)
    image_camera = np.zeros(n_cams, dtype=np.int32) if shared_camera else np.arange(n_cams, dtype=np.int32)
    X_gt = rng.uniform(-1, 1, (n_points, 3))
    obs_image = np.empty(n_points * obs_per_point, dtype=np.int32)
    for p in range(n_points):
        obs_image[p * obs_per_point:(p + 1) * obs_per_point] = rng.choice(n_cams, obs_per_point, replace=False)
    obs_point = np.repeat(np.arange(n_points, dtype=np.int32), obs_per_point)
    n_obs = len(obs_image)
    obs_patch = np.arange(n_obs, dtype=np.int64)
    centers = np.empty((n_obs, 2))
    for i in range(n_obs):
        im = obs_image[i]
        centers[i] = project(model, cam_params[image_camera[im]], q_gt[im], t_gt[im], X_gt[obs_point[i]])
    scales = np.tile(np.asarray(scale, dtype=np.float64), (n_obs, 1))
    # extractor.py:192-193: corner = int(kp*scale - ps/2)
    corners = np.floor(centers * scales - patch_size / 2.0).astype(np.int32)
    A = rng.normal(0, 1, (n_points, channels, N_BASIS))
    patches = render_patches(A[obs_point], centers, corners, scales, patch_size, dtype, noise, rng)
    refs = field(A, np.zeros(n_points), np.zeros(n_points))
    refs /= np.linalg.norm(refs, axis=-1, keepdims=True)
    qvec, tvec, xyz = q_gt.copy(), t_gt.copy(), X_gt.copy()
    if perturb:
        for i in range(n_cams):
            ax = rng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
            ang = np.deg2rad(rot_deg) * rng.uniform(0.5, 1.0)
            dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
            w0, v0 = dq[0], dq[1:]
            w1, v1 = qvec[i, 0], qvec[i, 1:]
            qvec[i] = np.concatenate([[w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)])
            tvec[i] += rng.normal(0, trans, 3)
        xyz += rng.normal(0, pt_sigma, xyz.shape)
    return dict(obs_image=obs_image, obs_point=obs_point, obs_patch=obs_patch, image_camera=image_camera,
                qvec=qvec, tvec=tvec, cam_model=cam_model, cam_params=cam_params, xyz=xyz, refs=refs,
                patches=patches, corners=corners, scales=scales,
                gt_qvec=q_gt, gt_tvec=t_gt, gt_xyz=X_gt, centers=centers)
