"""Randomised parity sweep, HIP kernels against the oracle (NOT part of the test suite; a bug hunt):
pxr_ba_eval / pxr_ka_eval / pxr_interpolate on random patch shapes, storage types, channel counts, scales, corners, camera
models 0-10, interpolation switches, keypoints inside / on the border / outside, points in front of and behind the camera.
python tests/fuzz/fuzz_eval_vs_oracle.py [n_trials] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pixel-perfect-sfm_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import pxo
from pixsfm_amd import engine
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss
from pixsfm_amd.ka_engine import KAProblem

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = Context(0)
KN = [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12]


def rel(a, b):
    m = np.isfinite(b)
    if not m.any():
        return 0.0
    if not np.array_equal(np.isfinite(a), m):
        return np.inf
    return float(np.abs(a[m] - b[m]).max() / max(1e-300, np.abs(b[m]).max()))


def cam_params(model, f, cx, cy):
    K = KN[model]
    p = np.zeros(12)
    if model in (1, 4, 5, 6, 7, 10):
        p[:4] = [f, f * 1.03, cx, cy]; e = 4
    else:
        p[:3] = [f, cx, cy]; e = 3
    extra = K - e
    if model == 7:
        p[4] = 0.5        # FOV omega
    elif extra:
        p[e:K] = rng.normal(0, 0.01, extra)
    return p


worst = dict(ba=0.0, ka=0.0, interp=0.0)
bad = 0
for trial in range(n_trials):
    dt = [np.float16, np.float32, np.float64][rng.integers(3)]
    C = int(rng.choice([128, 64]))
    H, W = int(rng.integers(4, 21)), int(rng.integers(4, 21))
    m = int(rng.integers(3, 40))
    l2, fs, cb = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
    patches = rng.normal(0, 1, (m, H, W, C)).astype(dt)
    scales = rng.uniform(0.2, 1.5, (m, 2)) if rng.integers(2) else np.ones((m, 2))
    model = int(rng.integers(0, 11))
    f, cx, cy = rng.uniform(300, 1500), rng.uniform(200, 900), rng.uniform(200, 900)
    params = np.stack([cam_params(model, f, cx, cy) for _ in range(m)])
    # one observation per image / camera / point
    uv = np.stack([rng.uniform(-1.5, W + 1.5, m), rng.uniform(-1.5, H + 1.5, m)], 1)     # local coords, some outside
    q = rng.normal(0, 1, (m, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = rng.normal(0, 1, (m, 3))
    z = rng.uniform(2, 8, m)
    if rng.integers(4) == 0:
        z[rng.integers(m)] *= -1                      # a point behind its camera
    corners = np.zeros((m, 2), np.int32); tvec = np.zeros((m, 3))
    for i in range(m):
        # choose the corner so that the projection of X lands at uv: project first with t putting X at depth z on a ray near the axis
        ray = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 1.0]) * z[i]
        w_, x_, y_, z_ = q[i]
        R = np.array([[1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - w_ * z_), 2 * (x_ * z_ + w_ * y_)],
                      [2 * (x_ * y_ + w_ * z_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - w_ * x_)],
                      [2 * (x_ * z_ - w_ * y_), 2 * (y_ * z_ + w_ * x_), 1 - 2 * (x_ * x_ + y_ * y_)]])
        tvec[i] = ray - R @ X[i]
        xy, _, _, _, _ = pxo.world_to_pixel(model, params[i, :KN[model]], q[i], tvec[i], X[i], jac=False)
        if not np.all(np.isfinite(xy)):
            xy = np.array([cx, cy])
        corners[i] = np.floor(xy * scales[i] - 0.5 - uv[i]).astype(np.int32)
    refs = rng.normal(0, 1, (m, C)); refs /= np.linalg.norm(refs, axis=1, keepdims=True)
    ids = np.arange(m, dtype=np.int32)
    prob = dict(obs_image=ids, obs_point=ids, obs_patch=np.arange(m, dtype=np.int64), image_camera=ids, qvec=q, tvec=tvec,
                cam_model=np.full(m, model, np.int32), cam_params=params, xyz=X, refs=refs, patches=patches, corners=corners, scales=scales)
    arena = PatchArena.from_numpy(ctx, patches, corners, scales)
    cfg = interp_cfg(l2_normalize=l2, use_float_simd=fs, check_bounds=cb)
    ocfg = pxo.cfg(l2, fs, cb)
    tol = 2e-6 if fs else 1e-9
    # ---- BA residual + full Jacobian ----
    ba = BAProblem(ctx, arena, prob)
    rec, r, gx, gy = ba.eval(cfg, with_jacobian=True, materialize=True)
    P = ba.projection_jacobian().download()
    r, gx, gy = r.download(), gx.download(), gy.download()
    J = gx[:, :, None] * P[:, None, 0, :] + gy[:, :, None] * P[:, None, 1, :]
    cost_o, r_o, J_o = pxo.ba_eval_batch(prob, ocfg, pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    e = max(rel(r, r_o), rel(J, J_o))
    cost = ba.cost(make_loss("cauchy", [0.25]))
    if np.isfinite(cost_o) != np.isfinite(cost) or (np.isfinite(cost_o) and abs(cost - cost_o) > tol * abs(cost_o)):
        e = np.inf
    worst["ba"] = max(worst["ba"], e if np.isfinite(e) else 0.0)
    if not e < tol:
        bad += 1
        print("BA trial %d: err %g  dtype %s C %d %dx%d model %d l2 %d fs %d cb %d cost %r vs %r" % (trial, e, np.dtype(dt).name, C, H, W, model, l2, fs, cb, cost, cost_o))
    # ---- keypoint-space interpolation + KA edges on the same patches ----
    kp = (uv + corners + 0.5) / scales
    desc, Jk = engine.interpolate(ctx, arena, cfg, kp, np.arange(m), jacobian=True)
    e = 0.0
    for i in range(m):
        of, ogx, ogy, _ = pxo.patch_eval(pxo.make_patch(patches[i], corners[i], scales[i]), kp[i], ocfg)
        e = max(e, rel(desc[i], of), rel(Jk[i, :, 0], ogx), rel(Jk[i, :, 1], ogy))
    worst["interp"] = max(worst["interp"], e if np.isfinite(e) else 0.0)
    if not e < tol:
        bad += 1
        print("interp trial %d: err %g  dtype %s C %d %dx%d l2 %d fs %d" % (trial, e, np.dtype(dt).name, C, H, W, l2, fs))
    if C in (128, 64) and m >= 4:
        me = m // 2
        kprob = dict(kp=kp, node_patch=np.arange(m, dtype=np.int64), node_const=np.zeros(m, np.uint8), node_problem=np.zeros(m, np.int32),
                     edge_src=np.arange(0, 2 * me, 2, dtype=np.int32), edge_dst=np.arange(1, 2 * me, 2, dtype=np.int32),
                     edge_w=rng.uniform(0.2, 1.0, me), patches=patches, corners=corners, scales=scales, n_problems=1)
        ka = KAProblem(ctx, arena, kprob)
        kc, kr, J1, J2 = ka.eval(cfg, make_loss("cauchy", [0.25]), materialize=True)
        kr, J1, J2, kc = kr.download(), J1.download(), J2.download(), kc.download()
        e = 0.0
        ls = pxo.loss("cauchy", 0.25)
        for j in range(me):
            a_, b_ = 2 * j, 2 * j + 1
            orr, oJ1, oJ2 = pxo.ka_residual(pxo.make_patch(patches[a_], corners[a_], scales[a_]), pxo.make_patch(patches[b_], corners[b_], scales[b_]), ocfg, kp[a_], kp[b_])
            oc = 0.5 * pxo.loss_eval(ls, float(orr @ orr), kprob["edge_w"][j])[0]
            e = max(e, rel(kr[j], orr), rel(J1[j], oJ1), rel(J2[j], oJ2), abs(kc[j] - oc) / max(1e-300, abs(oc)))
        worst["ka"] = max(worst["ka"], e if np.isfinite(e) else 0.0)
        if not e < tol:
            bad += 1
            print("KA trial %d: err %g  dtype %s C %d %dx%d l2 %d fs %d" % (trial, e, np.dtype(dt).name, C, H, W, l2, fs))
    arena.close()
print("trials %d  mismatches %d  worst finite relative errors %s" % (n_trials, bad, worst))
