"""TEST INFRASTRUCTURE: a bundle-adjustment problem of this repository's flat layout as a scipy.optimize.least_squares problem.

scipy's robust losses act on SCALAR residuals, rho(f_i^2); Ceres robustifies a whole residual block, rho(|r_b|^2).  The two
objectives coincide when scipy is handed one scalar per block, f_b = |r_b| (`block_norms`): then
    scipy:  cost = 1/2 sum_b C^2 rho0(f_b^2 / C^2),  C = f_scale   ==   Ceres: 1/2 sum_b rho(|r_b|^2)
for cauchy (rho0 = log1p), huber and soft_l1 = Ceres' CauchyLoss(a) / HuberLoss(a) / SoftLOneLoss(a) with a = f_scale.
Gauss-Newton on block norms converges slowly (one row per block), so the optimum is SEARCHED with the full 128-row blocks
robustified by hand (`robustified`, r~ = sqrt(rho(s) / s) r, loss 'linear': the same objective) and then CONFIRMED with scipy's own
loss on the block norms started from there (it must stay put and report the same cost).

Residuals and Jacobians come from the oracle's per-block evaluation (pxo.ba_eval_batch); the quaternion is optimised through a
3-vector in the tangent plane at the initial q (q = normalize(q0 + B d), B an orthonormal basis of q0's complement), which is a
valid chart for any local minimiser and independent of Ceres' manifold.  Constant blocks are left out of the unknowns."""
import numpy as np

import pxo


def _basis(q):
    q = q / np.linalg.norm(q)
    u, _, _ = np.linalg.svd(np.eye(4) - np.outer(q, q))
    return u[:, :3]


class ScipyBA:
    def __init__(self, prob, gauge, cfg=None):
        self.prob, self.cfg = prob, cfg or pxo.cfg()
        pose_const, tmask, cmask, pt_const = gauge
        n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
        self.K = [pxo.lib().pxo_camera_num_params(int(m)) for m in prob["cam_model"]]
        self.q0 = np.array([q / np.linalg.norm(q) for q in prob["qvec"]])
        self.B = np.array([_basis(q) for q in self.q0])
        col, n = {}, 0
        for i in range(n_img):
            if pose_const[i]:
                continue
            for a in range(3):
                col["q", i, a] = n; n += 1
            for a in range(3):
                if not (int(tmask[i]) >> a) & 1:
                    col["t", i, a] = n; n += 1
        for c in range(n_cam):
            for a in range(self.K[c]):
                if not (int(cmask[c]) >> a) & 1:
                    col["k", c, a] = n; n += 1
        for p in range(n_pt):
            if not pt_const[p]:
                for a in range(3):
                    col["X", p, a] = n; n += 1
        self.col, self.n = col, n
        oi, op = prob["obs_image"], prob["obs_point"]
        oc = prob["image_camera"][oi]
        idx = np.full((len(oi), 21), -1, np.int64)              # local columns: dq3 | t3 | X3 | k12
        for b in range(len(oi)):
            for a in range(3):
                idx[b, a] = col.get(("q", oi[b], a), -1)
                idx[b, 3 + a] = col.get(("t", oi[b], a), -1)
                idx[b, 6 + a] = col.get(("X", op[b], a), -1)
            for a in range(self.K[oc[b]]):
                idx[b, 9 + a] = col.get(("k", oc[b], a), -1)
        self.idx = idx

    def x0(self):
        x = np.zeros(self.n)
        for (kind, i, a), j in self.col.items():
            if kind == "t":
                x[j] = self.prob["tvec"][i, a]
            elif kind == "k":
                x[j] = self.prob["cam_params"][i, a]
            elif kind == "X":
                x[j] = self.prob["xyz"][i, a]
        return x

    def pack(self, qvec, tvec, cam_params, xyz):
        """unknown vector of given parameters (quaternions are mapped into the chart at q0; q and -q are the same rotation)"""
        x = np.zeros(self.n)
        for (kind, i, a), j in self.col.items():
            if kind == "q":
                q = qvec[i] / np.linalg.norm(qvec[i])
                q = q if q @ self.q0[i] > 0 else -q
                v = q / (q @ self.q0[i])                         # q0 + B d, up to scale
                x[j] = self.B[i][:, a] @ (v - self.q0[i])
            elif kind == "t":
                x[j] = tvec[i, a]
            elif kind == "k":
                x[j] = cam_params[i, a]
            else:
                x[j] = xyz[i, a]
        return x

    def unpack(self, x):
        p = dict(self.prob)
        q, t = self.q0.copy(), np.array(self.prob["tvec"], dtype=np.float64)
        k, X = np.array(self.prob["cam_params"], dtype=np.float64), np.array(self.prob["xyz"], dtype=np.float64)
        d = np.zeros((len(q), 3))
        for (kind, i, a), j in self.col.items():
            if kind == "q":
                d[i, a] = x[j]
            elif kind == "t":
                t[i, a] = x[j]
            elif kind == "k":
                k[i, a] = x[j]
            else:
                X[i, a] = x[j]
        v = self.q0 + np.einsum("iab,ib->ia", self.B, d)
        self._vnorm = np.linalg.norm(v, axis=1)
        p.update(qvec=v / self._vnorm[:, None], tvec=t, cam_params=k, xyz=X)
        return p

    def _blocks(self, x):
        """r (n, C) and the local Jacobians L (n, C, 21) with respect to [d (3) | t | X | camera parameters (12)]"""
        p = self.unpack(x)
        _, r, J = pxo.ba_eval_batch(p, self.cfg, pxo.loss("trivial"), want_r=True, want_J=True)
        oi = p["obs_image"]
        # dq/dd = (I - q q^t) B / |v|, and J_q q = 0 (the derivative runs through QuaternionRotatePoint's normalisation)
        Jd = np.einsum("ncq,nqa->nca", J[:, :, :4], self.B[oi]) / self._vnorm[oi][:, None, None]
        return r, np.concatenate([Jd, J[:, :, 4:]], axis=2)

    def _scatter(self, L):
        n, rows, _ = L.shape
        out = np.zeros((n * rows, self.n))
        for b in range(n):
            ok = self.idx[b] >= 0
            out[b * rows:(b + 1) * rows, self.idx[b][ok]] = L[b][:, ok]
        return out

    def block_norms(self, x):
        """f_b = |r_b| and its Jacobian: what scipy's own loss is applied to"""
        r, L = self._blocks(x)
        f = np.linalg.norm(r, axis=1)
        g = np.einsum("nc,ncj->nj", r, L) / f[:, None]
        return f, self._scatter(g[:, None, :])

    def robustified(self, x, rho):
        """r~_b = sqrt(rho(s_b) / s_b) r_b (so that |r~_b|^2 = rho(s_b)) and its Jacobian; rho(s) -> (rho, rho')"""
        r, L = self._blocks(x)
        s = (r * r).sum(1)
        r0, r1 = rho(s)
        w = np.sqrt(r0 / s)
        c = (r1 * s - r0) / (s * s) / w                          # dw = c r^t dr
        Lt = w[:, None, None] * L + r[:, :, None] * (c[:, None] * np.einsum("nc,ncj->nj", r, L))[:, None, :]
        return (r * w[:, None]).reshape(-1), self._scatter(Lt)


RHO = {   # Ceres loss_function.h [upstream]: rho(s), rho'(s)
    "cauchy": lambda a: (lambda s: (a * a * np.log1p(s / (a * a)), 1.0 / (1.0 + s / (a * a)))),
    "huber": lambda a: (lambda s: (np.where(s <= a * a, s, 2 * a * np.sqrt(s) - a * a), np.where(s <= a * a, 1.0, a / np.sqrt(np.maximum(s, 1e-300))))),
    "soft_l1": lambda a: (lambda s: (2 * a * a * (np.sqrt(1 + s / (a * a)) - 1), 1.0 / np.sqrt(1 + s / (a * a)))),
}


def solve(sp, loss, a, x_start=None, search_nfev=300, confirm_nfev=40):
    """-> (x_search, cost_search, x_confirm, cost_confirm, optimality): the optimum found with the hand-robustified full blocks,
    and what scipy's own `loss` (on the block norms, started there) makes of it."""
    from scipy.optimize import least_squares
    memo = {}

    def wrap(fn):
        def f(x):
            r, J = fn(x)
            memo["x"], memo["J"] = x.copy(), J
            return r

        def j(x):
            return memo["J"] if "x" in memo and np.array_equal(memo["x"], x) else fn(x)[1]
        return f, j
    x0 = sp.x0() if x_start is None else x_start
    kw = dict(x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-13, method="trf")
    if loss == "trivial":
        f, j = wrap(lambda x: sp.robustified(x, lambda s: (s, np.ones_like(s))))
        ra = least_squares(f, x0, jac=j, loss="linear", max_nfev=search_nfev, **kw)
        return ra.x, ra.cost, ra.x, ra.cost, ra.optimality
    f, j = wrap(lambda x: sp.robustified(x, RHO[loss](a)))
    ra = least_squares(f, x0, jac=j, loss="linear", max_nfev=search_nfev, **kw)
    f, j = wrap(sp.block_norms)
    rb = least_squares(f, ra.x, jac=j, loss=loss, f_scale=a, max_nfev=confirm_nfev, **kw)
    return ra.x, ra.cost, rb.x, rb.cost, ra.optimality


class ScipyKA:
    """A keypoint-adjustment problem (flat layout of pxr_ka_view, unit edge weights) for scipy: unknowns = the keypoints of the
    non-constant nodes, box bounds = KeypointOptimizerBase::ParameterizeKeypoints' (the oracle's pxo_ka.node_bounds)."""

    def __init__(self, prob, bound, cfg=None):
        import pxo_ka
        self.prob, self.cfg = prob, cfg or pxo.cfg()
        assert np.all(np.asarray(prob["edge_w"]) == 1.0), "scipy's loss cannot weight residuals: use unit weights"
        self.var = np.flatnonzero(np.asarray(prob["node_const"]) == 0)
        self.slot = {int(n): i for i, n in enumerate(self.var)}
        _, H, W, _ = prob["patches"].shape
        b = pxo_ka.node_bounds(prob["kp"], prob["corners"], prob["scales"], H, W, bound)
        self.lb, self.ub = b[self.var, :2].reshape(-1), b[self.var, 2:].reshape(-1)
        self.patch = [pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i]) for i in range(len(prob["kp"]))]

    def x0(self):
        return np.asarray(self.prob["kp"], np.float64)[self.var].reshape(-1).copy()

    def keypoints(self, x):
        kp = np.array(self.prob["kp"], dtype=np.float64)
        kp[self.var] = x.reshape(-1, 2)
        return kp

    def _blocks(self, x):
        kp = self.keypoints(x)
        src, dst = self.prob["edge_src"], self.prob["edge_dst"]
        R, L = [], []
        for s, d in zip(src, dst):
            r, J1, J2 = pxo.ka_residual(self.patch[s], self.patch[d], self.cfg, kp[s], kp[d])
            R.append(r); L.append(np.hstack([J1, J2]))
        return np.array(R), np.array(L)

    def _scatter(self, L):
        n, rows, _ = L.shape
        out = np.zeros((n * rows, 2 * len(self.var)))
        for b, (s, d) in enumerate(zip(self.prob["edge_src"], self.prob["edge_dst"])):
            for node, c0 in ((int(s), 0), (int(d), 2)):
                if node in self.slot:
                    j = 2 * self.slot[node]
                    out[b * rows:(b + 1) * rows, j:j + 2] += L[b][:, c0:c0 + 2]
        return out

    def block_norms(self, x):
        r, L = self._blocks(x)
        f = np.linalg.norm(r, axis=1)
        return f, self._scatter((np.einsum("nc,ncj->nj", r, L) / f[:, None])[:, None, :])

    def robustified(self, x, rho):
        r, L = self._blocks(x)
        s = (r * r).sum(1)
        r0, r1 = rho(s)
        w = np.sqrt(r0 / s)
        c = (r1 * s - r0) / (s * s) / w
        Lt = w[:, None, None] * L + r[:, :, None] * (c[:, None] * np.einsum("nc,ncj->nj", r, L))[:, None, :]
        return (r * w[:, None]).reshape(-1), self._scatter(Lt)


def solve_bounded(sp, loss, a, search_nfev=300, confirm_nfev=40):
    """like solve(), inside the box sp.lb <= x <= sp.ub (scipy's trust-region-reflective method keeps iterates interior)"""
    from scipy.optimize import least_squares
    memo = {}

    def wrap(fn):
        def f(x):
            r, J = fn(x)
            memo["x"], memo["J"] = x.copy(), J
            return r

        def j(x):
            return memo["J"] if "x" in memo and np.array_equal(memo["x"], x) else fn(x)[1]
        return f, j
    x0 = np.clip(sp.x0(), sp.lb + 1e-9, sp.ub - 1e-9)
    kw = dict(x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-13, method="trf", bounds=(sp.lb, sp.ub))
    f, j = wrap(lambda x: sp.robustified(x, RHO[loss](a)))
    ra = least_squares(f, x0, jac=j, loss="linear", max_nfev=search_nfev, **kw)
    f, j = wrap(sp.block_norms)
    rb = least_squares(f, ra.x, jac=j, loss=loss, f_scale=a, max_nfev=confirm_nfev, **kw)
    return ra.x, ra.cost, rb.x, rb.cost, ra.optimality
