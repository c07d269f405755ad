"""ctypes binding of the oracle's keypoint-adjustment solver (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C

import numpy as np

import pxo


class KaBatch(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("kp", C.c_void_p), ("node_patch", C.c_void_p), ("node_const", C.c_void_p),
                ("n_edges", C.c_int64), ("edge_src", C.c_void_p), ("edge_dst", C.c_void_p), ("edge_w", C.c_void_p),
                ("arena", C.c_void_p), ("dtype", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("corners", C.c_void_p), ("scales", C.c_void_p),
                ("n_unary", C.c_int64), ("unary_node", C.c_void_p), ("unary_ref", C.c_void_p), ("unary_w", C.c_void_p)]


def ka_solve(problem, config, ls, bound=4.0, opts=None):
    """Solves every sub-problem (node_problem labels) with the oracle LM.  Returns
    (refined keypoints copy, list of per-problem summary dicts)."""
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(problem[name], dtype=dt)
        keep[name] = a
        return a.ctypes.data

    kp = np.array(problem["kp"], dtype=np.float64, order="C", copy=True)
    patches = np.ascontiguousarray(problem["patches"])
    n_p, H, W, ch = patches.shape
    b = KaBatch(len(kp), kp.ctypes.data, arr("node_patch", np.int64), arr("node_const", np.uint8),
                len(problem["edge_src"]), arr("edge_src", np.int32), arr("edge_dst", np.int32), arr("edge_w", np.float64),
                patches.ctypes.data, pxo._NP2DT[patches.dtype], H, W, ch, arr("corners", np.int32),
                arr("scales", np.float64), 0, None, None, None)
    unary_problem = np.zeros(0, dtype=np.int64)
    if "unary_node" in problem and len(problem["unary_node"]):
        b.n_unary = len(problem["unary_node"])
        b.unary_node = arr("unary_node", np.int32)
        b.unary_ref = arr("unary_ref", np.float64)
        if problem.get("unary_w") is not None:
            b.unary_w = arr("unary_w", np.float64)
    opts = opts or pxo.lm_options(parameter_tolerance=1e-5)
    node_problem = np.asarray(problem["node_problem"])
    edge_problem = node_problem[np.asarray(problem["edge_src"], dtype=np.int64)]
    if b.n_unary:
        unary_problem = node_problem[np.asarray(problem["unary_node"], dtype=np.int64)]
    summaries = []
    lib = pxo.lib()
    for p in range(int(node_problem.max()) + 1):
        nodes = np.ascontiguousarray(np.nonzero(node_problem == p)[0], dtype=np.int32)
        edges = np.ascontiguousarray(np.nonzero(edge_problem == p)[0], dtype=np.int32)
        unary = np.ascontiguousarray(np.nonzero(unary_problem == p)[0], dtype=np.int32)
        s = pxo.LMSummary()
        rc = lib.pxo_ka_solve_problem_u(C.byref(b), pxo._p(nodes), len(nodes), pxo._p(edges), len(edges),
                                        pxo._p(unary), len(unary), C.byref(config),
                                        C.byref(ls), C.c_double(bound), C.byref(opts), C.byref(s))
        assert rc == 0
        summaries.append(s.as_dict())
    return kp, summaries


def node_bounds(kp, corners, scales, H, W, bound):
    """Box bounds (n, 4) = lower x, lower y, upper x, upper y of every keypoint, KeypointOptimizerBase::ParameterizeKeypoints
    (keypoint_optimizer.h:124-152): the patch extent in image pixels, intersected with +- bound / scale around the keypoint."""
    kp = np.ascontiguousarray(kp, dtype=np.float64)
    n = len(kp)
    corners = np.ascontiguousarray(corners, dtype=np.int32)
    scales = np.ascontiguousarray(scales, dtype=np.float64)
    node_patch = np.arange(n, dtype=np.int64)
    b = KaBatch(n, kp.ctypes.data, node_patch.ctypes.data, None, 0, None, None, None, None, 0, H, W, 128, corners.ctypes.data,
                scales.ctypes.data, 0, None, None, None)
    out = np.empty((n, 4))
    lo, hi = (C.c_double * 2)(), (C.c_double * 2)()
    lib = pxo.lib()
    for i in range(n):
        lib.pxo_ka_node_bounds(C.byref(b), C.c_int64(i), C.c_double(bound), lo, hi)
        out[i] = (lo[0], lo[1], hi[0], hi[1])
    return out
