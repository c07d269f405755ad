"""Golden vectors for the query refinements' problem construction from the REFERENCE's own code:
SingleQueryKeypointOptimizer::RunQuery + QueryKeypointOptimizer::ParameterizeKeypoint, SingleQueryBundleOptimizer::RunQuery +
QueryBundleOptimizer::ParameterizeQuery (localization/src/*.h) against a recording ceres::Problem, and FindNearestReferences
(localization/src/nearest_references.h), compiled in place (oracle/ref_loc_shim.cc -> oracle/_ref/libpxo_ref_loc.so).

    python tests/golden/make_golden_localization.py        # writes tests/golden/localization_ref.npz

Every reference descriptor carries a tag in its first entry (1000 i + 999 for correspondence i's own descriptor, 1000 i + r
for its r-th per-observation descriptor); the shim recovers from each recorded residual block which descriptor it holds."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_loc.so")
NPARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8}
N_QKA, N_QBA, N_NEAREST = 40, 40, 6


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def qka_case(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(1, 13))
    n_patches = n + int(rng.integers(0, 4))
    scales = rng.choice([1.0, 0.5, 0.25], size=(n_patches, 2)) * rng.uniform(0.9, 1.1, (n_patches, 1))
    corners = rng.integers(0, 400, size=(n_patches, 2)).astype(np.int32)
    patch_idxs = rng.permutation(n_patches)[:n].astype(np.int32) if seed % 3 == 0 else None
    own = np.arange(n) if patch_idxs is None else patch_idxs
    # keypoints inside their patch, image coordinates
    kp = (corners[own] + 0.5 + rng.uniform(2, 14, (n, 2))) / scales[own]
    mode = seed % 3 if seed % 7 else 2
    ref_count = {0: np.ones(n), 1: rng.integers(0, 4, n), 2: rng.choice([0, 0, 1, 3, 4], n)}[mode].astype(np.int32)
    inliers = None
    if seed % 4 == 1:
        inliers = (rng.uniform(size=n) < 0.6).astype(np.uint8)
    if seed == 9:
        inliers = np.zeros(n, np.uint8)                       # nothing left: RunQuery returns false
    return dict(kp=kp, corners=corners, scales=scales, sparse=int(seed % 5 != 4), bound=float(rng.choice([-1.0, 0.0, 4.0, 0.5, 40.0])),
                mode=mode, ref_count=ref_count, patch_idxs=patch_idxs, inliers=inliers)


def run_qka(c):
    lib = C.CDLL(LIB)
    n = len(c["kp"])
    cap = 64
    nb = C.c_int32()
    blk_kp, blk_tag = np.zeros(cap, np.int32), np.zeros(cap)
    lower, upper = np.zeros((n, 2)), np.zeros((n, 2))
    kp = np.ascontiguousarray(c["kp"], np.float64)
    rc = lib.pxo_ref_qka_setup(n, _p(kp), len(c["corners"]), _p(c["corners"]), _p(np.ascontiguousarray(c["scales"])), c["sparse"],
                               C.c_double(c["bound"]), c["mode"], _p(c["ref_count"]), _p(c["patch_idxs"]), _p(c["inliers"]), cap,
                               C.byref(nb), _p(blk_kp), _p(blk_tag), _p(lower), _p(upper))
    assert rc in (0, 1), rc
    return dict(solved=np.array(rc), blk_kp=blk_kp[:nb.value].copy(), blk_tag=blk_tag[:nb.value].copy(), lower=lower, upper=upper)


def qba_case(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 13))
    model = int(rng.integers(0, 5))
    base = {0: [1200.0, 500, 500], 1: [1200.0, 1180.0, 500, 500], 2: [1200.0, 500, 500, 0.02], 3: [1200.0, 500, 500, 0.02, -0.01],
            4: [1200.0, 1180.0, 500, 500, 0.02, -0.01, 1e-3, -5e-4]}[model]
    points = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), rng.uniform(3, 5, (n, 1))], 1)
    n_patches = n + int(rng.integers(0, 3))
    patch_idxs = rng.permutation(n_patches)[:n].astype(np.int32) if seed % 3 == 0 else None
    mode = seed % 3 if seed % 7 else 2
    ref_count = {0: np.ones(n), 1: rng.integers(0, 4, n), 2: rng.choice([0, 0, 1, 3, 4], n)}[mode].astype(np.int32)
    inliers = (rng.uniform(size=n) < 0.6).astype(np.uint8) if seed % 4 == 1 else None
    if seed == 9:
        inliers = np.zeros(n, np.uint8)
    return dict(points=points, model=model, params=np.array(base), qvec=np.array([1.0, 0, 0, 0]), tvec=np.zeros(3),
                corners=np.full((n_patches, 2), 490, np.int32), scales=np.ones((n_patches, 2)),
                refine=[int(x) for x in rng.integers(0, 2, 3)] if seed % 2 else [0, 0, 0], mode=mode, ref_count=ref_count,
                patch_idxs=patch_idxs, inliers=inliers)


def run_qba(c):
    lib = C.CDLL(LIB)
    n = len(c["points"])
    cap = 64
    nb, cam_const, quat = C.c_int32(), C.c_int32(), C.c_int32()
    blk_point, blk_tag, point_const = np.zeros(cap, np.int32), np.zeros(cap), np.zeros(n, np.uint8)
    rc = lib.pxo_ref_qba_setup(n, _p(np.ascontiguousarray(c["points"])), c["model"], len(c["params"]), _p(c["params"]), _p(c["qvec"]), _p(c["tvec"]),
                               len(c["corners"]), _p(c["corners"]), _p(c["scales"]), c["refine"][0], c["refine"][1], c["refine"][2], c["mode"],
                               _p(c["ref_count"]), _p(c["patch_idxs"]), _p(c["inliers"]), cap, C.byref(nb), _p(blk_point), _p(blk_tag),
                               _p(point_const), C.byref(cam_const), C.byref(quat))
    assert rc in (0, 1), rc
    return dict(solved=np.array(rc), blk_point=blk_point[:nb.value].copy(), blk_tag=blk_tag[:nb.value].copy(), point_const=point_const,
                camera_const=np.array(cam_const.value), quaternion=np.array(quat.value))


def nearest_case(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 9))
    patches = rng.normal(size=(n, 16, 16, 128)).astype(np.float16)
    corners = rng.integers(0, 300, size=(n, 2)).astype(np.int32)
    scales = np.tile(rng.choice([1.0, 0.5], 2), (n, 1))
    kp = (corners + 0.5 + rng.uniform(3, 13, (n, 2))) / scales
    cand_count = rng.integers(1, 6, n).astype(np.int32)
    cand = rng.normal(size=(int(cand_count.sum()), 128))
    cand /= np.linalg.norm(cand, axis=1, keepdims=True)
    return dict(patches=patches, corners=corners, scales=scales, kp=kp, cand_count=cand_count, cand=cand, l2=int(seed % 2 == 0))


def run_nearest(c):
    lib = C.CDLL(LIB)
    n = len(c["kp"])
    chosen, out = np.zeros(n, np.int32), np.zeros((n, 128))
    rc = lib.pxo_ref_nearest_references(n, _p(np.ascontiguousarray(c["kp"])), _p(c["patches"]), _p(c["corners"]), _p(np.ascontiguousarray(c["scales"])),
                                        _p(c["cand_count"]), _p(np.ascontiguousarray(c["cand"])), c["l2"], _p(chosen), _p(out))
    assert rc == 0, rc
    return dict(chosen=chosen, descriptor=out)


def main():
    store = {}
    for kind, n, case, run in (("qka", N_QKA, qka_case, run_qka), ("qba", N_QBA, qba_case, run_qba), ("nearest", N_NEAREST, nearest_case, run_nearest)):
        for s in range(n):
            for k, v in run(case(s)).items():
                store["%s%d|%s" % (kind, s, k)] = v
    np.savez_compressed(os.path.join(HERE, "localization_ref.npz"), **store)
    print("written", len(store), "arrays")


if __name__ == "__main__":
    main()
