"""Mirror of pixsfm's query keypoint adjustment (QKA) for the accelerated path:
`QueryKeypointAdjuster(conf).refine(pnp_points2D, fmap, references, point2D_idxs)`
(pixsfm/localization/main.py:89-192) and the `_localization.QueryKeypointOptimizer` it drives
(pixsfm/localization/bindings.cc:36-57, src/single_query_keypoint_optimizer.h:86-222).

One Ceres problem over all 2D-3D correspondences of a query, one FeatureReference2DCostFunctor
block per (keypoint, reference descriptor), box bounds from the patch extent and `bound`
(src/query_keypoint_optimizer.h:141-172).  Here: one pxr_ka_solve sub-problem whose terms are the
unary reference terms of pxr_ka_view; the keypoints are independent 2x2 components of it.

`QueryBundleAdjuster(conf).refine(qvec, tvec, camera, points3D, fmap, references, inliers,
point2D_idxs)` (main.py:194-258; src/single_query_bundle_optimizer.h:91-222, query_bundle_optimizer.h:
114-151) is the one-image BA: FeatureReferenceCostFunctor blocks against CONSTANT 3D points, pose on
the quaternion manifold, intrinsics constant unless refine_* says otherwise.  It runs on pxr_ba_solve
with every (correspondence, reference descriptor) pair as its own constant point.

PnP / retrieval / QueryLocalizer orchestration (main.py:261-560) stay outside the accelerated path
(control plane).
"""
from copy import deepcopy

import numpy as np

from ..engine import BAProblem, lm_options, make_loss, nearest_references
from ..ka_engine import KAProblem
from . import base, features
from .keypoint_adjustment import default_context


def resolve_level_indices(level_indices, n_levels):      # pixsfm/util/misc.py:19-23
    if level_indices not in [None, "all"]:
        return level_indices
    return list(reversed(range(n_levels)))


def _reference_descriptors(ref):
    """The descriptors one correspondence contributes (single_query_keypoint_optimizer.h:86-222)."""
    if isinstance(ref, features.Reference):
        return ref.observations if ref.has_observations() else [ref.descriptor]
    if isinstance(ref, (list, tuple)):
        return [np.asarray(r, dtype=np.float64).reshape(1, -1) for r in ref]
    return [np.asarray(ref, dtype=np.float64).reshape(1, -1)]


def _build_problem(keypoints, fmap, references, patch_idxs, inliers, per_keypoint_problems=False):
    n = len(keypoints)
    if len(references) != n:
        raise ValueError("references.size() != keypoints.rows()")           # THROW_CHECK_EQ, :96 / :134 / :180
    if patch_idxs is not None and len(patch_idxs) != n:
        raise ValueError("patch_idxs.size() != keypoints.rows()")           # :70-72
    rows, patches, unary_node, unary_ref = [], [], [], []
    for idx in range(n):
        if inliers is not None and not inliers[idx]:
            continue
        descs = _reference_descriptors(references[idx])
        if not descs:
            continue                                                        # added_to_problem == false, :140-154
        node = len(rows)
        rows.append(idx)
        patches.append(fmap.fpatch(idx if patch_idxs is None else patch_idxs[idx]))
        for d in descs:
            unary_node.append(node)
            unary_ref.append(d.reshape(-1))
    m = len(rows)
    prob = dict(kp=np.asarray(keypoints, dtype=np.float64)[rows].reshape(-1, 2),
                node_patch=np.arange(m, dtype=np.int64), node_const=np.zeros(m, np.uint8),
                node_problem=np.arange(m, dtype=np.int32) if per_keypoint_problems else np.zeros(m, np.int32),
                edge_src=np.zeros(0, np.int32), edge_dst=np.zeros(0, np.int32), edge_w=np.zeros(0),
                unary_node=np.asarray(unary_node, dtype=np.int32),
                unary_ref=(np.asarray(unary_ref, dtype=np.float64).reshape(len(unary_node), -1) if unary_node else np.zeros((0, 0))),
                unary_w=None)      # no residual at all (every correspondence an outlier): the callers return False
    return rows, patches, prob


def find_feature_inliers(p2Ds, fmap, references, interpolation_config, thresh=-1, ctx=None):
    """localization/main.py:20-35: keep a correspondence when |f(patch_i, p2D_i) - reference_i| <= thresh
    (only array references are tested; patch index = correspondence index, as in the reference).
    The descriptor distances come from the device: one zero-iteration sub-problem per keypoint,
    whose initial cost is 0.5 |f - ref|^2 under the trivial loss."""
    inliers = [True] * len(p2Ds)
    if thresh < 0.0 or len(p2Ds) == 0:
        return inliers
    tested = [i for i, r in enumerate(references) if isinstance(r, np.ndarray)]
    if not tested:
        return inliers
    ic = interpolation_config if isinstance(interpolation_config, base.InterpolationConfig) \
        else base.InterpolationConfig(interpolation_config)
    ctx = ctx or default_context()
    mask = [i in set(tested) for i in range(len(p2Ds))]
    rows, patches, prob = _build_problem(np.asarray(p2Ds, dtype=np.float64), fmap, references, None, mask,
                                         per_keypoint_problems=True)
    arena = features.to_arena(ctx, patches)
    prob['node_patch'] = arena.index
    ka = KAProblem(ctx, arena, prob)
    _, per = ka.solve(ic.to_engine(), make_loss('trivial', []), bound=0.0, options=lm_options(max_iterations=0),
                      per_problem=True)
    arena.close()
    for i, s in zip(rows, per):
        if np.sqrt(2.0 * s['initial_cost']) > thresh:
            inliers[i] = False
    return inliers


def find_nearest_references(query_fmap, references, keypoints, point3D_ids, interpolation_config, patch_idxs=None,
                            ctx=None):
    """_localization.find_nearest_references (bindings.cc:60; src/nearest_references.h:20-52): for every
    2D-3D correspondence, the observation descriptor of its 3D point's Reference that is nearest to the
    query descriptor interpolated at the keypoint.  `references`: {point3D_id: Reference} extracted with
    keep_observations=True.  Returns a list of (1, C) arrays."""
    ic = interpolation_config if isinstance(interpolation_config, base.InterpolationConfig) \
        else base.InterpolationConfig(interpolation_config)
    keypoints = np.asarray(keypoints, dtype=np.float64).reshape(-1, 2)
    n = len(keypoints)
    if len(point3D_ids) != n or (patch_idxs is not None and len(patch_idxs) != n):
        raise ValueError("keypoints, point3D_ids and patch_idxs must have the same length")
    if n == 0:
        return []
    cand, ptr = [], [0]
    for pid in point3D_ids:
        ref = references[pid]
        if not ref.has_observations():
            raise ValueError("Missing observations in references. Extract references with option "
                             "keep_observations=True.")                     # THROW_CHECK_MSG, :33-35
        cand.extend(o.reshape(-1) for o in ref.observations)
        ptr.append(len(cand))
    ctx = ctx or default_context()
    patches = [query_fmap.fpatch(i if patch_idxs is None else patch_idxs[i]) for i in range(n)]
    arena = features.to_arena(ctx, patches)
    cand = np.asarray(cand, dtype=np.float64)
    best, _, _ = nearest_references(ctx, arena, ic.to_engine(), keypoints, arena.index, ptr, cand)
    arena.close()
    return [cand[b].reshape(1, -1).copy() for b in best]


class QueryKeypointOptimizer:
    """_localization.QueryKeypointOptimizer: ctor (options, interpolation_config);
    run(keypoints, fmap, references, patch_idxs=None, inliers=None) refines `keypoints` in place and
    returns False when the problem has no residuals (query_keypoint_optimizer.h:56-59)."""

    # the C++ struct's own defaults (query_refinement_options.h:60-95); QueryKeypointAdjuster.default_conf carries the
    # Python-level ones (bound 4, parameter_tolerance 1e-5) and always passes them in full
    option_defaults = {
        'loss': {'name': 'trivial', 'params': []},
        'solver': {**base.solver_default_conf, 'parameter_tolerance': 1e-04, 'callbacks': []},
        'print_summary': True, 'bound': -1.0,
    }

    def __init__(self, options=None, interpolation_config=None, ctx=None):
        self.options = base.merge_conf(self.option_defaults, options)
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx = ctx
        self.last_summary = None

    def run(self, keypoints, fmap, references, patch_idxs=None, inliers=None):
        keypoints = np.asarray(keypoints)
        if keypoints.ndim != 2 or keypoints.shape[1] != 2 or keypoints.dtype != np.float64:
            raise ValueError("keypoints must be an (N, 2) float64 array (refined in place)")
        rows, patches, prob = _build_problem(keypoints, fmap, references, patch_idxs, inliers)
        if len(prob['unary_node']) == 0:
            return False
        ctx = self.ctx or default_context()
        arena = features.to_arena(ctx, patches)
        prob['node_patch'] = arena.index
        o, s = self.options, self.options['solver']
        # ParameterizeKeypoint: bounds when bound > 0 or the map is sparse (:145); the accelerated
        # path holds sparse maps only (features.to_arena), so the patch box always applies.
        ka = KAProblem(ctx, arena, prob)
        lm = lm_options(max_iterations=s['max_num_iterations'], function_tolerance=s['function_tolerance'],
                        gradient_tolerance=s['gradient_tolerance'], parameter_tolerance=s['parameter_tolerance'],
                        max_consecutive_invalid_steps=s['max_num_consecutive_invalid_steps'])
        total, _ = ka.solve(self.interpolation.to_engine(), make_loss(o['loss']['name'], o['loss']['params']),
                            bound=o['bound'], options=lm)
        keypoints[rows] = ka.keypoints()
        self.last_summary = total
        arena.close()
        return True

    def run_batch(self, queries):
        """Many queries in ONE launch ("batch across queries for throughput"): `queries` is a list of
        (keypoints, fmap, references, patch_idxs | None, inliers | None) tuples with run()'s meaning.  Each query
        is its own sub-problem -- its own trust region, step acceptance and termination, exactly as if run() had
        been called on it -- but all of them share one arena upload and one pxr_ka_solve call.  Returns the list
        of run()'s booleans; per-query summaries in self.last_summaries."""
        built = []
        for q in queries:
            keypoints, fmap, references = q[0], q[1], q[2]
            patch_idxs = q[3] if len(q) > 3 else None
            inliers = q[4] if len(q) > 4 else None
            keypoints = np.asarray(keypoints)
            if keypoints.ndim != 2 or keypoints.shape[1] != 2 or keypoints.dtype != np.float64:
                raise ValueError("keypoints must be an (N, 2) float64 array (refined in place)")
            built.append((keypoints,) + _build_problem(keypoints, fmap, references, patch_idxs, inliers))
        live = [i for i, b in enumerate(built) if len(b[3]['unary_node'])]
        ok = [False] * len(queries)
        self.last_summaries = [None] * len(queries)
        if not live:
            return ok
        patches, parts, n0, u0 = [], [], 0, 0
        for slot, i in enumerate(live):
            _, rows, plist, prob = built[i]
            m = len(rows)
            patches += plist
            parts.append(dict(kp=prob['kp'], node_patch=prob['node_patch'] + n0, node_problem=np.full(m, slot, np.int32),
                              unary_node=prob['unary_node'] + n0, unary_ref=prob['unary_ref']))
            n0 += m
        cat = {k: np.concatenate([p_[k] for p_ in parts]) for k in parts[0]}
        cat.update(node_const=np.zeros(n0, np.uint8), edge_src=np.zeros(0, np.int32), edge_dst=np.zeros(0, np.int32),
                   edge_w=np.zeros(0), unary_w=None)
        ctx = self.ctx or default_context()
        arena = features.to_arena(ctx, patches)
        cat['node_patch'] = arena.index[cat['node_patch']]
        o, s = self.options, self.options['solver']
        ka = KAProblem(ctx, arena, cat)
        lm = lm_options(max_iterations=s['max_num_iterations'], function_tolerance=s['function_tolerance'],
                        gradient_tolerance=s['gradient_tolerance'], parameter_tolerance=s['parameter_tolerance'],
                        max_consecutive_invalid_steps=s['max_num_consecutive_invalid_steps'])
        total, per = ka.solve(self.interpolation.to_engine(), make_loss(o['loss']['name'], o['loss']['params']),
                              bound=o['bound'], options=lm, per_problem=True)
        out = ka.keypoints()
        n0 = 0
        for slot, i in enumerate(live):
            keypoints, rows = built[i][0], built[i][1]
            keypoints[rows] = out[n0:n0 + len(rows)]
            n0 += len(rows)
            ok[i] = True
            self.last_summaries[i] = per[slot]
        self.last_summary = total
        arena.close()
        return ok


class QueryKeypointAdjuster:
    """pixsfm/localization/main.py:89-192."""

    default_conf = {
        'apply': True,
        'feature_inlier_thresh': -1,
        'interpolation': base.interpolation_default_conf,
        'level_indices': None,
        'stack_correspondences': False,
        'optimizer': {
            'loss': {'name': 'trivial', 'params': []},
            'solver': {**base.solver_default_conf, 'parameter_tolerance': 1e-05},
            'print_summary': False,
            'bound': 4.0,
        },
    }

    def __init__(self, conf=None, callbacks=None, ctx=None):
        self.conf = base.merge_conf(deepcopy(self.default_conf), conf)
        self.ctx = ctx
        self.solver = QueryKeypointOptimizer(self.conf['optimizer'], self.conf['interpolation'], ctx=ctx)

    def refine(self, pnp_points2D, fmap, references, point2D_idxs=None):
        qka_inliers = find_feature_inliers(pnp_points2D, fmap, references, self.conf['interpolation'],
                                           thresh=self.conf['feature_inlier_thresh'], ctx=self.ctx)
        if self.conf['stack_correspondences']:
            self.refine_stacked(pnp_points2D, fmap, references, point2D_idxs, inliers=qka_inliers)
        else:
            self.solver.run(pnp_points2D, fmap, references, patch_idxs=point2D_idxs, inliers=qka_inliers)

    def refine_multilevel(self, pnp_points2D, query_fmaps, references, point2D_idxs=None):
        for l_idx in resolve_level_indices(self.conf['level_indices'], len(query_fmaps)):
            self.refine(pnp_points2D, query_fmaps[l_idx], references[l_idx], point2D_idxs=point2D_idxs)

    def refine_batch(self, queries):
        """refine() for many queries at once: `queries` = [(pnp_points2D, fmap, references, point2D_idxs | None), ...];
        one GPU launch, each query solved as its own problem (not available with stack_correspondences)."""
        if self.conf['stack_correspondences']:
            raise ValueError("refine_batch does not stack correspondences; call refine() per query")
        batch = []
        for q in queries:
            pts, fmap, refs = q[0], q[1], q[2]
            idxs = q[3] if len(q) > 3 else None
            inl = find_feature_inliers(pts, fmap, refs, self.conf['interpolation'],
                                       thresh=self.conf['feature_inlier_thresh'], ctx=self.ctx)
            batch.append((pts, fmap, refs, idxs, inl))
        return self.solver.run_batch(batch)

    def refine_stacked(self, pnp_points2D, fmap, references, point2D_idxs, inliers=None):
        if point2D_idxs is None:
            raise ValueError("point2D_idxs must not be None in stacked QKA.")
        unique_p2D_idxs = list(dict.fromkeys(point2D_idxs))
        pos = {p: i for i, p in enumerate(unique_p2D_idxs)}
        old_to_new = [pos[p] for p in point2D_idxs]
        unique_kps = np.zeros((len(unique_p2D_idxs), 2), dtype=np.float64)
        for idx, new in enumerate(old_to_new):
            unique_kps[new] = pnp_points2D[idx]
        stacked_refs = [[] for _ in unique_p2D_idxs]
        for idx, query_ref in enumerate(references):
            if not isinstance(query_ref, np.ndarray):
                raise ValueError("Stacked QKA requires a np.ndarray reference for each 2D-3D correspondence. "
                                 "Consider setting target_references='nearest'.")
            stacked_refs[old_to_new[idx]].append(query_ref)
        # the reference forwards the per-correspondence inlier list to the per-keypoint problem
        # (main.py:186-188); it only has the right length when every point2D_idx is unique
        run_inliers = inliers if inliers is not None and len(inliers) == len(unique_p2D_idxs) else None
        self.solver.run(unique_kps, fmap, stacked_refs, patch_idxs=unique_p2D_idxs, inliers=run_inliers)
        for i in range(len(point2D_idxs)):
            pnp_points2D[i] = unique_kps[old_to_new[i]]


def _qba_observations(points3D, fmap, references, inliers, patch_idxs):
    """The residual blocks of a query BA (single_query_bundle_optimizer.h:92-221): one per correspondence and reference
    descriptor, inlier correspondences only, in that order.  -> (correspondence index, patch, xyz, descriptor) lists."""
    rows, patches, xyz, refs = [], [], [], []
    for idx in range(len(points3D)):
        if inliers is not None and not inliers[idx]:
            continue
        patch = fmap.fpatch(idx if patch_idxs is None else patch_idxs[idx])
        for d in _reference_descriptors(references[idx]):
            rows.append(idx)
            patches.append(patch)
            xyz.append(np.asarray(points3D[idx], dtype=np.float64).reshape(3))
            refs.append(d.reshape(-1))
    return rows, patches, xyz, refs


def _qba_camera_mask(camera, options):
    """ParameterizeQuery (query_bundle_optimizer.h:113-146): bit mask of the camera parameters held constant."""
    K = len(camera.params)
    if not (options['refine_focal_length'] or options['refine_principal_point'] or options['refine_extra_params']):
        return (1 << K) - 1                                                   # :121-126
    const = []
    if not options['refine_focal_length']:
        const += camera.focal_length_idxs()
    if not options['refine_principal_point']:
        const += camera.principal_point_idxs()
    if not options['refine_extra_params']:
        const += camera.extra_params_idxs()
    return sum(1 << a for a in const)


class QueryBundleOptimizer:
    """_localization.QueryBundleOptimizer: ctor (options, interpolation_config);
    run(qvec, tvec, camera, points3D, fmap, references, inliers=None, patch_idxs=None) refines qvec,
    tvec (numpy arrays) and camera.params in place; False when no residual was added."""

    # the C++ struct's own defaults (query_refinement_options.h:8-57)
    option_defaults = {
        'loss': {'name': 'cauchy', 'params': [0.25]},
        'solver': {**base.solver_default_conf, 'parameter_tolerance': 1e-05, 'callbacks': []},
        'print_summary': True,
        'refine_focal_length': False, 'refine_principal_point': False, 'refine_extra_params': False,
    }

    def __init__(self, options=None, interpolation_config=None, ctx=None):
        self.options = base.merge_conf(self.option_defaults, options)
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx = ctx
        self.last_summary = None

    def run(self, qvec, tvec, camera, points3D, fmap, references, inliers=None, patch_idxs=None):
        n = len(points3D)
        if len(references) != n:
            raise ValueError("references.size() != points3D.size()")          # THROW_CHECK_EQ, :100 / :141 / :183
        if patch_idxs is not None and len(patch_idxs) != n:
            raise ValueError("patch_idxs.size() != points3D.size()")
        if inliers is not None and len(inliers) != n:
            raise ValueError("inliers.size() != points3D.size()")
        for name, v, k in (("qvec", qvec, 4), ("tvec", tvec, 3)):
            if not isinstance(v, np.ndarray) or v.dtype != np.float64 or v.size != k:
                raise ValueError("%s must be a float64 numpy array of %d values (refined in place)" % (name, k))
        rows, patches, xyz, refs = _qba_observations(points3D, fmap, references, inliers, patch_idxs)
        if not patches:
            return False
        ctx = self.ctx or default_context()
        arena = features.to_arena(ctx, patches)
        m = len(patches)
        o, s = self.options, self.options['solver']
        q = qvec.reshape(4)
        params = np.zeros((1, 12))
        params[0, :len(camera.params)] = camera.params
        prob = dict(obs_image=np.zeros(m, np.int32), obs_point=np.arange(m, dtype=np.int32),
                    obs_patch=arena.index, image_camera=np.zeros(1, np.int32),
                    qvec=q.reshape(1, 4).copy(), tvec=tvec.reshape(1, 3).copy(),
                    cam_model=np.array([camera.model_id], np.int32), cam_params=params,
                    xyz=np.array(xyz), refs=np.array(refs))
        ba = BAProblem(ctx, arena, prob)
        K = len(camera.params)
        cam_mask = _qba_camera_mask(camera, o)
        lm = lm_options(max_iterations=s['max_num_iterations'], function_tolerance=s['function_tolerance'],
                        gradient_tolerance=s['gradient_tolerance'], parameter_tolerance=s['parameter_tolerance'],
                        max_consecutive_invalid_steps=s['max_num_consecutive_invalid_steps'],
                        use_inner_iterations=False)     # every point is constant: nothing for inner iterations
        callbacks = list(s.get('callbacks') or [])
        if callbacks:
            ctx.set_iteration_callbacks(callbacks)
        try:
            summary = ba.solve(self.interpolation.to_engine(), make_loss(o['loss']['name'], o['loss']['params']),
                               np.zeros(1, np.uint8), np.zeros(1, np.uint8), np.array([cam_mask], np.uint16),
                               np.ones(m, np.uint8), options=lm)
        finally:
            if callbacks:
                ctx.set_iteration_callbacks(None)
        q_out, t_out, cam_out, _ = ba.params()
        qvec.reshape(4)[:] = q_out[0]
        tvec.reshape(3)[:] = t_out[0]
        camera.params[:] = cam_out[0, :K]
        self.last_summary = summary
        arena.close()
        return True


class QueryBundleAdjuster:
    """pixsfm/localization/main.py:194-258."""

    default_conf = {
        'apply': True,
        'interpolation': base.interpolation_default_conf,
        'level_indices': None,
        'optimizer': {
            'loss': {'name': 'cauchy', 'params': [0.25]},
            'solver': {**base.solver_default_conf},
            'print_summary': False,
            'refine_focal_length': False, 'refine_principal_point': False, 'refine_extra_params': False,
        },
    }

    def __init__(self, conf=None, callbacks=None, ctx=None):
        self.conf = base.merge_conf(deepcopy(self.default_conf), conf)
        self.solver = QueryBundleOptimizer(self.conf['optimizer'], self.conf['interpolation'], ctx=ctx)

    def refine(self, qvec, tvec, camera, points3D, fmap, references, inliers=None, point2D_idxs=None):
        return self.solver.run(qvec, tvec, camera, points3D, fmap, references, inliers=inliers,
                               patch_idxs=point2D_idxs)

    def refine_multilevel(self, qvec, tvec, camera, points3D, fmaps, references, inliers=None, point2D_idxs=None):
        assert len(fmaps) == len(references)
        for level in resolve_level_indices(self.conf['level_indices'], len(fmaps)):
            self.refine(qvec, tvec, camera, points3D, fmaps[level], references[level], inliers=inliers,
                        point2D_idxs=point2D_idxs)
