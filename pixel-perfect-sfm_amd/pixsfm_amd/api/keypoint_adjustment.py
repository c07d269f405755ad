"""Mirror of pixsfm's keypoint-adjustment surface for the accelerated path:
`KeypointAdjuster.create(conf).refine_multilevel(keypoints, feature_manager, graph)`
(pixsfm/keypoint_adjustment/main.py:60-247) and the `_keypoint_adjustment` pybind classes it
drives (pixsfm/keypoint_adjustment/bindings.cc:17-81).  Same names, argument meaning and
defaults; the Ceres/C++ solve is replaced by pxr_ka_solve on the GPU.
"""
import ctypes as C
from copy import deepcopy

import numpy as np

from .. import _lib
from ..engine import Context, lm_options, make_loss
from ..ka_engine import KAProblem, pack_tracks_into_problems
from . import base, features

# pixsfm's name for the packing of tracks into sub-problems (keypoint_adjustment/main.py:13-57)
find_problem_labels = pack_tracks_into_problems

_default_ctx = None


def default_context():
    """The process-wide engine context: device PXR_DEVICE / LOCAL_RANK (one process per GPU), else 0."""
    global _default_ctx
    if _default_ctx is None:
        from ..parallel import local_device
        _default_ctx = Context(local_device())
    return _default_ctx


class KeypointAdjustmentSetup:
    """keypoint_adjustment/src/keypoint_adjustment_options.h:19-40."""

    def __init__(self):
        self.constant_images = set()
        self.constant_keypoints = {}

    def set_image_constant(self, image_id):
        self.constant_images.add(int(image_id))

    def set_keypoint_constant(self, image_id, feature_idx):
        self.constant_keypoints.setdefault(int(image_id), set()).add(int(feature_idx))

    def set_keypoints_constant(self, image_id, feature_idxs):
        for f in feature_idxs:
            self.set_keypoint_constant(image_id, f)

    def set_node_constant(self, node):
        self.set_keypoint_constant(node.image_id, node.feature_idx)

    def set_masked_nodes_constant(self, graph, mask):
        if len(mask) != len(graph.nodes):
            raise ValueError("mask size does not match the number of graph nodes")
        for node, m in zip(graph.nodes, mask):
            if m:
                self.set_node_constant(node)

    def is_keypoint_constant(self, image_id, feature_idx):
        return int(image_id) in self.constant_images or \
            int(feature_idx) in self.constant_keypoints.get(int(image_id), ())

    def is_node_constant(self, node):
        return self.is_keypoint_constant(node.image_id, node.feature_idx)


class Summary:
    """The ceres::Solver::Summary fields pixsfm reads (util/src/statistics.h:162-217)."""

    def __init__(self, d, num_residuals=0):
        self.initial_cost, self.final_cost = d["initial_cost"], d["final_cost"]
        self.num_iterations = d["iterations"]
        self.num_successful_steps = d["num_successful"]
        self.termination_type = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}[d["termination"]]
        self.total_time_in_seconds = d["total_ms"] * 1e-3
        self.num_residuals_reduced = num_residuals
        self.raw = d
        # ceres::Solver::Summary::iterations: per-iteration records of the BA solves (filled by the bundle optimizers through
        # pxr_set_iteration_callback); the keypoint adjustments run their LM loops inside one kernel and leave it empty
        self.iterations = []

    def BriefReport(self):
        return "pixsfm_amd: iterations %d, initial cost %e, final cost %e, termination %s" % (
            self.num_iterations, self.initial_cost, self.final_cost, self.termination_type)


def build_edges(graph, keypoints, track_labels, root_labels, nodes_in_problem=None, weight_by_sim=True,
                root_edges_only=False, root_regularize_weight=-1.0):
    """TopologicalKeypointOptimizer::SetUp + FeatureMetricKeypointOptimizer::AddIntraResiduals
    (topological_keypoint_optimizer.h:97-175, featuremetric_keypoint_optimizer.h:158-202).
    Returns (src, dst, weight) lists of residual blocks, in the reference's insertion order.
    Native host code (pxr_ka_build_edges), like the reference's."""
    lib = _lib.load()
    node_image, src, dst, sim = base._flat_graph(graph)
    n = len(node_image)
    node_feature = np.array([nd.feature_idx for nd in graph.nodes], dtype=np.int32)
    tl = np.ascontiguousarray(track_labels, dtype=np.int64)
    rl = np.ascontiguousarray(np.asarray(root_labels, dtype=bool), dtype=np.uint8)
    if len(tl) != n or len(rl) != n:
        raise ValueError("label arrays must have one entry per graph node")
    sub = None if nodes_in_problem is None else np.ascontiguousarray(list(nodes_in_problem), dtype=np.int64)
    cap = max(1, 3 * len(src))
    o_src, o_dst, o_w = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.float64)
    m = C.c_int64()
    _lib.check(lib.pxr_ka_build_edges(n, node_image.ctypes.data, node_feature.ctypes.data, len(src), src.ctypes.data,
                                      dst.ctypes.data, sim.ctypes.data, tl.ctypes.data, rl.ctypes.data,
                                      None if sub is None else sub.ctypes.data, 0 if sub is None else len(sub),
                                      int(bool(weight_by_sim)), int(bool(root_edges_only)), float(root_regularize_weight),
                                      o_src.ctypes.data, o_dst.ctypes.data, o_w.ctypes.data, C.byref(m)),
               "pxr_ka_build_edges")
    k = m.value
    return o_src[:k].tolist(), o_dst[:k].tolist(), o_w[:k].tolist()


def node_roles(setup, graph, src, dst, nodes_in_problem=None):
    """KeypointOptimizerBase::ParameterizeKeypoints (keypoint_optimizer.h:110-157) as per-node codes for pxr_ka_view.d_node_const:
    1 = held constant (KeypointAdjustmentSetup), 0 = variable inside its box bounds, 2 = variable WITHOUT bounds.
    RunSubset enumerates the out-matches of nodes_in_problem only (topological_keypoint_optimizer.h:108-113) but a match's
    destination becomes a parameter block wherever it lies; ParameterizeKeypoints visits nodes_in_problem only, so such a
    keypoint is neither held constant nor boxed.  Also returns the mask of the nodes that take part in this solve."""
    n = len(graph.nodes)
    node_const = np.array([setup.is_node_constant(nd) for nd in graph.nodes], np.uint8)
    if nodes_in_problem is None:
        return node_const, np.ones(n, bool)
    inside = np.zeros(n, bool)
    inside[np.fromiter(nodes_in_problem, dtype=np.int64)] = True
    touched = np.zeros(n, bool)
    touched[np.asarray(src, dtype=np.int64)] = True
    touched[np.asarray(dst, dtype=np.int64)] = True
    node_const[touched & ~inside] = 2
    return node_const, inside | touched


class FeatureMetricKeypointOptimizer:
    """_keypoint_adjustment.FeatureMetricKeypointOptimizer (bindings.cc:77-81):
    ctor (options, setup, interpolation_config); run(problem_labels, keypoints, graph, track_labels,
    root_labels, feature_set) or run(keypoints, graph, track_labels, root_labels, feature_set)."""

    option_defaults = {
        'loss': {'name': 'cauchy', 'params': [0.25]},
        'solver': {**base.solver_default_conf, 'parameter_tolerance': 1.0e-4, 'num_threads': 1, 'callbacks': []},
        'print_summary': True, 'bound': -1.0, 'num_threads': -1,      # KeypointOptimizerOptions' own defaults (keypoint_adjustment_options.h:46-80); KeypointAdjuster.default_conf passes bound 4
        'root_regularize_weight': -1.0, 'weight_by_sim': True, 'root_edges_only': False,
    }

    def __init__(self, options=None, setup=None, interpolation_config=None, ctx=None):
        self.options = base.merge_conf(self.option_defaults, options)
        self.setup = setup if setup is not None else KeypointAdjustmentSetup()
        ic = interpolation_config
        self.interpolation = ic if isinstance(ic, base.InterpolationConfig) else base.InterpolationConfig(ic)
        self.ctx = ctx
        self._summary = None
        self._used = False

    def _run(self, problem_labels, keypoints, graph, track_labels, root_labels, feature_set, nodes_in_problem=None):
        if keypoints is None:
            raise ValueError("keypoints cannot be NULL.")                       # topological_keypoint_optimizer.h:73-74
        if self._used and not getattr(self, "_resolving", False):
            raise ValueError("Cannot use the same KeypointOptimizer multiple times")   # :75-77
        self._used = True
        self._last = (problem_labels, keypoints, graph, track_labels, root_labels, feature_set, nodes_in_problem)
        n = len(graph.nodes)
        if len(track_labels) != n or len(root_labels) != n or (problem_labels is not None and len(problem_labels) != n):
            raise ValueError("label arrays must have one entry per graph node")  # THROW_CHECK_EQ, featuremetric_keypoint_optimizer.h:82-85
        ctx = self.ctx or default_context()
        o = self.options
        from ._timing import phase
        with phase("build_edges"):
            src, dst, w = build_edges(graph, keypoints, track_labels, root_labels, nodes_in_problem, o['weight_by_sim'],
                                      o['root_edges_only'], o['root_regularize_weight'])
        t_dump = phase("dump")
        t_dump.__enter__()
        names = [graph.image_id_to_name[nd.image_id] for nd in graph.nodes]
        kp = np.array([keypoints[nm][nd.feature_idx] for nm, nd in zip(names, graph.nodes)], dtype=np.float64).reshape(-1, 2)
        patches = [feature_set.fmap(nm).fpatch(nd.feature_idx) for nm, nd in zip(names, graph.nodes)]
        labels = np.zeros(n, np.int32) if problem_labels is None else np.asarray(problem_labels, dtype=np.int32)
        node_const, in_solve = node_roles(self.setup, graph, src, dst, nodes_in_problem)
        if nodes_in_problem is not None:
            labels = np.where(in_solve, 0, -1).astype(np.int32)
        prob = dict(kp=kp, node_patch=np.arange(n, dtype=np.int64), node_const=node_const,
                    node_problem=labels, edge_src=np.array(src, np.int32), edge_dst=np.array(dst, np.int32),
                    edge_w=np.array(w, np.float64),
                    # a label group of many keypoints (max_kps_per_problem = 1000 of configs/low_memory.yaml, or ONE problem with
                    # split_in_subproblems = false) is solved as chunks of whole tracks on several workgroups that share its trust
                    # region (ka_engine.chunk_label_groups, pxr_ka_view.d_prob_group): the tracks say where it may be cut
                    node_track=np.asarray(track_labels, dtype=np.int64))
        s = o['solver']
        lm = lm_options(max_iterations=s['max_num_iterations'], function_tolerance=s['function_tolerance'],
                        gradient_tolerance=s['gradient_tolerance'], parameter_tolerance=s['parameter_tolerance'],
                        max_consecutive_invalid_steps=s['max_num_consecutive_invalid_steps'])
        cfg, loss = self.interpolation.to_engine(), make_loss(o['loss']['name'], o['loss']['params'])
        # several ranks (torch.distributed initialised, one process per GPU): the sub-problems are dealt to the ranks
        # -- the reference's ParallelOptimizer over label groups (base/src/parallel_optimizer.h:77-211) with GPUs instead
        # of threads; every rank uploads only ITS patches and ends up with all refined keypoints
        from .. import parallel
        rank, world = parallel.world()
        shard, node_ids = (prob, np.arange(n)) if world == 1 else parallel.shard_ka_problem(prob, rank, world)
        channels = patches[0].shape[2]
        total = dict(iterations=0, num_successful=0, termination=0, initial_cost=0.0, final_cost=0.0, total_ms=0.0)
        out = kp.copy()
        t_dump.__exit__(None, None, None)
        if len(node_ids):
            with phase("upload"):
                arena = features.to_arena(ctx, [patches[i] for i in shard["node_patch" if world == 1 else "patch_ids"]],
                                          cache=getattr(self, "arena_cache", None) if world == 1 else None)
            with phase("problem_to_device"):
                shard = dict(shard, node_patch=arena.index)
                ka = KAProblem(ctx, arena, shard)
            with phase("solve"):
                total, _ = ka.solve(cfg, loss, bound=o['bound'], options=lm)
                out[node_ids] = ka.keypoints()
            arena.close()
        if world > 1:
            owned = np.zeros(n); owned[node_ids] = 1.0
            rows = parallel.gather_rows(out[node_ids], node_ids, n)
            owned = parallel.allreduce_host(owned)
            out = np.where(owned[:, None] > 0, rows, kp)
            acc = parallel.allreduce_host(np.array([total["iterations"], total["num_successful"], total["initial_cost"],
                                                    total["final_cost"], total["total_ms"],
                                                    1.0 if total["termination"] == 2 else 0.0], dtype=np.float64))
            total = dict(total, iterations=int(acc[0]), num_successful=int(acc[1]), initial_cost=float(acc[2]),
                         final_cost=float(acc[3]), total_ms=float(acc[4]), termination=2 if acc[5] > 0 else total["termination"])
        with phase("write_back"):
            for nm, nd, xy in zip(names, graph.nodes, out):       # in place, like featuremetric_keypoint_optimizer.h:195-196
                keypoints[nm][nd.feature_idx] = xy
        self._summary = Summary(total, num_residuals=len(src) * channels)
        return True

    def run_subset(self, nodes_in_problem, keypoints, graph, track_labels, root_labels, feature_set):
        """RunSubset (featuremetric_keypoint_optimizer.h:116-137): one problem over the given node indices only
        (what ParallelOptimizer hands to each worker); the other keypoints are left untouched."""
        self._run(None, keypoints, graph, track_labels, root_labels, feature_set,
                  nodes_in_problem=sorted(int(i) for i in nodes_in_problem))
        return self._summary

    def run(self, *args):
        if len(args) == 6:
            return self._run(*args)
        if len(args) == 5:
            return self._run(None, *args)
        raise TypeError("run(problem_labels, keypoints, graph, track_labels, root_labels, feature_set) or "
                        "run(keypoints, graph, track_labels, root_labels, feature_set)")

    def summary(self):
        return self._summary

    def solve_problem(self):
        """SolveProblem (keypoint_optimizer.h:77-107, bindings.cc:32): solve the problem of the last run() / run_subset() again,
        starting from the keypoints as they are now."""
        if getattr(self, "_last", None) is None:
            raise ValueError("no problem has been set up: call run() first")
        self._resolving = True
        try:
            return self._run(*self._last)
        finally:
            self._resolving = False


class TopologicalReferenceKeypointOptimizer(FeatureMetricKeypointOptimizer):
    """Option preset (topological_reference_keypoint_optimizer.h:8-15): star graph to the root.  The three values are the
    DEFAULTS of its Options struct -- keys the caller sets explicitly still win, as they do through the pybind constructor."""

    def __init__(self, options=None, setup=None, interpolation_config=None, ctx=None):
        options = {'weight_by_sim': False, 'root_regularize_weight': 1.0, 'root_edges_only': True, **dict(options or {})}
        super().__init__(options, setup, interpolation_config, ctx)


class KeypointAdjuster:
    """pixsfm/keypoint_adjustment/main.py:60-137."""
    default_conf = {
        'strategy': 'featuremetric',
        'apply': True,
        'interpolation': base.interpolation_default_conf,
        'level_indices': None,
        'max_kps_per_problem': 50,
        'optimizer': {
            'loss': {'name': 'cauchy', 'params': [0.25]},
            'solver': {**base.solver_default_conf, 'parameter_tolerance': 1.0e-5, 'num_threads': 1},
            'print_summary': False,
            'bound': 4.0,
            'num_threads': -1,
        },
        'split_in_subproblems': True,
    }

    @classmethod
    def create(cls, conf):
        strategy_to_solver = {"featuremetric": FeatureMetricKeypointAdjuster,
                              "topological_reference": TopologicalReferenceKeypointAdjuster}
        strategy = conf["strategy"] if "strategy" in conf else cls.default_conf["strategy"]
        return strategy_to_solver[strategy](conf)

    def refine(self, keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup=None):
        raise NotImplementedError

    def refine_multilevel(self, keypoints_dict, feature_manager, graph, track_labels=None, root_labels=None,
                          problem_setup=None):
        from ._timing import phase
        if track_labels is None and root_labels is None:
            # main.py:111-118 calls the three labellings one after the other on the host; here one call on the device
            # the solve runs on anyway (pxr_graph_labels_device: identical labels, scores and roots)
            with phase("labelling"):
                track_labels, _, root_labels = base.compute_labels_on_device(graph)
        if track_labels is None:
            track_labels = base.compute_track_labels(graph)
        if root_labels is None:
            score_labels = base.compute_score_labels(graph, track_labels)
            root_labels = base.compute_root_labels(graph, track_labels, score_labels)
        levels = self.conf['level_indices'] if self.conf['level_indices'] not in [None, "all"] else \
            list(reversed(range(feature_manager.num_levels)))
        from ._timing import gc_paused
        outputs = {}
        with gc_paused():
            for level_index in levels:
                out = self.refine(keypoints_dict, feature_manager.fset(level_index), graph, track_labels, root_labels,
                                  problem_setup=problem_setup)
                for k, v in out.items():
                    outputs.setdefault(k, []).append(v)
        return outputs

    _solver_cls = None

    def _refine(self, keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup):
        if problem_setup is None:
            problem_setup = KeypointAdjustmentSetup()
            problem_setup.set_masked_nodes_constant(graph, root_labels)          # main.py:175-177
        solver = self._solver_cls(deepcopy(self.conf['optimizer']), problem_setup, self.conf['interpolation'])
        from ._timing import phase
        from .. import parallel
        with features.SharedArena() as shared:
            if parallel.world()[1] == 1:             # (several ranks: every rank uploads its share only)
                # feature maps built by the reference's numpy constructor: their upload starts now, in a background thread, and runs
                # beside the edge construction and the walk over the keypoint / patch objects (features.SharedArena.prefetch)
                with phase("prefetch_start"):
                    seen = {}
                    for nd in graph.nodes:
                        seen.setdefault(nd.image_id, None)
                    shared.prefetch(solver.ctx or default_context(), feature_set, [graph.image_id_to_name[i] for i in seen])
                solver.arena_cache = shared
            if self.conf['split_in_subproblems']:
                problem_labels, _ = find_problem_labels(track_labels, self.conf['max_kps_per_problem'])
                solver.run(problem_labels, keypoints_dict, graph, track_labels, root_labels, feature_set)
            else:
                solver.run(keypoints_dict, graph, track_labels, root_labels, feature_set)
            solver.arena_cache = None
        return {"summary": solver.summary()}


class FeatureMetricKeypointAdjuster(KeypointAdjuster):
    """main.py:140-203."""
    default_conf = deepcopy(KeypointAdjuster.default_conf)
    default_conf["optimizer"] = {**default_conf["optimizer"], "root_regularize_weight": -1, "weight_by_sim": True,
                                 "root_edges_only": False, "num_threads": -1}
    _solver_cls = FeatureMetricKeypointOptimizer

    def __init__(self, conf):
        self.conf = base.merge_adjuster_conf(self.default_conf, conf)

    def refine(self, keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup=None):
        return self._refine(keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup)


class TopologicalReferenceKeypointAdjuster(KeypointAdjuster):
    """main.py:206-247."""
    default_conf = deepcopy(KeypointAdjuster.default_conf)
    default_conf["optimizer"] = {**default_conf["optimizer"], "num_threads": -1}
    _solver_cls = TopologicalReferenceKeypointOptimizer

    def __init__(self, conf):
        self.conf = base.merge_adjuster_conf(self.default_conf, conf)

    def refine(self, keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup=None):
        return self._refine(keypoints_dict, feature_set, graph, track_labels, root_labels, problem_setup)


def build_matching_graph(pairs, matches, scores=None):          # main.py:250-260
    graph = base.Graph()
    scores = scores if scores is not None else [None for _ in matches]
    for (name1, name2), m, s in zip(pairs, matches, scores):
        graph.register_matches(name1, name2, m, s)
    return graph
