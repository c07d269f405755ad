"""CPU: the pybind surface of the reference's module (pixsfm/*/bindings.cc) against the `_pixsfm`-shaped package.
  * live, when /root/reference is present: every class / function / attribute name the bindings define at module level exists
    in the matching sub-module, and every method / property bound on a class exists on an instance of the stand-in;
  * always: the semantics of the container helpers that are plain host code (FeaturePatch.to_corner / slice / get_entry,
    FeatureMap.shape, Graph.add_node / degrees / scores / edges, ...), restated from the sources cited in the classes."""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/pixsfm"
BINDINGS = {"_base": "base/bindings.cc", "_features": "features/bindings.cc", "_keypoint_adjustment": "keypoint_adjustment/bindings.cc",
            "_bundle_adjustment": "bundle_adjustment/bindings.cc", "_localization": "localization/bindings.cc",
            "_residuals": "residuals/bindings.cc", "_util": "util/bindings.cc"}
DEF = r'\.def(?:_readwrite|_readonly|_property|_property_readonly|_static)?\(\s*"([A-Za-z_0-9]+)"'
# `problem` is the live ceres::Problem of the reference's optimizers: there is no Ceres object to hand out here
NOT_OFFERED = {"problem"}


def _instances():
    from pixsfm_amd._pixsfm import _bundle_adjustment as ba, _features as ft, _keypoint_adjustment as ka, _localization as loc
    from pixsfm_amd.api import base, features
    from pixsfm_amd.api.bundle_adjustment import FeatureView
    from pixsfm_amd.api.reconstruction import Reconstruction
    patch = features.FeaturePatch(np.zeros((4, 4, 8), np.float16), (0, 0), (1.0, 1.0))
    fmap = features.FeatureMap({0: patch})
    fset = features.FeatureSet({"a.jpg": fmap})
    setup = ba.BundleAdjustmentSetup()
    return {"Graph": base.Graph(), "FeatureNode": base.FeatureNode(0, 0), "Match": base.Match(0, 1.0), "InterpolationConfig": base.InterpolationConfig(),
            "FeaturePatch": patch, "FeatureMap": fmap, "FeatureSet": fset, "FeatureView": FeatureView(fset, Reconstruction()),
            "FeatureManager": features.FeatureManager([fset]), "Reference": features.Reference(0, 0, np.zeros(8)),
            "PatchInterpolator": features.PatchInterpolator({}), "PatchStatus": ft.PatchStatus(),
            "KeypointAdjustmentSetup": ka.KeypointAdjustmentSetup(), "FeatureMetricKeypointOptimizer": ka.FeatureMetricKeypointOptimizer({}, None, None),
            "TopologicalReferenceKeypointOptimizer": ka.TopologicalReferenceKeypointOptimizer({}, None, None),
            "BundleAdjustmentSetup": setup, "FeatureReferenceBundleOptimizer": ba.FeatureReferenceBundleOptimizer({}, setup, None),
            "CostMapBundleOptimizer": ba.CostMapBundleOptimizer({}, setup, None), "ReferenceExtractor": ba.ReferenceExtractor({}, None),
            "CostMapExtractor": ba.CostMapExtractor({}, None), "QueryKeypointOptimizer": loc.QueryKeypointOptimizer({}, None),
            "QueryBundleOptimizer": loc.QueryBundleOptimizer({}, None)}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_every_bound_name_exists_in_the_shim():
    import importlib
    inst = _instances()
    problems = []
    for mod_name, rel in BINDINGS.items():
        src = re.sub(r"//[^\n]*", "", open(os.path.join(REF, rel)).read())          # bindings that are commented out do not count
        mod = importlib.import_module("pixsfm_amd._pixsfm." + mod_name)
        names = set(re.findall(r'py::(?:class|enum)_<[^;]*?>\(\s*\w+\s*,\s*\(?\s*"([A-Za-z_0-9]+)"', src, re.S))
        names |= set(re.findall(r'\bm\.def\(\s*"([A-Za-z_0-9]+)"', src)) | set(re.findall(r'\bm\.attr\(\s*"([A-Za-z_0-9]+)"', src))
        problems += ["%s.%s" % (mod_name, n) for n in sorted(names) if not hasattr(mod, n)]
        # methods / properties: classes named by a literal, and the template binders of the feature containers and optimizers
        for m in re.finditer(r'py::class_<[^;]*?>\(\s*\w+\s*,\s*\(?\s*"([A-Za-z_0-9]+)"([^;]*);', src, re.S):
            if m.group(1) in inst:
                problems += ["%s.%s.%s" % (mod_name, m.group(1), d) for d in sorted(set(re.findall(DEF, m.group(2))))
                             if not d.startswith("__") and d not in NOT_OFFERED and not hasattr(inst[m.group(1)], d)]
        for m in re.finditer(r'py::class_<[^;]*?>\(\s*m\s*,\s*\(\s*"([A-Za-z]+)"\s*\+\s*type_suffix\)[^;]*?\)([^;]*);', src, re.S):
            if m.group(1) in inst:
                problems += ["%s.%s.%s" % (mod_name, m.group(1), d) for d in sorted(set(re.findall(DEF, m.group(2))))
                             if not d.startswith("__") and not hasattr(inst[m.group(1)], d)]
        for m in re.finditer(r'void Bind(KeypointOptimizer|ParallelSolve|BundleOptimizer|SingleQueryKeypointOptimizer|SingleQueryBundleOptimizer)\([^)]*\)\s*\{(.*?)\n\}', src, re.S):
            targets = {"KeypointOptimizer": ["FeatureMetricKeypointOptimizer"], "ParallelSolve": ["FeatureMetricKeypointOptimizer"],
                       "BundleOptimizer": ["FeatureReferenceBundleOptimizer", "CostMapBundleOptimizer"],
                       "SingleQueryKeypointOptimizer": ["QueryKeypointOptimizer"], "SingleQueryBundleOptimizer": ["QueryBundleOptimizer"]}[m.group(1)]
            for t in targets:
                problems += ["%s.%s.%s" % (mod_name, t, d) for d in sorted(set(re.findall(DEF, m.group(2))))
                             if d not in NOT_OFFERED and not hasattr(inst[t], d)]
    assert not problems, "\n".join(problems)


def test_feature_patch_helpers():
    from pixsfm_amd.api import features
    rng = np.random.default_rng(0)
    data = rng.normal(size=(20, 30, 4)).astype(np.float32)
    p = features.FeaturePatch(data, (0, 0), (0.5, 0.25))
    assert (p.height, p.width, p.channels, p.size) == (20, 30, 4, 2400) and p.num_bytes() == 9600 == p.current_memory()
    assert p.get_entry(3, 7, 2) == data[3, 7, 2] and p.is_reference() and p.has_data() and p.data_ptr() == p.data.ctypes.data
    assert p.status.is_locked and p.status.reference_count == 1 and p.flush() == 0 and p.upsampling_factor == 1.0
    xy = np.array([33.0, 41.0])                                   # image coordinates
    uv = xy * p.scale - 0.5                                        # featurepatch.h:250-255, corner (0, 0)
    assert np.allclose(p.get_pixel_coords(xy), uv)
    # to_corner: trunc(uv - ps / 2), clamped into the patch (featurepatch.cc:322-334)
    assert np.array_equal(p.to_corner(xy, 8), [int(uv[0] - 4), int(uv[1] - 4)])
    assert np.array_equal(p.to_corner([0.0, 0.0], 8), [0, 0]) and np.array_equal(p.to_corner([1000.0, 1000.0], 8), [30 - 8, 20 - 8])
    s = p.slice(xy, 8)
    c = p.to_corner(xy, 8)
    assert s.shape == (8, 8, 4) and np.array_equal(s.corner, c) and np.array_equal(s.scale, p.scale)
    assert np.array_equal(s.data, data[c[1]:c[1] + 8, c[0]:c[0] + 8])
    with pytest.raises(ValueError):
        p.slice(xy, 25)


def test_feature_container_helpers():
    from pixsfm_amd.api import features
    a = features.FeaturePatch(np.zeros((4, 6, 8), np.float16), (1, 2), (1.0, 1.0))
    b = features.FeaturePatch(np.ones((4, 6, 8), np.float16), (3, 4), (1.0, 1.0))
    fm = features.FeatureMap()
    assert fm.channels == -1 and fm.shape() == [0, 0, 0, -1]
    fm.add_fpatch(7, a); fm.add_fpatch(9, b)
    assert fm.num_fpatches() == 2 and fm.shape() == [2, 4, 6, 8] and fm.channels == 8 and fm.size == 2 * 192
    assert fm.fpatches is fm.patches and fm.num_bytes() == 2 * 192 * 2 == fm.current_memory() and fm.flush() == 0
    fs = features.FeatureSet(channels=8)
    fs.add_fmap("a.jpg", fm)
    other = features.FeatureMap({1: a})
    fs.emplace("a.jpg", other)                                    # emplace keeps an existing entry
    assert fs.fmap("a.jpg") is fm and fs.keys() == ["a.jpg"] and fs.num_bytes() == fm.num_bytes() and fs.flush() == 0
    fs.flush_every_n(3); fs.use_parallel_io(True); fs.lock()
    mgr = features.FeatureManager([fs])
    assert mgr.fsets() is mgr.fsets and mgr.fsets[0] is fs and mgr.num_bytes() == fs.num_bytes() == mgr.current_memory()
    r = features.Reference(1, 2, np.zeros(8))
    assert (r.channels, r.n_nodes, r.costs, r.track) == (8, 1, [], []) and not r.has_observations()


def test_graph_helpers():
    from pixsfm_amd.api import base
    g = base.Graph()
    n0 = g.find_or_create_node("a.jpg", 3)
    n1 = g.find_or_create_node("b.jpg", 5)
    g.add_edge(n0, n1, 0.5)
    g.add_edge(n1, n0, 0.25)
    k = g.add_node("c.jpg", 1)                                     # graph.cc:105-113: registers the image, not the lookup
    assert k == 2 and g.image_name_to_id["c.jpg"] == 2 and (2, 1) not in g.node_map
    assert g.find_or_create_node("c.jpg", 1).node_idx == 3         # ... so the same (image, keypoint) gets a second node
    assert g.add_node(0, 9) == 4 and g.nodes[4].image_id == 0
    g.add_edge(g.nodes[2], n0, 2.0)
    assert g.degrees() == [3, 2, 1, 0, 0]                          # out-degree + in-degree (graph.cc:5-14)
    assert g.scores() == [2.75, 0.75, 2.0, 0.0, 0.0] and np.array_equal(g.get_scores(), g.scores())
    assert g.edges() == [(0, 1, 0.5), (1, 0, 0.25), (2, 0, 2.0)]
    m = base.Match(node_idx=1, similarity=0.5)
    m.similarity = 0.75
    assert m.sim == 0.75


def test_option_structs_behave_like_make_dataclass():
    """helpers.h:147-290: dict / kwargs constructors, struct defaults for what is left out, attribute access, mergedict()
    refusing unknown fields, summary()."""
    from pixsfm_amd._pixsfm import _bundle_adjustment as ba, _keypoint_adjustment as ka, _localization as loc
    o = ka.KeypointOptimizerOptions()
    assert o.bound == -1.0 and o.print_summary is True and o.solver["parameter_tolerance"] == 1e-4 and "weight_by_sim" not in o
    o.bound = 2.5
    assert o["bound"] == 2.5
    with pytest.raises(AttributeError):
        o.no_such_field = 1
    with pytest.raises(AttributeError):
        ka.KeypointOptimizerOptions({"boundd": 3})
    f = ka.FeatureMetricKeypointOptimizerOptions({"bound": 4.0, "solver": {"max_num_iterations": 7}}, weight_by_sim=False)
    assert (f.bound, f.weight_by_sim, f.root_edges_only, f.num_threads) == (4.0, False, False, -1)
    assert f.solver["max_num_iterations"] == 7 and f.solver["parameter_tolerance"] == 1e-4            # nested merge keeps the rest
    t = ka.TopologicalReferenceKeypointOptimizerOptions(root_edges_only=False)
    assert (t.weight_by_sim, t.root_regularize_weight, t.root_edges_only) == (False, 1.0, False) and isinstance(t, ka.FeatureMetricKeypointOptimizerOptions)
    b = ba.BundleOptimizerOptions(refine_focal_length=False)
    assert (b.refine_focal_length, b.refine_extra_params, b.min_track_length, b.loss["name"]) == (False, True, -1, "cauchy")
    assert ba.ReferenceConfig().iters == 10 and ba.ReferenceConfig(iters=100).iters == 100
    c = ba.CostMapConfig()
    assert c.get_effective_channels() == 3
    c.compute_cross_derivative = True
    assert c.get_effective_channels() == 4 and ba.CostMapConfig(as_gradientfield=False).get_effective_channels() == 1
    assert loc.QueryKeypointOptimizerOptions().bound == -1.0 and loc.QueryBundleOptimizerOptions().solver["parameter_tolerance"] == 1e-5
    # two instances do not share their nested dicts
    x, y = ba.BundleOptimizerOptions(), ba.BundleOptimizerOptions()
    x.solver["max_num_iterations"] = 3
    assert y.solver["max_num_iterations"] == 100
    assert "BundleOptimizerOptions:" in x.summary(True) and "refine_focal_length" in x.summary(False)
    # the optimizers take the structs as they take dicts
    opt = ka.FeatureMetricKeypointOptimizer(f, None, None)
    assert opt.options["bound"] == 4.0 and opt.options["solver"]["max_num_iterations"] == 7


def test_containers_accept_the_pybind_constructor_forms():
    """How the reference's own Python builds them: FeatureManager(channels_per_level, dtype_array) + fset(l).emplace(name,
    FeatureMap(patches, keypoint_ids, corners, metadata)) (extract.py:95-139), FeaturePatch(inarray, offset, scale, do_copy)
    (features/bindings.cc:47-53), Reference(descriptor=, observations=, costs=, source=, track=) (store_references.py:45-53)."""
    from pixsfm_amd._pixsfm import _features as ft
    from pixsfm_amd.api.reconstruction import Track, TrackElement
    rng = np.random.default_rng(0)
    patches = rng.normal(size=(3, 4, 4, 8)).astype(np.float16)
    corners = np.array([[1, 2], [3, 4], [5, 6]])
    meta = {"is_sparse": True, "scale": np.array([0.5, 0.25]), "patch_size": 4}
    mgr = ft.FeatureManager([8, 8], np.array([0], np.float16))
    assert mgr.num_levels == 2 and mgr.fset(1).channels == 8 and mgr.fset(0).keys() == []
    mgr.fset(0).emplace("a.jpg", ft.FeatureMap(patches, [10, 20, 30], corners, meta))
    fm = mgr.fset(0).fmap("a.jpg")
    assert fm.is_sparse and sorted(fm.keys()) == [10, 20, 30] and fm.shape() == [3, 4, 4, 8]
    p = fm.fpatch(20)
    assert np.array_equal(p.data, patches[1]) and np.array_equal(p.corner, [3, 4]) and np.array_equal(p.scale, [0.5, 0.25])
    assert np.shares_memory(p.data, patches)                      # a view, like the reference's numpy-backed patches
    dense = ft.FeatureMap(patches[:1], [7], corners[:1], {"is_sparse": False, "scale": np.ones(2)})
    assert not dense.is_sparse and dense.keys() == [ft.kDenseId] and dense.fpatch(123) is dense.fpatch(0)
    with pytest.raises(ValueError):
        ft.FeatureMap(patches, [1, 2, 3], corners, {"is_sparse": False, "scale": np.ones(2)})
    q = ft.FeaturePatch(inarray=patches[0], offset=np.array([9, 8]), scale=np.array([2.0, 2.0]), do_copy=True)
    assert np.array_equal(q.corner, [9, 8]) and not np.shares_memory(q.data, patches)
    fs = ft.FeatureSet(feature_dict={"a.jpg": fm}, channels=8)
    assert fs.has_fmap("a.jpg") and ft.FeatureSet(8).channels == 8
    r = ft.Reference(descriptor=np.ones((1, 8)), observations=[np.zeros((1, 8)), np.ones((1, 8))], costs=[8.0, 0.0],
                     source=TrackElement(3, 5), track=Track([TrackElement(3, 5), TrackElement(4, 1)]))
    assert (r.source.image_id, r.source.point2D_idx) == (3, 5) and r.source == (3, 5) and r.has_observations()
    assert [(e.image_id, e.point2D_idx) for e in r.track.elements] == [(3, 5), (4, 1)] and r.track.length() == 2 and r.costs == [8.0, 0.0]
    assert ft.Reference().channels == 0 and not ft.Reference().has_observations()
