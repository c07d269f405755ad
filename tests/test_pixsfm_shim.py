"""The `_pixsfm`-shaped adapter (pixsfm_amd/_pixsfm) against the REFERENCE's own, unmodified L4 Python: with the
adapter registered as `pixsfm._pixsfm`, pixsfm/keypoint_adjustment/main.py and pixsfm/bundle_adjustment/main.py are
imported from /root/reference and driven through `KeypointAdjuster.create(conf).refine_multilevel(...)` /
`BundleAdjuster.create(conf).refine_multilevel(...)`.  omegaconf / pyceres / pycolmap are absent in this image, so
minimal stand-ins are registered for them (attribute-access config dicts, the callback list type, our
pycolmap-shaped Reconstruction).  Without a GPU the run must get as far as the optimiser's solve and fail LOUDLY
there (no CPU fallback); the adapter's surface itself is checked against the names the reference's modules import.
Runs in the build container only -- /root/reference is not shipped to the GPU box."""
import importlib
import logging
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/pixsfm"


class _Cfg(dict):
    """OmegaConf stand-in: nested dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return _Cfg({k: _Cfg.wrap(v) for k, v in d.items()})
        if isinstance(d, (list, tuple)):
            return [_Cfg.wrap(v) for v in d]
        return d

    @staticmethod
    def unwrap(d):
        if isinstance(d, dict):
            return {k: _Cfg.unwrap(v) for k, v in d.items()}
        if isinstance(d, list):
            return [_Cfg.unwrap(v) for v in d]
        return d


def _merge(a, b):
    out = dict(a)
    for k, v in (b or {}).items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


@pytest.fixture()
def reference_pixsfm():
    if not os.path.isdir(REF):
        pytest.skip("reference tree absent")
    saved = dict(sys.modules)
    import pixsfm_amd._pixsfm as shim
    om = types.ModuleType("omegaconf")

    class OmegaConf:
        @staticmethod
        def create(d=None):
            return _Cfg.wrap(d or {})

        @staticmethod
        def merge(*confs):
            acc = {}
            for c in confs:
                acc = _merge(acc, _Cfg.unwrap(c))
            return _Cfg.wrap(acc)

        @staticmethod
        def to_container(cfg, resolve=False):
            return _Cfg.unwrap(cfg)

        @staticmethod
        def resolve(cfg):
            pass
    om.OmegaConf, om.DictConfig = OmegaConf, _Cfg
    om_sub = types.ModuleType("omegaconf.omegaconf")
    om_sub.OmegaConf, om_sub.DictConfig = OmegaConf, _Cfg
    om.omegaconf = om_sub
    sys.modules["omegaconf.omegaconf"] = om_sub
    pyceres = types.ModuleType("pyceres")
    pyceres.ListIterationCallback = list
    pyceres.IterationCallback = object
    from pixsfm_amd.api import reconstruction as rec_mod
    pycolmap = types.ModuleType("pycolmap")
    pycolmap.Reconstruction = rec_mod.Reconstruction
    pkg = types.ModuleType("pixsfm")
    pkg.__path__ = [REF]
    pkg.logger = logging.getLogger("pixsfm-test")
    sys.modules.update({"omegaconf": om, "pyceres": pyceres, "pycolmap": pycolmap, "pixsfm": pkg})
    shim.install_as("pixsfm._pixsfm")
    # pixsfm/features/__init__.py also pulls in the CNN extractors (torch models, h5py writers): out of scope here,
    # so `pixsfm.features` is the native half of that package only
    feats = types.ModuleType("pixsfm.features")
    feats.__dict__.update({k: v for k, v in vars(shim._features).items() if not k.startswith("__")})
    sys.modules["pixsfm.features"] = feats
    pkg.features = feats
    yield importlib.import_module
    for k in list(sys.modules):                 # only what this fixture registered (never torch & co.)
        if k == "pixsfm" or k.startswith("pixsfm.") or k in ("omegaconf", "omegaconf.omegaconf", "pyceres", "pycolmap"):
            if k in saved:
                sys.modules[k] = saved[k]
            else:
                del sys.modules[k]


def test_adapter_exports_what_the_reference_modules_import():
    import pixsfm_amd._pixsfm as shim
    want = {
        "_base": ["Graph", "FeatureNode", "Match", "InterpolationConfig", "InterpolatorType", "Map_NameKeypoints",
                  "compute_track_labels", "compute_score_labels", "compute_root_labels", "count_track_edges", "count_edges_AB"],
        "_features": ["FeaturePatch", "FeatureMap", "FeatureSet", "FeatureView", "FeatureManager", "Reference",
                      "PatchInterpolator", "kDenseId", "FeatureSet_f16", "FeatureManager_f32"],
        "_keypoint_adjustment": ["FeatureMetricKeypointOptimizer", "TopologicalReferenceKeypointOptimizer",
                                 "KeypointAdjustmentSetup", "KeypointOptimizerOptions"],
        "_bundle_adjustment": ["FeatureReferenceBundleOptimizer", "ReferenceExtractor", "BundleAdjustmentSetup",
                               "CostMapExtractor", "CostMapBundleOptimizer", "BundleOptimizerOptions",
                               "PatchWarpBundleOptimizer", "GeometricBundleOptimizer"],
        "_localization": ["QueryKeypointOptimizer", "QueryBundleOptimizer", "find_nearest_references"],
        "_util": ["free_memory", "total_memory", "used_memory"],
    }
    for sub, names in want.items():
        for n in names:
            assert hasattr(getattr(shim, sub), n), (sub, n)
    with pytest.raises(NotImplementedError):
        shim._bundle_adjustment.PatchWarpBundleOptimizer({}, None, {})
    assert shim._util.total_memory() >= shim._util.free_memory() > 0


def _ka_inputs():
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.api import features
    prob = synthetic_ka.make_ka_problem(n_tracks=6, track_len=4, seed=3, directed_both=False, channels=64)
    n, tl = 24, 4
    img, kid = np.arange(n) % tl, np.arange(n) // tl
    names = ["im%d" % k for k in range(tl)]
    keypoints = {names[k]: prob["kp"][img == k].copy() for k in range(tl)}
    fmaps = {names[k]: features.FeatureMap.from_arrays(prob["patches"][img == k], kid[img == k], prob["corners"][img == k],
                                                       (1.0, 1.0)) for k in range(tl)}
    pairs, matches, scores = [], [], []
    for a in range(tl):
        for b in range(a + 1, tl):
            sel = (img[prob["edge_src"]] == a) & (img[prob["edge_dst"]] == b)
            pairs.append((names[a], names[b]))
            matches.append(np.stack([kid[prob["edge_src"][sel]], kid[prob["edge_dst"][sel]]], 1))
            scores.append(prob["edge_w"][sel])
    return keypoints, features.FeatureManager([features.FeatureSet(fmaps)]), pairs, matches, scores


def test_reference_keypoint_adjuster_runs_on_the_adapter(reference_pixsfm):
    import torch
    from pixsfm_amd import PixsfmHipError
    ka_main = reference_pixsfm("pixsfm.keypoint_adjustment.main")          # the reference's file, unmodified
    assert ka_main.__file__.startswith(REF)
    from omegaconf import OmegaConf
    keypoints, fmanager, pairs, matches, scores = _ka_inputs()
    graph = ka_main.build_matching_graph(pairs, matches, scores)           # reference code driving our Graph
    assert len(graph.nodes) == 24
    for strategy in ("featuremetric", "topological_reference"):
        adjuster = ka_main.KeypointAdjuster.create(OmegaConf.create({"strategy": strategy, "optimizer": {"bound": 3.0}}))
        assert adjuster.conf.optimizer.bound == 3.0 and adjuster.conf.max_kps_per_problem == 50
        kp = {k: v.copy() for k, v in keypoints.items()}
        if torch.cuda.is_available():
            out = adjuster.refine_multilevel(kp, fmanager, graph)
            assert out["summary"][0].final_cost < out["summary"][0].initial_cost
        else:
            # everything up to the solve is the reference's Python + our host code; the solve needs the GPU and says so
            with pytest.raises(PixsfmHipError, match="pxr_ctx_create"):
                adjuster.refine_multilevel(kp, fmanager, graph)
    # the reference's packing function and ours agree on this graph's tracks
    labels = sys.modules["pixsfm._pixsfm"]._base.compute_track_labels(graph)
    from pixsfm_amd.ka_engine import pack_tracks_into_problems
    assert list(ka_main.find_problem_labels(labels, 50)[0]) == list(pack_tracks_into_problems(labels, 50)[0])


def test_reference_bundle_adjuster_runs_on_the_adapter(reference_pixsfm):
    import torch
    from pixsfm_amd import PixsfmHipError, synthetic
    from pixsfm_amd.api import features
    from pixsfm_amd.api.reconstruction import reconstruction_from_flat
    ba_main = reference_pixsfm("pixsfm.bundle_adjustment.main")            # the reference's file, unmodified
    assert ba_main.__file__.startswith(REF)
    from omegaconf import OmegaConf
    prob = synthetic.make_ba_problem(n_cams=4, n_points=30, obs_per_point=3, seed=2, channels=64)
    rec, patch_of = reconstruction_from_flat(prob)
    fmaps = {}
    for (image_id, p2d), pi in patch_of.items():
        fm = fmaps.setdefault(rec.images[image_id].name, features.FeatureMap())
        fm.patches[p2d] = features.FeaturePatch(prob["patches"][pi], prob["corners"][pi], prob["scales"][pi])
    fmanager = features.FeatureManager([features.FeatureSet(fmaps)])
    setup = ba_main.default_problem_setup(rec)                              # reference code on our BundleAdjustmentSetup
    assert setup.has_constant_pose(rec.reg_image_ids()[0]) and setup.has_constant_tvec(rec.reg_image_ids()[1])
    assert ba_main.find_problem_labels(rec, 10)[rec.point3D_ids()[-1]] == rec.point3D_ids()[-1] // 10
    adjuster = ba_main.BundleAdjuster.create(OmegaConf.create({"optimizer": {"solver": {"max_num_iterations": 5}}}))
    assert type(adjuster).__name__ == "FeatureReferenceBundleAdjuster"
    assert adjuster.conf.optimizer.solver.use_inner_iterations is True
    if torch.cuda.is_available():
        out = adjuster.refine_multilevel(rec, fmanager)
        assert out["summary"][0].final_cost < out["summary"][0].initial_cost and len(out["references"][0]) == 30
    else:
        with pytest.raises(PixsfmHipError, match="pxr_ctx_create"):       # ReferenceExtractor.run is the first device call
            adjuster.refine_multilevel(rec, fmanager)
    with pytest.raises(NotImplementedError):
        ba_main.BundleAdjuster.create(OmegaConf.create({"strategy": "geometric"})).refine(rec, fmanager.fset(0))



def test_reference_query_adjusters_run_on_the_adapter(reference_pixsfm):
    """pixsfm/localization/main.py, unmodified: QueryKeypointAdjuster / QueryBundleAdjuster construct the adapter's optimizers
    from their own default_conf (every key of it must be a field of the option structs) and drive them; the file's pure-Python
    helpers run as they are.  The CNN extractor, the cache loader and the config lookup it imports are outside the path
    (stubbed); QueryLocalizer's PnP needs pycolmap."""
    import torch
    from pixsfm_amd import PixsfmHipError, synthetic, synthetic_ka
    from pixsfm_amd.api import features
    for name, attrs in (("pixsfm.features.extractor", {"FeatureExtractor": type("FeatureExtractor", (), {"default_conf": {}})}),
                        ("pixsfm.extract", {"features_from_reconstruction": None, "load_features_from_cache": None}),
                        ("pixsfm.configs", {"parse_config_path": lambda p: p})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    sys.modules["pixsfm.features"].FeatureManager = features.FeatureManager
    sys.modules["pycolmap"].Camera = object
    loc_main = reference_pixsfm("pixsfm.localization.main")                 # the reference's file, unmodified
    assert loc_main.__file__.startswith(REF)
    # pure-Python helpers of the file
    assert loc_main.find_unique_inliers([3, 3, 5, 3], [True, False, True, True]) == [True, False, True, False]
    assert loc_main.find_unique_min_by_group([0.5, 0.2, 0.9, 0.1], [7, 7, 8, 8]) == [False, True, False, True]
    # QKA: one keypoint per track of a small KA scene, references = descriptors at the true positions
    base_ = synthetic_ka.make_ka_problem(n_tracks=8, track_len=2, seed=5, sigma=0.5, channels=128)
    q = np.arange(0, 16, 2)
    fmap = features.FeatureMap.from_arrays(base_["patches"][q], np.arange(8), base_["corners"][q], (1.0, 1.0))
    refs = [np.full((1, 128), 1.0 / np.sqrt(128.0)) for _ in q]
    qka = loc_main.QueryKeypointAdjuster({"optimizer": {"bound": 2.0}})
    assert qka.conf.optimizer.bound == 2.0 and qka.conf.optimizer.solver.parameter_tolerance == 1e-05
    assert type(qka.solver).__name__ == "QueryKeypointOptimizer" and qka.solver.options["bound"] == 2.0
    kp = base_["kp"][q].copy()
    # QBA: image 0 of a small BA scene
    full = synthetic.make_ba_problem(n_cams=3, n_points=30, obs_per_point=2, seed=6, model=2)
    sel = np.nonzero(full["obs_image"] == 0)[0]
    from pixsfm_amd.api.reconstruction import Camera
    cam = Camera(1, 2, 1000, 1000, full["cam_params"][full["image_camera"][0], :4].copy())
    qfmap = features.FeatureMap.from_arrays(full["patches"][sel], np.arange(len(sel)), full["corners"][sel], (1.0, 1.0))
    qba = loc_main.QueryBundleAdjuster({"optimizer": {"refine_focal_length": True}})
    assert qba.solver.options["refine_focal_length"] is True and qba.solver.options["loss"]["name"] == "cauchy"
    args = (full["qvec"][0].copy(), full["tvec"][0].copy(), cam, [full["gt_xyz"][p].copy() for p in full["obs_point"][sel]], qfmap,
            [full["refs"][p].copy() for p in full["obs_point"][sel]])
    if torch.cuda.is_available():
        qka.refine(kp, fmap, refs)
        assert qba.refine(*args)
    else:
        with pytest.raises(PixsfmHipError, match="pxr_ctx_create"):
            qka.refine(kp, fmap, refs)
        with pytest.raises(PixsfmHipError, match="pxr_ctx_create"):
            qba.refine(*args)
