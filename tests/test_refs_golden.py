"""Reference extraction against vectors produced by the REFERENCE's own ReferenceExtractor::RunSubset /
ComputeReference + RobustMeanIRLS compiled in place (tests/golden/make_golden_refs.py, oracle/ref_refs_shim.cc):
  * CPU: the oracle's restatement (pxo.compute_reference on pxo.ba_residual descriptors) reproduces the chosen observation,
    the descriptor handed to the BA (closest observation or robust mean) and the per-observation costs;
  * GPU: pxr_ba_compute_references through BAProblem.compute_references does;
  * live, when oracle/_ref/libpxo_ref_refs.so is present: the vectors are what the reference yields now.
Tolerance 1e-12 (descriptors are unit vectors; the stand-in for Eigen reduces left to right, real Eigen in packets)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden_refs", os.path.join(HERE, "golden", "make_golden_refs.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
NAMES = [s[0] for s in G.SCENES]
TOL = 1e-12


def _gold(name):
    z = np.load(os.path.join(HERE, "golden", "refs_ref.npz"))
    return {k.split("|", 1)[1]: z[k] for k in z.files if k.startswith(name + "|")}


def _scene(name):
    prob, opts, has = G.scene(name)
    gold = _gold(name)
    assert G.checksum(prob, has) == str(gold["checksum"]), "the synthetic scene generator changed: regenerate tests/golden/refs_ref.npz"
    return prob, opts, has, gold


def _visible(prob, has):
    keep = np.nonzero(has)[0]
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "obs_patch"):
        sub[k] = prob[k][keep]
    return sub, keep


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_reference_extraction(name):
    import pxo
    prob, opts, has, gold = _scene(name)
    cfg = pxo.cfg(l2_normalize=opts["l2_normalize"], use_float_simd=opts["use_float_simd"])
    ls = pxo.loss(opts["loss"][0], opts["loss"][1]) if opts["loss"][0] != "trivial" else pxo.loss("trivial")
    for p in range(len(prob["xyz"])):
        obs = np.nonzero((prob["obs_point"] == p) & has)[0]
        assert bool(gold["has_ref"][p]) == (len(obs) > 0)
        if len(obs) == 0:
            continue
        descs = []
        for i in obs:
            img = prob["obs_image"][i]
            cam = prob["image_camera"][img]
            q = prob["obs_patch"][i]
            patch = pxo.make_patch(prob["patches"][q], prob["corners"][q], prob["scales"][q])
            K = pxo.lib().pxo_camera_num_params(int(prob["cam_model"][cam]))
            f, *_ = pxo.ba_residual(patch, cfg, int(prob["cam_model"][cam]), prob["qvec"][img], prob["tvec"][img], prob["xyz"][p],
                                    prob["cam_params"][cam][:K], None, jac=False)
            descs.append(f)
        descs = np.array(descs)
        idx, ref, mean = pxo.compute_reference(descs, ls, opts["iters"], opts["l2_normalize"])
        assert obs[idx] == gold["src_obs"][p] and prob["obs_image"][obs[idx]] == gold["src_image"][p], (name, p)
        out = ref if opts["closest_to_robust_mean"] else mean
        assert np.abs(out - gold["descriptor"][p]).max() < TOL, (name, p)
        assert np.abs(((descs - mean) ** 2).sum(1) - gold["obs_cost"][obs]).max() < TOL, (name, p)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_the_reference_extraction(ctx, name):
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    prob, opts, has, gold = _scene(name)
    sub, keep = _visible(prob, has)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, sub)
    loss = make_loss(opts["loss"][0], [opts["loss"][1]] if opts["loss"][0] != "trivial" else [])
    chosen, mean = ba.compute_references(interp_cfg(l2_normalize=opts["l2_normalize"], use_float_simd=opts["use_float_simd"]), loss,
                                         iters=opts["iters"], keep_mean=True)
    refs = ba.d["refs"].download()
    ok = gold["has_ref"]
    assert np.array_equal(chosen >= 0, ok)
    assert np.array_equal(keep[chosen[ok]], gold["src_obs"][ok])
    out = refs if opts["closest_to_robust_mean"] else mean
    assert np.abs(out[ok] - gold["descriptor"][ok]).max() < 1e-10


def test_golden_vectors_are_what_the_reference_yields_now():
    if not os.path.isfile(G.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_refs.so not built (reference tree absent)")
    for name in NAMES:
        prob, opts, has, gold = _scene(name)
        now = G.run_reference(prob, opts, has)
        for k in ("has_ref", "src_image", "src_obs"):
            assert np.array_equal(now[k], gold[k]), (name, k)
        assert np.array_equal(now["descriptor"], gold["descriptor"]) and np.array_equal(now["obs_cost"], gold["obs_cost"], equal_nan=True)
