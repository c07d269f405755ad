"""CPU: the C-ABI library loads and exports every symbol include/pixsfm_hip.h declares (no compute
calls -- there is no GPU here), the ctypes layer covers all of them, and the host-side logic
(problem packing, sharding, option handling) behaves like the reference's Python."""
import ctypes
import os
import re
from collections import Counter

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pixsfm_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pxr_[a-z0-9_]+)\s*\(", txt)) - {"pxr_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    from pixsfm_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build() must have produced libpixsfm_hip.so (hipcc cross-compiles)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libpixsfm_hip.so does not export %s" % n
    assert lib.pxr_version() >= 100


def test_ctypes_layer_covers_the_header():
    from pixsfm_amd import _lib
    assert sorted(_lib.declared_symbols()) == _declared()


def test_struct_layouts_match_the_header_sizes():
    """sizeof checks against the C layout rules for the structs crossing the boundary."""
    from pixsfm_amd import _lib
    assert ctypes.sizeof(_lib.InterpCfg) == 12
    assert ctypes.sizeof(_lib.Loss) == 16
    assert ctypes.sizeof(_lib.BaView) == 8 * 14
    assert ctypes.sizeof(_lib.KaView) == 8 * 20
    assert ctypes.sizeof(_lib.LMOptions) == 8 * 16
    assert ctypes.sizeof(_lib.LMSummary) == 8 * 11


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """sizeof / offsetof of every struct of include/pixsfm_hip.h as gcc lays them out, against the ctypes mirror."""
    import shutil
    import subprocess
    from pixsfm_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    structs = {"pxr_interp_cfg": _lib.InterpCfg, "pxr_loss": _lib.Loss, "pxr_ba_view": _lib.BaView,
               "pxr_ka_view": _lib.KaView, "pxr_lm_options": _lib.LMOptions, "pxr_lm_summary": _lib.LMSummary}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pixsfm_hip.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in ct._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    seen = 0
    for ln in out:
        if not ln:
            continue
        cname, field, val = ln.split()
        ct = structs[cname]
        want = ctypes.sizeof(ct) if field == "sizeof" else getattr(ct, field).offset
        assert int(val) == want, (cname, field, val, want)
        seen += 1
    assert seen > 60


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    """Creating a context without a GPU must raise; nothing in the product imports oracle/."""
    import torch
    from pixsfm_amd import PixsfmHipError
    from pixsfm_amd.engine import Context
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(PixsfmHipError):
        Context(0)
    pkg = os.path.join(ROOT, "pixel-perfect-sfm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                # mentioning the oracle in a comment is fine; importing / loading / linking it is not
                assert "import pxo" not in src and "liboracle" not in src and "pxo_" not in src.replace("pxo_solve.c", ""), f


def test_interp_cfg_rejects_configurations_outside_the_path():
    from pixsfm_amd.engine import interp_cfg, make_loss
    c = interp_cfg(**{"nodes": [[0.0, 0.0]], "mode": "BICUBIC", "l2_normalize": True, "ncc_normalize": False,
                      "use_float_simd": False})          # pixsfm/base/main.py:1-7 defaults
    assert (c.l2_normalize, c.use_float_simd, c.check_bounds) == (1, 0, 0)
    for bad in ({"mode": "BILINEAR"}, {"nodes": [[0, 0], [1, 0]]}, {"ncc_normalize": True}):
        with pytest.raises(ValueError):
            interp_cfg(**bad)
    with pytest.raises(ValueError):
        make_loss("arctan", [1.0])
    assert make_loss("cauchy", [0.25]).a == 0.25


def test_find_problem_labels_first_fit_decreasing():
    """keypoint_adjustment/main.py:13-57: every track in one bin, bins <= max unless a single track is larger."""
    from pixsfm_amd.api.keypoint_adjustment import find_problem_labels
    rng = np.random.default_rng(0)
    sizes = rng.integers(2, 30, 40)
    track_labels = np.repeat(np.arange(40), sizes)
    labels, bins = find_problem_labels(track_labels, 50)
    labels = np.array(labels)
    for t in range(40):
        assert len(set(labels[track_labels == t])) == 1
    cnt = Counter(labels.tolist())
    assert sorted(cnt.values()) == sorted(bins) and max(bins) <= 50
    # an oversized track gets its own bin
    labels2, bins2 = find_problem_labels(np.repeat([0, 1, 2], [70, 5, 5]), 50)
    assert bins2[labels2[0]] == 70 and labels2[70] == labels2[75]
    # max_per_problem == -1: bins sized by the largest track
    _, bins3 = find_problem_labels(np.repeat([0, 1, 2], [7, 5, 2]), -1)
    assert max(bins3) == 7


def test_problem_packing_matches_reference_vectors():
    """pack_tracks_into_problems (our rewrite) against labels / bin sizes produced by the REFERENCE's own
    find_problem_labels (keypoint_adjustment/main.py:13-57, run by tests/golden/make_golden_packing.py)."""
    import importlib.util
    from pixsfm_amd.ka_engine import pack_tracks_into_problems
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_packing", os.path.join(here, "golden", "make_golden_packing.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(here, "golden", "packing_ref.npz"))
    n = 0
    for name, labels, cap, edge_counts in gen.cases():
        got_labels, got_bins = pack_tracks_into_problems(labels, cap, edge_counts)
        assert np.array_equal(np.asarray(got_labels), gold[name + "_labels"]), name
        assert np.array_equal(np.asarray(got_bins), gold[name + "_bins"]), name
        n += 1
    assert n >= 40
    if os.path.exists(gen.SRC):          # build container: also against the reference function run live
        ref = gen.load_reference_function()
        rng = np.random.default_rng(5)
        for _ in range(20):
            sizes = rng.integers(1, 40, int(rng.integers(1, 200)))
            labels = rng.permutation(np.repeat(np.arange(len(sizes)), sizes))
            want = ref([int(v) for v in labels], 50)
            got = pack_tracks_into_problems(labels, 50)
            assert list(want[0]) == list(got[0]) and list(want[1]) == list(got[1])
    with pytest.raises(ValueError):
        pack_tracks_into_problems(np.array([0, 2, 2]), 50)     # label 1 missing: not contiguous track ids


def test_ka_csr_grouping_and_validation():
    from pixsfm_amd.ka_engine import _csr
    ptr, ids = _csr([2, 0, 2, 1, 0], 4)
    assert ptr.tolist() == [0, 2, 3, 5, 5] and ids.tolist() == [1, 4, 3, 0, 2]


def test_shard_ba_problem_partitions_points_and_replicates_cameras():
    from pixsfm_amd import synthetic
    from pixsfm_amd.parallel import assign_problems_to_ranks, balanced_ranges, shard_ba_problem
    prob = synthetic.make_ba_problem(n_cams=4, n_points=23, obs_per_point=3, seed=1, channels=8, patch_size=8)
    seen = []
    for r in range(3):
        sh, ids = shard_ba_problem(prob, r, 3)
        seen.extend(ids.tolist())
        assert np.array_equal(sh["qvec"], prob["qvec"]) and np.array_equal(sh["cam_params"], prob["cam_params"])
        assert np.array_equal(sh["xyz"], prob["xyz"][ids]) and np.array_equal(sh["refs"], prob["refs"][ids])
        g = np.nonzero(np.isin(prob["obs_point"], ids))[0]
        assert np.array_equal(sh["patches"], prob["patches"][g])
        assert np.array_equal(ids[sh["obs_point"]], prob["obs_point"][g])
    assert sorted(seen) == list(range(23))
    assert balanced_ranges([1] * 10, 4) == [(0, 3), (3, 5), (5, 8), (8, 10)]
    owner = assign_problems_to_ranks([9, 1, 8, 2, 7, 3], 2)
    load = [sum(s for s, o in zip([9, 1, 8, 2, 7, 3], owner) if o == r) for r in range(2)]
    assert abs(load[0] - load[1]) <= 2


def test_iteration_callback_hook_logic_without_a_gpu():
    """Context.set_iteration_callbacks: the largest answer of the callbacks wins, None counts as continue, an exception inside a
    callback becomes SOLVER_ABORT and is re-raised when the hook is removed (ctypes alone would print it and return garbage)."""
    import ctypes as C
    from pixsfm_amd import _lib, engine

    class FakeLib:
        def pxr_set_iteration_callback(self, handle, fn, user):
            return 0
    ctx = engine.Context.__new__(engine.Context)
    ctx.lib, ctx.handle = FakeLib(), None
    summary = _lib.IterationSummary(iteration=3, cost=2.5)
    seen = []
    ctx.set_iteration_callbacks([lambda it: seen.append((it.iteration, it.cost)), lambda it: 2 if it.iteration == 3 else 0])
    assert ctx._iter_cb(C.pointer(summary), None) == 2 and seen == [(3, 2.5)]
    ctx.set_iteration_callbacks(None)

    def bad(it):
        raise KeyError("boom")
    ctx.set_iteration_callbacks([bad])
    assert ctx._iter_cb(C.pointer(summary), None) == 1
    with pytest.raises(KeyError, match="boom"):
        ctx.set_iteration_callbacks(None)
    ctx.set_iteration_callbacks(None)                       # nothing pending any more
    ctx.handle = None
