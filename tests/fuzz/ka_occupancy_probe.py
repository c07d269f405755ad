"""Debug probe for the KA solve kernel under a tighter register cap ("occupancy note"):
runs the default-options KA test problem through the library named by PXR_HIP_LIB (e.g. a build of csrc/pxr_ka.hip with
-DPXR_KA_WAVES=3) and lists, per sub-problem, where the in-kernel LM departs from the oracle.

    PXR_HIP_LIB=tools/debug/libpixsfm_hip_occ3.so python tests/fuzz/ka_occupancy_probe.py [max_iterations ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import pxo  # noqa: E402
import pxo_ka  # noqa: E402
from pixsfm_amd import synthetic_ka  # noqa: E402
from pixsfm_amd.engine import Context, PatchArena, interp_cfg, lm_options, make_loss  # noqa: E402
from pixsfm_amd.ka_engine import KAProblem  # noqa: E402

print("library:", os.environ.get("PXR_HIP_LIB", "(default)"))
ctx = Context(0)
prob = synthetic_ka.make_ka_problem(n_tracks=30, track_len=6, seed=5, max_kps_per_problem=50)
arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
for max_it in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 100]:
    ka = KAProblem(ctx, arena, prob)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0,
                          options=lm_options(parameter_tolerance=1e-5, max_iterations=max_it), per_problem=True)
    kp = ka.keypoints()
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0,
                                pxo.lm_options(parameter_tolerance=1e-5, max_iterations=max_it))
    bad = 0
    for i, (g, o) in enumerate(zip(per, sums)):
        nodes = np.nonzero(prob["node_problem"] == i)[0]
        dk = np.abs(kp[nodes] - kpo[nodes]).max()
        same = g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"] and \
            abs(g["final_cost"] - o["final_cost"]) <= 1e-9 * max(o["initial_cost"], 1e-300) and dk < 1e-6
        if not same:
            bad += 1
            print("  max_it %3d problem %2d: gpu it %d ok %d cost %.12e -> %.12e | oracle it %d ok %d cost %.12e -> %.12e | max |dkp| %.3e"
                  % (max_it, i, g["iterations"], g["num_successful"], g["initial_cost"], g["final_cost"], o["iterations"],
                     o["num_successful"], o["initial_cost"], o["final_cost"], dk))
    print("max_iterations %3d: %d of %d sub-problems differ from the oracle" % (max_it, bad, len(per)))
