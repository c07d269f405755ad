"""One `lm` + one `lm_no_inner` solve of the bench scene (10 iterations each) and nothing else -- the command traced by
rocprofv3 --kernel-trace --stats to see what an LM iteration is made of in a given mode (PXR_DETERMINISTIC / PXR_GRAM_CACHE in
the environment).  python tools/_lm_solve_once.py [points] [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
import numpy as np
import torch

import bench
from pixsfm_amd import synthetic_gpu
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
torch.cuda.set_device(0)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
prob, patches = synthetic_gpu.make_ba_problem_gpu(dev, n_cams=200, n_points=pts, obs_per_point=5, channels=128, patch_size=16, seed=2,
                                                  point_range=(0, pts))
arena = PatchArena(ctx, len(prob["obs_image"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, prob["corners"], prob["scales"])
ba = BAProblem(ctx, arena, prob)
pose_const, tmask, cmask, ptc = bench.default_gauge(200, len(prob["xyz"]))
out = {"deterministic": ctx.deterministic, "gram_cache": ctx.gram_cache}
for rep in range(2):                       # (the first pair warms the workspaces up)
    for key, inner in (("lm", True), ("lm_no_inner", False)):
        bench.reset_parameters(ba, prob)
        ctx.sync()
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pose_const, tmask, cmask, ptc,
                     options=lm_options(max_iterations=iters, use_inner_iterations=inner))
        out[key] = {"ms_per_iter": s["total_ms"] / max(1, s["iterations"]), "iterations": s["iterations"], "successful": s["num_successful"],
                    "final_cost": s["final_cost"], "setup_ms": s["setup_ms"]}
print(json.dumps(out))
