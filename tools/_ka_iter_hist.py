#!/usr/bin/env python
"""Distribution of the work over the sub-problems of the KA benchmark solve (BASELINE configs[1]): LM iterations, successful steps
and node stencils interpolated per sub-problem -- what the slowest workgroups of pxr_ka_solve's single launch do."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bench_ka import make_problem_gpu  # noqa: E402
from pixsfm_amd.engine import Context, PatchArena, interp_cfg, make_loss  # noqa: E402
from pixsfm_amd.ka_engine import KAProblem  # noqa: E402

dev = "cuda:0"
prob, patches = make_problem_gpu(dev, 10000, 10)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
arena = PatchArena(ctx, len(prob["kp"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, prob["corners"], prob["scales"])
ka = KAProblem(ctx, arena, prob)
tot, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
it = np.array([p["iterations"] for p in per]); su = np.array([p["num_successful"] for p in per])
st = np.array([p["linear_iterations"] for p in per]); nu = np.array([p["num_camera_unknowns"] for p in per])
nodes = np.bincount(prob["node_problem"])
probes = (st - (1 + su) * nodes) / np.maximum(1, nu // 2)
print(json.dumps({"kernel_ms": tot["total_ms"] - tot["setup_ms"],
                  "iterations_hist": np.bincount(it).tolist(), "successful_hist": np.bincount(su).tolist(),
                  "probes_per_problem": {"mean": float(probes.mean()), "hist": np.bincount(np.rint(probes).astype(int)).tolist()},
                  "probes_per_iteration_mean": float((probes / np.maximum(1, it)).mean()),
                  "stencils_per_problem": {"mean": float(st.mean()), "max": int(st.max()), "p95": float(np.percentile(st, 95))},
                  "heaviest": [{"prob": int(i), "iterations": int(it[i]), "successful": int(su[i]), "stencils": int(st[i]), "probes": float(probes[i]),
                                "unknowns": int(nu[i]), "initial_cost": per[i]["initial_cost"], "final_cost": per[i]["final_cost"]} for i in np.argsort(-st)[:8]],
                  "terminations": np.bincount(np.array([p["termination"] for p in per])).tolist()}))
