"""Experiment: what would a PERFECT heavy-first dispatch order buy the one-phase KA launch?  Solve configs[1] once, read every
sub-problem's work (node stencils interpolated), relabel the sub-problems so that the heaviest get the lowest indices (dispatch
is in index order), solve again and compare the kernel times.  python tools/_ka_oracle_schedule.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import bench_ka
from pixsfm_amd.engine import Context, PatchArena, interp_cfg, make_loss
from pixsfm_amd.ka_engine import KAProblem

os.environ["PXR_KA_TWO_PHASE"] = "0"
torch.cuda.set_device(0)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
prob, patches = bench_ka.make_problem_gpu("cuda:0", 10000, 10)
prob.pop("node_track")
arena = PatchArena(ctx, len(prob["kp"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, prob["corners"], prob["scales"])
kp0 = np.ascontiguousarray(prob["kp"], np.float64)


def timed(p, reps=6):
    ka = KAProblem(ctx, arena, p)
    ms, per = [], None
    for _ in range(reps):
        ka.d["kp"].upload(kp0); ctx.sync()
        total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
        ms.append(total["total_ms"] - total["setup_ms"])
    return float(np.mean(ms[1:])), float(np.min(ms)), per, total


base_ms, base_min, per, total = timed(prob)
work = np.array([s["linear_iterations"] for s in per])
its = np.array([s["iterations"] for s in per])
order = np.argsort(-work, kind="stable")                  # heaviest first
new_label = np.empty(len(order), np.int64); new_label[order] = np.arange(len(order))
p2 = dict(prob, node_problem=new_label[np.asarray(prob["node_problem"])].astype(np.int32))
best_ms, best_min, per2, total2 = timed(p2)
rev = dict(prob, node_problem=(len(order) - 1 - new_label[np.asarray(prob["node_problem"])]).astype(np.int32))
worst_ms, worst_min, _, _ = timed(rev)
print(json.dumps({"index_order_ms": base_ms, "index_order_min": base_min, "heaviest_first_ms": best_ms, "heaviest_first_min": best_min,
                  "lightest_first_ms": worst_ms, "heavy_sub_problems(>=7 iterations)": int((its >= 7).sum()),
                  "work_hist": np.percentile(work, [50, 90, 94, 97, 100]).tolist(), "same_final_cost": total["final_cost"] == total2["final_cost"]}))

# ---- how well does "a keypoint sits ON its bound after ONE LM iteration" predict the heavy sub-problems? ----------------------------
from pixsfm_amd.engine import lm_options
ka = KAProblem(ctx, arena, prob)
ka.d["kp"].upload(kp0); ctx.sync()
ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, options=lm_options(parameter_tolerance=1e-5, max_iterations=1))
kp1 = ka.keypoints()
moved = np.abs(kp1 - kp0).max(axis=1)
on_bound_node = np.abs(moved - 4.0) < 1e-9
flag = np.zeros(len(work), bool)
np.logical_or.at(flag, np.asarray(prob["node_problem"]), on_bound_node)
heavy = work >= 3000
print(json.dumps({"predicted_heavy": int(flag.sum()), "truly_heavy(work >= 3000 stencils)": int(heavy.sum()),
                  "heavy_and_predicted": int((flag & heavy).sum()), "heavy_missed": int((~flag & heavy).sum()),
                  "light_flagged": int((flag & ~heavy).sum()),
                  "medium(1000..3000)": int(((work >= 1000) & (work < 3000)).sum()), "medium_flagged": int((flag & (work >= 1000) & (work < 3000)).sum())}))
# the schedule that predictor gives: flagged first
order2 = np.argsort(~flag, kind="stable")
lab2 = np.empty(len(order2), np.int64); lab2[order2] = np.arange(len(order2))
p3 = dict(prob, node_problem=lab2[np.asarray(prob["node_problem"])].astype(np.int32))
ms3, min3, _, _ = timed(p3)
print(json.dumps({"flagged_first_ms": ms3, "flagged_first_min": min3}))
