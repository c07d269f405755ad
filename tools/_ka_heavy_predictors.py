"""Which early quantity tells the heavy KA sub-problems (long line searches) from the light ones?  configs[1]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_ka
from pixsfm_amd.engine import Context, PatchArena, interp_cfg, make_loss, lm_options
from pixsfm_amd.ka_engine import KAProblem
os.environ["PXR_KA_TWO_PHASE"] = "0"
torch.cuda.set_device(0)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
prob, patches = bench_ka.make_problem_gpu("cuda:0", 10000, 10); prob.pop("node_track")
arena = PatchArena(ctx, len(prob["kp"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, prob["corners"], prob["scales"])
kp0 = np.ascontiguousarray(prob["kp"], np.float64)
lab = np.asarray(prob["node_problem"])
def run(iters):
    ka = KAProblem(ctx, arena, prob); ka.d["kp"].upload(kp0); ctx.sync()
    t, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, options=lm_options(parameter_tolerance=1e-5, max_iterations=iters), per_problem=True)
    return ka.keypoints(), per
kpf, perf = run(100)
work = np.array([s["linear_iterations"] for s in perf]); heavy = work >= 3000
def per_prob_max(x):
    out = np.zeros(len(work)); np.maximum.at(out, lab, x); return out
res = {"heavy": int(heavy.sum())}
final_move = np.abs(kpf - kp0).max(1)
res["final_on_bound(any kp)"] = {"heavy": float((per_prob_max(final_move) > 4 - 1e-9)[heavy].mean()), "light": float((per_prob_max(final_move) > 4 - 1e-9)[~heavy].mean())}
for it in (1, 2, 3):
    kpi, peri = run(it)
    mv = per_prob_max(np.abs(kpi - kp0).max(1))
    cost = np.array([s["final_cost"] for s in peri])
    for thr in (2.5, 3.0, 3.5, 4 - 1e-9):
        f = mv > thr
        res["after_%d_iterations move>%.1f" % (it, thr)] = {"flagged": int(f.sum()), "heavy_caught": int((f & heavy).sum()), "light_flagged": int((f & ~heavy).sum())}
    res["after_%d cost: heavy median %.4f light median %.4f" % (it, np.median(cost[heavy]), np.median(cost[~heavy]))] = None
    for q in (90, 93, 95):
        thr = np.percentile(cost, q); f = cost > thr
        res["after_%d cost>p%d" % (it, q)] = {"flagged": int(f.sum()), "heavy_caught": int((f & heavy).sum())}
c0 = np.array([s["initial_cost"] for s in perf])
for q in (80, 90, 93):
    f = c0 > np.percentile(c0, q); res["initial cost>p%d" % q] = {"flagged": int(f.sum()), "heavy_caught": int((f & heavy).sum())}
print(json.dumps(res, indent=0))
