"""Runs the cost-map extraction back to back for a few seconds (to sample clocks / power beside it): python tools/_loop_costmap.py [seconds] [patch size]."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pixel-perfect-sfm_amd'))
import numpy as np
import torch
from pixsfm_amd import synthetic_gpu
from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
PS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
prob, patches = synthetic_gpu.make_ba_problem_gpu(dev, n_cams=200, n_points=40000, obs_per_point=5, channels=128, patch_size=PS, seed=2)
n = len(prob["obs_image"])
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
arena = PatchArena(ctx, n, PS, PS, 128, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, prob["corners"], prob["scales"])
ba = BAProblem(ctx, arena, prob)
ba.eval(interp_cfg(), with_jacobian=True)
trivial = make_loss("trivial", [])
cm = ba.extract_costmaps(trivial)
ctx.sync()
print("[costmap loop start]", flush=True)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < secs:
    ctx.timer_start()
    for _ in range(20):
        ba.extract_costmaps(trivial, out=cm)
    ms = ctx.timer_stop() / 20
    k += 1
print("[costmap loop end] last %.3f ms per %d maps of %dx%d = %.2f TB/s algorithmic" % (ms, n, PS, PS, n * (PS * PS * 256 + 1024 + PS * PS * 6) / ms / 1e9), flush=True)
print("[eval loop start]", flush=True)
t0 = time.perf_counter()
cfg = interp_cfg()
while time.perf_counter() - t0 < secs:
    ctx.timer_start()
    for _ in range(50):
        ba.eval(cfg, with_jacobian=True)
    ms = ctx.timer_stop() / 50
print("[eval loop end] last %.3f ms per %d observations" % (ms, n), flush=True)
