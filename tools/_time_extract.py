"""Times pxr_arena_extract (the sparse patch producer) on a synthetic dense feature map (tools/ helper):
python tools/_time_extract.py [n_keypoints] [C] [h] [w]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pixsfm_amd.engine import Context, PatchArena  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
h = int(sys.argv[3]) if len(sys.argv) > 3 else 256
w = int(sys.argv[4]) if len(sys.argv) > 4 else 320
torch.manual_seed(0)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
kps = ctx.to_device(np.stack([rng.uniform(0, 4 * w, n), rng.uniform(0, 4 * h, n)], 1), np.float64)
for src in (torch.float32, torch.float16):
    fmap = torch.randn(C, h, w, device="cuda", dtype=src)
    arena = PatchArena(ctx, n, 16, 16, C, np.float16)
    arena.extract(0, fmap, kps, (4 * w, 4 * h))
    ctx.sync()
    ctx.timer_start()
    reps = 5
    for _ in range(reps):
        arena.extract(0, fmap, kps, (4 * w, 4 * h))
    ms = ctx.timer_stop() / reps
    es = 4 if src == torch.float32 else 2
    rd, wr = n * 256 * C * es / 1e9, n * 256 * C * 2 / 1e9
    print("src %-8s %d patches 16x16x%d from a %dx%dx%d map: %.3f ms   gathered %.2f GB (map itself %.0f MB: L2/MALL resident) "
          "written %.2f GB -> %.0f GB/s of HBM writes, %.0f GB/s algorithmic" %
          (str(src).split(".")[1], n, C, C, h, w, ms, rd, C * h * w * es / 1e6, wr, wr / ms * 1e3, (rd + wr) / ms * 1e3))
    arena.close()
