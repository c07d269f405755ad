"""Times pxr_costmap_extract on synthetic fp16 patches (tools/ helper, not part of the bench contract):
python tools/_time_costmap.py [n_patches] [patch_size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pixsfm_amd._lib import check  # noqa: E402
from pixsfm_amd.engine import Context, PatchArena, make_loss  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Cn = 128
torch.manual_seed(0)
patches = torch.randn(n, ps, ps, Cn, device="cuda", dtype=torch.float16)
ctx = Context(0, stream=torch.cuda.current_stream().cuda_stream)
arena = PatchArena(ctx, n, ps, ps, Cn, np.float16, device_ptr=patches.data_ptr())
arena.upload(0, None, np.zeros((n, 2), np.int32), np.ones((n, 2)))
n_pts = n // 5
refs = ctx.to_device(np.random.default_rng(0).normal(size=(n_pts, Cn)), np.float64)
pidx = ctx.to_device(np.arange(n, dtype=np.int64), np.int64)
ridx = ctx.to_device((np.arange(n) // 5).astype(np.int32), np.int32)
for grad, name in ((1, "gradient field"), (0, "cost only")):
    out = PatchArena(ctx, n, ps, ps, 3 if grad else 1, np.float16)
    for loss in (make_loss("trivial", []), make_loss("cauchy", [0.25])):
        def run():
            check(ctx.lib.pxr_costmap_extract(ctx.handle, arena.handle, out.handle, 0, n, pidx.ptr, ridx.ptr, refs.ptr,
                                              C.byref(loss), grad, 0), "pxr_costmap_extract")
        run(); ctx.sync()
        ctx.timer_start()
        for _ in range(5):
            run()
        ms = ctx.timer_stop() / 5
        gb = n * (ps * ps * Cn * 2 + Cn * 8 + (3 if grad else 1) * ps * ps * 2) / 1e9
        print("%-15s loss %d: %.3f ms  %.0f GB/s  (%.2f of 8 TB/s)" % (name, loss.type, ms, gb / ms * 1e3, gb / ms * 1e3 / 8000))
    out.close()
