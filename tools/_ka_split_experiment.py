"""Does a 50-keypoint sub-problem run faster as a label group on SEVERAL workgroups (chunks of whole tracks, cross-workgroup sums)?
    python tools/_ka_split_experiment.py <keypoints per chunk> [bench_ka arguments]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pixsfm_amd import ka_engine  # noqa: E402

kps = int(sys.argv[1])
if kps > 0:
    ka_engine.CHUNK_KPS, ka_engine.CHUNK_FROM = kps, kps + 1
    ka_engine.chunk_label_groups.__defaults__ = (kps, kps + 1, ka_engine.MAX_CHUNKS)
import bench_ka  # noqa: E402
r = bench_ka.run(steps=5, solves=5)
s = r["solve"]
print("chunk", kps, {k: s[k] for k in ("kernel_ms", "kernel_ms_min", "successful_steps", "final_cost", "lm_iterations_max")}, "sub-problems", r.get("options"))
