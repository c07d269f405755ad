"""Where the set-up of pxr_ba_solve goes (PXR_VERBOSE marks) on the bench's scene: the first solve of a process and later ones.
    python tools/_lm_setup_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--marks" in sys.argv:
    os.environ["PXR_VERBOSE"] = "1"
import bench  # noqa: E402

args = bench.parse_args(["--no-cpu-baseline", "--no-api-e2e", "--no-costmap", "--no-telemetry", "--no-ka"])
job = bench.Job(args)
from pixsfm_amd.engine import interp_cfg, lm_options, make_loss  # noqa: E402
prob, patches, arena, ba, total_points = bench.make_scene(job)
cfg = interp_cfg()
pose_const, tmask, cmask, ptc = bench.default_gauge(args.cams, len(prob["xyz"]))
for name, inner in (("lm (first solve of the process)", True), ("lm_no_inner", False)) + (("lm again", True), ("lm_no_inner again", False)) * 4:
    bench.reset_parameters(ba, prob)
    job.ctx.sync()
    sys.stderr.write("==== %s\n" % name)
    sys.stderr.flush()
    import time
    t_wall = time.perf_counter()
    r = ba.solve(cfg, make_loss("cauchy", [0.25]), pose_const, tmask, cmask, ptc,
                 options=lm_options(max_iterations=10, use_inner_iterations=inner))
    t_wall = (time.perf_counter() - t_wall) * 1e3
    sys.stderr.write("     total %.3f ms, setup %.3f ms, initial %.3f ms, %d iterations -> %.3f ms / iteration; the call: %.3f ms wall\n" % (
        r["total_ms"], r["setup_ms"], r.get("initial_us", float("nan")) * 1e-3, r["iterations"], r["total_ms"] / r["iterations"], t_wall))
