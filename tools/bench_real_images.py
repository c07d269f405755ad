"""Throughput of the hot path on REAL image content (tests/real_scene.py: photographs of the reference's demo set) next to
the synthetic scene of the same dimensions.  One JSON object on stdout.  python tools/bench_real_images.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pixel-perfect-sfm_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import real_scene
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    ctx = Context(0)
    sc = real_scene.make_scene(n_views=8, n_points=400)
    n_obs, C = len(sc["obs_image"]), sc["channels"]
    order = np.argsort(sc["obs_image"], kind="stable")
    inv = np.empty(n_obs, np.int64); inv[order] = np.arange(n_obs)
    arena = PatchArena(ctx, n_obs, 16, 16, C, np.float16)
    first = 0
    for v in range(len(sc["fmaps"])):
        sel = order[sc["obs_image"][order] == v]
        arena.extract(first, torch.from_numpy(sc["fmaps"][v]).cuda().contiguous(), sc["detected"][sel], sc["image_size"])
        first += len(sel)
    out = {"scene": "%d views of a photograph (sacre_coeur crop) on a plane, %d points, %d observations, 128-ch 3x3 conv bank at "
                    "half resolution, L2-normalised fp16 16x16 patches" % (len(sc["fmaps"]), len(sc["xyz"]), n_obs)}

    def time_eval(ba, reps=200):
        cfg = interp_cfg()
        for _ in range(5):
            ba.eval(cfg, with_jacobian=True)
        ctx.timer_start()
        for _ in range(reps):
            ba.eval(cfg, with_jacobian=True)
        return ctx.timer_stop() / reps
    prob = {k: sc[k] for k in ("obs_image", "obs_point", "image_camera", "qvec", "tvec", "cam_model", "cam_params", "xyz")}
    prob["obs_patch"] = inv
    prob["refs"] = np.zeros((len(sc["xyz"]), C))
    ba = BAProblem(ctx, arena, prob)
    ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]))
    ms = time_eval(ba)
    syn = synthetic.make_ba_problem(n_cams=len(sc["fmaps"]), n_points=len(sc["xyz"]), obs_per_point=max(2, round(n_obs / len(sc["xyz"]))), seed=4)
    sarena = PatchArena.from_numpy(ctx, syn["patches"], syn["corners"], syn["scales"])
    ms_syn = time_eval(BAProblem(ctx, sarena, syn))
    out["ba_eval"] = {"real_ms": ms, "real_blocks_per_s": n_obs / (ms * 1e-3), "synthetic_same_size_ms": ms_syn,
                      "synthetic_blocks_per_s": len(syn["obs_image"]) / (ms_syn * 1e-3), "n_obs_real": n_obs, "n_obs_synthetic": len(syn["obs_image"]),
                      "note": "a few thousand observations do not fill the GPU (launch-bound): the comparison is real vs synthetic "
                              "content at equal size, not a throughput figure"}
    n_img = len(sc["qvec"])
    pc = np.zeros(n_img, np.uint8); pc[0] = 1
    tm = np.zeros(n_img, np.uint8); tm[1] = 1
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), pc, tm, np.full(n_img, 0b1111, np.uint16), np.zeros(len(sc["xyz"]), np.uint8),
                 options=lm_options(max_iterations=20, use_inner_iterations=True))
    q, t, k, X = ba.params()
    out["ba_solve"] = {k2: s[k2] for k2 in ("iterations", "num_successful", "initial_cost", "final_cost", "total_ms")}
    kprob = dict(kp=sc["detected"].copy(), node_patch=inv, node_const=sc["node_const"], node_problem=sc["node_problem"],
                 edge_src=sc["edge_src"], edge_dst=sc["edge_dst"], edge_w=sc["edge_w"])
    ka = KAProblem(ctx, arena, kprob)
    ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0)
    ka.d["kp"].upload(np.ascontiguousarray(kprob["kp"], dtype=np.float64))
    tot, _ = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0)
    out["ka_solve"] = {"sub_problems": int(sc["n_problems"]), "edges": int(len(sc["edge_src"])), "kernel_ms": tot["total_ms"] - tot["setup_ms"],
                       "lm_iterations_max": tot["iterations"], "initial_cost": tot["initial_cost"], "final_cost": tot["final_cost"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
