#!/usr/bin/env python
"""KA throughput on BASELINE.json configs[1]: synthetic 10k tracks / 100k observations (nodes),
complete intra-track graph (45 edges per 10-node track -> 450k residual blocks), 128-ch fp16
16x16 patches, Cauchy(0.25), bound 4 px, <= 50 keypoints per sub-problem.

Reports (one JSON line): per-edge residual+Jacobian evaluation rate (pxr_ka_eval, the
reference's unit of work: 2 interpolations per edge) and the full bounded-LM solve time
(pxr_ka_solve: node-centric evaluation inside one workgroup per sub-problem).
Not the headline bench (bench.py is); numbers are quoted in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_problem_gpu(dev, n_tracks, track_len, C=128, PS=16, seed=1, sigma=1.0, chunk=16384, max_kps_per_problem=50,
                     strategy="featuremetric"):
    """max_kps_per_problem: pixsfm's KeypointAdjuster option (50 by default, 1000 in configs/low_memory.yaml; 0 = ONE problem,
    split_in_subproblems = false, keypoint_adjustment/main.py:197-202).  strategy "topological_reference"
    (topological_reference_keypoint_optimizer.h:8-15): edges to the track's root only, unit weights."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.ka_engine import pack_tracks_into_problems
    rng = np.random.default_rng(seed)
    n = n_tracks * track_len
    track = np.repeat(np.arange(n_tracks), track_len)
    true_xy = rng.uniform(50, 950, (n, 2))
    kp0 = true_xy + rng.normal(0, sigma, (n, 2))
    corners = np.floor(kp0 - PS / 2.0).astype(np.int32)
    wx, wy, th = (torch.tensor(a, dtype=torch.float32, device=dev) for a in synthetic._basis())
    g = torch.Generator(device=dev); g.manual_seed(seed)
    A = torch.randn((n_tracks, C, synthetic.N_BASIS), generator=g, device=dev).half()
    patches = torch.empty((n, PS, PS, C), dtype=torch.float16, device=dev)
    ii = torch.arange(PS, device=dev, dtype=torch.float32)
    cr = torch.tensor(corners, device=dev, dtype=torch.float32)
    ct = torch.tensor(true_xy, device=dev, dtype=torch.float32)
    tr = torch.tensor(track, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        cx = cr[s:e, 0:1] + ii[None] + 0.5 - ct[s:e, 0:1]
        cy = cr[s:e, 1:2] + ii[None] + 0.5 - ct[s:e, 1:2]
        ph = torch.cos(cy[:, :, None, None] * wy + cx[:, None, :, None] * wx + th)
        val = torch.bmm(ph.reshape(e - s, PS * PS, -1), A[tr[s:e]].float().transpose(1, 2))
        patches[s:e] = (val / val.norm(dim=-1, keepdim=True)).reshape(e - s, PS, PS, C).half()
    # complete intra-track graph, one direction per pair
    a_idx, b_idx = np.triu_indices(track_len, 1)
    base = (np.arange(n_tracks) * track_len)[:, None]
    edge_src = (base + a_idx[None]).reshape(-1).astype(np.int32)
    edge_dst = (base + b_idx[None]).reshape(-1).astype(np.int32)
    edge_w = rng.uniform(0.5, 1.0, len(edge_src))
    score = np.zeros(n); np.add.at(score, edge_src, edge_w); np.add.at(score, edge_dst, edge_w)
    node_const = np.zeros(n, np.uint8)
    node_const[(np.arange(n_tracks) * track_len) + score.reshape(n_tracks, track_len).argmax(1)] = 1
    if strategy == "topological_reference":       # root_edges_only, weight_by_sim = false
        keep = (node_const[edge_src] | node_const[edge_dst]).astype(bool)
        edge_src, edge_dst, edge_w = edge_src[keep], edge_dst[keep], np.ones(int(keep.sum()))
    if max_kps_per_problem > 0:
        labels, bins = pack_tracks_into_problems(track, max_kps_per_problem)
    else:
        labels, bins = np.zeros(n, np.int32), [0]
    return dict(kp=kp0, node_patch=np.arange(n, dtype=np.int64), node_const=node_const,
                node_problem=np.array(labels, np.int32), node_track=track, edge_src=edge_src, edge_dst=edge_dst, edge_w=edge_w,
                corners=corners, scales=np.ones((n, 2)), true_xy=true_xy, n_problems=len(bins)), patches


def cpu_legs_on_sample(prob, patches, dev):
    """The CPU side of the KA half of the metric on a bounded sample (the first 8 x usable-CPUs sub-problems, at most all):
    the oracle's bounded LM, ONE single-threaded solve per task over a pool of threads (keypoint_adjustment/main.py:66-80),
    and the reference's own FeatureMetric2DCostFunctor on dual numbers per residual block -- both timed inside C
    (oracle/pxo_bench_harness.h)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pxo
    import pxo_cpubench
    logical, _, _ = pxo_cpubench.cpu_topology()
    n_prob = int(prob["n_problems"])
    take = min(n_prob, max(16, 8 * pxo_cpubench.usable_cpus(logical)))          # (a cgroup quota caps what more tasks would buy)
    node_sel = np.nonzero(prob["node_problem"] < take)[0]
    new_node = np.full(len(prob["kp"]), -1); new_node[node_sel] = np.arange(len(node_sel))
    edge_sel = np.nonzero(new_node[prob["edge_src"]] >= 0)[0]
    sub = dict(kp=prob["kp"][node_sel], node_patch=np.arange(len(node_sel), dtype=np.int64), node_const=prob["node_const"][node_sel],
               node_problem=prob["node_problem"][node_sel].astype(np.int32),
               edge_src=new_node[prob["edge_src"][edge_sel]].astype(np.int32),
               edge_dst=new_node[prob["edge_dst"][edge_sel]].astype(np.int32), edge_w=prob["edge_w"][edge_sel],
               patches=patches[torch.as_tensor(node_sel, device=dev)].cpu().numpy(),
               corners=prob["corners"][node_sel], scales=prob["scales"][node_sel])
    what = "the first %d of the %d sub-problems of the same workload (%d keypoints, %d residual blocks, %.1f GB of patches)" % (
        take, n_prob, len(node_sel), len(edge_sel), sub["patches"].nbytes / 1e9)
    out = {}
    solve = pxo_cpubench.ka_solve_port(sub, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    solve["sample"] = what + "; oracle C restatement of the bounded LM, one single-threaded solve per task, tasks taken from a shared counter"
    solve["projected_full_solve_ms"] = 1e3 * n_prob / solve["value"]
    out["cpu_baseline"] = solve
    return out


def run(tracks=10000, track_len=10, steps=20, device_index=0, ctx=None, rank=0, world=1, cpu_legs=False, telemetry=None, solves=8,
        max_kps_per_problem=50, patch_size=16, bound=4.0, strategy="featuremetric", chunk_groups=True):
    """Runs the KA benchmark and returns its result dict (bench.py attaches it as "ka").
    world > 1 (torch.distributed initialised by the caller): STRONG scaling of BASELINE configs[1] -- the sub-problems
    are dealt to the ranks (parallel.shard_ka_problem, SURVEY 8e: no collective during the solve), every rank times its
    share, the figures use the slowest rank, and the refined keypoints are gathered at the end."""
    import types
    args = types.SimpleNamespace(tracks=tracks, track_len=track_len, steps=steps)
    from pixsfm_amd import parallel
    from pixsfm_amd.engine import Context, PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    dev = "cuda:%d" % device_index
    torch.cuda.set_device(device_index)
    prob, patches = make_problem_gpu(dev, args.tracks, args.track_len, PS=patch_size, max_kps_per_problem=max_kps_per_problem, strategy=strategy)
    if not chunk_groups:
        prob.pop("node_track")          # every label group on ONE workgroup (the form of rounds 1-5)
    ctx = ctx or Context(device_index, stream=torch.cuda.current_stream().cuda_stream)
    full, node_ids = prob, np.arange(len(prob["kp"]))
    if world > 1:
        import torch.distributed as dist
        prob, node_ids = parallel.shard_ka_problem(full, rank, world)
        patches = patches[torch.as_tensor(prob["patch_ids"], device=dev)].contiguous()     # this rank's patches only
        prob["corners"], prob["scales"] = full["corners"][prob["patch_ids"]], full["scales"][prob["patch_ids"]]
        prob["true_xy"] = full["true_xy"][node_ids]
        torch.cuda.empty_cache()

    def slowest(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def summed(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        return t.item()
    n = len(prob["kp"])
    arena = PatchArena(ctx, n, patch_size, patch_size, 128, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    cfg, ls = interp_cfg(), make_loss("cauchy", [0.25])
    cost, _, _, _ = ka.eval(cfg, ls)
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.steps):
        ka.eval(cfg, ls, out=cost)              # (no allocation inside the timed loop: hipFree synchronises)
    ms = slowest(ctx.timer_stop() / args.steps)
    c0 = summed(float(cost.download().sum()))
    cpu = cpu_legs_on_sample(prob, patches, dev) if (cpu_legs and world == 1) else None
    cold, _ = ka.solve(cfg, ls, bound=bound)          # first call: grows the context workspace
    ka.d["kp"].upload(np.ascontiguousarray(prob["kp"], dtype=np.float64))
    ctx.sync()
    if world > 1:
        dist.barrier()
    # `solves` whole solves from the same initial keypoints, back to back (a single 5 ms launch says more about the clock the
    # GPU happened to idle at than about the kernel); the shader clock / power are sampled over the loop when asked for
    kp_init = np.ascontiguousarray(prob["kp"], dtype=np.float64)
    runs, walls = [], []

    def loop():
        for _ in range(max(1, solves)):
            ka.d["kp"].upload(kp_init)
            ctx.sync()
            t0 = time.perf_counter()
            s, _ = ka.solve(cfg, ls, bound=bound)
            walls.append(time.perf_counter() - t0)
            runs.append(s)
    tel = telemetry.sample_while(loop, interval=0.002) if (telemetry is not None and rank == 0) else (loop() or None)
    total = dict(runs[-1])
    kms = [r["total_ms"] - r["setup_ms"] for r in runs]
    total_kernel_first, total_kernel_min = kms[0], min(kms)
    total["total_ms"] = float(np.mean([r["total_ms"] for r in runs]))
    total["setup_ms"] = float(np.mean([r["setup_ms"] for r in runs]))
    wall = slowest(float(np.mean(walls)))
    kp = ka.keypoints()
    n_edges_total, n_problems_total = int(summed(ka.n_edges)), int(summed(ka.n_problems))
    gather_ms = 0.0
    if world > 1:                                   # disjoint rows -> every rank holds all refined keypoints
        t1 = time.perf_counter()
        kp_all = parallel.gather_rows(kp, node_ids, len(full["kp"]))
        gather_ms = slowest(time.perf_counter() - t1) * 1e3
        assert kp_all.shape == full["kp"].shape
        for k in ("initial_cost", "final_cost"):
            total[k] = summed(total[k])
        total["total_ms"], total["setup_ms"] = slowest(total["total_ms"]), slowest(total["setup_ms"])
        total["num_successful"] = int(summed(total["num_successful"]))
        total["linear_iterations"] = int(summed(total["linear_iterations"]))
        kp, prob = kp_all, full
    # expected optimum: true + (root offset)
    tl = args.track_len
    root = prob["node_const"].astype(bool)
    off = (prob["kp"][root] - prob["true_xy"][root]).repeat(tl, axis=0)
    err0 = np.linalg.norm(prob["kp"] - (prob["true_xy"] + off), axis=1)
    err1 = np.linalg.norm(kp - (prob["true_xy"] + off), axis=1)
    out = {"workload": "BASELINE.json configs[1]: %d tracks x %d nodes, %d edges, %d sub-problems, 128-ch fp16 %dx%d"
                       % (args.tracks, tl, n_edges_total, n_problems_total, patch_size, patch_size),
           "options": {"max_kps_per_problem": max_kps_per_problem, "patch_size": patch_size, "bound": bound, "strategy": strategy,
                       "label_groups": int(ka.n_labels), "workgroups": int(ka.n_problems),
                       "groups_chunked_over_workgroups": bool(ka.problem_group is not None)},
           "n_gpus": world, "partition": "sub-problems dealt to the ranks by edge count, no collective in the solve" if world > 1 else "none",
           "gather_ms": gather_ms,
           "edge_eval": {"edges_per_s": n_edges_total / (ms * 1e-3), "kernel_ms": ms,
                         "algorithmic_GBps": 8244 * n_edges_total / (ms * 1e-3) / 1e9},
           "solve": {"wall_ms": wall * 1e3, "first_call_ms": cold["total_ms"], "total_ms": total["total_ms"],
                     "kernel_ms": total["total_ms"] - total["setup_ms"], "kernel_ms_min": slowest(total_kernel_min),
                     "kernel_ms_first": slowest(total_kernel_first), "solves_timed": len(runs),
                     "lm_iterations_max": total["iterations"],
                     "successful_steps": total["num_successful"], "initial_cost": total["initial_cost"],
                     "final_cost": total["final_cost"], "initial_cost_check": c0},
           "accuracy_px": {"median_before": float(np.median(err0)), "median_after": float(np.median(err1)),
                           "p95_after": float(np.percentile(err1, 95))}}
    # roofline of the solve kernel: the algorithmic traffic is one 4 x 4 x C stencil per node and evaluation (node-centric:
    # every linearisation interpolates all nodes of a sub-problem, every line-search probe its variable nodes); the kernel
    # counts them (summary.linear_iterations).  HBM peak from MI355X_MICROARCH.md.
    kernel_ms = out["solve"]["kernel_ms"]
    stencils = int(total["linear_iterations"])
    bytes_solve = stencils * 16 * 128 * 2          # (a 4 x 4 stencil whatever the patch size)
    ach = bytes_solve / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                       "kernel": "ka_solve_kernel_occ2<f16,128>", "kernel_ms": kernel_ms, "algorithmic_bytes": bytes_solve,
                       "node_stencils_interpolated": stencils, "evaluations_per_node": stencils / max(1, len(prob["kp"])),
                       "traffic": None,
                       "note": "the whole bounded LM of every sub-problem is ONE launch; a sub-problem's evaluations are a serial "
                               "chain (LM iteration -> line-search probes), so the kernel is latency-bound well below the HBM rate"}
    if tel is not None:
        out["telemetry"] = tel
    if cpu is not None:
        out.update(cpu)
        out["gpu_over_cpu_solve"] = (out["cpu_baseline"]["projected_full_solve_ms"] / kernel_ms) if kernel_ms > 0 else None
    arena.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=10000)
    ap.add_argument("--track-len", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--solves", type=int, default=8)
    ap.add_argument("--max-kps-per-problem", type=int, default=50, help="50: pixsfm's default; 1000: configs/low_memory.yaml; 0: one problem (split_in_subproblems = false)")
    ap.add_argument("--patch-size", type=int, default=16, help="16: default; 8: configs/low_memory.yaml")
    ap.add_argument("--bound", type=float, default=4.0, help="4.0: default; 2.0: configs/low_memory.yaml")
    ap.add_argument("--strategy", choices=("featuremetric", "topological_reference"), default="featuremetric")
    ap.add_argument("--no-chunk", action="store_true", help="keep every label group on one workgroup (rounds 1-5)")
    args = ap.parse_args()
    print(json.dumps(run(args.tracks, args.track_len, args.steps, solves=args.solves, max_kps_per_problem=args.max_kps_per_problem,
                         patch_size=args.patch_size, bound=args.bound, strategy=args.strategy, chunk_groups=not args.no_chunk)))


if __name__ == "__main__":
    main()
