#!/usr/bin/env python
"""bench.py -- featuremetric residual+Jacobian throughput on the synthetic 1M-observation BA
problem (BASELINE.json config 3: 200 cams / 200k points / 1M obs, 128-ch fp16 16x16 patches).

    python bench.py --gpus N --steps K --warmup W

A "step" = one fused residual+Jacobian evaluation (pxr_ba_eval, with_jacobian=1) of every
observation owned by the rank: projection, bicubic interpolation of the patch stencil,
L2-normalisation with analytic gradient, reference subtraction and the six-scalar reduction
that stands in for the 128 x (10+K) Jacobian block.  Inputs are resident in HBM when the
timed region starts.  N > 1: points (with all their observations and patches) are
partitioned over the ranks, no data-path collective in the evaluation.  Default = STRONG scaling
(the target of BASELINE.json: the 1M-observation problem is sharded over the N ranks); --scaling weak lets
every rank own --points points instead (N x 1M observations in one scene, cameras shared).  The LM loop's
collective is the native RCCL all-reduce of the engine (pxr_comm_init; falls back to the torch.distributed
callback if the communicator cannot be created).

Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for the roofline and
cpu_baseline definitions.  Besides the contract's fields it carries `lm` / `lm_no_inner` (LM iterations/s
on the same problem), `ka` (BASELINE configs[1]) and `costmap` (the reference's low-memory strategy on the
same scene: cost-map extraction + cost-map BA).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_obs(C, K=4, elem=2):
    """stencil 16*C*sizeof(dtype) + reference C*8 + params 8*(10+K) + 4 indices."""
    return 16 * C * elem + C * 8 + 8 * (10 + K) + 16


def cpu_baseline(prob, patches, n_sample, budget_s=10.0, lm_gauge=None):
    """The CPU legs, timed on this box's host cores on a bounded sample of the same workload (whole points: the
    first n_sample observations' points with all their observations; every camera):
      cpu_baseline      -- the oracle (kind "port"): materialised 128 x (10+K) Jacobian blocks + loss per residual
                           block, threaded over blocks like Ceres (bundle_adjustment_options.h:58);
      cpu_baseline_lm_projected -- one LM iteration of the oracle's Schur path (oracle/pxo_lm_bench.c: Jacobian evaluation,
                           Schur elimination, Cholesky, back-substitution, residual-only evaluation of the candidate),
                           all cores; the per-observation stages are scaled to the full problem, the Cholesky is not;
      cpu_reference_kernel -- the REFERENCE's own AVX2/F16C bicubic kernels (cubic_hermite_spline_simd.h + grid2d.h
                           compiled in place, oracle/_ref): interpolation only (value + both derivatives of one
                           16x16x128 fp16 patch), no projection / normalisation / Jacobian / loss.
    The sample's touched working set (4 KiB stencil per observation + references) is stated next to each figure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    import pxo
    n_obs = len(prob["obs_image"])
    n_pts_s = int(prob["obs_point"][min(n_sample, n_obs) - 1]) + 1          # whole points (observations are point-sorted)
    n_sample = int(np.searchsorted(prob["obs_point"], n_pts_s, side="left")) if n_pts_s < len(prob["xyz"]) else n_obs
    sub = dict(prob)
    for k in ("obs_image", "obs_point", "obs_patch", "corners", "scales"):
        sub[k] = prob[k][:n_sample]
    sub["xyz"], sub["refs"] = prob["xyz"][:n_pts_s], prob["refs"][:n_pts_s]
    sub["patches"] = patches[:n_sample].cpu().numpy()
    cores = os.cpu_count() or 1
    touched_mb = n_sample * (16 * 128 * 2 + 128 * 8 / 5) / 1e6
    cfg, ls = pxo.cfg(), pxo.loss("cauchy", 0.25)
    pxo.ba_eval_batch(sub, cfg, ls, count=min(n_sample, 2048), n_threads=cores)       # warm-up / page-in
    t0 = time.perf_counter()
    passes = 0
    while True:
        pxo.ba_eval_batch(sub, cfg, ls, n_threads=cores)
        passes += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or passes >= 2000:
            break
    out = {"cpu_baseline": {
        "value": n_sample * passes / dt, "unit": "residual_blocks/s", "cores": cores, "kind": "port",
        "sample": "%d passes over the first %d observations (%d whole points) of the same workload (%.1f s), oracle C "
                  "restatement, materialised 128x(10+K) Jacobians + Cauchy loss, %d pthreads; touched working set "
                  "%.0f MB (stencils + references)" % (passes, n_sample, n_pts_s, dt, cores, touched_mb)}}
    # ---- one LM iteration on the host cores ------------------------------------------------------------------
    if lm_gauge is not None:
        pose_const, tmask, cmask, _ = lm_gauge
        best = None
        t0 = time.perf_counter()
        for _ in range(3):
            r = pxo.ba_lm_iteration_schur(sub, cfg, ls, pose_const, tmask, cmask, np.zeros(n_pts_s, np.uint8), radius=1e4,
                                          n_threads=cores, want_step=False)
            if best is None or r["total_ms"] < best["total_ms"]:
                best = r
            if time.perf_counter() - t0 > budget_s:
                break
        scale = n_obs / n_sample
        per_obs_ms = best["jacobian_eval_ms"] + best["schur_ms"] + best["backsub_ms"] + best["cost_eval_ms"]
        full_ms = per_obs_ms * scale + best["cholesky_ms"]
        out["cpu_baseline_lm_projected"] = {
            "value": 1e3 / full_ms, "unit": "LM iterations/s (PROJECTED from the sample, see `sample`)", "cores": cores, "kind": "port",
            "ms_per_iteration_projected": full_ms, "rc": best["rc"], "reduced_system": best["n_c"],
            "measured_on_sample_ms": {k: best[k] for k in ("jacobian_eval_ms", "schur_ms", "cholesky_ms", "backsub_ms",
                                                           "cost_eval_ms", "total_ms")},
            "sample": "one LM iteration (best of <= 3) on the first %d observations (%d whole points, all %d cameras) of "
                      "the same workload, oracle C restatement with OpenMP over observations / points (Schur "
                      "elimination with <= 32 private copies of S, blocked Cholesky); projected to the full problem: "
                      "per-observation stages x %.1f, Cholesky of the same %d x %d system unchanged; no inner iterations"
                      % (n_sample, n_pts_s, len(prob["image_camera"]), scale, best["n_c"], best["n_c"])}
    # ---- the reference's own AVX2 kernels (interpolation only) --------------------------------------------------
    try:
        ref = pxo.ref()
        arena = np.ascontiguousarray(sub["patches"]).view(np.uint16)
        rng = np.random.default_rng(0)
        rc = np.ascontiguousarray(rng.uniform(6.5, 8.5, (n_sample, 2)))              # around the patch centre
        idx = np.arange(n_sample, dtype=np.int64)
        chunks = np.array_split(np.arange(n_sample), cores)

        def work(c):
            if len(c) == 0:
                return 0.0
            return ref.pxo_ref_bicubic_many_half128(ctypes.c_void_p(arena.ctypes.data), ctypes.c_int64(len(c)), 16, 16,
                                                    ctypes.c_void_p(idx[c[0]:].ctypes.data),
                                                    ctypes.c_void_p(rc[c[0]:].ctypes.data), None)
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(work, chunks))                                              # warm-up
            t0 = time.perf_counter()
            passes = 0
            while True:
                list(pool.map(work, chunks))
                passes += 1
                dt = time.perf_counter() - t0
                if dt > budget_s / 2 or passes >= 2000:
                    break
        out["cpu_reference_kernel"] = {
            "value": n_sample * passes / dt, "unit": "bicubic interpolations/s (value + 2 derivatives, 128 channels)",
            "cores": cores, "kind": "reference-kernel",
            "sample": "%d passes over %d fp16 16x16x128 patches (%.1f s), the reference's cubic_hermite_spline_simd.h + "
                      "grid2d.h compiled in place (oracle/_ref), %d threads; INTERPOLATION ONLY -- no projection, "
                      "normalisation, Jacobian bridge or loss; touched working set %.0f MB"
                      % (passes, n_sample, dt, cores, n_sample * 4096 / 1e6)}
    except Exception as e:  # noqa: BLE001 -- oracle/_ref is optional (built from /root/reference in the build container)
        out["cpu_reference_kernel"] = {"value": None, "kind": "reference-kernel", "sample": "unavailable: %r" % (e,)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cams", type=int, default=200)
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--obs-per-point", type=int, default=5)
    ap.add_argument("--float-simd", action="store_true", help="InterpolationConfig.use_float_simd")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=131072,
                    help="observations of the CPU legs' sample (131072 -> 8.6 GB of patches, 0.5 GB touched: beyond the L3 of the box)")
    ap.add_argument("--lm-iters", type=int, default=10, help="LM iterations for the iters/s figure (0 = skip)")
    ap.add_argument("--no-ka", action="store_true", help="skip the keypoint-adjustment half of the metric (BASELINE configs[1])")
    ap.add_argument("--no-costmap", action="store_true", help="skip the cost-map extraction / cost-map BA figures")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default, BASELINE.json's target): --points points in total, sharded over the ranks; "
                         "weak: every rank owns --points points (N x the observations, cameras shared)")
    ap.add_argument("--patch-size", type=int, default=16, help="side of the square fp16 patches (16: pixsfm's default; 8: low_memory.yaml)")
    ap.add_argument("--preset", choices=("aachen",), default=None,
                    help="aachen: BASELINE.json configs[4]-shaped scene on ONE GPU -- 4000 cameras, 1M points, 5M observations, 8 x 8 "
                         "fp16 patches (82 GB; the reference runs Aachen with configs/low_memory.yaml: patch_size 8), iterative "
                         "solver by image count, cost-map strategy beside it; KA / CPU legs / API timing are skipped")
    ap.add_argument("--no-api-e2e", action="store_true",
                    help="skip the end-to-end timing of the drop-in API calls on host-resident inputs (tools/bench_api_e2e.py)")
    ap.add_argument("--linear-solver", default="auto", help="auto (by image count, bundle_optimizer.h:180-191) | direct | iterative")
    args = ap.parse_args()
    if args.preset == "aachen":
        args.cams, args.points, args.obs_per_point, args.patch_size = 4000, 1_000_000, 5, 8
        args.no_ka = args.no_api_e2e = args.no_cpu_baseline = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # PXR_BENCH_ONE_DEVICE=1 + PXR_BENCH_BACKEND=gloo: several ranks on ONE GPU, used to validate the
    # multi-process flow (sharding, barriers, max-over-ranks timing, all-reduce callback) on a 1-GPU box
    if os.environ.get("PXR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    # PXR_BENCH_FORCE_DIST=1 exercises the RCCL code path with a single rank (used to validate the
    # all-reduce plumbing on a 1-GPU box)
    dist_on = world > 1 or os.environ.get("PXR_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29531"
        backend = os.environ.get("PXR_BENCH_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pixsfm_amd import synthetic_gpu
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, make_loss

    C, PS = 128, args.patch_size
    # points (with their observations, patches and references) are the sharded unit; cameras and poses
    # are replicated.  weak: the scene grows with the ranks, strong: it is split.
    total_points = args.points * world if args.scaling == "weak" else args.points
    per = (total_points + world - 1) // world
    lo, hi = rank * per, min(total_points, (rank + 1) * per)
    prob, patches = synthetic_gpu.make_ba_problem_gpu(dev, n_cams=args.cams, n_points=total_points,
                                                      obs_per_point=args.obs_per_point, channels=C,
                                                      patch_size=PS, seed=2, point_range=(lo, hi),
                                                      # 8 x 8 patches leave +-2 px around the stencil: ~1 px initial errors
                                                      **(dict(rot_deg=0.04, trans=0.003, pt_sigma=0.003) if PS < 16 else {}))
    n_obs_local = len(prob["obs_image"])
    ctx = Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    collective = "none"
    if dist_on:
        from pixsfm_amd import parallel
        collective = "torch.distributed callback (%s)" % backend
        if backend == "nccl" and os.environ.get("PXR_BENCH_CALLBACK") != "1":
            ok = 1
            try:
                if world > 1:
                    parallel.init_native_comm(ctx)
                else:
                    ctx.comm_init(Context.comm_unique_id(), 0, 1)
                probe = ctx.to_device(np.full(4, float(rank + 1)), np.float64)      # one real all-reduce before relying on it
                ctx.allreduce_sum(probe)
                ctx.sync()
                if not np.array_equal(probe.download(), np.full(4, world * (world + 1) / 2.0)):
                    raise RuntimeError("native all-reduce returned %r" % (probe.download(),))
            except Exception as e:  # noqa: BLE001 -- keep the bench alive on the callback path
                ok = 0
                print("rank %d: native communicator unavailable (%r)" % (rank, e), file=sys.stderr)
            # every rank must take the same path: one rank on the callback while the others wait in ncclAllReduce would hang
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                collective = "native ncclAllReduce on the engine's stream (pxr_comm_init)"
            else:
                if ok:
                    ctx.comm_destroy()
                if world > 1:
                    ctx.comm_set_rank(rank, world)
                if rank == 0:
                    print("using the torch.distributed callback on every rank", file=sys.stderr)
    arena = PatchArena(ctx, n_obs_local, PS, PS, C, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    cfg = interp_cfg(use_float_simd=args.float_simd)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ba.eval(cfg, with_jacobian=True)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        ba.eval(cfg, with_jacobian=True)
    kernel_ms = ctx.timer_stop() / args.steps            # HIP events on the launch stream
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        ntot = torch.tensor([n_obs_local], dtype=torch.float64, device=dev)
        dist.all_reduce(ntot)
        n_obs_total = int(ntot.item())
    else:
        n_obs_total = n_obs_local
    # ---- what makes a multi-GPU run self-verifying: every rank's share and kernel time, the ranks the native communicator
    # really joined, and the time of the one collective of a direct LM iteration (the [S | rhs] all-reduce) on its own
    per_rank = None
    if dist_on:
        mine = torch.tensor([float(rank), float(n_obs_local), kernel_ms, float(local_rank)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu().numpy()
        comm_rank, comm_n = ctx.comm_rank()
        seen = torch.tensor([1.0 if collective.startswith("native") and comm_n == world else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(seen)
        per_rank = {"obs_per_gpu": [int(v) for v in every[:, 1]], "kernel_ms": [float(v) for v in every[:, 2]],
                    "kernel_ms_min": float(every[:, 2].min()), "kernel_ms_max": float(every[:, 2].max()),
                    "devices": [int(v) for v in every[:, 3]],
                    "nranks_seen": int(comm_n) if collective.startswith("native") else world,
                    "ranks_in_native_communicator": int(seen.item())}
    cost = ba.cost(make_loss("cauchy", [0.25]))
    if dist_on:                                   # cost of the whole (sharded) problem
        ctot = torch.tensor([cost], dtype=torch.float64, device=dev)
        dist.all_reduce(ctot)
        cost = ctot.item()

    # ---- second half of the metric: LM iterations / s on the same problem ------------------------
    # default gauge (bundle_adjustment/main.py:12-18) and refine flags (bundle_adjustment_options.h:66-76);
    # one iteration = linearise + Schur + Cholesky + back-substitution + evaluation at the trial point.
    lm, lm_extra = {}, {}
    if args.lm_iters > 0:
        from pixsfm_amd.engine import lm_options
        from pixsfm_amd.parallel import make_allreduce
        n_img = args.cams
        pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
        tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
        cmask = np.full(n_img, 0b0110, np.uint16)          # SIMPLE_RADIAL: refine f and k, keep cx, cy
        ptc = np.zeros(len(prob["xyz"]), np.uint8)
        # "lm": pixsfm's default BA configuration (use_inner_iterations = True, bundle_adjustment/main.py:43);
        # "lm_no_inner": the plain trust-region loop.  Same initial parameters for both.
        for key, inner in (("lm", True), ("lm_no_inner", False)):
            for name in ("qvec", "tvec", "cam_params", "xyz"):
                host = prob[name]
                if name == "cam_params":
                    host = np.zeros((len(prob["cam_model"]), 12)); host[:, :prob["cam_params"].shape[1]] = prob["cam_params"]
                ba.d[name].upload(host)
            barrier()
            if dist_on and collective.startswith("native") and "allreduce_ms" not in lm_extra:
                # the [S | rhs] buffer of the direct solver: (n_c + 1)^2 doubles, n_c = 8 per camera - 7 gauge columns
                n_c = 8 * n_img - 7
                buf = ctx.to_device(np.zeros((n_c + 1) * (n_c + 1)), np.float64)
                for _ in range(3):
                    ctx.allreduce_sum(buf)
                ctx.sync(); barrier()
                ctx.timer_start()
                for _ in range(10):
                    ctx.allreduce_sum(buf)
                lm_extra["allreduce_ms"] = ctx.timer_stop() / 10
                lm_extra["allreduce_bytes"] = int((n_c + 1) * (n_c + 1) * 8)
                del buf
                barrier()
            lm[key] = ba.solve(cfg, make_loss("cauchy", [0.25]), pose_const, tmask, cmask, ptc,
                               options=lm_options(max_iterations=args.lm_iters, use_inner_iterations=inner,
                                                  linear_solver=args.linear_solver),
                               allreduce=make_allreduce() if (dist_on and not collective.startswith("native")) else None)
            barrier()

    # ---- the reference's low-memory strategy on the same scene: cost-map extraction (one
    # HBM-bound pass over the feature arena) and the cost-map BA (3-channel maps, no reference descriptor)
    costmap = None
    if not args.no_costmap and world == 1:
        from pixsfm_amd.engine import lm_options
        for name in ("qvec", "tvec", "xyz"):
            ba.d[name].upload(prob[name])
        host = np.zeros((len(prob["cam_model"]), 12)); host[:, :prob["cam_params"].shape[1]] = prob["cam_params"]
        ba.d["cam_params"].upload(host)
        trivial = make_loss("trivial", [])
        cm = ba.extract_costmaps(trivial)                 # warm-up + the maps used below
        ctx.sync()
        reps = 20                                         # > 60 ms at 1M maps: the shader clock has settled (the kernel
        for _ in range(3):                                # runs at the package power limit)
            ba.extract_costmaps(trivial, out=cm)
        ctx.timer_start()
        for _ in range(reps):
            ba.extract_costmaps(trivial, out=cm)
        ex_ms = ctx.timer_stop() / reps
        ex_bytes = PS * PS * C * 2 + C * 8 + 3 * PS * PS * 2          # feature patch + reference in, 3-channel fp16 map out
        cba = ba.costmap_problem(cm)
        cfg_cm = interp_cfg(l2_normalize=False)                       # bundle_adjustment/main.py:270
        for _ in range(3):
            cba.eval(cfg_cm, with_jacobian=True)
        ctx.timer_start()
        for _ in range(20):
            cba.eval(cfg_cm, with_jacobian=True)
        ev_ms = ctx.timer_stop() / 20
        costmap = {"extract_ms": ex_ms, "maps_per_sec": n_obs_local / (ex_ms * 1e-3),
                   "extract_GBps": ex_bytes * n_obs_local / (ex_ms * 1e-3) / 1e9,
                   "extract_frac_of_peak": ex_bytes * n_obs_local / (ex_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "extract_bytes_per_map": ex_bytes, "kernel": "costmap_kernel_f16_split%s<f16> (%dx%dx128, gradients)" % ("8" if PS == 8 else "", PS, PS),
                   "map_arena_GB": n_obs_local * PS * PS * 3 * 2 / 1e9,
                   "extract_bound": "vector ALU at the package power limit: ~1800 4-cycle vector instructions per lane and map "
                                    "(the reference's double accumulation and half -> double conversions), 1366 W / shader clock "
                                    "2.10 GHz while it runs (profiles/r2_costmap_clock_power.txt, r2_costmap_split_pmc_sq.json)",
                   "eval_ms": ev_ms, "eval_blocks_per_sec": n_obs_local / (ev_ms * 1e-3)}
        if args.lm_iters > 0:
            n_img = args.cams
            pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
            tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
            for key, inner in (("lm", True), ("lm_no_inner", False)):
                for name in ("qvec", "tvec", "xyz"):
                    ba.d[name].upload(prob[name])
                ba.d["cam_params"].upload(host)
                s = cba.solve(cfg_cm, make_loss("cauchy", [0.25]), pose_const, tmask, np.full(n_img, 0b0110, np.uint16),
                              np.zeros(len(prob["xyz"]), np.uint8),
                              options=lm_options(max_iterations=args.lm_iters, use_inner_iterations=inner))
                costmap[key] = {"iters_per_sec": s["iterations"] / (s["total_ms"] * 1e-3), "iterations": s["iterations"],
                                "successful": s["num_successful"], "ms_per_iter": s["total_ms"] / max(1, s["iterations"]),
                                "initial_cost": s["initial_cost"], "final_cost": s["final_cost"], "inner_iterations": inner}

    # the other unit of work of the metric: one KA edge (A7).  BASELINE configs[1] (10k tracks / 100k keypoints /
    # 450k edges / 2000 sub-problems): per-edge residual+Jacobian rate and the whole bounded LM; with several ranks the
    # sub-problems are dealt to them (every rank takes part, rank 0 reports)
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_img = args.cams
        pc = np.zeros(n_img, np.uint8); pc[0] = 1
        tmk = np.zeros(n_img, np.uint8); tmk[1] = 1
        cpu_base = cpu_baseline(prob, patches, args.cpu_sample,
                                lm_gauge=(pc, tmk, np.full(n_img, 0b0110, np.uint16), None))
    ka_result = None
    if not args.no_ka:
        del ba, arena, patches
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_ka
        ka_result = bench_ka.run(device_index=local_rank, ctx=ctx, rank=rank, world=world)
    # ---- the drop-in calls end to end on HOST-resident inputs (rank 0, one GPU): what a pixsfm user pays, set-up included
    api_e2e = None
    if rank == 0 and world == 1 and not args.no_api_e2e:
        ba = arena = patches = None
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_api_e2e
        api_e2e = {"host_cores": os.cpu_count(),
                   "ba_host": bench_api_e2e.run_ba(dev, ctx, args.cams, total_points, args.obs_per_point, max(1, args.lm_iters), False)}
        if not args.no_ka:
            api_e2e["ka_host"] = bench_api_e2e.run_ka(dev, ctx, 10_000, 10, False)
        api_e2e["note"] = ("wall time of BundleAdjuster.create(conf).refine_multilevel(reconstruction, feature_manager) / "
                           "KeypointAdjuster...refine_multilevel(keypoints, feature_manager, graph) on host FeaturePatch objects "
                           "and Python scene objects, phases from pixsfm_amd.api._timing; building the inputs is not timed")

    if rank == 0:
        bpo = algorithmic_bytes_per_obs(C)
        # HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of
        # this same command (gpurun refuses/forbids mixing passes) and committed under profiles/.
        traffic, traffic_source = None, None
        for pmc_name in ("r2_ba_eval_pmc.json", "r1_ba_eval_pmc.json"):
            pmc_path = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc_path) and world == 1 and n_obs_total == 1_000_000 and not args.float_simd:
                with open(pmc_path) as fh:
                    traffic = json.load(fh).get("hbm_bytes_per_launch")
                traffic_source = "committed profile profiles/%s (separate rocprofv3 --pmc passes of this command; NOT " \
                                 "measured in this run)" % pmc_name
                break
        achieved = bpo * n_obs_local / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "featuremetric residuals+Jacobians evaluated/sec (1M obs)",
            "value": n_obs_total * args.steps / dt,
            "unit": "residual_blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 horizontal / f64 vertical+normalisation on f16 patches"
                     if not args.float_simd else "f32 splines / f64 normalisation on f16 patches",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json %s: synthetic %d cams / %d points / %d obs "
                                   "featuremetric BA residual+Jacobian evaluation, %d-ch fp16 %dx%d patches, "
                                   "SIMPLE_RADIAL, fused six-scalar Jacobian reduction"
                                   % ("configs[4] shape (Aachen scale) on one GPU" if args.preset == "aachen" else "configs[2]",
                                      args.cams, total_points, n_obs_total, C, PS, PS),
                       "n_obs": n_obs_total, "channels": C, "patch": PS,
                       "arena_GB": n_obs_total * PS * PS * C * 2 / 1e9,
                       "obs_per_gpu": n_obs_local,
                       "partition": "points sharded, cameras replicated" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes/launch, rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE",
                         "traffic_source": traffic_source,
                         "kernel": "ba_eval_kernel<f16,128,jac>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_obs": bpo},
            "initial_cost": cost,
        }
        for key, v in lm.items():
            out[key] = {"iters_per_sec": v["iterations"] / (v["total_ms"] * 1e-3), "iterations": v["iterations"],
                        "successful": v["num_successful"], "ms_per_iter": v["total_ms"] / max(1, v["iterations"]),
                        "setup_ms": v["setup_ms"], "initial_cost": v["initial_cost"], "final_cost": v["final_cost"],
                        "reduced_system": v["num_camera_unknowns"],
                        "linear_solver": "point Schur complement (LDS-privatised) + hand-written blocked dense Cholesky"
                                         if v["linear_solver"] == 1 else
                                         "implicit Schur complement, block-Jacobi preconditioned CG (ITERATIVE_SCHUR regime)",
                        "linear_iterations": v["linear_iterations"], "collective": collective,
                        "inner_iterations": key == "lm"}
        if cpu_base is not None:
            out.update(cpu_base)
        if ka_result is not None:
            out["ka"] = ka_result
        if costmap is not None:
            out["costmap"] = costmap
        if api_e2e is not None:
            out["api_e2e"] = api_e2e
        out["collective"] = collective
        if per_rank is not None:
            out["ranks"] = per_rank
        if "lm" in out:
            out["lm"].update(lm_extra)
        result_line = json.dumps(out)
    else:
        result_line = None
    # The JSON line must be the last thing on stdout: native libraries (RCCL's version banner, ...) write to the C
    # stdio buffer of every rank, which would otherwise be flushed at exit -- after the line.  Flush it now, wait for
    # all ranks, tear the process group down, then print.
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    sys.stdout.flush()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
