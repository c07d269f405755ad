"""GPU: the multi-rank paths of SURVEY 8e with the real solvers inside.  The GPU box has ONE MI355X and RCCL
refuses two ranks on one device, so: (a) two PROCESSES share the GPU, each solving its shard with pxr_ka_solve /
pxr_ba_solve, the BA collective being the callback form over gloo (tests/_multi_rank_worker.py) -- results must
match the single-process solve of the whole problem; (b) the native RCCL path (pxr_comm_*: dlopen of librccl,
ncclCommInitRank, ncclAllReduce on the context's stream) runs with a one-rank communicator."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _multi_rank_launch import run_ranks  # noqa: E402
import _multi_rank_worker as worker  # noqa: E402

pytestmark = pytest.mark.gpu


def test_ka_two_ranks_equal_one(ctx, tmp_path):
    from pixsfm_amd.engine import PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = worker.ka_problem()
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    total, _ = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, options=lm_options(parameter_tolerance=1e-5))
    want = ka.keypoints()
    res = run_ranks("ka", tmp_path, world=2)
    for r in res:
        # sub-problems are independent and the accumulation is order-independent (the deterministic default): a rank's solve
        # of its share IS the single-process solve of those sub-problems, bit for bit (the total cost is a host-side sum of
        # per-sub-problem costs in a different order)
        assert np.array_equal(r["kp"], want)
        assert abs(r["final_cost"][0] - total["final_cost"]) < 1e-12 * total["initial_cost"]
        assert abs(r["initial_cost"][0] - total["initial_cost"]) < 1e-12 * total["initial_cost"]
    assert np.array_equal(res[0]["kp"], res[1]["kp"])


@pytest.mark.parametrize("mode", ["ba_direct", "ba_iterative"])
def test_ba_two_ranks_equal_one(ctx, tmp_path, mode):
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = worker.ba_problem()
    gauge = worker.ba_gauge(prob)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    opts = dict(max_iterations=6)
    if mode == "ba_iterative":
        opts.update(linear_solver="iterative", eta=0.0, linear_r_tolerance=1e-13, max_linear_solver_iterations=1000)
    ref_obs, _ = ba.compute_references(interp_cfg(), make_loss("cauchy", [0.25]))       # at the initial parameters
    refs = ba.d["refs"].download()
    ba.d["refs"].upload(prob["refs"])
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**opts))
    q, t, k, X = ba.params()
    res = run_ranks(mode, tmp_path, world=2)
    for r in res:
        assert int(r["iterations"][0]) == s["iterations"] and int(r["successful"][0]) == s["num_successful"]
        if mode == "ba_direct":
            # every sum over observations / points / ranks is a sum of integers (fixed-point slots, limbs;
            # all-reduced as integers) -- the two-rank solve IS the one-rank solve, bit for bit
            assert r["final_cost"][0] == s["final_cost"] and r["initial_cost"][0] == s["initial_cost"]
            assert np.array_equal(r["q"], q) and np.array_equal(r["t"], t) and np.array_equal(r["k"], k) and np.array_equal(r["xyz"], X)
        assert abs(r["final_cost"][0] - s["final_cost"]) < 1e-8 * s["initial_cost"]
        assert abs(r["initial_cost"][0] - s["initial_cost"]) < 1e-12 * s["initial_cost"]
        assert np.abs(r["q"] - q).max() < 1e-8 and np.abs(r["t"] - t).max() < 1e-8
        assert np.abs(r["k"] - k).max() < 1e-6 and np.abs(r["xyz"] - X).max() < 1e-7
        # reference extraction on the shards (independent per point) + gather = the whole-problem extraction
        assert np.array_equal(r["ref_obs"], ref_obs)
        assert np.abs(r["refs"] - refs).max() < 1e-12
    # replicated parameters are bit-identical on the ranks (rank 0's camera step is broadcast)
    assert np.array_equal(res[0]["q"], res[1]["q"]) and np.array_equal(res[0]["t"], res[1]["t"])
    assert np.array_equal(res[0]["k"], res[1]["k"]) and np.array_equal(res[0]["xyz"], res[1]["xyz"])
    if mode == "ba_iterative":
        assert int(res[0]["linear_iterations"][0]) == int(res[1]["linear_iterations"][0]) > 0


def test_rank_counts_and_runs_give_the_same_bits(tmp_path):
    """The deterministic default: order- and partition-independent accumulation = the same bits on every rank, on every run
    AND for every rank count (two and three ranks sharing this box's GPU; the one-rank comparison is in the test above)."""
    for d in ("a", "b", "c"):
        (tmp_path / d).mkdir()
    first = run_ranks("ba_direct", tmp_path / "a", world=2)
    second = run_ranks("ba_direct", tmp_path / "b", world=2)
    three = run_ranks("ba_direct", tmp_path / "c", world=3)
    for key in ("q", "t", "k", "xyz", "final_cost", "iterations", "successful"):
        assert np.array_equal(first[0][key], first[1][key]), key                    # across the ranks
        assert np.array_equal(first[0][key], second[0][key]), key                   # across the runs
        assert np.array_equal(first[1][key], second[1][key]), key
        for r in three:
            assert np.array_equal(first[0][key], r[key]), key                       # across the rank counts


def test_floating_point_atomics_remain_as_the_opt_out(ctx, tmp_path, monkeypatch):
    """PXR_DETERMINISTIC=0: floating-point atomics, rank 0's replicated quantities broadcast -- the ranks agree bit for bit with
    each other, and with the deterministic solve to the tolerances of rounds 1-4."""
    monkeypatch.setenv("PXR_DETERMINISTIC", "0")
    res = run_ranks("ba_direct", tmp_path, world=2)
    monkeypatch.delenv("PXR_DETERMINISTIC")
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = worker.ba_problem()
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *worker.ba_gauge(prob), options=lm_options(max_iterations=6))
    q, t, k, X = ba.params()
    assert np.array_equal(res[0]["q"], res[1]["q"]) and np.array_equal(res[0]["xyz"], res[1]["xyz"])
    for r in res:
        assert int(r["iterations"][0]) == s["iterations"] and int(r["successful"][0]) == s["num_successful"]
        assert abs(r["final_cost"][0] - s["final_cost"]) < 1e-8 * s["initial_cost"]
        assert np.abs(r["q"] - q).max() < 1e-8 and np.abs(r["xyz"] - X).max() < 1e-7


def test_gradient_tolerance_is_decided_globally(ctx, tmp_path):
    """gradient_tolerance > 0 with sharded points: both ranks must stop at the same iteration (a rank-local
    max-norm would let one rank leave the loop while the other waits in the next all-reduce)."""
    res = run_ranks("ba_gradtol", tmp_path, world=2, timeout=300)
    assert int(res[0]["iterations"][0]) == int(res[1]["iterations"][0])
    assert int(res[0]["termination"][0]) == int(res[1]["termination"][0]) == 0       # CONVERGENCE by the gradient test
    assert int(res[0]["iterations"][0]) < 40


def test_native_rccl_communicator_single_rank():
    """pxr_comm_*: librccl resolved at run time, communicator of one rank on the context's device, in-place
    ncclAllReduce on the context's stream (identity for one rank), BA solve through the native path."""
    from pixsfm_amd import PixsfmHipError
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    c = Context(0)
    assert c.comm_rank() == (0, 1)
    uid = Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    c.comm_init(uid, 0, 1)
    assert c.comm_rank() == (0, 1)
    x = np.arange(1000, dtype=np.float64) * 0.5 - 3.0
    d = c.to_device(x)
    c.allreduce_sum(d)
    c.sync()
    assert np.array_equal(d.download(), x)
    with pytest.raises(PixsfmHipError):
        c.comm_init(uid, 0, 1)                      # a context owns at most one communicator
    with pytest.raises(PixsfmHipError):
        c.comm_set_rank(0, 2)                       # its rank is fixed by the communicator
    prob = worker.ba_problem()
    arena = PatchArena.from_numpy(c, prob["patches"], prob["corners"], prob["scales"])
    s = BAProblem(c, arena, prob).solve(interp_cfg(), make_loss("cauchy", [0.25]), *worker.ba_gauge(prob),
                                        options=lm_options(max_iterations=3))
    assert s["final_cost"] < s["initial_cost"]
    c.comm_destroy()
    assert c.comm_rank() == (0, 1)
    arena.close()
    c.close()


@pytest.mark.parametrize("mode", ["direct", "direct_inner", "iterative", "direct_fp_atomics"])
def test_forced_native_collective_equals_the_plain_solve_bit_for_bit(mode):
    """the multi-rank branch of pxr_ba_solve -- k_pack_upper -> ncclAllReduce(ncclInt64 / ncclFloat64) on the
    context's stream -> k_unpack_upper, the scalar limbs, diag(U) | g_c -- really RUNS on RCCL: a one-rank communicator with
    pxr_comm_force(ctx, 1).  A sum over one rank is the identity, so every parameter and both costs must equal the plain
    solve's bit for bit (deterministic default; with floating-point atomics: to rounding), and the context must have issued
    collectives."""
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    prob = worker.ba_problem()
    gauge = worker.ba_gauge(prob)
    opts = dict(max_iterations=6, use_inner_iterations=(mode == "direct_inner"))
    if mode == "iterative":
        opts.update(linear_solver="iterative", eta=0.0, linear_r_tolerance=1e-13, max_linear_solver_iterations=1000)
    out = []
    for forced in (False, True):
        c = Context(0)
        if mode == "direct_fp_atomics":
            c.deterministic = False
        if forced:
            c.comm_init(Context.comm_unique_id(), 0, 1)
            c.comm_force(True)
        arena = PatchArena.from_numpy(c, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(c, arena, prob)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**opts))
        calls, nbytes = c.comm_stats()
        out.append((s, ba.params(), calls, nbytes))
        arena.close()
        if forced:
            c.comm_destroy()
        c.close()
    (s0, p0, calls0, _), (s1, p1, calls1, bytes1) = out
    assert calls0 == 0 and calls1 >= 2 * s1["iterations"] and bytes1 > 0        # at least [S | rhs] (or CG vectors) + scalars per iteration
    assert s1["iterations"] == s0["iterations"] and s1["num_successful"] == s0["num_successful"]
    if mode != "iterative":
        n_c = s1["num_camera_unknowns"]
        assert s1["collective_kib"] == (n_c * (n_c + 3) // 2 * 8 + 1023) // 1024 and s0["collective_kib"] == 0
    if mode == "direct_fp_atomics":
        assert abs(s1["final_cost"] - s0["final_cost"]) < 1e-9 * s0["initial_cost"]
        for a, b in zip(p0, p1):
            assert np.abs(a - b).max() < 1e-7
    else:
        assert s1["initial_cost"] == s0["initial_cost"] and s1["final_cost"] == s0["final_cost"]
        for a, b in zip(p0, p1):
            assert np.array_equal(a, b)


def test_gathers_of_the_sharded_paths_run_on_rccl_with_one_rank(tmp_path):
    """KA keypoint gather and reference gather (parallel.gather_rows / allreduce_host) through torch.distributed's nccl backend
    (= RCCL) with a ONE-rank group in a fresh process: identity on the data."""
    import subprocess
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from pixsfm_amd import parallel\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))\n"
        "rows = np.arange(40, dtype=np.float64).reshape(20, 2) * 0.25 - 1\n"
        "ids = np.arange(20)[::-1].copy()\n"
        "full = parallel.gather_rows(rows, ids, 20)\n"
        "assert np.array_equal(full[ids], rows)\n"
        "refs = np.random.default_rng(0).normal(size=(7, 128))\n"
        "assert np.array_equal(parallel.gather_rows(refs, np.arange(7), 7), refs)\n"
        "assert np.array_equal(parallel.allreduce_host(np.array([3.0, 4.0])), [3.0, 4.0])\n"
        "dist.destroy_process_group()\n"
        "print('ok')\n") % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel-perfect-sfm_amd")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PXR_FORCE_COLLECTIVE"] = "1"
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout.split(), p.stderr[-3000:]           # (RCCL prints its banner after it)


def test_pixsfm_api_on_two_ranks(ctx, tmp_path, monkeypatch):
    """KeypointAdjuster / BundleAdjuster through the pixsfm-shaped API with torch.distributed initialised: two ranks
    (sharing the one GPU of this box, gloo) give what one process gives -- refined keypoints, references of all points,
    refined poses and points on EVERY rank."""
    monkeypatch.setenv("PXR_DEVICE", "0")                       # both ranks on the one GPU of this box
    one = worker.api_run()
    res = run_ranks("api", tmp_path, world=2)
    for r in res:
        assert np.array_equal(r["kp"], one["kp"]) and np.array_equal(r["xyz"], one["xyz"]) and np.array_equal(r["qvec"], one["qvec"])
        assert r["ba_cost"][1] == one["ba_cost"][1]                      # the deterministic default: the same bits for every rank count
        assert np.abs(r["kp"] - one["kp"]).max() < 1e-7
        assert np.abs(r["ka_cost"] - one["ka_cost"]).max() < 1e-9 * one["ka_cost"][0]
        assert np.array_equal(r["ref_ids"], one["ref_ids"]) and np.abs(r["ref_desc"] - one["ref_desc"]).max() < 1e-12
        assert int(r["ba_iters"][0]) == int(one["ba_iters"][0])
        assert np.abs(r["ba_cost"] - one["ba_cost"]).max() < 1e-8 * one["ba_cost"][0]
        assert np.abs(r["xyz"] - one["xyz"]).max() < 1e-7 and np.abs(r["qvec"] - one["qvec"]).max() < 1e-8
    assert np.array_equal(res[0]["xyz"], res[1]["xyz"]) and np.array_equal(res[0]["kp"], res[1]["kp"])


def _bench(n_gpus, extra_env=None, timeout=900):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_gpus), "--steps", "3", "--warmup", "1", "--cams", "24",
           "--points", "6000", "--lm-iters", "6", "--no-ka", "--no-costmap", "--no-cpu-baseline", "--no-api-e2e", "--no-telemetry"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]       # the JSON line and nothing else on stdout
    return json.loads(lines[0])


def test_plain_bench_command_on_two_ranks_is_the_one_rank_problem():
    """`python bench.py --gpus 2` with NO launcher around it (the shape of the driver's SCALE command): it spawns its two
    ranks itself (both on the one GPU of this box, gloo as the transport of the callback collective), the sharded scene is
    the one-rank scene bit for bit (same initial cost), the LM trajectory ends at the same cost, and the line says how many
    ranks took part and how many bytes the [S | rhs] collective moves."""
    one = _bench(1)
    two = _bench(2, {"PXR_BENCH_ONE_DEVICE": "1", "PXR_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["ranks"]["nranks_seen"] == 2
    assert two["ranks"]["obs_per_gpu"] == [15000, 15000] and two["config"]["n_obs"] == one["config"]["n_obs"] == 30000
    assert abs(two["initial_cost"] - one["initial_cost"]) <= 1e-12 * one["initial_cost"]
    for key in ("lm", "lm_no_inner"):
        assert two[key]["initial_cost"] == one[key]["initial_cost"]
        assert two[key]["final_cost"] == one[key]["final_cost"], key            # a future SCALE run verifies itself by `==`
        assert two[key]["iterations"] == one[key]["iterations"] and two[key]["successful"] == one[key]["successful"]
    n_c = two["lm"]["reduced_system"]
    assert two["lm"]["allreduce_bytes"] == n_c * (n_c + 3) // 2 * 8            # the packed upper triangle + rhs, not the square
    assert two["lm"]["collective_KiB_per_solve"] == (two["lm"]["allreduce_bytes"] + 1023) // 1024
    assert one["lm"]["collective_KiB_per_solve"] == 0 and two["lm"]["allreduce_ms"] > 0
