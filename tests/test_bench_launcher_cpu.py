"""bench.py as its own launcher (no GPU needed): a plain `python bench.py --gpus N` spawns N ranks, re-prints rank 0's JSON
line last, and a failing rank stops the others and makes the command fail (the driver's SCALE command is
the plain one; round 3 exited at once with '--gpus 8 but WORLD_SIZE=1').  PXR_BENCH_SELFTEST makes ranks succeed / fail /
hang before any GPU work."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(n, selftest, timeout=120, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PXR_BENCH_SELFTEST"] = selftest
    env.update(extra_env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p, time.time() - t0


def test_plain_command_spawns_its_ranks_and_reprints_rank0s_line_last():
    p, _ = _run(3, "ok:0,ok:1,ok:2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                       # ONLY rank 0's line on stdout
    assert json.loads(lines[0]) == {"selftest": True, "rank": 0}
    assert '"rank": 1' in p.stderr and '"rank": 2' in p.stderr      # the other ranks' stdout goes to stderr


def test_a_failing_rank_stops_the_others_and_fails_the_command():
    # rank 1 exits with code 3 at once, rank 0 would sleep for 10 minutes (a rank left alone in a collective)
    p, dt = _run(2, "fail:1,hang:0")
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert dt < 60, dt
    assert p.stdout.strip() == ""
    assert "rank 1 exited with code 3" in p.stderr


def test_launcher_times_out():
    p, dt = _run(2, "hang:0,hang:1", extra_env={"PXR_BENCH_LAUNCH_TIMEOUT": "2"})
    assert p.returncode != 0 and dt < 60
    assert "timed out" in p.stderr


def test_under_torchrun_the_process_is_a_rank_not_a_launcher():
    # RANK in the environment: no spawning; WORLD_SIZE must agree with --gpus
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", PXR_BENCH_SELFTEST="")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-ka"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (p.stderr + p.stdout)


def _newest_full_result():
    """The newest committed FULL bench result (profiles/r*_bench_detail.json, or round 4's one-line record)."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_detail.json")) + [os.path.join(ROOT, "profiles", "r4_bench_n1.json")]:
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and os.path.exists(path) and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    with open(best[1]) as fh:
        return json.load(fh)


def test_the_printed_line_fits_the_drivers_tail_and_ends_with_metric_2():
    """the 14.9 KB line overflowed the driver's tail, so LM iterations/s (the second half of BASELINE's
    metric) and the clocks were invisible.  The line is now the compact form of bench_detail.json: < 6000 bytes, with
    `telemetry`, `lm_no_inner` and `lm` LAST (inside the final 2 KB that even the shortest tail keeps)."""
    sys.path.insert(0, ROOT)
    import bench
    full = _newest_full_result()
    full["detail_file"] = "/root/repo/bench_detail.json"
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < bench.LINE_BUDGET_BYTES, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in back, k
    assert back["value"] == float("%.7g" % full["value"])
    assert set(back["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    tail = line[-2000:]
    assert '"lm":{' in tail and '"lm_no_inner":{' in tail and '"ms_per_iter"' in tail
    assert '"telemetry":{' in line[-4000:]
    assert list(back)[-1] == "lm" and list(back)[-2] == "lm_no_inner"
    assert back["lm"]["ms_per_iter"] == float("%.5g" % full["lm"]["ms_per_iter"])
    assert "ka" in back and "kernel_ms" in back["ka"]["solve"] and "frac" in back["ka"]["roofline"]


def test_plain_command_at_eight_ranks():
    """The driver's SCALE command at its largest size: `python bench.py --gpus 8` spawns eight ranks (RANK / LOCAL_RANK 0..7, one
    rendezvous port) and prints rank 0's line only."""
    p, _ = _run(8, ",".join("ok:%d" % r for r in range(8)), timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"selftest": True, "rank": 0}
    for r in range(1, 8):
        assert '"rank": %d' % r in p.stderr


def test_a_leg_that_never_returns_still_leaves_rank_0s_line_and_exit_code_0():
    """The legs after the headline measurement use collectives that development never ran on a second GPU.  bench.Watchdog: when
    they do not come back in time, rank 0 prints the line with what was measured and every rank exits 0 -- here every rank arms it
    (0.5 s) and then sleeps."""
    p, dt = _run(2, "watchdog:0,watchdog:1", timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert dt < 60
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"selftest": True, "watchdog": True, "rank": 0}
    assert "did not finish" in p.stderr


def test_the_watchdogs_partial_line_carries_the_contract():
    """What rank 0 prints when the watchdog fires is built from bench.headline(): every field of the contract + roofline."""
    sys.path.insert(0, ROOT)
    import types
    import bench
    args = bench.parse_args(["--gpus", "8"])
    job = types.SimpleNamespace(world=8, collective="native ncclAllReduce on the engine's stream (pxr_comm_init)")
    part = bench.headline(job, args, dt=0.0052, kernel_ms=0.104, n_obs_total=1_000_000, n_obs_local=125_000, total_points=200_000,
                          cost=32262.3, jac_ms=None)
    part["watchdog"] = "x"
    line = bench.compact_line(part)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 8 and line["config"]["obs_per_gpu"] == 125_000
    assert abs(line["value"] - 1_000_000 * args.steps / 0.0052) / line["value"] < 1e-6
