"""CPU legs of bench.py (TEST INFRASTRUCTURE ONLY -- imported by bench.py's cpu_baseline() and by tests/).

Every leg is timed INSIDE C by the persistent-thread harness of pxo_bench_harness.h (pinned threads created once,
thread-local first-touched copies of the sample, >= `min_seconds` of back-to-back passes between two barriers); this
module only sweeps the thread count and reports.  Legs:

  ba_eval_port       kind "port": the oracle's C restatement with analytic Jacobians + loss (oracle/pxo_cpubench.c);
  ka_solve_port      kind "port": keypoint-adjustment sub-problems, one single-threaded solve per task, a pool of threads
                     over the tasks (keypoint_adjustment/main.py:66-80).
There is no kind "reference" leg: the reference's C++ path is unbuildable in this image (rounds 1-5 timed a build
of its functors over builder-written Eigen / Jet / COLMAP stand-ins and called it "reference" -- removed in round 6).
"""
import ctypes as C
import os

import numpy as np

import pxo

HERE = os.path.dirname(os.path.abspath(__file__))
def cpu_topology():
    """(logical CPUs this process may run on, physical cores among them, NUMA nodes)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as fh:
                cores.add(fh.read().strip())
        except OSError:
            cores.add(str(c))
    try:
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        nodes = 1
    return len(allowed), max(1, len(cores)), max(1, nodes)


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable.
    The GPU boxes of this project are containers on a shared 256-thread node with `1600000 100000`: 16 CPUs."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = float(fq.read()), float(fp.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def usable_cpus(logical=None):
    """min(logical CPUs of the affinity mask, cgroup quota rounded up)."""
    logical = logical or cpu_topology()[0]
    quota = cgroup_cpu_quota()
    return logical if quota is None else max(1, min(logical, int(-(-quota // 1))))


def thread_counts(logical):
    """{n/4, n/2, n, 2n} threads for n = the CPUs the process can really use (the cgroup quota caps the affinity mask's
    count), plus the full logical count when that is larger -- so that a quota shows as a plateau, not as a guess."""
    n = usable_cpus(logical)
    return sorted({max(1, n // 4), max(1, n // 2), n, 2 * n, logical})


_host = None


def host_probe(counts=None):
    """What the box gives this process: the speed-up of a pure-ALU spin loop (registers only) at each thread count over one
    thread -- "cores' worth" of arithmetic; a shared node, a CPU quota or SMT show up here -- plus the cgroup CPU limit, its
    throttling counters and the load average.  Cached."""
    global _host
    if _host is not None and counts is None:
        return _host
    logical, physical, nodes = cpu_topology()
    fn = pxo.lib().pxo_bench_spin
    fn.restype = C.c_int

    def rate(threads):
        out = _out4()
        iters = 2_000_000
        if fn(C.c_int64(iters), int(threads), C.c_double(0.1), out):
            raise RuntimeError("pxo_bench_spin failed")
        return threads * iters * out[1] / out[0]

    def read(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except OSError:
            return None
    before = read("/sys/fs/cgroup/cpu.stat")
    one = rate(1)
    spin = {str(t): rate(t) / one for t in (counts or thread_counts(logical)) if t > 1}
    rep = {"logical_cpus": logical, "physical_cores": physical, "numa_nodes": nodes, "cgroup_cpu_quota": cgroup_cpu_quota(),
           "spin_speedup_over_one_thread": spin, "cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"),
           "cgroup_cpu_stat_before": before, "cgroup_cpu_stat_after": read("/sys/fs/cgroup/cpu.stat"),
           "loadavg": read("/proc/loadavg"),
           "note": "spin = 8 independent chains of dependent fp64 multiply-adds per thread, no memory traffic, same harness"}
    if counts is None:
        _host = rep
    return rep


def sweep(run, n_items, unit, min_seconds=0.25, single_items=None, counts=None):
    """run(n, n_threads, min_seconds) -> (seconds, passes, calib_seconds, pinned) over the first n items.
    Returns the report dict: best rate over the thread counts, the single-thread rate (on `single_items` items), the
    scaling efficiency and the harness_limited guard (best < 0.4 x physical cores x single-thread rate)."""
    logical, physical, nodes = cpu_topology()
    counts = counts or thread_counts(logical)
    n1 = int(min(n_items, single_items or max(64, n_items // max(1, logical))))

    def long_enough(n, t, want):
        """A timed region that really lasted `want` seconds.  The pass count comes from the calibration pass; under a cgroup CPU
        quota an over-subscribed calibration pass can be throttled (slow) and the timed passes then run inside one un-throttled
        burst of the next 100 ms period -- 256 threads on a 16-CPU quota reported 8 x the 16-thread rate that way.  So a region
        shorter than 0.6 x want is repeated with proportionally more passes asked for (up to three times)."""
        ask = want
        for _ in range(3):
            sec, passes, calib, pinned = run(n, t, ask)
            if want <= 0.0 or sec >= 0.6 * want:
                break
            ask = ask * max(2.0, 1.5 * want / max(sec, 1e-6))
        return sec, passes, calib, pinned
    sec, passes, _, pinned = long_enough(n1, 1, max(min_seconds, 0.2))
    single = n1 * passes / sec
    rates = {}
    for t in counts:
        if t == 1:
            rates[1] = single
            continue
        sec, passes, calib, pinned = long_enough(n_items, t, min_seconds)
        rates[t] = n_items * passes / sec
    best_t = max(rates, key=rates.get)
    best = rates[best_t]
    quota = cgroup_cpu_quota()
    cores_avail = physical if quota is None else min(physical, quota)        # cores' worth of time this process can get
    ideal = min(best_t, cores_avail) * single
    try:
        spin = host_probe()["spin_speedup_over_one_thread"].get(str(best_t))
    except Exception:  # noqa: BLE001
        spin = None
    return {"value": best, "unit": unit, "cores": best_t, "single_thread": single, "logical_cpus": logical,
            "physical_cores": physical, "numa_nodes": nodes, "sweep": {str(t): rates[t] for t in sorted(rates)},
            "scaling_efficiency": best / ideal if ideal > 0 else None,
            # the same ratio against what a pure-ALU loop gets out of `cores` threads on this box (host_probe): near 1 = the
            # leg scales like arithmetic does here, the box (not the harness) sets the ceiling
            "speedup_over_single": best / single, "host_spin_speedup_at_cores": spin,
            "scaling_vs_host_spin": (best / single) / spin if spin else None,
            "cgroup_cpu_quota": quota, "cores_available": cores_avail,
            # the guard of VERDICT r3 next-2, against the cores this process can actually get (quota-capped)
            "harness_limited": bool(best < 0.4 * cores_avail * single) if logical > 1 else False,
            "pinned_threads": bool(pinned), "timed_in": "C (persistent pinned pthreads, barriers, thread-local copies)",
            "n_items": int(n_items)}


def _out4():
    return (C.c_double * 8)()


def ba_eval_port(sub, config, ls, min_seconds=0.25):
    """Residual blocks / s of the oracle port on the sample `sub` (a BA problem dict with 'patches')."""
    b, keep = pxo.ba_batch(sub)
    fn = pxo.lib().pxo_bench_ba_eval
    fn.restype = C.c_int

    def run(n, threads, secs):
        out = _out4()
        rc = fn(C.byref(b), C.byref(config), C.byref(ls), C.c_int64(n), int(threads), C.c_double(secs), 1, out)
        if rc:
            raise RuntimeError("pxo_bench_ba_eval failed (%d)" % rc)
        return out[0], out[1], out[2], out[3]
    rep = sweep(run, b.n_obs, "residual_blocks/s", min_seconds)
    rep["kind"] = "port"
    return rep


def _ka_csr(problem):
    node_problem = np.asarray(problem["node_problem"], np.int64)
    n_prob = int(node_problem.max()) + 1
    order = np.argsort(node_problem, kind="stable").astype(np.int32)
    node_ptr = np.searchsorted(node_problem[order], np.arange(n_prob + 1)).astype(np.int64)
    edge_problem = node_problem[np.asarray(problem["edge_src"], np.int64)]
    eorder = np.argsort(edge_problem, kind="stable").astype(np.int32)
    edge_ptr = np.searchsorted(edge_problem[eorder], np.arange(n_prob + 1)).astype(np.int64)
    return n_prob, node_ptr, np.ascontiguousarray(order), edge_ptr, np.ascontiguousarray(eorder)


def ka_solve_port(problem, config, ls, bound=4.0, opts=None, counts=None):
    """Sub-problems / s of the oracle's bounded LM, one single-threaded solve per task (the sample `problem`: a KA problem
    dict with 'patches' and 'node_problem')."""
    import pxo_ka
    keep = {}

    def arr(name, dt):
        keep[name] = np.ascontiguousarray(problem[name], dtype=dt)
        return keep[name].ctypes.data
    kp = np.array(problem["kp"], dtype=np.float64, order="C", copy=True)
    patches = np.ascontiguousarray(problem["patches"])
    _, H, W, ch = patches.shape
    b = pxo_ka.KaBatch(len(kp), kp.ctypes.data, arr("node_patch", np.int64), arr("node_const", np.uint8),
                       len(problem["edge_src"]), arr("edge_src", np.int32), arr("edge_dst", np.int32), arr("edge_w", np.float64),
                       patches.ctypes.data, pxo._NP2DT[patches.dtype], H, W, ch, arr("corners", np.int32),
                       arr("scales", np.float64), 0, None, None, None)
    n_prob, node_ptr, nodes, edge_ptr, edges = _ka_csr(problem)
    opts = opts or pxo.lm_options(parameter_tolerance=1e-5)
    fn = pxo.lib().pxo_bench_ka_solve
    fn.restype = C.c_int
    iters = {}

    def run(n, threads, secs):
        out = _out4()
        rc = fn(C.byref(b), int(n), pxo._p(node_ptr), pxo._p(nodes), pxo._p(edge_ptr), pxo._p(edges), C.byref(config), C.byref(ls),
                C.c_double(bound), C.byref(opts), int(threads), C.c_double(0.0), out)
        if rc:
            raise RuntimeError("pxo_bench_ka_solve failed (%d)" % rc)
        iters[threads] = out[4] / (out[1] + 1) / max(1, n)
        return out[0], out[1], out[2], out[3]
    logical, physical, _ = cpu_topology()
    rep = sweep(run, n_prob, "sub-problems/s", 0.0, single_items=max(2, min(n_prob, 8)),
                counts=counts or [t for t in thread_counts(logical) if t <= n_prob] or [1])
    rep["kind"] = "port"
    rep["lm_iterations_per_sub_problem"] = iters.get(rep["cores"])
    rep["edges"] = int(len(problem["edge_src"]))
    return rep
