"""Timeline of an LM iteration from a rocprofv3 kernel trace: where the time of `ms_per_iter` goes that the kernels' own
durations do not explain (launch gaps, host polls).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python tools/_lm_solve_once.py
    python tools/lm_timeline.py gpurun_out/trace [n_last_iterations]

Takes the LAST solve of the trace (the `lm_no_inner` solve of the second pair), splits it at every k_pinv (one per LM attempt),
and prints per kernel: launches per iteration, busy time, and the idle time on the device BEFORE it (gap since the previous
kernel's end)."""
import collections
import csv
import glob
import json
import os
import sys


def load(path):
    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *kernel_trace.csv under " + path
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def short(name):
    name = name.split("(")[0]
    for pre in ("void pxr::", "pxr::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:60]


def main():
    rows = load(sys.argv[1])
    pinv = [i for i, r in enumerate(rows) if "k_pinv" in r[2]]
    # solves are separated by long pauses between k_pinv launches; take the last run of them
    runs, cur = [], [pinv[0]]
    for a, b in zip(pinv, pinv[1:]):
        if rows[b][0] - rows[a][0] > 20_000_000:      # 20 ms
            runs.append(cur); cur = []
        cur.append(b)
    runs.append(cur)
    last = runs[-1]
    want = int(sys.argv[2]) if len(sys.argv) > 2 else len(last) - 1
    first_it = max(0, len(last) - 1 - want)
    lo, hi = last[first_it], last[-1]
    its = len(last) - 1 - first_it
    span = rows[lo:hi]
    busy, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    prev_end = span[0][0]
    for s, e, n in span:
        k = short(n)
        busy[k] += e - s; cnt[k] += 1
        gap[k] += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    total = (rows[hi][0] - rows[lo][0]) / its
    tb, tg = sum(busy.values()) / its, sum(gap.values()) / its
    print("%d iterations (k_pinv to k_pinv): %.1f us per iteration = %.1f us kernels + %.1f us idle; %d launches per iteration"
          % (its, total / 1e3, tb / 1e3, tg / 1e3, sum(cnt.values()) / its))
    print("%-62s %8s %10s %12s" % ("kernel", "n/iter", "busy us", "idle before"))
    for k in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
        print("%-62s %8.1f %10.1f %12.1f" % (k, cnt[k] / its, busy[k] / its / 1e3, gap[k] / its / 1e3))
    # one iteration in order
    print("\none iteration, in order (start offset us, duration us, idle before us):")
    a, b = last[-2], last[-1]
    t0, prev_end = rows[a][0], rows[a][0]
    for s, e, n in rows[a:b]:
        print("  %9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3, short(n)))
        prev_end = max(prev_end, e)
    json.dump({"iterations": its, "us_per_iteration": total / 1e3, "kernel_us": tb / 1e3, "idle_us": tg / 1e3,
               "launches_per_iteration": sum(cnt.values()) / its,
               "per_kernel": {k: {"n": cnt[k] / its, "busy_us": busy[k] / its / 1e3, "idle_before_us": gap[k] / its / 1e3} for k in busy}},
              open(os.path.join(sys.argv[1], "lm_timeline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
