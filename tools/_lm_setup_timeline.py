"""The kernels of the LAST pxr_ba_solve of a rocprofv3 kernel trace from its first launch (k_count_indices) to the first k_pinv:
set-up, the evaluation at the initial point and the first linearisations, in order.
    python tools/_lm_setup_timeline.py <trace dir>"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from lm_timeline import load, short

rows = load(sys.argv[1])
starts = [i for i, r in enumerate(rows) if "k_count_indices" in r[2]]
lo = starts[-1]
hi = next(i for i in range(lo, len(rows)) if "k_pinv" in rows[i][2])
t0 = rows[lo][0]
prev = t0
print("start us   dur us  idle before  kernel")
for s, e, n in rows[lo:hi + 1]:
    print("%8.1f %8.1f %8.1f  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, max(0, s - prev) * 1e-3, short(n)))
    prev = max(prev, e)
