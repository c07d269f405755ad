"""Durations of the launches of one pxr_ka_solve from a rocprofv3 kernel trace (the two-launch schedule: solve kernel, order kernel,
solve kernel).   python tools/_ka_launch_durations.py <trace dir>"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from lm_timeline import load, short

rows = [r for r in load(sys.argv[1]) if "ka_" in r[2]]
for s, e, n in rows[-12:]:
    print("%10.1f us  %8.1f us  %s" % ((s - rows[-12][0]) * 1e-3, (e - s) * 1e-3, short(n)[:50]))
