"""Per-iteration time of the conjugate-gradient loop from a `rocprofv3 --kernel-trace --output-format csv` run: the median
distance between consecutive k_pre_apply launches of one solve.  usage: python tools/_cg_iter_time.py <dir>"""
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if "k_pre_apply" in r["Kernel_Name"]]
gaps = [b - a for a, b in zip(starts, starts[1:])]
inner = [g for g in gaps if g < 2_000_000]          # consecutive iterations of one solve (a new solve starts > 2 ms later)
print("k_pre_apply launches %d; consecutive-iteration distance: median %.1f us, min %.1f us, p90 %.1f us" %
      (len(starts), statistics.median(inner) / 1e3, min(inner) / 1e3, sorted(inner)[int(0.9 * len(inner))] / 1e3))
per = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("<")[0]
    if any(k in n for k in ("k_pre_apply", "k_cg_", "k_ublk", "k_pt_u", "k_img_wu")):
        per.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, v in per.items():
    print("  %-40s calls %5d  median %7.2f us" % (n[-40:], len(v), statistics.median(v) / 1e3))
