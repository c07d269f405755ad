"""Per-launch durations of the Cholesky step kernels from a `rocprofv3 --kernel-trace --output-format csv` run of
tools/_time_chol.py (last factorisation): usage python tools/_chol_trace.py <dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
steps = [r for r in rows if "k_chol" in r["Kernel_Name"] or "k_aug" in r["Kernel_Name"]]
last = len(steps) - 1 - [("k_aug_pack" in r["Kernel_Name"]) for r in reversed(steps)].index(True)
seq = steps[last:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = None
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    print("%-22s grid %6s  start %8.2f us  dur %7.2f us  gap %6.2f us" % (r["Kernel_Name"].split("(")[0][-22:], r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                                                                      (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3))
    prev_end = e
print("total %.2f us" % ((int(seq[-1]["End_Timestamp"]) - t0) / 1e3))
