"""Summarise a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run: per kernel (name pattern) the number of
dispatches and the mean of every collected counter, as JSON.

    python tools/pmc_summary.py <rocprof output dir> <kernel name pattern> [out.json]
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root, pat = sys.argv[1], sys.argv[2]
    files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under " + root)
    per = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if pat not in name:
                continue
            per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[name] = {k: r.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                                "Accum_VGPR_Count", "SGPR_Count") if r.get(k) is not None}
    out = {}
    for name, counters in per.items():
        out[name] = {"dispatches": max(len(v) for v in counters.values()), "launch": meta[name],
                     "mean": {c: sum(v) / len(v) for c, v in sorted(counters.items())}}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
