"""Merge the per-pass summaries of tools/pmc_collect.sh (<pass>_<kernel>.json, pass in fetch / write / sq / lds) into one file:
   python tools/pmc_merge.py <pmc dir> <out.json> <commit>"""
import glob
import json
import os
import sys

src, out, commit = sys.argv[1], sys.argv[2], sys.argv[3]
kernels = {}
for path in sorted(glob.glob(os.path.join(src, "*_*.json"))):
    base = os.path.basename(path)[:-5]
    ps = base.split("_", 1)[0]
    if ps not in ("fetch", "write", "sq", "lds"):
        continue
    key = base.split("_", 1)[1]
    try:
        data = json.load(open(path))
    except ValueError:
        continue
    for name, d in data.items():
        k = kernels.setdefault(key, {}).setdefault(name, {"launch": d.get("launch"), "dispatches": {}, "mean": {}})
        k["dispatches"][ps] = d.get("dispatches")
        k["mean"].update(d.get("mean", {}))
json.dump({"measured_at_commit": commit,
           "command": "rocprofv3 --pmc <one group per pass> --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 1 "
                      "--no-cpu-baseline --no-api-e2e --no-costmap --no-telemetry  (tools/pmc_collect.sh; passes: FETCH_SIZE | WRITE_SIZE | "
                      "eight SQ counters | LDS / VMEM / SALU / TCC counters)",
           "note": "means per dispatch; FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950: a wide coalesced read is tallied at half its "
                   "bytes -- MI355X_MICROARCH.md; not applied here)",
           "kernels": kernels}, open(out, "w"), indent=1)
print("wrote", out, "with", sum(len(v) for v in kernels.values()), "kernel instantiations")
