"""HBM bytes per launch of one kernel from two separate rocprofv3 passes (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`,
each with --kernel-trace --output-format csv): the file bench.py reads as `roofline.traffic`.

    python tools/pmc_traffic.py <fetch dir> <write dir> <kernel name pattern> <out.json> "<command the passes ran>" [commit]

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts KB and reports half of the bytes of 16 B / lane
coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE (KB) is taken as reported.  Only launches with with_jacobian
(the bench's step: the ones with the largest grid) are averaged -- pass the mangled template name to pick them."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PER = os.environ.get("PMC_PER_KERNEL")   # a kernel launched ONCE per unit of work: bytes are then per launch of THAT kernel
                                         # (pxr_ka_solve runs ka_solve_kernel twice per solve since round 6 and ka_order_kernel once)


def mean_counter(root, pattern, counter):
    vals, grids, units = [], [], 0
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            if pattern in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
                grids.append(int(r.get("Grid_Size") or 0))
            if PER and PER in r.get("Kernel_Name", ""):
                units += 1
    if not vals:
        raise SystemExit("no %s samples for %s under %s" % (counter, pattern, root))
    if PER:
        if not units:
            raise SystemExit("no launches of %s under %s" % (PER, root))
        return sum(vals) / units, units
    # only the launches of the bench's step: the LARGEST grid (the bench also launches the kernel on the 1/8 problem of its scaling
    # model and inside the texel-evaluation solves; until round 6 those were averaged in, which read 0.84x instead of 0.93x)
    top = max(grids)
    vals = [v for v, g in zip(vals, grids) if g == top]
    return sum(vals) / len(vals), len(vals)


def blob_hash(path):
    """what `git hash-object` prints (the box that collects the counters has no .git)"""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


fetch_dir, write_dir, pattern, out, cmd = sys.argv[1:6]
commit = sys.argv[6] if len(sys.argv) > 6 else "unrecorded"
sources = sys.argv[7:]            # the kernel's source files (paths relative to the repository root): their hashes gate staleness
f, nf = mean_counter(fetch_dir, pattern, "FETCH_SIZE")
w, nw = mean_counter(write_dir, pattern, "WRITE_SIZE")
rd, wr = 2.0 * f * 1024.0, w * 1024.0
res = {"kernel": pattern, "FETCH_SIZE_KB_per_launch_raw": f, "n_fetch_samples": nf, "WRITE_SIZE_KB_per_launch_raw": w,
       "n_write_samples": nw,
       "correction": "gfx950: read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
       "measured_at_commit": commit, "per": ("one launch of %s (all launches of the kernel summed)" % PER) if PER else "one launch of the kernel",
       "source_hashes": {os.path.relpath(s, ROOT): blob_hash(s) for s in sources},
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- " + cmd,
                    "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- " + cmd]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
