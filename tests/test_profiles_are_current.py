"""`roofline.traffic` of the bench line is read from a committed counter file (separate rocprofv3 --pmc passes cannot run inside
bench.py).  The file records the git blob hashes of the kernel's sources at the time of the measurement (tools/pmc_traffic.py):
if the kernel changes and the counters are not collected again, this test fails instead of the line silently carrying stale
traffic."""
import glob
import hashlib
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def _newest(pattern):
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


@pytest.mark.parametrize("pattern", ["r*_ba_eval_pmc.json", "r*_ka_solve_traffic.json"])
def test_committed_counter_file_was_measured_on_the_kernel_that_ships(pattern):
    path = _newest(pattern)
    assert path is not None, "no counter file " + pattern
    rec = json.load(open(path))
    hashes = rec.get("source_hashes")
    assert hashes, "%s carries no source hashes: collect the counters again with tools/pmc_collect.sh" % os.path.basename(path)
    for rel, want in hashes.items():
        assert _blob_hash(os.path.join(ROOT, rel)) == want, (
            "%s changed since %s was measured (commit %s): run tools/pmc_collect.sh on the GPU box and commit the new file"
            % (rel, os.path.basename(path), rec.get("measured_at_commit")))
    assert rec["hbm_bytes_per_launch"] > 0


def test_traffic_tool_averages_the_largest_grid_only(tmp_path):
    """tools/pmc_traffic.py: the bench launches the headline kernel at full size and on the 1/8 problem of its scaling model; only
    the full-size launches are the step (rounds 4-5 averaged all of them: 0.84x instead of 0.93x of the algorithmic bytes)."""
    import csv
    import json
    import subprocess
    import sys
    rows = [("K_full", 1000, 100.0), ("K_full", 1000, 102.0), ("K_full", 125, 12.0), ("other", 1000, 7.0)]
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        d = tmp_path / name
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            for k, g, v in rows:
                w.writerow([k, g, counter, v])
    out = tmp_path / "out.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "fetch"), str(tmp_path / "write"),
                           "K_full", str(out), "cmd", "abc1234"], stdout=subprocess.DEVNULL)
    rec = json.load(open(out))
    assert rec["n_fetch_samples"] == 2 and rec["FETCH_SIZE_KB_per_launch_raw"] == 101.0
    assert rec["hbm_read_bytes_per_launch"] == 2 * 101.0 * 1024 and rec["hbm_write_bytes_per_launch"] == 101.0 * 1024
