"""A/B timing of the inner-iteration kernels on the bench scene: `lm` (10 iterations) with the Gram-matrix kernel and with
the packed kernel (PXR_INNER_PACKED=1).  python tools/_time_inner.py [points]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pts = sys.argv[1] if len(sys.argv) > 1 else "200000"
for mode in ("gram", "packed"):
    env = dict(os.environ)
    if mode == "packed":
        env["PXR_INNER_PACKED"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", pts, "--steps", "3", "--warmup", "1", "--no-ka", "--no-costmap",
                        "--no-cpu-baseline", "--no-api-e2e", "--no-telemetry"], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        print(mode, "FAILED", p.stderr[-2000:])
        continue
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(mode, json.dumps({k: d[k] for k in ("lm", "lm_no_inner")}))
