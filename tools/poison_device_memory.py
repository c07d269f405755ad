"""Debug helper: dirty a large part of the device memory (0xFF bytes -> NaN doubles, -1 indices), release it, then run
the GPU tests in the SAME process, so that every kernel that reads memory it never wrote sees garbage instead of the
zeros a fresh box usually hands out.  python tools/poison_device_memory.py [GiB] [pytest args...]"""
import sys

import torch

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunks = [torch.full((1 << 30,), 255, dtype=torch.uint8, device="cuda") for _ in range(gib)]
torch.cuda.synchronize()
del chunks
torch.cuda.empty_cache()
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:] or ["tests", "-x", "-q", "-m", "gpu"]))
