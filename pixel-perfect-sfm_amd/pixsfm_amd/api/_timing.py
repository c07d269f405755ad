"""Wall-clock phases of the drop-in API calls (dump of the Python objects, native problem construction, upload, reference
extraction, solve, write-back): off unless a measurement (tools/bench_api_e2e.py, bench.py --api-e2e) switches it on."""
import contextlib
import time

_acc = None


def start():
    global _acc
    _acc = {}


def stop():
    """The phases accumulated since start() as {name: seconds}; timing is off again afterwards."""
    global _acc
    acc, _acc = _acc, None
    return acc or {}


@contextlib.contextmanager
def phase(name):
    if _acc is None:
        yield
        return
    t0 = time.perf_counter()
    try:
        yield
    finally:
        _acc[name] = _acc.get(name, 0.0) + time.perf_counter() - t0
