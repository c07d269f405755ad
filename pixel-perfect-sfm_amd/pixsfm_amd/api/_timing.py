"""Wall-clock phases of the drop-in API calls (dump of the Python objects, native problem construction, upload, reference
extraction, solve, write-back): off unless a measurement (tools/bench_api_e2e.py, bench.py --api-e2e) switches it on."""
import contextlib
import gc
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


@contextlib.contextmanager
def gc_paused():
    """The cyclic garbage collector off for the duration of a refinement call.  A scene of a million observations is a few
    million live Python objects (points2D, track elements, patch objects); the lists the host layer builds over them trigger
    full collections that traverse all of them -- measured: 0.3-0.8 s per call at BASELINE configs[2], as much as the walk over
    the scene itself.  Nothing the call allocates forms reference cycles that need collecting before it returns."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
