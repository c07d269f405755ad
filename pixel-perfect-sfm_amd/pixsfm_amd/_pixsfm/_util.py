"""`_pixsfm._util` (pixsfm/util/bindings.cc): host memory figures used by pixsfm's check_memory."""
import os


def _meminfo():
    out = {}
    try:
        with open("/proc/meminfo") as fh:
            for ln in fh:
                k, v = ln.split(":", 1)
                out[k] = int(v.split()[0]) * 1024
    except OSError:
        pass
    return out


def total_memory():
    return _meminfo().get("MemTotal", os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES"))


def free_memory():
    m = _meminfo()
    return m.get("MemAvailable", m.get("MemFree", 0))


def used_memory():
    return total_memory() - free_memory()
