"""Shared helpers of the test-suite (own code; no reference code)."""
import numpy as np


def relerr(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def widths(ncore, npad, width, factor):
    """Stretched cell widths: npad growing cells, ncore constant, npad growing."""
    pad = width * np.abs(factor) ** (np.arange(npad) + 1.0)
    return np.r_[pad[::-1], np.full(ncore, float(width)), pad]


def usable_cores(cap=16):
    """Threads worth using here: the affinity mask capped by the container's CPU quota (cgroup v2) and `cap`."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
            if q != 'max':
                n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))
