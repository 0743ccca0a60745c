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
