"""Stand-in for empymod: only the EMArray ndarray subclass the reference imports."""
import numpy as np

__version__ = "0.0.stub"


class EMArray(np.ndarray):
    def __new__(cls, data, dtype=None):
        return np.asarray(data, dtype=dtype).view(cls)

    def amp(self):
        return np.abs(self.view(np.ndarray))

    def pha(self, deg=False, unwrap=True, lag=True):
        pha = np.angle(self.view(np.ndarray))
        if unwrap and self.ndim > 0:
            pha = np.unwrap(pha)
        if not lag:
            pha = -pha
        if deg:
            pha = pha * 180 / np.pi
        return pha
