"""``core``-shaped entry points: same names, positional arguments and in-place semantics as
the reference's ``emg3d.core`` (reference emg3d/core.py:45-49), executed by the HIP kernels.

This is boundary B1 of SURVEY.md section 8(b): host NumPy arrays in, host NumPy arrays
out, through the ``emg3d_core_*`` (host-pointer) flavour of the C ABI -- what
``from emg3d import core`` would become inside the reference. Every call
stages its arrays through HBM, so this module is for parity tests and literal drop-in use;
the solver (emg3d_amd/solver.py) keeps everything device-resident and calls the
``emg3d_dev_*`` flavour instead.

Ordering: the smoothers visit nodes/lines in the four-colour order described in
include/emg3d_amd.h (a valid Gauss-Seidel ordering, not the reference's lexicographic
one): per-sweep values differ from the reference, fixed points do not.

``restrict_weights`` is 1-D host arithmetic in the reference too (O(n), results are
uploaded once per level); it is restated here with NumPy.
"""
import ctypes

import numpy as np

from emg3d_amd import _lib

__all__ = [
    'amat_x', 'gauss_seidel', 'gauss_seidel_x', 'gauss_seidel_y',
    'gauss_seidel_z', 'blocks_to_amat', 'solve', 'restrict',
    'restrict_weights',
]


def __dir__():
    return __all__


_vp = ctypes.c_void_p


def _is_complex(a):
    if a.dtype == np.complex128:
        return 1
    if a.dtype == np.float64:
        return 0
    raise TypeError(f"emg3d_amd.core: dtype must be float64 or complex128, got {a.dtype}.")


def _p(a, dtype=None):
    """Pointer to a Fortran-contiguous array (the reference's views are)."""
    if dtype is not None and a.dtype != dtype:
        raise TypeError(f"emg3d_amd.core: expected dtype {dtype}, got {a.dtype}.")
    if not (a.flags.f_contiguous or a.flags.c_contiguous and a.ndim == 1):
        raise ValueError("emg3d_amd.core: arrays must be Fortran-contiguous.")
    return _vp(a.ctypes.data)


def _model_args(eta_x, eta_y, eta_z, zeta, hx, hy, hz, dt, keep):
    """Pointers for (eta_x, eta_y, eta_z, zeta, hx, hy, hz); aliasing of the etas is kept."""
    out = []
    seen = []
    for a in (eta_x, eta_y, eta_z):
        for src, conv in seen:
            if src is a:
                out.append(_vp(conv.ctypes.data))
                break
        else:
            conv = np.asfortranarray(a, dtype=dt)
            seen.append((a, conv))
            out.append(_vp(conv.ctypes.data))
    z = np.asfortranarray(zeta, dtype=np.float64)
    h = [np.ascontiguousarray(x, dtype=np.float64) for x in (hx, hy, hz)]
    keep.extend([c for _, c in seen] + [z] + h)
    return out + [_vp(z.ctypes.data)] + [_vp(x.ctypes.data) for x in h], [x.size for x in h]


def amat_x(rx, ry, rz, ex, ey, ez, eta_x, eta_y, eta_z, zeta, hx, hy, hz):
    """Residual without/with source term, ``r -= A e`` in place (emg3d/core.py:57-206)."""
    dt = rx.dtype
    keep = []
    margs, (nx, ny, nz) = _model_args(eta_x, eta_y, eta_z, zeta, hx, hy, hz, dt, keep)
    _lib.check(_lib.lib().emg3d_core_amat_x(
        _p(rx), _p(ry, dt), _p(rz, dt), _p(ex, dt), _p(ey, dt), _p(ez, dt), *margs,
        nx, ny, nz, _is_complex(rx)), 'emg3d_core_amat_x')


def _gs(lr, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu):
    dt = ex.dtype
    keep = []
    margs, (nx, ny, nz) = _model_args(eta_x, eta_y, eta_z, zeta, hx, hy, hz, dt, keep)
    _lib.check(_lib.lib().emg3d_core_gauss_seidel(
        lr, _p(ex), _p(ey, dt), _p(ez, dt), _p(sx, dt), _p(sy, dt), _p(sz, dt), *margs,
        nx, ny, nz, int(nu), _is_complex(ex)), 'emg3d_core_gauss_seidel')


def gauss_seidel(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu):
    """Point-block Gauss-Seidel smoother (emg3d/core.py:210-503)."""
    _gs(0, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu)


def gauss_seidel_x(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu):
    """Gauss-Seidel with line relaxation in x (emg3d/core.py:506-783)."""
    _gs(1, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu)


def gauss_seidel_y(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu):
    """Gauss-Seidel with line relaxation in y (emg3d/core.py:786-1068)."""
    _gs(2, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu)


def gauss_seidel_z(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu):
    """Gauss-Seidel with line relaxation in z (emg3d/core.py:1071-1348)."""
    _gs(3, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu)


def blocks_to_amat(amat, bvec, middle, left, rhs, im, nc):
    """Insert middle, left and rhs into the banded arrays (emg3d/core.py:1351-1477)."""
    dt = amat.dtype
    middle = np.ascontiguousarray(middle, dtype=dt)
    left = np.ascontiguousarray(left, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=dt)
    _lib.check(_lib.lib().emg3d_core_blocks_to_amat(
        _p(amat), _p(bvec, dt), _p(middle), _p(left), _p(rhs), int(im), int(nc), bvec.size,
        _is_complex(amat)), 'emg3d_core_blocks_to_amat')


def solve(amat, bvec):
    """Solve A x = b by the non-standard Cholesky (LDL^T) factorisation
    (emg3d/core.py:1481-1616); amat is replaced by its factors, bvec by x."""
    _lib.check(_lib.lib().emg3d_core_solve(_p(amat), _p(bvec, amat.dtype), bvec.size,
                                           _is_complex(amat)), 'emg3d_core_solve')


def restrict(crx, cry, crz, rx, ry, rz, wx, wy, wz, sc_dir):
    """Restriction of the residual to the coarse grid (emg3d/core.py:1620-2001)."""
    dt = crx.dtype
    nx, ny, nz = rx.shape[0], ry.shape[1], rz.shape[2]     # fine cells
    keep = [np.ascontiguousarray(w, dtype=np.float64) for w in (*wx, *wy, *wz)]
    _lib.check(_lib.lib().emg3d_core_restrict(
        _p(crx), _p(cry, dt), _p(crz, dt), _p(rx, dt), _p(ry, dt), _p(rz, dt),
        *[_vp(w.ctypes.data) for w in keep], nx, ny, nz, int(sc_dir), _is_complex(crx)),
        'emg3d_core_restrict')


def restrict_weights(nodes, cell_centers, h, cnodes, ccell_centers, ch):
    """Restriction weights (wl, w0, wr) of one direction, Mulder (2006) Eq. 9
    (emg3d/core.py:2004-2076): distances between fine and coarse dual-cell boundaries
    divided by the fine dual-cell widths; half widths at the two domain boundaries."""
    nodes, cell_centers, h = (np.asarray(a, dtype=float) for a in (nodes, cell_centers, h))
    cnodes, ccell_centers, ch = (np.asarray(a, dtype=float) for a in (cnodes, ccell_centers, ch))
    n = cnodes.size

    d = np.empty(n + 1)                       # dual widths around the coarse nodes' neighbours
    d[0], d[-1] = h[0] / 2, h[-1] / 2
    d[1:-1] = (h[0:2 * (n - 1):2] + h[1:2 * (n - 1):2]) / 2.

    wl = 1 / d[:-1]
    wl[0] *= (nodes[0] - h[0] / 2) - (cnodes[0] - ch[0] / 2)
    wl[1:] *= cell_centers[1:2 * (n - 1):2] - ccell_centers[:n - 1]

    w0 = np.ones(n)

    wr = 1 / d[1:]
    wr[-1] *= (cnodes[-1] + ch[-1] / 2) - (nodes[-1] + h[-1] / 2)
    wr[:-1] *= ccell_centers[:n - 1] - cell_centers[0:2 * (n - 1):2]
    return wl, w0, wr
