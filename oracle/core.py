"""ORACLE -- TEST INFRASTRUCTURE ONLY (parity pinned, see oracle/core_oracle.c).

``core``-shaped Python front-end of the C restatement in ``core_oracle.c``: the
same nine functions, positional arguments and in-place semantics as the
reference's ``emg3d.core`` (reference emg3d/core.py:45-49), so that tests can be
written like the reference's ``tests/test_core.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module. The product package ``emg3d_amd`` never does.

Extra keyword ``order`` on the smoothers: 0 = the reference's lexicographic
order (default), 1 = the four-colour order of the HIP kernels.
"""
import ctypes
import os
import subprocess

import numpy as np

__all__ = [
    'amat_x', 'gauss_seidel', 'gauss_seidel_x', 'gauss_seidel_y',
    'gauss_seidel_z', 'blocks_to_amat', 'solve', 'restrict',
    'restrict_weights',
]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, 'liboracle.so')


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ('core_oracle.c', 'core_generic.h')]
    stale = (not os.path.exists(_LIBPATH) or
             any(os.path.getmtime(s) > os.path.getmtime(_LIBPATH) for s in src))
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'],
                              stdout=subprocess.DEVNULL)
    return _LIBPATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


_vp = ctypes.c_void_p
_ci = ctypes.c_int


def _suffix(a):
    if a.dtype == np.complex128:
        return '_z'
    if a.dtype == np.float64:
        return '_d'
    raise TypeError(f"oracle: unsupported dtype {a.dtype}")


def _p(a, dtype=None):
    """Pointer to the first element of an F-contiguous (or 1-D) array."""
    if dtype is not None and a.dtype != dtype:
        raise TypeError(f"oracle: expected dtype {dtype}, got {a.dtype}")
    if a.ndim > 1 and not a.flags.f_contiguous:
        raise ValueError("oracle: arrays must be Fortran-contiguous")
    if a.ndim == 1 and not a.flags.c_contiguous:
        raise ValueError("oracle: 1-D arrays must be contiguous")
    return _vp(a.ctypes.data)


def _h(h):
    return np.ascontiguousarray(h, dtype=np.float64)


def _eta(eta, dtype):
    # eta has the field dtype in the reference (complex for f-domain, real for
    # Laplace); cast defensively without copying when it already matches.
    return np.asfortranarray(eta, dtype=dtype)


def amat_x(rx, ry, rz, ex, ey, ez, eta_x, eta_y, eta_z, zeta, hx, hy, hz):
    """r -= A e; reference emg3d/core.py:57-206."""
    dt = rx.dtype
    hx, hy, hz = _h(hx), _h(hy), _h(hz)
    e1, e2, e3 = _eta(eta_x, dt), _eta(eta_y, dt), _eta(eta_z, dt)
    zeta = np.asfortranarray(zeta, dtype=np.float64)
    getattr(lib(), 'amat_x' + _suffix(rx))(
        _p(rx), _p(ry, dt), _p(rz, dt), _p(ex, dt), _p(ey, dt), _p(ez, dt),
        _p(e1), _p(e2), _p(e3), _p(zeta), _p(hx), _p(hy), _p(hz),
        _ci(hx.size), _ci(hy.size), _ci(hz.size))


def _gs(name, ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz,
        nu, order):
    dt = ex.dtype
    hx, hy, hz = _h(hx), _h(hy), _h(hz)
    e1, e2, e3 = _eta(eta_x, dt), _eta(eta_y, dt), _eta(eta_z, dt)
    zeta = np.asfortranarray(zeta, dtype=np.float64)
    getattr(lib(), name + _suffix(ex))(
        _p(ex), _p(ey, dt), _p(ez, dt), _p(sx, dt), _p(sy, dt), _p(sz, dt),
        _p(e1), _p(e2), _p(e3), _p(zeta), _p(hx), _p(hy), _p(hz),
        _ci(hx.size), _ci(hy.size), _ci(hz.size), _ci(int(nu)), _ci(int(order)))


def gauss_seidel(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                 hz, nu, order=0):
    """Point smoother; reference emg3d/core.py:210-503."""
    _gs('gauss_seidel', ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx,
        hy, hz, nu, order)


def gauss_seidel_x(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                   hz, nu, order=0):
    """x-line smoother; reference emg3d/core.py:506-783."""
    _gs('gauss_seidel_x', ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta,
        hx, hy, hz, nu, order)


def gauss_seidel_y(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                   hz, nu, order=0):
    """y-line smoother; reference emg3d/core.py:786-1068."""
    _gs('gauss_seidel_y', ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta,
        hx, hy, hz, nu, order)


def gauss_seidel_z(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                   hz, nu, order=0):
    """z-line smoother; reference emg3d/core.py:1071-1348."""
    _gs('gauss_seidel_z', ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta,
        hx, hy, hz, nu, order)


def blocks_to_amat(amat, bvec, middle, left, rhs, im, nc):
    """Reference emg3d/core.py:1351-1477 (``left`` is float64 there)."""
    dt = amat.dtype
    middle = np.ascontiguousarray(middle, dtype=dt)
    left = np.ascontiguousarray(left, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=dt)
    getattr(lib(), 'blocks_to_amat' + _suffix(amat))(
        _p(amat), _p(bvec, dt), _p(middle), _p(left), _p(rhs),
        _ci(int(im)), _ci(int(nc)))


def solve(amat, bvec):
    """Reference emg3d/core.py:1481-1616."""
    getattr(lib(), 'solve' + _suffix(amat))(
        _p(amat), _p(bvec, amat.dtype), _ci(bvec.size))


def restrict(crx, cry, crz, rx, ry, rz, wx, wy, wz, sc_dir):
    """Reference emg3d/core.py:1620-2001."""
    dt = crx.dtype
    cnx, cny, cnz = cry.shape[0], crx.shape[1], crx.shape[2]
    nx, ny, nz = ry.shape[0], rx.shape[1], rx.shape[2]
    w = [np.ascontiguousarray(a, dtype=np.float64)
         for a in (*wx, *wy, *wz)]
    getattr(lib(), 'restrict' + _suffix(crx))(
        _p(crx), _p(cry, dt), _p(crz, dt), _p(rx, dt), _p(ry, dt), _p(rz, dt),
        *[_p(a) for a in w],
        _ci(cnx), _ci(cny), _ci(cnz), _ci(nx), _ci(ny), _ci(nz),
        _ci(int(sc_dir)))


def restrict_weights(nodes, cell_centers, h, cnodes, ccell_centers, ch):
    """Reference emg3d/core.py:2004-2076; returns (wl, w0, wr)."""
    nodes, cell_centers, h = _h(nodes), _h(cell_centers), _h(h)
    cnodes, ccell_centers, ch = _h(cnodes), _h(ccell_centers), _h(ch)
    n = cnodes.size
    wl, w0, wr = np.empty(n), np.empty(n), np.empty(n)
    lib().restrict_weights(
        _p(nodes), _p(cell_centers), _p(h), _ci(h.size), _p(cnodes),
        _p(ccell_centers), _p(ch), _ci(n), _p(wl), _p(w0), _p(wr))
    return wl, w0, wr
