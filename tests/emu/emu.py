"""TEST INFRASTRUCTURE ONLY: ctypes front-end of tests/emu/emu.cpp (CPU walk of the HIP
launch grids calling the kernel bodies of emg3d_amd/csrc/stencil.h)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'libemu.so')
_SRC = [os.path.join(_HERE, 'emu.cpp')] + [
    os.path.join(_HERE, '..', '..', 'emg3d_amd', 'csrc', f) for f in ('stencil.h', 'launch.h', 'cplx.h')]


def build():
    if (not os.path.exists(_LIB) or
            any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in _SRC)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-w',
                               os.path.join(_HERE, 'emu.cpp'), '-o', _LIB])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emu_residual.restype = ctypes.c_double
        _lib.emu_residual_column.restype = ctypes.c_double
    return _lib


class LevelArgs(ctypes.Structure):
    _fields_ = [('nx', ctypes.c_int32), ('ny', ctypes.c_int32), ('nz', ctypes.c_int32),
                ('is_complex', ctypes.c_int32)] + [
        (n, ctypes.c_void_p) for n in ('ex', 'ey', 'ez', 'sx', 'sy', 'sz', 'eta_x', 'eta_y',
                                       'eta_z', 'zeta', 'ihx', 'ihy', 'ihz')]


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def make_level(e, s, vm, keep):
    """e, s: objects with fx/fy/fz F-order views; vm: eta_x..zeta + grid.h."""
    dt = e.fx.dtype
    cplx = dt == np.complex128
    eta = []
    for k in ('eta_x', 'eta_y', 'eta_z'):
        a = getattr(vm, k)
        b = np.asfortranarray(a, dtype=dt)
        for prev_a, prev_b in eta:   # preserve aliasing
            if prev_a is a:
                b = prev_b
        eta.append((a, b))
    zeta = np.asfortranarray(vm.zeta, dtype=np.float64)
    ih = [np.ascontiguousarray(1.0 / h) for h in vm.grid.h]
    keep.extend([b for _, b in eta] + [zeta] + ih)
    nx, ny, nz = vm.grid.shape_cells
    return LevelArgs(nx, ny, nz, int(cplx), _ptr(e.fx), _ptr(e.fy), _ptr(e.fz), _ptr(s.fx),
                     _ptr(s.fy), _ptr(s.fz), _ptr(eta[0][1]), _ptr(eta[1][1]), _ptr(eta[2][1]),
                     _ptr(zeta), _ptr(ih[0]), _ptr(ih[1]), _ptr(ih[2]))


def gauss_seidel(e, s, vm, lr, nu):
    keep = []
    lv = make_level(e, s, vm, keep)
    lib().emu_gauss_seidel(ctypes.byref(lv), int(lr), int(nu))


def gauss_seidel_fac(e, s, vm, lr, nu, fac, lfac):
    """Line sweeps with given factor records (numpy uint8 / float64 buffers in the layout of emg3d_dev_line_setup)."""
    keep = []
    lv = make_level(e, s, vm, keep)
    fac, lfac = np.ascontiguousarray(fac), np.ascontiguousarray(lfac)
    lib().emu_gauss_seidel_fac(ctypes.byref(lv), int(lr), int(nu), _ptr(fac), _ptr(lfac))


def residual(e, s, vm, r=None):
    """Returns sum |r|^2; fills r (object with fx/fy/fz) if given."""
    keep = []
    lv = make_level(e, s, vm, keep)
    if r is None:
        return lib().emu_residual(ctypes.byref(lv), None, None, None)
    return lib().emu_residual(ctypes.byref(lv), _ptr(r.fx), _ptr(r.fy), _ptr(r.fz))


def residual_column(e, s, vm, r, zb):
    """The residual through the HIP kernels' column walk (operands carried from cell to cell), zb planes per walk."""
    keep = []
    lv = make_level(e, s, vm, keep)
    return lib().emu_residual_column(ctypes.byref(lv), _ptr(r.fx), _ptr(r.fy), _ptr(r.fz), int(zb))


def line_coupling_mismatches(e, s, vm, direction, i1, i2):
    """Number of recomputed coupling entries (stencil.h: line_coupling) that differ bitwise from the stored lfac
    records of line (i1, i2) of the direction."""
    keep = []
    lv = make_level(e, s, vm, keep)
    return lib().emu_line_coupling_mismatches(ctypes.byref(lv), int(direction), int(i1), int(i2))


def line_blocks(e, s, vm, direction, i1, i2):
    """Blocks of the line system of line (i1, i2) (abstract transverse node indices) in direction
    0/1/2 as the kernels assemble them: (dg (n0,5) complex, mid (n0,5,5), left0 (n0,5), leftd (n0,5),
    rhs (n0,5) complex)."""
    keep = []
    lv = make_level(e, s, vm, keep)
    n0 = vm.grid.shape_cells[direction]
    dg, rhs = np.zeros((n0, 5), complex), np.zeros((n0, 5), complex)
    mid, l0, ld = np.zeros((n0, 5, 5)), np.zeros((n0, 5)), np.zeros((n0, 5))
    lib().emu_line_blocks(ctypes.byref(lv), int(direction), int(i1), int(i2), _ptr(dg), _ptr(mid), _ptr(l0), _ptr(ld), _ptr(rhs))
    return dg, mid, l0, ld, rhs


def restrict(c, r, w9, shape, sc_dir):
    arr = (ctypes.c_void_p * 9)(*[_ptr(a) if a is not None else None for a in w9])
    nx, ny, nz = shape
    lib().emu_restrict(_ptr(c.fx), _ptr(c.fy), _ptr(c.fz), _ptr(r.fx), _ptr(r.fy), _ptr(r.fz),
                       arr, nx, ny, nz, int(sc_dir), int(r.fx.dtype == np.complex128))


def prolong(f, c, il, w, shape, sc_dir):
    nx, ny, nz = shape
    lib().emu_prolong(_ptr(f.fx), _ptr(f.fy), _ptr(f.fz), _ptr(c.fx), _ptr(c.fy), _ptr(c.fz),
                      _ptr(il[0]), _ptr(il[1]), _ptr(il[2]), _ptr(w[0]), _ptr(w[1]), _ptr(w[2]),
                      nx, ny, nz, int(sc_dir), int(f.fx.dtype == np.complex128))


def restrict_param(p, sc_dir):
    nx, ny, nz = p.shape
    fx = 1 if sc_dir in (1, 5, 6) else 2
    fy = 1 if sc_dir in (2, 4, 6) else 2
    fz = 1 if sc_dir in (3, 4, 5) else 2
    out = np.zeros((nx // fx, ny // fy, nz // fz), dtype=p.dtype, order='F')
    p = np.asfortranarray(p)
    lib().emu_restrict_param(_ptr(out), _ptr(p), nx, ny, nz, int(sc_dir),
                             int(p.dtype == np.complex128))
    return out


def solve(amat, bvec):
    lib().emu_solve(_ptr(amat), _ptr(bvec), bvec.size, int(amat.dtype == np.complex128))


def blocks_to_amat(amat, bvec, middle, left, rhs, im, nc):
    dt = amat.dtype
    middle = np.ascontiguousarray(middle, dtype=dt)
    left = np.ascontiguousarray(left, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=dt)
    lib().emu_blocks_to_amat(_ptr(amat), _ptr(bvec), _ptr(middle), _ptr(left), _ptr(rhs),
                             int(im), int(nc), int(dt == np.complex128))
