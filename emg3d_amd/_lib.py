"""Loader of the HIP shared library (emg3d_amd/lib/libemg3d_amd.so) through ctypes.

The library is the product: there is no CPU fallback. If it cannot be loaded, or if a
device entry point is called without a GPU, an exception is raised.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIBDIR = os.path.join(_HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libemg3d_amd.so')
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'emg3d_amd.h')
SOURCES = [os.path.join(CSRC, f) for f in ('kernels.hip', 'stencil.h', 'launch.h', 'cplx.h', 'receivers.h', 'krylov.h', 'adjoint.h')] + [HEADER]

# -ffp-contract: hipcc's own default for HIP, spelled out because csrc/kernels.hip switches contraction off for its
# line-kernel section and back to THIS mode behind it (`#pragma clang fp contract(fast)`: the pragma can name a mode,
# not "whatever it was"; clang has no push / pop for it on this target -- `#pragma float_control` is ignored on amdgcn)
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast-honor-pragmas']


class Emg3dAmdError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    os.makedirs(LIBDIR, exist_ok=True)
    stale = (not os.path.exists(LIBPATH) or
             any(os.path.getmtime(s) > os.path.getmtime(LIBPATH) for s in SOURCES))
    if force or stale:
        hipcc = os.environ.get('HIPCC', 'hipcc')
        cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, 'kernels.hip'), '-o', LIBPATH]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIBPATH


_vp, _ci, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


class Level(ctypes.Structure):
    """Mirror of `emg3d_level` (include/emg3d_amd.h)."""
    _fields_ = [('nx', ctypes.c_int32), ('ny', ctypes.c_int32), ('nz', ctypes.c_int32),
                ('is_complex', ctypes.c_int32),
                ('ex', _vp), ('ey', _vp), ('ez', _vp),
                ('sx', _vp), ('sy', _vp), ('sz', _vp),
                ('eta_x', _vp), ('eta_y', _vp), ('eta_z', _vp),
                ('zeta', _vp), ('ihx', _vp), ('ihy', _vp), ('ihz', _vp),
                ('batch', ctypes.c_int32), ('flags', ctypes.c_int32), ('batch_stride', ctypes.c_int64)]


# name -> (restype, argtypes); every symbol declared in include/emg3d_amd.h
SIGNATURES = {
    'emg3d_version': (_ci, []),
    'emg3d_last_error': (ctypes.c_char_p, []),
    'emg3d_device_count': (_ci, []),
    'emg3d_set_option': (_ci, [ctypes.c_char_p, _ci]),
    'emg3d_get_option': (_ci, [ctypes.c_char_p]),
    'emg3d_option_count': (_ci, []),
    'emg3d_option_name': (ctypes.c_char_p, [_ci]),
    'emg3d_options_generation': (_ci, []),
    'emg3d_line_kernel_name': (ctypes.c_char_p, [_ci] * 6),
    'emg3d_core_amat_x': (_ci, [_vp] * 13 + [_ci] * 4),
    'emg3d_core_gauss_seidel': (_ci, [_ci] + [_vp] * 13 + [_ci] * 5),
    'emg3d_core_restrict': (_ci, [_vp] * 15 + [_ci] * 5),
    'emg3d_core_blocks_to_amat': (_ci, [_vp] * 5 + [_ci] * 4),
    'emg3d_core_solve': (_ci, [_vp, _vp, _ci, _ci]),
    'emg3d_gs_scratch_bytes': (_sz, [_ci] * 5),
    'emg3d_line_fac_bytes': (_sz, [_ci] * 5),
    'emg3d_line_fac_bytes_lv': (_sz, [ctypes.POINTER(Level), _ci]),
    'emg3d_line_compact_used': (_ci, [ctypes.POINTER(Level), _ci]),
    'emg3d_point_compact_used': (_ci, [ctypes.POINTER(Level)]),
    'emg3d_line_lfac_bytes': (_sz, [_ci] * 4),
    'emg3d_dev_line_setup': (_ci, [ctypes.POINTER(Level), _ci, _vp, _vp, _vp]),
    'emg3d_point_fac_bytes': (_sz, [_ci] * 4),
    'emg3d_point_fac_bytes_lv': (_sz, [ctypes.POINTER(Level)]),
    'emg3d_dev_eta_is_imaginary': (_ci, [ctypes.POINTER(Level), ctypes.POINTER(_ci), _vp]),
    'emg3d_dev_point_setup': (_ci, [ctypes.POINTER(Level), _vp, _vp]),
    'emg3d_dev_gauss_seidel': (_ci, [ctypes.POINTER(Level), _ci, _ci, _vp, _vp, _vp, _sz, _vp]),
    'emg3d_residual_ws_len': (_sz, [_ci] * 3),
    'emg3d_dev_residual': (_ci, [ctypes.POINTER(Level), _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'emg3d_dev_restrict': (_ci, [_vp] * 15 + [_ci] * 5 + [_vp]),
    'emg3d_dev_prolong': (_ci, [_vp] * 12 + [_ci] * 5 + [_vp]),
    'emg3d_dev_restrict_batch': (_ci, [_vp] * 15 + [_ci] * 6 + [_sz, _sz, _vp]),
    'emg3d_dev_restrict_clear_batch': (_ci, [_vp] * 18 + [_ci] * 6 + [_sz, _sz, _vp]),
    'emg3d_dev_prolong_batch': (_ci, [_vp] * 12 + [_ci] * 6 + [_sz, _sz, _vp]),
    'emg3d_dev_restrict_param': (_ci, [_vp, _vp] + [_ci] * 5 + [_vp]),
    'emg3d_dev_pec_zero': (_ci, [_vp] * 3 + [_ci] * 4 + [_vp]),
    'emg3d_krylov_ws_len': (_sz, []),
    'emg3d_dev_krylov_step': (_ci, [_sz, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _sz, _vp]),
    'emg3d_dev_apply_operator': (_ci, [ctypes.POINTER(Level), _vp, _vp, _vp, _vp]),
    'emg3d_dev_zero': (_ci, [_vp, _sz, _vp]),
    'emg3d_dev_copy': (_ci, [_vp, _vp, _sz, _vp]),
    'emg3d_dev_gradient_accumulate': (_ci, [_ci] * 4 + [_vp] * 6 + [ctypes.c_double] * 2 + [_vp] * 5),
    'emg3d_dev_source_field': (_ci, [_ci] * 4 + [_vp] * 7 + [_ci] + [ctypes.c_double] * 2 + [_vp] * 4),
    'emg3d_dev_volume_model': (_ci, [_ci] * 4 + [_vp] * 5 + [_ci] + [_vp] * 3 + [ctypes.c_double] * 4 + [_vp] * 5),
    'emg3d_dev_magnetic_field': (_ci, [_ci] * 4 + [_vp] * 7 + [ctypes.c_double] * 2 + [_vp] * 4),
    'emg3d_dev_spline_filter': (_ci, [_vp] + [_ci] * 4 + [_vp]),
    'emg3d_dev_spline_eval': (_ci, [_vp] + [_ci] * 4 + [_vp, _ci, _vp, _vp]),
    'emg3d_dev_linear_eval': (_ci, [_vp] + [_ci] * 4 + [_vp, _vp, _ci, _vp, _vp]),
    'emg3d_dev_volume_average': (_ci, [_vp] + [_ci] * 3 + [_vp] * 10 + [_ci] * 3 + [_vp, _ci, _vp]),
}

LEVEL_ETA_IMAG = 1      # emg3d_level.flags (include/emg3d_amd.h)
LEVEL_LINE_COMPACT = 2  # ... the level solves a correction equation: streamed line records may be single precision
LEVEL_POINT_COMPACT = 4  # ... and so may the eta edge sums of the tiled point smoother

_lib = None


def lib():
    """The loaded library (ctypes.CDLL) with argument types set. Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise ImportError(
                f"emg3d_amd: HIP library not found at {LIBPATH}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback.")
        cdll = ctypes.CDLL(LIBPATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)     # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def check(status, what=''):
    """Turn a non-zero C-ABI status into an exception (reference kernels raise nothing;
    this only reports HIP / argument errors, see include/emg3d_amd.h)."""
    if status != 0:
        msg = lib().emg3d_last_error()
        raise Emg3dAmdError(f"{what} failed with status {status}: "
                            f"{msg.decode() if msg else ''}")


_fingerprint = (None, None)


def options_fingerprint():
    """Current values of all run-time options of the library, in its own order (part of the key of
    everything that is built under them: captured graphs, option-dependent factor buffers). Read again
    only when the library's generation counter has moved (one ctypes call otherwise)."""
    global _fingerprint
    L = lib()
    gen = L.emg3d_options_generation()
    if _fingerprint[0] != gen:
        _fingerprint = (gen, tuple(L.emg3d_get_option(L.emg3d_option_name(i)) for i in range(L.emg3d_option_count())))
    return _fingerprint[1]


def require_gpu():
    if lib().emg3d_device_count() < 1:
        raise Emg3dAmdError("emg3d_amd: no HIP device visible; the MI355X path has no CPU "
                            "fallback.")
