"""Device-resident grid levels and thin wrappers around the ``emg3d_dev_*`` C ABI.

PyTorch is used for exactly three things here: HBM allocations (tensors), the HIP stream
the kernels are enqueued on, and (in emg3d_amd/parallel.py) the RCCL process group. All
arithmetic of the multigrid path happens in the hand-written HIP kernels of
emg3d_amd/csrc/; no torch operator touches field data in the cycle.

A ``DeviceLevel`` owns, for one grid of the hierarchy: the electric field ``e``, the source
``s`` and a residual buffer ``r`` (each ONE buffer ``[fx|fy|fz]`` like the reference's
``Field``, emg3d/fields.py:201-259), the volume-integrated model ``eta_x/eta_y/eta_z/zeta``
(aliased for isotropic / VTI / HTI models, emg3d/models.py:693-712), inverse cell widths,
and lazily-built coarse children per semicoarsening direction together with the
restriction weights (emg3d/solver.py:1721-1780) and prolongation tables
(emg3d/solver.py:1457-1462) that connect them. The reference re-creates coarse grids,
models and fields at every visit (emg3d/solver.py:849-944); here they are built once per
(level, sc_dir) and reused.
"""
import ctypes

import numpy as np
import torch

from emg3d_amd import _lib, meshes
from emg3d_amd import core as _core

_vp = ctypes.c_void_p


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t, offset_elems=0):
    return _vp(t.data_ptr() + offset_elems * t.element_size())


def coarsen_flags(sc_dir):
    """(cx, cy, cz): is the direction coarsened for this sc_dir? (solver.py:891-897)"""
    return (sc_dir not in (1, 5, 6), sc_dir not in (2, 4, 6), sc_dir not in (3, 4, 5))


def interp_table(cnodes, nodes):
    """Lower coarse node index and weight of the upper coarse node for every fine node,
    as ``np.searchsorted`` yields them in the reference (emg3d/solver.py:1457-1462)."""
    i = np.searchsorted(cnodes, nodes) - 1
    i[i < 0] = 0
    i[i > cnodes.size - 2] = cnodes.size - 2
    w = (nodes - cnodes[i]) / (cnodes[i + 1] - cnodes[i])
    return i.astype(np.int32), w.astype(np.float64)


class Workspace:
    """Scratch shared by all levels of one hierarchy (solver lifetime, not per call)."""

    def __init__(self, device):
        self.device = device
        self.gs_scratch = None
        self.gs_bytes = 0
        self._pin, self._pin_used, self._pins = None, 0, []
        self.ws = None
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        # line factors of the hierarchy's levels: 'resident' (every direction a level has used stays in
        # HBM: 304 B per cell and direction) or 'rebuild' (DeviceLevel.line_factors)
        self.factor_policy = 'resident'
        self.factor_epoch = 0          # advanced by every coarse-grid correction (_cycle.coarse_correction)
        # compact line records on the levels that stream them (solver.Hierarchy(line_compact=...)): every level of
        # the hierarchy carries LEVEL_LINE_COMPACT, and the finest level runs in residual form (_cycle.run_cycles)
        self.line_compact = False

    def upload(self, a):
        """Small host array -> device tensor through a pinned staging pool, asynchronously on
        the current stream. A plain ``.to(device)`` from pageable memory goes through the
        runtime's synchronous staging path: measured ~2.5 ms of idle GPU per call (and one
        70 ms outlier) -- with three uploads per level that was a third of a 128^3 solve."""
        a = np.ascontiguousarray(a)
        nbytes = a.nbytes
        if self._pin is None or self._pin_used + nbytes > self._pin.numel():
            # a fresh block; the old one stays referenced by the tensors staged from it
            self._pin = torch.empty(max(1 << 20, 2 * nbytes), dtype=torch.uint8).pin_memory()
            self._pin_used = 0
            self._pins.append(self._pin)
        view = self._pin[self._pin_used:self._pin_used + nbytes]
        self._pin_used += (nbytes + 63) // 64 * 64
        view.numpy()[:] = a.view(np.uint8).reshape(-1)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dev.copy_(view, non_blocking=True)
        return dev.view(torch.from_numpy(a[:0]).dtype)

    def need_gs(self, nbytes):
        """Line-smoother scratch. Sized once for the largest request of the hierarchy
        (DeviceLevel.from_host reserves the top level's maximum over the three directions):
        captured HIP graphs hold its address, so it must never be re-allocated later."""
        if nbytes > self.gs_bytes:
            self.gs_scratch = None   # release before growing
            self.gs_scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.gs_bytes = nbytes

    def need_ws(self, n):
        if self.ws is None or self.ws.numel() < n:
            self.ws = torch.empty(max(int(n), 1), dtype=torch.float64, device=self.device)


class DeviceLevel:
    """One grid level resident in HBM."""

    def __init__(self, grid, case, eta_x, eta_y, eta_z, zeta, dtype, work, device, batch=1, flags=None):
        self.grid = grid
        self.batch = int(batch)     # right-hand sides that share this level's model and factors
        self.case = case
        self.device = device
        self.work = work
        self.dtype = dtype                                  # torch.complex128 / float64
        self.is_complex = int(dtype == torch.complex128)
        self.eta_x, self.eta_y, self.eta_z, self.zeta = eta_x, eta_y, eta_z, zeta
        ihall = work.upload(np.concatenate([1.0 / np.asarray(h, dtype=np.float64) for h in grid.h]))
        nh = np.cumsum([0] + [len(h) for h in grid.h])
        self.ih = [ihall[nh[d]:nh[d + 1]] for d in range(3)]      # one upload, three views
        n = grid.n_edges
        # batch > 1: the buffers of the right-hand sides one behind the other (source b at b * n)
        self.e = torch.zeros(n * self.batch, dtype=dtype, device=device)
        self.s = torch.zeros(n * self.batch, dtype=dtype, device=device)
        self._r = None
        self.children = {}
        self._factors = {}
        self._parts = {}
        self.n_cells = grid.n_cells
        self._o1, self._o2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
        nx, ny, nz = grid.shape_cells
        self._c = _lib.Level(
            nx, ny, nz, self.is_complex,
            _ptr(self.e), _ptr(self.e, self._o1), _ptr(self.e, self._o2),
            _ptr(self.s), _ptr(self.s, self._o1), _ptr(self.s, self._o2),
            _ptr(eta_x), _ptr(eta_y), _ptr(eta_z), _ptr(zeta),
            _ptr(self.ih[0]), _ptr(self.ih[1]), _ptr(self.ih[2]), self.batch, 0, n)
        self._cref = ctypes.byref(self._c)
        if flags is None:
            # the finest level asks once (sums of purely imaginary eta stay purely imaginary:
            # coarse levels inherit the answer)
            res = ctypes.c_int(0)
            _lib.check(_lib.lib().emg3d_dev_eta_is_imaginary(self._cref, ctypes.byref(res), _stream()),
                       'emg3d_dev_eta_is_imaginary')
            flags = _lib.LEVEL_ETA_IMAG if res.value else 0
        self.flags = self._c.flags = int(flags)
        work.need_ws(_lib.lib().emg3d_residual_ws_len(nx, ny, nz) * self.batch)
        if self.batch > 1 and work.sumsq.numel() < self.batch:
            work.sumsq = torch.zeros(self.batch, dtype=torch.float64, device=device)

    # ---------------------------------------------------------------------------------
    @classmethod
    def from_host(cls, vmodel, device, work=None, batch=1, line_factors='resident'):
        """Upload a host ``VolumeModel`` (finest level). line_factors: 'resident' | 'rebuild' | 'single', the factor
        memory policy of the whole hierarchy (``line_factors`` / ``_line_factor_slots``)."""
        if line_factors not in ('resident', 'rebuild', 'single'):
            raise ValueError(f"`line_factors` must be 'resident', 'rebuild' or 'single'. Provided: {line_factors!r}.")
        if work is None:
            work = Workspace(device)
        elif getattr(work, '_policy_owner', False) and work.factor_policy != line_factors:
            # a shared workspace carries ONE policy: a second hierarchy with another one would switch the first
            # hierarchy's levels between the `_factors` and `_slots` schemes under its captured graphs
            raise ValueError(f"`work` already serves a hierarchy with line_factors={work.factor_policy!r}; "
                             f"provided: {line_factors!r}.")
        work.factor_policy = line_factors
        work._policy_owner = True
        if hasattr(vmodel, 'device_arrays'):
            # emg3d_amd.models.VolumeModel: form eta / zeta in HBM from the conductivities
            ex, ey, ez, zeta = vmodel.device_arrays(device)
            top = cls(meshes.BaseMesh(vmodel.grid.h, vmodel.grid.origin), vmodel.case, ex, ey, ez,
                      zeta, ex.dtype, work, device, batch)
            top.is_top = True
            nx, ny, nz = top.grid.shape_cells
            work.need_gs(batch * max(_lib.lib().emg3d_gs_scratch_bytes(lr, nx, ny, nz, top.is_complex)
                                     for lr in (1, 2, 3)))
            return top
        cplx = np.iscomplexobj(vmodel.eta_x)
        dtype = torch.complex128 if cplx else torch.float64
        ndt = np.complex128 if cplx else np.float64
        up = {}

        def upload(a, dt):
            key = id(a)
            if key not in up:      # preserves aliasing of eta_x/eta_y/eta_z
                up[key] = torch.from_numpy(np.asfortranarray(a, dtype=dt).ravel('F').copy()).to(device)
            return up[key]
        top = cls(meshes.BaseMesh(vmodel.grid.h, vmodel.grid.origin), vmodel.case,
                  upload(vmodel.eta_x, ndt), upload(vmodel.eta_y, ndt), upload(vmodel.eta_z, ndt),
                  upload(vmodel.zeta, np.float64), dtype, work, device, batch)
        top.is_top = True
        nx, ny, nz = top.grid.shape_cells
        work.need_gs(batch * max(_lib.lib().emg3d_gs_scratch_bytes(lr, nx, ny, nz, top.is_complex)
                                 for lr in (1, 2, 3)))
        return top

    def set_line_compact(self, on=True, point_here=True):
        """Mark this level (and the coarse levels made from it afterwards) as solving correction equations only:
        the streamed line passes keep their T and w records, the tiled point smoother its eta sums in single precision
        (include/emg3d_amd.h: EMG3D_LEVEL_LINE_COMPACT, _POINT_COMPACT). Before the first factorisation of the level.
        point_here = False: the point smoother of THIS level keeps fp64 sums (the coarse levels below it do not): a
        finest level that would run in residual form for their sake alone -- a plain multigrid solve at a loose tolerance
        -- loses more to the residual form than the narrower sums save (solver.Hierarchy(point_compact_top=))."""
        if self._factors or self.__dict__.get('_slots') or self.children:
            raise RuntimeError("set_line_compact: the level already has factors or coarse levels")
        bits = _lib.LEVEL_LINE_COMPACT | _lib.LEVEL_POINT_COMPACT
        base = self.flags & ~bits
        self._child_flags = (base | bits) if on else base
        self.flags = (base | _lib.LEVEL_LINE_COMPACT | (_lib.LEVEL_POINT_COMPACT if point_here else 0)) if on else base
        self._c.flags = self.flags
        self.work.line_compact = bool(on)

    def uses_line_compact(self, lines=True):
        """Does a smoother of THIS level keep compact records -- a line direction (the flag is set and the direction
        streams or runs the three-phase kernel on long lines; only asked if the solve relaxes lines at all: `lines`), or
        the tiled point smoother? The finest level of such a solve runs in residual form (_cycle.run_cycles)."""
        if not self.flags & (_lib.LEVEL_LINE_COMPACT | _lib.LEVEL_POINT_COMPACT):
            return False
        lib = _lib.lib()
        return ((lines and any(lib.emg3d_line_compact_used(self._cref, lr) for lr in (1, 2, 3))) or
                bool(lib.emg3d_point_compact_used(self._cref)))

    @property
    def r(self):
        if self._r is None:
            self._r = torch.empty(self.grid.n_edges * self.batch, dtype=self.dtype, device=self.device)
        return self._r

    def parts(self, t):
        """(px, py, pz) pointers into a 1-D buffer [fx|fy|fz] (cached per tensor: the level's
        own buffers never move)."""
        key = id(t)
        hit = self._parts.get(key)
        if hit is None or hit[0] is not t:
            hit = (t, (_ptr(t), _ptr(t, self._o1), _ptr(t, self._o2)))
            if t is self.e or t is self.s or t is self._r:
                self._parts[key] = hit
        return hit[1]

    # ------------------------------------------------------------------- kernels ------
    def _line_factor_slots(self, lr):
        """Policy 'rebuild': a level holds TWO factor buffers (the two directions of a line-relaxation
        code; 608 instead of 912 B per cell with all three directions in use; policy 'single': ONE buffer,
        304 B per cell, re-factorised at every change of direction) and re-factorises when a
        direction is asked for that neither holds -- emg3d_dev_line_setup into the same buffer, i.e. the same
        addresses whatever the direction. The finest level keeps its buffers from cycle to cycle: when the
        code advances (4 -> 5 -> 6: (y,z) -> (x,z) -> (x,y)) one direction is rebuilt per cycle, the
        direction that is not part of the new code makes room. A coarse level starts every coarse-grid
        correction with empty buffers (``work.factor_epoch``) and rebuilds at its first smoothing call of the
        correction: what a correction launches then does not depend on what ran before it -- the condition
        for replaying it from a captured graph."""
        lib = _lib.lib()
        work = self.work
        slots = self.__dict__.get('_slots')
        if slots is None:
            nx, ny, nz = self.grid.shape_cells
            nf = max(lib.emg3d_line_fac_bytes_lv(self._cref, d) for d in (1, 2, 3))      # (compact records: smaller)
            nl = max(lib.emg3d_line_lfac_bytes(d, nx, ny, nz) for d in (1, 2, 3))
            slots = self._slots = [{'dir': None, 'used': 0,
                                    'fac': torch.empty(nf, dtype=torch.uint8, device=self.device),
                                    'lfac': torch.empty(nl, dtype=torch.uint8, device=self.device)}
                                   for _ in range(1 if work.factor_policy == 'single' else 2)]
            self._slot_epoch, self._slot_clock, self.factor_rebuilds = work.factor_epoch, 0, 0
        if not self.__dict__.get('is_top', False) and self._slot_epoch != work.factor_epoch:
            for sl in slots:
                sl['dir'] = None
            self._slot_epoch = work.factor_epoch
        self._slot_clock += 1
        for sl in slots:
            if sl['dir'] == lr:
                sl['used'] = self._slot_clock
                return sl['fac'], sl['lfac']
        keep = self.__dict__.get('_factor_keep', ())
        free = [sl for sl in slots if sl['dir'] is None] or [sl for sl in slots if sl['dir'] not in keep]
        if free:
            victim = min(free, key=lambda sl: sl['used'])
        else:
            # more directions in use than buffers (line-relaxation code 7 = x, y, z with two buffers; any code with
            # one): under the cyclic access x, y, z, x, ... evicting the LEAST recently used buffer would miss on
            # every call (three factorisations per smoothing call, what 'single' does with half the memory) --
            # evict the MOST recently used one: the other buffer stays resident and hits once per round
            victim = max(slots, key=lambda sl: sl['used'])
        _lib.check(lib.emg3d_dev_line_setup(self._cref, lr, _ptr(victim['fac']), _ptr(victim['lfac']), _stream()),
                   'emg3d_dev_line_setup')
        victim['dir'], victim['used'] = lr, self._slot_clock
        self.factor_rebuilds += 1
        return victim['fac'], victim['lfac']

    def line_factors(self, lr):
        """Block factorisation of all lines of direction lr (1/2/3), built on first use
        and kept for the lifetime of the level (the model does not change in a solve) -- or, with the
        hierarchy's policy 'rebuild', held in one of two buffers per level (``_line_factor_slots``)."""
        if self.work.factor_policy in ('rebuild', 'single'):
            return self._line_factor_slots(lr)
        if lr not in self._factors:
            lib = _lib.lib()
            nx, ny, nz = self.grid.shape_cells
            # (sized for THIS level: 120 instead of 240 B per block where the direction keeps compact records,
            # emg3d_level.flags & LEVEL_LINE_COMPACT)
            fac = torch.empty(lib.emg3d_line_fac_bytes_lv(self._cref, lr), dtype=torch.uint8, device=self.device)
            lfac = torch.empty(lib.emg3d_line_lfac_bytes(lr, nx, ny, nz), dtype=torch.uint8,
                               device=self.device)
            _lib.check(lib.emg3d_dev_line_setup(self._cref, lr, _ptr(fac), _ptr(lfac), _stream()),
                       'emg3d_dev_line_setup')
            self._factors[lr] = (fac, lfac)
        return self._factors[lr]

    def point_factors(self):
        """Eta edge sums of the point smoother (emg3d_dev_point_setup), built on first use. Their
        layout follows the sweep schedule of the level (option point_tile_min): one buffer per value
        of that option, all kept -- graphs captured under an earlier value (``_cycle.coarse_correction``
        keys them on the option set) still hold its buffer's address."""
        lib = _lib.lib()
        key = ('point', lib.emg3d_get_option(b'point_tile_min'))
        if key not in self._factors:
            fac = torch.empty(lib.emg3d_point_fac_bytes_lv(self._cref), dtype=torch.uint8, device=self.device)
            _lib.check(lib.emg3d_dev_point_setup(self._cref, _ptr(fac), _stream()),
                       'emg3d_dev_point_setup')
            self._factors[key] = (fac, None)
        return self._factors[key][0]

    def smooth(self, lr, nu):
        """nu sweeps of smoother lr (0 point, 1/2/3 x/y/z line) on (e, s)."""
        lib = _lib.lib()
        nx, ny, nz = self.grid.shape_cells
        fac = lfac = scr = None
        nbytes = 0
        if lr == 0:
            fac = _ptr(self.point_factors())
        if lr:
            f, lf = self.line_factors(lr)
            fac, lfac = _ptr(f), _ptr(lf)
            nbytes = lib.emg3d_gs_scratch_bytes(lr, nx, ny, nz, self.is_complex) * self.batch
            self.work.need_gs(nbytes)
            scr = _ptr(self.work.gs_scratch)
        _lib.check(lib.emg3d_dev_gauss_seidel(self._cref, lr, nu, fac, lfac, scr, nbytes, _stream()),
                   'emg3d_dev_gauss_seidel')

    def keep_field(self):
        """Copy e aside (for ``extrapolate_field``)."""
        if self.__dict__.get('_e_kept') is None:
            self._e_kept = torch.empty_like(self.e)
        _lib.check(_lib.lib().emg3d_dev_copy(_ptr(self._e_kept), _ptr(self.e), self.e.numel() * self.e.element_size(),
                                             _stream()), 'emg3d_dev_copy')

    def extrapolate_field(self, omega):
        """e <- e_kept + omega (e - e_kept): one fused update (emg3d_dev_krylov_step with two
        immediate coefficients; the level's reduction workspace doubles as its unused scalar table)."""
        xs = (ctypes.c_void_p * 2)(self._e_kept.data_ptr(), self.e.data_ptr())
        slots = (ctypes.c_int * 2)(-1, -1)
        scales = (ctypes.c_double * 2)(1.0 - omega, omega)
        none_p, none_i = (ctypes.c_void_p * 1)(), (ctypes.c_int * 1)()
        w = self.work
        _lib.check(_lib.lib().emg3d_dev_krylov_step(
            self.e.numel(), int(self.is_complex), _ptr(self.e), 2, xs, slots, scales, 0, none_p, none_p, none_i, 0, none_i,
            _ptr(w.ws), _ptr(w.ws), w.ws.numel(), _stream()), 'emg3d_dev_krylov_step')

    # ---- finest level in residual form (_cycle.run_cycles): the cycle works on A d = r from d = 0
    def reserve_residual_equation(self):
        """The two field-sized buffers of the residual form (allocated on first use; a caller that switches
        to it in the middle of a solve reserves them first and stays in direct form if HBM is short)."""
        if self.__dict__.get('_x_kept') is None:
            self._x_kept, self._b_kept = torch.empty_like(self.e), torch.empty_like(self.s)
            self._b_valid = False
        _ = self.r

    # Residual form without per-cycle copies. While it lasts the solve's field and source live in `_x_kept` /
    # `_b_kept`; a cycle runs on (e, s) = (d, r): the residual buffer becomes the source by SWAPPING the two tensors
    # (and the pointers of the level struct), the correction is accumulated into `_x_kept`, and between cycles
    # `residual` works on (_x_kept, _b_kept). `_resmode`: None direct form, 'idle' between cycles, 'cycle' inside one.
    # Per cycle: one fill and one fused update (round 5: three copies more, ~1 ms of a 256^3 cycle).
    def _swap_s_r(self):
        _ = self.r
        self.s, self._r = self._r, self.s
        self._c.sx, self._c.sy, self._c.sz = (ctypes.c_void_p(p) if not isinstance(p, ctypes.c_void_p) else p
                                                for p in self.parts(self.s))

    def to_residual_equation(self):
        """Before a cycle: r = s - A x is in ``self.r`` (residual(store=True)). First time: keeps the field and the
        source aside. Then s <- r (tensor swap), e <- 0."""
        nbytes = self.e.numel() * self.e.element_size()
        self.reserve_residual_equation()
        cp = _lib.lib().emg3d_dev_copy
        if self.__dict__.get('_resmode') is None:
            _lib.check(cp(_ptr(self._x_kept), _ptr(self.e), nbytes, _stream()), 'emg3d_dev_copy')
            if not self._b_valid:                 # the source of a solve does not change between its cycles
                _lib.check(cp(_ptr(self._b_kept), _ptr(self.s), nbytes, _stream()), 'emg3d_dev_copy')
                self._b_valid = True
        self._swap_s_r()
        self.zero_field()
        self._resmode = 'cycle'

    def from_residual_equation(self):
        """After the cycle: x_kept += d (one fused update); the residual buffer is free again."""
        xs = (ctypes.c_void_p * 2)(self._x_kept.data_ptr(), self.e.data_ptr())
        slots = (ctypes.c_int * 2)(-1, -1)
        scales = (ctypes.c_double * 2)(1.0, 1.0)
        none_p, none_i = (ctypes.c_void_p * 1)(), (ctypes.c_int * 1)()
        w = self.work
        _lib.check(_lib.lib().emg3d_dev_krylov_step(
            self.e.numel(), int(self.is_complex), _ptr(self._x_kept), 2, xs, slots, scales, 0, none_p, none_p, none_i, 0,
            none_i, _ptr(w.ws), _ptr(w.ws), w.ws.numel(), _stream()), 'emg3d_dev_krylov_step')
        self._swap_s_r()
        self._resmode = 'idle'

    def solution(self):
        """The tensor that holds the solve's field right now (between the cycles of a residual-form solve: the
        accumulated ``_x_kept``; ``e`` otherwise)."""
        return self._x_kept if self.__dict__.get('_resmode') == 'idle' else self.e

    def leave_residual_form(self):
        """End of a residual-form solve: e <- the accumulated field, s <- the solve's source."""
        mode = self.__dict__.get('_resmode')
        if mode is None:
            return
        if mode == 'cycle':
            self._swap_s_r()
        nbytes = self.e.numel() * self.e.element_size()
        cp = _lib.lib().emg3d_dev_copy
        _lib.check(cp(_ptr(self.e), _ptr(self._x_kept), nbytes, _stream()), 'emg3d_dev_copy')
        _lib.check(cp(_ptr(self.s), _ptr(self._b_kept), nbytes, _stream()), 'emg3d_dev_copy')
        self._resmode = None

    def abandon_residual_equation(self):
        """A cycle in residual form was interrupted: e <- the field before it, s <- the solve's source."""
        self.leave_residual_form()

    def residual(self, store=True, norm=False):
        """r = s - A e into self.r (store) and/or its l2-norm (norm; synchronises)."""
        lib = _lib.lib()
        rx, ry, rz = self.parts(self.r) if store else (None, None, None)
        w = self.work
        cref = self._cref
        if self.__dict__.get('_resmode') == 'idle':      # between the cycles of a residual-form solve: the true equation
            cref = ctypes.byref(self._level_on(self._x_kept, self._b_kept))
        _lib.check(lib.emg3d_dev_residual(cref, rx, ry, rz, _ptr(w.ws), w.ws.numel(),
                                          _ptr(w.sumsq) if norm else None, _stream()),
                   'emg3d_dev_residual')
        if norm and self.batch > 1:
            return np.sqrt(w.sumsq[:self.batch].cpu().numpy())      # one norm per right-hand side
        if norm:
            return float(np.sqrt(w.sumsq[0].item()))
        return None

    def _level_on(self, e, s):
        """The level's C struct with other field / source vectors in place of e / s."""
        nx, ny, nz = self.grid.shape_cells
        return _lib.Level(nx, ny, nz, self.is_complex, *self.parts(e), *self.parts(s),
                          _ptr(self.eta_x), _ptr(self.eta_y), _ptr(self.eta_z), _ptr(self.zeta),
                          _ptr(self.ih[0]), _ptr(self.ih[1]), _ptr(self.ih[2]),
                          self.batch, self.flags, self.grid.n_edges)

    def apply_A(self, x, out):
        """out = A x for a vector x laid out like a field (the Krylov operator of
        emg3d/solver.py:686-702: core.amat_x into a zero field, negated)."""
        c = self._level_on(x, x)
        _lib.check(_lib.lib().emg3d_dev_apply_operator(ctypes.byref(c), *self.parts(out), _stream()),
                   'emg3d_dev_apply_operator')
        return out

    def residual_sumsq(self, x, b):
        """sum |b - A x|^2 as a device tensor (no synchronisation): the true residual of a Krylov
        iterate, without copying x / b into the level's own buffers."""
        c = self._level_on(x, b)
        w = self.work
        _lib.check(_lib.lib().emg3d_dev_residual(ctypes.byref(c), None, None, None, _ptr(w.ws), w.ws.numel(),
                                                 _ptr(w.sumsq), _stream()), 'emg3d_dev_residual')
        return w.sumsq

    def zero_field(self):
        """e <- 0 (a fill kernel on the stream)."""
        _lib.check(_lib.lib().emg3d_dev_zero(_ptr(self.e), self.e.numel() * self.e.element_size(), _stream()),
                   'emg3d_dev_zero')

    def pec_zero(self):
        nx, ny, nz = self.grid.shape_cells
        _lib.check(_lib.lib().emg3d_dev_pec_zero(*self.parts(self.e), nx, ny, nz,
                                                 self.is_complex, _stream()), 'emg3d_dev_pec_zero')

    # ------------------------------------------------------------- grid transfer ------
    def child(self, sc_dir):
        """Coarse level for semicoarsening code sc_dir (0..6), built on first use:
        coarse grid (solver.py:899-905), summed model parameters (:916-926), restriction
        weights (:931) and prolongation tables."""
        if sc_dir in self.children:
            return self.children[sc_dir]
        lib = _lib.lib()
        g = self.grid
        cx, cy, cz = coarsen_flags(sc_dir)
        rx, ry, rz = (2 if cx else 1), (2 if cy else 1), (2 if cz else 1)
        ch = [np.diff(g.nodes_x[::rx]), np.diff(g.nodes_y[::ry]), np.diff(g.nodes_z[::rz])]
        cgrid = meshes.BaseMesh(ch, g.origin)
        nx, ny, nz = g.shape_cells

        def restrict_param(p, is_complex, dtype):
            out = torch.empty(cgrid.n_cells, dtype=dtype, device=self.device)
            _lib.check(lib.emg3d_dev_restrict_param(_ptr(out), _ptr(p), nx, ny, nz, sc_dir,
                                                    is_complex, _stream()),
                       'emg3d_dev_restrict_param')
            return out
        ceta_x = restrict_param(self.eta_x, self.is_complex, self.dtype)
        ceta_y = (restrict_param(self.eta_y, self.is_complex, self.dtype)
                  if self.case in ('HTI', 'triaxial') else ceta_x)
        ceta_z = (restrict_param(self.eta_z, self.is_complex, self.dtype)
                  if self.case in ('VTI', 'triaxial') else ceta_x)
        czeta = restrict_param(self.zeta, 0, torch.float64)
        clevel = DeviceLevel(cgrid, self.case, ceta_x, ceta_y, ceta_z, czeta, self.dtype,
                             self.work, self.device, self.batch, self.__dict__.get('_child_flags', self.flags))

        # restriction weights (only for coarsened directions; others are never read) and
        # prolongation tables: all 1-D arrays of the link go up in ONE float64 and ONE int32
        # transfer (a dozen tiny uploads per level were a third of the level-build time)
        fparts, slots = [], []
        for d, coarsened in enumerate((cx, cy, cz)):
            if coarsened:
                nodes = (g.nodes_x, g.nodes_y, g.nodes_z)[d]
                cc = (g.cell_centers_x, g.cell_centers_y, g.cell_centers_z)[d]
                cnodes = (cgrid.nodes_x, cgrid.nodes_y, cgrid.nodes_z)[d]
                ccc = (cgrid.cell_centers_x, cgrid.cell_centers_y, cgrid.cell_centers_z)[d]
                for w in _core.restrict_weights(nodes, cc, g.h[d], cnodes, ccc, cgrid.h[d]):
                    slots.append(len(fparts))
                    fparts.append(np.ascontiguousarray(w, dtype=np.float64))
            else:
                slots += [None, None, None]
        tabs = [interp_table(cn, n) for cn, n in ((cgrid.nodes_x, g.nodes_x),
                                                  (cgrid.nodes_y, g.nodes_y),
                                                  (cgrid.nodes_z, g.nodes_z))]
        pw_slots = []
        for t in tabs:
            pw_slots.append(len(fparts))
            fparts.append(t[1])
        foff = np.cumsum([0] + [a.size for a in fparts])
        fbuf = self.work.upload(np.concatenate(fparts))
        ioff = np.cumsum([0] + [t[0].size for t in tabs])
        ibuf = self.work.upload(np.concatenate([t[0] for t in tabs]))
        wptr = [(_ptr(fbuf, int(foff[i])) if i is not None else None) for i in slots]
        pwptr = [_ptr(fbuf, int(foff[i])) for i in pw_slots]
        ilptr = [_ptr(ibuf, int(ioff[d])) for d in range(3)]
        link = {'level': clevel, 'buffers': (fbuf, ibuf), 'wptr': wptr, 'ilptr': ilptr, 'pwptr': pwptr}
        self.children[sc_dir] = link
        return link

    def restrict_to(self, sc_dir):
        """csfield <- R(self.r); cefield <- 0 (solver.py:937-941). Returns the child."""
        link = self.child(sc_dir)
        c = link['level']
        nx, ny, nz = self.grid.shape_cells
        _lib.check(_lib.lib().emg3d_dev_restrict_clear_batch(
            *c.parts(c.s), *c.parts(c.e), *self.parts(self.r), *link['wptr'], nx, ny, nz, sc_dir,
            self.is_complex, self.batch, self.grid.n_edges, c.grid.n_edges, _stream()), 'emg3d_dev_restrict')
        return c

    def prolong_from(self, sc_dir):
        """self.e += P child.e (solver.py:947-1019)."""
        link = self.children[sc_dir]
        c = link['level']
        nx, ny, nz = self.grid.shape_cells
        _lib.check(_lib.lib().emg3d_dev_prolong_batch(
            *self.parts(self.e), *c.parts(c.e), *link['ilptr'], *link['pwptr'], nx, ny, nz, sc_dir,
            self.is_complex, self.batch, self.grid.n_edges, c.grid.n_edges, _stream()),
            'emg3d_dev_prolong')
