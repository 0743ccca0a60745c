"""Model containers (host side).

``Model`` holds the electrical properties on the cells; ``VolumeModel`` turns them into
the two volume-integrated coefficient sets the kernels consume, with the semantics of
the reference's ``emg3d.models.VolumeModel`` (reference emg3d/models.py:627-717):

    eta_{x,y,z} = -s mu_0 V (sigma_{x,y,z} [+ s eps_0 eps_r]),      zeta = V / mu_r,

including its aliasing rule: ``eta_y`` / ``eta_z`` ARE ``eta_x`` unless the anisotropy
case has an independent y / z property (emg3d/models.py:693-712).
"""
import numpy as np

from emg3d_amd import meshes
from emg3d_amd.fields import EPSILON_0

__all__ = ['Model', 'VolumeModel']

_MAPS = {
    'Resistivity': lambda p: 1.0 / p,
    'Conductivity': lambda p: p,
    'LgResistivity': lambda p: 10.0 ** (-p),
    'LgConductivity': lambda p: 10.0 ** p,
    'LnResistivity': lambda p: np.exp(-p),
    'LnConductivity': lambda p: np.exp(p),
}


# codes of emg3d_dev_volume_model (include/emg3d_amd.h)
_MAP_CODES = {'Conductivity': 0, 'Resistivity': 1, 'LgConductivity': 2, 'LgResistivity': 3,
              'LnConductivity': 4, 'LnResistivity': 5}


class Model:
    """Resistivity/conductivity model on a tensor mesh (subset of emg3d.models.Model,
    reference emg3d/models.py:33-620, needed to feed the solver)."""

    def __init__(self, grid, property_x=1., property_y=None, property_z=None, mu_r=None,
                 epsilon_r=None, mapping='Resistivity'):
        if mapping not in _MAPS:
            raise ValueError(f"Unknown mapping {mapping!r}; known: {sorted(_MAPS)}.")
        self.grid = grid
        self.mapping = mapping
        self.shape = grid.shape_cells
        self.property_x = self._init(property_x, 'property_x')
        self.property_y = self._init(property_y, 'property_y')
        self.property_z = self._init(property_z, 'property_z')
        self.mu_r = self._init(mu_r, 'mu_r')
        self.epsilon_r = self._init(epsilon_r, 'epsilon_r')
        self.case = {(False, False): 'isotropic', (True, False): 'HTI', (False, True): 'VTI',
                     (True, True): 'triaxial'}[(self.property_y is not None,
                                                self.property_z is not None)]

    def __setattr__(self, name, value):
        # a replaced property array invalidates its HBM snapshot (parallel.broadcast_model)
        if name in ('property_x', 'property_y', 'property_z', 'mu_r', 'epsilon_r'):
            self.__dict__.get('_device_props', {}).pop(name, None)
        object.__setattr__(self, name, value)

    def _init(self, value, name):
        if value is None:
            return None
        value = np.asarray(value, dtype=np.float64)
        if value.ndim == 0:
            out = np.full(self.shape, float(value), order='F')
        elif value.shape == self.shape:
            out = np.asfortranarray(value)
        elif value.size == self.grid.n_cells:
            out = np.asfortranarray(value.reshape(self.shape, order='F'))
        else:
            raise ValueError(f"Shape of {name} must be () or {self.shape}; provided: "
                             f"{value.shape}.")
        if not np.all(np.isfinite(out)):
            raise ValueError(f"`{name}` must be all finite.")
        log_mapped = self.mapping.startswith('L') and name.startswith('property')
        if not log_mapped and np.any(out <= 0):
            raise ValueError(f"`{name}` must be all positive.")
        return out

    def __repr__(self):
        return (f"Model: {self.mapping}; {self.case}; {self.shape[0]} x {self.shape[1]} x "
                f"{self.shape[2]} ({self.grid.n_cells:,})")

    def conductivity(self, name):
        prop = getattr(self, name)
        return None if prop is None else _MAPS[self.mapping](prop)

    def interpolate_to_grid(self, grid, **interpolate_opts):
        """The model on another grid; same call as the reference's
        ``Model.interpolate_to_grid`` (emg3d/models.py:322-366): the model itself if the grids
        are identical, else every property by volume averaging (``method='volume'``, on a
        log10 scale unless the mapping is logarithmic already; nearest extrapolation outside
        the model grid), computed by ``emg3d_dev_volume_average`` (maps.py:555-664)."""
        if grid == self.grid:
            return self
        opts = {'method': 'volume', 'log': not self.mapping.startswith('L'), **interpolate_opts}
        if opts.pop('method') != 'volume':
            raise NotImplementedError("emg3d_amd: models are re-gridded with method='volume' only.")
        log = bool(opts.pop('log'))
        opts.pop('extrapolate', None)        # 'volume' always extrapolates with the nearest cell
        if opts:
            raise TypeError(f"Unexpected interpolation options: {sorted(opts)}")
        new = {}
        plan = _VolumeAverage(self.grid, grid)
        for name in ('property_x', 'property_y', 'property_z', 'mu_r', 'epsilon_r'):
            values = getattr(self, name)
            if values is not None:
                new[name] = plan(values, log)
        return Model(grid, mapping=self.mapping, **new)


def _volume_average_weights(x_i, x_o):
    """Segments of one axis for the volume averaging (maps._volume_average_weights, reference
    emg3d/maps.py:619-664): the union of input and output nodes cuts the axis; kept are the
    segments whose centre lies inside the output grid, with their length, input cell (clamped
    = nearest extrapolation) and output cell. Returned grouped by output cell: offsets (m+1),
    lengths, input cells."""
    xs = np.unique(np.concatenate((x_i, x_o)))
    centre = 0.5 * (xs[:-1] + xs[1:])
    keep = (x_o[0] <= centre) & (centre <= x_o[-1])
    centre, w = centre[keep], np.diff(xs)[keep]
    cell_in = np.clip(np.searchsorted(x_i[:-1], centre, side='right') - 1, 0, x_i.size - 1)
    cell_out = np.clip(np.searchsorted(x_o[:-1], centre, side='right') - 1, 0, x_o.size - 1)
    seg = np.searchsorted(cell_out, np.arange(x_o.size), side='left')       # cell_out is sorted
    return seg.astype(np.int32), w, cell_in.astype(np.int32)


class _VolumeAverage:
    """Tables of one (grid -> new grid) pair on the device; call with a property array."""

    def __init__(self, grid, new_grid):
        import torch
        from emg3d_amd import _lib
        _lib.require_gpu()
        self.dev = torch.device('cuda', torch.cuda.current_device())
        self.shape_in, self.shape_out = grid.shape_cells, new_grid.shape_cells
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)          # noqa: E731
        self.tabs = []
        for a, b in ((grid.nodes_x, new_grid.nodes_x), (grid.nodes_y, new_grid.nodes_y),
                     (grid.nodes_z, new_grid.nodes_z)):
            self.tabs.append([up(t) for t in _volume_average_weights(a, b)])
        self.vol = up(new_grid.cell_volumes)

    def __call__(self, values, log):
        import torch
        from emg3d_amd import _lib
        from emg3d_amd._device import _ptr, _stream
        v = torch.from_numpy(np.ascontiguousarray(values.ravel('F'))).to(self.dev)
        out = torch.empty(int(np.prod(self.shape_out)), dtype=torch.float64, device=self.dev)
        (sx, wx, ix), (sy, wy, iy), (sz, wz, iz) = self.tabs
        _lib.check(_lib.lib().emg3d_dev_volume_average(
            _ptr(v), *self.shape_in, _ptr(sx), _ptr(sy), _ptr(sz), _ptr(wx), _ptr(wy), _ptr(wz),
            _ptr(ix), _ptr(iy), _ptr(iz), _ptr(self.vol), *self.shape_out, _ptr(out), int(bool(log)), _stream()),
            'emg3d_dev_volume_average')
        return out.cpu().numpy().reshape(self.shape_out, order='F')

    def adjoint_add(self, nval, oval):
        """oval += P^T nval for device arrays: ``nval`` on the new grid (cells, F-order), ``oval`` on
        the original grid -- P the linear averaging this plan applies (``log=False``). The gradient's
        way back from a computational grid to the model grid (reference
        ``maps._interp_volume_average_adj``, emg3d/maps.py:722-750, which takes the operator from
        discretize; here it is the transpose of the plan's own tables: the same kernel walks the
        segments grouped by ORIGINAL cell)."""
        from emg3d_amd import _lib
        from emg3d_amd._device import _ptr, _stream
        if not hasattr(self, 'tabs_t'):
            import torch
            self.tabs_t = []
            for (seg, w, cell_in), n_in in zip(self.tabs, self.shape_in):
                seg, w, cell_in = seg.cpu().numpy(), w.cpu().numpy(), cell_in.cpu().numpy()
                cell_out = np.repeat(np.arange(seg.size - 1, dtype=np.int32), np.diff(seg))
                order = np.argsort(cell_in, kind='stable')
                seg_t = np.searchsorted(cell_in[order], np.arange(n_in + 1), side='left').astype(np.int32)
                self.tabs_t.append([torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
                                    for a in (seg_t, w[order], cell_out[order])])
        (sx, wx, ix), (sy, wy, iy), (sz, wz, iz) = self.tabs_t
        _lib.check(_lib.lib().emg3d_dev_volume_average(
            _ptr(nval), *self.shape_out, _ptr(sx), _ptr(sy), _ptr(sz), _ptr(wx), _ptr(wy), _ptr(wz),
            _ptr(ix), _ptr(iy), _ptr(iz), _ptr(self.vol), *self.shape_in, _ptr(oval), 2, _stream()),
            'emg3d_dev_volume_average')


class VolumeModel:
    """Volume-integrated eta_{x,y,z} and zeta for one frequency (taken from `sfield`);
    semantics of the reference's class (emg3d/models.py:627-717).

    The host arrays are formed on first access only: the solver builds the same quantities
    directly in HBM (``device_arrays``), which for a 128^3 model replaces ~150 ms of NumPy
    work and a 100 MB upload by a 17 MB upload per conductivity array and a few
    element-wise device kernels.
    """

    def __init__(self, model, sfield):
        self.case = model.case
        self.grid = meshes.BaseMesh(model.grid.h, model.grid.origin)
        if sfield.sval is None:
            raise ValueError("Source field is missing frequency information.")
        self._model = model
        self._sval, self._smu0 = sfield.sval, sfield.smu0
        self._host = None

    # ---- the formulas (emg3d/models.py:654-691): eta = -s mu0 V (sigma [+ s eps0 eps_r]),
    #      zeta = V / mu_r -- evaluated with NumPy (host) or torch (device), same order
    def _conductivities(self):
        return [self._model.conductivity(n) for n in ('property_x', 'property_y', 'property_z')]

    def _build_host(self):
        if self._host is not None:
            return self._host
        model = self._model
        vol = self.grid.cell_volumes.reshape(model.shape, order='F')
        etas = []
        for cond in self._conductivities():
            if cond is None:
                etas.append(None)
            elif model.epsilon_r is None:                       # diffusive approximation
                etas.append(np.asfortranarray(-self._smu0 * vol * cond))
            else:
                smu = self._sval * EPSILON_0 * model.epsilon_r
                etas.append(np.asfortranarray(-self._smu0 * vol * (cond + smu)))
        zeta = np.asfortranarray(vol.copy() if model.mu_r is None else vol / model.mu_r)
        self._host = (etas[0], etas[1], etas[2], zeta)
        return self._host

    def device_arrays(self, device):
        """(eta_x, eta_y, eta_z, zeta) as flat F-ordered torch tensors on `device`, with the
        reference's aliasing (eta_y / eta_z are eta_x's tensor unless the model has them), formed
        by ``emg3d_dev_volume_model`` from the property arrays: those that arrived through a
        device broadcast (``parallel.broadcast_model``) are already in HBM, the others go up as
        they are (17 MB per array at 128^3)."""
        import torch
        from emg3d_amd import _lib
        from emg3d_amd._device import _ptr, _stream
        model = self._model
        cplx = np.iscomplexobj(self._smu0)
        dtype = torch.complex128 if cplx else torch.float64
        resident = {k: v for k, v in getattr(model, '_device_props', {}).items() if v.device == torch.device(device)}

        def prop(name):
            if name in resident:
                return resident[name]
            a = getattr(model, name)
            if a is None:
                return None
            a = np.broadcast_to(np.asarray(a, dtype=np.float64), model.shape).ravel('F')
            if not a.flags.writeable or not a.flags.c_contiguous:
                a = np.array(a)                         # torch wants a writable, dense buffer
            return torch.from_numpy(a).to(device)
        px, eps, mu = prop('property_x'), prop('epsilon_r'), prop('mu_r')
        py = prop('property_y') if self.case in ('HTI', 'triaxial') else None
        pz = prop('property_z') if self.case in ('VTI', 'triaxial') else None
        h = torch.from_numpy(np.concatenate([np.asarray(w, dtype=np.float64) for w in self.grid.h])).to(device)
        nx, ny, nz = self.grid.shape_cells
        n = self.grid.n_cells
        ex = torch.empty(n, dtype=dtype, device=device)
        ey = torch.empty(n, dtype=dtype, device=device) if py is not None else ex
        ez = torch.empty(n, dtype=dtype, device=device) if pz is not None else ex
        zeta = torch.empty(n, dtype=torch.float64, device=device)
        smu0 = complex(self._smu0)
        seps0 = complex(self._sval) * EPSILON_0
        opt = lambda t: _ptr(t) if t is not None else None                    # noqa: E731
        _lib.check(_lib.lib().emg3d_dev_volume_model(
            nx, ny, nz, int(cplx), _ptr(px), opt(py), opt(pz), opt(eps), opt(mu), _MAP_CODES[model.mapping],
            _ptr(h), _ptr(h, nx), _ptr(h, nx + ny), smu0.real, smu0.imag, seps0.real, seps0.imag,
            _ptr(ex), opt(ey if py is not None else None), opt(ez if pz is not None else None), _ptr(zeta), _stream()),
            'emg3d_dev_volume_model')
        return ex, ey, ez, zeta

    @property
    def eta_x(self):
        return self._build_host()[0]

    @property
    def eta_y(self):
        h = self._build_host()
        return h[1] if self.case in ('HTI', 'triaxial') else h[0]

    @property
    def eta_z(self):
        h = self._build_host()
        return h[2] if self.case in ('VTI', 'triaxial') else h[0]

    @property
    def zeta(self):
        return self._build_host()[3]
