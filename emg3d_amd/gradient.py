"""Data misfit and adjoint-state gradient.

The reference obtains both from ``Simulation`` (emg3d/simulations.py:943-1094 ``gradient``,
:1097-1190 ``misfit``, :1193-1268 ``_bcompute`` / ``_get_rfield``) whose bookkeeping lives in
xarray datasets and whose solves run in a process pool. Here the same computation is a plain
function over (source, frequency) pairs, sharded over the ranks of the process group like the
forward solves of ``parallel.compute``:

    per pair, on its GPU
      1. forward solve                      -> efield stays in HBM
      2. responses at the receivers         (linear interpolation on the device, as the exact
                                             adjoint requires, simulations.py:975-983)
      3. residual = synthetic - observed;   misfit += sum w |residual|^2 / 2
      4. residual source field              = sum over receivers of a point dipole at the receiver
                                              with strength conj(residual w / (-s mu0))
      5. back-propagated solve, same hierarchy (levels, line factors, graphs)  -> bfield in HBM
      6. gradient += cells(real(bfield s mu0 efield))      (emg3d_dev_gradient_accumulate)
    once
      7. all-reduce of the cell gradient and of the misfit over the ranks (RCCL)
      8. anisotropy bookkeeping and the derivative chain of the property mapping (host, one pass)

Computational grids that differ from the model grid (``grids=``): the model goes to the pair's grid
by volume averaging (``Model.interpolate_to_grid``), the cell gradient comes back through the
adjoint of the linear averaging (``models._VolumeAverage.adjoint_add``; the reference takes that
operator from discretize, ``maps._interp_volume_average_adj``, emg3d/maps.py:722-750).

Magnetic point receivers (``magnetic=``): responses = ``get_magnetic_field`` interpolated linearly to the
point, adjoint source = the transpose of exactly that map (``fields.get_magnetic_point_source_field``;
the reference builds its ``_point_vector_magnetic`` from discretize's operators, emg3d/fields.py:749-789).

Limits as in the reference: no epsilon_r / mu_r.
"""
import numpy as np

from emg3d_amd import fields, models
from emg3d_amd.fields import Field

__all__ = ['misfit_and_gradient', 'residual_source_field']

_DCHAIN = {          # d sigma / d property, applied to the gradient w.r.t. conductivity (emg3d/maps.py:120-330)
    'Conductivity': lambda g, p: g,
    'Resistivity': lambda g, p: g * -(1.0 / p) ** 2,
    'LgConductivity': lambda g, p: g * (10.0 ** p) * np.log(10.0),
    'LgResistivity': lambda g, p: g * -(10.0 ** -p) * np.log(10.0),
    'LnConductivity': lambda g, p: g * np.exp(p),
    'LnResistivity': lambda g, p: g * -np.exp(-p),
}


def residual_source_field(grid, frequency, receivers, residual, weight, magnetic=None):
    """Source field of the back-propagation (``Simulation._get_rfield``, simulations.py:1235-1268):
    every receiver with data acts as a point dipole of strength ``conj(residual weight / (-s mu0))``
    -- an electric one (``TxElectricPoint``: the adjoint of the tri-linear receiver interpolation) or,
    for the receivers flagged in ``magnetic``, a magnetic one (``TxMagneticPoint``).
    ``receivers``: sequence of (x, y, z, azimuth, elevation); ``residual`` / ``weight``: one value per
    receiver (NaN residual: no data)."""
    rfield = Field(grid, frequency=frequency)
    strength = np.conj(np.asarray(residual) * np.asarray(weight) / -rfield.smu0)
    magnetic = np.zeros(len(strength), dtype=bool) if magnetic is None else np.asarray(magnetic, dtype=bool)
    index, value = [], []
    for rec, res, st, mag in zip(receivers, np.asarray(residual), strength, magnetic):
        if np.isnan(res):
            continue
        make = fields.get_magnetic_point_source_field if mag else fields.get_point_source_field
        part = make(grid, tuple(rec), frequency, strength=st)
        index.append(part._sparse[0])
        value.append(part._sparse[1])
    if index:
        index, value = np.concatenate(index), np.concatenate(value)
        np.add.at(rfield._field, index, value)
        nz = np.unique(index)
        rfield._sparse = (nz, rfield._field[nz].copy())
    else:
        rfield._sparse = (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=rfield._field.dtype))
    return rfield


def _receiver_tuple(receivers):
    r = np.asarray(receivers, dtype=float)
    return tuple(r[:, k] for k in range(5))


def misfit_and_gradient(model, sources, frequencies, receivers, observed, weights=None, solver_opts=None,
                        tol_gradient=1e-5, costs=None, grids=None, interpolate_opts=None, magnetic=None):
    """Misfit ``sum w |synthetic - observed|^2 / 2`` and its adjoint-state gradient with respect to
    the model properties (shape (nx, ny, nz) for isotropic models, (2, ...) HTI / VTI, (3, ...)
    tri-axial, as ``Simulation.gradient``).

    sources: dict name -> source coordinates; frequencies: dict name -> Hz; receivers: sequence of
    (x, y, z, azimuth, elevation) point receivers -- electric ones, or magnetic ones where the boolean
    sequence ``magnetic`` says so (responses in A/m: ``get_magnetic_field`` at the point); observed / weights: dict
    (source name, frequency name) -> one value per receiver (NaN: no data; weights default 1).
    grids: the computational grid of the pairs, if it is not the model's: one TensorMesh for all, or
    a dict (source name, frequency name) -> TensorMesh (pairs that are missing use the model grid);
    interpolate_opts: passed to ``Model.interpolate_to_grid`` (default: averaging on a log10 scale, as
    in the reference -- the way back is the adjoint of the LINEAR averaging either way, exact for
    ``{'log': False}`` with a conductivity model).
    With an initialised process group the pairs are sharded over the ranks and both results are
    all-reduced: every rank returns the complete misfit and gradient."""
    import torch
    from emg3d_amd import _lib, parallel, solver
    from emg3d_amd._device import _ptr, _stream
    _lib.require_gpu()
    for name, prop in (('el. permittivity', model.epsilon_r), ('magn. permeability', model.mu_r)):
        if prop is not None and not np.allclose(prop, 1.0):
            raise NotImplementedError(f"Gradient not implemented for {name}.")
    mgrid = model.grid
    opts = dict(solver_opts or {})
    opts.setdefault('sslsolver', True)
    rec = _receiver_tuple(receivers)
    mag = np.zeros(len(rec[0]), dtype=bool) if magnetic is None else np.asarray(magnetic, dtype=bool)
    pairs = parallel.srcfreq_pairs(sources, frequencies)
    rank, world = parallel.rank_and_world()
    mine = parallel.shard(len(pairs), rank, world, costs)
    dev = torch.device('cuda', torch.cuda.current_device())
    ncell = mgrid.n_cells
    grad = torch.zeros(3 * ncell, dtype=torch.float64, device=dev)
    misfit = 0.0
    hierarchies = {}
    on_grid = {}                 # per computational grid: (grid, model on it, cell volumes, averaging plan)
    info = {}

    def computational(pair):
        g = grids.get(pair) if isinstance(grids, dict) else grids
        if g is None or g == mgrid:
            g = mgrid
        key = id(g) if g is not mgrid else 0
        if key not in on_grid:
            vol = torch.from_numpy(np.ascontiguousarray(g.cell_volumes, dtype=np.float64)).to(dev)
            plan = None if g is mgrid else models._VolumeAverage(mgrid, g)
            on_grid[key] = (g, model.interpolate_to_grid(g, **(interpolate_opts or {})), vol, plan)
        return (key,) + on_grid[key]

    for i in mine:
        sname, fname = pairs[i]
        freq = frequencies[fname]
        gkey, grid, gmodel, vol, plan = computational((sname, fname))
        nx, ny, nz = grid.shape_cells
        sfield = fields.get_source_field(grid, sources[sname], freq)
        hkey = (complex(sfield.sval), gkey)
        hier = hierarchies.get(hkey)
        if hier is None:
            hierarchies.clear()                       # one (frequency, grid) at a time in HBM
            hier = hierarchies[hkey] = solver.Hierarchy(models.VolumeModel(gmodel, sfield))
        top = hier.top
        _, finfo = solver.solve(gmodel, sfield, return_info=True, always_return=True, hierarchy=hier, _download=False,
                                _sparse_source=True, **opts)
        e_fwd = torch.empty_like(top.e)
        _lib.check(_lib.lib().emg3d_dev_copy(_ptr(e_fwd), _ptr(top.e), top.e.numel() * top.e.element_size(), _stream()),
                   'emg3d_dev_copy')
        meta = Field(grid, frequency=freq)
        synthetic = fields.get_responses(meta, e_fwd, rec, 'linear', magnetic=mag)
        obs = np.asarray(observed[(sname, fname)])
        w = np.ones(obs.shape) if weights is None else np.asarray(weights[(sname, fname)], dtype=float)
        residual = synthetic - obs
        have = ~np.isnan(residual)
        misfit += float(np.sum(w[have] * (residual[have].conj() * residual[have])).real) / 2
        if not np.any(have & (residual != 0)):
            # no data (or a perfect fit) for this pair: no residual source, nothing back-propagated,
            # no contribution (the reference drops such pairs, emg3d/simulations.py:1120-1190)
            info[(sname, fname)] = {'forward': finfo, 'backward': None, 'synthetic': synthetic}
            continue
        rfield = residual_source_field(grid, freq, receivers, residual, w, mag)
        _, binfo = solver.solve(gmodel, rfield, return_info=True, always_return=True, hierarchy=hier, _download=False,
                                _sparse_source=True, **{**opts, 'tol': tol_gradient})
        smu0 = complex(sfield.smu0)
        o1, o2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
        if plan is None:
            gtarget, nc = grad, ncell
        else:                                          # cell gradient on the computational grid first
            nc = grid.n_cells
            gtarget = torch.empty(3 * nc, dtype=torch.float64, device=dev)
            _lib.check(_lib.lib().emg3d_dev_zero(_ptr(gtarget), gtarget.numel() * 8, _stream()), 'emg3d_dev_zero')
        _lib.check(_lib.lib().emg3d_dev_gradient_accumulate(
            nx, ny, nz, int(top.is_complex), _ptr(e_fwd), _ptr(e_fwd, o1), _ptr(e_fwd, o2),
            _ptr(top.e), _ptr(top.e, o1), _ptr(top.e, o2), smu0.real, smu0.imag, _ptr(vol),
            _ptr(gtarget), _ptr(gtarget, nc), _ptr(gtarget, 2 * nc), _stream()), 'emg3d_dev_gradient_accumulate')
        if plan is not None:                           # ... and back to the model grid: grad += P^T g
            for k in range(3):
                plan.adjoint_add(gtarget[k * nc:(k + 1) * nc], grad[k * ncell:(k + 1) * ncell])
        info[(sname, fname)] = {'forward': finfo, 'backward': binfo, 'synthetic': synthetic}
    # the one collective of the path: sum over the ranks
    tm = torch.tensor([misfit], dtype=torch.float64, device=dev)
    if world > 1:
        dist = parallel._dist()
        cdev = dev if dist.get_backend() == 'nccl' else torch.device('cpu')
        g, m = grad.to(cdev), tm.to(cdev)
        dist.all_reduce(g)
        dist.all_reduce(m)
        grad, tm = g, m
    misfit = float(tm.cpu()[0])
    g3 = np.stack([grad[k * ncell:(k + 1) * ncell].cpu().numpy().reshape(mgrid.shape_cells, order='F') for k in range(3)])

    # anisotropy bookkeeping + derivative chain of the mapping (simulations.py:1070-1090)
    chain = _DCHAIN[model.mapping]
    keep = [0]
    if model.case in ('HTI', 'triaxial'):
        g3[1] = chain(g3[1], model.property_y)
        keep.append(1)
    else:
        g3[0] += g3[1]
    if model.case in ('VTI', 'triaxial'):
        g3[2] = chain(g3[2], model.property_z)
        keep.append(2)
    else:
        g3[0] += g3[2]
    g3[0] = chain(g3[0], model.property_x)
    return misfit, np.asfortranarray(g3[keep].squeeze()), info

