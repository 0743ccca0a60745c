"""Generate the golden vectors in tests/golden/ from the reference (build container only).

The reference is imported un-jitted from /root/reference (stubs for numba /
empymod / scooby in tools/oracle_stubs; they contain no reference code). The
fixtures are DATA: inputs and the reference's outputs. Nothing of the
reference's source travels.

    python tools/make_golden.py            # all fixtures (takes a few minutes)

Fixtures written (each ≤ 1.5 MB):
    tests/golden/regression_small.npz   arrays of the reference's own golden file
                                        tests/data/regression.npz (res / reg_2 / lap)
    tests/golden/kernels.npz            per-function input/output vectors of emg3d.core
                                        and of the solver.py wrappers on small grids
    tests/golden/solves.npz             converged reference solves (tol 1e-10) incl.
                                        per-cycle error history
    tests/golden/receivers.npz          magnetic field and receiver responses (cubic / linear)
                                        of the reference for random fields
    tests/golden/gridding.npz           models re-gridded by the reference (volume averaging)
    tests/golden/sources.npz            (`sources`) source vectors incl. magnetic dipoles (square loops)
Metadata (scipy version, mu_0, seeds) is stored in every file.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, 'oracle_stubs'), '/root/reference',
                '/root/reference/tests']

import scipy  # noqa: E402
import scipy.constants  # noqa: E402
import emg3d  # noqa: E402  (the reference)
from emg3d import core as rcore, solver as rsolver  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
META = {
    'meta_scipy_version': scipy.__version__,
    'meta_numpy_version': np.__version__,
    'meta_mu_0': scipy.constants.mu_0,
    'meta_epsilon_0': scipy.constants.epsilon_0,
    'meta_reference': 'emsig/emg3d @ /root/reference (post v1.8.7), un-jitted',
}


def widths(ncore, npad, width, factor):
    pad = ((np.ones(npad) * np.abs(factor)) ** (np.arange(npad) + 1)) * width
    return np.r_[pad[::-1], np.ones(ncore) * width, pad]


def vm_arrays(vm):
    return {'eta_x': vm.eta_x, 'eta_y': vm.eta_y, 'eta_z': vm.eta_z, 'zeta': vm.zeta}


# ------------------------------------------------------------------------------
def regression_small():
    d = emg3d.load('/root/reference/tests/data/regression.npz', verb=0)
    out = dict(META)
    out['meta_source_file'] = 'tests/data/regression.npz (reference test data)'
    out['meta_file_version'] = d['_version']
    for key in ('res', 'lap'):
        r = d[key]
        g = r['grid']
        out[f'{key}_hx'], out[f'{key}_hy'], out[f'{key}_hz'] = g.h
        out[f'{key}_origin'] = np.asarray(g.origin, dtype=float)
        m = r['input_model']
        out[f'{key}_res_xyz'] = np.array([m['property_x'], m['property_y'], m['property_z']], float)
        out[f'{key}_source'] = np.asarray(r['input_source']['source'], float)
        out[f'{key}_frequency'] = float(r['input_source']['frequency'])
        out[f'{key}_sfield'] = np.asarray(r['sfield'].field)
        for k in ('Fresult', 'Wresult', 'Vresult', 'bicresult'):
            if k in r:
                out[f'{key}_{k}'] = np.asarray(r[k].field)
    r = d['reg_2']
    g = r['grid']
    out['reg2_hx'], out['reg2_hy'], out['reg2_hz'] = g.h
    out['reg2_origin'] = np.asarray(g.origin, dtype=float)
    out['reg2_res_x'] = np.asarray(r['model'].property_x)
    out['reg2_res_y'] = np.asarray(r['model'].property_y)
    out['reg2_res_z'] = np.asarray(r['model'].property_z)
    out['reg2_frequency'] = float(r['sfield'].frequency)
    out['reg2_sfield'] = np.asarray(r['sfield'].field)
    out['reg2_result'] = np.asarray(r['result'].field)
    for k, v in r['inp'].items():
        out[f'reg2_inp_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'regression_small.npz'), **out)
    print('regression_small.npz written')


# ------------------------------------------------------------------------------
def kernel_cases():
    """(name, shape, dtype, case) small stretched grids; 2-cell directions included."""
    return [
        ('c_tri', (6, 4, 8), np.complex128, 'triaxial'),
        ('c_iso', (4, 8, 6), np.complex128, 'isotropic'),
        ('c_vti', (8, 6, 4), np.complex128, 'VTI'),
        ('r_tri', (6, 8, 4), np.float64, 'triaxial'),
        ('r_iso', (4, 4, 4), np.float64, 'isotropic'),
        ('c_x2', (2, 6, 4), np.complex128, 'triaxial'),
        ('c_y2', (6, 2, 4), np.complex128, 'HTI'),
        ('c_z2', (4, 6, 2), np.complex128, 'triaxial'),
    ]


def build_case(rng, shape, dtype, case):
    nx, ny, nz = shape
    hx = widths(nx - 2 * (nx // 4), nx // 4, 40., 1.3) if nx > 2 else np.array([40., 55.])
    hy = widths(ny - 2 * (ny // 4), ny // 4, 50., 1.2) if ny > 2 else np.array([50., 45.])
    hz = widths(nz - 2 * (nz // 4), nz // 4, 30., 1.4) if nz > 2 else np.array([30., 42.])
    grid = emg3d.TensorMesh([hx, hy, hz], origin=(-hx.sum() / 2, -hy.sum() / 2, -hz.sum()))
    n = grid.n_cells
    px = 10 ** rng.uniform(-1, 2, n)
    kw = {'property_x': px}
    if case in ('HTI', 'triaxial'):
        kw['property_y'] = px * rng.uniform(0.5, 2, n)
    if case in ('VTI', 'triaxial'):
        kw['property_z'] = px * rng.uniform(1, 3, n)
    model = emg3d.Model(grid, mapping='Resistivity', **kw)
    freq = 0.77 if dtype == np.complex128 else -1.9
    sfield = emg3d.Field(grid, frequency=freq)
    vm = emg3d.models.VolumeModel(model, sfield)

    def rand_field(pec):
        f = emg3d.Field(grid, frequency=freq)
        v = rng.standard_normal(f.field.size)
        if dtype == np.complex128:
            v = v + 1j * rng.standard_normal(f.field.size)
        f.field = v
        if pec:
            f.fx[:, 0, :] = f.fx[:, -1, :] = 0.
            f.fx[:, :, 0] = f.fx[:, :, -1] = 0.
            f.fy[0, :, :] = f.fy[-1, :, :] = 0.
            f.fy[:, :, 0] = f.fy[:, :, -1] = 0.
            f.fz[0, :, :] = f.fz[-1, :, :] = 0.
            f.fz[:, 0, :] = f.fz[:, -1, :] = 0.
        return f
    return grid, model, vm, freq, rand_field


def kernels():
    rng = np.random.default_rng(20260928)
    out = dict(META)
    out['meta_seed'] = 20260928
    names = []
    for name, shape, dtype, case in kernel_cases():
        names.append(name)
        grid, model, vm, freq, rand_field = build_case(rng, shape, dtype, case)
        p = name + '_'
        out[p + 'hx'], out[p + 'hy'], out[p + 'hz'] = grid.h
        out[p + 'origin'] = np.asarray(grid.origin, float)
        out[p + 'case'] = case
        out[p + 'frequency'] = freq
        out[p + 'res_x'] = model.property_x
        if model.property_y is not None:
            out[p + 'res_y'] = model.property_y
        if model.property_z is not None:
            out[p + 'res_z'] = model.property_z
        for k, v in vm_arrays(vm).items():
            out[p + k] = np.asarray(v)
        h = grid.h
        vma = (vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta)

        # amat_x with non-zero boundary entries
        e, s = rand_field(False), rand_field(False)
        out[p + 'amat_e'], out[p + 'amat_r_in'] = e.field.copy(), s.field.copy()
        r = s.copy()
        rcore.amat_x(r.fx, r.fy, r.fz, e.fx, e.fy, e.fz, *vma, *h)
        out[p + 'amat_r_out'] = r.field.copy()

        # smoothers through the wrapper's kernels, nu = 1 and 2 (first sweep backward)
        e, s = rand_field(True), rand_field(True)
        out[p + 'gs_e_in'], out[p + 'gs_s'] = e.field.copy(), s.field.copy()
        for fn in ('gauss_seidel', 'gauss_seidel_x', 'gauss_seidel_y', 'gauss_seidel_z'):
            for nu in (1, 2):
                f = e.copy()
                getattr(rcore, fn)(f.fx, f.fy, f.fz, s.fx, s.fy, s.fz, *vma, *h, nu)
                out[p + f'{fn}_nu{nu}'] = f.field.copy()

        # residual norm (solver.residual)
        out[p + 'residual_norm'] = rsolver.residual(vm, s, e, True)

        # restriction (model + field) and prolongation for every sc_dir that is valid
        res = rand_field(False)
        out[p + 'restrict_res'] = res.field.copy()
        for sc_dir in range(7):
            rx, ry, rz = [1 if sc_dir in s_ else 2 for s_ in ([1, 5, 6], [2, 4, 6], [3, 4, 5])]
            if any(r_ == 2 and (n_ % 2 != 0 or n_ < 4) for r_, n_ in zip((rx, ry, rz), shape)):
                continue
            cmodel, cs, ce = rsolver.restriction(vm, s, res, sc_dir)
            q = p + f'sc{sc_dir}_'
            out[q + 'csfield'] = cs.field.copy()
            out[q + 'ceta_x'] = np.asarray(cmodel.eta_x)
            out[q + 'ceta_y'] = np.asarray(cmodel.eta_y)
            out[q + 'ceta_z'] = np.asarray(cmodel.eta_z)
            out[q + 'czeta'] = np.asarray(cmodel.zeta)
            wx, wy, wz = rsolver._get_restriction_weights(vm.grid, cmodel.grid, sc_dir)
            for nm, w in zip('xyz', (wx, wy, wz)):
                out[q + f'w{nm}'] = np.array(w)
            # prolongation of a random coarse field onto a random fine field
            ce.field = (rng.standard_normal(ce.field.size) +
                        (1j * rng.standard_normal(ce.field.size) if dtype == np.complex128 else 0))
            fine = rand_field(True)
            out[q + 'prol_c'] = ce.field.copy()
            out[q + 'prol_f_in'] = fine.field.copy()
            rsolver.prolongation(fine, ce, sc_dir)
            out[q + 'prol_f_out'] = fine.field.copy()
    out['meta_cases'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'kernels.npz'), **out)
    print('kernels.npz written')


# ------------------------------------------------------------------------------
def solves():
    rng = np.random.default_rng(7)
    out = dict(META)
    names = []

    def run(name, grid, model, src, freq, **kw):
        t0 = time.time()
        sfield = emg3d.get_source_field(grid, src, freq)
        ef, info = rsolver.solve(model, sfield, return_info=True, sslsolver=False, verb=0, **kw)
        vm = emg3d.models.VolumeModel(model, sfield)
        p = name + '_'
        names.append(name)
        out[p + 'hx'], out[p + 'hy'], out[p + 'hz'] = grid.h
        out[p + 'origin'] = np.asarray(grid.origin, float)
        out[p + 'case'] = model.case
        out[p + 'frequency'] = float(freq)
        out[p + 'source'] = np.asarray(src, float)
        out[p + 'res_x'] = model.property_x
        if model.property_y is not None:
            out[p + 'res_y'] = model.property_y
        if model.property_z is not None:
            out[p + 'res_z'] = model.property_z
        for k, v in vm_arrays(vm).items():
            out[p + k] = np.asarray(v)
        out[p + 'sfield'] = np.asarray(sfield.field)
        out[p + 'efield'] = np.asarray(ef.field)
        out[p + 'it_mg'] = info['it_mg']
        out[p + 'exit_message'] = info['exit_message']
        out[p + 'error_at_cycle'] = info['error_at_cycle']
        out[p + 'ref_error'] = info['ref_error']
        for k, v in kw.items():
            out[p + 'kw_' + k] = v
        print(f"  {name}: {info['exit_message']} it={info['it_mg']} "
              f"rel={info['rel_error']:.3e}  ({time.time() - t0:.1f} s)")

    # (a) config-1 family: uniform fullspace, plain F-cycle (docs/dev/tests.rst:193-219), nx=16
    h = np.ones(16) * 50.
    grid = emg3d.TensorMesh([h, h, h], origin=(-400, -400, -400))
    model = emg3d.Model(grid, property_x=1., mapping='Resistivity')
    run('uni16_F', grid, model, (0, 0, 0, 0, 0), 1.0, semicoarsening=False,
        linerelaxation=False, cycle='F', tol=1e-10)

    # (b) stretched marine VTI, W-cycle + semicoarsening + line relaxation (config-2 family)
    hx = widths(8, 4, 50, 1.3); hz = widths(8, 4, 25, 1.4)
    grid = emg3d.TensorMesh([hx, hx, hz], origin=(-hx.sum() / 2, -hx.sum() / 2, -hz[:10].sum()))
    zc = grid.cell_centers_z
    rh = np.where(zc > -200, 0.3, 1.0)
    rv = np.where(zc > -200, 0.3, 2.0)
    px = np.tile(rh[None, None, :], (16, 16, 1)).ravel('F')
    pz = np.tile(rv[None, None, :], (16, 16, 1)).ravel('F')
    model = emg3d.Model(grid, property_x=px, property_z=pz, mapping='Resistivity')
    run('marine16_W', grid, model, (0, 0, -150, 0, 0), 1.0, semicoarsening=True,
        linerelaxation=True, cycle='W', tol=1e-10)

    # (c) tri-axial random blocky model, F-cycle, sc=123 lr=456, non-cubic grid
    hx = widths(4, 4, 30, 1.2); hy = widths(4, 2, 40, 1.3); hz = widths(8, 4, 20, 1.25)
    grid = emg3d.TensorMesh([hx, hy, hz], origin=(-hx.sum() / 2, -hy.sum() / 2, -hz.sum() / 2))
    lat = 10 ** rng.uniform(-0.5, 1.5, (3, 2, 4))
    px = np.kron(lat, np.ones((4, 4, 4))).ravel('F')
    model = emg3d.Model(grid, property_x=px, property_y=1.5 * px, property_z=2.5 * px,
                        mapping='Resistivity')
    run('tri12x8x16_F', grid, model, (5, -3, 2, 30, 10), 0.5, semicoarsening=123,
        linerelaxation=456, cycle='F', tol=1e-10)

    # (d) Laplace domain (real arithmetic), V-cycle, line relaxation 7
    h = widths(4, 2, 20, 1.3)
    grid = emg3d.TensorMesh([h, h, h], origin=(-h.sum() / 2, -h.sum() / 2, -h.sum() / 2))
    model = emg3d.Model(grid, property_x=1.5, property_y=2.0, property_z=3.3, mapping='Resistivity')
    run('lap8_V', grid, model, (3, 2, 1, 20, 40), -2 * np.pi, semicoarsening=False,
        linerelaxation=7, cycle='V', tol=1e-10)

    out['meta_cases'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'solves.npz'), **out)
    print('solves.npz written')


def solves32():
    """32^3 solves of the reference itself (several minutes, un-jitted): BASELINE.json config 1
    -- h = 50 m x 32, unit fullspace, x-dipole at the origin, 1 Hz, plain F-cycle -- at tol 1e-6
    (cycle count and error of SURVEY.md 8d: 6 cycles, 1.784e-07) and at tol 1e-10 (the field the
    GPU solve is compared with), and a 32^3 stretched marine VTI model with W-cycle,
    semicoarsening and line relaxation at tol 1e-10. Only inputs that cannot be re-derived and the
    converged fields are stored."""
    out = dict(META)
    names = []

    def run(name, grid, model, src, freq, tols, **kw):
        sfield = emg3d.get_source_field(grid, src, freq)
        p = name + '_'
        names.append(name)
        out[p + 'hx'], out[p + 'hy'], out[p + 'hz'] = grid.h
        out[p + 'origin'] = np.asarray(grid.origin, float)
        out[p + 'case'] = model.case
        out[p + 'frequency'] = float(freq)
        out[p + 'source'] = np.asarray(src, float)
        out[p + 'res_x'] = model.property_x
        if model.property_z is not None:
            out[p + 'res_z'] = model.property_z
        for k, v in kw.items():
            out[p + 'kw_' + k] = v
        for tol in tols:
            t0 = time.time()
            ef, info = rsolver.solve(model, sfield, return_info=True, sslsolver=False, verb=0, tol=tol, **kw)
            q = p + f"tol{tol:.0e}_"
            out[q + 'it_mg'] = info['it_mg']
            out[q + 'exit_message'] = info['exit_message']
            out[q + 'error_at_cycle'] = info['error_at_cycle']
            out[q + 'rel_error'] = info['rel_error']
            out[q + 'ref_error'] = info['ref_error']
            if tol == min(tols):
                out[p + 'efield'] = np.asarray(ef.field)
            print(f"  {name} tol={tol:.0e}: {info['exit_message']} it={info['it_mg']} "
                  f"rel={info['rel_error']:.3e}  ({time.time() - t0:.1f} s)", flush=True)

    h = np.ones(32) * 50.
    grid = emg3d.TensorMesh([h, h, h], origin=(-800, -800, -800))
    model = emg3d.Model(grid, property_x=1., mapping='Resistivity')
    run('uni32_F', grid, model, (0, 0, 0, 0, 0), 1.0, (1e-6, 1e-10), semicoarsening=False,
        linerelaxation=False, cycle='F')

    hx = widths(16, 8, 50, 1.2); hz = widths(16, 8, 25, 1.25)
    grid = emg3d.TensorMesh([hx, hx, hz], origin=(-hx.sum() / 2, -hx.sum() / 2, -hz[:20].sum()))
    zc = grid.cell_centers_z
    rh = np.where(zc > -200, 0.3, 1.0)
    rv = np.where(zc > -200, 0.3, 2.0)
    px = np.tile(rh[None, None, :], (32, 32, 1)).ravel('F')
    pz = np.tile(rv[None, None, :], (32, 32, 1)).ravel('F')
    model = emg3d.Model(grid, property_x=px, property_z=pz, mapping='Resistivity')
    run('marine32_W', grid, model, (0, 0, -150, 0, 0), 1.0, (1e-10,), semicoarsening=True,
        linerelaxation=True, cycle='W')

    out['meta_cases'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'solves32.npz'), **out)
    print('solves32.npz written')


def gradient():
    """Misfit and adjoint-state gradient of a small problem, computed with the reference's own
    functions in the order Simulation.gradient / _get_rfield use them (emg3d/simulations.py:
    1041-1090, 1235-1268; Survey / Simulation themselves need xarray, which is not installed):
    forward solve, linear receiver responses, residual source, back-propagated solve,
    real(bfield smu0 efield), maps.interp_edges_to_vol_averages, derivative chain of the
    resistivity mapping. Also raw input / output vectors of interp_edges_to_vol_averages."""
    from emg3d import maps as rmaps, fields as rfields
    rng = np.random.default_rng(21)
    out = dict(META)
    hx = widths(6, 3, 40, 1.3); hy = widths(4, 3, 50, 1.25); hz = widths(4, 2, 30, 1.4)
    grid = emg3d.TensorMesh([hx, hy, hz], origin=(-hx.sum() / 2, -hy.sum() / 2, -hz[:5].sum()))
    shape = grid.shape_cells
    rho_h = 10 ** rng.uniform(-0.3, 0.7, shape)
    rho_v = rho_h * rng.uniform(1.0, 2.5, shape)
    model = emg3d.Model(grid, property_x=rho_h, property_z=rho_v, mapping='Resistivity')
    freq, src = 0.8, (-45., 10., -20., 15., 5.)
    recs = np.array([[60., -20., -35., 0., 0.], [95., 30., -35., 90., 0.], [-110., 40., -50., 30., 10.],
                     [20., 75., -15., 0., 90.]])
    rec_t = tuple(recs[:, k] for k in range(5))
    solver_opts = dict(sslsolver=True, semicoarsening=True, linerelaxation=True, verb=0, tol=1e-9)
    sfield = emg3d.get_source_field(grid, src, freq)
    efield = rsolver.solve(model, sfield, **solver_opts)
    synthetic = rfields.get_receiver(efield, rec_t, method='linear')
    observed = synthetic * (1 + 0.15 * (rng.standard_normal(4) + 1j * rng.standard_normal(4)))
    observed[3] = np.nan                                   # a receiver without data
    weights = 1.0 / (0.05 * np.abs(observed)) ** 2
    weights[3] = 0.0
    residual = synthetic - observed
    have = ~np.isnan(residual)
    misfit = np.sum(weights[have] * (residual[have].conj() * residual[have])).real / 2
    rfield = emg3d.Field(grid, frequency=freq)
    strength = np.conj(residual * weights / -rfield.smu0)
    for i in range(4):
        if np.isnan(residual[i]):
            continue
        # the adjoint source of an electric point receiver (emg3d/electrodes.py:683)
        adj = emg3d.electrodes.TxElectricPoint(tuple(recs[i]), strength=strength[i])
        rfield.field += adj.get_field(grid=grid, frequency=freq).field
    bfield = rsolver.solve(model, rfield, **{**solver_opts, 'tol': 1e-9})
    gfield = emg3d.Field(grid, data=np.real(bfield.field * efield.smu0 * efield.field))
    grad = np.zeros((3, *shape), order='F')
    vol = grid.cell_volumes.reshape(shape, order='F')
    rmaps.interp_edges_to_vol_averages(ex=gfield.fx, ey=gfield.fy, ez=gfield.fz, volumes=vol,
                                       ox=grad[0], oy=grad[1], oz=grad[2])
    raw = grad.copy()
    mp = rmaps.MapResistivity()
    mp.derivative_chain(grad[2], model.property_z)
    grad[0] += grad[1]
    mp.derivative_chain(grad[0], model.property_x)
    out.update(hx=hx, hy=hy, hz=hz, origin=np.asarray(grid.origin, float), res_x=rho_h, res_z=rho_v,
               frequency=freq, source=np.asarray(src), receivers=recs, observed=observed, weights=weights,
               synthetic=synthetic, misfit=misfit, gradient=grad[[0, 2]], rfield=rfield.field,
               efield=efield.field, bfield=bfield.field, grad_cells_raw=raw)
    np.savez_compressed(os.path.join(OUT, 'gradient.npz'), **out)
    print(f'gradient.npz written: misfit {misfit:.6e}, |grad| {np.linalg.norm(grad[[0, 2]]):.4e}')


def receivers():
    """Magnetic field and receiver responses: inputs and the outputs of
    the reference's fields.get_magnetic_field / fields.get_receiver (cubic and linear), for a
    frequency-domain and a Laplace-domain field, receivers inside, in the outermost cell and
    outside of the grid."""
    rng = np.random.default_rng(2209)
    out = dict(META)
    out['meta_seed'] = 2209
    hx, hy, hz = widths(6, 3, 50., 1.3), widths(4, 3, 60., 1.2), widths(4, 2, 40., 1.25)
    origin = (-hx.sum() / 2, -hy.sum() / 2 + 13., -hz.sum() + 200.)
    grid = emg3d.TensorMesh([hx, hy, hz], origin)
    shp = grid.shape_cells
    model = emg3d.Model(grid, property_x=rng.uniform(0.5, 2, shp), mu_r=rng.uniform(0.8, 1.5, shp))
    out.update(hx=hx, hy=hy, hz=hz, origin=np.array(origin), mu_r=model.mu_r, property_x=model.property_x)
    n = 14
    x = rng.uniform(grid.nodes_x[2], grid.nodes_x[-3], n)
    y = rng.uniform(grid.nodes_y[2], grid.nodes_y[-3], n)
    z = rng.uniform(grid.nodes_z[2], grid.nodes_z[-3], n)
    # special places: exactly on a node / centre, in the outermost cells, outside the grid
    x[0], y[0], z[0] = grid.nodes_x[5], grid.nodes_y[4], grid.nodes_z[3]
    x[1], y[1], z[1] = grid.cell_centers_x[6], grid.cell_centers_y[5], grid.cell_centers_z[4]
    x[2] = 0.5 * (grid.nodes_x[0] + grid.nodes_x[1])
    y[3] = 0.5 * (grid.nodes_y[-1] + grid.nodes_y[-2])
    z[4] = grid.nodes_z[-1] + 10.
    x[5] = grid.nodes_x[0] - 1.
    x[6], y[6], z[6] = grid.nodes_x[1] + 1., grid.nodes_y[1] + 1., grid.nodes_z[1] + 1.
    x[7], y[7], z[7] = grid.nodes_x[-2] - 1., grid.nodes_y[-2] - 1., grid.nodes_z[-2] - 1.
    az = rng.uniform(-180, 180, n)
    el = rng.uniform(-90, 90, n)
    az[8], el[8] = 0., 0.
    az[9], el[9] = 90., 0.
    az[10], el[10] = 0., 90.
    out.update(rec_x=x, rec_y=y, rec_z=z, rec_azimuth=az, rec_elevation=el)
    for tag, freq, dtype in (('f', 1.2, complex), ('s', -2.0, float)):
        data = rng.standard_normal(grid.n_edges)
        if dtype is complex:
            data = data + 1j * rng.standard_normal(grid.n_edges)
        efield = emg3d.Field(grid, data=data, frequency=freq)
        hfield = emg3d.get_magnetic_field(model, efield)
        out[tag + '_frequency'] = freq
        out[tag + '_efield'] = efield.field
        out[tag + '_hfield'] = hfield.field
        for method in ('cubic', 'linear'):
            out[f'{tag}_e_{method}'] = np.asarray(efield.get_receiver((x, y, z, az, el), method=method))
            out[f'{tag}_h_{method}'] = np.asarray(hfield.get_receiver((x, y, z, az, el), method=method))
    np.savez_compressed(os.path.join(OUT, 'receivers.npz'), **out)
    print('receivers.npz written')


def gridding():
    """Model re-gridding: Model.interpolate_to_grid (volume averaging on a
    log10 scale, emg3d/models.py:322-366 -> maps.interp_volume_average, maps.py:555-664) from a
    model grid to computational grids that are finer, coarser, shifted and larger (nearest
    extrapolation), for the mappings with and without log scale."""
    rng = np.random.default_rng(1107)
    out = dict(META)
    out['meta_seed'] = 1107
    hx, hy, hz = widths(6, 2, 100., 1.3), widths(4, 2, 120., 1.2), widths(4, 1, 80., 1.5)
    origin = np.array([-hx.sum() / 2, -hy.sum() / 2, -hz.sum() + 100.])
    grid = emg3d.TensorMesh([hx, hy, hz], origin)
    shp = grid.shape_cells
    out.update(in_hx=hx, in_hy=hy, in_hz=hz, in_origin=origin)
    targets = {
        'fine': ([np.full(24, 40.), np.full(20, 45.), np.full(12, 50.)], origin + np.array([55., 30., 20.])),
        'coarse': ([widths(2, 2, 300., 1.4), widths(2, 1, 350., 1.3), np.array([200., 150., 100., 150.])],
                   origin - np.array([400., 300., 150.])),
        'same_nodes': ([hx[1:-1], hy, hz[:-1]], origin + np.array([hx[0], 0., 0.])),
    }
    for t, (h, o) in targets.items():
        out.update({f'{t}_hx': h[0], f'{t}_hy': h[1], f'{t}_hz': h[2], f'{t}_origin': np.asarray(o)})
    props = {'property_x': rng.uniform(0.3, 30., shp), 'property_z': rng.uniform(0.5, 50., shp),
             'mu_r': rng.uniform(0.9, 1.6, shp), 'epsilon_r': rng.uniform(1., 9., shp)}
    for mapping in ('Resistivity', 'Conductivity', 'LgConductivity'):
        p = dict(props)
        if mapping.startswith('L'):
            p['property_x'] = np.log10(p['property_x'])
            p['property_z'] = np.log10(p['property_z'])
        model = emg3d.Model(grid, mapping=mapping, **p)
        for k, v in p.items():
            out[f'{mapping}_in_{k}'] = v
        for t, (h, o) in targets.items():
            new = model.interpolate_to_grid(emg3d.TensorMesh(h, o))
            for k in p:
                out[f'{mapping}_{t}_{k}'] = getattr(new, k)
    np.savez_compressed(os.path.join(OUT, 'gridding.npz'), **out)
    print('gridding.npz written')


def sources():
    """Source vectors of the reference's get_source_field: magnetic dipoles (electric=False: the
    square loop of TxMagneticDipole) in the point and the two-electrode format, next to an electric
    dipole and a wire -- sparse (index, value) of the bare vector (frequency=None) and of the
    frequency- and Laplace-domain fields."""
    hx, hy, hz = widths(6, 3, 40., 1.3), widths(4, 3, 50., 1.25), widths(4, 2, 30., 1.4)
    grid = emg3d.TensorMesh([hx, hy, hz], (-hx.sum() / 2, -hy.sum() / 2, -hz[:5].sum()))
    out = dict(META)
    out['hx'], out['hy'], out['hz'], out['origin'] = hx, hy, hz, np.asarray(grid.origin, float)
    cases = [
        ('mag_point', (13., -7., 5., 37., -21.), dict(electric=False, length=1.0, strength=1.0)),
        ('mag_point_big', (-20., 25., -3., 0., 90.), dict(electric=False, length=900.0, strength=2.5)),
        ('mag_two', (-20., 25., -3., 17., 7., 31.), dict(electric=False, strength=1.0)),
        ('el_point', (13., -7., 5., 37., -21.), dict(strength=3.0)),
        ('el_wire', np.array([[-60., -40., -10.], [-10., -40., -10.], [-10., 30., 5.], [45., 30., 20.]]), dict(strength=1.5)),
    ]
    out['names'] = np.array([c[0] for c in cases])
    for name, src, kw in cases:
        out[f'{name}_source'] = np.asarray(src, dtype=float)
        out[f'{name}_strength'] = kw.get('strength', 1.0)
        out[f'{name}_length'] = kw.get('length', 1.0)
        out[f'{name}_electric'] = kw.get('electric', True)
        for tag, freq in (('vec', None), ('f', 0.9), ('s', -1.7)):
            sf = emg3d.get_source_field(grid, src, freq, **kw)
            nz = np.flatnonzero(sf.field)
            out[f'{name}_{tag}_index'], out[f'{name}_{tag}_value'] = nz, sf.field[nz]
    np.savez_compressed(os.path.join(OUT, 'sources.npz'), **out)
    print('sources.npz written:', [c[0] for c in cases])


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['regression', 'kernels', 'solves', 'receivers', 'gridding']
    if 'gridding' in which:
        gridding()
    if 'receivers' in which:
        receivers()
    if 'regression' in which:
        regression_small()
    if 'kernels' in which:
        kernels()
    if 'solves' in which:
        solves()
    if 'solves32' in which:
        solves32()
    if 'gradient' in which:
        gradient()
    if 'sources' in which:
        sources()
