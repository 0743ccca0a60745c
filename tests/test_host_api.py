"""Host-side logic of emg3d_amd (no GPU needed): data containers, parameter handling, cycle
control helpers, restriction weights, and that the C-ABI library loads and exports every
symbol declared in include/emg3d_amd.h."""
import os
import re

import numpy as np
import pytest

import emg3d_amd as emg3d
from emg3d_amd import _lib, core, solver
from oracle import core as ocore
from helpers import relerr, widths


def emg3d_line_records(n0, lines):
    """Block records of a direction (stencil.h: line_padded): middle m = (n0 // 2) // 4 * 4, m + 2 records up to
    the middle pair, the rest padded to a multiple of four."""
    m = (n0 // 2) // 4 * 4
    return (m + 2 + (n0 - 2 - m + 3) // 4 * 4) * lines


def test_library_loads_and_exports_all_declared_symbols():
    lib = _lib.lib()
    assert lib.emg3d_version() == 100
    header = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'emg3d_amd.h')).read()
    declared = set(re.findall(r'\b(emg3d_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.emg3d_device_count() >= 0
    # pure-arithmetic helpers of the ABI work without a device
    assert lib.emg3d_gs_scratch_bytes(0, 8, 8, 8, 1) == 0
    # records per line of the two-sided factorisation (stencil.h: line_padded):
    # n0 = 8 -> 4 top + 2 middle + 2 bottom padded to 4 = 10;  n0 = 5 (granule 2) -> 2 + 2 + 1 -> 2 = 6
    assert lib.emg3d_gs_scratch_bytes(1, 8, 6, 4, 1) == (5 * 10 * 3 * 2 + 80) * 16
    # T records (15 entries) + the N records of the wide form (16 entries) on levels small enough to hold them
    assert lib.emg3d_line_fac_bytes(1, 8, 6, 4, 1) == (15 + 16) * 10 * 5 * 3 * 16
    assert lib.emg3d_line_fac_bytes(1, 128, 6, 4, 1) == 15 * emg3d_line_records(128, 5 * 3) * 16      # lines too long
    assert lib.emg3d_line_fac_bytes(3, 300, 300, 64, 1) == 15 * 66 * 299 * 299 * 16                   # too many records
    assert lib.emg3d_line_lfac_bytes(3, 8, 6, 5) == 8 * 6 * 7 * 5 * 8
    assert lib.emg3d_point_fac_bytes(8, 6, 4, 1) == (8 * 7 * 5 + 9 * 6 * 5 + 9 * 7 * 4) * 16
    assert lib.emg3d_residual_ws_len(64, 8, 4) == 2 * 3 * 5


def test_no_cpu_fallback_without_gpu():
    if _lib.lib().emg3d_device_count() > 0:
        pytest.skip("GPU present")
    a = np.zeros(36)
    with pytest.raises(_lib.Emg3dAmdError, match="no HIP device"):
        core.solve(a, np.zeros(6))
    grid = emg3d.TensorMesh([np.ones(4)] * 3, (0, 0, 0))
    sfield = emg3d.get_source_field(grid, (2, 2, 2, 0, 0), 1.0)
    with pytest.raises(_lib.Emg3dAmdError):
        emg3d.solve(emg3d.Model(grid), sfield, sslsolver=False)


def test_field_layout_and_views():
    grid = emg3d.TensorMesh([np.ones(3), np.ones(4), np.ones(5)], (0, 0, 0))
    f = emg3d.Field(grid, frequency=2.0)
    assert f.field.dtype == np.complex128 and f.field.size == grid.n_edges
    assert f.fx.shape == (3, 5, 6) and f.fy.shape == (4, 4, 6) and f.fz.shape == (4, 5, 5)
    f.fx[1, 2, 3] = 7
    assert f.field[1 + 3 * (2 + 5 * 3)] == 7          # Fortran order, x fastest
    f.fz[3, 4, 4] = 9
    assert f.field[-1] == 9
    assert f.fx.flags.f_contiguous and not f.fx.flags.owndata
    assert np.isclose(f.sval, 2j * np.pi * 2.0) and np.isclose(f.smu0, f.sval * 1.25663706127e-06)
    lap = emg3d.Field(grid, frequency=-3.0)
    assert lap.field.dtype == np.float64 and lap.sval == 3.0 and lap.frequency == 3.0
    with pytest.raises(ValueError, match="must be f>0"):
        emg3d.Field(grid, frequency=0)
    g = f.copy()
    assert g == f
    g.fy[1, 1, 1] = 1e-3
    assert g != f


def test_source_field_and_volume_model_match_reference_fixtures(golden_solves):
    g = golden_solves
    for name in g['meta_cases']:
        p = str(name) + '_'
        grid = emg3d.TensorMesh([g[p + 'hx'], g[p + 'hy'], g[p + 'hz']], g[p + 'origin'])
        sf = emg3d.get_source_field(grid, g[p + 'source'], float(g[p + 'frequency']))
        assert relerr(sf.field, g[p + 'sfield']) < 1e-12
        kw = {k: g[p + 'res_' + k[-1]] for k in ('property_x', 'property_y', 'property_z')
              if p + 'res_' + k[-1] in g}
        model = emg3d.Model(grid, **kw)
        assert model.case == str(g[p + 'case'])
        vm = emg3d.models.VolumeModel(model, sf)
        for k in ('eta_x', 'eta_y', 'eta_z', 'zeta'):
            assert relerr(getattr(vm, k), g[p + k]) < 1e-15
        if model.case in ('isotropic', 'VTI'):
            assert vm.eta_y is vm.eta_x                 # aliasing rule
        if model.case in ('isotropic', 'HTI'):
            assert vm.eta_z is vm.eta_x
    with pytest.raises(ValueError, match="outside grid"):
        emg3d.get_source_field(grid, (1e9, 0, 0, 0, 0), 1.0)


def test_source_field_has_no_dense_buffer_until_somebody_looks():
    """A source from get_source_field is its few deposited entries until its dense buffer is asked for
    (Field._untouched): solve() then takes norm and upload from those entries; once `field` / `fx` ... has been handed
    out the dense buffer is the truth (it may have been modified), and the deposited entries are exactly what it held."""
    grid = emg3d.TensorMesh([np.full(12, 10.)] * 3, (-60., -60., -60.))
    sf = emg3d.get_source_field(grid, (1., 2., 3., 20., 10.), 1.0)
    assert sf._untouched and sf.dtype == np.complex128 and sf._dense is None
    idx, val = sf._sparse
    assert np.unique(idx).size == idx.size
    e = emg3d.Field(grid, frequency=1.0)
    assert e._untouched and e.dtype == np.complex128
    dense = sf.field                        # first look: zeros + the entries
    assert not sf._untouched and sf._lazy is None
    assert np.count_nonzero(dense) == idx.size and np.array_equal(dense[idx], val)
    assert np.linalg.norm(dense) == pytest.approx(np.linalg.norm(val), rel=1e-15)
    assert sf.fx.base is not None and sf.fx.shape == grid.shape_edges_x
    # deposits into a field that already has its buffer land in it
    g = emg3d.Field(grid, frequency=1.0)
    g.field[3] = 7.
    g._deposit(np.array([5]), np.array([2. + 0j]))
    assert g.field[3] == 7. and g.field[5] == 2. and not g._untouched
    # a Laplace-domain source is real
    assert emg3d.get_source_field(grid, (1., 2., 3., 20., 10.), -1.0).dtype == np.float64


def test_source_field_moment_and_finite_dipole():
    grid = emg3d.TensorMesh([widths(4, 2, 10, 1.5)] * 3, (-50, -50, -50))
    vec = emg3d.get_source_field(grid, (1.3, -2.2, 4.1, 30, 10), None)
    d = emg3d.fields._direction(30, 10)
    assert np.isclose(vec.fx.sum(), d[0]) and np.isclose(vec.fy.sum(), d[1])
    assert np.isclose(vec.fz.sum(), d[2])
    fin = emg3d.get_source_field(grid, (-20, 25, -3, -3, 7, 7), None, strength=2.0)
    assert np.isclose(fin.fx.sum(), 90.0) and fin.fy.sum() == 0 and fin.fz.sum() == 0


def test_restrict_weights_known_answer_and_vs_oracle():
    edges = np.array([0., 500, 1200, 2000, 3000])
    width = edges[1:] - edges[:-1]
    centr = edges[:-1] + width / 2
    c_edges = edges[::2]
    c_width = c_edges[1:] - c_edges[:-1]
    c_centr = c_edges[:-1] + c_width / 2
    wl, w0, wr = core.restrict_weights(edges, centr, width, c_edges, c_centr, c_width)
    assert np.allclose(wl, [350 / 250, 250 / 600, 400 / 900], rtol=1e-15)
    assert np.allclose(w0, 1.0)
    assert np.allclose(wr, [350 / 600, 500 / 900, 400 / 500], rtol=1e-15)
    grid = emg3d.TensorMesh([widths(2, 3, 200, 1.8), [1, 1], [1, 1]], (-1e5, 0, 0))
    cg = emg3d.TensorMesh([np.diff(grid.nodes_x[::2]), [1, 1], [1, 1]], grid.origin)
    a = core.restrict_weights(grid.nodes_x, grid.cell_centers_x, grid.h[0], cg.nodes_x,
                              cg.cell_centers_x, cg.h[0])
    b = ocore.restrict_weights(grid.nodes_x, grid.cell_centers_x, grid.h[0], cg.nodes_x,
                               cg.cell_centers_x, cg.h[0])
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-14, atol=0)


def test_mgparameters():
    """Rules of emg3d/solver.py:1202-1381 (cf. the reference's TestMGParameters)."""
    var = solver.MGParameters(verb=0, sslsolver=False, semicoarsening=False, linerelaxation=False,
                              shape_cells=(2 ** 3, 2 ** 5, 2 ** 4))
    assert list(var.clevel) == [4, 4, 3, 4] and var.cycmax == 2 and var.maxcycle == 1
    assert var.sc_dir == 0 and var.lr_dir == 0 and not var.sc_cycle and not var.lr_cycle
    var = solver.MGParameters(verb=0, sslsolver=True, semicoarsening=True, linerelaxation=True,
                              shape_cells=(20, 20, 20), cycle='V', maxit=33)
    assert var.sslsolver == 'bicgstab' and var.ssl_maxit == 33 and var.maxit == 3
    assert var.cycmax == 1 and var.sc_dir == 1 and var.lr_dir == 4
    assert [next(var.sc_cycle) for _ in range(4)] == [2, 3, 1, 2]
    assert [next(var.lr_cycle) for _ in range(4)] == [5, 6, 4, 5]
    assert list(var.clevel) == [2, 2, 2, 2]        # 20 -> 10 -> 5 (odd)
    assert 'not optimal' in var._repr_clevel['message']
    var = solver.MGParameters(verb=0, sslsolver=False, semicoarsening=1213, linerelaxation=7,
                              shape_cells=(16, 16, 16), clevel=2)
    assert var.sc_dir == 1 and var.maxcycle == 4 and list(var.clevel) == [2, 2, 2, 2]
    assert [next(var.sc_cycle) for _ in range(4)] == [2, 1, 3, 1]
    for bad in (dict(semicoarsening=5), dict(linerelaxation=9), dict(sslsolver='cg'),
                dict(cycle='G'), dict(cycle=None, sslsolver=False)):
        kw = dict(verb=0, sslsolver=False, semicoarsening=False, linerelaxation=False,
                  shape_cells=(8, 8, 8))
        kw.update(bad)
        with pytest.raises(ValueError):
            solver.MGParameters(**kw)
    with pytest.raises(ValueError, match="at least two"):
        solver.MGParameters(verb=0, sslsolver=False, semicoarsening=False, linerelaxation=False,
                            shape_cells=(1, 8, 8))
    assert 'MG-cycle' in repr(var) and 'Coarsest grid' in repr(var)


def test_current_sc_and_lr_dir():
    """emg3d/solver.py:1482-1588 (cf. reference tests test_current_sc_dir/_lr_dir)."""
    mk = lambda n: emg3d.TensorMesh([np.ones(n[0]), np.ones(n[1]), np.ones(n[2])], (0, 0, 0))
    g = mk((4, 2, 2))
    assert [solver._current_sc_dir(s, g) for s in range(4)] == [4, 6, 4, 4]
    g = mk((4, 4, 4))
    assert [solver._current_sc_dir(s, g) for s in range(4)] == [0, 1, 2, 3]
    g = mk((4, 4, 2))
    assert [solver._current_sc_dir(s, g) for s in range(4)] == [3, 5, 4, 3]
    g = mk((3, 4, 8))
    assert [solver._current_sc_dir(s, g) for s in range(4)] == [1, 1, 6, 5]
    g = mk((4, 4, 4))
    assert [solver._current_lr_dir(c, g) for c in range(8)] == list(range(8))
    g = mk((2, 4, 4))
    assert [solver._current_lr_dir(c, g) for c in range(8)] == [0, 0, 2, 3, 4, 3, 2, 4]
    g = mk((4, 2, 4))
    assert [solver._current_lr_dir(c, g) for c in range(8)] == [0, 1, 0, 3, 3, 5, 1, 5]
    g = mk((4, 4, 2))
    assert [solver._current_lr_dir(c, g) for c in range(8)] == [0, 1, 2, 0, 2, 1, 6, 6]
    g = mk((2, 2, 4))
    assert solver._current_lr_dir(7, g) == 3 and solver._current_lr_dir(6, g) == 0


def test_terminate():
    """emg3d/solver.py:1591-1664 (cf. reference test_terminate)."""
    class V:
        tol, l2_refe, sslsolver, maxit, verb, exit_message = 1e-3, 1e-3, False, 5, 0, ''

        def cprint(self, *a, **k):
            pass
    v = V()
    assert solver._terminate(v, 1e-7, 1, 1) and v.exit_message == 'CONVERGED'
    assert solver._terminate(v, np.inf, 1, 1) and v.exit_message == 'DIVERGED'
    assert solver._terminate(v, 1e-1, 1, 1) and v.exit_message == 'DIVERGED'
    assert solver._terminate(v, 1e-5, 1e-6, 3) and v.exit_message == 'STAGNATED'
    assert not solver._terminate(v, 1e-5, 1e-6, 2)
    assert solver._terminate(v, 1e-5, 1e-4, 5) and v.exit_message.startswith('MAX. ITERATION')
    v.sslsolver = True
    with pytest.raises(solver._ConvergenceError):
        solver._terminate(v, np.nan, 1, 1)


def test_regular_grid_prolongator_vs_scipy():
    import scipy.interpolate as si
    cx, cy = np.array([0., 1, 3, 7]), np.array([-2., 0, 5])
    x, y = np.array([0., .5, 1, 2, 3, 5, 7]), np.array([-2., -1, 0, 2.5, 5])
    vals = np.arange(12.).reshape(4, 3) ** 1.5
    fn = solver.RegularGridProlongator(cx, cy, x, y)
    ref = si.RegularGridInterpolator((cx, cy), vals, bounds_error=False, fill_value=None)
    X, Y = np.meshgrid(x, y, indexing='ij')
    assert np.allclose(fn(vals).reshape(7, 5, order='F'), ref((X, Y)), rtol=1e-14)


def test_get_restriction_weights_shapes():
    grid = emg3d.TensorMesh([widths(2, 1, 10, 1.5), widths(4, 2, 10, 1.2), np.ones(2)], (0, 0, 0))
    cg = emg3d.TensorMesh([np.diff(grid.nodes_x[::2]), grid.h[1], grid.h[2]], grid.origin)
    wx, wy, wz = solver._get_restriction_weights(grid, cg, 4)
    assert wx[0].size == cg.shape_nodes[0]
    assert wy[0].size == grid.shape_nodes[1] and np.all(wy[0] == 0) and np.all(wy[1] == 1)
    assert wz[1].size == grid.shape_nodes[2]


def test_coarse_schedule_matches_recursive_definition():
    """The flat visiting order of _cycle.coarse_schedule against the recursive definition of the
    cycle (emg3d/solver.py:512-649: a level loops cycmax times, hands `cycmax - done` to the next
    coarser level, the coarsest level is smoothed once) written out here as a plain recursion."""
    from emg3d_amd._cycle import coarse_schedule, SMOOTH, DOWN, UP

    def recurse(cycle, cycmax_all, depth, level, budget, out):
        visits = 1 if level == depth else (cycmax_all if budget == 0 or cycle != 'F' else budget)
        for done in range(visits):
            if level == depth:
                out.append((SMOOTH, level, 1, done, visits))
            else:
                out.append((SMOOTH, level, 2, done, visits))
                out.append((DOWN, level))
                recurse(cycle, cycmax_all, depth, level + 1, visits - done, out)
                out.append((UP, level))
                out.append((SMOOTH, level, 3, done, visits))

    for cycle, cycmax in (('V', 1), ('W', 2), ('F', 2)):
        for depth in (1, 2, 3, 6):
            for budget in (1, 2):
                want = []
                recurse(cycle, cycmax, depth, 1, budget, want)
                got = [(s[0], s[1]) + ((s[2], s[4], s[5]) if s[0] == SMOOTH else ())
                       for s in coarse_schedule(cycle, cycmax, depth, 1, budget, 2, 1, 3) if s[0] in (SMOOTH, DOWN, UP)]
                assert got == want, (cycle, depth, budget)
    # number of coarsest-level visits: V 1, W 2^(depth-1), F depth (one more per level)
    count = lambda c, m, d: sum(1 for s in coarse_schedule(c, m, d, 1, m, 2, 1, 2) if s[0] == SMOOTH and s[1] == d)
    assert [count('V', 1, d) for d in (1, 2, 3, 4)] == [1, 1, 1, 1]
    assert [count('W', 2, d) for d in (1, 2, 3, 4)] == [1, 2, 4, 8]
    assert [count('F', 2, d) for d in (1, 2, 3, 4)] == [1, 2, 3, 4]


def test_direction_schedule_and_stop_rules():
    from emg3d_amd._params import _DirectionSchedule, coarsening_depths
    from emg3d_amd._cycle import stop_reason
    s = _DirectionSchedule(True, (1, 2, 3), 4, "{}")
    assert s and len(s) == 3 and s.first() == 1 and [next(s) for _ in range(4)] == [2, 3, 1, 2]
    s = _DirectionSchedule(2, (1, 2, 3), 4, "{}")
    assert not s and s.first() == 2 and len(s) == 1
    s = _DirectionSchedule(1213, (1, 2, 3), 4, "{}")
    assert s and s.first() == 1 and [next(s) for _ in range(5)] == [2, 1, 3, 1, 2]
    for bad in (4, 15, 'x'):
        with pytest.raises(ValueError):
            _DirectionSchedule(bad, (1, 2, 3), 4, "bad {}")
    d, c = coarsening_depths((48, 20, 2), -1)
    assert list(d) == [4, 2, 0] and c == (3, 5, 2)
    d, c = coarsening_depths((48, 20, 2), 1)
    assert list(d) == [1, 1, 0] and c == (24, 10, 2)

    class V:
        tol, l2_refe, maxit = 1e-3, 1.0, 4
    assert stop_reason(V, 1e-4, 1, 1) == ("CONVERGED", False)
    assert stop_reason(V, 11.0, 1, 1) == ("DIVERGED", True)
    assert stop_reason(V, 0.5, 0.4, 3) == ("STAGNATED", True)
    assert stop_reason(V, 0.5, 0.6, 4)[0].startswith("MAX.") and stop_reason(V, 0.5, 0.6, 3) is None


def test_residual_source_field_vs_reference(golden_gradient):
    """gradient.residual_source_field (Simulation._get_rfield, emg3d/simulations.py:1235-1268): the
    residual source the reference's own functions assemble for the fixture's receivers."""
    from emg3d_amd import gradient
    g = golden_gradient
    grid = emg3d.TensorMesh([g['hx'], g['hy'], g['hz']], g['origin'])
    residual = g['synthetic'] - g['observed']
    rf = gradient.residual_source_field(grid, float(g['frequency']), g['receivers'], residual, g['weights'])
    assert relerr(rf.field, g['rfield']) < 1e-12
    idx, val = rf._sparse
    dense = np.zeros_like(rf.field)
    dense[idx] = val
    assert np.array_equal(dense, rf.field)


def test_smoother_omega_is_validated():
    from emg3d_amd import solver
    assert solver._check_omega(1) == 1.0 and solver._check_omega(1.25) == 1.25
    for bad in (0, 2, -0.5, 2.5):
        with pytest.raises(ValueError, match="smoother_omega"):
            solver._check_omega(bad)


def test_source_vectors_incl_magnetic_dipoles_vs_reference(golden_sources):
    """get_source_field against the reference's own source vectors (tools/make_golden.py sources):
    magnetic dipoles (``electric=False``: the square loop of TxMagneticDipole, point and two-electrode
    format), an electric dipole and a wire -- bare vector, frequency and Laplace domain."""
    g = golden_sources
    grid = emg3d.TensorMesh([g['hx'], g['hy'], g['hz']], g['origin'])
    for name in g['names']:
        src = g[f'{name}_source']
        kw = dict(strength=float(g[f'{name}_strength']))
        if src.size == 5:
            kw['length'] = float(g[f'{name}_length'])
        if not bool(g[f'{name}_electric']):
            kw['electric'] = False
        for tag, freq in (('vec', None), ('f', 0.9), ('s', -1.7)):
            sf = emg3d.get_source_field(grid, src, freq, **kw)
            want = np.zeros(sf.field.size, dtype=g[f'{name}_{tag}_value'].dtype)
            want[g[f'{name}_{tag}_index']] = g[f'{name}_{tag}_value']
            assert sf.field.dtype == want.dtype
            scale = np.abs(want).max()
            assert np.abs(sf.field - want).max() < 1e-12 * scale, (name, tag)


def test_residual_form_auto_rule():
    """`solve(residual_form='auto')`: the finest level goes into residual form where eps / (|s| mu0 sigma h^2)
    of the worst cell is not well below the tolerance (air layers, very low frequencies, tight tolerances)."""
    from emg3d_amd import solver
    h = np.full(8, 50.)
    grid = emg3d.TensorMesh([h, h, h], (0., 0., 0.))
    sf = emg3d.get_source_field(grid, (200., 200., 200., 0., 0.), 1.0)
    benign = emg3d.Model(grid, property_x=np.full(grid.shape_cells, 1.0))
    rho = np.full(grid.shape_cells, 1.0)
    rho[:, :, 6:] = 1e8
    air = emg3d.Model(grid, property_x=rho)

    def rule(model, choice='auto', **kw):
        var = solver.MGParameters(0, kw.pop('sslsolver', False), True, True, model.shape, **kw)
        return solver._residual_form(choice, var, model, sf)
    assert rule(benign, tol=1e-6) is False and rule(air, tol=1e-6) is True
    assert rule(benign, tol=1e-12) is True
    assert rule(air, tol=1e-6, sslsolver=True) is False          # a preconditioner is in residual form anyway
    assert rule(benign, True, tol=1e-6) is True and rule(air, False, tol=1e-6) is False
    with pytest.raises(ValueError):
        rule(benign, 'yes', tol=1e-6)


def test_line_compact_auto_rule():
    """solver.block_condition / COMPACT_COND_MAX: 'auto' keeps the streamed line records in single precision where the
    block condition estimate 1 / (|s| mu0 sigma_min h_min^2) is at most 3e4 -- the bench workloads are (configs 2, 3,
    5: 4e2, 1.6e4, 2e4), an air layer of 1e8 Ohm m is not (2e10), nor is a model with a zero conductivity -- from a
    Model (property arrays + mapping) and from a bare eta / zeta holder alike."""
    from emg3d_amd import solver, models
    import bench
    for name, lo, hi in (('marine64', 1e2, 1e3), ('triaxial64', 5e3, 5e4), ('salt96', 1e2, 1e5)):
        wl = bench.workload(name)
        grid = emg3d.TensorMesh(wl['h'], wl['origin'])
        model = emg3d.Model(grid, **wl['res'])
        sf = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        vm = models.VolumeModel(model, sf)
        c = solver.block_condition(vm)
        assert lo < c < hi and c <= solver.COMPACT_COND_MAX, (name, c)
        assert sf._untouched            # (the rule never looks at the source's dense buffer)
    # air: 1e8 Ohm m on top
    wl = bench.workload('marine32')
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    res = {k: np.array(v, dtype=float) for k, v in wl['res'].items()}
    for v in res.values():
        v[:, :, -3:] = 1e8
    sf = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    vm = models.VolumeModel(emg3d.Model(grid, **res), sf)
    assert solver.block_condition(vm) > 1e9
    # the same number from eta / zeta alone
    class Holder:
        pass
    h = Holder()
    h.grid, h._sval = vm.grid, vm._sval
    h.eta_x, h.eta_y, h.eta_z, h.zeta = vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta
    assert solver.block_condition(h) == pytest.approx(solver.block_condition(vm), rel=1e-12)
    # a cell that does not conduct at all: beyond any bound, never compact
    res0 = {k: np.array(v, dtype=float) for k, v in wl['res'].items()}
    for v in res0.values():
        v[0, 0, 0] = 1e300
    assert solver.block_condition(models.VolumeModel(emg3d.Model(grid, **res0), sf)) > 1e200


def test_option_table_and_fingerprint():
    """The run-time options of the library: enumerable, settable by name, unknown names refused; the
    fingerprint that keys captured graphs follows a change."""
    lib = _lib.lib()
    names = [lib.emg3d_option_name(i).decode() for i in range(lib.emg3d_option_count())]
    assert lib.emg3d_option_name(len(names)) is None and lib.emg3d_option_name(-1) is None
    for want in ('point_tile_min', 'skip_repeat', 'line_stream', 'line_lds', 'residual_zb'):
        assert want in names
    assert len(set(names)) == len(names)
    before = _lib.options_fingerprint()
    assert len(before) == len(names)
    old = lib.emg3d_get_option(b'line_stream')
    try:
        assert lib.emg3d_set_option(b'line_stream', 1 - old) == 0
        assert lib.emg3d_get_option(b'line_stream') == 1 - old
        assert _lib.options_fingerprint() != before
    finally:
        lib.emg3d_set_option(b'line_stream', old)
    assert _lib.options_fingerprint() == before
    assert lib.emg3d_set_option(b'no_such_option', 1) != 0 and lib.emg3d_get_option(b'no_such_option') == -1
    assert lib.emg3d_set_option(b'line_lpw', 5) != 0                      # only 0, 4, 8, 16, 32
    assert lib.emg3d_set_option(b'residual_zb', 0) == 0 and lib.emg3d_get_option(b'residual_zb') == 1
    lib.emg3d_set_option(b'residual_zb', 8)
    # options that select the sweep order take their defined values only; the wrong-results debug switch
    # needs the environment's consent
    assert lib.emg3d_set_option(b'line_order', 3) != 0 and lib.emg3d_set_option(b'point_order', 2) != 0
    assert lib.emg3d_get_option(b'line_order') == 1 and lib.emg3d_get_option(b'point_order') == 1
    had = os.environ.pop('EMG3D_AMD_ALLOW_DEBUG', None)
    try:
        assert lib.emg3d_set_option(b'line_debug', 1) != 0 and lib.emg3d_get_option(b'line_debug') == 0
        assert lib.emg3d_set_option(b'line_debug', 0) == 0
    finally:
        if had is not None:
            os.environ['EMG3D_AMD_ALLOW_DEBUG'] = had


def test_bench_helpers_without_a_gpu():
    """bench.py: ranks > 0 of a multi-GPU run never build the model (`with_model=False`), the level-0
    kernel label follows the line length, and the PMC figure names where it comes from."""
    import bench
    for name in ('marine64', 'triaxial64', 'salt96', 'uniform32'):
        lean, full = bench.workload(name, with_model=False), bench.workload(name)
        assert lean['res'] is None and full['res'] is not None
        assert all(np.array_equal(a, b) for a, b in zip(lean['h'], full['h'])) and lean['source'] == full['source']
    # the label is the library's own dispatch decision (emg3d_line_kernel_name), per direction and batch
    for lr in (1, 2, 3):
        assert bench.line_kernel_name(lr, (256, 256, 256)) == 'k_line_stream'
        assert bench.line_kernel_name(lr, (128, 128, 128)) == 'k_line_stream'
        assert bench.line_kernel_name(lr, (64, 64, 64)) == 'k_line_colour'
        assert bench.line_kernel_name(lr, (256, 256, 256), batch=4) == 'k_line_stream'
        assert bench.line_kernel_name(lr, (64, 64, 64), batch=4) == 'k_line_colour'
    assert bench.line_kernel_name(1, (384, 256, 256)) == 'k_line_stream' and bench.line_kernel_name(2, (256, 32, 32)) == 'k_line_wide' and bench.line_kernel_name(2, (256, 64, 64)) == 'k_line_colour'
    assert bench.line_kernel_name(2, (128, 128, 128), batch=8) == 'k_line_stream'
    lib = _lib.lib()
    old = lib.emg3d_get_option(b'line_lpw')
    try:                        # 32 lines per workgroup: the streamed kernels (16 lines) must not be chosen
        lib.emg3d_set_option(b'line_lpw', 32)
        assert bench.line_kernel_name(2, (256, 256, 256)) == 'k_line_colour'
    finally:
        lib.emg3d_set_option(b'line_lpw', old)
    for bad in (3, 6, 36, -4):
        assert lib.emg3d_set_option(b'line_stream_r', bad) != 0
    assert lib.emg3d_set_option(b'line_stream_r', 8) == 0 and lib.emg3d_set_option(b'line_stream_r', 0) == 0
    traffic, src = bench.pmc_traffic('triaxial256', 'k_gs_line<y>')
    assert traffic > 3e9 and src['file'].startswith('profiles/r') and src['measured_in_this_run'] is False
    # the summary names the sources it was measured on; `stale` says whether they are the ones in the tree
    assert src['stale'] == (src.get('csrc_sha16') != bench.csrc_sha16())
    assert bench.pmc_traffic('no_such_workload', 'k') == (None, None)


def test_model_drops_a_device_snapshot_when_a_property_is_replaced():
    """parallel.broadcast_model leaves HBM copies of the property arrays on the model it RETURNS
    (`_device_props`); replacing a property array must drop that array's copy."""
    grid = emg3d.TensorMesh([np.ones(4), np.ones(4), np.ones(4)], (0, 0, 0))
    model = emg3d.Model(grid, 2.0, property_z=3.0)
    model.__dict__['_device_props'] = {'property_x': 'X', 'property_z': 'Z'}
    model.property_x = np.full(grid.shape_cells, 5.0)
    assert model._device_props == {'property_z': 'Z'}
    model.mapping = model.mapping                      # other attributes do not touch it
    assert model._device_props == {'property_z': 'Z'}


@pytest.mark.parametrize('auto', [True, False])
def test_stall_switch_control_flow_without_a_gpu(auto, monkeypatch):
    """_cycle.run_cycles: a direct form that stagnates above the tolerance goes on in residual form when the
    caller left residual_form at 'auto' (and only then); scripted residual norms, no device."""
    from emg3d_amd import _cycle
    from emg3d_amd._params import MGParameters
    script = [1.0, 1e-2, 1e-4, 1e-6, 1.1e-6, 1e-8, 1e-10, 1e-12]

    class Top:
        batch = 1
        _b_valid = True

        def __init__(self):
            self.norms, self.calls = iter(script), []

        def residual(self, store=True, norm=False):
            self.calls.append(('residual', store, norm))
            return next(self.norms) if norm else None

        def reserve_residual_equation(self):          # (the two extra buffers, reserved before the switch)
            self.calls.append('reserve')

        def to_residual_equation(self):
            self.calls.append('to')

        def from_residual_equation(self):
            self.calls.append('from')

    monkeypatch.setattr(_cycle, '_one_cycle', lambda top, var, it, loud: top.calls.append('cycle'))
    var = MGParameters(verb=0, sslsolver=False, semicoarsening=False, linerelaxation=False, shape_cells=(8, 8, 8),
                       tol=1e-9, maxit=20)
    var.l2_refe = 1.0
    var.error_at_cycle[0] = 1.0
    var.residual_form, var.residual_form_auto = False, auto
    top = Top()
    _cycle.run_cycles(top, var)
    if not auto:
        assert var.exit_message == 'STAGNATED' and var.it == 4 and 'to' not in top.calls
        return
    assert var.exit_message == 'CONVERGED' and var.it == 6 and var.l2 == 1e-10
    assert var.residual_form is True and var.residual_form_switched is True
    # four cycles in direct form (norm only), the stored residual at the switch, then two cycles in residual form
    assert top.calls.count('cycle') == 6 and top.calls.count('to') == top.calls.count('from') == 2
    first_to = top.calls.index('to')
    assert top.calls[first_to - 1] == ('residual', True, False) and top.calls[:first_to].count('cycle') == 4
    assert top.calls.index('reserve') < first_to
    assert top._b_valid is False


def test_fast_div_is_exact():
    """launch.h: fast_div (single-precision quotient of (a + 1/2) / b, what the small line kernels use instead of the
    integer-division sequence) equals a // b on the whole range it is used on (a < 2^19, every divisor up to 4096,
    and larger divisors at the quotient boundaries)."""
    a = np.arange(1 << 19, dtype=np.int64)
    for b in list(range(1, 4097, 7)) + [1, 2, 3, 5, 8, 63, 64, 65, 127, 191, 255, 256, 4095, 4096]:
        q = ((a.astype(np.float32) + np.float32(0.5)) / np.float32(b)).astype(np.int64)
        assert np.array_equal(q, a // b), b
    for b in (10007, 65537, 262147, (1 << 19) - 1):
        aa = np.unique(np.clip(np.concatenate([np.arange(0, 1 << 19, b) + d for d in (-1, 0, 1)]), 0, (1 << 19) - 1))
        q = ((aa.astype(np.float32) + np.float32(0.5)) / np.float32(b)).astype(np.int64)
        assert np.array_equal(q, aa // b), b
