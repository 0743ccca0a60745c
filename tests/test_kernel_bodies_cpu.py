"""Kernel BODIES (emg3d_amd/csrc/stencil.h) on the CPU: tests/emu walks the HIP launch grids
on host arrays and calls the same per-thread functions the GPU runs. Compared with

* the oracle in its four-colour order (same update order -> tight tolerance),
* the golden vectors generated from the reference (order-independent kernels).

The GPU parity tests (tests/test_gpu_*.py, -m gpu) repeat the comparisons through the C ABI.
"""
import numpy as np
import pytest

from oracle import core as ocore
from oracle import mg_ref
from emu import emu
from helpers import relerr

LR = {'gauss_seidel': 0, 'gauss_seidel_x': 1, 'gauss_seidel_y': 2, 'gauss_seidel_z': 3}


def _case(g, name):
    p = name + '_'
    grid = mg_ref.Grid([g[p + 'hx'], g[p + 'hy'], g[p + 'hz']], g[p + 'origin'])
    ex = np.asfortranarray(g[p + 'eta_x'])
    case = str(g[p + 'case'])
    ey = np.asfortranarray(g[p + 'eta_y']) if case in ('HTI', 'triaxial') else ex
    ez = np.asfortranarray(g[p + 'eta_z']) if case in ('VTI', 'triaxial') else ex
    vm = mg_ref.VModel(grid, ex, ey, ez, np.asfortranarray(g[p + 'zeta']), case)
    return grid, vm


def prolong_tables(cgrid, grid):
    il, w = [], []
    for cn, n in ((cgrid.nodes_x, grid.nodes_x), (cgrid.nodes_y, grid.nodes_y),
                  (cgrid.nodes_z, grid.nodes_z)):
        i, ww = mg_ref._interp_1d(cn, n)
        il.append(np.ascontiguousarray(i, dtype=np.int32))
        w.append(np.ascontiguousarray(ww, dtype=np.float64))
    return il, w


def test_smoothers_match_oracle_four_colour(golden_kernels):
    g = golden_kernels
    for name in g['meta_cases']:
        name = str(name)
        p = name + '_'
        grid, vm = _case(g, name)
        s = mg_ref.Field(grid, g[p + 'gs_s'].copy())
        for fn, lr in LR.items():
            for nu in (1, 2, 3):
                a = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
                b = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
                getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y,
                                   vm.eta_z, vm.zeta, *grid.h, nu, order=1)
                emu.gauss_seidel(b, s, vm, lr, nu)
                assert relerr(b.field, a.field) < 2e-12, (name, fn, nu)


def test_compact_line_walk_is_a_small_perturbation_of_the_fp64_walk(golden_kernels):
    """The CPU walk of the line solve with COMPACT records (tests/emu: T records rounded to single precision by the
    set-up, w records rounded as the forward pass stores them, all arithmetic fp64 -- the definition the HIP kernel
    k_line_stream<.., COMPACT> is checked against on the GPU) against the fp64 walk: different, and within eps32 x cond
    of the blocks; the fp64 walk itself is untouched by the option."""
    g = golden_kernels
    lib = emu.lib()
    seen = 0
    for name in g['meta_cases']:
        name = str(name)
        p = name + '_'
        grid, vm = _case(g, name)
        s = mg_ref.Field(grid, g[p + 'gs_s'].copy())
        for fn, lr in LR.items():
            if lr == 0:
                continue
            a = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            b = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            c = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            emu.gauss_seidel(a, s, vm, lr, 2)
            lib.emu_set_line_compact(1)
            try:
                emu.gauss_seidel(b, s, vm, lr, 2)
            finally:
                lib.emu_set_line_compact(0)
            emu.gauss_seidel(c, s, vm, lr, 2)
            assert np.array_equal(a.field, c.field)
            d = relerr(b.field, a.field)
            assert 1e-10 < d < 1e-3, (name, fn, d)
            seen += 1
    assert seen >= 3


def test_residual_matches_reference_vectors(golden_kernels):
    g = golden_kernels
    for name in g['meta_cases']:
        name = str(name)
        p = name + '_'
        grid, vm = _case(g, name)
        e = mg_ref.Field(grid, g[p + 'amat_e'].copy())
        s = mg_ref.Field(grid, g[p + 'amat_r_in'].copy())
        r = mg_ref.Field(grid, dtype=e.field.dtype)
        ss = emu.residual(e, s, vm, r)
        assert relerr(r.field, g[p + 'amat_r_out']) < 1e-13
        assert abs(ss / np.linalg.norm(g[p + 'amat_r_out']) ** 2 - 1) < 1e-13
        # in-place form (r aliases s), as core.amat_x is used by the host-flavour ABI
        s2 = mg_ref.Field(grid, g[p + 'amat_r_in'].copy())
        emu.residual(e, s2, vm, s2)
        assert relerr(s2.field, g[p + 'amat_r_out']) < 1e-13
        # norm-only form
        e = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
        s = mg_ref.Field(grid, g[p + 'gs_s'].copy())
        assert abs(np.sqrt(emu.residual(e, s, vm)) / g[p + 'residual_norm'] - 1) < 1e-13


def test_restrict_prolong_param_match_reference_vectors(golden_kernels):
    g = golden_kernels
    for name in g['meta_cases']:
        name = str(name)
        p = name + '_'
        grid, vm = _case(g, name)
        res = mg_ref.Field(grid, g[p + 'restrict_res'].copy())
        for sc_dir in range(7):
            q = p + f'sc{sc_dir}_'
            if q + 'csfield' not in g:
                continue
            rx = 1 if sc_dir in (1, 5, 6) else 2
            ry = 1 if sc_dir in (2, 4, 6) else 2
            rz = 1 if sc_dir in (3, 4, 5) else 2
            cgrid = mg_ref.Grid([np.diff(grid.nodes_x[::rx]), np.diff(grid.nodes_y[::ry]),
                                 np.diff(grid.nodes_z[::rz])], grid.origin)
            w9 = []
            for d, f in zip('xyz', (rx, ry, rz)):
                w = g[q + f'w{d}']
                w9 += [np.ascontiguousarray(w[i]) if f == 2 else None for i in range(3)]
            c = mg_ref.Field(cgrid, dtype=res.field.dtype)
            emu.restrict(c, res, w9, grid.shape_cells, sc_dir)
            assert relerr(c.field, g[q + 'csfield']) < 1e-14, (name, sc_dir)
            for k in ('eta_x', 'eta_y', 'eta_z', 'zeta'):
                assert relerr(emu.restrict_param(getattr(vm, k), sc_dir), g[q + 'c' + k]) < 1e-15
            ce = mg_ref.Field(cgrid, g[q + 'prol_c'].copy())
            fine = mg_ref.Field(grid, g[q + 'prol_f_in'].copy())
            il, w = prolong_tables(cgrid, grid)
            emu.prolong(fine, ce, il, w, grid.shape_cells, sc_dir)
            assert relerr(fine.field, g[q + 'prol_f_out']) < 1e-14, (name, sc_dir)


def test_band_solve_and_blocks_to_amat():
    rng = np.random.default_rng(11)
    for dtype in (np.float64, np.complex128):
        for n in (6, 11, 36):
            amat = rng.standard_normal(6 * n).astype(dtype)
            if dtype == np.complex128:
                amat = amat + 1j * rng.standard_normal(6 * n)
            amat[::6] += 25
            b = rng.standard_normal(n).astype(dtype)
            a1, b1, a2, b2 = amat.copy(), b.copy(), amat.copy(), b.copy()
            ocore.solve(a1, b1)
            emu.solve(a2, b2)
            assert relerr(b2, b1) < 1e-13 and relerr(a2, a1) < 1e-13
        for nc in (2, 3, 6):
            n = 5 * nc - 4
            a1, a2 = np.zeros(6 * n, dtype), np.zeros(6 * n, dtype)
            b1, b2 = np.zeros(n, dtype), np.zeros(n, dtype)
            for im in range(nc):
                mid = rng.standard_normal(25).astype(dtype)
                left = rng.standard_normal(25)
                rhs = rng.standard_normal(5).astype(dtype)
                ocore.blocks_to_amat(a1, b1, mid, left, rhs, im, nc)
                emu.blocks_to_amat(a2, b2, mid, left, rhs, im, nc)
            assert np.array_equal(a1, a2) and np.array_equal(b1, b2)


@pytest.mark.parametrize('shape', [(16, 8, 12), (5, 7, 9), (3, 3, 3), (2, 2, 2), (2, 9, 4)])
def test_smoothers_odd_and_tiny_grids(shape):
    """Ragged sizes (odd cell counts, 2- and 3-cell directions): every node/line must be
    visited exactly once per sweep -- compared with the oracle's four-colour order."""
    rng = np.random.default_rng(5)
    nx, ny, nz = shape
    grid = mg_ref.Grid([rng.uniform(10, 30, nx), rng.uniform(10, 30, ny), rng.uniform(10, 30, nz)],
                       (0, 0, 0))
    vm = mg_ref.volume_model(grid, 1.3, 10 ** rng.uniform(-1, 1, shape),
                             10 ** rng.uniform(-1, 1, shape), 10 ** rng.uniform(-1, 1, shape))
    s = mg_ref.Field(grid)
    s.field[:] = rng.standard_normal(s.field.size) + 1j * rng.standard_normal(s.field.size)
    e0 = mg_ref.Field(grid)
    e0.field[:] = rng.standard_normal(s.field.size) + 1j * rng.standard_normal(s.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    for fn, lr in LR.items():
        a, b = e0.copy(), e0.copy()
        getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                           vm.zeta, *grid.h, 2, order=1)
        emu.gauss_seidel(b, s, vm, lr, 2)
        assert relerr(b.field, a.field) < 2e-12, (shape, fn)


@pytest.mark.parametrize('dtype', [complex, float])
def test_oracle_threads_change_nothing(dtype):
    """oracle_set_threads: the four-colour orders (point 1 / 2, lines 1) with 1 and 6 threads, bit for bit."""
    rng = np.random.default_rng(8)
    shape = (37, 13, 15)            # two tiles of 32 nodes in x, several in y and z
    grid = mg_ref.Grid([rng.uniform(10, 30, n) for n in shape], (0, 0, 0))
    vm = mg_ref.volume_model(grid, 1.3 if dtype is complex else -1.3, *[10 ** rng.uniform(-1, 1, shape) for _ in range(3)])
    s, e0 = mg_ref.Field(grid, dtype=dtype), mg_ref.Field(grid, dtype=dtype)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size)
    out = {}
    try:
        for nt in (1, 6):
            ocore.lib().oracle_set_threads(nt)
            for fn, lr in LR.items():
                for order in ((1, 2) if lr == 0 else (1,)):
                    a = e0.copy()
                    getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                                       vm.zeta, *grid.h, 3, order=order)
                    out[(nt, fn, order)] = a.field.copy()
    finally:
        ocore.lib().oracle_set_threads(1)
    for (nt, fn, order), v in out.items():
        if nt == 6:
            assert np.array_equal(v, out[(1, fn, order)]), (fn, order)
            assert np.any(v != e0.field)


@pytest.mark.parametrize('order', [0, 2])
def test_line_orders_other_than_the_default_match_oracle(order):
    """launch.h: line_sweep_colour with the mirrored (0) and the repeated (2) sequence of the line passes --
    the kernel bodies walked on the CPU against the oracle in the same order (nu = 3: every sweep of a call
    differs under the cyclic default, none under order 2)."""
    rng = np.random.default_rng(50 + order)
    shape = (9, 6, 7)
    grid = mg_ref.Grid([rng.uniform(10, 30, n) for n in shape], (0, 0, 0))
    vm = mg_ref.volume_model(grid, 1.3, *[10 ** rng.uniform(-1, 1, shape) for _ in range(3)])
    s, e0 = mg_ref.Field(grid), mg_ref.Field(grid)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size) + 1j * rng.standard_normal(f.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    out = {}
    try:
        for o in (1, order):
            emu.lib().emu_set_line_order(o)
            ocore.lib().oracle_set_line_order(o, 1, 2, 3, 0)
            for fn, lr in LR.items():
                if lr == 0:
                    continue
                a, b = e0.copy(), e0.copy()
                getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                                   vm.zeta, *grid.h, 3, order=1)
                emu.gauss_seidel(b, s, vm, lr, 3)
                assert relerr(b.field, a.field) < 2e-12, (o, fn)
                out[(o, fn)] = b.field.copy()
    finally:
        emu.lib().emu_set_line_order(1)
        ocore.lib().oracle_set_line_order(1, 1, 2, 3, 0)
    assert all(relerr(out[(order, fn)], out[(1, fn)]) > 1e-6 for fn, lr in LR.items() if lr)


@pytest.mark.parametrize('slab', [1, 2, 3, 5, 64])
def test_point_slab_schedule_is_bit_identical(golden_kernels, slab):
    """The skewed plane-slab launch schedule of the point smoother (launch.h) must give
    exactly the values of the plain one-launch-per-colour schedule."""
    g = golden_kernels
    for name in ('c_tri', 'c_iso', 'r_tri', 'c_z2'):
        p = name + '_'
        grid, vm = _case(g, name)
        s = mg_ref.Field(grid, g[p + 'gs_s'].copy())
        for nu in (1, 2):
            a = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            b = mg_ref.Field(grid, g[p + 'gs_e_in'].copy())
            emu.lib().emu_set_point_slab(0)
            emu.gauss_seidel(a, s, vm, 0, nu)
            emu.lib().emu_set_point_slab(slab)
            try:
                emu.gauss_seidel(b, s, vm, 0, nu)
            finally:
                emu.lib().emu_set_point_slab(0)
            assert np.array_equal(a.field, b.field), (name, nu, slab)


@pytest.mark.parametrize('shape', [(36, 20, 18), (16, 8, 8), (18, 34, 10), (4, 6, 2)])
@pytest.mark.parametrize('dtype', [complex, float])
def test_point_tiled_schedule_matches_oracle_tile_order(shape, dtype):
    """Tiled schedule of the point smoother (launch.h: tile_load / tile_colour / tile_store,
    eight tile-colour launches) against the oracle's order 2, on grids with several and with
    partial tiles; tri-axial model."""
    rng = np.random.default_rng(sum(shape))
    h = [rng.uniform(0.5, 2.0, n) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sig = [10 ** rng.uniform(-1, 1, shape) for _ in range(3)]
    vm = mg_ref.volume_model(grid, 1.3 if dtype is complex else -1.3, *sig)
    s = mg_ref.Field(grid, dtype=dtype)
    e0 = mg_ref.Field(grid, dtype=dtype)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size)
        if dtype is complex:
            f.field[:] += 1j * rng.standard_normal(f.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    emu.lib().emu_set_point_tile_min(1)
    try:
        for nu in (1, 2, 3):
            a, b = e0.copy(), e0.copy()
            ocore.gauss_seidel(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                               vm.zeta, *grid.h, nu, order=2)
            emu.gauss_seidel(b, s, vm, 0, nu)
            assert relerr(b.field, a.field) < 5e-10, (shape, nu)   # random widths: roundoff
            # and it IS a different ordering than the plain four-colour one when > 1 tile
            if shape[0] > 16:
                c = e0.copy()
                ocore.gauss_seidel(c.fx, c.fy, c.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y,
                                   vm.eta_z, vm.zeta, *grid.h, nu, order=1)
                assert relerr(c.field, a.field) > 1e-6
    finally:
        emu.lib().emu_set_point_tile_min(1 << 20)


@pytest.mark.parametrize('freq', [1.0, 0.01, -3.0])
def test_two_sided_line_factorisation_keeps_accuracy_at_low_frequency(freq):
    """CPU emulation of the two-sided (twisted) line factorisation with the mirrored far
    half: same accuracy against the oracle's one-sided LDL^T as frequency drops (nearly
    singular blocks); lines of 64, 6 and 10 blocks exercise middle positions 32, 0 and 4."""
    shape = (64, 6, 10)
    rng = np.random.default_rng(1)
    h = [np.ones(n) * 20. for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sig = [10 ** rng.uniform(-1, 1, shape) for _ in range(3)]
    vm = mg_ref.volume_model(grid, freq, *sig)
    dtype = complex if freq > 0 else float
    s = mg_ref.Field(grid, dtype=dtype)
    e0 = mg_ref.Field(grid, dtype=dtype)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size)
        if dtype is complex:
            f.field[:] += 1j * rng.standard_normal(f.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    tol = {1.0: 2e-12, 0.01: 2e-10, -3.0: 2e-11}[freq]
    for fn, lr in LR.items():
        if lr == 0:
            continue
        a, b = e0.copy(), e0.copy()
        getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                           vm.zeta, *grid.h, 1, order=1)
        emu.gauss_seidel(b, s, vm, lr, 1)
        assert relerr(b.field, a.field) < tol, (fn, freq)


def test_bench_workloads_build():
    """Every bench.py workload builds (shapes, positive resistivities, source inside)."""
    from bench import workload
    for name, shape in (('marine32', (32, 32, 32)), ('triaxial64', (64, 64, 64)),
                        ('uniform32', (32, 32, 32)), ('salt96', (96, 64, 64))):
        for idx in (0, 3):
            wl = workload(name, idx)
            assert tuple(h.size for h in wl['h']) == shape
            for v in wl['res'].values():
                assert np.shape(v) == shape and np.all(np.asarray(v) > 0)
            for d in range(3):
                assert wl['origin'][d] < wl['source'][d] < wl['origin'][d] + wl['h'][d].sum()
    assert {workload('salt96', i)['frequency'] for i in range(8)} == {0.25, 0.5, 1.0, 2.0}


@pytest.mark.parametrize('shape', [(9, 12, 7), (24, 6, 10), (5, 5, 40)])
def test_recomputed_coupling_entries_equal_the_stored_records(shape):
    """k_line_stream's producers recompute the eight coupling entries of a block from zeta and the widths
    (stencil.h: line_coupling) instead of fetching the lfac record: for every line direction, several lines and
    every record the half-chains read (top blocks, mirrored blocks, the last block, the identity padding) the
    recomputed values must equal the set-up's stored ones bit for bit."""
    rng = np.random.default_rng(sum(shape))
    h = [rng.uniform(5., 15., n) * 1.1 ** np.abs(np.arange(n) - n // 2) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    vm = mg_ref.volume_model(grid, 0.7, *[10 ** rng.uniform(-1, 1, shape) for _ in range(3)],
                             mu_r=rng.uniform(0.8, 2.0, shape))
    e, s = mg_ref.Field(grid), mg_ref.Field(grid)
    for direction in (0, 1, 2):
        n1 = shape[(direction + 1) % 3]
        n2 = shape[(direction + 2) % 3]
        for i1, i2 in ((1, 1), (n1 - 1, n2 - 1), (max(n1 // 2, 1), 1), (1, max(n2 // 2, 1))):
            assert emu.line_coupling_mismatches(e, s, vm, direction, i1, i2) == 0, (direction, i1, i2)


@pytest.mark.parametrize('dtype', [complex, float])
@pytest.mark.parametrize('zb', [1, 3, 8])
def test_residual_column_walk_equals_cell_by_cell(dtype, zb):
    """The residual kernels walk a column of cells and carry the operands a cell shares with the cell below it
    (stencil.h: residual_load_roll): same values, same norm as cell by cell, bit for bit, for any number of planes
    per walk (also when the walk ends on the top boundary plane)."""
    shape = (7, 9, 11)
    rng = np.random.default_rng(zb)
    h = [rng.uniform(5., 15., n) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    vm = mg_ref.volume_model(grid, 0.7 if dtype is complex else -0.7, *[10 ** rng.uniform(-1, 1, shape) for _ in range(3)])
    e, s = mg_ref.Field(grid, dtype=dtype), mg_ref.Field(grid, dtype=dtype)
    for f in (e, s):
        f.field[:] = rng.standard_normal(f.field.size) + (1j * rng.standard_normal(f.field.size) if dtype is complex else 0)
    r1, r2 = mg_ref.Field(grid, dtype=dtype), mg_ref.Field(grid, dtype=dtype)
    n1 = emu.residual(e, s, vm, r1)
    n2 = emu.residual_column(e, s, vm, r2, zb)
    assert np.array_equal(r1.field, r2.field)
    assert n2 == pytest.approx(n1, rel=1e-13)


def _random_case(shape, freq, seed, stretch=True):
    rng = np.random.default_rng(seed)
    h = [rng.uniform(10, 30, n) if stretch else np.ones(n) * 20. for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sig = [10 ** rng.uniform(-1, 1, shape) for _ in range(3)]
    vm = mg_ref.volume_model(grid, freq, *sig)
    dtype = complex if freq > 0 else float
    s = mg_ref.Field(grid, dtype=dtype)
    e0 = mg_ref.Field(grid, dtype=dtype)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size)
        if dtype is complex:
            f.field[:] += 1j * rng.standard_normal(f.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    return grid, vm, s, e0


@pytest.mark.parametrize('shape,freq', [((16, 8, 12), 1.3), ((5, 7, 9), 1.3), ((3, 3, 3), 1.3), ((2, 9, 4), 1.3),
                                        ((14, 4, 6), 1.3), ((32, 6, 10), -2.0), ((64, 6, 10), 1.0),
                                        ((64, 6, 10), 0.01)])
def test_wide_line_form_matches_oracle(shape, freq):
    """stencil.h: line_wide_ref -- the line solve restated with 4 x 4 chains (what k_line_wide computes per line:
    N records, g / y chains, per-block c / w0 / g', middle block, h chains, x) against the oracle's four-colour
    order, per call; ragged line lengths (2 ... 64 blocks: middle positions 0 ... 32, halves of unequal length),
    complex and real, and at a low frequency (nearly singular blocks) with the tolerance the two-sided
    factorisation itself is held to."""
    grid, vm, s, e0 = _random_case(shape, freq, 7)
    tol = 2e-10 if freq == 0.01 else 2e-12
    try:
        emu.lib().emu_set_line_wide(1)
        for fn, lr in LR.items():
            if lr == 0:
                continue
            a, b = e0.copy(), e0.copy()
            getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                               vm.zeta, *grid.h, 2, order=1)
            emu.gauss_seidel(b, s, vm, lr, 2)
            assert relerr(b.field, a.field) < tol, (shape, fn)
    finally:
        emu.lib().emu_set_line_wide(0)
