"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Compact CPU restatement of the reference's multigrid driver (emg3d/solver.py)
on top of the C restatement of its kernels (oracle/core.py). It exists to

* check the HIP path at sizes where the un-jitted reference is too slow,
* be the CPU baseline that bench.py times next to the GPU numbers
  (``cpu_baseline.kind = "port"``).

Parity status: PINNED -- ``solve`` reproduces the reference's own golden file
(tests/data/regression.npz, re-exported to tests/golden/regression_small.npz)
and reference solves generated in the build container (tools/make_golden.py);
see tests/test_oracle.py.

Every function cites the reference lines it follows. No product code is
imported here and the product never imports this module.
"""
import itertools
import time

import numpy as np

from oracle import core

MU_0 = 1.25663706127e-06        # scipy 1.15.3 value
EPSILON_0 = 8.8541878188e-12


class Grid:
    """Attribute subset of emg3d.meshes.BaseMesh (emg3d/meshes.py:72-111)."""

    def __init__(self, h, origin):
        self.origin = np.array(origin, dtype=float)
        self.h = [np.array(h[0], dtype=float), np.array(h[1], dtype=float),
                  np.array(h[2], dtype=float)]
        self.shape_cells = tuple(x.size for x in self.h)
        self.shape_nodes = tuple(x.size + 1 for x in self.h)
        self.nodes_x = np.r_[0., self.h[0].cumsum()] + self.origin[0]
        self.nodes_y = np.r_[0., self.h[1].cumsum()] + self.origin[1]
        self.nodes_z = np.r_[0., self.h[2].cumsum()] + self.origin[2]
        self.cell_centers_x = (self.nodes_x[1:] + self.nodes_x[:-1]) / 2
        self.cell_centers_y = (self.nodes_y[1:] + self.nodes_y[:-1]) / 2
        self.cell_centers_z = (self.nodes_z[1:] + self.nodes_z[:-1]) / 2
        nx, ny, nz = self.shape_cells
        self.shape_edges_x = (nx, ny + 1, nz + 1)
        self.shape_edges_y = (nx + 1, ny, nz + 1)
        self.shape_edges_z = (nx + 1, ny + 1, nz)
        self.n_edges_x = int(np.prod(self.shape_edges_x))
        self.n_edges_y = int(np.prod(self.shape_edges_y))
        self.n_edges_z = int(np.prod(self.shape_edges_z))
        self.n_edges = self.n_edges_x + self.n_edges_y + self.n_edges_z
        self.n_cells = nx * ny * nz

    @property
    def cell_volumes(self):
        # emg3d/meshes.py:119-126
        return (self.h[0][None, None, :] * self.h[1][None, :, None] *
                self.h[2][:, None, None]).ravel()


class Field:
    """1-D buffer [fx|fy|fz] with F-order views (emg3d/fields.py:88-259)."""

    def __init__(self, grid, data=None, dtype=np.complex128):
        self.grid = grid
        if data is None:
            self.field = np.zeros(grid.n_edges, dtype=dtype)
        else:
            self.field = np.asarray(data)
        n1, n2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
        self.fx = self.field[:n1].reshape(grid.shape_edges_x, order='F')
        self.fy = self.field[n1:n2].reshape(grid.shape_edges_y, order='F')
        self.fz = self.field[n2:].reshape(grid.shape_edges_z, order='F')

    def copy(self):
        return Field(self.grid, self.field.copy())


class VModel:
    """eta/zeta holder; semantics of emg3d.models.VolumeModel (models.py:627-717)."""

    def __init__(self, grid, eta_x, eta_y, eta_z, zeta, case):
        self.grid, self.case = grid, case
        self.eta_x, self.eta_y, self.eta_z, self.zeta = eta_x, eta_y, eta_z, zeta


def volume_model(grid, frequency, cond_x, cond_y=None, cond_z=None, mu_r=None,
                 epsilon_r=None):
    """eta = -s mu0 V (sigma [+ s eps0 eps_r]); zeta = V/mu_r (models.py:654-691).

    Conductivities (S/m) are F-ordered (nx,ny,nz) arrays or scalars.
    """
    sval = -frequency if frequency < 0 else 2j * np.pi * frequency
    smu0 = sval * MU_0
    vol = grid.cell_volumes.reshape(grid.shape_cells, order='F')

    def eta(c):
        c = np.broadcast_to(np.asarray(c, dtype=float), grid.shape_cells)
        if epsilon_r is None:
            return np.asfortranarray(-smu0 * vol * c)
        return np.asfortranarray(-smu0 * vol * (c + sval * EPSILON_0 * epsilon_r))
    ex = eta(cond_x)
    ey = eta(cond_y) if cond_y is not None else ex
    ez = eta(cond_z) if cond_z is not None else ex
    case = {(False, False): 'isotropic', (True, False): 'HTI',
            (False, True): 'VTI', (True, True): 'triaxial'}[
                (cond_y is not None, cond_z is not None)]
    zeta = np.asfortranarray(vol.copy() if mu_r is None else vol / mu_r)
    return VModel(grid, ex, ey, ez, zeta, case)


# --- helper routines -----------------------------------------------------------
def current_sc_dir(sc_dir, grid):
    """solver.py:1482-1531."""
    n = grid.shape_cells
    xs = n[0] % 2 != 0 or n[0] < 3 or sc_dir == 1
    ys = n[1] % 2 != 0 or n[1] < 3 or sc_dir == 2
    zs = n[2] % 2 != 0 or n[2] < 3 or sc_dir == 3
    if xs:
        return 6 if ys else (5 if zs else 1)
    if ys:
        return 4 if zs else 2
    return 3 if zs else 0


def current_lr_dir(lr_dir, grid):
    """solver.py:1534-1588."""
    c = int(lr_dir)
    n = grid.shape_cells
    if n[0] == 2:
        c = {1: 0, 5: 3, 6: 2, 7: 4}.get(c, c)
    if n[1] == 2:
        c = {2: 0, 4: 3, 6: 1, 7: 5}.get(c, c)
    if n[2] == 2:
        c = {3: 0, 4: 2, 5: 1, 7: 6}.get(c, c)
    return c


def restrict_model_parameters(p, sc_dir):
    """Sum of 2/4/8 fine cells; solver.py:1667-1718."""
    a, b = slice(None, -1, 2), slice(1, None, 2)
    al = slice(None)
    if sc_dir == 1:
        return p[:, a, a] + p[:, b, a] + p[:, a, b] + p[:, b, b]
    if sc_dir == 2:
        return p[a, :, a] + p[b, :, a] + p[a, :, b] + p[b, :, b]
    if sc_dir == 3:
        return p[a, a, :] + p[b, a, :] + p[a, b, :] + p[b, b, :]
    if sc_dir == 4:
        return p[a, al, al] + p[b, al, al]
    if sc_dir == 5:
        return p[al, a, al] + p[al, b, al]
    if sc_dir == 6:
        return p[al, al, a] + p[al, al, b]
    out = p[a, a, a] + p[b, a, a]
    out = out + p[a, a, b] + p[b, a, b]
    out = out + p[a, b, a] + p[b, b, a]
    out = out + p[a, b, b] + p[b, b, b]
    return out


def restriction_weights(grid, cgrid, sc_dir):
    """solver.py:1721-1780."""
    out = []
    for d, skip in enumerate(([1, 5, 6], [2, 4, 6], [3, 4, 5])):
        nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)[d]
        cc = (grid.cell_centers_x, grid.cell_centers_y, grid.cell_centers_z)[d]
        cnodes = (cgrid.nodes_x, cgrid.nodes_y, cgrid.nodes_z)[d]
        ccc = (cgrid.cell_centers_x, cgrid.cell_centers_y, cgrid.cell_centers_z)[d]
        if sc_dir not in skip:
            out.append(core.restrict_weights(nodes, cc, grid.h[d], cnodes, ccc, cgrid.h[d]))
        else:
            z = np.zeros(grid.shape_nodes[d])
            out.append((z, np.ones(grid.shape_nodes[d]), z))
    return out


# Levels with at least this many interior nodes run the point smoother in the tiled order
# (order 2) when a coloured order is requested -- the rule of the HIP library
# (include/emg3d_amd.h, option "point_tile_min").
POINT_TILE_MIN = 1 << 20


def smoothing(model, sfield, efield, nu, lr_dir, order=0, tile_min=POINT_TILE_MIN):
    """solver.py:788-846."""
    inp = (sfield.fx, sfield.fy, sfield.fz, model.eta_x, model.eta_y, model.eta_z,
           model.zeta, model.grid.h[0], model.grid.h[1], model.grid.h[2], nu)
    c = current_lr_dir(lr_dir, model.grid)
    e = (efield.fx, efield.fy, efield.fz)
    if c == 0:
        n = model.grid.shape_cells
        tiled = order == 1 and tile_min > 0 and (n[0] - 1) * (n[1] - 1) * (n[2] - 1) >= tile_min
        core.gauss_seidel(*e, *inp, order=2 if tiled else order)
    if c in (1, 5, 6, 7):
        core.gauss_seidel_x(*e, *inp, order=order)
    if c in (2, 4, 6, 7):
        core.gauss_seidel_y(*e, *inp, order=order)
    if c in (3, 4, 5, 7):
        core.gauss_seidel_z(*e, *inp, order=order)
    return c


def residual(model, sfield, efield, norm=False):
    """solver.py:1022-1070."""
    r = sfield.copy()
    core.amat_x(r.fx, r.fy, r.fz, efield.fx, efield.fy, efield.fz, model.eta_x,
                model.eta_y, model.eta_z, model.zeta, *model.grid.h)
    if norm:
        return float(np.linalg.norm(r.field))
    return r


def restriction(model, sfield, res, sc_dir):
    """solver.py:849-944."""
    g = model.grid
    rx = 1 if sc_dir in (1, 5, 6) else 2
    ry = 1 if sc_dir in (2, 4, 6) else 2
    rz = 1 if sc_dir in (3, 4, 5) else 2
    ch = [np.diff(g.nodes_x[::rx]), np.diff(g.nodes_y[::ry]), np.diff(g.nodes_z[::rz])]
    cgrid = Grid(ch, g.origin)
    ex = np.asfortranarray(restrict_model_parameters(model.eta_x, sc_dir))
    ey = (np.asfortranarray(restrict_model_parameters(model.eta_y, sc_dir))
          if model.case in ('HTI', 'triaxial') else ex)
    ez = (np.asfortranarray(restrict_model_parameters(model.eta_z, sc_dir))
          if model.case in ('VTI', 'triaxial') else ex)
    zeta = np.asfortranarray(restrict_model_parameters(model.zeta, sc_dir))
    cmodel = VModel(cgrid, ex, ey, ez, zeta, model.case)
    wx, wy, wz = restriction_weights(g, cgrid, sc_dir)
    cs = Field(cgrid, dtype=sfield.field.dtype)
    core.restrict(cs.fx, cs.fy, cs.fz, res.fx, res.fy, res.fz, wx, wy, wz, sc_dir)
    ce = Field(cgrid, dtype=sfield.field.dtype)
    return cmodel, cs, ce


def _interp_1d(cx, x):
    """Lower index + weight of linear interpolation (solver.py:1457-1462)."""
    i = np.searchsorted(cx, x) - 1
    i[i < 0] = 0
    i[i > cx.size - 2] = cx.size - 2
    return i, (x - cx[i]) / (cx[i + 1] - cx[i])


def _bilinear(values, ia, wa, ib, wb):
    """RegularGridProlongator.__call__ (solver.py:1424-1473): sum over the four
    corner combinations in itertools.product order (a, b) = (0,0),(0,1),(1,0),(1,1)."""
    A, B = np.meshgrid(np.arange(ia.size), np.arange(ib.size), indexing='ij')
    out = 0.
    for da, db in itertools.product((0, 1), (0, 1)):
        w = (np.where(da == 0, 1 - wa, wa)[A]) * (np.where(db == 0, 1 - wb, wb)[B])
        out = out + values[(ia + da)[A], (ib + db)[B]] * w
    return out


def prolongation(efield, cefield, sc_dir):
    """solver.py:947-1019."""
    cg, g = cefield.grid, efield.grid
    iy, wy = _interp_1d(cg.nodes_y, g.nodes_y)
    iz, wz = _interp_1d(cg.nodes_z, g.nodes_z)
    ix, wx = _interp_1d(cg.nodes_x, g.nodes_x)
    for ixc in range(cg.shape_cells[0]):
        hh = _bilinear(cefield.fx[ixc, :, :], iy, wy, iz, wz)
        if sc_dir not in (1, 5, 6):
            efield.fx[2 * ixc, 1:-1, 1:-1] += hh[1:-1, 1:-1]
            efield.fx[2 * ixc + 1, 1:-1, 1:-1] += hh[1:-1, 1:-1]
        else:
            efield.fx[ixc, 1:-1, 1:-1] += hh[1:-1, 1:-1]
    for iyc in range(cg.shape_cells[1]):
        hh = _bilinear(cefield.fy[:, iyc, :], ix, wx, iz, wz)
        if sc_dir not in (2, 4, 6):
            efield.fy[1:-1, 2 * iyc, 1:-1] += hh[1:-1, 1:-1]
            efield.fy[1:-1, 2 * iyc + 1, 1:-1] += hh[1:-1, 1:-1]
        else:
            efield.fy[1:-1, iyc, 1:-1] += hh[1:-1, 1:-1]
    for izc in range(cg.shape_cells[2]):
        hh = _bilinear(cefield.fz[:, :, izc], ix, wx, iy, wy)
        if sc_dir not in (3, 4, 5):
            efield.fz[1:-1, 1:-1, 2 * izc] += hh[1:-1, 1:-1]
            efield.fz[1:-1, 1:-1, 2 * izc + 1] += hh[1:-1, 1:-1]
        else:
            efield.fz[1:-1, 1:-1, izc] += hh[1:-1, 1:-1]


class Params:
    """Subset of solver.MGParameters (solver.py:1074-1381) needed by multigrid()."""

    def __init__(self, shape_cells, cycle='F', semicoarsening=False, linerelaxation=False,
                 tol=1e-6, maxit=50, nu_init=0, nu_pre=2, nu_coarse=1, nu_post=2,
                 clevel=-1, order=0, tile_min=POINT_TILE_MIN):
        self.cycle, self.tol, self.maxit = cycle, tol, maxit
        self.nu_init, self.nu_pre, self.nu_coarse, self.nu_post = nu_init, nu_pre, nu_coarse, nu_post
        self.order, self.tile_min = order, tile_min
        self.it, self.l2, self.l2_refe = 0, 1.0, 1.0
        self.exit_message = ''
        self.error_at_cycle = [0.]
        self.runtime_at_cycle = [0.]
        self.smooth_work = 0           # sum of nu * n_cells over smoother calls
        self.t0 = time.perf_counter()
        # _max_level, solver.py:1202-1232
        cl = np.zeros(3, dtype=int)
        for i in range(3):
            n = shape_cells[i]
            while n % 2 == 0 and n > 2:
                cl[i] += 1
                n /= 2
        for i in range(3):
            if -1 < clevel < cl[i]:
                cl[i] = clevel
        self.clevel = [max(cl), max(cl[1], cl[2]), max(cl[0], cl[2]), max(cl[0], cl[1])]
        # _semicoarsening / _linerelaxation, solver.py:1272-1339
        self.sc_cycle, sc_list = self._cyc(semicoarsening, [1, 2, 3], 4)
        self.lr_cycle, lr_list = self._cyc(linerelaxation, [4, 5, 6], 8)
        self.sc_dir = next(self.sc_cycle) if self.sc_cycle else sc_list[0]
        self.lr_dir = next(self.lr_cycle) if self.lr_cycle else lr_list[0]
        self.cycmax = 2 if cycle in ('F', 'W') else 1       # solver.py:1361-1364
        self.maxcycle = max(len(sc_list), len(lr_list))     # solver.py:1376

    @staticmethod
    def _cyc(val, true_list, nmax):
        if val is True:
            return itertools.cycle(true_list), true_list
        if val is False or int(val) in range(nmax):
            return False, [int(val)]
        lst = [int(x) for x in str(abs(int(val)))]
        return itertools.cycle(lst), lst


# number of smoother kernel calls per smoothing() for a (current) lr_dir
NDIR = {0: 1, 1: 1, 2: 1, 3: 1, 4: 2, 5: 2, 6: 2, 7: 3}


def terminate(var, l2_last, l2_stag, it):
    """solver.py:1591-1664 (stand-alone multigrid branch)."""
    if l2_last < var.tol * var.l2_refe:
        var.exit_message = "CONVERGED"
    elif l2_last > 10 * var.l2_refe or not np.isfinite(l2_last):
        var.exit_message = "DIVERGED"
    elif it > 2 and l2_last >= l2_stag:
        var.exit_message = "STAGNATED"
    elif it == var.maxit:
        var.exit_message = "MAX. ITERATION REACHED, NOT CONVERGED"
    else:
        return False
    return True


def multigrid(model, sfield, efield, var, level=0, new_cycmax=0):
    """solver.py:471-649."""
    it = 0
    if level == var.clevel[var.sc_dir]:
        cycmax = 1
    elif new_cycmax == 0 or var.cycle != 'F':
        cycmax = var.cycmax
    else:
        cycmax = new_cycmax
    cyc = 0
    ncell = model.grid.n_cells

    if level == 0:  # the norm on coarse levels is unused by the control flow
        l2_last = residual(model, sfield, efield, True)
        l2_stag = np.ones(var.maxcycle) * l2_last
        if var.nu_init > 0:
            c = smoothing(model, sfield, efield, var.nu_init, var.lr_dir, var.order, var.tile_min)
            var.smooth_work += var.nu_init * ncell * NDIR[c]

    while level == 0 or it < cycmax:
        if level == 0:
            l2_prev = l2_last  # noqa: F841
            l2_stag[(it - 1) % var.maxcycle] = l2_last
        if level == var.clevel[var.sc_dir]:
            c = smoothing(model, sfield, efield, var.nu_coarse, var.lr_dir, var.order, var.tile_min)
            var.smooth_work += var.nu_coarse * ncell * NDIR[c]
        else:
            if var.nu_pre > 0:
                c = smoothing(model, sfield, efield, var.nu_pre, var.lr_dir, var.order, var.tile_min)
                var.smooth_work += var.nu_pre * ncell * NDIR[c]
            sc_dir = current_sc_dir(var.sc_dir, model.grid)
            res = residual(model, sfield, efield)
            cmodel, csfield, cefield = restriction(model, sfield, res, sc_dir)
            multigrid(cmodel, csfield, cefield, var, level + 1, cycmax - cyc)
            prolongation(efield, cefield, sc_dir)
            if var.nu_post > 0:
                c = smoothing(model, sfield, efield, var.nu_post, var.lr_dir, var.order, var.tile_min)
                var.smooth_work += var.nu_post * ncell * NDIR[c]
        it += 1
        if level == 0:
            var.it += 1
        if level > 0:
            cyc += 1
        else:
            l2_last = residual(model, sfield, efield, True)
            var.error_at_cycle.append(l2_last)
            var.runtime_at_cycle.append(time.perf_counter() - var.t0)
            if var.sc_cycle:
                var.sc_dir = next(var.sc_cycle)
            if var.lr_cycle:
                var.lr_dir = next(var.lr_cycle)
            if terminate(var, l2_last, l2_stag[(it - 1) % var.maxcycle], it):
                break
    if level == 0:
        var.l2 = l2_last


def solve(model, sfield, efield=None, **kwargs):
    """Stand-alone multigrid branch of solver.solve (solver.py:52-449).

    Returns (efield, info)."""
    var = Params(model.grid.shape_cells, **kwargs)
    var.l2_refe = float(np.linalg.norm(sfield.field))
    var.error_at_cycle[0] = var.l2_refe
    if efield is None:
        efield = Field(model.grid, dtype=sfield.field.dtype)
    multigrid(model, sfield, efield, var)
    info = {'exit': int(var.exit_message != 'CONVERGED'), 'exit_message': var.exit_message,
            'abs_error': var.l2, 'rel_error': var.l2 / var.l2_refe, 'ref_error': var.l2_refe,
            'tol': var.tol, 'it_mg': var.it, 'error_at_cycle': np.array(var.error_at_cycle),
            'runtime_at_cycle': np.array(var.runtime_at_cycle), 'smooth_work': var.smooth_work}
    return efield, info
