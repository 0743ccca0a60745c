"""Build-container-only check: oracle/ (C restatement) against the reference itself.

Imports the reference un-jitted from /root/reference (stub
packages in tools/oracle_stubs contain no reference code) and compares every
``core`` function of the oracle with the reference's on seeded random inputs,
real and complex, isotropic (aliased eta) and tri-axial, on small stretched grids.

Run:  python tools/check_oracle_vs_reference.py
This script never runs on the GPU box (no /root/reference there).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, 'oracle_stubs'), '/root/reference', ROOT]

import emg3d  # noqa: E402  (the reference)
from emg3d import core as rcore  # noqa: E402
from oracle import core as ocore  # noqa: E402


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    nb = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (nb if nb > 0 else 1.0)


def make_case(rng, shape, dtype, aniso):
    nx, ny, nz = shape
    hx = 50 * 1.1 ** rng.uniform(-1, 3, nx)
    hy = 40 * 1.2 ** rng.uniform(-1, 3, ny)
    hz = 30 * 1.3 ** rng.uniform(-1, 3, nz)
    grid = emg3d.TensorMesh([hx, hy, hz], origin=(0, 0, 0))
    vol = grid.cell_volumes.reshape(grid.shape_cells, order='F')
    s = 2j * np.pi * 0.7 if dtype == np.complex128 else 1.3
    smu0 = s * 1.25663706127e-06

    def eta():
        sig = 10 ** rng.uniform(-2, 1, grid.shape_cells)
        return np.asfortranarray(-smu0 * vol * sig).astype(dtype)
    eta_x = eta()
    eta_y = eta() if aniso else eta_x
    eta_z = eta() if aniso else eta_x
    zeta = np.asfortranarray(vol / rng.uniform(0.8, 1.5, grid.shape_cells))

    def field(pec=True):
        f = emg3d.Field(grid, dtype=dtype)
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
    return grid, (eta_x, eta_y, eta_z, zeta), field


def main():
    rng = np.random.default_rng(20260928)
    worst = 0.0
    for shape in [(4, 6, 8), (8, 4, 6), (2, 4, 4), (4, 2, 6), (6, 4, 2), (6, 6, 6)]:
        for dtype in (np.complex128, np.float64):
            for aniso in (False, True):
                grid, model, field = make_case(rng, shape, dtype, aniso)
                h = grid.h
                # amat_x (non-zero boundary values on purpose: App. B.7)
                e, r = field(pec=False), field(pec=False)
                r1, r2 = r.copy(), r.copy()
                rcore.amat_x(r1.fx, r1.fy, r1.fz, e.fx, e.fy, e.fz, *model, *h)
                ocore.amat_x(r2.fx, r2.fy, r2.fz, e.fx, e.fy, e.fz, *model, *h)
                err = relerr(r2.field, r1.field); worst = max(worst, err)
                assert err < 1e-13, ('amat_x', shape, dtype, aniso, err)
                # smoothers
                for name in ('gauss_seidel', 'gauss_seidel_x', 'gauss_seidel_y',
                             'gauss_seidel_z'):
                    for nu in (1, 2, 3):
                        e, s = field(), field()
                        e1, e2 = e.copy(), e.copy()
                        getattr(rcore, name)(e1.fx, e1.fy, e1.fz, s.fx, s.fy, s.fz,
                                             *model, *h, nu)
                        getattr(ocore, name)(e2.fx, e2.fy, e2.fz, s.fx, s.fy, s.fz,
                                             *model, *h, nu)
                        err = relerr(e2.field, e1.field); worst = max(worst, err)
                        assert err < 1e-12, (name, shape, dtype, aniso, nu, err)
        print(f"shape {shape}: ok (worst so far {worst:.2e})")

    # solve / blocks_to_amat
    for dtype in (np.complex128, np.float64):
        for n in (6, 11, 16, 5 * 8 - 4):
            amat = rng.standard_normal(6 * n).astype(dtype)
            if dtype == np.complex128:
                amat = amat + 1j * rng.standard_normal(6 * n)
            amat[::6] += 20
            b = rng.standard_normal(n).astype(dtype)
            a1, a2, b1, b2 = amat.copy(), amat.copy(), b.copy(), b.copy()
            rcore.solve(a1, b1); ocore.solve(a2, b2)
            assert relerr(b2, b1) < 1e-13 and relerr(a2, a1) < 1e-13
        for nc in (2, 3, 5):
            n = 5 * nc - 4
            a1 = np.zeros(6 * n, dtype); a2 = a1.copy()
            b1 = np.zeros(n, dtype); b2 = b1.copy()
            for im in range(nc):
                mid = rng.standard_normal(25).astype(dtype)
                left = rng.standard_normal(25)
                rhs = rng.standard_normal(5).astype(dtype)
                rcore.blocks_to_amat(a1, b1, mid, left, rhs, im, nc)
                ocore.blocks_to_amat(a2, b2, mid, left, rhs, im, nc)
            assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    print("solve / blocks_to_amat: ok")

    # restrict + restrict_weights, all seven sc_dir
    from emg3d import solver as rsolver
    for dtype in (np.complex128, np.float64):
        for shape in [(8, 4, 12), (4, 4, 4), (6, 8, 4)]:
            grid, model, field = make_case(rng, shape, dtype, True)
            res = field(pec=False)
            for sc_dir in range(7):
                rx, ry, rz = [1 if sc_dir in s else 2 for s in
                              ([1, 5, 6], [2, 4, 6], [3, 4, 5])]
                ch = [np.diff(grid.nodes_x[::rx]), np.diff(grid.nodes_y[::ry]),
                      np.diff(grid.nodes_z[::rz])]
                cgrid = emg3d.meshes.BaseMesh(ch, grid.origin)
                wx, wy, wz = rsolver._get_restriction_weights(grid, cgrid, sc_dir)
                c1 = emg3d.Field(cgrid, dtype=dtype); c2 = emg3d.Field(cgrid, dtype=dtype)
                rcore.restrict(c1.fx, c1.fy, c1.fz, res.fx, res.fy, res.fz, wx, wy, wz, sc_dir)
                ocore.restrict(c2.fx, c2.fy, c2.fz, res.fx, res.fy, res.fz, wx, wy, wz, sc_dir)
                err = relerr(c2.field, c1.field); worst = max(worst, err)
                assert err < 1e-14, ('restrict', shape, dtype, sc_dir, err)
            w1 = rcore.restrict_weights(grid.nodes_x, grid.cell_centers_x, grid.h[0],
                                        cgrid.nodes_x, cgrid.cell_centers_x, cgrid.h[0]) \
                if False else None
        # weights on a coarsened direction
        g = emg3d.meshes.BaseMesh([50 * 1.2 ** np.arange(8.), [1, 1], [1, 1]], (3, 0, 0))
        cg = emg3d.meshes.BaseMesh([np.diff(g.nodes_x[::2]), [1, 1], [1, 1]], (3, 0, 0))
        w1 = rcore.restrict_weights(g.nodes_x, g.cell_centers_x, g.h[0],
                                    cg.nodes_x, cg.cell_centers_x, cg.h[0])
        w2 = ocore.restrict_weights(g.nodes_x, g.cell_centers_x, g.h[0],
                                    cg.nodes_x, cg.cell_centers_x, cg.h[0])
        for a, b in zip(w1, w2):
            assert np.allclose(a, b, rtol=1e-15, atol=0)
    print("restrict / restrict_weights: ok")
    print(f"ALL OK; worst rel-L2 deviation oracle vs reference = {worst:.3e}")


if __name__ == '__main__':
    main()
