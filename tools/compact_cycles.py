"""Oracle-side study BEFORE the kernel (review of round 5, item 1a): what do single-precision STORED line factors
and w records cost in multigrid cycles?

The multigrid driver is the oracle's (oracle/mg_ref.py) with the finest level in residual form (every cycle solves
A d = s - A e from d = 0: what `emg3d_amd.solve(residual_form=True)` does); the line smoothers are the CPU walk of the
library's own two-sided block factorisation (tests/emu: stencil.h compiled with g++), once with fp64 records and once
in the COMPACT form -- T records rounded to single precision when the set-up stores them, w records rounded when the
forward pass stores them, every operation in fp64 -- on EVERY level and direction (the GPU uses it on the two
largest levels only: this is the pessimistic case). Same four-colour cyclic order on both sides.

    python tools/compact_cycles.py [--workloads triaxial64,marine64,salt96] [--tol 1e-10] [--air RHO]

Output: cycles to tol, final relative error, rel-L2 between the two converged fields, and eps32 * cond estimate of
the model (the quantity the library's `line_compact='auto'` rule bounds). TEST INFRASTRUCTURE (oracle + emu) only.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import mg_ref                                   # noqa: E402
from emu import emu                                         # noqa: E402
import bench                                                # noqa: E402
import emg3d_amd as emg3d                                   # noqa: E402


class _F:
    def __init__(self, fx, fy, fz):
        self.fx, self.fy, self.fz = fx, fy, fz


class _VM:
    pass


def _emu_smoother(lr):
    def fn(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, nu, order=1):
        vm = _VM()
        vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta = eta_x, eta_y, eta_z, zeta
        vm.grid = mg_ref.Grid([hx, hy, hz], (0, 0, 0))
        emu.gauss_seidel(_F(ex, ey, ez), _F(sx, sy, sz), vm, lr, nu)
    return fn


def solve_residual_form(vm, sfield, tol, maxit, opts):
    var = mg_ref.Params(vm.grid.shape_cells, tol=tol, maxit=1, order=1, **opts)
    l2_refe = float(np.linalg.norm(sfield.field))
    e = mg_ref.Field(vm.grid, dtype=sfield.field.dtype)
    hist = []
    for it in range(maxit):
        r = mg_ref.residual(vm, sfield, e)
        d = mg_ref.Field(vm.grid, dtype=sfield.field.dtype)
        var.it, var.maxit = 0, 1
        mg_ref.multigrid(vm, r, d, var)
        e.field += d.field
        l2 = mg_ref.residual(vm, sfield, e, True)
        hist.append(l2 / l2_refe)
        print(f"      cycle {it + 1:2d}: {l2 / l2_refe:.3e}", flush=True)
        if l2 < tol * l2_refe or not np.isfinite(l2) or l2 > 10 * l2_refe:
            break
    return e, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workloads', default='triaxial64,marine64,salt96')
    ap.add_argument('--tol', type=float, default=1e-10)
    ap.add_argument('--maxit', type=int, default=40)
    ap.add_argument('--air', type=float, default=0.0, help='resistivity of an air layer put on top of the model (0: none)')
    args = ap.parse_args()
    for fn, lr in (('gauss_seidel_x', 1), ('gauss_seidel_y', 2), ('gauss_seidel_z', 3)):
        setattr(mg_ref.core, fn, _emu_smoother(lr))
    lib = emu.lib()
    for name in args.workloads.split(','):
        wl = bench.workload(name)
        grid = emg3d.TensorMesh(wl['h'], wl['origin'])
        sf = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        og = mg_ref.Grid(grid.h, grid.origin)
        res = {k: np.array(v, dtype=float) for k, v in wl['res'].items()}
        if args.air > 0:
            for v in res.values():
                v[:, :, -max(2, v.shape[2] // 8):] = args.air
        cond = {k: 1.0 / v for k, v in res.items()}
        vm = mg_ref.volume_model(og, wl['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
        hmin = min(float(np.min(h)) for h in grid.h)
        sig_min = min(float(np.min(c)) for c in cond.values())
        blockcond = 1.0 / (2 * np.pi * wl['frequency'] * mg_ref.MU_0 * sig_min * hmin ** 2)
        print(f"{name}: {grid.shape_cells}, {wl['opts']}, air={args.air}: block cond estimate {blockcond:.2e}, "
              f"eps32 x cond = {blockcond * 2 ** -24:.2e}", flush=True)
        out = {}
        for compact in (0, 1):
            lib.emu_set_line_compact(compact)
            print(f"   {'compact (fp32-stored T and w)' if compact else 'fp64 records'}:", flush=True)
            t0 = time.perf_counter()
            e, hist = solve_residual_form(vm, mg_ref.Field(og, sf.field.copy()), args.tol, args.maxit, wl['opts'])
            out[compact] = (e, hist, time.perf_counter() - t0)
        lib.emu_set_line_compact(0)
        (e0, h0, t0_), (e1, h1, t1_) = out[0], out[1]
        diff = np.linalg.norm(e0.field - e1.field) / np.linalg.norm(e0.field)
        print(f"   => cycles fp64 / compact: {len(h0)} / {len(h1)}; final rel. error {h0[-1]:.2e} / {h1[-1]:.2e}; "
              f"rel-L2 between the converged fields {diff:.2e}; largest ratio of per-cycle errors "
              f"{max(b / a for a, b in zip(h0, h1)):.3f} ({t0_:.0f} + {t1_:.0f} s)", flush=True)


if __name__ == '__main__':
    main()
