"""GPU micro-benchmarks of single kernels at full size (run through gpurun).

    python tools/microbench.py point --n 256 --slabs 0,4,8,16,32
    python tools/microbench.py lines --n 128
    python tools/microbench.py residual --n 256
    python tools/microbench.py around --n 256      (Krylov steps, operator, gradient gather: SURVEY.md 8f)
Values: complex standard normal fields (PEC zeroed), model from the config.
Timing: torch.cuda events on the launch stream, warm-up 2, median of `reps`.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emg3d_amd import _lib                      # noqa: E402
from emg3d_amd._device import DeviceLevel       # noqa: E402
import emg3d_amd as emg3d                       # noqa: E402
from bench import widths, BYTES_PER_CELL_SWEEP  # noqa: E402


def make_level(n, case, stretch=1.03, shape=None, eta_real=False, batch=1):
    shape = shape or (n, n, n)
    rng = np.random.default_rng(1)
    h = [widths(m - 2 * (m // 4), m // 4, 25., stretch) for m in shape]
    grid = emg3d.TensorMesh(h, (0, 0, 0))
    vol = grid.cell_volumes.reshape(shape, order='F')
    smu0 = 2j * np.pi * 1.25663706127e-06

    class VM:
        pass
    vm = VM()
    vm.grid, vm.case = grid, case
    sig = 10 ** rng.uniform(-1.5, 0.5, shape)
    vm.eta_x = np.asfortranarray(-smu0 * vol * sig)
    if eta_real:            # as with epsilon_r given: eta gets a real part (16-byte eta sums)
        vm.eta_x = np.asfortranarray(vm.eta_x + 1e-3 * np.abs(vm.eta_x.imag))
    vm.eta_y = np.asfortranarray(vm.eta_x / 1.5) if case == 'triaxial' else vm.eta_x
    vm.eta_z = np.asfortranarray(vm.eta_x / 2.5) if case in ('VTI', 'triaxial') else vm.eta_x
    vm.zeta = np.asfortranarray(vol)
    lv = DeviceLevel.from_host(vm, torch.device('cuda'), batch=batch)
    gen = torch.Generator(device='cuda').manual_seed(1)
    one = DeviceLevel.from_host(vm, torch.device('cuda')) if batch > 1 else lv
    for t in (lv.e, lv.s):
        for b in range(batch):                    # every right-hand side its own values, PEC faces zeroed
            one.e.copy_(torch.complex(torch.randn(grid.n_edges, generator=gen, device='cuda', dtype=torch.float64),
                                      torch.randn(grid.n_edges, generator=gen, device='cuda', dtype=torch.float64)))
            one.pec_zero()
            t[b * grid.n_edges:(b + 1) * grid.n_edges].copy_(one.e)
    del one
    return lv, grid


def timeit(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def report(name, ms, ncells, nsweeps, case):
    b = BYTES_PER_CELL_SWEEP[case] * ncells * nsweeps
    print(f"{name:40s} {ms:9.3f} ms  {ncells * nsweeps / ms / 1e6:8.2f} Gcell-sweeps/s  "
          f"{b / ms / 1e6:8.1f} GB/s algorithmic  ({100 * b / ms / 1e6 / 8000:5.1f}% of 8 TB/s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['point', 'lines', 'residual', 'around', 'all', 'batchcmp', 'optcmp'])
    ap.add_argument('--variant', action='append', default=[], help="optcmp: option set 'name=value,name=value' ('' = defaults); repeatable")
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--case', default='triaxial')
    ap.add_argument('--slabs', default='0,2,4,8,16,32')
    ap.add_argument('--nu', type=int, default=2)
    ap.add_argument('--shape', default='', help='nx,ny,nz instead of n^3')
    ap.add_argument('--fused-only', action='store_true', help='line smoothers: the default fused launches only')
    ap.add_argument('--eta-real', action='store_true', help='eta with a real part (as with epsilon_r)')
    ap.add_argument('--opt', action='append', default=[], help='library option name=value (repeatable)')
    ap.add_argument('--batch', type=int, default=1, help='right-hand sides that share the level (lines only)')
    args = ap.parse_args()
    lib = _lib.lib()
    for o in args.opt:
        k, v = o.split('=')
        assert lib.emg3d_set_option(k.encode(), int(v)) == 0, o
    shape = tuple(int(x) for x in args.shape.split(',')) if args.shape else None
    if args.what == 'batchcmp':
        return batchcmp(args, shape)
    if args.what == 'optcmp':
        return optcmp(args, shape)
    lv, grid = make_level(args.n, args.case, shape=shape, eta_real=args.eta_real, batch=args.batch)
    nc = grid.n_cells
    print(f"# {shape or args.n} {args.case}, nu={args.nu}, batch={args.batch}")
    if args.what in ('point', 'all'):
        lib.emg3d_set_option(b'point_tile_min', 1)
        med, mn = timeit(lambda: lv.smooth(0, args.nu))
        report("gauss_seidel (point) tiled", med, nc, args.nu, args.case)
        lib.emg3d_set_option(b'point_tile_min', 0)
        for slab in [int(x) for x in args.slabs.split(',') if x != '']:
            lib.emg3d_set_option(b'point_slab', slab)
            med, mn = timeit(lambda: lv.smooth(0, args.nu))
            report(f"gauss_seidel (point) slab={slab}", med, nc, args.nu, args.case)
        lib.emg3d_set_option(b'point_slab', 0)
        lib.emg3d_set_option(b'point_tile_min', 1 << 20)
    if args.what in ('lines', 'all'):
        for fuse in ((1,) if args.fused_only else (0, 1)):
            lib.emg3d_set_option(b'line_fuse', fuse)
            for lr in (1, 2, 3):
                med, mn = timeit(lambda: lv.smooth(lr, args.nu), reps=5, warm=1)
                kn = lib.emg3d_line_kernel_name(lr, *grid.shape_cells, 1, args.batch).decode()
                report(f"gauss_seidel_{'xyz'[lr - 1]} fuse={fuse} {kn} /source", med / args.batch, nc, args.nu, args.case)
        lib.emg3d_set_option(b'line_fuse', 2)
    if args.what in ('residual', 'all'):
        med, mn = timeit(lambda: lv.residual(store=True, norm=False))
        report("residual (store)", med, nc, 1, args.case)
        med, mn = timeit(lambda: lv.residual(store=False, norm=True))
        report("residual (norm only, incl. sync)", med, nc, 1, args.case)
    if args.what in ('around', 'all'):
        around(lv, grid)


def batchcmp(args, shape):
    """Line passes with 1, 2 and 4 right-hand sides on ONE level in ONE process, interleaved over several rounds
    (consecutive processes on one box differ by up to 8 %: clocks): ms per source and call, medians over the
    rounds, and the ratio to the single source of the same round. The batched levels share the single level's
    factor buffers."""
    lib = _lib.lib()
    one, grid = make_level(args.n, args.case, shape=shape)
    levels = {1: one}
    for b in (2, 4):
        levels[b], _ = make_level(args.n, args.case, shape=shape, batch=b)
        levels[b]._factors = one._factors
    for lr in (1, 2, 3):
        one.smooth(lr, args.nu)
    rounds = 5
    res = {(b, lr): [] for b in levels for lr in (1, 2, 3)}
    for r in range(rounds):
        for lr in (1, 2, 3):
            for b, lv in levels.items():
                med, _ = timeit(lambda: lv.smooth(lr, args.nu), reps=3, warm=1)
                res[(b, lr)].append(med / b)
    print(f"# {grid.shape_cells} {args.case}, nu={args.nu}: ms per source and call of {4 * args.nu - (args.nu - 1)} launches, "
          f"median of {rounds} interleaved rounds (ratio to one source: median of the per-round ratios)")
    for lr in (1, 2, 3):
        base = np.array(res[(1, lr)])
        line = f"gauss_seidel_{'xyz'[lr - 1]}:"
        for b in levels:
            v = np.array(res[(b, lr)])
            kn = lib.emg3d_line_kernel_name(lr, *grid.shape_cells, 1, b).decode()
            line += f"  B={b} {kn} {np.median(v):.3f} ms ({np.median(v / base):.3f} x)"
        print(line, flush=True)


def optcmp(args, shape):
    """Line passes under several option sets on ONE box in ONE process, interleaved over several rounds (consecutive
    processes differ by up to 8 %): every variant gets its own level (its factors are built under its options), the
    options are set before each timed call. ms per call of 4 nu - (nu - 1) launches, median over the rounds, and the
    median of the per-round ratios to the first variant."""
    lib = _lib.lib()
    variants = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in v.split(',') if kv) for v in (args.variant or [''])]
    names = sorted({k for v in variants for k in v})
    defaults = {k: lib.emg3d_get_option(k.encode()) for k in names}

    def apply(v):
        for k in names:
            assert lib.emg3d_set_option(k.encode(), v.get(k, defaults[k])) == 0, k
    levels = []
    for v in variants:
        apply(v)
        lv, grid = make_level(args.n, args.case, shape=shape)
        for lr in (1, 2, 3):
            lv.smooth(lr, args.nu)
        levels.append(lv)
    res = {(i, lr): [] for i in range(len(variants)) for lr in (1, 2, 3)}
    for r in range(args.rounds):
        for lr in (1, 2, 3):
            for i, (v, lv) in enumerate(zip(variants, levels)):
                apply(v)
                med, _ = timeit(lambda: lv.smooth(lr, args.nu), reps=3, warm=1)
                res[(i, lr)].append(med)
    apply({})
    nl = 4 * args.nu - (args.nu - 1)
    print(f"# {grid.shape_cells} {args.case}, nu={args.nu}: ms per call of {nl} launches (ms per launch; % of the 8 TB/s roofline per "
          f"launch on {BYTES_PER_CELL_SWEEP[args.case]} B per cell-sweep), median of {args.rounds} interleaved rounds")
    for i, v in enumerate(variants):
        line = f"{str(v):60s}"
        for lr in (1, 2, 3):
            t = np.array(res[(i, lr)])
            base = np.array(res[(0, lr)])
            frac = BYTES_PER_CELL_SWEEP[args.case] * grid.n_cells / 4 / (np.median(t) / nl * 1e-3) / 8e12
            line += f"  {'xyz'[lr - 1]} {np.median(t):6.3f} ({np.median(t) / nl:.3f}; {100 * frac:4.1f} %; {np.median(t / base):.3f} x)"
        print(line, flush=True)


def around(lv, grid):
    """Kernels either side of the cycle: bytes they must move / time."""
    import ctypes
    from emg3d_amd import _krylov
    from emg3d_amd._device import _ptr, _stream
    lib = _lib.lib()
    n, nc = grid.n_edges, grid.n_cells

    def line(name, ms, nbytes):
        print(f"{name:52s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:5.1f}% of 8 TB/s)",
              flush=True)

    V = _krylov.Vectors(lv)
    a, b, c, d = (V.new() for _ in range(4))
    for t in (a, b, c, d):
        t.copy_(lv.e)
    for nm in ('beta', 'nbo', 'alpha', 'omega'):
        V.slot(nm)
    V.table[:8] = 0.5
    med, _ = timeit(lambda: V.step(a, [(b, 1.0), (a, 'beta'), (c, 'nbo')]))
    line("krylov step  p = r + beta p - beta omega v", med, 4 * 16 * n)
    med, _ = timeit(lambda: V.step(a, [(a, 1.0), (b, 'alpha')], dots=[('rr', a, a), ('rho', d, a)]))
    line("krylov step  r -= alpha v ; r.r ; rt.r", med, 4 * 16 * n)
    med, _ = timeit(lambda: V.step(None, dots=[('ts', a, b), ('tt', a, a)]))
    line("krylov step  t.s ; t.t (no update)", med, 2 * 16 * n)
    med, _ = timeit(lambda: lv.apply_A(a, b))
    line("operator     y = A x", med, (2 * 48 + 56) * nc)          # field in, out, eta x3 (8 B) + zeta + 3 h
    med, _ = timeit(lambda: lv.residual_sumsq(a, b))
    line("true residual |b - A x|^2 (no store)", med, (2 * 48 + 56) * nc)
    med, _ = timeit(lambda: V.copy(a, b))
    line("copy", med, 2 * 16 * n)
    # adjoint gradient gather
    nx, ny, nz = grid.shape_cells
    o1, o2 = grid.n_edges_x, grid.n_edges_x + grid.n_edges_y
    grad = torch.zeros(3 * nc, dtype=torch.float64, device='cuda')
    vol = torch.from_numpy(np.ascontiguousarray(grid.cell_volumes)).cuda()
    med, _ = timeit(lambda: _lib.check(lib.emg3d_dev_gradient_accumulate(
        nx, ny, nz, 1, _ptr(a), _ptr(a, o1), _ptr(a, o2), _ptr(b), _ptr(b, o1), _ptr(b, o2), 0.0, 7.9e-6, _ptr(vol),
        _ptr(grad), _ptr(grad, nc), _ptr(grad, 2 * nc), _stream()), 'grad'))
    line("gradient gather  g += cells(re(b s mu0 e))", med, (2 * 48 + 8 + 2 * 24) * nc)
    # before a solve: source vector on the device, model re-gridding, eta / zeta
    from emg3d_amd import fields, models
    cx, cy, cz = (0.5 * (g[0] + g[-1]) for g in (grid.nodes_x, grid.nodes_y, grid.nodes_z))
    pts = np.array([[cx - 50.3, cy + 3.1, cz - 7.7], [cx + 50.1, cy - 2.2, cz + 9.9]])
    med, _ = timeit(lambda: fields.source_field_device(grid, pts, 1.0, out=a))
    line("source vector of a 100 m dipole (incl. zero fill)", med, 16 * n)
    model = emg3d.Model(grid, property_x=np.full(grid.shape_cells, 1.0), property_z=np.full(grid.shape_cells, 2.0),
                        mapping='Resistivity')
    sf = emg3d.Field(grid, frequency=1.0)
    med, _ = timeit(lambda: models.VolumeModel(model, sf).device_arrays(torch.device('cuda')), reps=5, warm=1)
    line("eta, zeta from the properties (incl. 2 x 134 MB upload)", med, (2 * 8 + 2 * 16 + 8) * nc)
    h2 = [np.r_[g[0], 0.5 * (g[:-1:3] + g[1::3])[1:], g[-1]] for g in (grid.nodes_x, grid.nodes_y, grid.nodes_z)]
    grid2 = emg3d.TensorMesh([np.diff(x) for x in h2], (h2[0][0], h2[1][0], h2[2][0]))
    plan = models._VolumeAverage(grid, grid2)
    vals = np.asfortranarray(10 ** np.random.default_rng(3).uniform(-1, 1, grid.shape_cells))
    med, _ = timeit(lambda: plan(vals, True), reps=5, warm=1)
    line(f"volume averaging {grid.shape_cells} -> {grid2.shape_cells}, log scale (host in / out)", med, 8 * (nc + grid2.n_cells))


if __name__ == '__main__':
    main()
