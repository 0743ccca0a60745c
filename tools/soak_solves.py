"""Soak test (through gpurun): random small solves -- grids, stretching, anisotropy, air layers, frequency and
Laplace domain, multigrid / BiCGSTAB / CGS, cycle types -- against the oracle's converged fields. (This
kind of run found the accuracy floor of the direct-form finest level)
    python tools/soak_solves.py           (SEEDS=23,26 python ... : only these)
Known non-failures of the library: seeds 23 and 26 -- CGS on a model with an air layer does not reach
tol 1e-10 within 80 iterations (identical with the round-2 library; the device CGS follows SciPy's
iteration step by step, tests/test_gpu_parity.py::test_device_krylov_matches_scipy_iteration). With
SEED_BASE=31000 seed 28 (plain V-cycle with semicoarsening on an air-layer model): 80 cycles are not enough in
the four-colour order (the oracle needs 83 in that order, 55 in the reference's: the ordering's price, DESIGN.md
4.1), likewise CGS seeds 31033 / 47026."""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import emg3d_amd as emg3d
from oracle import mg_ref
from helpers import widths
bad = 0
cycles_total = 0
by_solver = {}        # solver -> [solves both converged, GPU cycles, oracle (lexicographic) cycles]
if os.environ.get('LINE_ORDER') or os.environ.get('POINT_ORDER'):      # compare the sweep orders (cycle totals below)
    from emg3d_amd import _lib
    _lib.lib().emg3d_set_option(b'line_order', int(os.environ.get('LINE_ORDER', 1)))
    _lib.lib().emg3d_set_option(b'point_order', int(os.environ.get('POINT_ORDER', 1)))
t0 = time.time()
for seed in [int(x) for x in os.environ.get("SEEDS","").split(",")] if os.environ.get("SEEDS") else range(40):
    rng = np.random.default_rng(int(os.environ.get("SEED_BASE", 11000)) + seed)
    shape = tuple(int(rng.choice([8, 10, 12, 16, 20, 24])) for _ in range(3))
    h = [widths(max(n // 2, 2), (n - max(n // 2, 2)) // 2, 20., float(rng.choice([1.05, 1.15, 1.3]))) for n in shape]
    shape = tuple(len(x) for x in h)
    grid = emg3d.TensorMesh(h, [-w.sum() / 2 for w in h])
    rho = 10 ** rng.uniform(-0.5, 1.0, shape)
    if rng.integers(0, 3) == 0:                     # an "air" layer
        rho[:, :, -max(shape[2] // 4, 1):] = 1e8
    case = int(rng.integers(0, 3))
    props = [(rho,), (rho, None, 2.0 * rho), (rho, 1.5 * rho, 2.5 * rho)][case]
    model = emg3d.Model(grid, *props)
    freq = float(rng.choice([0.8, 0.05, -1.5]))
    sfield = emg3d.get_source_field(grid, (3., -2., -11., float(rng.uniform(0, 90)), float(rng.uniform(-30, 30))), freq)
    ssl = str(rng.choice(['bicgstab', 'cgs', 'False']))
    kw = dict(cycle=str(rng.choice(['V', 'W', 'F'])), semicoarsening=bool(rng.integers(0, 2)), linerelaxation=bool(rng.integers(0, 2)), maxit=80)
    try:
        e, info = emg3d.solve(model, sfield, sslsolver=False if ssl == 'False' else ssl, tol=1e-10, return_info=True, **kw)
        og = mg_ref.Grid(grid.h, grid.origin)
        inv = lambda p: None if p is None else 1 / p
        vm = mg_ref.volume_model(og, freq, *[inv(p) for p in (props + (None, None))[:3]])
        eo, io = mg_ref.solve(vm, mg_ref.Field(og, sfield.field.copy()), tol=1e-10, **dict(kw, maxit=80))
        err = np.linalg.norm(e.field - eo.field) / np.linalg.norm(eo.field)
        air = rho.max() > 1e7
        cycles_total += int(info['it_mg'])
        if info['exit'] == 0 and io['exit'] == 0:
            b = by_solver.setdefault(ssl, [0, 0, 0]); b[0] += 1; b[1] += int(info['it_mg']); b[2] += int(io['it_mg'])
        ok = info['exit'] == 0 and io['exit'] == 0 and err < (1e-5 if air else 1e-8)
        if not ok:
            bad += 1
            print('SEED', seed, shape, 'case', case, 'f', freq, ssl, kw, 'air', air, '| exit', info['exit'], io['exit'], info['exit_message'], 'it', info['it_mg'], info.get('it_ssl'), io['it_mg'], 'err %.1e' % err, flush=True)
    except Exception as exc:
        bad += 1
        print('SEED', seed, shape, ssl, kw, 'EXC', repr(exc)[:300], flush=True)
    if time.time() - t0 > 1100:
        print('time limit at', seed); break
print('converged solves, cycles GPU / lexicographic oracle, by solver:', by_solver)
print('done, failures:', bad, 'multigrid cycles in all solves:', cycles_total, 'seconds %.0f' % (time.time() - t0))
