"""Soak test (through gpurun): the adjoint-state gradient against central finite differences of the misfit for
every anisotropy case x property mapping, electric and magnetic point receivers mixed (agreement to 6-7
digits: linear receiver interpolation, both solves at tol 1e-10).
    python tools/soak_gradient.py [-v]"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import emg3d_amd as emg3d
from emg3d_amd import gradient
from helpers import widths
bad = 0
t0 = time.time()
maps = {'Conductivity': lambda r: 1 / r, 'Resistivity': lambda r: r, 'LgConductivity': lambda r: np.log10(1 / r),
        'LgResistivity': lambda r: np.log10(r), 'LnConductivity': lambda r: np.log(1 / r), 'LnResistivity': lambda r: np.log(r)}
for seed in range(18):
    rng = np.random.default_rng(15000 + seed)
    hx, hz = widths(4, 3, 50., 1.3), widths(4, 2, 40., 1.3)
    grid = emg3d.TensorMesh([hx, hx, hz], (-hx.sum() / 2, -hx.sum() / 2, -hz[:4].sum()))
    shape = grid.shape_cells
    rho = 10 ** rng.uniform(-0.2, 0.5, shape)
    case = ['isotropic', 'HTI', 'VTI', 'triaxial'][seed % 4]
    mapping = list(maps)[seed % 6]
    fac = {'isotropic': (1, None, None), 'HTI': (1, 1.6, None), 'VTI': (1, None, 2.2), 'triaxial': (1, 1.6, 2.2)}[case]
    m = maps[mapping]
    def model_of(props):
        kw = {k: v for k, v in zip(('property_x', 'property_y', 'property_z'), props) if v is not None}
        return emg3d.Model(grid, mapping=mapping, **kw)
    base = [None if f is None else m(rho * f) for f in fac]
    true = [None if f is None else m(rho * f * (1.3 if i != 1 else 0.8)) for i, f in enumerate(fac)]
    srcs = {'a': (-60., 0., -30., 0., 0.), 'b': (40., 30., -30., 90., 0.)}
    freqs = {'f': float(rng.choice([1.0, 0.3]))}
    recs = np.array([[70., 10., -40., 0., 0.], [-30., -60., -40., 90., 0.], [10., 80., -25., 45., 10.], [-75., 20., -35., 30., 20.]])
    mag = np.array([False, bool(seed % 2), False, True])
    opts = dict(tol=1e-10, sslsolver=True)
    rt = tuple(recs[:, k] for k in range(5))
    tm = model_of(true)
    obs = {}
    for s in srcs:
        ef = emg3d.solve(tm, emg3d.get_source_field(grid, srcs[s], freqs['f']), **opts)
        obs[(s, 'f')] = emg3d.fields.get_responses(ef, None, rt, 'linear', magnetic=mag, efield=ef)
    wts = {k: 1.0 / (0.05 * np.abs(v)) ** 2 for k, v in obs.items()}
    def phi(props):
        return gradient.misfit_and_gradient(model_of(props), srcs, freqs, recs, obs, wts, solver_opts=opts, tol_gradient=1e-10, magnetic=mag)
    m0, g0, _ = phi(base)
    nprop = sum(p is not None for p in base)
    g0 = g0.reshape((nprop,) + shape) if nprop > 1 else g0[None]
    which = [i for i, p in enumerate(base) if p is not None]
    worst = 0.0
    for gi, pi in enumerate(which):
        cand = np.abs(g0[gi]).copy()
        cand[:2], cand[-2:], cand[:, :2], cand[:, -2:], cand[:, :, :2], cand[:, :, -2:] = 0, 0, 0, 0, 0, 0
        cell = np.unravel_index(np.argmax(cand), shape)
        d = 1e-4 * max(abs(base[pi][cell]), 0.1)
        pp, pm = [None if p is None else p.copy() for p in base], [None if p is None else p.copy() for p in base]
        pp[pi][cell] += d; pm[pi][cell] -= d
        fd = (phi(pp)[0] - phi(pm)[0]) / (2 * d)
        nrmsd = 200 * abs(g0[gi][cell] - fd) / (abs(g0[gi][cell]) + abs(fd))
        worst = max(worst, nrmsd)
        if '-v' in sys.argv: print('   prop', pi, 'cell', cell, 'adjoint %.6e  fd %.6e  misfit %.4e' % (g0[gi][cell], fd, m0))
    flag = '' if worst < 2.0 else '   <<<<<'
    if worst >= 2.0: bad += 1
    print('seed', seed, case, mapping, 'f', freqs['f'], 'mag', mag.tolist(), 'worst NRMSD %.2f %%' % worst, flag, flush=True)
    if time.time() - t0 > 1100: break
print('done, failures:', bad, 'seconds %.0f' % (time.time() - t0))
