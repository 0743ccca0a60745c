"""Soak test (through gpurun): a sequence of solves with mixed options (solver, cycle, tolerance, residual form,
source) on ONE reused hierarchy against the same solves on fresh hierarchies: bit-identical fields expected.
    python tools/soak_reuse.py"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import emg3d_amd as emg3d
from emg3d_amd import solver, models
from helpers import widths
rng = np.random.default_rng(5)
shape = (24, 16, 20)
h = [widths(n // 2, n // 4, 20., 1.2) for n in shape]
grid = emg3d.TensorMesh(h, [-w.sum() / 2 for w in h])
rho = 10 ** rng.uniform(-0.5, 1.0, shape)
model = emg3d.Model(grid, rho, 1.5 * rho, 2.0 * rho)
srcs = [(3., -2., 1., 20., 30.), (-40., 10., -20., 90., 0.), (10., 10., 10., 0., 90.)]
sfs = [emg3d.get_source_field(grid, s, 1.2) for s in srcs]
hier = solver.Hierarchy(models.VolumeModel(model, sfs[0]))
bad = 0
for k in range(24):
    i = int(rng.integers(0, 3))
    kw = dict(sslsolver=[False, True, 'cgs', 'gcrotmk'][int(rng.integers(0, 4))] if k % 3 else False,
              cycle=str(rng.choice(['F', 'W', 'V'])), semicoarsening=bool(rng.integers(0, 2)), linerelaxation=bool(rng.integers(0, 2)),
              tol=float(rng.choice([1e-6, 1e-9])), residual_form=[True, False, 'auto'][int(rng.integers(0, 3))], maxit=40)
    sf = sfs[i]
    if kw['sslsolver'] == 'gcrotmk':
        sf = emg3d.Field(grid, sf.field * (100 / np.linalg.norm(sf.field)), frequency=1.2)
    a, ia = emg3d.solve(model, sf, return_info=True, hierarchy=hier, **kw)
    b, ib = emg3d.solve(model, sf, return_info=True, **kw)
    same = np.array_equal(a.field, b.field) and ia['it_mg'] == ib['it_mg'] and ia['exit'] == ib['exit']
    if not same:
        bad += 1
        d = np.linalg.norm(a.field - b.field) / np.linalg.norm(b.field)
        print('step', k, 'source', i, kw, 'NOT identical: rel diff %.1e' % d, 'it', ia['it_mg'], ib['it_mg'], 'exit', ia['exit'], ib['exit'], flush=True)
print('done, differing solves:', bad)
