"""Converged parity at FULL size (through gpurun): BASELINE.json configs 3 / 2 / 5 exactly as bench.py builds them,
solved to tol 1e-10 on the GPU and by the oracle's multigrid driver in the same smoother ordering (order 1, its
independent classes walked by threads: bit-identical with the serial walk), rel-L2 of the converged fields and the
cycle counts. The reference order (sequential) is not affordable at 256^3 (~30 min per cycle); its converged
field is the same fixed point -- compared on the reduced copies in tests/test_gpu_parity.py.
    python tools/full_size_converged.py [triaxial256 marine128 salt384]     (profiles/r03_full_size_converged.txt)"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import torch
import emg3d_amd as emg3d
from oracle import core as ocore, mg_ref
from helpers import relerr, usable_cores
from bench import workload

names = sys.argv[1:] or ['marine128', 'triaxial256', 'salt384']
tol = float(os.environ.get('TOL', 1e-10))
nt = usable_cores()
ocore.lib().oracle_set_threads(nt)
print('oracle threads', nt, 'tol', tol, flush=True)
for name in names:
    wl = workload(name)
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    model = emg3d.Model(grid, **wl['res'])
    sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, info = emg3d.solve(model, sfield, sslsolver=False, tol=tol, return_info=True, **wl['opts'])
    torch.cuda.synchronize(); tg = time.perf_counter() - t0
    ogrid = mg_ref.Grid(grid.h, grid.origin)
    cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in wl['res'].items()}
    vm = mg_ref.volume_model(ogrid, wl['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
    t0 = time.perf_counter()
    eo, io = mg_ref.solve(vm, mg_ref.Field(ogrid, sfield.field.copy()), tol=tol, order=1, **wl['opts'])
    to = time.perf_counter() - t0
    print(name, grid.shape_cells, wl['opts'], '| GPU exit', info['exit'], 'cycles', info['it_mg'], 'rel. error %.2e' % info['rel_error'],
          '%.2f s' % tg, '| oracle (same order) exit', io['exit'], 'cycles', io['it_mg'], 'rel. error %.2e' % io['rel_error'],
          '%.0f s' % to, '| rel-L2 of the fields %.2e' % relerr(e.field, eo.field), flush=True)
    del model, sfield, e, eo, vm
    torch.cuda.empty_cache()
