import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, numpy as np
import emg3d_amd as emg3d
from bench import workload
wl = workload('marine128')
grid = emg3d.TensorMesh(wl['h'], wl['origin'])
model = emg3d.Model(grid, **wl['res'])
sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
opts = dict(wl['opts']); opts.update(sslsolver=False)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, info = emg3d.solve(model, sfield, return_info=True, tol=1e-6, verb=0, **opts)
    torch.cuda.synchronize(); print('run', r, (time.perf_counter()-t0)*1e3, info['it_mg'])
pr = cProfile.Profile(); pr.enable()
e, info = emg3d.solve(model, sfield, return_info=True, tol=1e-6, verb=0, **opts)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
