"""One default solve (BiCGSTAB + multigrid, the reference's defaults) of the 128^3 marine model --
the command behind profiles/r02_default_solve_kernel_stats.txt:
    rocprofv3 --kernel-trace --stats -d OUT -o run -- python tools/default_solve.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import emg3d_amd as emg3d          # noqa: E402
from bench import workload         # noqa: E402

wl = workload('marine128')
grid = emg3d.TensorMesh(wl['h'], wl['origin'])
model = emg3d.Model(grid, **wl['res'])
sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
for r in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e, info = emg3d.solve(model, sfield, return_info=True, verb=0)        # sslsolver=True, sc + lr, tol 1e-6
    torch.cuda.synchronize()
    print('solve', r, f'{(time.perf_counter() - t0) * 1e3:.1f} ms', info['it_mg'], 'cycles', info['it_ssl'], 'Krylov iterations',
          info['exit_message'])
