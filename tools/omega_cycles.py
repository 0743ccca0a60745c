"""Cycles and time to tolerance against `smoother_omega` (extrapolated smoothing calls, solver.solve)
on the bench workloads (through gpurun):
    python tools/omega_cycles.py triaxial64 triaxial256 marine128 uniform128 salt96
One hierarchy per workload (levels, line factors, graphs are shared by all runs); every value of
omega is solved twice, the time is that of the second solve."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import emg3d_amd as emg3d                   # noqa: E402
from emg3d_amd import models, solver        # noqa: E402
from bench import workload                  # noqa: E402

OMEGAS = (1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6)

for name in sys.argv[1:] or ['triaxial64']:
    wl = workload(name)
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    model = emg3d.Model(grid, **{k: np.asfortranarray(v) for k, v in wl['res'].items()})
    sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    hier = solver.Hierarchy(models.VolumeModel(model, sfield))
    row = []
    for w in OMEGAS:
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.time()
            e, info = emg3d.solve(model, sfield, return_info=True, verb=0, tol=1e-8, maxit=60, sslsolver=False,
                                  smoother_omega=w, always_return=True, hierarchy=hier, **wl['opts'])
            torch.cuda.synchronize()
            dt = time.time() - t
        row.append(f"{info['it_mg']:3d} {dt:6.3f}s" if info['exit'] == 0 else f"({info['exit_message'].split()[0].lower():>9s})")
    print(f"{name:12s} " + "  ".join(f"{w}: {r}" for w, r in zip(OMEGAS, row)), flush=True)
