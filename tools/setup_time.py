"""Where does the time of a whole solve go besides its cycles? (through gpurun)  One solve of a bench workload on a
fresh hierarchy, split into: source field + VolumeModel on the host, hierarchy construction (eta / zeta on the device),
the cycles one by one (the first ones build coarse levels, line factorisations and capture graphs), download.
    python tools/setup_time.py [workload] [tol]"""
import os, sys, time, cProfile, pstats
root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
import emg3d_amd as emg3d
from emg3d_amd import solver, models, _cycle
from bench import workload

name = sys.argv[1] if len(sys.argv) > 1 else 'salt384'
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
wl = workload(name)
grid = emg3d.TensorMesh(wl['h'], wl['origin'])
model = emg3d.Model(grid, **wl['res'])


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


emg3d.solve(emg3d.Model(emg3d.TensorMesh([np.ones(8)] * 3, (0, 0, 0))), emg3d.get_source_field(emg3d.TensorMesh([np.ones(8)] * 3, (0, 0, 0)), (4., 4., 4., 0, 0), 1.0), sslsolver=False, verb=0)   # library warm-up
for rep in range(2):
    t0 = sync()
    sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    t1 = sync()
    vmodel = models.VolumeModel(model, sfield)
    t2 = sync()
    hier = solver.Hierarchy(vmodel)
    t3 = sync()
    marks = []
    orig = _cycle._one_cycle

    def timed(top, var, it, loud):
        a = sync()
        orig(top, var, it, loud)
        marks.append(sync() - a)
    _cycle._one_cycle = timed
    prof = cProfile.Profile()
    prof.enable()
    e, info = emg3d.solve(model, sfield, sslsolver=False, tol=tol, return_info=True, hierarchy=hier, verb=0, **wl['opts'])
    prof.disable()
    _cycle._one_cycle = orig
    t4 = sync()
    print(f"{name} rep {rep}: source field {1e3 * (t1 - t0):.0f} ms, VolumeModel {1e3 * (t2 - t1):.0f} ms, Hierarchy {1e3 * (t3 - t2):.0f} ms, "
          f"solve {1e3 * (t4 - t3):.0f} ms ({info['it_mg']} cycles: " + ' '.join(f'{1e3 * m:.0f}' for m in marks) +
          f" ms; outside the cycles {1e3 * (t4 - t3 - sum(marks)):.0f} ms)", flush=True)
    if rep == 0:
        pstats.Stats(prof).sort_stats('cumulative').print_stats(28)
    del hier, e
    torch.cuda.empty_cache()
