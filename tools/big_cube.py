"""One W-cycle of the tri-axial workload at 512^3 (or any n) with the factor-memory policy 'rebuild', through gpurun:
memory, time per cycle, error reduction -- and, with ORACLE=1, the same cycle by the oracle's driver in the same
ordering (threaded classes) with the rel-L2 of the fields.
    python tools/big_cube.py [n] [policy]        (profiles/r04_big_cube.txt)"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import torch
import emg3d_amd as emg3d
from emg3d_amd import solver
from bench import workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
policy = sys.argv[2] if len(sys.argv) > 2 else 'rebuild'
cycles = int(os.environ.get('CYCLES', 3))
wl = workload(f'triaxial{n}')
grid = emg3d.TensorMesh(wl['h'], wl['origin'])
model = emg3d.Model(grid, **wl['res'])
sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
vmodel = emg3d.models.VolumeModel(model, sfield)
t0 = time.perf_counter()
hier = solver.Hierarchy(vmodel, line_factors=policy)
for it in (1, cycles):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, info = emg3d.solve(model, sfield, sslsolver=False, tol=1e-30, maxit=it, return_info=True, hierarchy=hier,
                          _download=(it == 1), **wl['opts'])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{grid.shape_cells} policy {policy}: {it} cycle(s) {dt:.2f} s, rel. error {info['rel_error']:.3e}, "
          f"errors {np.array2string(info['error_at_cycle'][:it + 1] / info['ref_error'], precision=3)}, "
          f"HBM allocated (max) {torch.cuda.max_memory_allocated() / 1e9:.1f} GB, "
          f"{torch.cuda.max_memory_allocated() / grid.n_cells:.0f} B per cell, rebuilds on the finest level "
          f"{getattr(hier.top, 'factor_rebuilds', 0)}", flush=True)
    if it == 1:
        field1 = e.field.copy()
if os.environ.get('ORACLE'):
    from oracle import core as ocore, mg_ref
    from helpers import relerr, usable_cores
    ocore.lib().oracle_set_threads(usable_cores())
    del hier
    torch.cuda.empty_cache()
    ogrid = mg_ref.Grid(grid.h, grid.origin)
    cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in wl['res'].items()}
    vm = mg_ref.volume_model(ogrid, wl['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
    t0 = time.perf_counter()
    eo, io = mg_ref.solve(vm, mg_ref.Field(ogrid, sfield.field.copy()), tol=1e-30, maxit=1, order=1, **wl['opts'])
    print(f"oracle, same ordering, {usable_cores()} threads: {time.perf_counter() - t0:.0f} s, rel-L2 of the fields after "
          f"one cycle {relerr(field1, eo.field):.2e}, abs. error {info['error_at_cycle'][1]:.6e} / {io['abs_error']:.6e}", flush=True)
