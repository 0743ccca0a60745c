"""Cycles and seconds of whole solves at full size with the mirrored (line_order = 0) and the cyclic (1) sequence of
the line smoothers' colour passes (through gpurun; round 3: profiles/r03_order_cycles.txt).
    python tools/order_cycles.py"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
import emg3d_amd as emg3d
from emg3d_amd import _lib
from bench import workload
for name in ('triaxial256', 'marine128', 'salt384'):
    wl = workload(name)
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    model = emg3d.Model(grid, **wl['res'])
    sf = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    for order in [int(x) for x in os.environ.get("ORDERS", "0,1").split(",")]:
        _lib.lib().emg3d_set_option(b'line_order', order)
        for tol in (1e-6, 1e-8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, info = emg3d.solve(model, sf, sslsolver=False, tol=tol, return_info=True, **wl['opts'])
            torch.cuda.synchronize()
            print(name, 'line_order', order, 'tol', tol, 'cycles', info['it_mg'], 'seconds %.3f' % (time.perf_counter() - t0), flush=True)
    del model, sf
    torch.cuda.empty_cache()
